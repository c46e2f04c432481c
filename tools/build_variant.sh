#!/bin/bash
# build a variant of the library for same-box A/Bs: tools/build_variant.sh <name> "<-D flags>" file1.hip [file2.hip ...]
#   compiles the listed sources with the extra flags, links them with the release objects of the others -> build/x/<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/dict_tts_amd/csrc; X=$R/build/x; mkdir -p $X
NAME=$1; FLAGS=$2; shift 2
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed"
ALL="conv1d vconv rblock vpair flowstack ops context"
[ -f $C/rblock2.o ] && grep -q rblock2 $C/Makefile && ALL="$ALL rblock2"
OBJS=""
for s in $ALL; do
  if [[ " $* " == *" $s.hip "* ]]; then
    (cd $C && /opt/rocm/bin/hipcc $F $FLAGS -Rpass-analysis=kernel-resource-usage -c $s.hip -o $X/${NAME}_$s.o 2> $X/${NAME}_$s.log) || { grep -v "remark:" $X/${NAME}_$s.log | head -30; exit 1; }
    OBJS="$OBJS $X/${NAME}_$s.o"
  else
    OBJS="$OBJS $C/$s.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $OBJS -o $X/$NAME.so
ls -la $X/$NAME.so
