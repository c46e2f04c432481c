#!/usr/bin/env python3
"""Where a short-sequence convolution (conv1d_short_kernel, the T_w ~ 27 word encoder) spends its time: per-workgroup cycle stamps at
kernel start / tile staged / contraction done / stores issued, for the four shapes of one encoder FFT block (FFN conv_1, QKV, FFN
conv_2, O) — what `rocprofv3 --kernel-trace` cannot split.  Needs the stamped build of the library:
    make -C dict_tts_amd/csrc prof          # build/libdicttts_hip_prof.so (conv1d.hip with -DC1D_PROF=1)
    python tools/c1d_phase_prof.py
(DTTS_C1D_KS=1|2|4 forces the contraction split for an A/B.)"""
import ctypes, os, sys, shutil
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import numpy as np
import torch
from dict_tts_amd import abi, model, synth
lib = abi.load_library(os.path.join(R, "build", "libdicttts_hip_prof.so"))   # before anything else binds the plain build
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
sd = synth.dict_tts_state_dict(1234)
sd["dur_predictor.linear.0.bias"] = np.array([3.09], np.float32)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
st = synth.biaobei_struct()
table = synth.dict_table(1234)
m.upload_dict_table(table)
B = 60
ib = synth.make_id_batch(st["sentences"][:B], table)
dev = torch.device("cuda")
d = {k: T(ib[k]).to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
ptr = lambda t: t.data_ptr()
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, B, ib["word_tokens"].shape[1], ib["L_k"], ib["P"], s)
torch.cuda.synchronize()
buf = np.zeros((4, 2048, 8), np.uint64)
rc = lib.dtts_debug_c1d_prof(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0
names = ["FFN1 k5 192->768", "QKV k1 192->576", "FFN2 k1 768->192", "O k1 192->192"]
for i, nm in enumerate(names):
    a = buf[i].astype(np.int64)
    n = int((a[:, 5] > 0).sum())
    if n == 0:
        continue
    a = a[:n]
    live = a[:, 5] == 2
    al = a[live]
    rt0, rt1 = al[:, 6], al[:, 7]
    span = (a[:, 7].max() - a[:, 6].min()) / 100.0      # 100 MHz
    cyc = al[:, 3] - al[:, 0]
    us_per_cyc = ((rt1 - rt0) / 100.0).sum() / max(cyc.sum(), 1)
    ph = [(al[:, k + 1] - al[:, k]).mean() * us_per_cyc for k in range(3)]
    print(f"{nm}: {n} WGs ({int(live.sum())} live), kernel span {span:.1f} us, first->last WG start {(a[:,6].max()-a[:,6].min())/100.0:.1f} us, "
          f"per-WG: stage {ph[0]:.2f} us  loop {ph[1]:.2f} us  epilogue {ph[2]:.2f} us  total {sum(ph):.2f} us (max {(rt1-rt0).max()/100.0:.1f}); clk {1/us_per_cyc:.0f} MHz")
