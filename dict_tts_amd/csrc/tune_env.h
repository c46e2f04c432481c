// A/B switches read from the process environment exist ONLY in -DDTTS_ABLATE builds (`make ablate` -> libdicttts_abl.so, used by
// tools/ab_*.sh / abl_voc.sh).  The release library never reads the environment: its arithmetic, layout and schedule follow
// dtts_config alone (tune_flags = 0: the measured defaults).
#pragma once
#include <cstdlib>

namespace dtts {
inline const char* ablate_env(const char* name) {
#ifdef DTTS_ABLATE
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
} // namespace dtts
