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

// experiment (ablation builds): the persistent ResBlock grids leave DTTS_CU_RESERVE compute units' worth of workgroup slots free, so that
// the other stream's text->mel kernels find a slot at any time instead of only at the vocoder's launch boundaries
inline int cu_reserve() {
    static const int v = [] { const char* e = ablate_env("DTTS_CU_RESERVE"); return e ? atoi(e) : 0; }();
    return v;
}

// dtts_config.tune_flags (include/dicttts_hip.h).  The RELEASE library honours only the bits that have a parity / bit-identity test behind
// them (tests/test_gpu_parity.py): 8 prior flow launch by launch on the exact-fp32 kernels, 9 all ResBlocks of a C <= 64 stage in one launch,
// 12 the first two ResBlocks of the C = 32 stage in one launch, 13 two-product fp16 ups.1, 14 512-row tiles for every k at C = 64, 15 the fp32
// inter-iteration stream of the per-iteration ResBlock kernels (round 5's form; default since round 6: fp16).  Every
// other bit is an untested tuning experiment: its code exists only in -DDTTS_ABLATE builds (`make ablate`), the branches fold away in the
// release library, and dtts_create REJECTS such a bit there (DTTS_E_INVAL) instead of ignoring it.
constexpr int TUNE_RELEASE_MASK = (1 << 8) | (1 << 9) | (1 << 12) | (1 << 13) | (1 << 14) | (1 << 15);
#ifdef DTTS_ABLATE
constexpr int TUNE_MASK = ~0;
#else
constexpr int TUNE_MASK = TUNE_RELEASE_MASK;
#endif
#define DTTS_TUNE(h, bit) ((((bit) & dtts::TUNE_MASK) != 0) && ((h)->tune & (bit)) != 0)
} // namespace dtts
