// Shared device/host helpers for the gfx950 two-tower hot path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tt_hotpath.h"

#define TT_WAVE 64

#define TT_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

// Launch-error check without synchronising (launch errors surface immediately,
// execution errors at the caller's next sync).
#define TT_CHECK_LAUNCH()                    \
  do {                                       \
    hipError_t _e = hipGetLastError();       \
    if (_e != hipSuccess) return (int)_e;    \
  } while (0)

#define TT_HIP(call)                         \
  do {                                       \
    hipError_t _e = (call);                  \
    if (_e != hipSuccess) return (int)_e;    \
  } while (0)

// LDS bytes a gather workgroup must occupy so that at most `blocks_per_cu` of them fit the CU's 160 KB
// (gather_pool.hip, "Sharing the chip").
static inline unsigned tt_k1_lds_for_cap(int blocks_per_cu) {
  if (blocks_per_cu <= 0 || blocks_per_cu >= 16) return 0;
  return (160u * 1024u) / (unsigned)blocks_per_cu - 1024u;  // one granule of slack: n fit, n + 1 do not
}

// hipEventCreateWithFlags flags for tt_tuning.event_scope / tt_event_create's `scope`
static inline unsigned tt_event_flags(int scope) {
  return hipEventDisableTiming | (scope == 1 ? hipEventDisableSystemFence : scope == 2 ? hipEventReleaseToDevice : 0u);
}

// the id that marks padding inside an ids array (gather_pool.hip, tt_is_pad)
#ifndef TT_PAD_ID32
#define TT_PAD_ID32 (-2147483647LL - 1LL)
#endif

static inline bool tt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef __HIPCC__
// Butterfly sum over the 64 lanes of a wave; every lane ends with the total.
// Fixed order => bitwise reproducible.
__device__ __forceinline__ float tt_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, TT_WAVE);
  return v;
}
#endif
