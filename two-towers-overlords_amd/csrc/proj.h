// Projection (K2) forward/backward as grouped MFMA GEMM launches.  Internal.
#pragma once
#include "gemm_f32.h"

namespace ttp {

// One tower's slice of a grouped projection call.
struct Tower {
  const float* X;   // [M, D] pooled-normalised inputs
  const float* W1;  // [P, D]
  const float* b1;  // [P]
  const float* W2;  // [P, P]
  const float* b2;  // [P]
  float* H;         // [M, P] relu(X W1^T + b1)
  float* Out;       // [M, P]
  const float* dOut;  // [M, P]          (backward)
  float* dH;          // [M, P] scratch  (backward)
  float* gW1;         // slab 0 of dW1 [P, D]; slab z at + z*slab_stride
  float* gb1;
  float* gW2;
  float* gb2;
  int M;
};

// Block-tile policy of the [rows x P] projection GEMMs: stand-alone choice; 128x64 where 64x64 would do (half the
// weight re-reads: beside a capped K1); 64x64 below the 128x128 threshold (beside a K1 reserved just above that
// tile's LDS, engine.hip).
// kTileStream: beside the streaming gather of an HBM-resident table (gather_pool.hip): 128x128 tiles with K-step 16.
enum TilePolicy : int { kTileAuto = 0, kTileLowTraffic = 1, kTileSmall = 2, kTileStream = 3 };
// How a grouped projection call is tiled: the policy that pairs it with whatever shares the chip (the engine's
// business) and the caller's explicit choices (tt_tuning, 0 = auto), which win over the policy.
struct Plan {
  int policy = kTileAuto;
  bool beside_gather = false;  // token-id gathers of another stream are sharing the chip with this call (engine.hip)
  tt_tuning t{};
};
inline Plan plan_of(const tt_tuning* t, int policy = kTileAuto) {
  Plan p;
  p.policy = policy;
  if (t) p.t = *t;
  return p;
}
// k_per_split for `M` rows over `nsplit` slabs (multiple of 32, >= 32)
int k_per_split(int M, int nsplit);
// split count used for a standalone backward over M rows
int choose_nsplit(long long M);
// split count for the engine's two towers (B query rows, 2B document rows) at projection width P over inputs of width D
int choose_nsplit_towers(long long B, int P, int D);
// Largest static LDS (bytes) of a workgroup of the projection kernels forward()/backward_*() launch for
// `rows` rows in total with the given plan — what a gather workgroup's LDS reservation is sized against.
size_t chain_lds_bytes(long long rows, int P, const Plan& plan);

int forward(const Tower* t, int count, int D, int P, hipStream_t st, const Plan& plan = Plan{});
// writes nsplit partial slabs of (dW1, db1, dW2, db2) per tower; dH is scratch
int backward(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st,
             const Plan& plan = Plan{});
// The three launches of `backward`, separately (dW2/db2 is independent of the dH -> dW1/db1
// chain, so a caller may run it on another stream):
int backward_dw2(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st,
                 const Plan& plan = Plan{});
int backward_dh(const Tower* t, int count, int D, int P, hipStream_t st, const Plan& plan = Plan{});
// dW2/db2 and dH as ONE launch (they are independent of each other)
int backward_dw2_dh(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st,
                    const Plan& plan = Plan{});
int backward_dw1(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st,
                 const Plan& plan = Plan{});

}  // namespace ttp
