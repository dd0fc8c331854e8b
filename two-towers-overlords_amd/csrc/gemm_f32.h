// Exact-fp32 MFMA GEMM building block (gfx950, v_mfma_f32_32x32x2_f32) used by the
// projection forward/backward (K2).  Internal to the library — not part of the C ABI.
#pragma once
#include "tt_common.h"

namespace ttg {

enum Layout : int {
  kNT = 0,  // C[m,n] = sum_k A[m,k] * B[n,k]   (A [M,K], B [N,K]; both K-contiguous)  X W^T
  kNN = 1,  // C[m,n] = sum_k A[m,k] * B[k,n]   (A [M,K], B [K,N])                       dOut W2
  kTN = 2   // C[m,n] = sum_k A[k,m] * B[k,n]   (A [K,M], B [K,N]; contraction over rows) dOut^T H
};

enum Epilogue : int {
  kStore = 0,     // C = acc
  kBias = 1,      // C = acc + bias[n]
  kBiasRelu = 2,  // C = max(acc + bias[n], 0)
  kReluMask = 3,  // C = aux[m,n] > 0 ? acc : 0
  kHinge = 4      // in-batch-negatives loss (inbatch.hip): acc = S[m,n] = cos(q_m, d_n); C = dL/dS[m,n], see HingeEpi
};

// kHinge: C[m,n] = g where n is a valid negative of query m and S[m,n] - s_pos[m] + margin > 0, else 0; every
// workgroup also leaves its rows' hinge sums and active counts in part_loss / part_cnt [M, n_tiles] (column-tile
// major per row), which a finishing pass adds up in tile order (fixed order: bitwise reproducible).
struct HingeEpi {
  const int* pos_index;        // [M] column of the positive (null: the diagonal)
  const long long* qid_q;      // [M] query ids (null: none)
  const long long* qid_d;      // [N]
  const float* s_pos;          // [M] S[m, pos_m]
  float margin, g;
  float* part_loss;
  float* part_cnt;
  int n_tiles;                 // column tiles of the launch (row stride of the partials)
};

struct Problem {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // kBias / kBiasRelu: [N]
  const float* aux;   // kReluMask: [M,N], leading dim ldaux
  float* colsum;      // kTN only (may be null): colsum[m] = sum_k A[k,m], written per split like C
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int k_per_split;  // kTN split-K: split z covers k in [z*k_per_split, (z+1)*k_per_split); multiple of 32
};

constexpr int kMaxGroup = 4;

struct Batch {
  Problem p[kMaxGroup];
  int count;            // problems in this launch (blockIdx.z / nsplit selects)
  int nsplit;           // split-K factor (kTN, else 1); split z writes C + z*slab_stride
  long long slab_stride;  // floats between consecutive split slabs of C / colsum
  // XCD-aware workgroup order (1-D grid; workgroups with equal index % 8 share an XCD — a placement label, speed
  // only): tiles that re-read the same operand block run on ONE XCD, back to back, so the block crosses the fabric
  // once instead of once per tile.  kNT / kNN: the n-tiles of a row panel; kTN: all tiles of one split-K slab.
  // Worth it where the fabric is the shared bottleneck (beside the streaming gather); 0 = plain 3-D grid.
  HingeEpi hinge;  // kHinge only
  int xcd_order;
  int gx, gy, gz;         // tile grid (filled by the launcher when xcd_order is set)
};

// Static LDS of one workgroup of the (layout, wm, wn) kernel the launcher would pick, in bytes.
size_t tile_lds_bytes(Layout layout, int wm, int wn, int bk = 32);

// Launches one grouped GEMM.  wm/wn pick the block tile: (64*wm) x (64*wn).
// bk: K-step, 32 or 16 (16 only changes the 128x128 tiles: 40 KB of LDS instead of 72).
int launch(Layout layout, Epilogue epi, const Batch& b, int wm, int wn, hipStream_t st, int bk = 32);

// One launch for an independent (kTN split-K, kStore) + (kNN, kReluMask) pair; falls back to two
// launches for shapes the vector path does not cover.
int launch_tn_nn_pair(const Batch& tn, const Batch& nn, int wm, int wn, hipStream_t st);

}  // namespace ttg
