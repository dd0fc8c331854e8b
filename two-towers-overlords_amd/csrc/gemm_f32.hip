// Exact-fp32 MFMA GEMM for the projection (K2) — gfx950, v_mfma_f32_32x32x2_f32.
//
// Tiling: 256 threads = 4 waves arranged 2x2; each wave owns WM x WN tiles of
// 32x32 (16 accumulator registers each), so the block tile is (64*WM) x (64*WN)
// over K-steps of 32.  Operands are staged global -> registers -> LDS with the
// next tile's global loads issued before the current tile's MFMAs (two LDS
// buffers, one barrier per K-step).
//
// The MFMA sums over two k-slots per instruction (lane half h = lane>>5), and
// any assignment of k to (instruction, half) is legal as long as A and B agree.
// We use k = 8j + 4h + s for instruction s of group j, so a K-contiguous
// operand ([rows][K] in memory: X, W, dOut-as-A in NN) is read from its
// [rows][32+4] LDS image with ONE ds_read_b128 per four MFMAs (the +4 pad makes
// the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots); a
// K-strided operand ([K][cols] in memory: W2 in NN, dOut/H/X in TN) sits in a
// [32][cols] image and is read with conflict-free ds_read_b32 (lanes =
// consecutive columns).
//
// Numerics: each C element is a k-ordered fmaf chain in fp32 (what the MFMA
// computes, bitwise), K-steps in ascending order; split-K (kTN) partials go to
// separate slabs that are summed in fixed order later => run-to-run bitwise
// reproducible, no float atomics.
#include "gemm_f32.h"

namespace ttg {
namespace {

constexpr int BK = 32;
constexpr int BKP = BK + 4;
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- global -> register staging ------------------------------------------------
// K-contiguous operand: tile [R rows][BK], element (row, k) at P[(r0+row)*ld + k0+k].
template <int R>
__device__ __forceinline__ void gload_kc(float4 (&v)[R * BK / 4 / kThreads], const float* __restrict__ P,
                                         int ld, int r0, int rmax, int k0, int kend, bool vec, int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int row = f >> 3, kq = f & 7;
    const int gr = r0 + row, gk = k0 + kq * 4;
    const float* __restrict__ src = P + (long long)gr * ld + gk;
    if (gr < rmax && gk + 3 < kend && vec) {
      v[i] = *reinterpret_cast<const float4*>(src);
    } else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rmax) {
        if (gk + 0 < kend) t.x = src[0];
        if (gk + 1 < kend) t.y = src[1];
        if (gk + 2 < kend) t.z = src[2];
        if (gk + 3 < kend) t.w = src[3];
      }
      v[i] = t;
    }
  }
}
template <int R>
__device__ __forceinline__ void sstore_kc(const float4 (&v)[R * BK / 4 / kThreads], float* __restrict__ T, int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int row = f >> 3, kq = f & 7;
    *reinterpret_cast<float4*>(T + row * BKP + kq * 4) = v[i];
  }
}
// K-strided operand: tile [BK][R cols], element (k, col) at P[(k0+k)*ld + r0+col].
template <int R>
__device__ __forceinline__ void gload_ks(float4 (&v)[R * BK / 4 / kThreads], const float* __restrict__ P,
                                         int ld, int r0, int rmax, int k0, int kend, bool vec, int tid) {
  constexpr int RQ = R / 4;
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int krow = f / RQ, rq = f % RQ;
    const int gk = k0 + krow, gr = r0 + rq * 4;
    const float* __restrict__ src = P + (long long)gk * ld + gr;
    if (gk < kend && gr + 3 < rmax && vec) {
      v[i] = *reinterpret_cast<const float4*>(src);
    } else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < kend) {
        if (gr + 0 < rmax) t.x = src[0];
        if (gr + 1 < rmax) t.y = src[1];
        if (gr + 2 < rmax) t.z = src[2];
        if (gr + 3 < rmax) t.w = src[3];
      }
      v[i] = t;
    }
  }
}
template <int R>
__device__ __forceinline__ void sstore_ks(const float4 (&v)[R * BK / 4 / kThreads], float* __restrict__ T, int tid) {
  constexpr int RQ = R / 4;
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int krow = f / RQ, rq = f % RQ;
    *reinterpret_cast<float4*>(T + krow * R + rq * 4) = v[i];
  }
}

// ---- LDS -> MFMA fragments: 4 values for k = 8j + 4h + {0,1,2,3} ------------------
__device__ __forceinline__ float4 frag_kc(const float* __restrict__ T, int row, int j, int h) {
  return *reinterpret_cast<const float4*>(T + row * BKP + 8 * j + 4 * h);
}
template <int R>
__device__ __forceinline__ float4 frag_ks(const float* __restrict__ T, int col, int j, int h) {
  const float* __restrict__ p = T + (8 * j + 4 * h) * R + col;
  return make_float4(p[0], p[R], p[2 * R], p[3 * R]);
}

template <int LAYOUT, int EPI, int WM, int WN>
__global__ __launch_bounds__(kThreads) void gemm_f32_kernel(const Batch b) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr bool A_KC = (LAYOUT != kTN);
  constexpr bool B_KC = (LAYOUT == kNT);
  constexpr int A_TILE = A_KC ? BM * BKP : BK * BM;
  constexpr int B_TILE = B_KC ? BN * BKP : BK * BN;
  constexpr int NVA = BM * BK / 4 / kThreads, NVB = BN * BK / 4 / kThreads;

  __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];

  const int grp = blockIdx.z / b.nsplit;
  const int split = blockIdx.z - grp * b.nsplit;
  const Problem& pr = b.p[grp];
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= pr.M || n0 >= pr.N) return;  // grouped launch: grid is sized for the largest problem

  const int kbeg = (LAYOUT == kTN) ? min(pr.K, split * pr.k_per_split) : 0;
  const int kend = (LAYOUT == kTN) ? min(pr.K, kbeg + pr.k_per_split) : pr.K;
  const long long slab = (long long)split * b.slab_stride;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wrow = wave >> 1, wcol = wave & 1;

  const bool vecA = ((pr.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.A) & 15) == 0);
  const bool vecB = ((pr.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.B) & 15) == 0);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[NVA], rb[NVB];
  float csum = 0.f;  // kTN column sums of A (bias gradient), threads tid < BM of n-tile 0
  const bool do_colsum = (LAYOUT == kTN) && pr.colsum != nullptr && blockIdx.x == 0;

  auto gload = [&](int k0) {
    if (A_KC) gload_kc<BM>(ra, pr.A, pr.lda, m0, pr.M, k0, kend, vecA, tid);
    else      gload_ks<BM>(ra, pr.A, pr.lda, m0, pr.M, k0, kend, vecA, tid);
    if (B_KC) gload_kc<BN>(rb, pr.B, pr.ldb, n0, pr.N, k0, kend, vecB, tid);
    else      gload_ks<BN>(rb, pr.B, pr.ldb, n0, pr.N, k0, kend, vecB, tid);
  };
  auto sstore = [&](int buf) {
    float* As = lds + buf * (A_TILE + B_TILE);
    float* Bs = As + A_TILE;
    if (A_KC) sstore_kc<BM>(ra, As, tid); else sstore_ks<BM>(ra, As, tid);
    if (B_KC) sstore_kc<BN>(rb, Bs, tid); else sstore_ks<BN>(rb, Bs, tid);
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    gload(kbeg);
    sstore(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const float* __restrict__ As = lds + cur * (A_TILE + B_TILE);
    const float* __restrict__ Bs = As + A_TILE;
    if (kt + 1 < nk) gload(kbeg + (kt + 1) * BK);

#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = (wrow * WM + i) * 32 + r;
        fa[i] = A_KC ? frag_kc(As, row, j, h) : frag_ks<BM>(As, row, j, h);
      }
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        const int col = (wcol * WN + i) * 32 + r;
        fb[i] = B_KC ? frag_kc(Bs, col, j, h) : frag_ks<BN>(Bs, col, j, h);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jj = 0; jj < WN; ++jj) {
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[jj].x, acc[i][jj], 0, 0, 0);
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[jj].y, acc[i][jj], 0, 0, 0);
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[jj].z, acc[i][jj], 0, 0, 0);
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[jj].w, acc[i][jj], 0, 0, 0);
        }
    }
    if (do_colsum && tid < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) csum += As[k * BM + tid];
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*h
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int jj = 0; jj < WN; ++jj) {
      const int n = n0 + (wcol * WN + jj) * 32 + r;
      if (n >= pr.N) continue;
      float bias = 0.f;
      if (EPI == kBias || EPI == kBiasRelu) bias = pr.bias[n];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wrow * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= pr.M) continue;
        float v = acc[i][jj][e];
        if (EPI == kBias) v += bias;
        if (EPI == kBiasRelu) v = fmaxf(v + bias, 0.f);
        if (EPI == kReluMask) v = pr.aux[(long long)m * pr.ldaux + n] > 0.f ? v : 0.f;
        pr.C[slab + (long long)m * pr.ldc + n] = v;
      }
    }
  if (do_colsum && tid < BM && m0 + tid < pr.M) pr.colsum[slab + m0 + tid] = csum;
}

template <int LAYOUT, int EPI>
int launch_le(const Batch& b, int wm, int wn, hipStream_t st) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < b.count; ++i) {
    maxM = b.p[i].M > maxM ? b.p[i].M : maxM;
    maxN = b.p[i].N > maxN ? b.p[i].N : maxN;
  }
  if (maxM == 0 || maxN == 0) return TT_OK;
  const int BM = 64 * wm, BN = 64 * wn;
  const dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, b.count * b.nsplit), block(kThreads);
  if (wm == 1 && wn == 1) hipLaunchKernelGGL((gemm_f32_kernel<LAYOUT, EPI, 1, 1>), grid, block, 0, st, b);
  else if (wm == 2 && wn == 1) hipLaunchKernelGGL((gemm_f32_kernel<LAYOUT, EPI, 2, 1>), grid, block, 0, st, b);
  else if (wm == 2 && wn == 2) hipLaunchKernelGGL((gemm_f32_kernel<LAYOUT, EPI, 2, 2>), grid, block, 0, st, b);
  else return TT_ERR_UNSUPPORTED;
  TT_CHECK_LAUNCH();
  return TT_OK;
}

}  // namespace

int launch(Layout layout, Epilogue epi, const Batch& b, int wm, int wn, hipStream_t st) {
  if (b.count < 1 || b.count > kMaxGroup || b.nsplit < 1) return TT_ERR_BAD_SHAPE;
  if (layout == kNT && epi == kBiasRelu) return launch_le<kNT, kBiasRelu>(b, wm, wn, st);
  if (layout == kNT && epi == kBias) return launch_le<kNT, kBias>(b, wm, wn, st);
  if (layout == kNT && epi == kStore) return launch_le<kNT, kStore>(b, wm, wn, st);
  if (layout == kNN && epi == kReluMask) return launch_le<kNN, kReluMask>(b, wm, wn, st);
  if (layout == kNN && epi == kStore) return launch_le<kNN, kStore>(b, wm, wn, st);
  if (layout == kTN && epi == kStore) return launch_le<kTN, kStore>(b, wm, wn, st);
  return TT_ERR_UNSUPPORTED;
}

}  // namespace ttg
