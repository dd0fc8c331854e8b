// 96x96-tile exact-fp32 MFMA GEMM on v_mfma_f32_16x16x4_f32 (gfx950) for the projection's [rows x P] products.
//
// Why a second tile shape.  The 32x32x2 kernel (gemm_f32.hip) builds its block tile from four 32x32 wave tiles, so
// the finest split of a [6144 x 384] output (BASELINE.json configs[1]: three encodes of 2048 rows, P = 384) is 576
// tiles of 64x64 on 256 CUs: 2.25 per CU, paid as 3.  A 96x96 tile is 6 x 6 accumulator blocks of 16x16: twelve waves
// (three per SIMD) own a 16x48 strip each (three accumulators, one A fragment reused over three B fragments), and a
// single [6144 x 384] matrix would be exactly 256 of them.  Same arithmetic as the other kernel: each output is a
// k-ordered fp32 fmaf chain (what the MFMA computes, bitwise), K-steps in ascending order.
// STATUS: opt-in (TT_G16=1).  The step's product is two towers of 2048 + 4096 rows (260 tiles, four CUs get two) and
// measured slower than the 64x64 tiles in every mode (profiles/r03_gemm16_ab.txt).
//
// Work per K-step of 32 and workgroup: every thread stages ONE 16-byte chunk of A and one of B (768 threads = the
// 96 x 32 / 4 chunks of a tile), two K-steps ahead of their use (two register sets, two LDS buffers, one barrier per
// step).  Lane l of a wave holds A[i = l & 15][k-slot q = l >> 4]; MFMA s of a 16-wide K chunk uses k = 4q + s for
// slot q, so the four values a lane needs are contiguous and come from ONE ds_read_b128.  LDS image of a
// K-contiguous operand: [chunk][row][16 floats] with the four 16-byte pieces of a row stored at piece ^ (row & 8 ?
// 3 : 0): every 16-lane group of the ds_read_b128 then touches 16 distinct 16-byte bank slots (derivation in
// DESIGN.md §3).  The K-strided operand of kNN (W2 read as [K][N]) stays in a [32][96 + 4] image and is read with
// ds_read_b32: lanes i = 0..15 walk consecutive words, k-slots q sit 4 rows = 400 words = 16 banks apart.
#include <cstdlib>
#include "gemm_f32.h"

namespace ttg {
namespace {

constexpr int T = 96;           // tile edge
constexpr int BK = 32;          // K-step
constexpr int kThreads16 = 768; // 12 waves
constexpr int KS_LD = T + 4;    // row of the K-strided image
constexpr int CP = T + 4;       // padded row of the epilogue's C tile
constexpr int OPERAND = 2 * T * 16;          // floats of a K-contiguous operand tile: [2 chunks][96][16]
constexpr int OPERAND_KS = BK * KS_LD;       // floats of a K-strided operand tile
constexpr int BUF = OPERAND + (OPERAND > OPERAND_KS ? OPERAND : OPERAND_KS);
constexpr int LDS_FLOATS16 = (2 * BUF > T * CP) ? 2 * BUF : T * CP;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int row) { return (row & 8) ? 3 : 0; }

// global -> registers: the chunk (row = t / 8, k-quad = t % 8) of a K-contiguous tile [96][32]
__device__ __forceinline__ float4 load_kc(const float* __restrict__ P, int ld, int r0, int rmax, int k0, int kend, int t,
                                          bool full) {
  const int row = t >> 3, kq = t & 7;
  const int gr = r0 + row, gk = k0 + kq * 4;
  if (full) return *reinterpret_cast<const float4*>(P + (long long)gr * ld + gk);
  const bool ok = gr < rmax && gk < kend;  // K % 4 == 0: a chunk is all in or all out
  const int cr = min(gr, rmax - 1), ck = min(gk, kend - 4);
  const float4 v = *reinterpret_cast<const float4*>(P + (long long)cr * ld + ck);
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void store_kc(float* __restrict__ img, const float4& v, int t) {
  const int row = t >> 3, kq = t & 7;
  *reinterpret_cast<float4*>(img + (((kq >> 2) * T + row) << 4) + 4 * ((kq & 3) ^ swz(row))) = v;
}
// global -> registers: the chunk (k = t / 24, column quad = t % 24) of a K-strided tile [32][96]
__device__ __forceinline__ float4 load_ks(const float* __restrict__ P, int ld, int c0, int cmax, int k0, int kend, int t,
                                          bool full) {
  const int krow = t / 24, cq = t - krow * 24;
  const int gk = k0 + krow, gc = c0 + cq * 4;
  if (full) return *reinterpret_cast<const float4*>(P + (long long)gk * ld + gc);
  const bool ok = gk < kend && gc < cmax;  // N % 4 == 0
  const int ck = min(gk, kend - 1), cc = min(gc, cmax - 4);
  const float4 v = *reinterpret_cast<const float4*>(P + (long long)ck * ld + cc);
  return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void store_ks(float* __restrict__ img, const float4& v, int t) {
  const int krow = t / 24, cq = t - krow * 24;
  *reinterpret_cast<float4*>(img + krow * KS_LD + cq * 4) = v;
}

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(kThreads16) void gemm16_kernel(const Batch b) {
  static_assert(LAYOUT == kNT || LAYOUT == kNN, "weight-gradient (kTN) products stay on the split-K 32x32 kernel");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS16];
  constexpr bool B_KC = LAYOUT == kNT;

  const int grp = blockIdx.z;
  const float* __restrict__ gA = b.p[grp].A;
  const float* __restrict__ gB = b.p[grp].B;
  float* __restrict__ gC = b.p[grp].C;
  const float* __restrict__ gbias = b.p[grp].bias;
  const float* __restrict__ gaux = b.p[grp].aux;
  const int M = b.p[grp].M, N = b.p[grp].N, K = b.p[grp].K;
  const int lda = b.p[grp].lda, ldb = b.p[grp].ldb, ldc = b.p[grp].ldc, ldaux = b.p[grp].ldaux;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  if (m0 >= M || n0 >= N) return;  // grouped launch: the grid is sized for the largest problem

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int rb = wave % 6, ch = wave / 6;  // 16-row block, 48-column half
  const bool full = (m0 + T <= M) && (n0 + T <= N) && (K % BK == 0);

  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto gload = [&](float4& ra, float4& rb_, int k0) {
    ra = load_kc(gA, lda, m0, M, k0, K, t, full);
    rb_ = B_KC ? load_kc(gB, ldb, n0, N, k0, K, t, full) : load_ks(gB, ldb, n0, N, k0, K, t, full);
  };
  auto sstore = [&](const float4& ra, const float4& rb_, int buf) {
    float* As = lds + buf * BUF;
    store_kc(As, ra, t);
    if (B_KC) store_kc(As + OPERAND, rb_, t);
    else store_ks(As + OPERAND, rb_, t);
  };
  auto compute = [&](int buf) {
    const float* __restrict__ As = lds + buf * BUF;
    const float* __restrict__ Bs = As + OPERAND;
    const int arow = rb * 16 + i;
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-wide K chunks per step
      const float4 fa = *reinterpret_cast<const float4*>(As + ((c * T + arow) << 4) + 4 * (q ^ swz(arow)));
      float4 fb[3];
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        const int brow = ch * 48 + cb * 16 + i;  // output column = row of B in kNT, column of B in kNN
        if (B_KC) {
          fb[cb] = *reinterpret_cast<const float4*>(Bs + ((c * T + brow) << 4) + 4 * (q ^ swz(brow)));
        } else {
          const float* __restrict__ p = Bs + (c * 16 + 4 * q) * KS_LD + brow;
          fb[cb] = make_float4(p[0], p[KS_LD], p[2 * KS_LD], p[3 * KS_LD]);
        }
      }
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb[cb].x, acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb[cb].y, acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb[cb].z, acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb[cb].w, acc[cb], 0, 0, 0);
    }
  };

  // K loop: same two-ahead pipeline as gemm_f32.hip (set 0 / set 1 alternate, so both are static registers)
  const int nk = (K + BK - 1) / BK;
  float4 ra0, rb0, ra1, rb1;
  ra0 = rb0 = ra1 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nk > 0) {
    gload(ra0, rb0, 0);
    if (nk > 1) gload(ra1, rb1, BK);
    sstore(ra0, rb0, 0);
  }
  __syncthreads();
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    gload(ra0, rb0, (kt + 2) * BK);
    compute(0);
    sstore(ra1, rb1, 1);
    __syncthreads();
    gload(ra1, rb1, (kt + 3) * BK);
    compute(1);
    sstore(ra0, rb0, 0);
    __syncthreads();
  }
  if (kt < nk) {  // 1..3 tiles left (LDS 0 = tile kt; set 1 = tile kt+1 if it exists)
    const bool t1 = kt + 1 < nk, t2 = kt + 2 < nk;
    if (t2) gload(ra0, rb0, (kt + 2) * BK);
    compute(0);
    if (t1) {
      sstore(ra1, rb1, 1);
      __syncthreads();
      compute(1);
      if (t2) {
        sstore(ra0, rb0, 0);
        __syncthreads();
        compute(0);
      }
    }
    __syncthreads();
  }

  // epilogue through LDS: C/D of the 16x16 MFMA is col = lane & 15, row = 4 * (lane >> 4) + reg
  float* __restrict__ Cs = lds;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) Cs[(rb * 16 + 4 * q + e) * CP + ch * 48 + cb * 16 + i] = acc[cb][e];
  __syncthreads();
  const bool vecC = ((ldc & 3) == 0) && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(gC) & 15) == 0) &&
                    (EPI != kReluMask || (((ldaux & 3) == 0) && ((reinterpret_cast<uintptr_t>(gaux) & 15) == 0)));
  // 96 x 24 float4 chunks over 768 threads: thread -> 4 columns c4, rows row_first + 32 * j
  const int c4 = (t % 24) * 4, n = n0 + c4;
  const int row_first = t / 24;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = row_first + 32 * j, m = m0 + row;
    if (m >= M || n >= N) continue;
    float4 x = *reinterpret_cast<const float4*>(Cs + row * CP + c4);
    float* xe = &x.x;
    if (vecC) {
      if (EPI == kBias || EPI == kBiasRelu) {
        const float4 bv = *reinterpret_cast<const float4*>(gbias + n);
        x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w;
      }
      if (EPI == kBiasRelu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      if (EPI == kReluMask) {
        const float4 a = *reinterpret_cast<const float4*>(gaux + (long long)m * ldaux + n);
        x.x = a.x > 0.f ? x.x : 0.f; x.y = a.y > 0.f ? x.y : 0.f; x.z = a.z > 0.f ? x.z : 0.f; x.w = a.w > 0.f ? x.w : 0.f;
      }
      *reinterpret_cast<float4*>(gC + (long long)m * ldc + n) = x;
    } else {
      for (int c = 0; c < 4 && n + c < N; ++c) {
        float v = xe[c];
        if (EPI == kBias || EPI == kBiasRelu) v += gbias[n + c];
        if (EPI == kBiasRelu) v = fmaxf(v, 0.f);
        if (EPI == kReluMask) v = gaux[(long long)m * ldaux + n + c] > 0.f ? v : 0.f;
        gC[(long long)m * ldc + n + c] = v;
      }
    }
  }
}

// what the staged loads need: 16-byte aligned operands, leading dimensions and K multiples of 4; kNN also N % 4
bool g16_ok(Layout layout, const Problem& p) {
  const bool base = tt_aligned16(p.A) && tt_aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0) && p.K % 4 == 0 && p.K >= 4;
  if (!base) return false;
  return layout == kNT || (p.N % 4 == 0 && p.N >= 4);
}

}  // namespace

bool gemm16_supported(Layout layout, Epilogue epi, const Batch& b) {
  if (b.nsplit != 1 || b.count < 1 || b.count > kMaxGroup) return false;
  // kNN (dH = dOut W2, W2 read K-strided through ds_read_b32) is built and parity-tested but measured slower than the
  // 64x64 kernel (41 vs 27 us at [6144 x 384 x 384]): TT_G16_NN=1 turns it on for A/B runs only.
  static const bool nn_on = getenv("TT_G16_NN") && getenv("TT_G16_NN")[0] == '1';
  if (!((layout == kNT && (epi == kBiasRelu || epi == kBias)) || (nn_on && layout == kNN && epi == kReluMask))) return false;
  for (int i = 0; i < b.count; ++i)
    if (b.p[i].M > 0 && !g16_ok(layout, b.p[i])) return false;
  return true;
}

size_t gemm16_lds_bytes() { return (size_t)LDS_FLOATS16 * sizeof(float); }

int launch16(Layout layout, Epilogue epi, const Batch& b, hipStream_t st) {
  if (!gemm16_supported(layout, epi, b)) return TT_ERR_UNSUPPORTED;
  int maxM = 0, maxN = 0;
  for (int i = 0; i < b.count; ++i) {
    maxM = b.p[i].M > maxM ? b.p[i].M : maxM;
    maxN = b.p[i].N > maxN ? b.p[i].N : maxN;
  }
  if (maxM == 0 || maxN == 0) return TT_OK;
  const dim3 grid((maxN + T - 1) / T, (maxM + T - 1) / T, b.count), block(kThreads16);
  // TT_G16_LDS_KB (A/B): dynamic LDS on top of the static tile buffers — at > 80 KB in all, one workgroup per CU
  static const unsigned pad = getenv("TT_G16_LDS_KB") ? (unsigned)atoi(getenv("TT_G16_LDS_KB")) * 1024u : 0u;
  if (layout == kNT && epi == kBiasRelu) hipLaunchKernelGGL((gemm16_kernel<kNT, kBiasRelu>), grid, block, pad, st, b);
  else if (layout == kNT) hipLaunchKernelGGL((gemm16_kernel<kNT, kBias>), grid, block, pad, st, b);
  else hipLaunchKernelGGL((gemm16_kernel<kNN, kReluMask>), grid, block, pad, st, b);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

}  // namespace ttg
