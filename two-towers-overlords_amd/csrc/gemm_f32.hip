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
// consecutive columns).  (Round 4 built the alternative for kTN — both K-strided operands transposed in the register
// file on their way in, four consecutive k of one column per thread, so that its fragments are one ds_read_b128 as well;
// bitwise the same products — and measured it: no gain at any shape, one-stream configs[3] 1.6 % slower;
// profiles/r04_tn_transpose_ab.txt.  Removed: 32 ds_read_b32 per K-step and wave are 256 LDS clocks against 1024 MFMA clocks.)
// (Round 4 also built the K loop with FOUR register staging sets for the 64x64 tiles — a global load requested four
// K-steps ahead of its use instead of two, 124 instead of 92 VGPRs, bitwise the same products — on the reading that beside
// the gather the tiles wait for loads: BASELINE.json configs[1] 175.1 -> 186.1 us per step, configs[2]'s share 142.7 ->
// 145.9; profiles/r04_proj_depth_ab.txt.  Removed: the tiles are not short of requests in flight, and the registers
// cost a resident wave per SIMD.)
//
// (Round 5 built K-step 64 for the 64x64 tiles — one barrier per 32 MFMAs instead of 16, 70 KB of LDS, 116-120 registers,
// bitwise the same products — on in-kernel phase stamps (tools/gemm_stamps.hip) that show a workgroup's K loop taking 2.3 x
// its MFMA time with the chip otherwise idle: K loop 11.96 -> 11.44 us at configs[2]'s share but the prologue 1.24 -> 1.84,
// launches 19.8 -> 20.3 us; at configs[1] the 70 KB stop the third workgroup of a CU from being resident: 27.1 -> 29.6 us;
// profiles/r05_gemm_stamps.txt.  Removed.  What the stamps do show: all workgroups of a launch start within 0.6 us, the
// prologue is 1.2-1.6 us, the epilogue 0.6-1 us — the launches are bound by the CUs that hold the most tiles (2 of 64x64
// at 1.5 tiles per CU, 3 at 2.25), each tile's K loop running at 0.7-0.86 of the MFMA rate while it shares its SIMDs.)
//
// (Round 5 also built BALANCED launches for the tile quantisation the stamps show — the left-over tile-units of a launch as
// two 32x64 half tiles each, K split between wave pairs inside the workgroup — after a probe (half-K full tiles) had run the
// same work 15-22 % faster: the real half tile, overhead-bound at 8 MFMAs per barrier and LDS-heavy at 16, gained 3 % alone
// at best and lost 10-18 % beside the gather; docs/LOG_r05.md section 1, profiles/r05_gemm_stamps.txt.  Removed.)
//
// Numerics: each C element is a k-ordered fmaf chain in fp32 (what the MFMA
// computes, bitwise), K-steps in ascending order; split-K (kTN) partials go to
// separate slabs that are summed in fixed order later => run-to-run bitwise
// reproducible, no float atomics.
#include <type_traits>
#include "gemm_f32.h"

#ifdef TT_GEMM_STAMPS
// Diagnostic builds only (tools/gemm_stamps.hip): thread 0 of every workgroup leaves the 100 MHz wall clock at entry, after
// the prologue (first operand tile in LDS), after the K loop and after the epilogue.
__device__ unsigned long long tt_gemm_stamps[4 * 8192];
__device__ unsigned long long tt_gemm_cycles[4 * 8192];  // the shader clock at the same points (clock64: follows DVFS)
#define TT_STAMP(i)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0) {                                                                                      \
      const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                       \
      if (wg_ < 8192) {                                                                                          \
        tt_gemm_stamps[4 * wg_ + (i)] = wall_clock64();                                                          \
        tt_gemm_cycles[4 * wg_ + (i)] = clock64();                                                               \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#else
#define TT_STAMP(i)
#endif

namespace ttg {
namespace {

constexpr int kThreads = 256;
// K-step: 32 by default; 16 for the 128x128 tiles that share a CU with the streaming gather (half the operand
// LDS — 40 KB instead of 72 — and 32 staging registers fewer, so two tiles AND the gather's waves stay resident).
// A K-contiguous operand's LDS row is BK + 4 floats: 36 or 20 words = 9 or 5 sixteen-byte slots, both coprime
// with the 16 slots of a bank row, so the 16-lane groups of ds_read_b128 stay conflict-free.

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- global -> register staging ------------------------------------------------
// VEC path (16-byte aligned base, ld % 4 == 0, and for K-contiguous operands K % 4 == 0):
// branch-free — the address is clamped in range, the float4 load is unconditional and
// out-of-range chunks are zeroed with a select, so the loads of tile t+1 stay in flight
// under the MFMAs of tile t.  The scalar path (any shape) is correct, not fast.
//
// K-contiguous operand: tile [R rows][BK], element (row, k) at P[(r0+row)*ld + k0+k].
template <int R, int BK, bool VEC, bool FULL>
__device__ __forceinline__ void gload_kc(float4 (&v)[R * BK / 4 / kThreads], const float* __restrict__ P,
                                         int ld, int r0, int rmax, int k0, int kend, int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int row = f / (BK / 4), kq = f % (BK / 4);
    const int gr = r0 + row, gk = k0 + kq * 4;
    if (FULL) {  // interior tile: nothing to guard, so the load stays unconditional and countable
      v[i] = *reinterpret_cast<const float4*>(P + (long long)gr * ld + gk);
    } else if (VEC) {
      const bool ok = gr < rmax && gk < kend;  // K % 4 == 0: a chunk is all in or all out
      const int cr = min(gr, rmax - 1), ck = min(gk, kend - 4);
      const float4 t = *reinterpret_cast<const float4*>(P + (long long)cr * ld + ck);
      v[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* __restrict__ src = P + (long long)gr * ld + gk;
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
template <int R, int BK>
__device__ __forceinline__ void sstore_kc(const float4 (&v)[R * BK / 4 / kThreads], float* __restrict__ T, int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int row = f / (BK / 4), kq = f % (BK / 4);
    *reinterpret_cast<float4*>(T + row * (BK + 4) + kq * 4) = v[i];
  }
}
// K-strided operand: tile [BK][R cols], element (k, col) at P[(k0+k)*ld + r0+col].
// VEC additionally needs rmax % 4 == 0 (a 4-column chunk is all in or all out).
template <int R, int BK, bool VEC, bool FULL>
__device__ __forceinline__ void gload_ks(float4 (&v)[R * BK / 4 / kThreads], const float* __restrict__ P,
                                         int ld, int r0, int rmax, int k0, int kend, int tid) {
  constexpr int RQ = R / 4;
#pragma unroll
  for (int i = 0; i < R * BK / 4 / kThreads; ++i) {
    const int f = tid + i * kThreads;
    const int krow = f / RQ, rq = f % RQ;
    const int gk = k0 + krow, gr = r0 + rq * 4;
    if (FULL) {
      v[i] = *reinterpret_cast<const float4*>(P + (long long)gk * ld + gr);
    } else if (VEC) {
      const bool ok = gk < kend && gr < rmax;
      const int ck = min(gk, kend - 1), cr = min(gr, rmax - 4);
      const float4 t = *reinterpret_cast<const float4*>(P + (long long)ck * ld + cr);
      v[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* __restrict__ src = P + (long long)gk * ld + gr;
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
template <int R, int BK>
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
template <int BK>
__device__ __forceinline__ float4 frag_kc(const float* __restrict__ T, int row, int j, int h) {
  return *reinterpret_cast<const float4*>(T + row * (BK + 4) + 8 * j + 4 * h);
}
template <int R>
__device__ __forceinline__ float4 frag_ks(const float* __restrict__ T, int col, int j, int h) {
  const float* __restrict__ p = T + (8 * j + 4 * h) * R + col;
  return make_float4(p[0], p[R], p[2 * R], p[3 * R]);
}

template <int LAYOUT, int WM, int WN, int BK = 32>
struct TileCfg {
  static constexpr int BM = 64 * WM, BN = 64 * WN, BKP = BK + 4;
  static constexpr bool A_KC = (LAYOUT != kTN);   // K-contiguous LDS image
  static constexpr bool B_KC = (LAYOUT == kNT);
  static constexpr int A_TILE = A_KC ? BM * BKP : BK * BM;
  static constexpr int B_TILE = B_KC ? BN * BKP : BK * BN;
  static constexpr int CP = BN + 4;  // padded row of the epilogue's C tile
  // the C tile leaves through LDS in EP passes of BM / EP rows (one pass per wave row), so that it needs no more
  // LDS than the operand buffers do when those are small (BK = 16)
  static constexpr int EP = (2 * (A_TILE + B_TILE) >= BM * CP) ? 1 : WM;
  static constexpr int EP_ROWS = BM / EP;
  static constexpr int LDS_FLOATS = (2 * (A_TILE + B_TILE) > EP_ROWS * CP) ? 2 * (A_TILE + B_TILE) : EP_ROWS * CP;
};

// One workgroup's tile: (bx, by) tile coordinates, bz = group * nsplit + split.
template <int LAYOUT, int EPI, int WM, int WN, bool VEC, int BK = 32>
__device__ __forceinline__ void gemm_body(const Batch& b, float* __restrict__ lds, int bx, int by, int bz) {
  using Cfg = TileCfg<LAYOUT, WM, WN, BK>;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  [[maybe_unused]] constexpr int BKP = Cfg::BKP;
  constexpr bool A_KC = Cfg::A_KC;
  constexpr bool B_KC = Cfg::B_KC;
  constexpr int A_TILE = Cfg::A_TILE;
  constexpr int B_TILE = Cfg::B_TILE;
  constexpr int NVA = BM * BK / 4 / kThreads, NVB = BN * BK / 4 / kThreads;
  static_assert(NVA >= 1 && NVB >= 1, "a K-step of the tile is at least one 16-byte chunk per thread");

  constexpr int CP = Cfg::CP;  // padded row of the epilogue's C tile

  const int grp = bz / b.nsplit;
  const int split = bz - grp * b.nsplit;
  // Copy the problem descriptor out of the kernarg segment once: stores to C could
  // alias it as far as the compiler knows, which would force a reload + vmcnt(0)
  // after every store.
  const float* __restrict__ gA = b.p[grp].A;
  const float* __restrict__ gB = b.p[grp].B;
  float* __restrict__ gC = b.p[grp].C;
  const float* __restrict__ gbias = b.p[grp].bias;
  const float* __restrict__ gaux = b.p[grp].aux;
  float* __restrict__ gcolsum = b.p[grp].colsum;
  const int M = b.p[grp].M, N = b.p[grp].N, K = b.p[grp].K;
  const int lda = b.p[grp].lda, ldb = b.p[grp].ldb, ldc = b.p[grp].ldc, ldaux = b.p[grp].ldaux;
  const int kps = b.p[grp].k_per_split;

  const int m0 = by * BM, n0 = bx * BN;
  if (m0 >= M || n0 >= N) return;  // grouped launch: grid is sized for the largest problem
  TT_STAMP(0);

  const bool ksplit = (LAYOUT == kTN) || b.nsplit > 1;  // split z covers k in [z * kps, (z + 1) * kps)
  const int kbeg = ksplit ? min(K, split * kps) : 0;
  const int kend = ksplit ? min(K, kbeg + kps) : K;
  const long long slab = (long long)split * b.slab_stride;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wrow = wave >> 1, wcol = wave & 1;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Two register staging sets: tile t+2 is requested while tile t is multiplied and
  // tile t+1 (requested one step earlier) is written to LDS, so a global load has two
  // K-steps to land (the K loop is unrolled by two so that both sets are static).
  float4 ra0[NVA], rb0[NVB], ra1[NVA], rb1[NVB];
  float csum = 0.f;  // kTN column sums of A (bias gradient), threads tid < BM of n-tile 0
  const bool do_colsum = (LAYOUT == kTN) && gcolsum != nullptr && bx == 0;

  // FULL: this block's tiles lie wholly inside A, B (rows and k) — true for every block of
  // the batch shapes the trainer uses; edge blocks take the guarded loads.
  const bool full = VEC && (m0 + BM <= M) && (n0 + BN <= N) && ((kend - kbeg) % BK == 0);
  auto gload = [&](float4 (&ra)[NVA], float4 (&rb)[NVB], int k0, auto full_tag) {
    constexpr bool F = decltype(full_tag)::value;
    if constexpr (A_KC) gload_kc<BM, BK, VEC, F>(ra, gA, lda, m0, M, k0, kend, tid);
    else gload_ks<BM, BK, VEC, F>(ra, gA, lda, m0, M, k0, kend, tid);
    if constexpr (B_KC) gload_kc<BN, BK, VEC, F>(rb, gB, ldb, n0, N, k0, kend, tid);
    else gload_ks<BN, BK, VEC, F>(rb, gB, ldb, n0, N, k0, kend, tid);
  };
  auto sstore = [&](const float4 (&ra)[NVA], const float4 (&rb)[NVB], int buf) {
    float* As = lds + buf * (A_TILE + B_TILE);
    float* Bs = As + A_TILE;
    if (A_KC) sstore_kc<BM, BK>(ra, As, tid); else sstore_ks<BM, BK>(ra, As, tid);
    if (B_KC) sstore_kc<BN, BK>(rb, Bs, tid); else sstore_ks<BN, BK>(rb, Bs, tid);
  };
  auto compute = [&](int buf) {
    const float* __restrict__ As = lds + buf * (A_TILE + B_TILE);
    const float* __restrict__ Bs = As + A_TILE;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 fa[WM], fb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = (wrow * WM + i) * 32 + r;
        fa[i] = A_KC ? frag_kc<BK>(As, row, j, h) : frag_ks<BM>(As, row, j, h);
      }
#pragma unroll
      for (int i = 0; i < WN; ++i) {
        const int col = (wcol * WN + i) * 32 + r;
        fb[i] = B_KC ? frag_kc<BK>(Bs, col, j, h) : frag_ks<BN>(Bs, col, j, h);
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
    if constexpr (LAYOUT == kTN) {
      if (do_colsum && tid < BM) {
#pragma unroll
        for (int k = 0; k < BK; ++k) csum += As[k * BM + tid];
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  auto kloop = [&](auto full_tag) {
    if (nk > 0) {
      gload(ra0, rb0, kbeg, full_tag);
      if (nk > 1) gload(ra1, rb1, kbeg + BK, full_tag);
      sstore(ra0, rb0, 0);
    }
    __syncthreads();
    TT_STAMP(1);
    int kt = 0;
    // steady state, two K-steps per trip, no conditionals inside (so the compiler can keep
    // the newest request in flight behind a counted vmcnt): on entry LDS 0 holds tile kt
    // and set1 holds tile kt+1 (in flight).
    for (; kt + 3 < nk; kt += 2) {
      gload(ra0, rb0, kbeg + (kt + 2) * BK, full_tag);
      compute(0);
      sstore(ra1, rb1, 1);
      __syncthreads();
      gload(ra1, rb1, kbeg + (kt + 3) * BK, full_tag);
      compute(1);
      sstore(ra0, rb0, 0);
      __syncthreads();
    }
    // tail: 1..3 tiles left (LDS 0 = tile kt; set1 = tile kt+1 if it exists)
    if (kt < nk) {
      const bool t1 = kt + 1 < nk, t2 = kt + 2 < nk;
      if (t2) gload(ra0, rb0, kbeg + (kt + 2) * BK, full_tag);
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
  };
  if (VEC && full) kloop(std::true_type{});
  else kloop(std::false_type{});
  TT_STAMP(2);

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*h.
  // The tile goes through LDS (free after the K loop's last barrier) so that global
  // stores are whole 16-byte chunks of contiguous rows (4 per thread instead of 16
  // dword stores whose address registers the compiler serialises with vmcnt(0)).
  // With EP > 1 (small operand buffers, BK = 16) it leaves in EP passes of EP_ROWS rows: pass `ep` is the rows of
  // the waves with wrow == ep.
  float* __restrict__ Cs = lds;
  constexpr int EP = Cfg::EP, EP_ROWS = Cfg::EP_ROWS;
  static_assert(EP == 1 || EP == WM, "one pass per wave row");
  const bool vecC = VEC && ((ldc & 3) == 0) && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(gC) & 15) == 0) &&
                    ((b.slab_stride & 3) == 0) &&
                    (EPI != kReluMask || (((ldaux & 3) == 0) && ((reinterpret_cast<uintptr_t>(gaux) & 15) == 0)));
  constexpr int NI = EP_ROWS * BN / 4 / kThreads;  // 16-byte chunks of one pass per thread
  static_assert(kThreads % (BN / 4) == 0, "every chunk of a thread lies in the same 4 columns");
  const int c4 = (tid % (BN / 4)) * 4, n = n0 + c4;
  const int row_first = tid / (BN / 4);
  constexpr int ROW_STEP = kThreads / (BN / 4);
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    if (ep > 0) __syncthreads();  // the previous pass has been read out
    if (EP == 1 || wrow == ep) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jj = 0; jj < WN; ++jj) {
          const int col = (wcol * WN + jj) * 32 + r;
          const int rbase = ((EP == 1 ? wrow * WM : 0) + i) * 32 + 4 * h;
#pragma unroll
          for (int e = 0; e < 16; ++e) Cs[(rbase + (e & 3) + 8 * (e >> 2)) * CP + col] = acc[i][jj][e];
        }
    }
    __syncthreads();
    const int mrow0 = m0 + ep * EP_ROWS;  // first output row of this pass
    if (EPI == kHinge) {
      // S tile -> dL/dS tile; per row: hinge sum and active count of this tile's columns (reduced over the BN / 4
      // threads that hold the row: consecutive lanes of one wave) -> partials[row][column tile]
      constexpr int RT = BN / 4;  // threads per row: 16 or 32 consecutive lanes
      const HingeEpi& hp = b.hinge;
      for (int i = 0; i < NI; ++i) {
        const int row = row_first + i * ROW_STEP;
        const int m = mrow0 + row;
        const float4 x = *reinterpret_cast<const float4*>(Cs + row * CP + c4);
        const float xe[4] = {x.x, x.y, x.z, x.w};
        float ge[4] = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f, c = 0.f;
        const bool row_ok = m < M;
        if (row_ok && n < N) {
          const int pi = hp.pos_index ? hp.pos_index[m] : m;
          const float sp = hp.s_pos[m];
          const long long myq = hp.qid_q ? hp.qid_q[m] : 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = n + e;
            if (j < N) {
              const bool valid = j != pi && !(hp.qid_q && hp.qid_d[j] == myq);
              const float h = xe[e] - sp + hp.margin;  // = (1 - s_pos) - (1 - s[j]) + margin
              if (valid && h > 0.f) {
                l += h;
                c += 1.f;
                ge[e] = hp.g;
              }
            }
          }
          float* __restrict__ dst = gC + (long long)m * ldc + n;
          if (vecC && n + 3 < N) {
            *reinterpret_cast<float4*>(dst) = make_float4(ge[0], ge[1], ge[2], ge[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < N) dst[e] = ge[e];
          }
        }
#pragma unroll
        for (int off = RT / 2; off > 0; off >>= 1) {
          l += __shfl_xor(l, off, TT_WAVE);
          c += __shfl_xor(c, off, TT_WAVE);
        }
        if ((tid % RT) == 0 && row_ok) {
          hp.part_loss[(long long)m * hp.n_tiles + bx] = l;
          hp.part_cnt[(long long)m * hp.n_tiles + bx] = c;
        }
      }
      continue;
    }
    if (vecC && full) {
      // Interior tile: no guards, so nothing splits the stores into exec-masked branches
      // (each of which the compiler fences with vmcnt(0)).  All loads first (tile chunks from
      // LDS, bias once, ReLU-mask rows), then NI stores back to back.
      float4 v[NI], am[NI];
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == kBias || EPI == kBiasRelu) bv = *reinterpret_cast<const float4*>(gbias + n);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = row_first + i * ROW_STEP;
        v[i] = *reinterpret_cast<const float4*>(Cs + row * CP + c4);
        if (EPI == kReluMask) am[i] = *reinterpret_cast<const float4*>(gaux + (long long)(mrow0 + row) * ldaux + n);
      }
      float* __restrict__ cbase = gC + slab + (long long)(mrow0 + row_first) * ldc + n;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float4 x = v[i];
        if (EPI == kBias || EPI == kBiasRelu) { x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w; }
        if (EPI == kBiasRelu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        if (EPI == kReluMask) {
          x.x = am[i].x > 0.f ? x.x : 0.f; x.y = am[i].y > 0.f ? x.y : 0.f;
          x.z = am[i].z > 0.f ? x.z : 0.f; x.w = am[i].w > 0.f ? x.w : 0.f;
        }
        *reinterpret_cast<float4*>(cbase + (long long)i * ROW_STEP * ldc) = x;
      }
    } else if (vecC) {
      const bool col_ok = n < N;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((EPI == kBias || EPI == kBiasRelu) && col_ok) bv = *reinterpret_cast<const float4*>(gbias + n);
      for (int i = 0; i < NI; ++i) {
        const int row = row_first + i * ROW_STEP;
        const int m = mrow0 + row;
        if (m >= M || !col_ok) continue;
        float4 x = *reinterpret_cast<const float4*>(Cs + row * CP + c4);
        if (EPI == kBias || EPI == kBiasRelu) { x.x += bv.x; x.y += bv.y; x.z += bv.z; x.w += bv.w; }
        if (EPI == kBiasRelu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        if (EPI == kReluMask) {
          const float4 a = *reinterpret_cast<const float4*>(gaux + (long long)m * ldaux + n);
          x.x = a.x > 0.f ? x.x : 0.f; x.y = a.y > 0.f ? x.y : 0.f; x.z = a.z > 0.f ? x.z : 0.f; x.w = a.w > 0.f ? x.w : 0.f;
        }
        *reinterpret_cast<float4*>(gC + slab + (long long)m * ldc + n) = x;
      }
    } else {
      for (int i = 0; i < NI; ++i) {
        const int row = row_first + i * ROW_STEP;
        const int m = mrow0 + row;
        if (m >= M) continue;
        const float4 v = *reinterpret_cast<const float4*>(Cs + row * CP + c4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (n + c >= N) break;
          float x = c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
          if (EPI == kBias || EPI == kBiasRelu) x += gbias[n + c];
          if (EPI == kBiasRelu) x = fmaxf(x, 0.f);
          if (EPI == kReluMask) x = gaux[(long long)m * ldaux + n + c] > 0.f ? x : 0.f;
          gC[slab + (long long)m * ldc + n + c] = x;
        }
      }
    }
  }
  if (do_colsum && tid < BM && m0 + tid < M) gcolsum[slab + m0 + tid] = csum;
  TT_STAMP(3);
}

// Tile coordinates of this workgroup.  Plain: the 3-D grid.  XCD-aware (Batch::xcd_order): a 1-D grid where
// workgroup i belongs to class i % 8 and is the (i / 8)-th of its class; a class walks ITS share of the sharing
// groups (row panels for kNT / kNN, split-K slabs for kTN: group g goes to class g % 8) one group after the other.
template <int LAYOUT>
__device__ __forceinline__ bool tile_coords(const Batch& b, int& bx, int& by, int& bz) {
  if (!b.xcd_order) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    return true;
  }
  const int cls = blockIdx.x & 7, t = blockIdx.x >> 3;
  if (LAYOUT == kTN) {
    const int per = b.gx * b.gy;                 // tiles of one slab
    const int slab = (t / per) * 8 + cls, tile = t % per;
    if (slab >= b.gz) return false;
    bx = tile % b.gx; by = tile / b.gx; bz = slab;
  } else {
    const int panel = (t / b.gx) * 8 + cls;      // row panel index over (group, m-tile)
    if (panel >= b.gy * b.gz) return false;
    bx = t % b.gx; by = panel % b.gy; bz = panel / b.gy;
  }
  return true;
}

template <int LAYOUT, int EPI, int WM, int WN, bool VEC, int BK = 32>
__global__ __launch_bounds__(kThreads) void gemm_f32_kernel(const Batch b) {
  __shared__ __attribute__((aligned(16))) float lds[TileCfg<LAYOUT, WM, WN, BK>::LDS_FLOATS];
  int bx, by, bz;
  if (!tile_coords<LAYOUT>(b, bx, by, bz)) return;
  gemm_body<LAYOUT, EPI, WM, WN, VEC, BK>(b, lds, bx, by, bz);
}

// The BK = 16 tile shares a CU with the streaming gather's workgroup (gather_pool.hip): three waves per SIMD's worth
// of registers (<= 168) leave room for two of these workgroups plus the gather's 4 waves of up to 176.
// (A 128-register build — a few dwords of scratch outside the K loop — changed nothing at configs[3]: 6.14-6.31 vs
// 6.13-6.31 ms, profiles/r03_c4_128_register_tiles_ab.txt.)
template <int LAYOUT, int EPI, int WM, int WN>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 8)))
void gemm_f32_bk16_kernel(const Batch b) {
  __shared__ __attribute__((aligned(16))) float lds[TileCfg<LAYOUT, WM, WN, 16>::LDS_FLOATS];
  int bx, by, bz;
  if (!tile_coords<LAYOUT>(b, bx, by, bz)) return;
  gemm_body<LAYOUT, EPI, WM, WN, true, 16>(b, lds, bx, by, bz);
}

// Two independent GEMMs of the projection backward in ONE launch: the split-K dW2/db2 (kTN,
// 64x64 tiles) and dH = (dOut W2) * [H > 0] (kNN).  Workgroups [0, n_tn) take the TN tiles, the
// rest the NN tiles (1-D grid, so no empty workgroups); saves a launch, and the two kinds of
// tile fill each other's ramp-up / ramp-down.
struct PairGeom {
  int tn_gx, tn_gy, tn_blocks;  // TN tile grid (x, y) and total TN workgroups (gx*gy*gz)
  int nn_gx, nn_gy;
  int tn_head, nn_blocks;       // workgroup order: TN [0, tn_head), all NN, the remaining TN (longest K loops first)
};

template <int WM, int WN>
__global__ __launch_bounds__(kThreads) void gemm_tn_nn_kernel(const Batch bt, const Batch bn, const PairGeom g) {
  constexpr int LT = TileCfg<kTN, 1, 1>::LDS_FLOATS, LN = TileCfg<kNN, WM, WN>::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LT > LN ? LT : LN];
  int id = blockIdx.x;
  const bool is_tn = id < g.tn_head || id >= g.tn_head + g.nn_blocks;
  if (is_tn) {
    if (id >= g.tn_head) id -= g.nn_blocks;
    const int per_z = g.tn_gx * g.tn_gy;
    const int z = id / per_z, rem = id - z * per_z;
    gemm_body<kTN, kStore, 1, 1, true>(bt, lds, rem % g.tn_gx, rem / g.tn_gx, z);
  } else {
    id -= g.tn_head;
    const int per_z = g.nn_gx * g.nn_gy;
    const int z = id / per_z, rem = id - z * per_z;
    gemm_body<kNN, kReluMask, WM, WN, true>(bn, lds, rem % g.nn_gx, rem / g.nn_gx, z);
  }
}

template <int LAYOUT>
bool problem_vec_ok(const Problem& p);

#ifdef TT_GEMM_KSPLIT  // tools/gemm_stamps.hip only: round 5's K-split 96-row tiles (tools/gemm_ksplit_tiles.inc)
#include "gemm_ksplit_tiles.inc"
#endif

// VEC requirements of one problem (see gload_kc / gload_ks)
template <int LAYOUT>
bool problem_vec_ok(const Problem& p) {
  const bool base = tt_aligned16(p.A) && tt_aligned16(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
  if (!base) return false;
  if (LAYOUT == kNT) return p.K % 4 == 0 && p.K >= 4;                       // A, B K-contiguous
  if (LAYOUT == kNN) return p.K % 4 == 0 && p.K >= 4 && p.N % 4 == 0 && p.N >= 4;  // B K-strided over N
  return p.M % 4 == 0 && p.N % 4 == 0 && p.M >= 4 && p.N >= 4 && p.K >= 1;  // kTN: both K-strided
}

template <int LAYOUT, int EPI, int WM, int WN, int BK>
void launch_tile(const Batch& b, dim3 grid, hipStream_t st) {
  const dim3 block(kThreads);
  if constexpr (BK == 16) hipLaunchKernelGGL((gemm_f32_bk16_kernel<LAYOUT, EPI, WM, WN>), grid, block, 0, st, b);
  else hipLaunchKernelGGL((gemm_f32_kernel<LAYOUT, EPI, WM, WN, true>), grid, block, 0, st, b);
}

template <int LAYOUT, int EPI>
int launch_le(const Batch& b_in, int wm, int wn, hipStream_t st, int bk = 32) {
  Batch b = b_in;
  int maxM = 0, maxN = 0;
  for (int i = 0; i < b.count; ++i) {
    maxM = b.p[i].M > maxM ? b.p[i].M : maxM;
    maxN = b.p[i].N > maxN ? b.p[i].N : maxN;
  }
  if (maxM == 0 || maxN == 0) return TT_OK;
#ifdef TT_GEMM_KSPLIT
  if (wm == 3) {  // K-split 96-row tiles where the shapes allow them, the 64x64 tiles elsewhere
    const int rc = launch_ks<LAYOUT, EPI>(b, wn, st);
    if (rc != TT_ERR_UNSUPPORTED) return rc;
    wm = wn = 1;
  }
#endif
  const int BM = 64 * wm, BN = 64 * wn;
  dim3 grid((maxN + BN - 1) / BN, (maxM + BM - 1) / BM, b.count * b.nsplit);
  const dim3 block(kThreads);
  bool vec = true;
  for (int i = 0; i < b.count; ++i) vec = vec && (b.p[i].M == 0 || problem_vec_ok<LAYOUT>(b.p[i]));
  if (!vec) b.xcd_order = 0;
  if (b.xcd_order) {  // 1-D grid, 8 classes x (their share of the sharing groups, rounded up) x tiles per group
    b.gx = (int)grid.x; b.gy = (int)grid.y; b.gz = (int)grid.z;
    const long long groups = LAYOUT == kTN ? b.gz : (long long)b.gy * b.gz;
    const long long per = LAYOUT == kTN ? (long long)b.gx * b.gy : b.gx;
    const long long total = 8 * ((groups + 7) / 8) * per;
    if (total > 0x7fffffffLL) b.xcd_order = 0;
    else grid = dim3((unsigned)total, 1, 1);
  }
  if (!vec) hipLaunchKernelGGL((gemm_f32_kernel<LAYOUT, EPI, 1, 1, false>), dim3((maxN + 63) / 64, (maxM + 63) / 64, b.count * b.nsplit), block, 0, st, b);
  else if (wm == 1 && wn == 1) launch_tile<LAYOUT, EPI, 1, 1, 32>(b, grid, st);
  else if (wm == 2 && wn == 1) launch_tile<LAYOUT, EPI, 2, 1, 32>(b, grid, st);
  else if (wm == 2 && wn == 2 && bk == 16) launch_tile<LAYOUT, EPI, 2, 2, 16>(b, grid, st);  // shares a CU with the streaming gather
  else if (wm == 2 && wn == 2) launch_tile<LAYOUT, EPI, 2, 2, 32>(b, grid, st);
  else return TT_ERR_UNSUPPORTED;
  TT_CHECK_LAUNCH();
  return TT_OK;
}

}  // namespace

int launch_tn_nn_pair(const Batch& bt, const Batch& bn, int wm, int wn, hipStream_t st) {
  bool vec = true;
  for (int i = 0; i < bt.count; ++i) vec = vec && (bt.p[i].M == 0 || problem_vec_ok<kTN>(bt.p[i]));
  for (int i = 0; i < bn.count; ++i) vec = vec && (bn.p[i].M == 0 || problem_vec_ok<kNN>(bn.p[i]));
  if (!vec || !((wm == 1 && wn == 1) || (wm == 2 && wn == 1))) {  // odd shapes: two ordinary launches
    int rc = launch(kTN, kStore, bt, 1, 1, st);
    if (rc != TT_OK) return rc;
    return launch(kNN, kReluMask, bn, wm, wn, st);
  }
  int tM = 0, tN = 0, nM = 0, nN = 0;
  for (int i = 0; i < bt.count; ++i) { tM = bt.p[i].M > tM ? bt.p[i].M : tM; tN = bt.p[i].N > tN ? bt.p[i].N : tN; }
  for (int i = 0; i < bn.count; ++i) { nM = bn.p[i].M > nM ? bn.p[i].M : nM; nN = bn.p[i].N > nN ? bn.p[i].N : nN; }
  if (tM == 0 || tN == 0 || nM == 0 || nN == 0) return TT_ERR_BAD_SHAPE;
  PairGeom g{};
  g.tn_gx = (tN + 63) / 64;
  g.tn_gy = (tM + 63) / 64;
  g.tn_blocks = g.tn_gx * g.tn_gy * bt.count * bt.nsplit;
  g.nn_gx = (nN + 64 * wn - 1) / (64 * wn);
  g.nn_gy = (nM + 64 * wm - 1) / (64 * wm);
  const long long nn_total = (long long)g.nn_gx * g.nn_gy * bn.count;
  const long long total = (long long)g.tn_blocks + nn_total;
  if (total > 0x7fffffffLL) return TT_ERR_BAD_SHAPE;
  g.nn_blocks = (int)nn_total;
  // TN problems arrive longest-K first (proj.hip): the first problem's workgroups lead, the NN tiles (K = P) follow,
  // shorter TN problems close the grid (measured against "all TN first" in round 1: 198 -> 196 us per step).
  g.tn_head = (bt.count > 1 && bt.p[0].K > bt.p[1].K) ? g.tn_gx * g.tn_gy * bt.nsplit : g.tn_blocks;
  const dim3 grid((unsigned)total), block(kThreads);
  if (wm == 1) hipLaunchKernelGGL((gemm_tn_nn_kernel<1, 1>), grid, block, 0, st, bt, bn, g);
  else hipLaunchKernelGGL((gemm_tn_nn_kernel<2, 1>), grid, block, 0, st, bt, bn, g);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

template <int WM, int WN, int BK>
static size_t lds_of(Layout layout) {
  const int f = layout == kNT ? TileCfg<kNT, WM, WN, BK>::LDS_FLOATS
                              : layout == kNN ? TileCfg<kNN, WM, WN, BK>::LDS_FLOATS : TileCfg<kTN, WM, WN, BK>::LDS_FLOATS;
  return (size_t)f * sizeof(float);
}

size_t tile_lds_bytes(Layout layout, int wm, int wn, int bk) {
#ifdef TT_GEMM_KSPLIT
  if (wm == 3) {
    const int f = layout == kNT ? (wn == 3 ? KsCfg<kNT, 3>::LDS_FLOATS : KsCfg<kNT, 2>::LDS_FLOATS)
                                : (wn == 3 ? KsCfg<kNN, 3>::LDS_FLOATS : KsCfg<kNN, 2>::LDS_FLOATS);
    return (size_t)f * sizeof(float);
  }
#endif
  if (wm == 2 && wn == 2 && bk == 16) return lds_of<2, 2, 16>(layout);
  if (wm == 2 && wn == 2) return lds_of<2, 2, 32>(layout);
  if (wm == 2 && wn == 1) return lds_of<2, 1, 32>(layout);
  return lds_of<1, 1, 32>(layout);
}

int launch(Layout layout, Epilogue epi, const Batch& b, int wm, int wn, hipStream_t st, int bk) {
  if (b.count < 1 || b.count > kMaxGroup || b.nsplit < 1 || (bk != 32 && bk != 16)) return TT_ERR_BAD_SHAPE;
  if (layout == kNT && epi == kBiasRelu) return launch_le<kNT, kBiasRelu>(b, wm, wn, st, bk);
  if (layout == kNT && epi == kBias) return launch_le<kNT, kBias>(b, wm, wn, st, bk);
  if (layout == kNT && epi == kStore) return launch_le<kNT, kStore>(b, wm, wn, st, bk);
  if (layout == kNT && epi == kHinge) return launch_le<kNT, kHinge>(b, wm, wn, st, bk);
  if (layout == kNN && epi == kReluMask) return launch_le<kNN, kReluMask>(b, wm, wn, st, bk);
  if (layout == kNN && epi == kStore) return launch_le<kNN, kStore>(b, wm, wn, st, bk);
  if (layout == kTN && epi == kStore) return launch_le<kTN, kStore>(b, wm, wn, st, bk);
  return TT_ERR_UNSUPPORTED;
}

}  // namespace ttg
