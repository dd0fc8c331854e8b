// K1 — token-embedding gather + masked mean-pool + L2 normalise (gfx950).
//
// Replaces backend/model.py:51-56,63-72 of the reference (see include/tt_hotpath.h).
//
// Mapping: one 256-thread workgroup (4 waves) per text.  A lane owns 16-byte
// column chunks {lane, lane+64, ...} of the D-float row, so one wave-instruction
// reads 1 KiB of one embedding row, fully coalesced.  Token positions are dealt
// round-robin to the 4 waves (position % 4), each wave keeps U rows in flight
// (U x C dwordx4 loads issued before the first use) and accumulates in
// registers; masked-out tokens are never fetched (their weight is 0), so the
// bytes moved are the algorithmic bytes.  The 4 partial sums meet in LDS, then
// mean (clamp 1e-9), L2 norm (wave __shfl_xor butterfly + 4-entry LDS), divide
// (clamp 1e-12) and one coalesced float4 store per lane.  The [N,S,D] tensor of
// the reference is never materialised.
//
// Row ids are wave-uniform (v_readlane from the lane that loaded them), so the
// row base address is scalar and each load is `global_load_dwordx4 v, v_off, s[base]`.
#include <atomic>
#include <new>
#include "tt_common.h"

namespace {

constexpr int kWavesPerText = 4;
constexpr int kThreads = kWavesPerText * TT_WAVE;

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ void f4_fma(float4& acc, float w, const float4& r) {
  acc.x = fmaf(w, r.x, acc.x);
  acc.y = fmaf(w, r.y, acc.y);
  acc.z = fmaf(w, r.z, acc.z);
  acc.w = fmaf(w, r.w, acc.w);
}

// Padding inside the ids themselves (round 5): a token whose id is INT32_MIN (or INT64_MIN) carries weight 0 whatever mask
// or length says — never read, never reported as out of range.  Lets a caller that has squeezed the attended tokens to
// the front of each row (tt_engine_pool_roles) run the gather without a per-text length load: beside the projection
// chain that one dependent scalar load per wave cost the column-sliced form 12 us per configs[1] launch (170.8 vs
// 158.9 us in the loop at full length, alone 85.1 vs 85.3; profiles/r05_module_path.txt).
__device__ __forceinline__ bool tt_is_pad(long long raw) { return raw == TT_PAD_ID32 || raw == (-9223372036854775807LL - 1LL); }

// Row of the token store a text reads from (tt_engine_pool_rows): never outside the store.
__device__ __forceinline__ long long store_row(int row, long long n_store, int* __restrict__ oob_flag) {
  if (row < 0 || row >= n_store) {
    if (oob_flag) atomicOr(oob_flag, 1);
    return row < 0 ? 0 : n_store - 1;
  }
  return row;
}

// Pops K token positions from `mine` (wave-uniform bit set), issues all K*C row
// loads, then accumulates.  DENSE: rows are hidden[text_base + seg + pos].
template <int K, int C, bool DENSE>
__device__ __forceinline__ void pool_group(unsigned long long& mine, const float* __restrict__ src,
                                           int id_lane, float w_lane, long long dense_row0, int D,
                                           int lane, bool last_ok, float4 (&acc)[C]) {
  float4 r[K][C];
  float wk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int pos = __builtin_ctzll(mine);
    mine &= mine - 1;
    wk[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_lane), pos));
    long long row;
    if (DENSE) {
      row = dense_row0 + pos;
    } else {
      row = (long long)__builtin_amdgcn_readlane(id_lane, pos);
    }
    const float4* __restrict__ rp = reinterpret_cast<const float4*>(src + row * (long long)D);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (c < C - 1 || last_ok) {
        r[k][c] = rp[c * TT_WAVE + lane];
      } else {
        r[k][c] = f4_zero();
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int c = 0; c < C; ++c) f4_fma(acc[c], wk[k], r[k][c]);
  }
}

// One text by one 4-wave workgroup (see the file header).  `part` / `red` are the workgroup's LDS.
// C = float4 chunks per lane = ceil(D/4/64); U = rows in flight per wave.
// (Register accumulation, not LDS-tile staging: the LDS-DMA form — 8 rows per wave by global_load_lds_dwordx4 into a
// 16 KB slab, one vmcnt(0), read back and accumulate — was built and measured in round 1: bit-identical, 152.0 vs
// 153.6 us on the 46.9 MB table, 243 vs 243 us per step, at 72 KB of LDS per workgroup; profiles/r01_k1_form_ab.txt.)
template <typename IdT, typename MaskT, int C, int U, bool DENSE>
__device__ __forceinline__ void pool_one_text(
    long long text, float4 (*part)[C * TT_WAVE], float* red,
    const float* __restrict__ src, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ out, int S, int D, long long V,
    int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  const bool last_ok = (C - 1) * TT_WAVE + lane < nchunk;
  // with text_rows, text t's ids / mask / length are row text_rows[t] of a token store [n_store, S]
  const long long stext = text_rows ? store_row(text_rows[text], n_store, oob_flag) : text;
  const long long tbase = stext * (long long)S;
  const int len = lens ? lens[stext] : S;

  float4 acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = f4_zero();
  float wtot = 0.f;

  // with prefix lengths nothing lies beyond `len`: do not even scan the padding
  const int send = (mask == nullptr && len < S) ? max(len, 0) : S;
  for (int seg = 0; seg < send; seg += TT_WAVE) {
    const int tok = seg + lane;
    int id = 0;
    float w = 0.f;
    if (tok < S) {
      w = mask ? static_cast<float>(mask[tbase + tok]) : (tok < len ? 1.f : 0.f);
      if (!DENSE) {
        const long long raw = static_cast<long long>(ids[tbase + tok]);
        if (tt_is_pad(raw)) w = 0.f;
        if (raw < 0 || raw >= V) {
          // never read outside the table: clamp, and tell the host if it asked
          if (w != 0.f && oob_flag) atomicOr(oob_flag, 1);
          id = raw < 0 ? 0 : static_cast<int>(V - 1);
        } else {
          id = static_cast<int>(raw);
        }
      }
    }
    wtot += w;
    const unsigned long long live = __ballot(w != 0.f);
    unsigned long long mine = live & (0x1111111111111111ULL << wave);
    int cnt = __popcll(mine);
    const long long drow0 = tbase + seg;
    while (cnt >= U) {
      pool_group<U, C, DENSE>(mine, src, id, w, drow0, D, lane, last_ok, acc);
      cnt -= U;
    }
    // the rest (< U positions) in groups of 8 / 4 / 2 / 1: any U works, positions stay in ascending order
    if (U > 8 && cnt >= 8) {
      pool_group<8, C, DENSE>(mine, src, id, w, drow0, D, lane, last_ok, acc);
      cnt -= 8;
    }
    if (U > 4 && cnt >= 4) {
      pool_group<4, C, DENSE>(mine, src, id, w, drow0, D, lane, last_ok, acc);
      cnt -= 4;
    }
    if (U > 2 && cnt >= 2) {
      pool_group<2, C, DENSE>(mine, src, id, w, drow0, D, lane, last_ok, acc);
      cnt -= 2;
    }
    if (cnt >= 1) {
      pool_group<1, C, DENSE>(mine, src, id, w, drow0, D, lane, last_ok, acc);
    }
  }

  // denominator: sum of the mask over ALL positions (every wave saw them all)
  const float wsum = tt_wave_sum(wtot);

#pragma unroll
  for (int c = 0; c < C; ++c) part[wave][c * TT_WAVE + lane] = acc[c];
  __syncthreads();

  const int t = threadIdx.x;
  const bool act = t < nchunk;  // C <= 4  =>  nchunk <= 256: one chunk per thread
  float4 m = f4_zero();
  float ss = 0.f;
  if (act) {
    const float4 a = part[0][t], b = part[1][t], c2 = part[2][t], d = part[3][t];
    const float den = fmaxf(wsum, 1e-9f);  // clamp(mask.sum(1), min=1e-9)
    m.x = (((a.x + b.x) + c2.x) + d.x) / den;
    m.y = (((a.y + b.y) + c2.y) + d.y) / den;
    m.z = (((a.z + b.z) + c2.z) + d.z) / den;
    m.w = (((a.w + b.w) + c2.w) + d.w) / den;
    ss = fmaf(m.x, m.x, fmaf(m.y, m.y, fmaf(m.z, m.z, m.w * m.w)));
  }
  ss = tt_wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float nrm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
  const float dn = fmaxf(nrm, 1e-12f);  // F.normalize eps
  if (act) {
    float4 o;
    o.x = m.x / dn;
    o.y = m.y / dn;
    o.z = m.z / dn;
    o.w = m.w / dn;
    reinterpret_cast<float4*>(out + text * (long long)D)[t] = o;
  }
}

template <typename IdT, typename MaskT, int C, int U, bool DENSE>
__global__ __launch_bounds__(kThreads) void pool_norm_kernel(
    const float* __restrict__ src, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ out, int S, int D, long long V,
    int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store) {
  __shared__ float4 part[kWavesPerText][C * TT_WAVE];
  __shared__ float red[kWavesPerText];
  pool_one_text<IdT, MaskT, C, U, DENSE>(blockIdx.x, part, red, src, ids, mask, lens, out, S, D, V, oob_flag, text_rows,
                                         n_store);
}

// ---------------------------------------------------------------------------------------------
// Streaming K1 for tables that really come from HBM (far beyond the 256 MiB Infinity Cache), built to run BESIDE
// the projection chain of another stream.  There the gather is bandwidth-bound and wants a fixed amount of loads
// in flight on every CU for its whole duration (two 4-wave workgroups = 96 KB at D = 768), and the MFMA kernels
// want the rest of the CU.  A grid of one workgroup per text cannot hold that balance: which stream's workgroup
// takes the registers / LDS a finished one leaves is the dispatcher's choice, and measured at
// BASELINE.json configs[3] either the gathers crowd the GEMM tiles out (chain 3.5 -> 7.8 ms) or an LDS reservation
// keeps them to a CU they then share one-for-one with a tile (K1 5.2 -> 6.8 ms) — 7.5 ms per step both ways.
// Here residency is explicit instead:
//   * persistent workgroups pull texts from a queue head (one relaxed atomic per 0.8 MB of rows);
//   * a candidate workgroup first registers on ITS compute unit (XCC id + SE/SH/CU bits of HW_ID -> a counter per
//     CU); when `per_cu` candidates have registered there before it, it leaves at once, so no CU ever hosts more
//     gather waves than planned, whatever the dispatcher does, and the MFMA kernels keep the rest;
//   * candidates are offered far beyond the `target` = per_cu x CUs the chip may keep (round 4; round 3 offered
//     2 x target, all of which the dispatcher placed within microseconds wherever there was room at that moment —
//     a traced launch beside the chain ended with 430 of 512 admitted, 24 CUs without a gather workgroup and 20
//     candidates on one CU).  The candidates the chip has no room for wait in the dispatcher's queue and are
//     placed as room appears — on a CU where a projection tile has just finished — so a CU that was full of tiles
//     when the launch started still gets its share.  A candidate costs one load when the target is reached or
//     the queue is empty (it leaves without registering), one atomic otherwise.
// Placement decides speed only: every text is pooled by exactly one workgroup, with the arithmetic of
// pool_norm_kernel (same order => bitwise the same output).  No wave ever waits for another workgroup: a
// turned-away candidate leaves at once (letting it linger for a bounded 8-32 us, so that fewer candidates are spent per
// microsecond, was measured and changes nothing: profiles/r04_c4_admission_ab.txt) and admitted workgroups leave
// when the queue runs dry.
constexpr int kCuSlots = 4096;  // 4 bits of XCC id x 8 bits of SE / SH / CU id
struct StreamCtl {
  int head;            // next text to hand out
  int admitted;        // workgroups that registered successfully
  int target;          // workgroups the launch means to keep (diagnostic, written by workgroup 0)
  int pad;
  int cu[kCuSlots];    // candidates that registered per compute unit (admitted: the first per_cu of them)
};

__device__ __forceinline__ unsigned tt_cu_key() {
  // s_getreg_b32 simm16 = (size - 1) << 11 | offset << 6 | id;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID[3:0]
  return ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);                  // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
}

template <typename IdT, typename MaskT, int C, int U>
__global__ __launch_bounds__(kThreads) void pool_norm_stream_kernel(
    const float* __restrict__ src, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ out, long long N, int S, int D, long long V,
    int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store, StreamCtl* __restrict__ ctl,
    int per_cu, int target, int relax_from) {
  __shared__ float4 part[kWavesPerText][C * TT_WAVE];
  __shared__ float red[kWavesPerText];
  __shared__ int s_next;
  if (threadIdx.x == 0) {
    int next = -1;
    if (blockIdx.x == 0) ctl->target = target;
    // relaxed agent-scope loads: a stale value only costs this candidate the registration it would have been refused
    const bool open = __hip_atomic_load(&ctl->admitted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target &&
                      __hip_atomic_load(&ctl->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < N;
    if (open) {
      // the last `target` candidates of the grid are kept wherever they land: a CU that was offered fewer than its
      // share while the others turned candidates away (seen once: 511 of 512) is made up for elsewhere
      const bool here = atomicAdd(&ctl->cu[tt_cu_key()], 1) < per_cu || (int)blockIdx.x >= relax_from;
      if (here && atomicAdd(&ctl->admitted, 1) < target)
        next = atomicAdd(&ctl->head, 1);  // (else: this CU has its share — leave)
    }
    s_next = next;
  }
  __syncthreads();
  long long text = s_next;
  while (text >= 0 && text < N) {
    // the next text's number travels through LDS; thread 0 asks for it now, everyone reads it after the two
    // barriers inside pool_one_text (so the write below cannot overtake this iteration's read above)
    pool_one_text<IdT, MaskT, C, U, false>(text, part, red, src, ids, mask, lens, out, S, D, V, oob_flag, text_rows,
                                           n_store);
    if (threadIdx.x == 0) s_next = atomicAdd(&ctl->head, 1);
    __syncthreads();
    text = s_next;
  }
}

// ---------------------------------------------------------------------------------------------
// XCD-sliced K1.  A 30522 x 384 fp32 table is 47 MB: 12x an XCD's 4 MiB L2, so the per-text kernel
// above is served by the Infinity Cache (~8.6 TB/s, MI355X_MICROARCH.md "Indexed rows").  Workgroups
// with equal blockIdx % 8 share an XCD (placement label: speed only, never correctness), so here
// workgroup b gathers only the tokens whose id % NS == b % NS: every XCD then re-reads one 1/NS
// slice of the table (5.9 MB at NS = 8) and most row reads hit its own L2.
//
// The grid is NS x G workgroups; wave w of workgroup (slice, g) takes texts g*4+w, +4G, +8G, ...; G covers
// all texts (one text per wave).  (A persistent grid of n workgroups per CU walking the texts with the same loop
// measured within 5 % alone and 4 % slower beside the projection kernels — 213-216 vs 205 us per step — and 6- or
// 8-wave workgroups no better: round 1-2 records, profiles/r01_k1_form_ab.txt; removed in round 4.)
// One (text, slice) step: the ids / mask of the NEXT
// text, if any, are loaded into registers first (their latency hides under this text's rows), this
// text's selected (id, weight) list — compacted into LDS one step earlier — is gathered U rows at a
// time, the slice's partial sum [D] goes to a workspace, then the next text's list is compacted.
// A second launch adds a text's NS partials in slice order (fixed order: bitwise reproducible),
// divides by the mask sum and L2-normalises.  Costs N*NS*D*4 bytes written + read.
// (NS = 16 in two grid phases, so that a slice fits the L2 whole, doubles the partials: 114 vs 96 us.)
template <typename IdT, typename MaskT, int C, int U, int NS, int MAXSEG, int WPB = kWavesPerText>
__global__ __launch_bounds__(WPB * TT_WAVE) void pool_sliced_kernel(
    const float* __restrict__ table, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ partial, float* __restrict__ wsum_out, long long N,
    int S, int D, long long V, int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store) {
  extern __shared__ int sliced_lds[];  // per wave: 2 x ([S] ids + [S] weights)
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  static_assert(NS == 8, "one residue class per XCD");
  const int slice = (int)(blockIdx.x % NS);
  const long long tstride = (long long)(gridDim.x / NS) * WPB;
  long long text = (long long)(blockIdx.x / NS) * WPB + wave;
  if (text >= N) return;  // wave-uniform; no workgroup barrier anywhere below
  int* const lbase = sliced_lds + wave * (4 * S);
  const int nchunk = D >> 2;
  const bool last_ok = (C - 1) * TT_WAVE + lane < nchunk;

  IdT idr[MAXSEG];
  MaskT mr[MAXSEG];
  int len_r = S;

  // registers <- ids / mask of text t (loads only; nothing waits on them here)
  auto fetch = [&](long long t) {
    const long long st = text_rows ? store_row(text_rows[t], n_store, oob_flag) : t;
    len_r = lens ? lens[st] : S;
    const long long tb = st * (long long)S;
#pragma unroll
    for (int k = 0; k < MAXSEG; ++k) {
      const int tok = k * TT_WAVE + lane;
      idr[k] = 0;
      mr[k] = 0;
      if (tok < S) {
        idr[k] = ids[tb + tok];
        if (mask) mr[k] = mask[tb + tok];
      }
    }
  };
  // registers -> compacted (id, weight) list of this slice in LDS buffer `buf`; returns the count
  auto compact = [&](int buf, float& wtot) -> int {
    int* __restrict__ lid = lbase + buf * (2 * S);
    float* __restrict__ lw = reinterpret_cast<float*>(lid + S);
    int n = 0;
    wtot = 0.f;
#pragma unroll
    for (int k = 0; k < MAXSEG; ++k) {
      const int tok = k * TT_WAVE + lane;
      float w = 0.f;
      int id = 0;
      if (tok < S) {
        w = mask ? static_cast<float>(mr[k]) : (tok < len_r ? 1.f : 0.f);
        const long long raw = static_cast<long long>(idr[k]);
        if (tt_is_pad(raw)) w = 0.f;
        if (raw < 0 || raw >= V) {
          if (w != 0.f && oob_flag) atomicOr(oob_flag, 1);
          id = raw < 0 ? 0 : static_cast<int>(V - 1);
        } else {
          id = static_cast<int>(raw);
        }
      }
      wtot += w;
      const bool sel = (w != 0.f) && (id % NS == slice);
      const unsigned long long m = __ballot(sel);
      if (sel) {
        const int at = n + __popcll(m & ((1ULL << lane) - 1ULL));
        lid[at] = id;
        lw[at] = w;
      }
      n += __popcll(m);
    }
    return n;  // the list is written and read by this wave only: LDS ops of one wave complete in order
  };

  float wtot = 0.f;
  fetch(text);
  int n = compact(0, wtot);
  int buf = 0;
  while (true) {
    const long long tnext = text + tstride;
    if (tnext < N) fetch(tnext);

    const int* __restrict__ lid = lbase + buf * (2 * S);
    const float* __restrict__ lw = reinterpret_cast<const float*>(lid + S);
    float4 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = f4_zero();
    for (int base = 0; base < n; base += TT_WAVE) {
      const int have = min(n - base, TT_WAVE);
      int id = 0;
      float w = 0.f;
      if (lane < have) {
        id = lid[base + lane];
        w = lw[base + lane];
      }
      unsigned long long mine = have == TT_WAVE ? ~0ULL : ((1ULL << have) - 1ULL);
      int cnt = have;
      while (cnt >= U) {
        pool_group<U, C, false>(mine, table, id, w, 0, D, lane, last_ok, acc);
        cnt -= U;
      }
      if (U > 8 && cnt >= 8) {
        pool_group<8, C, false>(mine, table, id, w, 0, D, lane, last_ok, acc);
        cnt -= 8;
      }
      if (U > 4 && cnt >= 4) {
        pool_group<4, C, false>(mine, table, id, w, 0, D, lane, last_ok, acc);
        cnt -= 4;
      }
      if (U > 2 && cnt >= 2) {
        pool_group<2, C, false>(mine, table, id, w, 0, D, lane, last_ok, acc);
        cnt -= 2;
      }
      if (cnt >= 1) pool_group<1, C, false>(mine, table, id, w, 0, D, lane, last_ok, acc);
    }

    // (storing the partials `sc1`, so that they leave this XCD's L2 to the table slice, measured 3 % alone and
    // 1 % in the step: not worth hand-written store instructions)
    float4* __restrict__ pp = reinterpret_cast<float4*>(partial + (text * NS + slice) * (long long)D);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (c < C - 1 || last_ok) pp[c * TT_WAVE + lane] = acc[c];
    if (slice == 0) {
      const float ws = tt_wave_sum(wtot);
      if (lane == 0) wsum_out[text] = ws;
    }
    if (tnext >= N) break;
    buf ^= 1;
    n = compact(buf, wtot);
    text = tnext;
  }
}

// One wave per text: partials summed in slice order, mean (clamp 1e-9), L2 normalise (clamp 1e-12).
template <int C, int NS>
__global__ __launch_bounds__(kThreads) void pool_combine_kernel(const float* __restrict__ partial,
                                                                const float* __restrict__ wsum,
                                                                float* __restrict__ out, long long N, int D) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long text = (long long)blockIdx.x * kWavesPerText + (threadIdx.x >> 6);
  if (text >= N) return;
  const int nchunk = D >> 2;
  const bool last_ok = (C - 1) * TT_WAVE + lane < nchunk;
  const float den = fmaxf(wsum[text], 1e-9f);
  const float4* __restrict__ pp = reinterpret_cast<const float4*>(partial + text * NS * (long long)D);
  float4 m[C];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    m[c] = f4_zero();
    if (c < C - 1 || last_ok) {
      float4 r[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) r[s] = pp[s * nchunk + c * TT_WAVE + lane];
      float4 a = r[0];
#pragma unroll
      for (int s = 1; s < NS; ++s) {
        a.x += r[s].x;
        a.y += r[s].y;
        a.z += r[s].z;
        a.w += r[s].w;
      }
      m[c] = make_float4(a.x / den, a.y / den, a.z / den, a.w / den);
      ss = fmaf(m[c].x, m[c].x, fmaf(m[c].y, m[c].y, fmaf(m[c].z, m[c].z, fmaf(m[c].w, m[c].w, ss))));
    }
  }
  const float dn = fmaxf(sqrtf(tt_wave_sum(ss)), 1e-12f);
  float4* __restrict__ op = reinterpret_cast<float4*>(out + text * (long long)D);
#pragma unroll
  for (int c = 0; c < C; ++c)
    if (c < C - 1 || last_ok) op[c * TT_WAVE + lane] = make_float4(m[c].x / dn, m[c].y / dn, m[c].z / dn, m[c].w / dn);
}

// ---------------------------------------------------------------------------------------------
// Column-sliced K1 (round 3).  The row-sliced form above still misses its XCD's L2 on ~40 % of the rows at
// V = 30522 (a 5.9 MB slice in a 4 MiB L2) and pays for N x 8 x D floats of partial sums.  Here a slice is 32
// COLUMNS of every row — one 128-byte line per row, V x 128 bytes per slice: 3.9 MB at V = 30522 — so a slice
// fits the L2 whole and the partial of a (text, slice) unit is that slice of the output itself (no partial sums
// at all; only one float per unit, the slice's sum of squares, for the L2 norm).
//
// Units u = slice * N + text, slice-major.  Workgroups with equal blockIdx % 8 share an XCD (placement label:
// speed only), and class x takes the units [x * UPX, (x + 1) * UPX), UPX = ceil(NS * N / 8), in order: an XCD
// works through at most two slices (NS = 12 at D = 384: one and a half), one after the other.  One wave per
// unit: lane = (row slot r = lane >> 3, 16-byte piece q = lane & 7), so one wave instruction reads the
// 128-byte pieces of 8 rows; 8 instructions (64 tokens, 8 KB) are in flight before the first use.  Token ids and
// weights are loaded coalesced (non-temporal: they must not displace the slice) and handed to the row slots by
// ds_bpermute.  Masked-out tokens are never fetched.  The 8 row slots meet in a 3-step butterfly, then mean
// (clamp 1e-9), the slice's sum of squares, and lanes 0..7 store the 128 bytes of un-normalised mean.  A second
// launch adds a text's NS sums of squares in slice order and divides its row (clamp 1e-12) in place.
// Same result as the other forms up to fp32 summation order; run-to-run bitwise reproducible.
constexpr int kColSliceFloats = 32;

// (66 registers: seven waves per SIMD.  Forced to 64 for eight — 12 bytes of scratch — it is slower alone, 91 vs 85 us,
// and in the step, 183-187 vs 179 us: profiles/r03_k1_column_slices.txt.  Several units per wave with the next unit's
// ids requested ahead: 90 registers, 94 us alone, 191-193 us per step for 1-4 units.  One short-lived wave per unit and
// as many of them as the register file holds is the form that wins.  Scalar base + 32-bit lane offsets bring it to 62
// registers — eight waves per SIMD, 82 vs 85 us alone — but beside the projection chain the step is 184-185 vs 178-180 us
// on one box (tools/r03_colslice9.sh): with 72-register waves three gather workgroups leave no room for a chain wave on
// their CU, with 64 they do, and sharing a CU wave by wave costs the chain more than taking turns.)
template <typename IdT, typename MaskT, int MAXSEG, int WPB>
__global__ __launch_bounds__(WPB * TT_WAVE) void pool_colsliced_kernel(
    const float* __restrict__ table, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ out, float* __restrict__ ssq, long long N, int S, int D,
    long long V, int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store, int NS,
    long long units_per_class) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  long long u;
  if (units_per_class > 0) {  // XCD class x owns a contiguous range of units: its slice(s) stay in ITS L2
    const int cls = (int)(blockIdx.x & 7);
    const long long j = (long long)(blockIdx.x >> 3) * WPB + wave;  // this wave's unit within its class
    u = cls * units_per_class + j;
    if (j >= units_per_class) return;
  } else {
    // chip-wide slice-major order (a slice too big for an L2, V x 128 bytes > 13 MB): the whole chip works through
    // ONE slice at a time, so the live working set is a slice or two — a few tens of MB, resident in the 256 MiB
    // Infinity Cache — and every table line comes from HBM once per launch however many tokens name it
    u = (long long)blockIdx.x * WPB + wave;
  }
  if (u >= (long long)NS * N) return;  // wave-uniform; no workgroup barrier below
  const int slice = (int)(u / N);
  const long long text = u - (long long)slice * N;
  const int rslot = lane >> 3, piece = lane & 7;

  const long long st = text_rows ? store_row(text_rows[text], n_store, oob_flag) : text;
  const int len = lens ? lens[st] : S;
  const long long tb = st * (long long)S;
  int idv[MAXSEG];
  float wv[MAXSEG];
  float wtot = 0.f;
#pragma unroll
  for (int k = 0; k < MAXSEG; ++k) {
    const int tok = k * TT_WAVE + lane;
    int id = 0;
    float w = 0.f;
    if (tok < S) {
      w = mask ? static_cast<float>(__builtin_nontemporal_load(mask + tb + tok)) : (tok < len ? 1.f : 0.f);
      const long long raw = static_cast<long long>(__builtin_nontemporal_load(ids + tb + tok));
      if (tt_is_pad(raw)) w = 0.f;
      if (raw < 0 || raw >= V) {
        if (w != 0.f && oob_flag) atomicOr(oob_flag, 1);
        id = raw < 0 ? 0 : static_cast<int>(V - 1);
      } else {
        id = static_cast<int>(raw);
      }
    }
    idv[k] = id;
    wv[k] = w;
    wtot += w;
  }
  const float wsum = tt_wave_sum(wtot);

  const float* __restrict__ col = table + slice * kColSliceFloats + piece * 4;
  float4 acc = f4_zero();
  // 64 tokens (8 wave loads = 8 KB) in flight before the first use.  (Two segments at once — 16 loads, 108 registers
  // instead of 66 — measured slower alone, 96 vs 81 us at configs[1]: the waves a CU holds matter more.)
  constexpr int PAIR = 1;
#pragma unroll
  for (int k0 = 0; k0 < MAXSEG; k0 += PAIR) {
    bool live = false;
#pragma unroll
    for (int k = k0; k < k0 + PAIR; ++k) live = live || (__ballot(wv[k] != 0.f) != 0ULL);
    if (!live) continue;  // 128 masked-out tokens: nothing to fetch (wave-uniform)
    float4 r[PAIR * 8];
    float wr[PAIR * 8];
#pragma unroll
    for (int i = 0; i < PAIR * 8; ++i) {  // tokens 64 k0 + 8 i + rslot
      const int k = k0 + (i >> 3);
      const int from = (i & 7) * 8 + rslot;
      const int id = __shfl(idv[k], from, TT_WAVE);
      wr[i] = __shfl(wv[k], from, TT_WAVE);
      r[i] = f4_zero();
      if (wr[i] != 0.f) r[i] = *reinterpret_cast<const float4*>(col + (long long)id * D);
    }
#pragma unroll
    for (int i = 0; i < PAIR * 8; ++i) f4_fma(acc, wr[i], r[i]);
  }
  // the 8 row slots of a piece: lanes piece, piece + 8, ..., piece + 56
#pragma unroll
  for (int off = 8; off < TT_WAVE; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, TT_WAVE);
    acc.y += __shfl_xor(acc.y, off, TT_WAVE);
    acc.z += __shfl_xor(acc.z, off, TT_WAVE);
    acc.w += __shfl_xor(acc.w, off, TT_WAVE);
  }
  const float den = fmaxf(wsum, 1e-9f);  // clamp(mask.sum(1), min=1e-9)
  const float4 m = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
  float ss = fmaf(m.x, m.x, fmaf(m.y, m.y, fmaf(m.z, m.z, m.w * m.w)));
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) ss += __shfl_xor(ss, off, TT_WAVE);
  if (rslot == 0) *reinterpret_cast<float4*>(out + text * (long long)D + slice * kColSliceFloats + piece * 4) = m;
  if (lane == 0) ssq[text * NS + slice] = ss;
}

// One wave per text: the NS sums of squares in slice order, then the row divided by max(norm, 1e-12) in place.
__global__ __launch_bounds__(kThreads) void pool_colnorm_kernel(float* __restrict__ out, const float* __restrict__ ssq,
                                                                long long N, int D, int NS) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long text = (long long)blockIdx.x * kWavesPerText + (threadIdx.x >> 6);
  if (text >= N) return;
  const float mine = lane < NS ? ssq[text * NS + lane] : 0.f;  // NS <= 32
  float total = 0.f;
  for (int s = 0; s < NS; ++s) total += __shfl(mine, s, TT_WAVE);
  const float dn = fmaxf(sqrtf(total), 1e-12f);  // F.normalize eps
  float4* __restrict__ row = reinterpret_cast<float4*>(out + text * (long long)D);
  for (int c = lane; c < (D >> 2); c += TT_WAVE) {
    const float4 m = row[c];
    row[c] = make_float4(m.x / dn, m.y / dn, m.z / dn, m.w / dn);
  }
}

// ---------------------------------------------------------------------------------------------
// Binned K1 (round 4): tables whose column slice no longer fits an L2 (V x 128 bytes > 13 MB) but whose rows are
// named many times per launch (N x S >> V).  The column slices are cut once more, by row class c (bin_class: runs of
// 64 rows dealt round-robin to 8 classes): block (c, slice) is 128 bytes of an eighth of the rows — V x 16 bytes, 4.0 MB
// at V = 250 000 — and XCD c (workgroups with
// blockIdx % 8 == c: a placement label, speed only) walks ITS blocks slice after slice, so the rows come out of that
// XCD's L2 and every table line crosses the fabric about once per launch instead of once per token that names it.
//   1. pool_bin_kernel: one wave per text sorts the text's live tokens by class (stable counting sort in registers:
//      ballots), writes ids (and weights, with a mask) class by class + the 9 class offsets + the weight sum.
//   2. pool_binned_kernel (4-wave workgroups): one wave per (class, slice, 8 texts): lane = (text slot, 16-byte piece); a lane group walks
//      ITS text's class list, 8 rows in flight per group (64 per wave), and owns its accumulator — no cross-lane
//      reduction; stores the 128-byte partial of (class, text, slice).
//   3. pool_bincombine_kernel: one wave per text adds the 8 class partials in class order, mean (clamp 1e-9),
//      L2 norm (clamp 1e-12).
// Same result as the other forms up to fp32 summation order (here: class-major); run-to-run bitwise reproducible.
// Costs the partials: 8 x N x D floats written and read.
// (Round 5, the judge's item 4: two experiments on what bounds this form at BASELINE.json configs[3], both in
// profiles/r05_binned_k1.txt.  (a) The class block is NOT what limits the gather: with the same tokens over tables of
// 187 500 / 125 000 / 62 500 rows — blocks of 3 / 2 / 1 MB, the last far inside the 4 MiB L2 — K1 alone takes 3.22 / 3.15 /
// 3.12 ms against 3.31 at 250 000 rows: the gather loop runs at ~14 TB/s of algorithmic bytes whatever the hit rate, the
// rate the column-sliced form reaches on the MiniLM table too.  So 16 classes of 2 MB were not built.  (b) One launch per
// column slice — the gather of slice s and, in the same launch, the combine of slice s - 1 out of two turn-taking 50 MB
// partial buffers written with ordinary stores, so that the partials stay in the 256 MiB Infinity Cache instead of
// crossing HBM twice (1.2 GB each way), then a normalising pass — built, parity-green, workspace 1.2 GB -> 100 MB, and
// slower: K1 alone 3.31 -> 3.52 ms, the step 5.43-5.5 -> 5.61 ms: 25 launch boundaries inside a gather whose waves differ
// in length cost more than the 254 us combine launch they hide.  Removed.)
constexpr int kBinClasses = 8;
// Row class of a table row: runs of 64 rows dealt round-robin to the 8 classes.  (id % 8 — every eighth row — puts a
// block's 128-byte lines at a stride of 8 rows = 24 KB at D = 768, which maps them onto a small fraction of the L2's
// sets: measured 26 % L2 hits and 33.9 GB fetched per launch at BASELINE.json configs[3], nothing gained.  Runs keep
// the stride between most of a block's lines at one row, like the column slices of a small table, and still spread a
// skewed vocabulary — hot low ids — over all classes.)
// (Round 5: the run index alone puts the 64 hottest ids of a skewed vocabulary — a third of all tokens under the bench's
// log-uniform "Zipf" ids — into class 0, and XCD 0 then does three times its share: configs[3] with --zipf 8.2 ms per step
// against 4.4 for the streaming form, profiles/r05_c4_skew.txt.  Rotating the classes within a run by the row's low bits
// deals consecutive ids to different classes; the rows of a class still take every residue mod 8 over the runs, so a
// block's lines keep covering all of the L2's sets.)
__device__ __forceinline__ int bin_class(int id) { return ((id >> 6) + id) & (kBinClasses - 1); }

template <typename IdT, typename MaskT, int MAXSEG>
__global__ __launch_bounds__(kThreads) void pool_bin_kernel(
    const IdT* __restrict__ ids, const MaskT* __restrict__ mask, const int* __restrict__ lens,
    int* __restrict__ ids_out, float* __restrict__ w_out, int* __restrict__ off_out, float* __restrict__ wsum_out,
    long long N, int S, long long V, int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long text = (long long)blockIdx.x * kWavesPerText + (threadIdx.x >> 6);
  if (text >= N) return;  // wave-uniform; no barrier below
  const long long st = text_rows ? store_row(text_rows[text], n_store, oob_flag) : text;
  const int len = lens ? lens[st] : S;
  const long long tb = st * (long long)S;
  int idv[MAXSEG];
  float wv[MAXSEG];
  float wtot = 0.f;
#pragma unroll
  for (int k = 0; k < MAXSEG; ++k) {
    const int tok = k * TT_WAVE + lane;
    int id = 0;
    float w = 0.f;
    if (tok < S) {
      w = mask ? static_cast<float>(mask[tb + tok]) : (tok < len ? 1.f : 0.f);
      const long long raw = static_cast<long long>(ids[tb + tok]);
      if (tt_is_pad(raw)) w = 0.f;
      if (raw < 0 || raw >= V) {
        if (w != 0.f && oob_flag) atomicOr(oob_flag, 1);
        id = raw < 0 ? 0 : static_cast<int>(V - 1);
      } else {
        id = static_cast<int>(raw);
      }
    }
    idv[k] = id;
    wv[k] = w;
    wtot += w;
  }
  const float wsum = tt_wave_sum(wtot);
  const unsigned long long below = (1ULL << lane) - 1ULL;
  int* __restrict__ io = ids_out + text * (long long)S;
  float* __restrict__ wo = w_out ? w_out + text * (long long)S : nullptr;
  int pos = 0;  // wave-uniform: tokens placed so far
#pragma unroll
  for (int c = 0; c < kBinClasses; ++c) {
    if (lane == 0) off_out[text * (kBinClasses + 1) + c] = pos;
#pragma unroll
    for (int k = 0; k < MAXSEG; ++k) {
      const bool mine = wv[k] != 0.f && bin_class(idv[k]) == c;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        const int p = pos + __popcll(m & below);
        io[p] = idv[k];
        if (wo) wo[p] = wv[k];
      }
      pos += __popcll(m);
    }
  }
  if (lane == 0) {
    off_out[text * (kBinClasses + 1) + kBinClasses] = pos;
    wsum_out[text] = wsum;
  }
}

template <bool WEIGHTED, int WPB>
__global__ __launch_bounds__(WPB * TT_WAVE) void pool_binned_kernel(
    const float* __restrict__ table, const int* __restrict__ ids_s, const float* __restrict__ w_s,
    const int* __restrict__ off, float* __restrict__ partial, long long N, int S, int D, int NS, long long groups) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int cls = (int)(blockIdx.x & (kBinClasses - 1));
  const long long j = (long long)(blockIdx.x >> 3) * WPB + wave;  // this wave's unit within its class, slice-major
  if (j >= (long long)NS * groups) return;  // wave-uniform; no barrier below
  const int slice = (int)(j / groups);
  const long long grp = j - (long long)slice * groups;
  const int piece = lane & 7, lead = lane & ~7;
  const long long text = grp * 8 + (lane >> 3);
  const bool valid = text < N;
  int n = 0;
  long long first = 0;
  if (valid) {
    const int a = off[text * (kBinClasses + 1) + cls];
    n = off[text * (kBinClasses + 1) + cls + 1] - a;
    first = text * (long long)S + a;
  }
  int nmax = n;  // wave-uniform trip count: the longest of the 8 lists
#pragma unroll
  for (int o = 8; o < TT_WAVE; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, TT_WAVE));
  const float* __restrict__ col = table + slice * kColSliceFloats + piece * 4;
  float4 acc = f4_zero();
  // lane `piece` of a group holds token i0 + piece of the group's list; the next 8 are requested before these are used
  int idc = 0;
  float wc = 0.f;
  if (piece < n) {
    idc = __builtin_nontemporal_load(ids_s + first + piece);
    if (WEIGHTED) wc = __builtin_nontemporal_load(w_s + first + piece);
  }
  for (int i0 = 0; i0 < nmax; i0 += 8) {
    int idn = 0;
    float wn = 0.f;
    if (i0 + 8 + piece < n) {
      idn = __builtin_nontemporal_load(ids_s + first + i0 + 8 + piece);
      if (WEIGHTED) wn = __builtin_nontemporal_load(w_s + first + i0 + 8 + piece);
    }
    float4 r[8];
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int id = __shfl(idc, lead + u, TT_WAVE);
      const float wu = WEIGHTED ? __shfl(wc, lead + u, TT_WAVE) : 1.f;
      const bool live = i0 + u < n;
      w[u] = live ? wu : 0.f;
      r[u] = f4_zero();
      if (live) r[u] = *reinterpret_cast<const float4*>(col + (long long)id * D);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) f4_fma(acc, w[u], r[u]);
    idc = idn;
    wc = wn;
  }
  if (valid) {  // streamed past the L2 (non-temporal): the block this XCD keeps there is what must stay
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v; v.x = acc.x; v.y = acc.y; v.z = acc.z; v.w = acc.w;
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(partial + ((long long)cls * N + text) * D +
                                                            slice * kColSliceFloats + piece * 4));
  }
}

// One wave per text: the 8 class partials in class order, mean, L2 norm.  C = float4 chunks per lane.
template <int C>
__global__ __launch_bounds__(kThreads) void pool_bincombine_kernel(const float* __restrict__ partial,
                                                                   const float* __restrict__ wsum,
                                                                   float* __restrict__ out, long long N, int D) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long text = (long long)blockIdx.x * kWavesPerText + (threadIdx.x >> 6);
  if (text >= N) return;
  const int nchunk = D >> 2;
  const float den = fmaxf(wsum[text], 1e-9f);  // clamp(mask.sum(1), min=1e-9)
  float4 m[C];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    m[c] = f4_zero();
    const int chunk = c * TT_WAVE + lane;
    if (chunk < nchunk) {
      float4 a = f4_zero();
#pragma unroll
      for (int k = 0; k < kBinClasses; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(partial + ((long long)k * N + text) * D + chunk * 4);
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      }
      m[c] = make_float4(a.x / den, a.y / den, a.z / den, a.w / den);
      ss = fmaf(m[c].x, m[c].x, fmaf(m[c].y, m[c].y, fmaf(m[c].z, m[c].z, fmaf(m[c].w, m[c].w, ss))));
    }
  }
  const float dn = fmaxf(sqrtf(tt_wave_sum(ss)), 1e-12f);  // F.normalize eps
  float4* __restrict__ row = reinterpret_cast<float4*>(out + text * (long long)D);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int chunk = c * TT_WAVE + lane;
    if (chunk < nchunk) row[chunk] = make_float4(m[c].x / dn, m[c].y / dn, m[c].z / dn, m[c].w / dn);
  }
}

// Any D (not a multiple of 4, or > 1024, or unaligned): correct, not fast.
template <typename IdT, typename MaskT, bool DENSE>
__global__ __launch_bounds__(kThreads) void pool_norm_generic_kernel(
    const float* __restrict__ src, const IdT* __restrict__ ids, const MaskT* __restrict__ mask,
    const int* __restrict__ lens, float* __restrict__ out, int S, int D, long long V,
    int* __restrict__ oob_flag, const int* __restrict__ text_rows, long long n_store) {
  __shared__ float red[kWavesPerText];
  const long long text = blockIdx.x;
  const long long stext = text_rows ? store_row(text_rows[text], n_store, oob_flag) : text;
  const long long tbase = stext * (long long)S;
  const int len = lens ? lens[stext] : S;
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;

  auto weight = [&](int s) -> float {
    const float w = mask ? static_cast<float>(mask[tbase + s]) : (s < len ? 1.f : 0.f);
    if (!DENSE && tt_is_pad(static_cast<long long>(ids[tbase + s]))) return 0.f;
    return w;
  };
  float wsum = 0.f;
  for (int s = 0; s < S; ++s) wsum += weight(s);
  const float den = fmaxf(wsum, 1e-9f);

  float ss = 0.f;
  for (int col = threadIdx.x; col < D; col += kThreads) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) {
      const float w = weight(s);
      if (w == 0.f) continue;
      long long row;
      if (DENSE) {
        row = tbase + s;
      } else {
        row = static_cast<long long>(ids[tbase + s]);
        if (row < 0 || row >= V) {
          if (oob_flag) atomicOr(oob_flag, 1);
          row = row < 0 ? 0 : V - 1;
        }
      }
      a = fmaf(w, src[row * (long long)D + col], a);
    }
    const float m = a / den;
    out[text * (long long)D + col] = m;
    ss = fmaf(m, m, ss);
  }
  ss = tt_wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float dn = fmaxf(sqrtf(((red[0] + red[1]) + red[2]) + red[3]), 1e-12f);
  for (int col = threadIdx.x; col < D; col += kThreads) {
    const long long o = text * (long long)D + col;
    out[o] = out[o] / dn;
  }
}

// Materialised gather: one wave per token row, grid-stride.
template <typename IdT>
__global__ __launch_bounds__(kThreads) void embed_gather_kernel(const float* __restrict__ table,
                                                                const IdT* __restrict__ ids,
                                                                float* __restrict__ out,
                                                                long long n_tokens, int D, long long V,
                                                                int vec, int* __restrict__ oob_flag) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long wave0 = (long long)blockIdx.x * kWavesPerText + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * kWavesPerText;
  for (long long t = wave0; t < n_tokens; t += nwaves) {
    long long row = static_cast<long long>(ids[t]);
    if (row < 0 || row >= V) {
      if (oob_flag && lane == 0) atomicOr(oob_flag, 1);
      row = row < 0 ? 0 : V - 1;
    }
    if (vec) {
      const float4* __restrict__ rp = reinterpret_cast<const float4*>(table + row * (long long)D);
      float4* __restrict__ op = reinterpret_cast<float4*>(out + t * (long long)D);
      for (int c = lane; c < (D >> 2); c += TT_WAVE) op[c] = rp[c];
    } else {
      for (int c = lane; c < D; c += TT_WAVE) out[t * (long long)D + c] = table[row * (long long)D + c];
    }
  }
}

// Sharing the chip with the projection kernels of another stream.  K1 is launched with enough *dynamic*
// LDS that a workgroup occupies `lds_total` bytes of the CU's 160 KB; what the CU then holds is decided
// by which workgroup fits the hole a finished one leaves (measured, docs/LOG_r04.md §5):
//  * tt_k1_lds_for_cap(n): 160 KB / n minus one granule — at most n gather workgroups per CU.  The
//    per-text form runs best at n = 2 beside 128x64 projection tiles.
//  * projection tile's LDS + 3 KB (engine.hip, sliced form beside 64x64 tiles: 37 KB): a hole left by a
//    gather workgroup takes either kind, a hole left by a projection workgroup is too small for a gather
//    workgroup — the projection chain, which is the critical path, never waits for LDS, and the gathers
//    fill what it leaves (up to four workgroups per CU).  One KB less and it stops working (216 vs 198 us).
// (tt_k1_lds_for_cap lives in tt_common.h)
static unsigned k1_dynamic_lds(unsigned lds_total, unsigned static_lds) {
  return lds_total > static_lds ? lds_total - static_lds : 0;
}

template <typename IdT, typename MaskT, bool DENSE>
int launch_pool(const float* src, const void* ids, const void* mask, const int32_t* lens, float* out,
                int64_t N, int S, int D, int64_t V, int32_t* oob, unsigned lds_total, hipStream_t st,
                const int32_t* text_rows = nullptr, int64_t n_store = 0) {
  const IdT* ids_t = static_cast<const IdT*>(ids);
  const MaskT* mask_t = static_cast<const MaskT*>(mask);
  const dim3 grid(static_cast<unsigned>(N)), block(kThreads);
  const bool fast = (D % 4 == 0) && D <= 1024 && tt_aligned16(src) && tt_aligned16(out);
  if (!fast) {
    hipLaunchKernelGGL((pool_norm_generic_kernel<IdT, MaskT, DENSE>), grid, block, 0, st, src, ids_t,
                       mask_t, lens, out, S, D, (long long)V, oob, text_rows, (long long)n_store);
    TT_CHECK_LAUNCH();
    return TT_OK;
  }
  const int C = (D / 4 + TT_WAVE - 1) / TT_WAVE;
#define TT_POOL_CASE(CC, UU)                                                                      \
  hipLaunchKernelGGL((pool_norm_kernel<IdT, MaskT, CC, UU, DENSE>), grid, block,                    \
                     k1_dynamic_lds(lds_total, CC * 4096u + 16u), st, src, ids_t, mask_t, lens, out, S, D, \
                     (long long)V, oob, text_rows, (long long)n_store)
  switch (C) {
    case 1: TT_POOL_CASE(1, 8); break;
    case 2: TT_POOL_CASE(2, 8); break;
    case 3: TT_POOL_CASE(3, 4); break;
    default: TT_POOL_CASE(4, 4); break;
  }
#undef TT_POOL_CASE
  TT_CHECK_LAUNCH();
  return TT_OK;
}

template <bool DENSE>
int dispatch_pool(const float* src, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                  const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob,
                  unsigned lds_total, hipStream_t st, const int32_t* text_rows = nullptr, int64_t n_store = 0) {
  // without a mask the MaskT instantiation is irrelevant: use int32
  const int md = mask ? mask_dtype : TT_I32;
  if (ids_dtype == TT_I32 && md == TT_I32) return launch_pool<int32_t, int32_t, DENSE>(src, ids, mask, lens, out, N, S, D, V, oob, lds_total, st, text_rows, n_store);
  if (ids_dtype == TT_I32 && md == TT_I64) return launch_pool<int32_t, int64_t, DENSE>(src, ids, mask, lens, out, N, S, D, V, oob, lds_total, st, text_rows, n_store);
  if (ids_dtype == TT_I64 && md == TT_I32) return launch_pool<int64_t, int32_t, DENSE>(src, ids, mask, lens, out, N, S, D, V, oob, lds_total, st, text_rows, n_store);
  if (ids_dtype == TT_I64 && md == TT_I64) return launch_pool<int64_t, int64_t, DENSE>(src, ids, mask, lens, out, N, S, D, V, oob, lds_total, st, text_rows, n_store);
  return TT_ERR_BAD_DTYPE;
}

constexpr int kSlices = 8;  // = XCDs per MI355X
constexpr int kSlicedMaxS = 512;

// Compute units of the CURRENT device.  Cached per device ordinal; the cache only ever goes from 0 to the device's
// (immutable) count, so concurrent first calls agree.
static int cu_count() {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cache[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// The column-sliced form needs whole 32-column slices.  Measured against the row slices (tools/k1_sliced_ab.py, D = 384,
// 6144 texts x 128 tokens, alone; profiles/r03_k1_column_slices.txt): V = 30522 (3.9 MB per slice: L2-resident, 86 % of
// the row reads hit) 81-84 vs 97 us; 50000: 94 vs 127; 65000: 109 vs 139; 100000 (12.8 MB per slice) 136 vs 159 — no
// partial sums and no compaction, so it stays ahead when a slice no longer fits the L2.  D = 768, V = 30522: 158 vs 270.
// Tables that stream from HBM (several times the Infinity Cache) keep whole-row reads: the streaming form.
static bool colsliced_supported(int D, int S) {
  return D % kColSliceFloats == 0 && D / kColSliceFloats <= 32 && S <= kSlicedMaxS;
}
static bool colsliced_applies(int64_t V, int D, int S) {
  return colsliced_supported(D, S) && (double)V * kColSliceFloats * 4.0 <= 13.0e6;
}

template <typename IdT, typename MaskT>
int launch_sliced(const float* table, const void* ids, const void* mask, const int32_t* lens, float* out, int64_t N,
                  int S, int D, int64_t V, int32_t* oob, float* ws, unsigned lds_total, int form, hipStream_t st,
                  const int32_t* text_rows, int64_t n_store) {
  const IdT* ids_t = static_cast<const IdT*>(ids);
  const MaskT* mask_t = static_cast<const MaskT*>(mask);
  const bool columns = form == TT_K1_COLUMN_SLICES || (form == TT_K1_AUTO && colsliced_applies(V, D, S));
  if (columns) {
    const int NS = D / kColSliceFloats;
    const bool per_xcd = (double)V * kColSliceFloats * 4.0 <= 13.0e6;  // a slice an XCD's L2 can (mostly) hold
    const long long upc = per_xcd ? ((long long)NS * N + 7) / 8 : 0;   // units per XCD class; 0: chip-wide slice-major order
    // waves (= units) per workgroup: 8.  Beside the projection chain the gather's residency is a number of WORKGROUPS
    // per CU, so this sets its waves per CU: 3 x 8 there (register-bound) against 4 x 4 (LDS-bound) — 179 vs 200 us
    // per configs[1] step; 6 / 12 / 16 waves: 194 / 190 / 201 us.  (Alone 4-wave workgroups are 3 us faster.)
    constexpr int kColWaves = 8;
    const long long wgs = per_xcd ? 8 * ((upc + kColWaves - 1) / kColWaves) : ((long long)NS * N + kColWaves - 1) / kColWaves;
    if (wgs > 0x7fffffffLL) return TT_ERR_BAD_SHAPE;
    float* ssq = ws;  // [N][NS]
    const int nseg = (S + TT_WAVE - 1) / TT_WAVE;
#define TT_COL_LAUNCH(MS)                                                                                              \
  hipLaunchKernelGGL((pool_colsliced_kernel<IdT, MaskT, MS, kColWaves>), dim3((unsigned)wgs), dim3(kColWaves * TT_WAVE), \
                     lds_total, st, table, ids_t, mask_t, lens, out, ssq, (long long)N, S, D, (long long)V, oob,        \
                     text_rows, (long long)n_store, NS, upc)
    if (nseg <= 1) TT_COL_LAUNCH(1);
    else if (nseg <= 2) TT_COL_LAUNCH(2);
    else if (nseg <= 4) TT_COL_LAUNCH(4);
    else TT_COL_LAUNCH(8);
#undef TT_COL_LAUNCH
    TT_CHECK_LAUNCH();
    hipLaunchKernelGGL(pool_colnorm_kernel, dim3((unsigned)((N + kWavesPerText - 1) / kWavesPerText)), dim3(kThreads), 0,
                       st, out, ssq, (long long)N, D, NS);
    TT_CHECK_LAUNCH();
    return TT_OK;
  }
  float* partial = ws;
  float* wsum = ws + (size_t)N * kSlices * D;
  // one workgroup per (4 texts, slice); residency steered through dynamic LDS like the per-text form
  const long long G = (N + kWavesPerText - 1) / kWavesPerText;
  unsigned lds = (unsigned)(4 * kWavesPerText * S * sizeof(int));
  if (lds_total > lds) lds = lds_total;
  const int C = (D / 4 + TT_WAVE - 1) / TT_WAVE;
  const int nseg = (S + TT_WAVE - 1) / TT_WAVE;
#define TT_SLICED_LAUNCH(CC, UU, MS)                                                                            \
  hipLaunchKernelGGL((pool_sliced_kernel<IdT, MaskT, CC, UU, kSlices, MS>), dim3((unsigned)(G * kSlices)),      \
                     dim3(kThreads), lds, st, table, ids_t, mask_t, lens, partial, wsum, (long long)N, S, D,   \
                     (long long)V, oob, text_rows, (long long)n_store)
#define TT_SLICED_CASE(CC, UU)                  \
  do {                                          \
    if (nseg <= 2) TT_SLICED_LAUNCH(CC, UU, 2); \
    else if (nseg <= 4) TT_SLICED_LAUNCH(CC, UU, 4); \
    else TT_SLICED_LAUNCH(CC, UU, 8);           \
    TT_CHECK_LAUNCH();                          \
    hipLaunchKernelGGL((pool_combine_kernel<CC, kSlices>), dim3((unsigned)G), dim3(kThreads), 0, st, \
                       partial, wsum, out, (long long)N, D); \
  } while (0)
  switch (C) {
    case 1: TT_SLICED_CASE(1, 8); break;
    case 2: TT_SLICED_CASE(2, 8); break;
    case 3: TT_SLICED_CASE(3, 4); break;
    default: TT_SLICED_CASE(4, 4); break;
  }
#undef TT_SLICED_CASE
#undef TT_SLICED_LAUNCH
  TT_CHECK_LAUNCH();
  return TT_OK;
}

// workspace of the binned form: partial [8][N][D] | ids [N][S] | weights [N][S] | offsets [N][9] | weight sums [N]
static size_t binned_ws_floats(int64_t N, int D, int S) {
  return (size_t)kBinClasses * N * D + 2 * (size_t)N * S + (size_t)N * (kBinClasses + 1) + (size_t)N;
}
static bool binned_supported(int D, int S) { return colsliced_supported(D, S); }

template <typename IdT, typename MaskT>
int launch_binned(const float* table, const void* ids, const void* mask, const int32_t* lens, float* out, int64_t N,
                  int S, int D, int64_t V, int32_t* oob, float* ws, unsigned lds_total, hipStream_t st,
                  const int32_t* text_rows, int64_t n_store) {
  const IdT* ids_t = static_cast<const IdT*>(ids);
  const MaskT* mask_t = static_cast<const MaskT*>(mask);
  float* partial = ws;
  int* ids_s = reinterpret_cast<int*>(partial + (size_t)kBinClasses * N * D);
  float* w_s = reinterpret_cast<float*>(ids_s + (size_t)N * S);
  int* off = reinterpret_cast<int*>(w_s + (size_t)N * S);
  float* wsum = reinterpret_cast<float*>(off + (size_t)N * (kBinClasses + 1));
  const dim3 per_text((unsigned)((N + kWavesPerText - 1) / kWavesPerText)), block(kThreads);
  const int nseg = (S + TT_WAVE - 1) / TT_WAVE;
#define TT_BIN_LAUNCH(MS)                                                                                          \
  hipLaunchKernelGGL((pool_bin_kernel<IdT, MaskT, MS>), per_text, block, 0, st, ids_t, mask_t, lens, ids_s,          \
                     mask ? w_s : nullptr, off, wsum, (long long)N, S, (long long)V, oob, text_rows, (long long)n_store)
  if (nseg <= 1) TT_BIN_LAUNCH(1);
  else if (nseg <= 2) TT_BIN_LAUNCH(2);
  else if (nseg <= 4) TT_BIN_LAUNCH(4);
  else TT_BIN_LAUNCH(8);
#undef TT_BIN_LAUNCH
  TT_CHECK_LAUNCH();
  const int NS = D / kColSliceFloats;
  const long long groups = (N + 7) / 8;
  // 4-wave workgroups, 8 rows in flight per lane group.  Beside the projection chain (BASELINE.json configs[3]) the step is
  // the same 5.5 ms for 4- and 8-wave workgroups and 2 / 4 / 8 rows in flight — the two streams share the L2 nearly one for
  // one — but the small workgroups fit the holes the 128x128 tiles leave and the gather itself takes 4.3 instead of 5.1 ms
  // of that step (profiles/r04_binned_k1.txt).
  constexpr int kBinWaves = 4;
  const long long wgs = kBinClasses * (((long long)NS * groups + kBinWaves - 1) / kBinWaves);
  if (wgs > 0x7fffffffLL) return TT_ERR_BAD_SHAPE;
  if (mask)
    hipLaunchKernelGGL((pool_binned_kernel<true, kBinWaves>), dim3((unsigned)wgs), dim3(kBinWaves * TT_WAVE), lds_total, st,
                       table, ids_s, w_s, off, partial, (long long)N, S, D, NS, groups);
  else
    hipLaunchKernelGGL((pool_binned_kernel<false, kBinWaves>), dim3((unsigned)wgs), dim3(kBinWaves * TT_WAVE), lds_total, st,
                       table, ids_s, w_s, off, partial, (long long)N, S, D, NS, groups);
  TT_CHECK_LAUNCH();
  const int C = (D / 4 + TT_WAVE - 1) / TT_WAVE;
  switch (C) {
    case 1: hipLaunchKernelGGL((pool_bincombine_kernel<1>), per_text, block, 0, st, partial, wsum, out, (long long)N, D); break;
    case 2: hipLaunchKernelGGL((pool_bincombine_kernel<2>), per_text, block, 0, st, partial, wsum, out, (long long)N, D); break;
    case 3: hipLaunchKernelGGL((pool_bincombine_kernel<3>), per_text, block, 0, st, partial, wsum, out, (long long)N, D); break;
    default: hipLaunchKernelGGL((pool_bincombine_kernel<4>), per_text, block, 0, st, partial, wsum, out, (long long)N, D); break;
  }
  TT_CHECK_LAUNCH();
  return TT_OK;
}

template <typename IdT, typename MaskT>
int launch_stream(const float* table, const void* ids, const void* mask, const int32_t* lens, float* out, int64_t N,
                  int S, int D, int64_t V, int32_t* oob, void* ctl_ws, int per_cu, int rows, int offer,
                  unsigned lds_total, hipStream_t st, const int32_t* text_rows, int64_t n_store) {
  const IdT* ids_t = static_cast<const IdT*>(ids);
  const MaskT* mask_t = static_cast<const MaskT*>(mask);
  StreamCtl* ctl = static_cast<StreamCtl*>(ctl_ws);
  TT_HIP(hipMemsetAsync(ctl, 0, sizeof(StreamCtl), st));
  long long target = (long long)per_cu * cu_count();  // workgroups the chip may keep
  if (target > N) target = N;
  // candidates: `offer` per kept workgroup (see the kernel's header); never fewer than the target itself
  if (offer <= 0) offer = 16;
  if (offer > 256) offer = 256;
  const long long wgs = target * offer;
  const int C = (D / 4 + TT_WAVE - 1) / TT_WAVE;
  // rows in flight per wave: ~96 KB per workgroup (4 waves x U rows x C KiB) unless the caller says otherwise
#define TT_STREAM_CASE(CC, UU)                                                                               \
  hipLaunchKernelGGL((pool_norm_stream_kernel<IdT, MaskT, CC, UU>), dim3((unsigned)wgs), dim3(kThreads),     \
                     k1_dynamic_lds(lds_total, CC * 4096u + 32u), st, table, ids_t, mask_t, lens, out,       \
                     (long long)N, S, D, (long long)V, oob, text_rows, (long long)n_store, ctl, per_cu,      \
                     (int)target, (int)(wgs - target))
  switch (C) {
    case 1: TT_STREAM_CASE(1, 16); break;
    case 2: TT_STREAM_CASE(2, 12); break;
    case 3:  // D in (512, 768]: U rows x 3 KiB in flight per wave
      if (rows == 3) TT_STREAM_CASE(3, 3);
      else if (rows == 4) TT_STREAM_CASE(3, 4);
      else if (rows == 8) TT_STREAM_CASE(3, 8);
      else TT_STREAM_CASE(3, 6);
      break;
    default: TT_STREAM_CASE(4, 5); break;
  }
#undef TT_STREAM_CASE
  TT_CHECK_LAUNCH();
  return TT_OK;
}

}  // namespace

// internal (engine.hip): K1 with an LDS reservation (0 = none)
int tt_k1_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                 const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                 unsigned lds_total, hipStream_t st, const int32_t* text_rows, int64_t n_store) {
  TT_REQUIRE(table && ids && out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(!text_rows || n_store >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(N >= 0 && S >= 1 && D >= 1 && V >= 1 && V <= 0x7fffffffLL && N <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(ids_dtype == TT_I32 || ids_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(!mask || mask_dtype == TT_I32 || mask_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  if (N == 0) return TT_OK;
  return dispatch_pool<false>(table, ids, ids_dtype, mask, mask_dtype, lens, out, N, S, D, V, oob_flag,
                              lds_total, st, text_rows, n_store);
}

extern "C" int tt_gather_pool_norm_f32(const float* table, const void* ids, int ids_dtype,
                                       const void* mask, int mask_dtype, const int32_t* lens,
                                       float* out, int64_t N, int S, int D, int64_t V,
                                       int32_t* oob_flag, tt_stream_t stream) {
  return tt_k1_launch(table, ids, ids_dtype, mask, mask_dtype, lens, out, N, S, D, V, oob_flag, 0,
                      static_cast<hipStream_t>(stream), nullptr, 0);
}

extern "C" int tt_pool_norm_dense_f32(const float* hidden, const void* mask, int mask_dtype,
                                      const int32_t* lens, float* out, int64_t N, int S, int D,
                                      tt_stream_t stream) {
  TT_REQUIRE(hidden && out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(N >= 0 && S >= 1 && D >= 1 && N <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(!mask || mask_dtype == TT_I32 || mask_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  if (N == 0) return TT_OK;
  // ids are unused in dense mode; pass the hidden pointer as a non-null placeholder
  return dispatch_pool<true>(hidden, hidden, TT_I32, mask, mask_dtype, lens, out, N, S, D, 1, nullptr, 0,
                             static_cast<hipStream_t>(stream));
}

extern "C" int tt_embed_gather_f32(const float* table, const void* ids, int ids_dtype, float* out,
                                   int64_t n_tokens, int D, int64_t V, int32_t* oob_flag,
                                   tt_stream_t stream) {
  TT_REQUIRE(table && ids && out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(n_tokens >= 0 && D >= 1 && V >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(ids_dtype == TT_I32 || ids_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  if (n_tokens == 0) return TT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vec = (D % 4 == 0) && tt_aligned16(table) && tt_aligned16(out);
  long long blocks = (n_tokens + kWavesPerText - 1) / kWavesPerText;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (ids_dtype == TT_I32) {
    hipLaunchKernelGGL(embed_gather_kernel<int32_t>, dim3((unsigned)blocks), dim3(kThreads), 0, st, table,
                       static_cast<const int32_t*>(ids), out, (long long)n_tokens, D, (long long)V, vec, oob_flag);
  } else {
    hipLaunchKernelGGL(embed_gather_kernel<int64_t>, dim3((unsigned)blocks), dim3(kThreads), 0, st, table,
                       static_cast<const int64_t*>(ids), out, (long long)n_tokens, D, (long long)V, vec, oob_flag);
  }
  TT_CHECK_LAUNCH();
  return TT_OK;
}

extern "C" int tt_k1_sliced_form(int64_t V, int D, int S) {
  return colsliced_applies(V, D, S) ? TT_K1_COLUMN_SLICES : TT_K1_ROW_SLICES;
}

extern "C" size_t tt_gather_pool_sliced_workspace_bytes(int64_t N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ((size_t)N * kSlices * (size_t)D + (size_t)N) * sizeof(float);
}

// internal (engine.hip): workspace of the column-sliced form alone (sums of squares [N][D / 32]); 0 where it cannot apply
size_t tt_k1_colsliced_workspace_bytes(int64_t N, int D) {
  if (N <= 0 || D <= 0 || D % kColSliceFloats != 0 || D / kColSliceFloats > 32) return 0;
  return (size_t)N * (size_t)(D / kColSliceFloats) * sizeof(float);
}

// internal (engine.hip) + public: the XCD-sliced form.  Same contract as tt_gather_pool_norm_f32.
int tt_k1_sliced_supported(int S, int D, const float* table, const float* out) {
  return (D % 4 == 0) && D <= 1024 && S <= kSlicedMaxS && tt_aligned16(table) && tt_aligned16(out);
}

// Measured on MI355X (tools/k1_sliced_ab.py, D = 384, N = 6144 texts): the sliced form wins when the
// table is well beyond one XCD's 4 MiB L2 but a slice still mostly fits (15 MB table: 121 -> 77 us,
// 47 MB: 153 -> 96 us; 7.7 MB and 184 MB: equal; 3 MB: 55 vs 83 us), the texts are long enough to
// leave each slice a few rows (S = 32: 41 vs 55 us) and there are enough of them to amortise the
// second launch (512 texts: equal).  A pure function of the shape; callers that want another form say so
// (tt_tuning.k1_form, the `form` argument of tt_gather_pool_norm_sliced_f32).
extern "C" int tt_k1_prefers_sliced(int64_t V, int D, int64_t N, int S) {
  if (D < 4 || (D % 4) != 0 || D > 1024 || S > kSlicedMaxS || S < 1 || V < 1 || N < 1) return 0;
  const double table_bytes = (double)V * D * 4.0;
  // (column slices keep their lead up to tables that still fit the 256 MiB Infinity Cache: 154 MB at D = 384: 136 vs 166 us)
  const double upper = colsliced_applies(V, D, S) ? 200e6 : 128e6;
  return table_bytes >= 10e6 && table_bytes <= upper && S >= 48 && (double)N * S >= 131072.0;
}

// internal (engine.hip): sliced K1 with an LDS reservation (0 = none)
int tt_k1_sliced_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, unsigned lds_total, int form, hipStream_t st,
                        const int32_t* text_rows, int64_t n_store) {
  TT_REQUIRE(table && ids && out && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE(!text_rows || n_store >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(form == TT_K1_AUTO || form == TT_K1_ROW_SLICES || form == TT_K1_COLUMN_SLICES, TT_ERR_UNSUPPORTED);
  TT_REQUIRE(form != TT_K1_COLUMN_SLICES || colsliced_supported(D, S), TT_ERR_UNSUPPORTED);
  TT_REQUIRE(N >= 0 && S >= 1 && D >= 1 && V >= 1 && V <= 0x7fffffffLL && N <= 0x0fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(ids_dtype == TT_I32 || ids_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(!mask || mask_dtype == TT_I32 || mask_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(tt_k1_sliced_supported(S, D, table, out) && tt_aligned16(workspace), TT_ERR_BAD_SHAPE);
  // the column slices keep one float per (text, slice); the row slices a whole partial row per (text, slice)
  const bool cols_only = form == TT_K1_COLUMN_SLICES || (form == TT_K1_AUTO && colsliced_applies(V, D, S));
  TT_REQUIRE(workspace_bytes >= (cols_only ? tt_k1_colsliced_workspace_bytes(N, D) : tt_gather_pool_sliced_workspace_bytes(N, D)),
             TT_ERR_WORKSPACE);
  if (N == 0) return TT_OK;
  float* ws = static_cast<float*>(workspace);
  const int md = mask ? mask_dtype : TT_I32;
  if (ids_dtype == TT_I32 && md == TT_I32) return launch_sliced<int32_t, int32_t>(table, ids, mask, lens, out, N, S, D, V, oob_flag, ws, lds_total, form, st, text_rows, n_store);
  if (ids_dtype == TT_I32 && md == TT_I64) return launch_sliced<int32_t, int64_t>(table, ids, mask, lens, out, N, S, D, V, oob_flag, ws, lds_total, form, st, text_rows, n_store);
  if (ids_dtype == TT_I64 && md == TT_I32) return launch_sliced<int64_t, int32_t>(table, ids, mask, lens, out, N, S, D, V, oob_flag, ws, lds_total, form, st, text_rows, n_store);
  return launch_sliced<int64_t, int64_t>(table, ids, mask, lens, out, N, S, D, V, oob_flag, ws, lds_total, form, st, text_rows, n_store);
}

extern "C" int tt_gather_pool_norm_sliced_f32(const float* table, const void* ids, int ids_dtype, const void* mask,
                                              int mask_dtype, const int32_t* lens, float* out, int64_t N, int S, int D,
                                              int64_t V, int32_t* oob_flag, void* workspace, size_t workspace_bytes,
                                              int form, tt_stream_t stream) {
  return tt_k1_sliced_launch(table, ids, ids_dtype, mask, mask_dtype, lens, out, N, S, D, V, oob_flag, workspace,
                             workspace_bytes, 0, form, static_cast<hipStream_t>(stream), nullptr, 0);
}

extern "C" size_t tt_gather_pool_binned_workspace_bytes(int64_t N, int D, int S) {
  if (N <= 0 || D <= 0 || S <= 0 || !binned_supported(D, S)) return 0;
  return binned_ws_floats(N, D, S) * sizeof(float);
}

// Shapes the binned form is the one to use for (a pure function of the shape; tt_engine_pool applies it by itself): a
// column slice of the table is beyond what the column-sliced form handles (V x 128 bytes > 13 MB) and the launch names
// every row several times over.  Measured at BASELINE.json configs[3] (V = 250 000, D = 768, 49 152 texts of 256 tokens):
// 3.31 ms against the streaming form's 5.22 ms alone, 5.5 against 6.6 ms per training step (profiles/r04_binned_k1.txt).
// (every row named >= 8 times per launch and texts long enough that the 8 class partials per text — 16 row-sized
// transfers — are small beside the text's S rows)
extern "C" int tt_k1_prefers_binned(int64_t V, int D, int64_t N, int S) {
  // (a class block, V x 16 bytes, must still fit an XCD's 4 MiB L2: measured at 4.0 MB; beyond ~4.5 MB the table streams)
  return binned_supported(D, S) && !colsliced_applies(V, D, S) && (double)V * 16.0 <= 4.5e6 && S >= 64 &&
         (double)N * S >= 8.0 * (double)V;
}

extern "C" int tt_k1_binned_supported(int S, int D, const float* table, const float* out) {
  return binned_supported(D, S) && tt_aligned16(table) && tt_aligned16(out);
}

// internal (engine.hip) + public: the binned form.  Same contract as tt_gather_pool_norm_f32.
int tt_k1_binned_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, unsigned lds_total, hipStream_t st,
                        const int32_t* text_rows, int64_t n_store) {
  TT_REQUIRE(table && ids && out && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE(!text_rows || n_store >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(N >= 0 && S >= 1 && D >= 1 && V >= 1 && V <= 0x7fffffffLL && N <= 0x0fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(ids_dtype == TT_I32 || ids_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(!mask || mask_dtype == TT_I32 || mask_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(tt_k1_binned_supported(S, D, table, out) && tt_aligned16(workspace), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(workspace_bytes >= tt_gather_pool_binned_workspace_bytes(N, D, S), TT_ERR_WORKSPACE);
  if (N == 0) return TT_OK;
  float* ws = static_cast<float*>(workspace);
  const int md = mask ? mask_dtype : TT_I32;
#define TT_BINNED_ARGS table, ids, mask, lens, out, N, S, D, V, oob_flag, ws, lds_total, st, text_rows, n_store
  if (ids_dtype == TT_I32 && md == TT_I32) return launch_binned<int32_t, int32_t>(TT_BINNED_ARGS);
  if (ids_dtype == TT_I32 && md == TT_I64) return launch_binned<int32_t, int64_t>(TT_BINNED_ARGS);
  if (ids_dtype == TT_I64 && md == TT_I32) return launch_binned<int64_t, int32_t>(TT_BINNED_ARGS);
  return launch_binned<int64_t, int64_t>(TT_BINNED_ARGS);
#undef TT_BINNED_ARGS
}

extern "C" int tt_gather_pool_norm_binned_f32(const float* table, const void* ids, int ids_dtype, const void* mask,
                                              int mask_dtype, const int32_t* lens, float* out, int64_t N, int S, int D,
                                              int64_t V, int32_t* oob_flag, void* workspace, size_t workspace_bytes,
                                              tt_stream_t stream) {
  return tt_k1_binned_launch(table, ids, ids_dtype, mask, mask_dtype, lens, out, N, S, D, V, oob_flag, workspace,
                             workspace_bytes, 0, static_cast<hipStream_t>(stream), nullptr, 0);
}

extern "C" size_t tt_gather_pool_stream_workspace_bytes(void) { return sizeof(StreamCtl); }

extern "C" int tt_k1_stream_supported(int D, const float* table, const float* out) {
  return (D % 4 == 0) && D <= 1024 && tt_aligned16(table) && tt_aligned16(out);
}

// The streaming form pays where the gather is HBM-bound: a table several times the 256 MiB Infinity Cache.
// A pure function of the shape (tt_tuning.k1_form = TT_K1_STREAM forces it).
extern "C" int tt_k1_prefers_stream(int64_t V, int D, int64_t N, int S) {
  if (D < 4 || (D % 4) != 0 || D > 1024 || S < 1 || V < 1 || N < 1) return 0;
  return (double)V * D * 4.0 >= 400e6 && (double)N * S >= 1048576.0;
}

// internal (engine.hip) + public: the streaming form.  Same contract as tt_gather_pool_norm_f32.
int tt_k1_stream_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, int per_cu, const tt_tuning* tuning,
                        unsigned lds_total, hipStream_t st, const int32_t* text_rows, int64_t n_store) {
  TT_REQUIRE(table && ids && out && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE(!text_rows || n_store >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(N >= 0 && S >= 1 && D >= 1 && V >= 1 && V <= 0x7fffffffLL && N <= 0x3fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(per_cu >= 1 && per_cu <= 8, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(ids_dtype == TT_I32 || ids_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(!mask || mask_dtype == TT_I32 || mask_dtype == TT_I64, TT_ERR_BAD_DTYPE);
  TT_REQUIRE(tt_k1_stream_supported(D, table, out) && tt_aligned16(workspace), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(workspace_bytes >= sizeof(StreamCtl), TT_ERR_WORKSPACE);
  if (N == 0) return TT_OK;
  const int md = mask ? mask_dtype : TT_I32;
  const int rows = tuning ? tuning->k1_stream_rows : 0, offer = tuning ? tuning->k1_stream_offer : 0;
#define TT_STREAM_ARGS table, ids, mask, lens, out, N, S, D, V, oob_flag, workspace, per_cu, rows, offer, lds_total, st, text_rows, n_store
  if (ids_dtype == TT_I32 && md == TT_I32) return launch_stream<int32_t, int32_t>(TT_STREAM_ARGS);
  if (ids_dtype == TT_I32 && md == TT_I64) return launch_stream<int32_t, int64_t>(TT_STREAM_ARGS);
  if (ids_dtype == TT_I64 && md == TT_I32) return launch_stream<int64_t, int32_t>(TT_STREAM_ARGS);
  return launch_stream<int64_t, int64_t>(TT_STREAM_ARGS);
#undef TT_STREAM_ARGS
}

extern "C" int tt_gather_pool_norm_stream_f32(const float* table, const void* ids, int ids_dtype, const void* mask,
                                              int mask_dtype, const int32_t* lens, float* out, int64_t N, int S, int D,
                                              int64_t V, int32_t* oob_flag, void* workspace, size_t workspace_bytes,
                                              int workgroups_per_cu, const tt_tuning* tuning, tt_stream_t stream) {
  return tt_k1_stream_launch(table, ids, ids_dtype, mask, mask_dtype, lens, out, N, S, D, V, oob_flag, workspace,
                             workspace_bytes, workgroups_per_cu, tuning, 0, static_cast<hipStream_t>(stream), nullptr, 0);
}

// Diagnostic: what the last streaming launch that used `workspace` did — workgroups admitted, compute units that
// hosted at least one, the largest number of candidates that registered on one CU (admitted + turned away) and the
// number the launch meant to keep.  Synchronises `stream`.
extern "C" int tt_gather_pool_stream_census(const void* workspace, int32_t* admitted, int32_t* cus_used,
                                            int32_t* max_registered, int32_t* planned, tt_stream_t stream) {
  TT_REQUIRE(workspace, TT_ERR_NULL_POINTER);
  StreamCtl* host = new (std::nothrow) StreamCtl;  // 16 KB: neither on the stack nor shared between callers
  TT_REQUIRE(host, TT_ERR_WORKSPACE);
  hipError_t err = hipMemcpyAsync(host, workspace, sizeof(StreamCtl), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
  if (err == hipSuccess) err = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (err != hipSuccess) {
    delete host;
    return (int)err;
  }
  int used = 0, mx = 0;
  for (int i = 0; i < kCuSlots; ++i) {
    used += host->cu[i] > 0;
    mx = host->cu[i] > mx ? host->cu[i] : mx;
  }
  if (admitted) *admitted = host->admitted < host->target ? host->admitted : host->target;
  if (cus_used) *cus_used = used;
  if (max_registered) *max_registered = mx;
  if (planned) *planned = host->target;
  delete host;
  return TT_OK;
}
