// f2 — device-side retrieval evaluation: cosine-similarity matrix (MFMA) and NDCG@k.
// Replaces the per-query arithmetic of backend/training.py:281-311 (evaluate_model):
//   similarities = torch.cosine_similarity(query_embed.cpu(), doc_embeds.cpu(), dim=1)
//   ndcg_score(true_relevance, similarities, k=10 | 5 | 1)          (sklearn)
// for ALL queries at once, without the .cpu() round trips.
#include "gemm_f32.h"

namespace {

constexpr int kThreads = 256;

// rows -> unit rows, x / max(|x|, eps): the first half of F.cosine_similarity (eps 1e-8)
__global__ __launch_bounds__(kThreads) void unit_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             long long rows, int P, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* __restrict__ xr = x + row * P;
  float ss = 0.f;
  for (int c = lane; c < P; c += 64) ss = fmaf(xr[c], xr[c], ss);
  const float den = fmaxf(sqrtf(tt_wave_sum(ss)), eps);
  for (int c = lane; c < P; c += 64) y[row * P + c] = xr[c] / den;
}

// Block-wide (max value < bound, its multiplicity, the gain summed over its occurrences).
struct Group {
  float v;
  float cnt;
  float gain;
};
__device__ __forceinline__ Group merge(Group a, Group b) {
  if (a.v > b.v) return a;
  if (b.v > a.v) return b;
  return Group{a.v, a.cnt + b.cnt, a.gain + b.gain};
}

// One workgroup per query row.  sklearn.metrics.ndcg_score(y_true, y_score, k, ignore_ties=False):
// DCG with tied scores sharing the average gain of their group (_tie_averaged_dcg), divided by
// the ideal DCG (gains sorted descending).  Walks the distinct score values from the top: each pass
// finds the next lower distinct score, how many documents carry it and their summed gain; the
// group occupies ranks [pos, pos+cnt) and contributes mean_gain * sum of the discounts 1/log2(r+2)
// of its ranks below k.  At most k passes.  The same walk over the gains themselves gives the IDCG.
// `valid` (optional, per element) restricts a row to its candidate pool.
__global__ __launch_bounds__(kThreads) void ndcg_kernel(const float* __restrict__ scores,
                                                        const float* __restrict__ gains,
                                                        const unsigned char* __restrict__ valid, long long N,
                                                        const int* __restrict__ ks, int nk,
                                                        float* __restrict__ out) {
  __shared__ Group red[kThreads / 64];
  __shared__ Group result;
  const long long row = blockIdx.x;
  const float* __restrict__ s = scores + row * N;
  const float* __restrict__ g = gains + row * N;
  const unsigned char* __restrict__ vm = valid ? valid + row * N : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int kmax = 0;
  for (int i = 0; i < nk; ++i) kmax = max(kmax, ks[i]);

  float dcg[8], idcg[8];
  for (int i = 0; i < 8; ++i) dcg[i] = idcg[i] = 0.f;

  for (int which = 0; which < 2; ++which) {  // 0: ranking by score -> DCG; 1: ranking by gain -> IDCG
    const float* __restrict__ key = which == 0 ? s : g;
    float bound = INFINITY;
    bool first = true;
    long long pos = 0;
    while (pos < kmax) {
      Group best{-INFINITY, 0.f, 0.f};
      for (long long j = threadIdx.x; j < N; j += kThreads) {
        if (vm && !vm[j]) continue;
        const float v = key[j];
        if (!(first || v < bound)) continue;
        if (v > best.v) best = Group{v, 1.f, g[j]};
        else if (v == best.v) { best.cnt += 1.f; best.gain += g[j]; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        Group o{__shfl_xor(best.v, off, 64), __shfl_xor(best.cnt, off, 64), __shfl_xor(best.gain, off, 64)};
        best = merge(best, o);
      }
      if (lane == 0) red[wave] = best;
      __syncthreads();
      if (threadIdx.x == 0) result = merge(merge(red[0], red[1]), merge(red[2], red[3]));
      __syncthreads();
      const Group r = result;
      __syncthreads();
      if (r.cnt == 0.f) break;  // ran out of documents
      const long long cnt = (long long)r.cnt;
      const float mean_gain = r.gain / r.cnt;
      for (int i = 0; i < nk && i < 8; ++i) {
        float dsum = 0.f;
        for (long long rnk = pos; rnk < pos + cnt && rnk < ks[i]; ++rnk) dsum += 1.f / log2f((float)rnk + 2.f);
        if (which == 0) dcg[i] += mean_gain * dsum; else idcg[i] += mean_gain * dsum;
      }
      pos += cnt;
      bound = r.v;
      first = false;
    }
  }
  if (threadIdx.x == 0)
    for (int i = 0; i < nk && i < 8; ++i) out[row * nk + i] = idcg[i] > 0.f ? dcg[i] / idcg[i] : 0.f;
}

}  // namespace

extern "C" size_t tt_cosine_scores_workspace_bytes(int64_t Q, int64_t N, int P) {
  return (size_t)((Q + N) * P + 4) * sizeof(float);
}

extern "C" int tt_cosine_scores_f32(const float* q, const float* d, float* out, int64_t Q, int64_t N, int P,
                                    void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(q && d && out && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE(Q >= 1 && N >= 1 && P >= 1 && Q <= 0x7fffffffLL && N <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(workspace_bytes >= tt_cosine_scores_workspace_bytes(Q, N, P), TT_ERR_WORKSPACE);
  TT_REQUIRE(tt_aligned16(workspace), TT_ERR_ALIGNMENT);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* qn = static_cast<float*>(workspace);
  float* dn = qn + ((Q * P + 3) / 4) * 4;
  hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(kThreads), 0, st, q, qn, (long long)Q, P, 1e-8f);
  TT_CHECK_LAUNCH();
  hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(kThreads), 0, st, d, dn, (long long)N, P, 1e-8f);
  TT_CHECK_LAUNCH();
  ttg::Batch b{};
  b.count = 1;
  b.nsplit = 1;
  ttg::Problem& p = b.p[0];
  p.A = qn; p.B = dn; p.C = out;
  p.M = (int)Q; p.N = (int)N; p.K = P; p.lda = P; p.ldb = P; p.ldc = (int)N;
  const long long tiles = ((Q + 63) / 64) * ((N + 63) / 64);
  const int w = tiles >= 2048 ? 2 : 1;
  return ttg::launch(ttg::kNT, ttg::kStore, b, w, w, st);
}

extern "C" int tt_ndcg_at_k_f32(const float* scores, const float* gains, const uint8_t* valid, int64_t Q,
                                int64_t N, const int32_t* ks_host, int nk, int32_t* ks_device, float* out,
                                tt_stream_t stream) {
  TT_REQUIRE(scores && gains && ks_host && ks_device && out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(Q >= 1 && N >= 1 && nk >= 1 && nk <= 8 && Q <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
  for (int i = 0; i < nk; ++i) TT_REQUIRE(ks_host[i] >= 1, TT_ERR_BAD_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  TT_HIP(hipMemcpyAsync(ks_device, ks_host, sizeof(int32_t) * nk, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ndcg_kernel, dim3((unsigned)Q), dim3(kThreads), 0, st, scores, gains, valid, (long long)N,
                     ks_device, nk, out);
  TT_CHECK_LAUNCH();
  return TT_OK;
}
