// f4 — in-batch-negatives triplet loss (BASELINE.json configs[4]): every other document of the
// batch is a negative for query i; the B x B cosine-similarity matrix and both gradient products
// run on the fp32 MFMA GEMM.  The reference has no such loss (its negatives are ONE sampled
// in-batch positive per item, backend/data.py:113-137), so parity is against this repo's own CPU
// oracle (oracle/two_towers_oracle.py::inbatch_triplet_loss) — "parity unpinned".
//
//   S[i,j]  = cos(q_i, d_j)                                   (eps 1e-8, like F.cosine_similarity)
//   L       = loss_scale * sum_i sum_{j valid} relu(S[i,j] - S[i,pos_i] + margin)
//   valid j : j != pos_i and (no query ids given or qid_d[j] != qid_q[i])
#include "gemm_f32.h"

int tt_launch_loss_mean(const float* hinge_rows, long long B, float denom, float* loss, double* acc_sum,
                        long long* acc_count, hipStream_t st);

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void unit_rows_norm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  float* __restrict__ nrm, long long rows, int P,
                                                                  float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* __restrict__ xr = x + row * P;
  float ss = 0.f;
  for (int c = lane; c < P; c += 64) ss = fmaf(xr[c], xr[c], ss);
  const float n = sqrtf(tt_wave_sum(ss));
  const float den = fmaxf(n, eps);
  for (int c = lane; c < P; c += 64) y[row * P + c] = xr[c] / den;
  if (lane == 0) nrm[row] = n;
}

// One workgroup per query row: hinge over the row of S, S overwritten by dL/dS.
__global__ __launch_bounds__(kThreads) void inbatch_hinge_kernel(float* __restrict__ S, const int* __restrict__ pos_index,
                                                                 const long long* __restrict__ qid_q,
                                                                 const long long* __restrict__ qid_d, long long Bd,
                                                                 float margin, float g, float* __restrict__ row_loss) {
  __shared__ float red_l[kThreads / 64], red_c[kThreads / 64];
  const long long i = blockIdx.x;
  float* __restrict__ s = S + i * Bd;
  const int pi = pos_index ? pos_index[i] : (int)i;
  const float s_pos = s[pi];
  const long long my_q = qid_q ? qid_q[i] : 0;
  __syncthreads();  // everyone has read s[pi] before anyone overwrites it
  float loss = 0.f, cnt = 0.f;
  for (long long j = threadIdx.x; j < Bd; j += kThreads) {
    float gij = 0.f;
    const bool valid = j != pi && !(qid_q && qid_d && qid_d[j] == my_q);
    if (valid) {
      const float h = s[j] - s_pos + margin;  // = (1 - s_pos) - (1 - s[j]) + margin
      if (h > 0.f) {
        loss += h;
        cnt += 1.f;
        gij = g;
      }
    }
    if (j != pi) s[j] = gij;
  }
  loss = tt_wave_sum(loss);
  cnt = tt_wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) {
    red_l[threadIdx.x >> 6] = loss;
    red_c[threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    row_loss[i] = ((red_l[0] + red_l[1]) + red_l[2]) + red_l[3];
    s[pi] = -g * (((red_c[0] + red_c[1]) + red_c[2]) + red_c[3]);
  }
}

// d(x / max(|x|, eps))^T applied to the upstream gradient u of the unit row:
// dx = (u - xhat * (xhat . u) * r) / c,  c = max(|x|, eps), r = c/|x| (|x| > 0) else 0
__global__ __launch_bounds__(kThreads) void unit_rows_bwd_kernel(const float* __restrict__ xhat,
                                                                 const float* __restrict__ nrm,
                                                                 const float* __restrict__ u, float* __restrict__ dx,
                                                                 long long rows, int P, float eps, int accumulate) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long o = row * P;
  float t = 0.f;
  for (int c = lane; c < P; c += 64) t = fmaf(xhat[o + c], u[o + c], t);
  t = tt_wave_sum(t);
  const float n = nrm[row], cden = fmaxf(n, eps), r = n > 0.f ? cden / n : 0.f;
  for (int c = lane; c < P; c += 64) {
    const float v = (u[o + c] - xhat[o + c] * (t * r)) / cden;
    dx[o + c] = accumulate ? dx[o + c] + v : v;
  }
}

inline size_t r4(size_t x) { return (x + 3) / 4 * 4; }

}  // namespace

extern "C" size_t tt_inbatch_workspace_bytes(int64_t Bq, int64_t Bd, int P) {
  const size_t f = r4((size_t)Bq * P) * 2 + r4((size_t)Bd * P) * 2 + r4(Bq) * 2 + r4(Bd) + r4((size_t)Bq * Bd);
  return f * sizeof(float);
}

extern "C" int tt_inbatch_triplet_fwd_bwd_f32(const float* q, const float* d, const int32_t* pos_index,
                                              const int64_t* qid_q, const int64_t* qid_d, float margin,
                                              float grad_scale, float loss_denominator, int64_t Bq, int64_t Bd,
                                              int P, float* loss, float* dq, float* dd, int accumulate_dd,
                                              void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(q && d && dq && dd && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE((qid_q == nullptr) == (qid_d == nullptr), TT_ERR_NULL_POINTER);
  TT_REQUIRE(Bq >= 1 && Bd >= 2 && P >= 1 && Bq <= 0x7fffffffLL && Bd <= 0x7fffffffLL && loss_denominator > 0.f,
             TT_ERR_BAD_SHAPE);
  TT_REQUIRE(pos_index || Bq == Bd, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(workspace_bytes >= tt_inbatch_workspace_bytes(Bq, Bd, P), TT_ERR_WORKSPACE);
  TT_REQUIRE(tt_aligned16(workspace), TT_ERR_ALIGNMENT);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* w = static_cast<float*>(workspace);
  float* qh = w;            w += r4((size_t)Bq * P);
  float* dh = w;            w += r4((size_t)Bd * P);
  float* uq = w;            w += r4((size_t)Bq * P);
  float* ud = w;            w += r4((size_t)Bd * P);
  float* nq = w;            w += r4(Bq);
  float* row_loss = w;      w += r4(Bq);
  float* nd = w;            w += r4(Bd);
  float* S = w;

  hipLaunchKernelGGL(unit_rows_norm_kernel, dim3((unsigned)((Bq + 3) / 4)), dim3(kThreads), 0, st, q, qh, nq, (long long)Bq, P, 1e-8f);
  TT_CHECK_LAUNCH();
  hipLaunchKernelGGL(unit_rows_norm_kernel, dim3((unsigned)((Bd + 3) / 4)), dim3(kThreads), 0, st, d, dh, nd, (long long)Bd, P, 1e-8f);
  TT_CHECK_LAUNCH();

  auto tile = [](long long m, long long n) { return ((m + 63) / 64) * ((n + 63) / 64) >= 2048 ? 2 : 1; };
  {  // S = qh dh^T                                         [Bq, Bd]
    ttg::Batch b{};
    b.count = 1; b.nsplit = 1;
    ttg::Problem& p = b.p[0];
    p.A = qh; p.B = dh; p.C = S; p.M = (int)Bq; p.N = (int)Bd; p.K = P; p.lda = P; p.ldb = P; p.ldc = (int)Bd;
    const int t = tile(Bq, Bd);
    int rc = ttg::launch(ttg::kNT, ttg::kStore, b, t, t, st);
    if (rc != TT_OK) return rc;
  }
  hipLaunchKernelGGL(inbatch_hinge_kernel, dim3((unsigned)Bq), dim3(kThreads), 0, st, S, pos_index,
                     reinterpret_cast<const long long*>(qid_q), reinterpret_cast<const long long*>(qid_d), (long long)Bd,
                     margin, grad_scale, row_loss);
  TT_CHECK_LAUNCH();
  if (loss) {
    int rc = tt_launch_loss_mean(row_loss, Bq, loss_denominator, loss, nullptr, nullptr, st);
    if (rc != TT_OK) return rc;
  }
  {  // uq = G dh                                           [Bq, P]   (G = dL/dS, in S)
    ttg::Batch b{};
    b.count = 1; b.nsplit = 1;
    ttg::Problem& p = b.p[0];
    p.A = S; p.B = dh; p.C = uq; p.M = (int)Bq; p.N = P; p.K = (int)Bd; p.lda = (int)Bd; p.ldb = P; p.ldc = P;
    int rc = ttg::launch(ttg::kNN, ttg::kStore, b, 1, 1, st);
    if (rc != TT_OK) return rc;
  }
  {  // ud = G^T qh                                         [Bd, P]
    ttg::Batch b{};
    b.count = 1; b.nsplit = 1;
    ttg::Problem& p = b.p[0];
    p.A = S; p.B = qh; p.C = ud; p.M = (int)Bd; p.N = P; p.K = (int)Bq; p.lda = (int)Bd; p.ldb = P; p.ldc = P;
    p.k_per_split = (int)((Bq + 31) / 32 * 32);
    int rc = ttg::launch(ttg::kTN, ttg::kStore, b, 1, 1, st);
    if (rc != TT_OK) return rc;
  }
  hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((unsigned)((Bq + 3) / 4)), dim3(kThreads), 0, st, qh, nq, uq, dq, (long long)Bq, P, 1e-8f, 0);
  TT_CHECK_LAUNCH();
  hipLaunchKernelGGL(unit_rows_bwd_kernel, dim3((unsigned)((Bd + 3) / 4)), dim3(kThreads), 0, st, dh, nd, ud, dd, (long long)Bd, P, 1e-8f, accumulate_dd);
  TT_CHECK_LAUNCH();
  return TT_OK;
}
