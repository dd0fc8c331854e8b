// f4 — in-batch-negatives triplet loss (BASELINE.json configs[4]): every other document of the
// batch is a negative for query i; the B x B cosine-similarity matrix and both gradient products
// run on the fp32 MFMA GEMM.  The reference has no such loss (its negatives are ONE sampled
// in-batch positive per item, backend/data.py:113-137), so parity is against this repo's own CPU
// oracle (oracle/two_towers_oracle.py::inbatch_triplet_loss) — "parity unpinned".
//
//   S[i,j]  = cos(q_i, d_j)                                   (eps 1e-8, like F.cosine_similarity)
//   L       = loss_scale * sum_i sum_{j valid} relu(S[i,j] - S[i,pos_i] + margin)
//   valid j : j != pos_i and (no query ids given or qid_d[j] != qid_q[i])
#include <cstdlib>
#include "gemm_f32.h"

int tt_launch_loss_mean(const float* hinge_rows, long long B, float denom, float* loss, double* acc_sum,
                        long long* acc_count, hipStream_t st);

namespace {

constexpr int kThreads = 256;

// Unit rows and the positives' similarities in one launch.  Rows [0, Bq): query i -> qhat_i = q_i / max(|q_i|, eps),
// |q_i| and s_pos[i] = qhat_i . dhat_{pos_i}, which every column tile of row i needs before the S product's epilogue
// can turn S into dL/dS; rows [Bq, Bq + Bd): document j -> dhat_j, |d_j|.  s_pos is formed from the positive's RAW
// row, normalised on the fly (divide, then fma chain), so it does not have to wait for dhat.
__global__ __launch_bounds__(kThreads) void inbatch_prep_kernel(const float* __restrict__ q, const float* __restrict__ d,
                                                                const int* __restrict__ pos_index, float* __restrict__ qh,
                                                                float* __restrict__ dh, float* __restrict__ nq,
                                                                float* __restrict__ nd, float* __restrict__ s_pos,
                                                                long long Bq, long long Bd, int P, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Bq + Bd) return;
  const bool is_q = row < Bq;
  const long long r = is_q ? row : row - Bq;
  const float* __restrict__ xr = (is_q ? q : d) + r * P;
  float* __restrict__ yr = (is_q ? qh : dh) + r * P;
  float ss = 0.f;
  for (int c = lane; c < P; c += 64) ss = fmaf(xr[c], xr[c], ss);
  const float n = sqrtf(tt_wave_sum(ss));
  const float den = fmaxf(n, eps);
  for (int c = lane; c < P; c += 64) yr[c] = xr[c] / den;
  if (lane == 0) (is_q ? nq : nd)[r] = n;
  if (!is_q) return;
  long long pi = pos_index ? pos_index[r] : r;
  pi = pi < 0 ? 0 : (pi >= Bd ? Bd - 1 : pi);
  const float* __restrict__ pr = d + pi * P;
  float sp = 0.f;
  for (int c = lane; c < P; c += 64) sp = fmaf(pr[c], pr[c], sp);
  const float pden = fmaxf(sqrtf(tt_wave_sum(sp)), eps);
  float t = 0.f;
  for (int c = lane; c < P; c += 64) t = fmaf(xr[c] / den, pr[c] / pden, t);
  t = tt_wave_sum(t);
  if (lane == 0) s_pos[r] = t;
}

// After the S product (epilogue kHinge: S already replaced by dL/dS except on the positive's column): per query row,
// add the column tiles' hinge sums and active counts in tile order, store the row loss and
// dL/dS[i, pos_i] = -g * (number of active negatives of row i).  One wave per row.
__global__ __launch_bounds__(kThreads) void inbatch_finish_kernel(float* __restrict__ G, const int* __restrict__ pos_index,
                                                                  const float* __restrict__ part_loss,
                                                                  const float* __restrict__ part_cnt, int n_tiles,
                                                                  long long Bq, long long Bd, float g,
                                                                  float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Bq) return;
  float l = 0.f, c = 0.f;
  for (int t0 = 0; t0 < n_tiles; t0 += 64) {  // fixed order: chunks of 64 tiles, butterfly inside a chunk
    const int t = t0 + lane;
    const float lv = t < n_tiles ? part_loss[i * n_tiles + t] : 0.f;
    const float cv = t < n_tiles ? part_cnt[i * n_tiles + t] : 0.f;
    l += tt_wave_sum(lv);
    c += tt_wave_sum(cv);
  }
  if (lane == 0) {
    long long pi = pos_index ? pos_index[i] : i;
    pi = pi < 0 ? 0 : (pi >= Bd ? Bd - 1 : pi);
    row_loss[i] = l;
    G[i * Bd + pi] = -g * c;
  }
}

// d(x / max(|x|, eps))^T applied to the upstream gradient u of the unit row:
// dx = (u - xhat * (xhat . u) * r) / c,  c = max(|x|, eps), r = c/|x| (|x| > 0) else 0
// For queries (rows [0, Bq)) and documents (rows [Bq, Bq + Bd)) in one launch; `u` arrives as split-K partial
// slabs (slab z at u + z * slab_stride), summed here in slab order.
__global__ __launch_bounds__(kThreads) void unit_rows_bwd2_kernel(
    const float* __restrict__ qh, const float* __restrict__ nq, const float* __restrict__ uq, int sq, long long stride_q,
    float* __restrict__ dq, long long Bq, const float* __restrict__ dh, const float* __restrict__ nd,
    const float* __restrict__ ud, int sd, long long stride_d, float* __restrict__ dd, long long Bd, int P, float eps,
    int accumulate_dd) {
  const int lane = threadIdx.x & 63;
  const long long row0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row0 >= Bq + Bd) return;
  const bool is_q = row0 < Bq;
  const long long row = is_q ? row0 : row0 - Bq;
  const float* __restrict__ xhat = is_q ? qh : dh;
  const float* __restrict__ nrm = is_q ? nq : nd;
  const float* __restrict__ u = is_q ? uq : ud;
  const int nslabs = is_q ? sq : sd;
  const long long slab_stride = is_q ? stride_q : stride_d;
  float* __restrict__ dx = is_q ? dq : dd;
  const int accumulate = is_q ? 0 : accumulate_dd;
  const long long o = row * P;
  auto uval = [&](int c) {
    float v = u[o + c];
    for (int z = 1; z < nslabs; ++z) v += u[z * slab_stride + o + c];
    return v;
  };
  float t = 0.f;
  for (int c = lane; c < P; c += 64) t = fmaf(xhat[o + c], uval(c), t);
  t = tt_wave_sum(t);
  const float n = nrm[row], cden = fmaxf(n, eps), r = n > 0.f ? cden / n : 0.f;
  for (int c = lane; c < P; c += 64) {
    const float v = (uval(c) - xhat[o + c] * (t * r)) / cden;
    dx[o + c] = accumulate ? dx[o + c] + v : v;
  }
}

inline size_t r4(size_t x) { return (x + 3) / 4 * 4; }

}  // namespace

// split-K factor of the two gradient products (G dhat, G^T qhat): [B x P] outputs are few 64x64 tiles (B = 4096,
// P = 384: 384 tiles on 256 CUs), so the long contraction (K = B) is cut until there are about three per CU
static int grad_splits(long long rows, int P, long long K) {
  const long long tiles = ((rows + 63) / 64) * ((P + 63) / 64);
  int s = (int)((3 * 256 + tiles - 1) / tiles);
  if (s < 1) s = 1;
  if (s > 8) s = 8;
  while (s > 1 && K / s < 256) --s;  // keep at least 8 K-steps per split
  return s;
}
// block tile of the S product, in units of 64.  64x64 everywhere: measured against 128x128 tiles in round 3 it is never
// slower and 3-4 % of the step faster at B = 4096 / 8192 beside the gathers (profiles/r03_inbatch_tile_ab.txt) — with
// K = P = 384 a 128x128 tile is 12 K-steps between a prologue and a 64 KB hinge epilogue, and only two of them fit a CU.
static int s_tile(long long, long long, int) { return 1; }
static int col_tiles(long long Bq, long long Bd, int P) {  // column tiles of the S product: row stride of the partials
  const int t = s_tile(Bq, Bd, P);
  return (int)((Bd + 64 * t - 1) / (64 * t));
}

extern "C" size_t tt_inbatch_workspace_bytes(int64_t Bq, int64_t Bd, int P) {
  const size_t sq = grad_splits(Bq, P, Bd), sd = grad_splits(Bd, P, Bq);
  const size_t f = r4((size_t)Bq * P) * (1 + sq) + r4((size_t)Bd * P) * (1 + sd) + r4(Bq) * 3 + r4(Bd) +
                   2 * r4((size_t)Bq * col_tiles(Bq, Bd, P)) + r4((size_t)Bq * Bd);
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
  const int sq = grad_splits(Bq, P, Bd), sd = grad_splits(Bd, P, Bq), nt = col_tiles(Bq, Bd, P);
  const long long stride_q = (long long)r4((size_t)Bq * P), stride_d = (long long)r4((size_t)Bd * P);
  float* w = static_cast<float*>(workspace);
  float* qh = w;            w += stride_q;
  float* dh = w;            w += stride_d;
  float* uq = w;            w += stride_q * sq;
  float* ud = w;            w += stride_d * sd;
  float* nq = w;            w += r4(Bq);
  float* row_loss = w;      w += r4(Bq);
  float* s_pos = w;         w += r4(Bq);
  float* nd = w;            w += r4(Bd);
  float* part_loss = w;     w += r4((size_t)Bq * nt);
  float* part_cnt = w;      w += r4((size_t)Bq * nt);
  float* S = w;

  hipLaunchKernelGGL(inbatch_prep_kernel, dim3((unsigned)((Bq + Bd + 3) / 4)), dim3(kThreads), 0, st, q, d, pos_index, qh, dh, nq,
                     nd, s_pos, (long long)Bq, (long long)Bd, P, 1e-8f);
  TT_CHECK_LAUNCH();

  {  // G = dL/dS of S = qh dh^T, the hinge applied in the product's epilogue: S itself is never stored   [Bq, Bd]
    ttg::Batch b{};
    b.count = 1; b.nsplit = 1;
    ttg::Problem& p = b.p[0];
    p.A = qh; p.B = dh; p.C = S; p.M = (int)Bq; p.N = (int)Bd; p.K = P; p.lda = P; p.ldb = P; p.ldc = (int)Bd;
    b.hinge.pos_index = pos_index;
    b.hinge.qid_q = reinterpret_cast<const long long*>(qid_q);
    b.hinge.qid_d = reinterpret_cast<const long long*>(qid_d);
    b.hinge.s_pos = s_pos;
    b.hinge.margin = margin;
    b.hinge.g = grad_scale;
    b.hinge.part_loss = part_loss;
    b.hinge.part_cnt = part_cnt;
    b.hinge.n_tiles = nt;
    const int t = s_tile(Bq, Bd, P);
    int rc = ttg::launch(ttg::kNT, ttg::kHinge, b, t, t, st);
    if (rc != TT_OK) return rc;
  }
  hipLaunchKernelGGL(inbatch_finish_kernel, dim3((unsigned)((Bq + 3) / 4)), dim3(kThreads), 0, st, S, pos_index, part_loss, part_cnt,
                     nt, (long long)Bq, (long long)Bd, grad_scale, row_loss);
  TT_CHECK_LAUNCH();
  if (loss) {
    int rc = tt_launch_loss_mean(row_loss, Bq, loss_denominator, loss, nullptr, nullptr, st);
    if (rc != TT_OK) return rc;
  }
  {  // uq = G dh                                           [Bq, P]   (sq split-K slabs)
    ttg::Batch b{};
    b.count = 1; b.nsplit = sq; b.slab_stride = stride_q;
    ttg::Problem& p = b.p[0];
    p.A = S; p.B = dh; p.C = uq; p.M = (int)Bq; p.N = P; p.K = (int)Bd; p.lda = (int)Bd; p.ldb = P; p.ldc = P;
    p.k_per_split = (int)(((Bd + sq - 1) / sq + 31) / 32 * 32);
    int rc = ttg::launch(ttg::kNN, ttg::kStore, b, 1, 1, st);
    if (rc != TT_OK) return rc;
  }
  {  // ud = G^T qh                                         [Bd, P]   (sd split-K slabs)
    ttg::Batch b{};
    b.count = 1; b.nsplit = sd; b.slab_stride = stride_d;
    ttg::Problem& p = b.p[0];
    p.A = S; p.B = qh; p.C = ud; p.M = (int)Bd; p.N = P; p.K = (int)Bq; p.lda = (int)Bd; p.ldb = P; p.ldc = P;
    p.k_per_split = (int)(((Bq + sd - 1) / sd + 31) / 32 * 32);
    int rc = ttg::launch(ttg::kTN, ttg::kStore, b, 1, 1, st);
    if (rc != TT_OK) return rc;
  }
  hipLaunchKernelGGL(unit_rows_bwd2_kernel, dim3((unsigned)((Bq + Bd + 3) / 4)), dim3(kThreads), 0, st, qh, nq, uq, sq, stride_q, dq,
                     (long long)Bq, dh, nd, ud, sd, stride_d, dd, (long long)Bd, P, 1e-8f, accumulate_dd);
  TT_CHECK_LAUNCH();
  return TT_OK;
}
