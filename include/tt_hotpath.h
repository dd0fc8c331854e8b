/*
 * tt_hotpath.h — C ABI of the MI355X (gfx950) two-tower triplet hot path.
 *
 * Drop-in boundary.  The reference (freemvmt/two-towers-overlords) is pure Python
 * and has no FFI of its own; its boundary for this path is the class surface of
 * backend/model.py and backend/training.py.  This header is the C ABI *under*
 * that surface: each entry point replaces the torch op sequence of the reference
 * lines it cites (paths relative to the reference repo).  The Python host side
 * (two-towers-overlords_amd/{model,training}.py) mirrors the reference classes
 * and binds these symbols with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     marked host; the caller (PyTorch) owns every buffer; nothing is retained
 *     after the call returns except inside an explicit tt_engine handle.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream); no call synchronises the device except
 *     tt_engine_read_stats and the create/destroy pairs.
 *   - return value: 0 = TT_OK, otherwise a tt_status (negative = argument error
 *     detected on the host before any launch, positive = hipError_t).  No C++
 *     exception crosses the boundary, nothing calls exit().
 *   - row-major fp32 everywhere; token ids / masks are int32 or int64 (the HF
 *     tokenizer's int64 tensors are consumed as they are: backend/model.py:43-48).
 *   - re-entrant, no global mutable state, no environment variables: every choice a caller can make is an
 *     argument (tt_tuning below) or a property of an explicit handle; one engine handle per device/rank.
 */
#ifndef TT_HOTPATH_H
#define TT_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_ABI_VERSION 4

typedef void* tt_stream_t; /* hipStream_t */

typedef enum {
  TT_OK = 0,
  TT_ERR_NULL_POINTER = -1,
  TT_ERR_BAD_SHAPE = -2,
  TT_ERR_BAD_DTYPE = -3,
  TT_ERR_ALIGNMENT = -4,
  TT_ERR_WORKSPACE = -5,
  TT_ERR_BAD_HANDLE = -6,
  TT_ERR_UNSUPPORTED = -7,
  TT_ERR_WRONG_DEVICE = -8 /* tt_engine_create: the calling thread's current HIP device is not cfg->device, or the
                              workspace lives on another device (one engine handle per device/rank: hipSetDevice first) */
} tt_status;

/* integer element widths accepted for ids / masks */
#define TT_I32 4
#define TT_I64 8
/* a token id that means "padding" (see tt_gather_pool_norm_f32) */
#define TT_PAD_ID (-2147483647 - 1)

/*
 * Tuning: which of several equivalent kernels / tilings / schedules a call uses.  Every field 0 = "auto", the
 * measured-best choice for the shape (the measurements behind each rule: docs/LOG_r04.md sections 3, 5, 8).  Speed only: no field changes a result beyond fp32
 * summation order, and the projection tiles do not even do that (every output element is a k-ordered fmaf chain
 * whatever the tile).  Passed per call (NULL = all auto) or bound to an engine handle (tt_engine_set_tuning); there
 * is no process-wide state behind it.  Fields a call has no use for are ignored; values a shape does not support
 * fall back to auto.
 */
#define TT_K1_AUTO 0
#define TT_K1_PER_TEXT 1      /* one workgroup per text */
#define TT_K1_ROW_SLICES 2    /* XCD-sliced, token ids mod 8 */
#define TT_K1_COLUMN_SLICES 3 /* XCD-sliced, 32-column slices (D % 32 == 0) */
#define TT_K1_STREAM 4        /* persistent workgroups + per-CU registry (tables that stream from HBM) */
#define TT_K1_BINNED 5        /* tokens sorted by row class (id % 8) + 32-column slices: an (XCD, slice) block of a big table
                                 fits the XCD's L2 (D % 32 == 0) */
#define TT_BWD_AUTO 0
#define TT_BWD_FORK 1
#define TT_BWD_FUSED 2
typedef struct {
  int32_t k1_form;         /* TT_K1_*: form of the gather+pool kernel */
  int32_t k1_stream_wgs;   /* streaming K1: gather workgroups kept per compute unit, 1..8 (auto: 2) */
  int32_t k1_stream_rows;  /* streaming K1: table rows in flight per wave (auto: by D; 4, 6 or 8 at 512 < D <= 768) */
  int32_t k1_stream_offer; /* streaming K1: candidate workgroups launched per workgroup the chip may keep (auto: 16) */
  int32_t k1_share_lds_kb; /* engine: KiB of LDS a gather workgroup occupies while it shares the chip with the
                              projection kernels (auto: paired with the projection tile; -1: no reservation) */
  int32_t proj_tile;       /* projection block tile, rows x columns in units of 64: 11, 21, 22 (auto: by shape).  Every tile
                              gives bitwise the same product.  (ABI 3 also took 33 / 32, K-split panels of 96 rows: left
                              the library in ABI 4 — no policy ever picked them; any other value means auto) */
  int32_t proj_bk;         /* K-step of the 128x128 projection tile: 16 or 32 */
  int32_t tn_tile;         /* weight-gradient (split-K) tile: 11 or 22 */
  int32_t xcd_order;       /* XCD-aware workgroup order of the projection GEMMs: 1 on, -1 off (auto: beside the streaming
                              gather, and beside the sliced gather from 6144 rows per launch) */
  int32_t backward;        /* engine: TT_BWD_* schedule of phase B's backward */
  int32_t event_scope;     /* engine: release scope of the events that order its two streams: 1 no system-scope fence
                              (hipEventDisableSystemFence; auto), 2 device-scope release (hipEventReleaseToDevice),
                              3 HIP's default (system-scope fence when the event is recorded) */
  int32_t reserved[5];     /* must be 0 */
} tt_tuning;

int tt_abi_version(void);
/* Static string for a tt_status / hipError_t value (host pointer, never freed). */
const char* tt_status_string(int status);

/* ------------------------------------------------------------------------- */
/* K1 — token-embedding gather + masked mean-pool + L2 normalise             */
/* ------------------------------------------------------------------------- */

/*
 * out[n,:] = normalize( sum_s w[n,s] * table[ids[n,s],:] / max(sum_s w[n,s], 1e-9) ), eps 1e-12
 *
 * Replaces, in `embed` encoder mode, backend/model.py:51-56:
 *   output = self.pretrained_model(**tokens)            (word-embedding gather)
 *   pooled = self._mean_pooling(output, attention_mask) (model.py:63-72)
 *   pooled = F.normalize(pooled, p=2, dim=1)
 * without materialising the [N,S,D] tensor.
 *
 *   table [V,D] f32, 16-byte aligned, D % 4 == 0
 *   ids   [N,S] int32 or int64 (ids_dtype = TT_I32 / TT_I64).  TT_PAD_ID (INT32_MIN; INT64_MIN in int64 ids too) marks
 *         padding inside the ids themselves: such a token has weight 0 whatever mask / lens say, is never read and never
 *         reported — every gather form honours it (what tt_engine_pool_roles writes behind the attended tokens)
 *   w:    mask  [N,S] int32/int64 (mask_dtype) attention mask, any 0/1 pattern
 *         (values are cast to float as the reference does, so weights other
 *         than 0/1 are honoured), or, if mask == NULL, lens [N] int32 prefix
 *         lengths (w = s < lens[n]), or all ones if both are NULL.
 *   out   [N,D] f32
 *   oob_flag: optional int32[1]; set to 1 if an id outside [0,V) is met.  Such
 *         ids are clamped into range, so the kernel never reads outside `table`.
 */
int tt_gather_pool_norm_f32(const float* table, const void* ids, int ids_dtype,
                            const void* mask, int mask_dtype, const int32_t* lens,
                            float* out, int64_t N, int S, int D, int64_t V,
                            int32_t* oob_flag, tt_stream_t stream);

/*
 * XCD-sliced form of tt_gather_pool_norm_f32 (same arguments, same result to fp32 summation order):
 * the workgroups that share an XCD read one SLICE of the table, so that slice is served by the XCD's own
 * L2.  Two ways of slicing, picked per shape (tt_k1_sliced_form):
 *   0  rows: token ids of one residue class mod 8 (an eighth of the rows); per-slice partial sums meet in
 *      `workspace` and a second launch adds them in slice order, divides and normalises;
 *   1  columns: 32 columns of every row (one 128-byte line per row: V x 128 bytes per slice), where
 *      D % 32 == 0 and such a slice fits the 4 MiB L2 — the partial of a (text, slice) is that slice of
 *      the output itself; the second launch only applies the L2 norm.
 * `workspace`: tt_gather_pool_sliced_workspace_bytes(N, D) bytes, 16-byte aligned.  Requires D % 4 == 0,
 * D <= 1024, S <= 512.  Faster than the per-text form when the table is far larger than one L2 (4 MiB) —
 * BERT-sized vocabularies.
 */
size_t tt_gather_pool_sliced_workspace_bytes(int64_t N, int D);
/* which slicing tt_gather_pool_norm_sliced_f32 picks for this shape when asked for TT_K1_AUTO:
 * TT_K1_ROW_SLICES or TT_K1_COLUMN_SLICES (a pure function of the shape) */
int tt_k1_sliced_form(int64_t V, int D, int S);
/* 1 if the sliced form is the faster one for this shape on MI355X (a pure function of the shape, from
 * measurements; tt_engine_pool applies it by itself). */
int tt_k1_prefers_sliced(int64_t V, int D, int64_t N, int S);
/* form: TT_K1_AUTO (tt_k1_sliced_form), TT_K1_ROW_SLICES or TT_K1_COLUMN_SLICES (needs D % 32 == 0, D <= 1024;
 * else TT_ERR_UNSUPPORTED) */
int tt_gather_pool_norm_sliced_f32(const float* table, const void* ids, int ids_dtype,
                                   const void* mask, int mask_dtype, const int32_t* lens,
                                   float* out, int64_t N, int S, int D, int64_t V,
                                   int32_t* oob_flag, void* workspace, size_t workspace_bytes,
                                   int form, tt_stream_t stream);

/*
 * Streaming form of tt_gather_pool_norm_f32 for tables that really come from HBM (several times the 256 MiB
 * Infinity Cache; BASELINE.json configs[3]), built to share the chip with MFMA kernels of another stream:
 * persistent workgroups pull texts from a queue and at most `workgroups_per_cu` of them stay on any compute
 * unit, so the gather keeps a fixed number of row loads in flight per CU and the rest of every CU stays free.
 * Candidate workgroups are offered well beyond that number (tt_tuning.k1_stream_offer per kept workgroup): a
 * candidate that finds its CU full leaves at once, the ones the chip has no room for wait in the dispatcher and
 * arrive as room appears, so every CU gets its share whatever else is running; once the planned number is
 * admitted (or the queue is empty) the remaining candidates leave without touching anything.  Same arguments
 * and bitwise the same result as tt_gather_pool_norm_f32.  `workspace`: tt_gather_pool_stream_workspace_bytes()
 * bytes, 16-byte aligned, private to one launch at a time (it holds the queue head and the per-CU registry;
 * cleared by the call).  `tuning` (may be NULL): k1_stream_rows, k1_stream_offer.
 * Requires D % 4 == 0, D <= 1024.  Replaces the same reference lines (backend/model.py:51-56, 63-72).
 */
size_t tt_gather_pool_stream_workspace_bytes(void);
int tt_k1_stream_supported(int D, const float* table, const float* out);
/* 1 if the streaming form is the one to use for this shape on MI355X (a pure function of the shape;
 * tt_engine_pool applies it by itself) */
int tt_k1_prefers_stream(int64_t V, int D, int64_t N, int S);
int tt_gather_pool_norm_stream_f32(const float* table, const void* ids, int ids_dtype,
                                   const void* mask, int mask_dtype, const int32_t* lens,
                                   float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                                   void* workspace, size_t workspace_bytes, int workgroups_per_cu,
                                   const tt_tuning* tuning, tt_stream_t stream);
/* Diagnostic (synchronises `stream`): workgroups the last streaming launch on `workspace` admitted, compute
 * units that hosted at least one ADMITTED workgroup, the largest number of candidates that tried one CU, and the
 * workgroups the launch meant to keep (workgroups_per_cu x compute units, or fewer for a short queue). */
int tt_gather_pool_stream_census(const void* workspace, int32_t* admitted, int32_t* cus_used,
                                 int32_t* max_registered, int32_t* planned, tt_stream_t stream);

/*
 * Binned form of tt_gather_pool_norm_f32 for tables too big for the column slices (V x 128 bytes beyond an L2) whose
 * rows are named many times per launch (N x S >> V; BASELINE.json configs[3]: 12.6 M tokens over 250 000 rows): each
 * text's live tokens are sorted by row class id % 8 once, then XCD c gathers only class c, 32 columns at a time — block
 * (c, slice) of the table is V x 16 bytes (4.0 MB at V = 250 000) and stays in that XCD's L2, so a table line crosses the
 * fabric about once per launch instead of once per token that names it; the 8 class partials of a text are added in
 * class order, then mean (clamp 1e-9) and L2 norm (clamp 1e-12).  Same arguments as tt_gather_pool_norm_f32 and the
 * same result up to fp32 summation order; run-to-run bitwise reproducible.  `workspace`:
 * tt_gather_pool_binned_workspace_bytes(N, D, S) bytes, 16-byte aligned (8 x N x D floats of partials + the sorted
 * ids / weights).  Requires D % 32 == 0, D <= 1024, S <= 512.  Replaces backend/model.py:51-56, 63-72.
 */
size_t tt_gather_pool_binned_workspace_bytes(int64_t N, int D, int S);
int tt_k1_binned_supported(int S, int D, const float* table, const float* out);
/* 1 if the binned form is the one to use for this shape on MI355X (a pure function of the shape; tt_engine_pool applies
 * it by itself): a column slice of the table beyond 13 MB but a class block (V x 16 bytes) within 4.5 MB, texts of >= 64
 * tokens, every row named >= 8 times per launch */
int tt_k1_prefers_binned(int64_t V, int D, int64_t N, int S);
int tt_gather_pool_norm_binned_f32(const float* table, const void* ids, int ids_dtype,
                                   const void* mask, int mask_dtype, const int32_t* lens,
                                   float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                                   void* workspace, size_t workspace_bytes, tt_stream_t stream);

/*
 * Same pooling + normalisation over dense hidden states [N,S,D] (the `bert`
 * encoder mode: the frozen encoder of backend/model.py:51-52 is run by stock
 * PyTorch-ROCm, its last_hidden_state is pooled here).  Replaces
 * backend/model.py:55-56,63-72.
 */
int tt_pool_norm_dense_f32(const float* hidden, const void* mask, int mask_dtype,
                           const int32_t* lens, float* out, int64_t N, int S, int D,
                           tt_stream_t stream);

/*
 * Plain row gather out[t,:] = table[ids[t],:] for t < n_tokens — the
 * materialised form of backend/model.py:51-52 (`embed` mode).  Used to prove
 * the gather indices bit-exact, and by callers that need token embeddings.
 */
int tt_embed_gather_f32(const float* table, const void* ids, int ids_dtype, float* out,
                        int64_t n_tokens, int D, int64_t V, int32_t* oob_flag,
                        tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K2 — per-tower projection Linear(D,P) -> ReLU -> Linear(P,P), fp32 MFMA     */
/* ------------------------------------------------------------------------- */

/*
 * H = relu(X W1^T + b1),  Out = H W2^T + b2.     Replaces backend/model.py:33-38,59.
 *   X [M,D], W1 [P,D], b1 [P], W2 [P,P], b2 [P], H [M,P] (saved for backward), Out [M,P]
 * Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): each output is a k-ordered fmaf chain.
 * `tuning` (may be NULL): proj_tile, proj_bk, xcd_order.
 */
int tt_proj_fwd_f32(const float* X, const float* W1, const float* b1, const float* W2,
                    const float* b2, float* H, float* Out, int64_t M, int D, int P,
                    const tt_tuning* tuning, tt_stream_t stream);

/* Bytes of scratch tt_proj_bwd_f32 needs for (M, D, P). */
size_t tt_proj_bwd_workspace_bytes(int64_t M, int D, int P);

/*
 * Backward of the projection for the parameters only (the tower input does not
 * require grad: the encoder is frozen, backend/model.py:29-30,51).  Replaces
 * what loss.backward() (backend/training.py:50) does through model.py:33-38:
 *   dW2 = dOut^T H, db2 = sum_m dOut, dH = (dOut W2) * [H > 0],
 *   dW1 = dH^T X,   db1 = sum_m dH
 * accumulate != 0 adds into dW1/db1/dW2/db2 (the document tower is called for
 * positives and negatives, backend/training.py:41-42), else overwrites.
 * workspace: tt_proj_bwd_workspace_bytes() bytes, 16-byte aligned.
 * `tuning` (may be NULL): proj_tile, proj_bk, tn_tile, xcd_order.
 */
int tt_proj_bwd_f32(const float* dOut, const float* X, const float* H, const float* W2,
                    float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                    size_t workspace_bytes, int accumulate, int64_t M, int D, int P,
                    const tt_tuning* tuning, tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K3 — cosine-distance triplet loss, forward + gradient                      */
/* ------------------------------------------------------------------------- */

/*
 * pos_dist = 1 - cos(q,p), neg_dist = 1 - cos(q,n) (F.cosine_similarity, eps 1e-8),
 * loss = mean_b relu(pos_dist - neg_dist + margin).   Replaces backend/model.py:132-145.
 * If dq/dp/dn are non-NULL also writes grad_scale * d(sum_b hinge_b)/d{q,p,n}
 * (grad_scale = 1/B reproduces loss.backward(); 1/(B*k) gradient accumulation over
 * k batches, backend/training.py:93; 1/B_global data-parallel shards).
 *   q,p,n [B,P]; loss [1]; pos_dist,neg_dist [B] (may be NULL); hinge_rows [B] scratch
 */
int tt_triplet_fwd_bwd_f32(const float* q, const float* p, const float* n, float margin,
                           float grad_scale, int64_t B, int P, float* loss,
                           float* pos_dist, float* neg_dist, float* hinge_rows,
                           float* dq, float* dp, float* dn, tt_stream_t stream);

/*
 * dist[b] = 1 - cosine_similarity(x[b,:], y[b,:])  (eps 1e-8) — `TripletLoss.cosine_distance`, backend/model.py:132-135 —
 * and its backward: dx, dy = grad_dist[b] * d dist[b] / d{x, y}[b,:] (torch's autograd through
 * `1 - F.cosine_similarity(x, y, dim=1)`, including its behaviour for rows with |x| < eps), so that a caller may build
 * its own loss from the distances like it can with the reference's plain-torch method.   x, y [B,P]; dist, grad_dist [B].
 */
int tt_cosine_distance_f32(const float* x, const float* y, int64_t B, int P, float* dist, tt_stream_t stream);
int tt_cosine_distance_bwd_f32(const float* x, const float* y, const float* grad_dist, int64_t B, int P,
                               float* dx, float* dy, tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K4 — Adam                                                                   */
/* ------------------------------------------------------------------------- */

/*
 * One torch.optim.Adam step (defaults of backend/training.py:436: betas
 * (0.9,0.999), eps 1e-8, weight_decay 0, no amsgrad) over a flat fp32 array:
 *   g = grad_scale * sum_{z<nslabs} grads[z*slab_stride + i]
 *   m = m + (g - m)(1-b1);  v = v*b2 + (1-b2) g g
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)        t = step (1-based)
 * nslabs > 1 folds the split-K partial gradients of tt_engine into the update.
 * lr/betas/eps are doubles because torch derives 1-b1, 1-b2 and the bias
 * corrections in Python double precision before casting to fp32.
 */
int tt_adam_f32(float* params, const float* grads, int nslabs, int64_t slab_stride,
                float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                double beta2, double eps, int64_t step, float grad_scale, tt_stream_t stream);

/*
 * The same step over SEPARATE tensors in one launch (16 tensors per launch; more are taken in chunks) — what
 * `optimizer.step()` (backend/training.py:51) does for the 8 trainable tensors of model.parameters()
 * (backend/training.py:436) when they are ordinary nn.Parameters with autograd-produced .grad tensors rather than
 * views of one flat buffer.  `tensors`: HOST array; every device pointer 16-byte aligned; all tensors take step `step`.
 */
typedef struct {
  float* param;        /* [n] updated in place */
  const float* grad;   /* [n] */
  float* exp_avg;      /* [n] */
  float* exp_avg_sq;   /* [n] */
  int64_t n;
} tt_adam_tensor;
int tt_adam_multi_f32(const tt_adam_tensor* tensors, int count, double lr, double beta1, double beta2,
                      double eps, int64_t step, tt_stream_t stream);

/* grad_out[i] (+)= sum_z slabs[z*slab_stride+i] — sums split-K partial gradient slabs. */
int tt_reduce_slabs_f32(const float* slabs, int nslabs, int64_t slab_stride, float* out,
                        int64_t n, int accumulate, tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* In-batch-negatives triplet loss (BASELINE.json configs[4])  */
/* ------------------------------------------------------------------------- */

size_t tt_inbatch_workspace_bytes(int64_t Bq, int64_t Bd, int P);
/*
 * Every other document of the batch is a negative for query i (the reference samples ONE in-batch
 * positive per item as its negative, backend/data.py:113-137; it has no all-pairs loss, so this
 * entry point is pinned only by this repo's own CPU oracle):
 *   S[i,j] = cos(q_i, d_j)  (eps 1e-8);  positives: d row pos_index[i] (NULL: i, needs Bq == Bd)
 *   *loss  = (sum_i sum_{j valid} relu(S[i,j] - S[i,pos_i] + margin)) / loss_denominator
 *   valid j: j != pos_i and, if qid_q/qid_d are given, qid_d[j] != qid_q[i]
 *   dq [Bq,P], dd [Bd,P] = grad_scale * d(sum of hinges)/d{q,d}   (dd is added to if accumulate_dd)
 * The [Bq,Bd] similarity matrix, dS * d and dS^T * q are fp32 MFMA GEMMs.  Bq != Bd serves the
 * data-parallel case: local queries against the all-gathered documents of every rank.
 */
int tt_inbatch_triplet_fwd_bwd_f32(const float* q, const float* d, const int32_t* pos_index,
                                   const int64_t* qid_q, const int64_t* qid_d, float margin,
                                   float grad_scale, float loss_denominator, int64_t Bq, int64_t Bd,
                                   int P, float* loss, float* dq, float* dd, int accumulate_dd,
                                   void* workspace, size_t workspace_bytes, tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Evaluation — the arithmetic of evaluate_model on the device */
/* ------------------------------------------------------------------------- */

size_t tt_cosine_scores_workspace_bytes(int64_t Q, int64_t N, int P);
/*
 * out[i,j] = cosine_similarity(q[i,:], d[j,:]) (eps 1e-8: rows are divided by max(|x|, eps),
 * then dotted) for every (query, document) pair: two row-normalisation launches and one
 * exact-fp32 MFMA GEMM.  Replaces, for all queries at once,
 *   torch.cosine_similarity(query_embed.cpu(), doc_embeds.cpu(), dim=1)   (backend/training.py:297-299)
 *   q [Q,P], d [N,P], out [Q,N]; workspace: tt_cosine_scores_workspace_bytes(), 16-byte aligned.
 */
int tt_cosine_scores_f32(const float* q, const float* d, float* out, int64_t Q, int64_t N, int P,
                         void* workspace, size_t workspace_bytes, tt_stream_t stream);

/*
 * out[i, t] = sklearn.metrics.ndcg_score(gains[i:i+1], scores[i:i+1], k=ks[t]) (ignore_ties=False:
 * tied scores share the average gain of their tie group) — backend/training.py:301-311.
 *   scores, gains [Q,N] f32 (gains = true relevance, >= 0); valid [Q,N] u8 or NULL restricts
 *   row i to its candidate pool; ks_host: nk <= 8 cut-offs (host array); ks_device: int32[nk]
 *   device scratch; out [Q,nk].  A row without any positive gain scores 0, as in sklearn.
 */
int tt_ndcg_at_k_f32(const float* scores, const float* gains, const uint8_t* valid, int64_t Q,
                     int64_t N, const int32_t* ks_host, int nk, int32_t* ks_device, float* out,
                     tt_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Engine — the whole train_epoch batch body as one host call                  */
/* ------------------------------------------------------------------------- */

/*
 * Replaces the per-batch body of backend/training.py:36-53 (zero_grad, three
 * encodes, criterion, backward, optimizer.step, loss accumulation) for one
 * replica.  The engine borrows the caller's table, flat parameter / Adam state
 * buffers and its activation + split-K workspace; it allocates no device memory.
 *
 * Flat parameter layout (floats), the reference's model.parameters() order
 * (backend/model.py:33-38,84-85):
 *   query_tower.projection.0.weight [P,D], .0.bias [P], .2.weight [P,P], .2.bias [P],
 *   document_tower.projection.0.weight, .0.bias, .2.weight, .2.bias
 */
typedef struct tt_engine tt_engine;

/* pooled-input slots: gathers may run up to TT_ENGINE_SLOTS - 1 batches ahead of fwd_bwd */
#define TT_ENGINE_SLOTS 3

typedef struct {
  int64_t V;        /* table rows */
  int32_t D;        /* embedding dim */
  int32_t P;        /* projection dim */
  int32_t max_B;    /* max triplets per step on this replica */
  int32_t max_S;    /* max padded sequence length */
  int32_t device;   /* HIP device ordinal: must be the calling thread's current device at tt_engine_create (the handle's
                       second stream and events are created there) and the device of `workspace`, else
                       TT_ERR_WRONG_DEVICE — a rank that forgot hipSetDevice(local_rank) fails here, not later */
  int32_t k1_blocks_per_cu; /* How tt_engine_pool shares the chip with the MFMA kernels of
                               tt_engine_fwd_bwd running on another stream (both sides are steered
                               through the LDS a gather workgroup occupies):
                                 0                 not shared: no reservation, stand-alone tile choice
                                 TT_K1_SHARE_AUTO  the measured-best pairing for the table: XCD-sliced K1
                                                   sized just above a 64x64 projection tile's LDS, or the
                                                   per-text K1 at 2 workgroups per CU beside 128x64 tiles
                                 n >= 2            at most n gather workgroups per CU, 128x64 tiles (A/B) */
  int32_t grad_slabs; /* split-K slabs of the weight-gradient GEMMs (1..16; fixed for the handle's lifetime: the slab
                         workspace is sized by it); 0 = about 256 query rows per split, at least four where the batch
                         allows them (measured, docs/LOG_r04.md section 8) */
  int32_t k1_form;    /* TT_K1_* form a later tt_engine_set_tuning will force on this handle, or 0: the per-slot partial-sum
                         workspace is sized for the forms the shape rule can pick by itself plus this one (ABI 3; before,
                         every handle carried up to 64 MB per slot for forms nobody had asked for).  A tuning that forces a
                         form whose workspace was not sized falls back to the shape rule. */
  int32_t pack_roles; /* != 0: the workspace also holds, per slot, int32 staging ids [3, max_B, max_S] + prefix lengths for
                         tt_engine_pool_roles (the drop-in train_epoch route: 12 bytes per token slot; off for callers that
                         hand over packed [roles, B, S] ids themselves) */
} tt_engine_config;
#define TT_K1_SHARE_AUTO (-1)

/* Device bytes of activation + split-K workspace an engine of this config needs. */
size_t tt_engine_workspace_bytes(const tt_engine_config* cfg);
/* workspace: caller-owned device buffer (256-byte aligned, >= tt_engine_workspace_bytes),
 * borrowed until tt_engine_destroy; it is zero-filled on `stream` here. */
int tt_engine_create(const tt_engine_config* cfg, void* workspace, size_t workspace_bytes,
                     tt_stream_t stream, tt_engine** out);
int tt_engine_destroy(tt_engine* e);
/* floats in the flat parameter buffer: 2*(P*D + P + P*P + P) */
int64_t tt_engine_param_count(const tt_engine* e);
/* number of split-K gradient slabs the backward writes (>= 1) */
int tt_engine_grad_slabs(const tt_engine* e);

/*
 * Host-side hint (no device work): with k1_blocks_per_cu > 0, say whether token-id gathers
 * (tt_engine_pool on another stream) are actually running beside phase B right now.  Phase B then
 * picks its backward schedule: second-stream fork (shortest critical path under sharing) or the
 * fused single-stream form (fewer launches; faster when the chip is otherwise idle, e.g. batches
 * fed from the frozen-feature cache).  Default: sharing.
 */
int tt_engine_set_sharing(tt_engine* e, int gathers_share_chip);

/*
 * Force phase B's backward schedule (host-side, no device work): TT_BWD_AUTO (default: the rule
 * above), TT_BWD_FORK (loss mean + dW2/db2 on the engine's second stream under dH -> dW1/db1) or
 * TT_BWD_FUSED (loss mean inside the triplet launch, dW2/db2 + dH in one launch).  Both write
 * bitwise the same gradient slabs; the logged loss may differ in its last bit (summation order).
 * Which is faster depends on shape and on what shares the chip; TripletTrainer.calibrate() times
 * both on the device at hand.
 */
int tt_engine_set_backward(tt_engine* e, int mode);

/*
 * Bind a tuning to this handle (host-side, no device work; NULL = all auto).  Every later tt_engine_pool* /
 * tt_engine_fwd_bwd / tt_engine_forward / tt_engine_backward on the handle uses it; other handles are not
 * affected.  tuning->backward != TT_BWD_AUTO has the effect of tt_engine_set_backward.  Must be set before the
 * first tt_engine_pool* if it changes the K1 form (the LDS pairing of gather and projection tile is derived
 * from it per call, so it may also be changed between steps).
 */
int tt_engine_set_tuning(tt_engine* e, const tt_tuning* tuning);
int tt_engine_get_tuning(const tt_engine* e, tt_tuning* out);

/* Borrow caller-owned device buffers (valid until rebound or destroy). */
int tt_engine_bind(tt_engine* e, const float* table, float* params, float* exp_avg,
                   float* exp_avg_sq);

/*
 * Phase A (weight-independent): K1 over the 3B texts of one batch into pooled
 * slot `slot` (0 .. TT_ENGINE_SLOTS-1, so that the gathers of the next batches can run on
 * another stream under the projection kernels / gradient all-reduce of this one).
 *   ids  [roles,B,S] (q, pos[, neg] — the call order of backend/training.py:40-42)
 *   mask [roles,B,S] or NULL; lens [roles,B] int32 or NULL
 *   roles 3: independent negatives.  roles 2: negatives are in-batch positives and will be
 *   given to tt_engine_fwd_bwd as an index (no third gather) — backend/data.py:113-137 builds
 *   negatives exactly that way (`negatives[i] = batch[neg_idx]["positive"]`).
 */
int tt_engine_pool(tt_engine* e, int slot, const void* ids, int ids_dtype, const void* mask,
                   int mask_dtype, const int32_t* lens, int32_t roles, int32_t B, int32_t S,
                   tt_stream_t stream);

/*
 * Phase A through a row indirection: the collection's texts are tokenised once into a store
 * (store_ids [n_store, S] int32/int64 zero-padded rows, store_lens [n_store] int32 prefix
 * lengths or NULL = full length) that stays in HBM; a batch is then just rows [roles,B] int32 —
 * text t of the batch reads store row rows[t] — so the reference's per-batch tokenizer call +
 * `.to(device)` of three [B,S] int64 tensors (backend/model.py:43-48) becomes 12 bytes per
 * triplet, and no [roles,B,S] id tensor is materialised.  Same arithmetic and bitwise the same
 * result as tt_engine_pool on the gathered rows.  Rows outside [0, n_store) are clamped and
 * reported by tt_engine_read_stats like out-of-range token ids.
 */
int tt_engine_pool_rows(tt_engine* e, int slot, const void* store_ids, int ids_dtype,
                        const int32_t* store_lens, int64_t n_store, const int32_t* rows,
                        int32_t roles, int32_t B, int32_t S, tt_stream_t stream);

/*
 * K1 over roles handed over the way the reference's tokenizer call returns them (backend/model.py:43-48: one
 * `input_ids` [B, S_r] and one `attention_mask` [B, S_r] per encode call, int64, every role padded to ITS longest
 * text): one pass on `stream` squeezes every text's attended tokens (mask != 0), in order, into int32 ids
 * [n_roles, B, S_max] (TT_PAD_ID behind them) in the engine's workspace (tt_engine_config.pack_roles), then K1 runs
 * without mask or lengths — no [roles, B, S] int64 stacks on the caller's side and no mask re-reads in the gather
 * (measured: 17 us of a 210 us step at BASELINE.json configs[1]).  For 0/1 masks this is the reference's masked mean
 * (backend/model.py:63-72: a token with mask 0 adds nothing to either sum), bitwise the prefix-length form for right-padded
 * rows; a mask value other than 0 / 1 (a weight) cannot be expressed: it raises the flag tt_engine_read_stats reports as
 * TT_ERR_UNSUPPORTED — use tt_engine_pool with the mask for weighted pooling.  mask == NULL: every token attended.
 */
typedef struct {
  const void* ids;     /* [B, S] row-major, device */
  const void* mask;    /* [B, S] or NULL */
  int32_t ids_dtype;   /* TT_I32 / TT_I64 */
  int32_t mask_dtype;  /* TT_I32 / TT_I64 */
  int32_t S;           /* this role's padded length, 1..max_S */
  int32_t reserved;    /* must be 0 */
} tt_role_tokens;
int tt_engine_pool_roles(tt_engine* e, int slot, const tt_role_tokens* roles, int32_t n_roles, int32_t B,
                         tt_stream_t stream);
/* The two halves of tt_engine_pool_roles for a caller whose gather stream is the busier one: the packing pass on one
 * stream (*S_out = the padded length of the packed rows, the longest role's), the gather over the packed rows of `slot` on
 * another — ordered by the caller (an event between them).  Same results as the one-call form. */
int tt_engine_pack_roles(tt_engine* e, int slot, const tt_role_tokens* roles, int32_t n_roles, int32_t B, int32_t* S_out,
                         tt_stream_t stream);
int tt_engine_pool_packed(tt_engine* e, int slot, int32_t n_roles, int32_t B, int32_t S, tt_stream_t stream);

/*
 * Phase A from a frozen-feature cache.  The encoder is frozen (backend/model.py:29-30,51), so
 * the pooled-normalised vector of a text never changes: compute it once for every unique text
 * with tt_gather_pool_norm_f32 (`embed` mode) or encoder + tt_pool_norm_dense_f32 (`bert`
 * mode) into cache [cache_rows, D]; every later use of the text is one row copy.
 *   rows [roles,B] int32/int64: cache row of each text.
 */
int tt_engine_pool_cached(tt_engine* e, int slot, const float* cache, int64_t cache_rows,
                          const void* rows, int rows_dtype, int32_t roles, int32_t B,
                          tt_stream_t stream);

/*
 * Phase B: projection forward (both towers), triplet loss, backward.  Leaves
 * the parameter gradient as tt_engine_grad_slabs() partial slabs (summed by
 * tt_engine_reduce_grads or inside tt_engine_adam) and adds the batch loss to
 * the engine's device-side loss accumulator (no host sync:
 * backend/training.py:53's loss.item() is deferred to tt_engine_read_stats).
 *   grad_scale: 1/B for the reference semantics (mean over the batch);
 *               1/B_global under data parallelism; 1/(B*k) under accumulation.
 *   loss_denominator: the logged batch loss is (sum_b hinge_b) / loss_denominator —
 *               B for the reference semantics; B_global on a data-parallel shard,
 *               so that the per-rank values add up to the global batch loss.
 *   neg_index:  NULL (slot holds q, pos, neg), or int32[B] with values in [0,B) (slot holds q,
 *               pos): negative i is positive neg_index[i].  The document tower then runs on B
 *               rows instead of 2B and each row's gradient is dp[j] + sum_{i: neg_index[i]=j} dn[i]
 *               — the same numbers as the three-encode form up to fp32 summation order.
 */
int tt_engine_fwd_bwd(tt_engine* e, int slot, int32_t B, const int32_t* neg_index, float margin,
                      float grad_scale, float loss_denominator, tt_stream_t stream);

/*
 * Phase B in two halves, for a loss computed by the caller between them (the in-batch-negatives
 * loss, or any other function of the projected embeddings):
 *   tt_engine_forward : projection forward of the slot's rows -> tt_engine_embeddings() [roles*B, P]
 *   (caller writes d loss / d embeddings, same row layout, into tt_engine_dout())
 *   tt_engine_backward: parameter gradients -> slabs, as tt_engine_fwd_bwd leaves them
 * tt_engine_add_loss folds a device-side loss value into the engine's running loss statistics.
 */
int tt_engine_forward(tt_engine* e, int slot, int32_t B, int32_t roles, tt_stream_t stream);
int tt_engine_backward(tt_engine* e, int slot, int32_t B, int32_t roles, tt_stream_t stream);
float* tt_engine_dout(const tt_engine* e);                     /* [roles*B, P] */
int tt_engine_add_loss(tt_engine* e, const float* loss_value, tt_stream_t stream);

/* Diagnostic (synchronises `stream`): tt_gather_pool_stream_census of the streaming gather last launched into pooled
 * slot `slot` by tt_engine_pool* (all zeros when the engine does not use the streaming form for its table). */
int tt_engine_k1_census(tt_engine* e, int slot, int32_t* admitted, int32_t* cus_used, int32_t* max_registered,
                        int32_t* planned, tt_stream_t stream);

/* grad_out[param_count] (+)= sum of slabs — the flat bucket handed to the RCCL all-reduce. */
int tt_engine_reduce_grads(tt_engine* e, float* grad_out, int accumulate, tt_stream_t stream);
/*
 * The same over the parameters [offset, offset + count) only (grad_out is still the whole flat bucket):
 * lets the data-parallel step reduce and exchange the second-layer gradients (dW2 / db2, complete before
 * the dH -> dW1 chain starts) while the first-layer gradients are still being computed.
 */
int tt_engine_reduce_grads_range(tt_engine* e, float* grad_out, int64_t offset, int64_t count, int accumulate,
                                 tt_stream_t stream);
/*
 * Optional hand-off point inside tt_engine_fwd_bwd / tt_engine_backward: `event` (a hipEvent_t, NULL = none)
 * is recorded right after the launch that completes the dW2 / db2 slabs of both towers — on the caller's
 * stream (fused backward) or on the engine's second stream (fork backward).  A stream that waits for it may
 * read those slabs while dH -> dW1 / db1 are still running.
 */
typedef void* tt_event_t; /* hipEvent_t */
int tt_engine_set_dw2_event(tt_engine* e, tt_event_t event);

/*
 * Optional hand-off point inside tt_engine_fwd_bwd / tt_engine_backward: `event` (NULL = none) is recorded on the caller's
 * stream right after the loss kernel — everything phase B of EARLIER batches did lies before it in stream order, and
 * so does this batch's forward.  In the fork backward it IS the fork point (the engine's second stream waits for this
 * record instead of one of its own), so a caller whose gather stream needs "phase B of the previous batch has
 * finished with its pooled slot" gets that edge without an event packet of its own on the critical stream.
 * Setting a non-NULL event ARMS it (see tt_engine_arm_loss_event): both backward schedules record it from then on, so a
 * caller that sets the event and waits for it is always waiting for a record that happened.  NULL disarms.
 */
int tt_engine_set_loss_event(tt_engine* e, tt_event_t event);
/*
 * The fork backward always records the loss event (it IS its fork point).  The fused backward (one stream, no event of
 * its own) records it only while `armed` != 0 — a caller that is NOT going to wait for it this step may disarm it and
 * save the packet on the critical chain (round-4 advisor finding).  Off after tt_engine_create, on after
 * tt_engine_set_loss_event(non-NULL); waiting for the event of a step during which it was disarmed under the fused
 * backward waits for an OLDER record (or none).
 */
int tt_engine_arm_loss_event(tt_engine* e, int32_t armed);

/*
 * One-shot: the NEXT tt_engine_fwd_bwd / tt_engine_backward, if it runs the fork backward, makes its second stream wait
 * for `event` before it records the join — so that the caller's stream, which waits for the join anyway, is thereby also
 * ordered behind `event` (e.g. the gather of the next batch, training.py:39-48 of the step after) with ONE wait packet
 * instead of two on the critical stream.  *accepted = 1 if the engine took it (fork schedule), 0 if it did not (fused
 * schedule, NULL event): then the caller orders its stream itself.  Cleared by that call either way.
 */
int tt_engine_set_join_wait(tt_engine* e, tt_event_t event, int32_t* accepted);

/*
 * Events that order streams of ONE device against each other — what the host side needs around tt_engine_pool* on a
 * side stream: "the gather may overwrite this slot once phase B of three batches ago has finished", "phase B may read
 * the slot once the gather has finished".  A default hipEvent performs a system-scope release (cache write-back)
 * when it is recorded, which the kernel behind it on the same stream waits for; same-device ordering needs no more than
 * device scope.  scope: 0 default, 1 hipEventDisableSystemFence, 2 hipEventReleaseToDevice.  Timing is disabled.
 * tt_event_done: 1 = everything recorded before the event has completed, 0 = not yet.
 */
int tt_event_create(int scope, tt_event_t* out);
int tt_event_destroy(tt_event_t event);
int tt_event_record(tt_event_t event, tt_stream_t stream);
int tt_stream_wait_event(tt_stream_t stream, tt_event_t event);
int tt_event_done(tt_event_t event);

/*
 * Phase C: Adam over the bound flat buffers.  grads == NULL: consume the
 * engine's own slabs (single replica); else a reduced flat gradient
 * (after all-reduce / accumulation).  step is 1-based.
 */
int tt_engine_adam(tt_engine* e, const float* grads, double lr, double beta1, double beta2,
                   double eps, int64_t step, tt_stream_t stream);

/*
 * Phase C with the step counter on the device: every call advances it by one and derives the
 * bias corrections there (in double, like torch does on the host), so the launch sequence has
 * no step-dependent argument and can be captured once in a hipGraph and replayed.
 * tt_engine_set_step sets the counter (0 = fresh optimizer) — e.g. after steps taken with
 * tt_engine_adam, which does not touch it.
 */
int tt_engine_adam_auto(tt_engine* e, const float* grads, double lr, double beta1, double beta2,
                        double eps, tt_stream_t stream);
int tt_engine_set_step(tt_engine* e, int64_t step, tt_stream_t stream);

/* A+B+C on one stream: one reference batch body.  */
int tt_engine_train_step(tt_engine* e, const void* ids, int ids_dtype, const void* mask,
                         int mask_dtype, const int32_t* lens, int32_t B, int32_t S,
                         float margin, double lr, int64_t step, tt_stream_t stream);

/* Device pointers into the engine's activations for the last fwd_bwd on `slot`
 * (rows: [q(B); pos(B); neg(B)]).  Valid until the next call on that slot. */
const float* tt_engine_pooled(const tt_engine* e, int slot);   /* [roles*B, D] */
const float* tt_engine_embeddings(const tt_engine* e);         /* [roles*B, P] projected */
const float* tt_engine_grad_slab0(const tt_engine* e);         /* [slabs, param_count] */
/* The device scalar every loss kernel leaves the batch loss in (what tt_engine_read_stats returns as last_loss): a caller
 * that wants a batch's loss without stopping the stream copies it out stream-ordered (backend/training.py:59-61 prints
 * every tenth batch's loss; a synchronous read there drains the pipeline). */
const float* tt_engine_last_loss(const tt_engine* e);

/* Host-side read of {sum of batch losses, number of batches, last loss}; synchronises
 * `stream`.  reset != 0 zeroes the accumulator afterwards. */
int tt_engine_read_stats(tt_engine* e, double* loss_sum, int64_t* n_batches, float* last_loss,
                         int reset, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TT_HOTPATH_H */
