// tt_engine — the per-batch body of backend/training.py:36-53 for one replica,
// as stream-ordered launches over caller-owned table / parameter / Adam-state
// buffers and a caller-owned workspace (activations + split-K gradient slabs).
//
//   pool     : K1 over [q; pos; neg] (3B texts, one launch)            -> pooled[slot]
//   fwd_bwd  : K2 fwd (both towers grouped: query rows 0..B, document rows B..3B so
//              the positive and negative calls of training.py:41-42 share one GEMM
//              and their gradients accumulate inside it), K3, K2 bwd     -> grad slabs
//   adam     : K4, folding the slab sum into the update
#include <new>
#include "proj.h"

#include <mutex>
#include <vector>

// text_rows != NULL: text t reads row text_rows[t] of a token store [n_store, S] (ids / mask / lens are the store's)
int tt_k1_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                 const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                 unsigned lds_total, hipStream_t st, const int32_t* text_rows, int64_t n_store);
int tt_k1_sliced_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, unsigned lds_total, int form, hipStream_t st,
                        const int32_t* text_rows, int64_t n_store);
int tt_k1_sliced_supported(int S, int D, const float* table, const float* out);
int tt_k1_binned_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, unsigned lds_total, hipStream_t st,
                        const int32_t* text_rows, int64_t n_store);
size_t tt_k1_colsliced_workspace_bytes(int64_t N, int D);
int tt_k1_stream_launch(const float* table, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                        const int32_t* lens, float* out, int64_t N, int S, int D, int64_t V, int32_t* oob_flag,
                        void* workspace, size_t workspace_bytes, int per_cu, const tt_tuning* tuning,
                        unsigned lds_total, hipStream_t st, const int32_t* text_rows, int64_t n_store);
int tt_launch_adam_auto(float* params, const float* grads, int nslabs, long long slab_stride, float* exp_avg,
                        float* exp_avg_sq, long long n, double lr, double beta1, double beta2, double eps,
                        long long* dev_step, float* dev_scal, hipStream_t st);
int tt_launch_set_step(long long* dev_step, long long value, hipStream_t st);
int tt_launch_loss_mean(const float* hinge_rows, long long B, float denom, float* loss, double* acc_sum,
                        long long* acc_count, hipStream_t st);
int tt_launch_triplet_rows(const float* q, const float* p, const float* n, float margin, float grad_scale,
                           long long B, int P, float* pos_dist, float* neg_dist, float* hinge_rows,
                           float* dq, float* dp, float* dn, const int* neg_index, hipStream_t st,
                           float* mean_partials = nullptr, unsigned* mean_ticket = nullptr, float mean_denom = 1.f,
                           float* mean_loss = nullptr, double* acc_sum = nullptr, long long* acc_count = nullptr,
                           int* csr_ws = nullptr);

struct tt_engine {
  uint32_t magic;
  tt_engine_config cfg;
  int64_t tower_n;   // floats per tower: P*D + P + P*P + P
  int64_t nparam;    // 2 * tower_n
  int64_t slab_stride;
  int nslabs;
  // borrowed
  const float* table;
  float* params;
  float* exp_avg;
  float* exp_avg_sq;
  // carved out of the caller's workspace
  char* arena;
  float* pooled[TT_ENGINE_SLOTS];
  float* k1_partial[TT_ENGINE_SLOTS];  // XCD-sliced K1: per-slice partial sums + mask sums of one batch
  size_t k1_partial_bytes;
  int32_t* stage_ids[TT_ENGINE_SLOTS];   // tt_engine_pool_roles: packed int32 ids [3, max_B, max_S] of one batch (cfg.pack_roles)
  int32_t staged[TT_ENGINE_SLOTS][3];    // (n_roles, B, S_out) of the last tt_engine_pack_roles into the slot; zeros = nothing staged
  int32_t* stage_lens[TT_ENGINE_SLOTS];  // ... and their prefix lengths [3, max_B]
  void* k1_stream_ctl[TT_ENGINE_SLOTS];  // streaming K1: queue head + per-CU registry of one launch (every engine has them: 16 KB each)
  bool k1_streamed[TT_ENGINE_SLOTS];     // the last tt_engine_pool* into this slot used the streaming form
  bool stream_beside;                    // the most recent tt_engine_pool* streamed: phase B pairs its tiles with that
  tt_tuning tuning;                      // tt_engine_set_tuning (all zero: auto)
  // sharing with the gathers (cfg.k1_blocks_per_cu != 0), see gather_pool.hip "Sharing the chip"
  bool shared;           // gathers run beside phase B on another stream
  bool tile_above;       // sliced K1 reserved just above the 64x64 projection tile's LDS (else: capped K1, 128x64 tiles)
  int cap;               // capped mode: gather workgroups per CU
  float* H;
  float* emb;
  float* dOut;
  float* dH;
  float* slabs;
  float* hinge;
  float* pos_dist;
  float* neg_dist;
  float* loss;
  double* loss_sum;
  long long* loss_cnt;
  int32_t* oob;
  float* mean_partials;  // per-workgroup hinge sums of the triplet kernel (in-launch loss mean)
  unsigned* mean_ticket;
  int* neg_csr;         // 4 * max_B + 1 ints: sources of every negative-by-index destination row (large batches)
  long long* dev_step;  // optimizer steps taken, device copy (tt_engine_adam_auto)
  float* dev_scal;      // {step_size, bc2_sqrt} of the current device step
  // second stream: dW2/db2 run there, under the dH -> dW1/db1 chain (see backward_chain)
  hipStream_t aux;
  hipEvent_t ev_fork[3], ev_join[3];  // [tt_tuning.event_scope]: default / no system fence / device-scope release
  bool gathers_share_chip;  // tt_engine_set_sharing: token-id gathers are running beside phase B
  int bwd_mode;             // tt_engine_set_backward: TT_BWD_AUTO / TT_BWD_FORK / TT_BWD_FUSED
  hipEvent_t ev_dw2;        // tt_engine_set_dw2_event: recorded once the dW2 / db2 slabs are complete (may be null)
  hipEvent_t ev_loss;       // tt_engine_set_loss_event: recorded on the caller's stream right after the loss kernel (may be null)
  bool loss_event_armed;    // tt_engine_arm_loss_event: the fused backward records ev_loss too (the fork backward always does: it IS its fork point)
  hipEvent_t join_wait;     // tt_engine_set_join_wait: one-shot, the second stream waits for it before the next join (may be null)
};

namespace {
// The engine's second stream and its fork / join events are RECYCLED between handles of a device, never destroyed — what
// torch does with its own stream pool.  Round 6: with engines created and destroyed between the captures of hipGraphs (a
// trainer with use_graphs per test), the HIP runtime's graph launch crashed in hip::Graph::UpdateStreams depending on which
// engines had been destroyed before (tools/r06_graph_probe2.py); a process that only ever creates these objects does not
// exercise that path.  The free list is the library's one piece of process-wide state: a cache of idle HIP objects, no
// behaviour hangs on it.
struct ForkKit {
  int device;
  hipStream_t aux;
  hipEvent_t ev_fork[3], ev_join[3];
};
std::mutex g_kits_mu;
std::vector<ForkKit> g_kits;

bool take_fork_kit(int device, tt_engine* e) {
  std::lock_guard<std::mutex> lock(g_kits_mu);
  for (size_t i = 0; i < g_kits.size(); ++i) {
    if (g_kits[i].device != device) continue;
    e->aux = g_kits[i].aux;
    for (int k = 0; k < 3; ++k) { e->ev_fork[k] = g_kits[i].ev_fork[k]; e->ev_join[k] = g_kits[i].ev_join[k]; }
    g_kits.erase(g_kits.begin() + (long)i);
    return true;
  }
  return false;
}

void give_fork_kit(tt_engine* e) {
  ForkKit kit{};
  kit.device = e->cfg.device;
  kit.aux = e->aux;
  for (int k = 0; k < 3; ++k) { kit.ev_fork[k] = e->ev_fork[k]; kit.ev_join[k] = e->ev_join[k]; }
  std::lock_guard<std::mutex> lock(g_kits_mu);
  g_kits.push_back(kit);
}

constexpr uint32_t kMagic = 0x7754454eu;
inline bool ok(const tt_engine* e) { return e && e->magic == kMagic; }
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += up256(n * sizeof(T));
    return p;
  }
};

void carve(tt_engine* e, char* base, size_t* total) {
  Carver c{base};
  const size_t rows = 3 * (size_t)e->cfg.max_B;
  for (int i = 0; i < TT_ENGINE_SLOTS; ++i) e->pooled[i] = c.take<float>(rows * e->cfg.D);
  // per-slot partial sums of the sliced / binned K1 forms: sized for what the shape rule can pick for this table by
  // itself (any N, S up to the handle's maxima) and for the form the caller says a tuning will force (cfg.k1_form)
  const int want = e->cfg.k1_form;
  const bool auto_sliced = tt_k1_prefers_sliced(e->cfg.V, e->cfg.D, (int64_t)1 << 40, 128);
  const bool rows_form = want == TT_K1_ROW_SLICES ||
                         (auto_sliced && tt_k1_sliced_form(e->cfg.V, e->cfg.D, 128) == TT_K1_ROW_SLICES);
  e->k1_partial_bytes = rows_form ? tt_gather_pool_sliced_workspace_bytes((int64_t)rows, e->cfg.D)
                                  : tt_k1_colsliced_workspace_bytes((int64_t)rows, e->cfg.D);  // (the column form's is tiny)
  const size_t binned_bytes = tt_gather_pool_binned_workspace_bytes((int64_t)rows, e->cfg.D, e->cfg.max_S);
  if (binned_bytes > e->k1_partial_bytes &&
      (want == TT_K1_BINNED || tt_k1_prefers_binned(e->cfg.V, e->cfg.D, (int64_t)rows, e->cfg.max_S)))
    e->k1_partial_bytes = binned_bytes;
  for (int i = 0; i < TT_ENGINE_SLOTS; ++i)
    e->k1_partial[i] = e->k1_partial_bytes ? c.take<float>(e->k1_partial_bytes / sizeof(float)) : nullptr;
  for (int i = 0; i < TT_ENGINE_SLOTS; ++i) e->k1_stream_ctl[i] = c.take<char>(tt_gather_pool_stream_workspace_bytes());
  for (int i = 0; i < TT_ENGINE_SLOTS; ++i) {
    e->stage_ids[i] = e->cfg.pack_roles ? c.take<int32_t>(rows * (size_t)e->cfg.max_S) : nullptr;
    e->stage_lens[i] = e->cfg.pack_roles ? c.take<int32_t>(rows) : nullptr;
  }
  e->H = c.take<float>(rows * e->cfg.P);
  e->emb = c.take<float>(rows * e->cfg.P);
  e->dOut = c.take<float>(rows * e->cfg.P);
  e->dH = c.take<float>(rows * e->cfg.P);
  e->slabs = c.take<float>((size_t)e->nslabs * e->slab_stride);
  e->hinge = c.take<float>(e->cfg.max_B);
  e->pos_dist = c.take<float>(e->cfg.max_B);
  e->neg_dist = c.take<float>(e->cfg.max_B);
  e->loss = c.take<float>(4);
  e->loss_sum = c.take<double>(2);
  e->loss_cnt = c.take<long long>(2);
  e->oob = c.take<int32_t>(4);
  e->mean_partials = c.take<float>((size_t)e->cfg.max_B / 4 + 4);
  e->mean_ticket = c.take<unsigned>(4);
  e->neg_csr = c.take<int>(4 * (size_t)e->cfg.max_B + 1);
  e->dev_step = c.take<long long>(2);
  e->dev_scal = c.take<float>(4);
  *total = c.off;
}
}  // namespace

static int fill_dims(tt_engine* e, const tt_engine_config* cfg) {
  TT_REQUIRE(cfg->V >= 1 && cfg->V <= 0x7fffffffLL && cfg->D >= 1 && cfg->P >= 1 && cfg->P <= 1024 &&
                 cfg->max_B >= 1 && cfg->max_S >= 1 && (int64_t)cfg->max_B * 3 <= 0x7fffffffLL &&
                 cfg->k1_blocks_per_cu >= TT_K1_SHARE_AUTO && cfg->grad_slabs >= 0 && cfg->grad_slabs <= 16 &&
                 cfg->k1_form >= TT_K1_AUTO && cfg->k1_form <= TT_K1_BINNED,
             TT_ERR_BAD_SHAPE);
  e->cfg = *cfg;
  e->tower_n = (int64_t)cfg->P * cfg->D + cfg->P + (int64_t)cfg->P * cfg->P + cfg->P;
  e->nparam = 2 * e->tower_n;
  e->slab_stride = (e->nparam + 3) / 4 * 4;
  e->nslabs = cfg->grad_slabs > 0 ? cfg->grad_slabs : ttp::choose_nsplit_towers(cfg->max_B, cfg->P, cfg->D);
  return TT_OK;
}

extern "C" size_t tt_engine_workspace_bytes(const tt_engine_config* cfg) {
  if (!cfg) return 0;
  tt_engine tmp{};
  if (fill_dims(&tmp, cfg) != TT_OK) return 0;
  size_t total = 0;
  carve(&tmp, nullptr, &total);
  return total;
}

extern "C" int tt_engine_create(const tt_engine_config* cfg, void* workspace, size_t workspace_bytes,
                                tt_stream_t stream, tt_engine** out) {
  TT_REQUIRE(cfg && out && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, TT_ERR_ALIGNMENT);
  {  // one handle per device/rank: the second stream and the events below are created on the CURRENT device
    int cur = -1;
    TT_REQUIRE(hipGetDevice(&cur) == hipSuccess && cur == cfg->device, TT_ERR_WRONG_DEVICE);
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, workspace) == hipSuccess) {
      TT_REQUIRE(attr.device == cfg->device, TT_ERR_WRONG_DEVICE);
    } else {
      (void)hipGetLastError();  // a pointer the runtime cannot describe: left to the first launch to reject
    }
  }
  tt_engine* e = new (std::nothrow) tt_engine();
  TT_REQUIRE(e, TT_ERR_WORKSPACE);
  int rc = fill_dims(e, cfg);
  size_t total = 0;
  if (rc == TT_OK) {
    carve(e, nullptr, &total);
    if (workspace_bytes < total) rc = TT_ERR_WORKSPACE;
  }
  if (rc == TT_OK) {
    hipError_t herr = hipMemsetAsync(workspace, 0, total, static_cast<hipStream_t>(stream));
    if (herr != hipSuccess) rc = (int)herr;
  }
  if (rc != TT_OK) {
    delete e;
    return rc;
  }
  e->arena = static_cast<char*>(workspace);
  carve(e, e->arena, &total);
  // the fork/join stream is optional: without it the backward is one chain on the caller's stream
  if (!take_fork_kit(cfg->device, e)) {
    if (hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking) != hipSuccess) e->aux = nullptr;
    for (int k = 0; k < 3 && e->aux; ++k) {
      const unsigned flags = tt_event_flags(k);
      if (hipEventCreateWithFlags(&e->ev_fork[k], flags) != hipSuccess) {
        e->ev_fork[k] = nullptr;
      } else if (hipEventCreateWithFlags(&e->ev_join[k], flags) != hipSuccess) {
        (void)hipEventDestroy(e->ev_fork[k]);
        e->ev_fork[k] = nullptr;
      }
      if (!e->ev_fork[k]) {  // without its events the second stream is of no use: one chain on the caller's stream
        for (int j = 0; j < k; ++j) {
          (void)hipEventDestroy(e->ev_fork[j]);
          (void)hipEventDestroy(e->ev_join[j]);
        }
        (void)hipStreamDestroy(e->aux);
        e->aux = nullptr;
      }
    }
  }
  e->gathers_share_chip = true;
  e->shared = cfg->k1_blocks_per_cu != 0;
  e->tile_above = cfg->k1_blocks_per_cu == TT_K1_SHARE_AUTO && tt_k1_prefers_sliced(cfg->V, cfg->D, 3 * (int64_t)cfg->max_B, cfg->max_S) &&
                  e->k1_partial_bytes >= (tt_k1_sliced_form(cfg->V, cfg->D, cfg->max_S) == TT_K1_ROW_SLICES
                                              ? tt_gather_pool_sliced_workspace_bytes(3 * (int64_t)cfg->max_B, cfg->D)
                                              : tt_k1_colsliced_workspace_bytes(3 * (int64_t)cfg->max_B, cfg->D));
  e->cap = cfg->k1_blocks_per_cu > 0 ? cfg->k1_blocks_per_cu : (cfg->k1_blocks_per_cu == TT_K1_SHARE_AUTO ? 2 : 0);
  e->magic = kMagic;
  *out = e;
  return TT_OK;
}

extern "C" int tt_engine_destroy(tt_engine* e) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  if (e->aux) {
    (void)hipStreamSynchronize(e->aux);
    give_fork_kit(e);  // the stream and its six events go back to the free list: never destroyed (see ForkKit)
  }
  e->magic = 0;
  delete e;
  return TT_OK;
}

extern "C" int tt_engine_set_sharing(tt_engine* e, int gathers_share_chip) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  e->gathers_share_chip = gathers_share_chip != 0;
  return TT_OK;
}

extern "C" int tt_engine_set_backward(tt_engine* e, int mode) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(mode == TT_BWD_AUTO || mode == TT_BWD_FORK || mode == TT_BWD_FUSED, TT_ERR_BAD_SHAPE);
  e->bwd_mode = mode;
  return TT_OK;
}

extern "C" int tt_engine_set_tuning(tt_engine* e, const tt_tuning* tuning) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  const tt_tuning t = tuning ? *tuning : tt_tuning{};
  TT_REQUIRE(t.k1_form >= TT_K1_AUTO && t.k1_form <= TT_K1_BINNED, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(t.k1_stream_wgs >= 0 && t.k1_stream_wgs <= 8 && t.k1_share_lds_kb >= -1 && t.k1_share_lds_kb <= 160, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(t.backward == TT_BWD_AUTO || t.backward == TT_BWD_FORK || t.backward == TT_BWD_FUSED, TT_ERR_BAD_SHAPE);
  for (int r : t.reserved) TT_REQUIRE(r == 0, TT_ERR_BAD_SHAPE);
  e->tuning = t;
  if (t.backward != TT_BWD_AUTO) e->bwd_mode = t.backward;
  return TT_OK;
}

extern "C" int tt_engine_get_tuning(const tt_engine* e, tt_tuning* out) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(out, TT_ERR_NULL_POINTER);
  *out = e->tuning;
  return TT_OK;
}

extern "C" int64_t tt_engine_param_count(const tt_engine* e) { return ok(e) ? e->nparam : -1; }
extern "C" int tt_engine_grad_slabs(const tt_engine* e) { return ok(e) ? e->nslabs : -1; }
extern "C" const float* tt_engine_pooled(const tt_engine* e, int slot) {
  return ok(e) && slot >= 0 && slot < TT_ENGINE_SLOTS ? e->pooled[slot] : nullptr;
}
extern "C" const float* tt_engine_embeddings(const tt_engine* e) { return ok(e) ? e->emb : nullptr; }
extern "C" const float* tt_engine_grad_slab0(const tt_engine* e) { return ok(e) ? e->slabs : nullptr; }
extern "C" const float* tt_engine_last_loss(const tt_engine* e) { return ok(e) ? e->loss : nullptr; }

extern "C" int tt_engine_bind(tt_engine* e, const float* table, float* params, float* exp_avg,
                              float* exp_avg_sq) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(table && params && exp_avg && exp_avg_sq, TT_ERR_NULL_POINTER);
  TT_REQUIRE(tt_aligned16(table) && tt_aligned16(params) && tt_aligned16(exp_avg) && tt_aligned16(exp_avg_sq),
             TT_ERR_ALIGNMENT);
  e->table = table;
  e->params = params;
  e->exp_avg = exp_avg;
  e->exp_avg_sq = exp_avg_sq;
  return TT_OK;
}

// The projection plan phase B runs with: tiles that pair with how the gathers are reserved (gather_pool.hip "Sharing
// the chip") + the handle's tuning.  Beside the streaming gather: 128x128 K-step-16 tiles in XCD order — chosen from
// what the most recent tt_engine_pool* actually launched, not from the table alone (a short batch of a big table
// does not stream).
static ttp::Plan chain_plan(const tt_engine* e) {
  int policy = !e->shared ? ttp::kTileAuto : (e->tile_above ? ttp::kTileSmall : ttp::kTileLowTraffic);
  if (e->shared && e->stream_beside) policy = ttp::kTileStream;
  ttp::Plan pl = ttp::plan_of(&e->tuning, policy);
  pl.beside_gather = e->shared && e->gathers_share_chip;
  return pl;
}

static int engine_k1(tt_engine* e, int slot, const void* ids, int ids_dtype, const void* mask, int mask_dtype,
                     const int32_t* lens, int32_t roles, int32_t B, int32_t S, const int32_t* text_rows,
                     int64_t n_store, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->table, TT_ERR_NULL_POINTER);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS && (roles == 2 || roles == 3), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B && S >= 1 && S <= e->cfg.max_S, TT_ERR_BAD_SHAPE);
  const tt_tuning& tn = e->tuning;
  const int64_t N = roles * (int64_t)B;
  const bool auto_share = e->shared && e->cfg.k1_blocks_per_cu == TT_K1_SHARE_AUTO;
  // which form: the tuning's, else the measured rule for the shape
  int form = tn.k1_form;
  if (form == TT_K1_STREAM && !tt_k1_stream_supported(e->cfg.D, e->table, e->pooled[slot])) form = TT_K1_AUTO;
  const size_t rows_ws = tt_gather_pool_sliced_workspace_bytes(N, e->cfg.D), cols_ws = tt_k1_colsliced_workspace_bytes(N, e->cfg.D);
  if ((form == TT_K1_ROW_SLICES || form == TT_K1_COLUMN_SLICES) &&
      !(tt_k1_sliced_supported(S, e->cfg.D, e->table, e->pooled[slot]) &&
        (form == TT_K1_ROW_SLICES ? e->k1_partial_bytes >= rows_ws : (cols_ws != 0 && e->k1_partial_bytes >= cols_ws))))
    form = TT_K1_AUTO;
  if (form == TT_K1_BINNED && !(tt_k1_binned_supported(S, e->cfg.D, e->table, e->pooled[slot]) &&
                                e->k1_partial_bytes >= tt_gather_pool_binned_workspace_bytes(N, e->cfg.D, S)))
    form = TT_K1_AUTO;
  if (form == TT_K1_AUTO) {
    if (tt_k1_prefers_binned(e->cfg.V, e->cfg.D, N, S) && tt_k1_binned_supported(S, e->cfg.D, e->table, e->pooled[slot]) &&
        e->k1_partial_bytes >= tt_gather_pool_binned_workspace_bytes(N, e->cfg.D, S))
      form = TT_K1_BINNED;  // a big table whose rows are named many times per batch: served from the XCDs' L2s
    else if (auto_share && tt_k1_prefers_stream(e->cfg.V, e->cfg.D, N, S) &&
        tt_k1_stream_supported(e->cfg.D, e->table, e->pooled[slot]))
      form = TT_K1_STREAM;  // beside the projection chain when the table really comes from HBM (gather_pool.hip)
    else if (tt_k1_prefers_sliced(e->cfg.V, e->cfg.D, N, S) && tt_k1_sliced_supported(S, e->cfg.D, e->table, e->pooled[slot]) &&
             e->k1_partial_bytes >= (tt_k1_sliced_form(e->cfg.V, e->cfg.D, S) == TT_K1_ROW_SLICES ? rows_ws : cols_ws) &&
             (tt_k1_sliced_form(e->cfg.V, e->cfg.D, S) == TT_K1_ROW_SLICES || cols_ws != 0))
      form = tt_k1_sliced_form(e->cfg.V, e->cfg.D, S);  // table several times an XCD's L2 (tt_k1_prefers_sliced)
    else
      form = TT_K1_PER_TEXT;
  }
  e->k1_streamed[slot] = form == TT_K1_STREAM;
  e->stream_beside = form == TT_K1_STREAM || form == TT_K1_BINNED;  // the projection tiles that pair with a big table's gather
  // LDS a gather workgroup occupies while it shares the chip (0 = no reservation)
  unsigned lds_total = 0;
  if (e->shared)
    lds_total = e->tile_above ? (unsigned)ttp::chain_lds_bytes(3 * (long long)e->cfg.max_B, e->cfg.P,
                                                               ttp::plan_of(&e->tuning, ttp::kTileSmall)) + 3072u
                              : tt_k1_lds_for_cap(e->cap);
  if (e->shared && tn.k1_share_lds_kb > 0) lds_total = (unsigned)tn.k1_share_lds_kb * 1024u;
  if (e->shared && tn.k1_share_lds_kb < 0) lds_total = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (form == TT_K1_STREAM) {
    // two 4-wave workgroups x 6 rows x C KiB in flight per CU; leaves room for two 128x128 K-step-16 tiles.
    // The streaming workgroups hold their own 12 KB; a reservation applies only when the tuning asks for one.
    const int per_cu = tn.k1_stream_wgs > 0 ? tn.k1_stream_wgs : 2;
    const unsigned lds = tn.k1_share_lds_kb > 0 ? lds_total : 0;
    return tt_k1_stream_launch(e->table, ids, ids_dtype, mask, mask_dtype, lens, e->pooled[slot], N, S, e->cfg.D, e->cfg.V,
                               e->oob, e->k1_stream_ctl[slot], tt_gather_pool_stream_workspace_bytes(), per_cu, &e->tuning,
                               lds, st, text_rows, n_store);
  }
  if (form == TT_K1_BINNED)
    return tt_k1_binned_launch(e->table, ids, ids_dtype, mask, mask_dtype, lens, e->pooled[slot], N, S, e->cfg.D, e->cfg.V,
                               e->oob, e->k1_partial[slot], e->k1_partial_bytes, tn.k1_share_lds_kb != 0 ? lds_total : 0, st,
                               text_rows, n_store);
  if (form == TT_K1_ROW_SLICES || form == TT_K1_COLUMN_SLICES) {
    if (e->shared && !e->tile_above && auto_share && tn.k1_share_lds_kb == 0) lds_total = tt_k1_lds_for_cap(3);
    return tt_k1_sliced_launch(e->table, ids, ids_dtype, mask, mask_dtype, lens, e->pooled[slot], N, S, e->cfg.D, e->cfg.V,
                               e->oob, e->k1_partial[slot], e->k1_partial_bytes, lds_total, form, st, text_rows, n_store);
  }
  return tt_k1_launch(e->table, ids, ids_dtype, mask, mask_dtype, lens, e->pooled[slot], N, S, e->cfg.D, e->cfg.V, e->oob,
                      lds_total, st, text_rows, n_store);
}

extern "C" int tt_engine_pool(tt_engine* e, int slot, const void* ids, int ids_dtype, const void* mask,
                              int mask_dtype, const int32_t* lens, int32_t roles, int32_t B, int32_t S,
                              tt_stream_t stream) {
  return engine_k1(e, slot, ids, ids_dtype, mask, mask_dtype, lens, roles, B, S, nullptr, 0, stream);
}

extern "C" int tt_engine_pool_rows(tt_engine* e, int slot, const void* store_ids, int ids_dtype,
                                   const int32_t* store_lens, int64_t n_store, const int32_t* rows, int32_t roles,
                                   int32_t B, int32_t S, tt_stream_t stream) {
  TT_REQUIRE(store_ids && rows, TT_ERR_NULL_POINTER);
  TT_REQUIRE(n_store >= 1 && n_store <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
  return engine_k1(e, slot, store_ids, ids_dtype, nullptr, TT_I32, store_lens, roles, B, S, rows, n_store, stream);
}

// One wave per (role, text): the attended tokens (mask != 0, id not already TT_PAD_ID) of a [B, S_r] block, in order, as int32
// ids at the front of row (role, text) of out_ids [roles, B, S_out]; TT_PAD_ID behind them; lens = their count.  flags[1] |= 1
// where a mask value is neither 0 nor 1 (weighted pooling cannot be packed), ids outside int32 become -1 (K1 reports them).
__device__ __forceinline__ bool tt_is_pad_host(long long raw) { return raw == TT_PAD_ID32 || raw == (-9223372036854775807LL - 1LL); }
struct PackRoles {
  const void* ids[3];
  const void* mask[3];
  int ids_w[3], mask_w[3], S[3];
  int n_roles;
};
__global__ __launch_bounds__(256) void pack_roles_kernel(const PackRoles a, int B, int S_out, int32_t* __restrict__ out_ids,
                                                         int32_t* __restrict__ out_lens, int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= (long long)a.n_roles * B) return;
  const int role = (int)(t / B);
  const long long text = t - (long long)role * B;
  const int S = a.S[role];
  const void* __restrict__ ip = a.ids[role];
  const void* __restrict__ mp = a.mask[role];
  const bool i64 = a.ids_w[role] == TT_I64, m64 = a.mask_w[role] == TT_I64;
  int32_t* __restrict__ dst = out_ids + t * (long long)S_out;
  const long long src = text * (long long)S;
  int count = 0;
  bool weird = false;
  for (int base = 0; base < S; base += 64) {
    const int tok = base + lane;
    const bool in = tok < S;
    long long m = 1, id = 0;
    if (in) {
      if (mp) m = m64 ? static_cast<const long long*>(mp)[src + tok] : (long long)static_cast<const int32_t*>(mp)[src + tok];
      id = i64 ? static_cast<const long long*>(ip)[src + tok] : (long long)static_cast<const int32_t*>(ip)[src + tok];
    }
    const bool keep = in && m != 0 && !tt_is_pad_host(id);
    weird = weird || (keep && m != 1);
    const unsigned long long b = __ballot(keep);
    const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
    if (keep) dst[pos] = (id < 0 || id > 0x7fffffffLL) ? -1 : (int32_t)id;
    count += __popcll(b);
  }
  for (int tok = count + lane; tok < S_out; tok += 64) dst[tok] = (int32_t)TT_PAD_ID32;  // padding marked in the ids themselves
  if (lane == 0) out_lens[t] = count;
  if (__any(weird) && lane == 0) atomicOr(flags + 1, 1);
}

static int pack_roles(tt_engine* e, int slot, const tt_role_tokens* roles, int32_t n_roles, int32_t B, int* S_out_p,
                      hipStream_t st) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(roles, TT_ERR_NULL_POINTER);
  TT_REQUIRE(e->cfg.pack_roles, TT_ERR_WORKSPACE);  // the handle was created without the staging area
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS && (n_roles == 2 || n_roles == 3), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B, TT_ERR_BAD_SHAPE);
  PackRoles a{};
  a.n_roles = n_roles;
  int S_out = 0;
  for (int r = 0; r < n_roles; ++r) {
    TT_REQUIRE(roles[r].ids, TT_ERR_NULL_POINTER);
    TT_REQUIRE(roles[r].reserved == 0 && roles[r].S >= 1 && roles[r].S <= e->cfg.max_S, TT_ERR_BAD_SHAPE);
    TT_REQUIRE((roles[r].ids_dtype == TT_I32 || roles[r].ids_dtype == TT_I64) &&
                   (!roles[r].mask || roles[r].mask_dtype == TT_I32 || roles[r].mask_dtype == TT_I64), TT_ERR_BAD_DTYPE);
    a.ids[r] = roles[r].ids; a.mask[r] = roles[r].mask;
    a.ids_w[r] = roles[r].ids_dtype; a.mask_w[r] = roles[r].mask ? roles[r].mask_dtype : TT_I32;
    a.S[r] = roles[r].S;
    S_out = roles[r].S > S_out ? roles[r].S : S_out;
  }
  const long long texts = (long long)n_roles * B;
  hipLaunchKernelGGL(pack_roles_kernel, dim3((unsigned)((texts + 3) / 4)), dim3(256), 0, st, a, (int)B, S_out,
                     e->stage_ids[slot], e->stage_lens[slot], e->oob);
  TT_CHECK_LAUNCH();
  e->staged[slot][0] = n_roles; e->staged[slot][1] = B; e->staged[slot][2] = S_out;
  *S_out_p = S_out;
  return TT_OK;
}

extern "C" int tt_engine_pack_roles(tt_engine* e, int slot, const tt_role_tokens* roles, int32_t n_roles, int32_t B,
                                    int32_t* S_out, tt_stream_t stream) {
  TT_REQUIRE(S_out, TT_ERR_NULL_POINTER);
  int s = 0;
  const int rc = pack_roles(e, slot, roles, n_roles, B, &s, static_cast<hipStream_t>(stream));
  *S_out = s;
  return rc;
}

extern "C" int tt_engine_pool_packed(tt_engine* e, int slot, int32_t n_roles, int32_t B, int32_t S, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->cfg.pack_roles, TT_ERR_WORKSPACE);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS, TT_ERR_BAD_SHAPE);
  // the staging rows are laid out [n_roles, B, S_out] as the last pack wrote them: any other (n_roles, B, S) would read
  // rows of another layout, or of an older batch
  TT_REQUIRE(e->staged[slot][0] == n_roles && e->staged[slot][1] == B && e->staged[slot][2] == S, TT_ERR_BAD_SHAPE);
  // no mask and no lengths: the packed rows mark their own padding (TT_PAD_ID32)
  return engine_k1(e, slot, e->stage_ids[slot], TT_I32, nullptr, TT_I32, nullptr, n_roles, B, S, nullptr, 0, stream);
}

extern "C" int tt_engine_pool_roles(tt_engine* e, int slot, const tt_role_tokens* roles, int32_t n_roles, int32_t B,
                                    tt_stream_t stream) {
  int S_out = 0;
  const int rc = pack_roles(e, slot, roles, n_roles, B, &S_out, static_cast<hipStream_t>(stream));
  if (rc != TT_OK) return rc;
  return engine_k1(e, slot, e->stage_ids[slot], TT_I32, nullptr, TT_I32, nullptr, n_roles, B, S_out, nullptr, 0, stream);
}

extern "C" int tt_engine_pool_cached(tt_engine* e, int slot, const float* cache, int64_t cache_rows,
                                     const void* rows, int rows_dtype, int32_t roles, int32_t B,
                                     tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(cache && rows, TT_ERR_NULL_POINTER);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS && (roles == 2 || roles == 3), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B && cache_rows >= 1, TT_ERR_BAD_SHAPE);
  return tt_embed_gather_f32(cache, rows, rows_dtype, e->pooled[slot], roles * (int64_t)B, e->cfg.D, cache_rows,
                             e->oob, stream);
}

// The two towers' row ranges of a pooled slot: query rows [0,B), document rows [B, B + doc_rows).
static void fill_towers(tt_engine* e, int slot, int32_t B, int32_t doc_rows, ttp::Tower (&t)[2]) {
  const int D = e->cfg.D, P = e->cfg.P;
  for (int i = 0; i < 2; ++i) {
    const int64_t row0 = i == 0 ? 0 : B;
    float* w = e->params + i * e->tower_n;
    float* g = e->slabs + i * e->tower_n;
    t[i] = ttp::Tower{};
    t[i].M = i == 0 ? B : doc_rows;
    t[i].X = e->pooled[slot] + row0 * D;
    t[i].W1 = w;
    t[i].b1 = w + (int64_t)P * D;
    t[i].W2 = t[i].b1 + P;
    t[i].b2 = t[i].W2 + (int64_t)P * P;
    t[i].H = e->H + row0 * P;
    t[i].Out = e->emb + row0 * P;
    t[i].dOut = e->dOut + row0 * P;
    t[i].dH = e->dH + row0 * P;
    t[i].gW1 = g;
    t[i].gb1 = g + (int64_t)P * D;
    t[i].gW2 = t[i].gb1 + P;
    t[i].gb2 = t[i].gW2 + (int64_t)P * P;
  }
}

// Backward of phase B.  dW2/db2 and dH are independent; dW1/db1 needs dH.  Two forms, A/B-measured on
// one box with interleaved runs (tt_tuning.backward / tt_engine_set_backward override the choice):
//   fused: loss mean inside the triplet launch, ONE launch for dW2/db2 + dH, then dW1: 8 launches and no
//          event traffic per step.  The default: alone on the chip (frozen-feature cache: 125 vs 129 us),
//          and beside gathers reserved just above the projection tile's LDS, where it ties with the fork
//          at B = 2048 (199 vs 200 us) and wins below (B = 512: 100 vs 113 us) because it costs the host
//          a third less (52-69 vs 85-100 us of API time per step).
//   fork : dW2/db2 and the loss mean on the engine's second stream under the dH -> dW1 chain, joined
//          before the caller's next stream-ordered call.  Shortest critical path beside a CAPPED K1
//          (explicit k1_blocks_per_cu, per-text form): 232 vs 248 us per step; there the in-launch loss
//          mean also loses (its ticket round trip queues behind the gather's loads).
static bool use_fork(const tt_engine* e) {
  if (e->bwd_mode == TT_BWD_FORK) return e->aux != nullptr;
  if (e->bwd_mode == TT_BWD_FUSED) return false;
  return e->shared && !e->tile_above && e->aux != nullptr && e->gathers_share_chip;
}

static int backward_chain(tt_engine* e, const ttp::Tower (&t)[2], int32_t loss_rows, float loss_denominator,
                          hipStream_t st) {
  const int D = e->cfg.D, P = e->cfg.P;
  const ttp::Plan shared = chain_plan(e);
  int rc;
  if (!use_fork(e)) {
    e->join_wait = nullptr;
    if (e->ev_loss && e->loss_event_armed) TT_HIP(hipEventRecord(e->ev_loss, st));  // (an extra packet on the critical chain)
    if (loss_rows > 0) {
      rc = tt_launch_loss_mean(e->hinge, loss_rows, loss_denominator, e->loss, e->loss_sum, e->loss_cnt, st);
      if (rc != TT_OK) return rc;
    }
    rc = ttp::backward_dw2_dh(t, 2, D, P, e->nslabs, e->slab_stride, st, shared);
    if (rc != TT_OK) return rc;
    if (e->ev_dw2) TT_HIP(hipEventRecord(e->ev_dw2, st));
    return ttp::backward_dw1(t, 2, D, P, e->nslabs, e->slab_stride, st, shared);
  }
  // auto (0) = 1: measured 179.5 -> 176.3 us per configs[1] step, 150.3 -> 145.5 at configs[2]'s share (with the trainer's
  // fence events made the same way; profiles/r04_event_scope_ab.txt); 3 = HIP's default system-scope events
  const int es = e->tuning.event_scope == 3 ? 0 : (e->tuning.event_scope == 2 ? 2 : 1);
  // the fork point: ONE event on the caller's stream, the caller's if it gave one (tt_engine_set_loss_event) — a second
  // stream of the caller's may wait for the same record instead of paying for a packet of its own
  hipEvent_t fork = e->ev_loss ? e->ev_loss : e->ev_fork[es];
  TT_HIP(hipEventRecord(fork, st));
  TT_HIP(hipStreamWaitEvent(e->aux, fork, 0));
  if (loss_rows > 0) {
    rc = tt_launch_loss_mean(e->hinge, loss_rows, loss_denominator, e->loss, e->loss_sum, e->loss_cnt, e->aux);
    if (rc != TT_OK) return rc;
  }
  rc = ttp::backward_dw2(t, 2, D, P, e->nslabs, e->slab_stride, e->aux, shared);
  if (rc != TT_OK) return rc;
  if (e->ev_dw2) TT_HIP(hipEventRecord(e->ev_dw2, e->aux));
  if (e->join_wait) {  // the caller's next dependency rides on the join: one wait packet on the caller's stream, not two
    TT_HIP(hipStreamWaitEvent(e->aux, e->join_wait, 0));
    e->join_wait = nullptr;
  }
  TT_HIP(hipEventRecord(e->ev_join[es], e->aux));
  rc = ttp::backward_dh(t, 2, D, P, st, shared);
  if (rc != TT_OK) return rc;
  rc = ttp::backward_dw1(t, 2, D, P, e->nslabs, e->slab_stride, st, shared);
  if (rc != TT_OK) return rc;
  TT_HIP(hipStreamWaitEvent(st, e->ev_join[es], 0));
  return TT_OK;
}

extern "C" int tt_engine_fwd_bwd(tt_engine* e, int slot, int32_t B, const int32_t* neg_index, float margin,
                                 float grad_scale, float loss_denominator, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->params, TT_ERR_NULL_POINTER);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B && loss_denominator > 0.f, TT_ERR_BAD_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int D = e->cfg.D, P = e->cfg.P;
  ttp::Tower t[2];
  fill_towers(e, slot, B, neg_index ? B : 2 * B, t);
  int rc = ttp::forward(t, 2, D, P, st, chain_plan(e));
  if (rc != TT_OK) return rc;
  const float* q = e->emb;
  const float* p = e->emb + (int64_t)B * P;
  // with neg_index the negatives ARE rows of the positives' embeddings (no third encode)
  const float* n = neg_index ? p : e->emb + 2 * (int64_t)B * P;
  if (use_fork(e)) {  // loss mean as its own launch on the second stream
    rc = tt_launch_triplet_rows(q, p, n, margin, grad_scale, B, P, e->pos_dist, e->neg_dist, e->hinge, e->dOut,
                                e->dOut + (int64_t)B * P, e->dOut + 2 * (int64_t)B * P, neg_index, st, nullptr, nullptr, 1.f,
                                nullptr, nullptr, nullptr, e->neg_csr);
    if (rc != TT_OK) return rc;
    return backward_chain(e, t, B, loss_denominator, st);
  }
  // the batch loss is reduced inside the triplet launch (last workgroup), not by a launch of its own
  rc = tt_launch_triplet_rows(q, p, n, margin, grad_scale, B, P, e->pos_dist, e->neg_dist, e->hinge, e->dOut,
                              e->dOut + (int64_t)B * P, e->dOut + 2 * (int64_t)B * P, neg_index, st,
                              e->mean_partials, e->mean_ticket, loss_denominator, e->loss, e->loss_sum, e->loss_cnt, e->neg_csr);
  if (rc != TT_OK) return rc;
  return backward_chain(e, t, 0, 1.f, st);
}

extern "C" int tt_engine_forward(tt_engine* e, int slot, int32_t B, int32_t roles, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->params, TT_ERR_NULL_POINTER);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS && (roles == 2 || roles == 3), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B, TT_ERR_BAD_SHAPE);
  ttp::Tower t[2];
  fill_towers(e, slot, B, (roles - 1) * B, t);
  return ttp::forward(t, 2, e->cfg.D, e->cfg.P, static_cast<hipStream_t>(stream), chain_plan(e));
}

extern "C" int tt_engine_backward(tt_engine* e, int slot, int32_t B, int32_t roles, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->params, TT_ERR_NULL_POINTER);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS && (roles == 2 || roles == 3), TT_ERR_BAD_SHAPE);
  TT_REQUIRE(B >= 1 && B <= e->cfg.max_B, TT_ERR_BAD_SHAPE);
  ttp::Tower t[2];
  fill_towers(e, slot, B, (roles - 1) * B, t);
  return backward_chain(e, t, 0, 1.f, static_cast<hipStream_t>(stream));
}

extern "C" float* tt_engine_dout(const tt_engine* e) { return ok(e) ? e->dOut : nullptr; }

__global__ void add_loss_kernel(const float* __restrict__ v, float* __restrict__ last, double* __restrict__ sum,
                                long long* __restrict__ cnt) {
  *last = *v;
  *sum += static_cast<double>(*v);
  *cnt += 1;
}

extern "C" int tt_engine_add_loss(tt_engine* e, const float* loss_value, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(loss_value, TT_ERR_NULL_POINTER);
  hipLaunchKernelGGL(add_loss_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), loss_value, e->loss,
                     e->loss_sum, e->loss_cnt);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

extern "C" int tt_engine_reduce_grads(tt_engine* e, float* grad_out, int accumulate, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(grad_out, TT_ERR_NULL_POINTER);
  return tt_reduce_slabs_f32(e->slabs, e->nslabs, e->slab_stride, grad_out, e->nparam, accumulate, stream);
}

extern "C" int tt_engine_reduce_grads_range(tt_engine* e, float* grad_out, int64_t offset, int64_t count,
                                            int accumulate, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(grad_out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(offset >= 0 && count >= 0 && offset + count <= e->nparam, TT_ERR_BAD_SHAPE);
  return tt_reduce_slabs_f32(e->slabs + offset, e->nslabs, e->slab_stride, grad_out + offset, count, accumulate, stream);
}

extern "C" int tt_engine_set_dw2_event(tt_engine* e, tt_event_t event) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  e->ev_dw2 = static_cast<hipEvent_t>(event);
  return TT_OK;
}

extern "C" int tt_engine_set_loss_event(tt_engine* e, tt_event_t event) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  e->ev_loss = static_cast<hipEvent_t>(event);
  // a caller that hands over an event means to wait for it: recorded by BOTH backward schedules until the caller says
  // otherwise (tt_engine_arm_loss_event(0)) — a wait on an event the fused backward never recorded would be a no-op and the
  // gather could overwrite a slot still in use (round-5 advisor finding)
  e->loss_event_armed = event != nullptr;
  return TT_OK;
}

extern "C" int tt_engine_arm_loss_event(tt_engine* e, int32_t armed) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  e->loss_event_armed = armed != 0;
  return TT_OK;
}

extern "C" int tt_engine_set_join_wait(tt_engine* e, tt_event_t event, int32_t* accepted) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  const bool take = event != nullptr && use_fork(e);
  e->join_wait = take ? static_cast<hipEvent_t>(event) : nullptr;
  if (accepted) *accepted = take ? 1 : 0;
  return TT_OK;
}

extern "C" int tt_engine_k1_census(tt_engine* e, int slot, int32_t* admitted, int32_t* cus_used, int32_t* max_registered,
                                   int32_t* planned, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(slot >= 0 && slot < TT_ENGINE_SLOTS, TT_ERR_BAD_SHAPE);
  if (!e->k1_streamed[slot]) {  // the last gather into this slot did not use the streaming form
    if (admitted) *admitted = 0;
    if (cus_used) *cus_used = 0;
    if (max_registered) *max_registered = 0;
    if (planned) *planned = 0;
    return TT_OK;
  }
  return tt_gather_pool_stream_census(e->k1_stream_ctl[slot], admitted, cus_used, max_registered, planned, stream);
}

extern "C" int tt_engine_adam(tt_engine* e, const float* grads, double lr, double beta1, double beta2,
                              double eps, int64_t step, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->params, TT_ERR_NULL_POINTER);
  if (grads) return tt_adam_f32(e->params, grads, 1, 0, e->exp_avg, e->exp_avg_sq, e->nparam, lr, beta1, beta2, eps, step, 1.f, stream);
  return tt_adam_f32(e->params, e->slabs, e->nslabs, e->slab_stride, e->exp_avg, e->exp_avg_sq, e->nparam, lr,
                     beta1, beta2, eps, step, 1.f, stream);
}

extern "C" int tt_engine_adam_auto(tt_engine* e, const float* grads, double lr, double beta1, double beta2,
                                   double eps, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(e->params, TT_ERR_NULL_POINTER);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (grads) {
    TT_REQUIRE(tt_aligned16(grads), TT_ERR_ALIGNMENT);
    return tt_launch_adam_auto(e->params, grads, 1, 0, e->exp_avg, e->exp_avg_sq, e->nparam, lr, beta1, beta2, eps,
                               e->dev_step, e->dev_scal, st);
  }
  return tt_launch_adam_auto(e->params, e->slabs, e->nslabs, e->slab_stride, e->exp_avg, e->exp_avg_sq, e->nparam,
                             lr, beta1, beta2, eps, e->dev_step, e->dev_scal, st);
}

extern "C" int tt_engine_set_step(tt_engine* e, int64_t step, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  TT_REQUIRE(step >= 0, TT_ERR_BAD_SHAPE);
  return tt_launch_set_step(e->dev_step, step, static_cast<hipStream_t>(stream));
}

extern "C" int tt_engine_train_step(tt_engine* e, const void* ids, int ids_dtype, const void* mask,
                                    int mask_dtype, const int32_t* lens, int32_t B, int32_t S, float margin,
                                    double lr, int64_t step, tt_stream_t stream) {
  int rc = tt_engine_pool(e, 0, ids, ids_dtype, mask, mask_dtype, lens, 3, B, S, stream);
  if (rc != TT_OK) return rc;
  rc = tt_engine_fwd_bwd(e, 0, B, nullptr, margin, 1.0f / (float)B, (float)B, stream);
  if (rc != TT_OK) return rc;
  return tt_engine_adam(e, nullptr, lr, 0.9, 0.999, 1e-8, step, stream);
}

extern "C" int tt_engine_read_stats(tt_engine* e, double* loss_sum, int64_t* n_batches, float* last_loss,
                                    int reset, tt_stream_t stream) {
  TT_REQUIRE(ok(e), TT_ERR_BAD_HANDLE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  double s = 0;
  long long c = 0;
  float l = 0;
  int32_t oob[2] = {0, 0};  // [0] a token id / store row outside its range was clamped, [1] tt_engine_pool_roles met a weight mask
  TT_HIP(hipMemcpyAsync(&s, e->loss_sum, sizeof(double), hipMemcpyDeviceToHost, st));
  TT_HIP(hipMemcpyAsync(&c, e->loss_cnt, sizeof(long long), hipMemcpyDeviceToHost, st));
  TT_HIP(hipMemcpyAsync(&l, e->loss, sizeof(float), hipMemcpyDeviceToHost, st));
  TT_HIP(hipMemcpyAsync(oob, e->oob, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (reset) {
    TT_HIP(hipMemsetAsync(e->loss_sum, 0, sizeof(double), st));
    TT_HIP(hipMemsetAsync(e->loss_cnt, 0, sizeof(long long), st));
    TT_HIP(hipMemsetAsync(e->oob, 0, 2 * sizeof(int32_t), st));  // the flags are reported once
    // the in-launch loss mean's arrival ticket: zero between launches by construction (the last workgroup
    // re-zeroes it); cleared here as well so that a launch torn down mid-grid cannot leave it poisoned
    TT_HIP(hipMemsetAsync(e->mean_ticket, 0, sizeof(unsigned), st));
  }
  TT_HIP(hipStreamSynchronize(st));
  if (loss_sum) *loss_sum = s;
  if (n_batches) *n_batches = c;
  if (last_loss) *last_loss = l;
  if (oob[1]) return TT_ERR_UNSUPPORTED;  // a mask value other than 0 / 1 reached tt_engine_pool_roles
  return oob[0] ? TT_ERR_BAD_SHAPE : TT_OK;  // an out-of-range token id was clamped
}

// Events for ordering streams of ONE device against each other (the trainer's gather fence / slot-ready events).  A
// default hipEvent performs a system-scope release when it is recorded — a cache write-back the host and other
// devices would need, and that the kernel behind it on the same stream waits for; streams of the same device need no
// more than device scope.  scope: 0 default, 1 hipEventDisableSystemFence, 2 hipEventReleaseToDevice.
extern "C" int tt_event_create(int scope, tt_event_t* out) {
  TT_REQUIRE(out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(scope >= 0 && scope <= 2, TT_ERR_BAD_SHAPE);
  hipEvent_t ev = nullptr;
  TT_HIP(hipEventCreateWithFlags(&ev, tt_event_flags(scope)));
  *out = ev;
  return TT_OK;
}
extern "C" int tt_event_destroy(tt_event_t ev) {
  TT_REQUIRE(ev, TT_ERR_NULL_POINTER);
  TT_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev)));
  return TT_OK;
}
extern "C" int tt_event_record(tt_event_t ev, tt_stream_t stream) {
  TT_REQUIRE(ev, TT_ERR_NULL_POINTER);
  TT_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)));
  return TT_OK;
}
extern "C" int tt_stream_wait_event(tt_stream_t stream, tt_event_t ev) {
  TT_REQUIRE(ev, TT_ERR_NULL_POINTER);
  TT_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0));
  return TT_OK;
}
// 1: everything recorded before the event has completed; 0: not yet; < 0 / other: error
extern "C" int tt_event_done(tt_event_t ev) {
  TT_REQUIRE(ev, TT_ERR_NULL_POINTER);
  const hipError_t r = hipEventQuery(static_cast<hipEvent_t>(ev));
  if (r == hipSuccess) return 1;
  if (r == hipErrorNotReady) return 0;
  return TT_ERR_BAD_HANDLE;
}
