// K2 — per-tower projection Linear(D,P) -> ReLU -> Linear(P,P) and its parameter
// gradients, as grouped exact-fp32 MFMA GEMMs (both towers in one launch).
// Replaces backend/model.py:33-38,59 and the part of loss.backward()
// (backend/training.py:50) that runs through it.
#include "proj.h"

namespace ttp {

int k_per_split(int M, int nsplit) {
  int k = (M + nsplit - 1) / nsplit;
  k = ((k + 31) / 32) * 32;
  return k < 32 ? 32 : k;
}

int choose_nsplit(long long M) {
  long long s = (M + 255) / 256;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return (int)s;
}

// Split count for the engine's weight-gradient launches (one launch holds both towers' tiles; the document tower's
// are twice as long).  About 256 query rows per split as in rounds 1-3, but never fewer than four splits where the
// batch allows them: measured in round 4 (profiles/r04_grad_slabs_ab.txt) B = 512 runs 94.1 / 85.4 / 82.1 us per step
// with 2 / 3 / 4 splits (the old rule gave 2).  A longest-first packing model of the tiles onto the 256 compute units
// — which prefers 7 splits to 8 at BASELINE.json configs[1]: 504 tiles, two per CU, instead of 576 = 2.25 per CU paid
// as three rounds — was built and is falsified by the same measurement: 8 splits 173.5 us, 7 176.9, 6 177.2, 4 186.6,
// 10 172.9; beside the gather the weight-gradient tiles are latency-bound, not round-bound, and shorter tiles win.
int choose_nsplit_towers(long long B, int P, int D) {
  (void)P;
  (void)D;
  if (B < 1) return 1;
  long long s = (B + 255) / 256;
  const long long floor4 = (B + 31) / 32 < 4 ? (B + 31) / 32 : 4;  // splits shorter than one K-step stay empty
  if (s < floor4) s = floor4;
  if (s > 16) s = 16;
  return (int)s;
}

// XCD-aware workgroup order for the projection GEMMs (gemm_f32.h): on beside the streaming gather, where every
// re-read of an operand block that misses the XCD's L2 competes with the gather for the fabric — and beside the
// sliced gather from 6144 rows per launch (B >= 2048 triplets) up, where an XCD's L2 then sees one eighth of the
// activations beside the table slice it keeps for the gather: measured (profiles/r04_xcd_order_ab.txt) B = 2048
// 172.6 -> 170.1 us, B = 4096 338.7 -> 334.9, P = 512 B = 2048 221.7 -> 220.2, by-index 145.2 -> 143.9; below that the
// eight classes' rounding costs more than it saves (B = 1024 117.8 -> 119.7, B = 512 93.9 -> 96.7): off there, and off
// when nothing gathers beside the chain (frozen-feature cache: 123.5 vs 124.2).
static long long rows_of(const Tower* t, int count) {
  long long rows = 0;
  for (int i = 0; i < count; ++i) rows += t[i].M;
  return rows;
}
static int xcd_order_for(const Plan& pl, long long rows) {
  if (pl.t.xcd_order != 0) return pl.t.xcd_order > 0;
  return pl.policy == kTileStream || (pl.policy == kTileSmall && pl.beside_gather && rows >= 6144);
}

// Tile of the split-K weight-gradient GEMMs: 64x64 (many short workgroups; pairs with the gather's LDS reservation at
// BERT sizes), or — beside the streaming gather — 128x128 with K-step 16: a quarter of the operand re-reads.
static void pick_tn_tile(int M, int N, const Plan& pl, int& wm, int& wn, int& bk) {
  wm = wn = 1; bk = 32;
  const bool big = pl.t.tn_tile ? pl.t.tn_tile == 22 : pl.policy == kTileStream;
  if (big && M >= 256 && N >= 256) { wm = wn = 2; bk = 16; }
}

// Block tile for a grouped [rows x N] product.  (96x96 tiles on v_mfma_f32_16x16x4_f32 — 256 tiles for a single
// [6144 x 384] matrix where 64x64 gives 576 = 2.25 per CU — were built in round 3, parity-green, and measured slower at
// BASELINE.json configs[1]: the two towers' 2048 + 4096 rows make 260 such tiles, cache-fed 132 vs 129 us, one stream
// 253 vs 235 us, beside the gathers 233 vs 194 us; profiles/r03_gemm16_ab.txt.  Removed in round 4.)
// (Round 5: K-split 96-row tiles, proj_tile = 33 / 32 — ONE tile per CU at BASELINE.json configs[1] and configs[2]'s share.
// Alone on the chip a launch was 19.2 / 23.4 / 24.6 us against 19.8 / 24.1 / 25.3 at configs[2]'s share and 27.5 / 27.1 / 28.6
// against 27.4 / 27.2 / 27.9 at configs[1] (profiles/r05_gemm_stamps_ksplit.txt), no step got faster and no policy picked
// them: out of the library in round 6 — tools/gemm_ksplit_tiles.inc, built only by tools/gemm_stamps.hip.)
static void pick_tile(long long total_rows, int N, int& wm, int& wn, int& bk, const Plan& pl) {
  const bool low_traffic = pl.policy == kTileLowTraffic;
  bk = 32;
#ifdef TT_GEMM_KSPLIT
  if (pl.t.proj_tile == 33 || pl.t.proj_tile == 32) {  // K-split 96-row tiles, 96 or 64 columns (tools/gemm_ksplit_tiles.inc)
    wm = 3; wn = pl.t.proj_tile % 10;
    return;
  }
#endif
  if (pl.t.proj_tile == 11 || pl.t.proj_tile == 21 || pl.t.proj_tile == 22) {
    wm = pl.t.proj_tile / 10; wn = pl.t.proj_tile % 10;
    if (wm == 2 && wn == 2 && pl.t.proj_bk == 16) bk = 16;
    return;
  }
  const long long tiles64 = ((total_rows + 63) / 64) * ((N + 63) / 64);
  if (tiles64 >= 2048 && N >= 128) {
    wm = 2; wn = 2;
    // beside the streaming gather: K-step 16 (40 KB of LDS, <= 168 registers), two tiles + the gather's waves per CU
    bk = pl.policy == kTileStream ? 16 : 32;
    if (pl.t.proj_bk == 16 || pl.t.proj_bk == 32) bk = pl.t.proj_bk;
  }
  else if (pl.policy == kTileSmall) { wm = 1; wn = 1; }
  else if (tiles64 >= 1024 || (low_traffic && tiles64 >= 512)) { wm = 2; wn = 1; }
  else { wm = 1; wn = 1; }
}

size_t chain_lds_bytes(long long rows, int P, const Plan& pl) {
  int wm, wn, bk;
  pick_tile(rows, P, wm, wn, bk, pl);
  size_t m = ttg::tile_lds_bytes(ttg::kNT, wm, wn, bk);
  const size_t nn = ttg::tile_lds_bytes(ttg::kNN, wm, wn, bk);
  const size_t tn = ttg::tile_lds_bytes(ttg::kTN, 1, 1);
  m = nn > m ? nn : m;
  return tn > m ? tn : m;
}

int forward(const Tower* t, int count, int D, int P, hipStream_t st, const Plan& pl) {
  ttg::Batch b1{}, b2{};
  b1.count = b2.count = count;
  b1.nsplit = b2.nsplit = 1;
  b1.xcd_order = b2.xcd_order = xcd_order_for(pl, rows_of(t, count));
  long long rows = 0;
  for (int i = 0; i < count; ++i) {
    rows += t[i].M;
    ttg::Problem& p = b1.p[i];
    p.A = t[i].X; p.B = t[i].W1; p.C = t[i].H; p.bias = t[i].b1;
    p.M = t[i].M; p.N = P; p.K = D; p.lda = D; p.ldb = D; p.ldc = P;
    ttg::Problem& q = b2.p[i];
    q.A = t[i].H; q.B = t[i].W2; q.C = t[i].Out; q.bias = t[i].b2;
    q.M = t[i].M; q.N = P; q.K = P; q.lda = P; q.ldb = P; q.ldc = P;
  }
  int wm, wn, bk;
  pick_tile(rows, P, wm, wn, bk, pl);
  int rc = ttg::launch(ttg::kNT, ttg::kBiasRelu, b1, wm, wn, st, bk);
  if (rc != TT_OK) return rc;
  return ttg::launch(ttg::kNT, ttg::kBias, b2, wm, wn, st, bk);
}

// Weight-gradient launches put the towers with more rows (longer K loops) first: workgroups are placed in
// index order, so the short ones then fill in behind the long ones instead of the other way round.
static int tn_slot(const Tower* t, int count, int i) {
  if (count != 2) return i;
  return (t[1].M > t[0].M) ? 1 - i : i;
}

int backward_dw2(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st, const Plan& pl) {
  ttg::Batch g2{};
  g2.count = count;
  g2.nsplit = nsplit;
  g2.xcd_order = xcd_order_for(pl, rows_of(t, count));
  g2.slab_stride = slab_stride;
  for (int i = 0; i < count; ++i) {
    // dW2[i,j] = sum_m dOut[m,i] H[m,j];  db2[i] = sum_m dOut[m,i]
    ttg::Problem& a = g2.p[tn_slot(t, count, i)];
    a.A = t[i].dOut; a.B = t[i].H; a.C = t[i].gW2; a.colsum = t[i].gb2;
    a.M = P; a.N = P; a.K = t[i].M; a.lda = P; a.ldb = P; a.ldc = P; a.k_per_split = k_per_split(t[i].M, nsplit);
  }
  int wm, wn, bk;
  pick_tn_tile(P, P, pl, wm, wn, bk);
  return ttg::launch(ttg::kTN, ttg::kStore, g2, wm, wn, st, bk);
}

int backward_dh(const Tower* t, int count, int D, int P, hipStream_t st, const Plan& pl) {
  ttg::Batch gh{};
  gh.count = count;
  gh.nsplit = 1;
  gh.xcd_order = xcd_order_for(pl, rows_of(t, count));
  long long rows = 0;
  for (int i = 0; i < count; ++i) {
    rows += t[i].M;
    // dH = (dOut W2) * [H > 0]
    ttg::Problem& h = gh.p[i];
    h.A = t[i].dOut; h.B = t[i].W2; h.C = t[i].dH; h.aux = t[i].H;
    h.M = t[i].M; h.N = P; h.K = P; h.lda = P; h.ldb = P; h.ldc = P; h.ldaux = P;
  }
  int wm, wn, bk;
  pick_tile(rows, P, wm, wn, bk, pl);
  return ttg::launch(ttg::kNN, ttg::kReluMask, gh, wm, wn, st, bk);
}

int backward_dw2_dh(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st,
                    const Plan& pl) {
  ttg::Batch g2{}, gh{};
  g2.count = gh.count = count;
  g2.nsplit = nsplit;
  g2.slab_stride = slab_stride;
  gh.nsplit = 1;
  g2.xcd_order = gh.xcd_order = xcd_order_for(pl, rows_of(t, count));
  long long rows = 0;
  for (int i = 0; i < count; ++i) {
    rows += t[i].M;
    ttg::Problem& a = g2.p[tn_slot(t, count, i)];
    a.A = t[i].dOut; a.B = t[i].H; a.C = t[i].gW2; a.colsum = t[i].gb2;
    a.M = P; a.N = P; a.K = t[i].M; a.lda = P; a.ldb = P; a.ldc = P; a.k_per_split = k_per_split(t[i].M, nsplit);
    ttg::Problem& h = gh.p[i];
    h.A = t[i].dOut; h.B = t[i].W2; h.C = t[i].dH; h.aux = t[i].H;
    h.M = t[i].M; h.N = P; h.K = P; h.lda = P; h.ldb = P; h.ldc = P; h.ldaux = P;
  }
  int wm, wn, bk;
  pick_tile(rows, P, wm, wn, bk, pl);
  int tm, tn, tbk;
  pick_tn_tile(P, P, pl, tm, tn, tbk);
  if ((wm == 2 && wn == 2) || wm == 3 || tm != 1) {  // the fused kernel carries 64x64 TN tiles beside 64x64 / 128x64 NN tiles only
    int rc = ttg::launch(ttg::kTN, ttg::kStore, g2, tm, tn, st, tbk);
    if (rc != TT_OK) return rc;
    return ttg::launch(ttg::kNN, ttg::kReluMask, gh, wm, wn, st, bk);
  }
  return ttg::launch_tn_nn_pair(g2, gh, wm, wn, st);
}

int backward_dw1(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st, const Plan& pl) {
  ttg::Batch g1{};
  g1.count = count;
  g1.nsplit = nsplit;
  g1.xcd_order = xcd_order_for(pl, rows_of(t, count));
  g1.slab_stride = slab_stride;
  for (int i = 0; i < count; ++i) {
    // dW1[i,j] = sum_m dH[m,i] X[m,j];  db1[i] = sum_m dH[m,i]
    ttg::Problem& c = g1.p[tn_slot(t, count, i)];
    c.A = t[i].dH; c.B = t[i].X; c.C = t[i].gW1; c.colsum = t[i].gb1;
    c.M = P; c.N = D; c.K = t[i].M; c.lda = P; c.ldb = D; c.ldc = D; c.k_per_split = k_per_split(t[i].M, nsplit);
  }
  int wm, wn, bk;
  pick_tn_tile(P, D, pl, wm, wn, bk);
  return ttg::launch(ttg::kTN, ttg::kStore, g1, wm, wn, st, bk);
}

int backward(const Tower* t, int count, int D, int P, int nsplit, long long slab_stride, hipStream_t st, const Plan& pl) {
  int rc = backward_dw2(t, count, D, P, nsplit, slab_stride, st, pl);
  if (rc != TT_OK) return rc;
  rc = backward_dh(t, count, D, P, st, pl);
  if (rc != TT_OK) return rc;
  return backward_dw1(t, count, D, P, nsplit, slab_stride, st, pl);
}

}  // namespace ttp

extern "C" int tt_proj_fwd_f32(const float* X, const float* W1, const float* b1, const float* W2,
                               const float* b2, float* H, float* Out, int64_t M, int D, int P,
                               const tt_tuning* tuning, tt_stream_t stream) {
  TT_REQUIRE(X && W1 && b1 && W2 && b2 && H && Out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(M >= 0 && M <= 0x7fffffffLL && D >= 1 && P >= 1, TT_ERR_BAD_SHAPE);
  if (M == 0) return TT_OK;
  ttp::Tower t{};
  t.X = X; t.W1 = W1; t.b1 = b1; t.W2 = W2; t.b2 = b2; t.H = H; t.Out = Out; t.M = (int)M;
  return ttp::forward(&t, 1, D, P, static_cast<hipStream_t>(stream), ttp::plan_of(tuning));
}

static int64_t tower_floats(int D, int P) { return (int64_t)P * D + P + (int64_t)P * P + P; }
static int64_t round4(int64_t x) { return (x + 3) / 4 * 4; }

extern "C" size_t tt_proj_bwd_workspace_bytes(int64_t M, int D, int P) {
  const int ns = ttp::choose_nsplit(M);
  return (size_t)(round4(M * P) + (int64_t)ns * round4(tower_floats(D, P))) * sizeof(float);
}

extern "C" int tt_proj_bwd_f32(const float* dOut, const float* X, const float* H, const float* W2,
                               float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                               size_t workspace_bytes, int accumulate, int64_t M, int D, int P,
                               const tt_tuning* tuning, tt_stream_t stream) {
  TT_REQUIRE(dOut && X && H && W2 && dW1 && db1 && dW2 && db2 && workspace, TT_ERR_NULL_POINTER);
  TT_REQUIRE(M >= 1 && M <= 0x7fffffffLL && D >= 1 && P >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(workspace_bytes >= tt_proj_bwd_workspace_bytes(M, D, P), TT_ERR_WORKSPACE);
  TT_REQUIRE(tt_aligned16(workspace), TT_ERR_ALIGNMENT);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ns = ttp::choose_nsplit(M);
  const int64_t stride = round4(tower_floats(D, P));
  float* ws = static_cast<float*>(workspace);
  float* dH = ws;
  float* slabs = ws + round4(M * P);
  ttp::Tower t{};
  t.X = X; t.W2 = W2; t.H = const_cast<float*>(H); t.dOut = dOut; t.dH = dH; t.M = (int)M;
  t.gW1 = slabs;
  t.gb1 = slabs + (int64_t)P * D;
  t.gW2 = t.gb1 + P;
  t.gb2 = t.gW2 + (int64_t)P * P;
  int rc = ttp::backward(&t, 1, D, P, ns, stride, st, ttp::plan_of(tuning));
  if (rc != TT_OK) return rc;
  // outputs laid out like a tower's slab (dW1 | db1 | dW2 | db2 contiguous): one reduction launch instead of four
  if (db1 == dW1 + (int64_t)P * D && dW2 == db1 + P && db2 == dW2 + (int64_t)P * P)
    return tt_reduce_slabs_f32(t.gW1, ns, stride, dW1, tower_floats(D, P), accumulate, stream);
  rc = tt_reduce_slabs_f32(t.gW1, ns, stride, dW1, (int64_t)P * D, accumulate, stream);
  if (rc != TT_OK) return rc;
  rc = tt_reduce_slabs_f32(t.gb1, ns, stride, db1, P, accumulate, stream);
  if (rc != TT_OK) return rc;
  rc = tt_reduce_slabs_f32(t.gW2, ns, stride, dW2, (int64_t)P * P, accumulate, stream);
  if (rc != TT_OK) return rc;
  return tt_reduce_slabs_f32(t.gb2, ns, stride, db2, P, accumulate, stream);
}
