// Where does a small projection GEMM's time go when it has the chip to itself?  (diagnostic, round 5)
// Builds the product's gemm_f32.hip with TT_GEMM_STAMPS: thread 0 of every workgroup records the 100 MHz wall clock at
// entry / after the prologue (first operand tile in LDS) / after the K loop / after the epilogue.  Prints, per shape of
// the chain at BASELINE.json configs[2]'s one-GPU share (3072 rows, P = 512, D = 384) and configs[1] (6144 rows, 384):
// the launch's span (first entry -> last exit), the spread of workgroup start times, and the median / max length of each
// phase, next to the MFMA time the tile's K-steps need (1024 cycles per K-step of 32 on a 64x64 tile at 2.4 GHz).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTT_GEMM_STAMPS -DTT_GEMM_KSPLIT -Iinclude -Itwo-towers-overlords_amd/csrc -Itools tools/gemm_stamps.hip -o /tmp/gemm_stamps
#include "../two-towers-overlords_amd/csrc/gemm_f32.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static float* dev_rand(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  float* d; CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

static void report(const char* name, int wgs, int ksteps_long) {
  std::vector<unsigned long long> st(4 * 8192);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tt_gemm_stamps), st.size() * sizeof(unsigned long long)));
  std::vector<unsigned long long> cy(4 * 8192);
  CK(hipMemcpyFromSymbol(cy.data(), HIP_SYMBOL(tt_gemm_cycles), cy.size() * sizeof(unsigned long long)));
  std::vector<double> start, pro, kl, epi, tot, ghz;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int w = 0; w < wgs && w < 8192; ++w) {
    const unsigned long long* p = &st[4 * w];
    if (p[0] == 0 || p[3] < p[0]) continue;  // an empty workgroup of a grouped launch
    t0 = std::min(t0, p[0]); t1 = std::max(t1, p[3]);
  }
  for (int w = 0; w < wgs && w < 8192; ++w) {
    const unsigned long long* p = &st[4 * w];
    if (p[0] == 0 || p[3] < p[0]) continue;
    start.push_back((p[0] - t0) * 0.01); pro.push_back((p[1] - p[0]) * 0.01); kl.push_back((p[2] - p[1]) * 0.01);
    epi.push_back((p[3] - p[2]) * 0.01); tot.push_back((p[3] - p[0]) * 0.01);
    if (p[2] > p[1]) ghz.push_back((double)(cy[4 * w + 2] - cy[4 * w + 1]) / ((p[2] - p[1]) * 10.0));  // shader cycles per ns over the K loop
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  auto mx = [](const std::vector<double>& v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
  auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() * 9 / 10]; };
  printf("%-34s wgs %4zu  span %6.2f us | start med %5.2f p90 %5.2f max %5.2f | prologue med %5.2f max %5.2f | K loop med %5.2f p90 %5.2f max %5.2f"
         " (MFMA of %2d K-steps: %5.2f) | epilogue med %5.2f max %5.2f | workgroup med %5.2f max %5.2f | clock64/wall over the K loop: %.3f GHz\n",
         name, tot.size(), (t1 - t0) * 0.01, med(start), p90(start), mx(start), med(pro), mx(pro), med(kl), p90(kl), mx(kl), ksteps_long,
         ksteps_long * 1024 / 2400.0, med(epi), mx(epi), med(tot), mx(tot), med(ghz));
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Shape { const char* name; int rows_q, P, D; };
  const Shape shapes[] = {{"configs[2] share (B=1024,P=512,D=384)", 1024, 512, 384}, {"configs[1] (B=2048,P=384,D=384)", 2048, 384, 384}};
  const int bk = 32;
  for (const Shape& sh : shapes) {
    const int Mq = sh.rows_q, Md = 2 * sh.rows_q, P = sh.P, D = sh.D, rows = Mq + Md;
    float* X = dev_rand((size_t)rows * D, 1); float* W1 = dev_rand((size_t)2 * P * D, 2); float* W2 = dev_rand((size_t)2 * P * P, 3);
    float* b = dev_rand(2 * P, 4); float* H = dev_rand((size_t)rows * P, 5); float* O = dev_rand((size_t)rows * P, 6);
    float* dH = dev_rand((size_t)rows * P, 7);
    const int nsplit = sh.rows_q / 256; const long long stride = ((long long)2 * (P * D + P + P * P + P) + 3) / 4 * 4;
    float* slabs; CK(hipMalloc(&slabs, (size_t)nsplit * stride * sizeof(float)));
    printf("== %s: %d rows, two towers (%d + %d), 64x64 tiles, K-step %d, chip idle otherwise\n", sh.name, rows, Mq, Md, bk);
    auto timed = [&](const char* name, int wgs, int ks, auto&& fn) {
      { static std::vector<unsigned long long> z(4 * 8192, 0ull); CK(hipMemcpyToSymbol(HIP_SYMBOL(tt_gemm_stamps), z.data(), z.size() * sizeof(unsigned long long))); }
      for (int i = 0; i < 3; ++i) fn();  // warm
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 10; ++i) fn();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      char buf[96]; snprintf(buf, sizeof buf, "%s [%.1f us/launch]", name, ms * 100.0);
      report(buf, wgs, ks);
    };
    ttg::Batch b1{}; b1.count = 2; b1.nsplit = 1;
    for (int i = 0; i < 2; ++i) {
      ttg::Problem& p = b1.p[i];
      p.A = X + (i ? (size_t)Mq * D : 0); p.B = W1 + (size_t)i * P * D; p.C = H + (i ? (size_t)Mq * P : 0); p.bias = b + i * P;
      p.M = i ? Md : Mq; p.N = P; p.K = D; p.lda = D; p.ldb = D; p.ldc = P;
    }
    const int nt_wgs = (P / 64) * (Md / 64) * 2;
    timed("NT layer 1 (K = D)", nt_wgs, D / 32, [&] { ttg::launch(ttg::kNT, ttg::kBiasRelu, b1, 1, 1, st, bk); });
    const int rn = P % 96 == 0 ? 3 : 2;  // K-split 96-row tiles (round 5): one tile per CU at these shapes
    const int ks_wgs = (rows / 96) * (P / (32 * rn));
    timed("NT layer 1, K-split 96-row tiles", ks_wgs, D / 32 * 3 * rn / 4, [&] { ttg::launch(ttg::kNT, ttg::kBiasRelu, b1, 3, rn, st, bk); });
    {  // the same two launches over all-zero operands: if the K loop is shorter, what limits it is power (operand toggling), not issue
      float* Xz; float* Wz; CK(hipMalloc(&Xz, (size_t)rows * D * sizeof(float))); CK(hipMalloc(&Wz, (size_t)2 * P * D * sizeof(float)));
      CK(hipMemset(Xz, 0, (size_t)rows * D * sizeof(float))); CK(hipMemset(Wz, 0, (size_t)2 * P * D * sizeof(float)));
      ttg::Batch bz = b1;
      for (int i = 0; i < 2; ++i) { bz.p[i].A = Xz + (i ? (size_t)Mq * D : 0); bz.p[i].B = Wz + (size_t)i * P * D; }
      timed("  zeros: NT layer 1 (K = D)", nt_wgs, D / 32, [&] { ttg::launch(ttg::kNT, ttg::kBiasRelu, bz, 1, 1, st, bk); });
      timed("  zeros: NT layer 1, K-split tiles", ks_wgs, D / 32 * 3 * rn / 4, [&] { ttg::launch(ttg::kNT, ttg::kBiasRelu, bz, 3, rn, st, bk); });
      CK(hipFree(Xz)); CK(hipFree(Wz));
    }
    ttg::Batch b2 = b1;
    for (int i = 0; i < 2; ++i) { ttg::Problem& p = b2.p[i]; p.A = H + (i ? (size_t)Mq * P : 0); p.B = W2 + (size_t)i * P * P; p.C = O + (i ? (size_t)Mq * P : 0); p.K = P; p.lda = P; p.ldb = P; }
    timed("NT layer 2 (K = P)", nt_wgs, P / 32, [&] { ttg::launch(ttg::kNT, ttg::kBias, b2, 1, 1, st, bk); });
    timed("NT layer 2, K-split 96-row tiles", ks_wgs, P / 32 * 3 * rn / 4, [&] { ttg::launch(ttg::kNT, ttg::kBias, b2, 3, rn, st, bk); });
    {  // Probe (round 5): what a launch of the same total work would cost if a third of its tile-units were HALF units —
       // as many full tiles as CUs would hold one each, the rest as tiles with half the K loop (a stand-in for K-split half
       // tiles: same MFMA time, same workgroup shape), full tiles first in the grid.  Wrong results by construction.
      const int units = (P / 64) * (rows / 64);             // 64x64 tile-units of the real launch
      const int full_tiles = units >= 512 ? 384 : 256, half_tiles = 2 * (units - full_tiles);
      ttg::Batch pb{}; pb.count = 2; pb.nsplit = 1;
      for (int i = 0; i < 2; ++i) {
        ttg::Problem& p = pb.p[i];
        p.A = X; p.B = W1; p.C = H; p.bias = b;
        p.M = (i == 0 ? full_tiles : half_tiles) / (P / 64) * 64; p.N = P; p.K = i == 0 ? D : D / 2; p.lda = D; p.ldb = D; p.ldc = P;
      }
      if (pb.p[0].M <= rows && pb.p[1].M <= rows && pb.p[1].M > 0)
        timed("  probe: NT1 as full + half-K tiles", full_tiles + half_tiles, D / 32, [&] { ttg::launch(ttg::kNT, ttg::kBiasRelu, pb, 1, 1, st, bk); });
    }
    ttg::Batch gh = b2;
    for (int i = 0; i < 2; ++i) { ttg::Problem& p = gh.p[i]; p.A = O + (i ? (size_t)Mq * P : 0); p.C = dH + (i ? (size_t)Mq * P : 0); p.aux = H + (i ? (size_t)Mq * P : 0); p.ldaux = P; p.bias = nullptr; }
    timed("NN dH (K = P)", nt_wgs, P / 32, [&] { ttg::launch(ttg::kNN, ttg::kReluMask, gh, 1, 1, st, bk); });
    timed("NN dH, K-split 96-row tiles", ks_wgs, P / 32 * 3 * rn / 4, [&] { ttg::launch(ttg::kNN, ttg::kReluMask, gh, 3, rn, st, bk); });
    ttg::Batch g1{}; g1.count = 2; g1.nsplit = nsplit; g1.slab_stride = stride;
    for (int i = 0; i < 2; ++i) {  // document tower (longer K) first, as proj.hip orders them
      const int t = 1 - i; ttg::Problem& p = g1.p[i];
      p.A = dH + (t ? (size_t)Mq * P : 0); p.B = X + (t ? (size_t)Mq * D : 0); p.C = slabs + (size_t)t * (P * D + P + P * P + P); p.colsum = p.C + (size_t)P * D;
      p.M = P; p.N = D; p.K = t ? Md : Mq; p.lda = P; p.ldb = D; p.ldc = D; p.k_per_split = p.K / nsplit;
    }
    timed("TN dW1 split-K", (D / 64) * (P / 64) * 2 * nsplit, Md / nsplit / 32, [&] { ttg::launch(ttg::kTN, ttg::kStore, g1, 1, 1, st, bk); });
    CK(hipFree(X)); CK(hipFree(W1)); CK(hipFree(W2)); CK(hipFree(b)); CK(hipFree(H)); CK(hipFree(O)); CK(hipFree(dH)); CK(hipFree(slabs));
  }
  return 0;
}
