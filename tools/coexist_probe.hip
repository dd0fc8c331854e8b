// What do an MFMA-bound kernel and an L2-bound gather cost each other when they share the chip?  (diagnostic, not part of
// the product; docs/LOG_r04.md section 8 "Round 4"; round 5: docs/LOG_r05.md section 3)
//
//   kernel M: every wave runs a chain of v_mfma_f32_32x32x2_f32 on registers only (4 independent accumulators) — no
//             memory traffic at all, `mw` waves per SIMD (one 4-wave workgroup per wave-per-SIMD), optionally holding
//             `lds_kb` of LDS so that fewer fit a CU;
//   kernel G: every wave gathers random 128-byte lines from a 2 MB region per XCD (L2-resident after the first touch),
//             8 lines in flight per 8-lane group (64 per wave), like the product's sliced gathers; `gw` 4-wave
//             workgroups per CU.
// Each alone, then both on two streams; prints the times and the sum of the two speeds (alone time / together time).
//   hipcc -O3 --offload-arch=gfx950 tools/coexist_probe.hip -o /tmp/coexist_probe && /tmp/coexist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// GAP > 0: s_sleep GAP (64 x GAP cycles) after every four MFMAs (4 x 64 cycles of matrix-core time): the matrix core is busy
// 4 / (4 + GAP) of the time
template <int GAP>
__global__ __launch_bounds__(256) void mfma_kernel(float* __restrict__ out, int iters) {
  extern __shared__ float pad[];  // occupancy control only
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    if (GAP > 0) __builtin_amdgcn_s_sleep(GAP);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s + pad[0];  // never true: keeps the chain alive
}

// Round 5: the same register-only chain on other matrix opcodes — is "a busy matrix core stops the SIMD's other waves" a
// property of the chip or of the f32-input forms (whose rate IS the vector rate)?  MT 1: v_mfma_f32_16x16x4_f32,
// MT 2: v_mfma_f32_32x32x16_bf16 (a true matrix-core opcode: 16 x the f32 form's FLOP per clock).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MT>
__global__ __launch_bounds__(256) void mfma_kernel_t(float* __restrict__ out, int iters) {
  float s = 0.f;
  if constexpr (MT == 1) {
    f32x4v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][e];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = (__bf16)(1.0f + (threadIdx.x + e) * 1e-3f);
      b[e] = (__bf16)(1.0f - (threadIdx.x + e) * 1e-3f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][e];
  }
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;  // never true: keeps the chain alive
}

// region of XCD x (workgroups with blockIdx % 8 == x): lines [x * lines_per_xcd, (x + 1) * lines_per_xcd)
template <int PRIO>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ buf, unsigned lines_per_xcd, int iters,
                                                     float4* __restrict__ out) {
  if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
  const unsigned lane = threadIdx.x & 63, piece = lane & 7, grp = (blockIdx.x * 256 + threadIdx.x) >> 3;
  const float4* __restrict__ base = buf + (size_t)(blockIdx.x & 7) * lines_per_xcd * 8 + piece;
  unsigned h = grp * 2654435761u + 12345u;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    float4 r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      const unsigned line = (h >> 8) & (lines_per_xcd - 1);
      r[u] = base[(size_t)line * 8];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w; }
  }
  if (acc.x == 12345.678f) out[grp] = acc;
}

// The same two kinds of work inside ONE wave: every iteration requests 8 lines per lane group, then issues MPI x 4 MFMAs,
// then adds the lines — does the gather run in the matrix core's shadow when both belong to the same instruction stream?
template <int MPI>
__global__ __launch_bounds__(256) void fused_kernel(const float4* __restrict__ buf, unsigned lines_per_xcd, int iters,
                                                    float4* __restrict__ out) {
  const unsigned lane = threadIdx.x & 63, piece = lane & 7, grp = (blockIdx.x * 256 + threadIdx.x) >> 3;
  const float4* __restrict__ base = buf + (size_t)(blockIdx.x & 7) * lines_per_xcd * 8 + piece;
  unsigned h = grp * 2654435761u + 12345u;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 macc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) macc[i][e] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  // software-pipelined by hand: the lines of iteration it + 1 are requested, then this iteration's MFMAs are issued, then
  // the lines requested one iteration ago are added — the requests are in flight for the whole MFMA block
  float4 r[8], q[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u;
    r[u] = base[(size_t)((h >> 8) & (lines_per_xcd - 1)) * 8];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      q[u] = base[(size_t)((h >> 8) & (lines_per_xcd - 1)) * 8];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MPI; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) macc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, macc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w; }
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = q[u];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += macc[i][e];
  if (acc.x + s == 12345.678f) out[grp] = acc;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const unsigned lines_per_xcd = 2u * 1024 * 1024 / 128;  // 2 MB per XCD (a power of two: the line index is a mask, not a modulo)
  float4* buf;
  CHECK(hipMalloc(&buf, (size_t)8 * lines_per_xcd * 128));
  CHECK(hipMemset(buf, 0, (size_t)8 * lines_per_xcd * 128));
  float* out;
  CHECK(hipMalloc(&out, 64u << 20));
  hipStream_t sm, sg;
  CHECK(hipStreamCreate(&sm));
  CHECK(hipStreamCreate(&sg));
  hipEvent_t e0, e1, f0, f1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));

  // work sized so that each kernel alone takes ~2 ms: M: waves x iters x 4 MFMAs x 64 cycles; G: bytes / ~12 TB/s
  const int m_total_iters = 18000;        // per SIMD, split over the waves that share it
  const long long g_total_iters = 3200;   // per 4-wave workgroup at 2 workgroups per CU; scaled with the workgroup count
  int m_gap = 0;
  auto run_m = [&](int mw, int lds_kb, hipStream_t st) {
    if (m_gap == 4)
      hipLaunchKernelGGL(mfma_kernel<4>, dim3(cus * mw), dim3(256), (size_t)lds_kb * 1024, st, out, m_total_iters / mw / 2);
    else if (m_gap == 12)
      hipLaunchKernelGGL(mfma_kernel<12>, dim3(cus * mw), dim3(256), (size_t)lds_kb * 1024, st, out, m_total_iters / mw / 4);
    else
      hipLaunchKernelGGL(mfma_kernel<0>, dim3(cus * mw), dim3(256), (size_t)lds_kb * 1024, st, out, m_total_iters / mw);
  };
  int g_prio = 0;
  auto run_g = [&](int gw, hipStream_t st) {
    if (g_prio)
      hipLaunchKernelGGL(gather_kernel<3>, dim3(cus * gw), dim3(256), 0, st, buf, lines_per_xcd, (int)(g_total_iters * 2 / gw),
                         reinterpret_cast<float4*>(out) + (8u << 20) / 16);
    else
      hipLaunchKernelGGL(gather_kernel<0>, dim3(cus * gw), dim3(256), 0, st, buf, lines_per_xcd, (int)(g_total_iters * 2 / gw),
                         reinterpret_cast<float4*>(out) + (8u << 20) / 16);
  };
  auto time_alone = [&](auto&& fn, hipStream_t st) {
    fn(st);  // warm-up (and L2 fill for the gather)
    CHECK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      CHECK(hipEventRecord(e0, st));
      fn(st);
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    return best;
  };
  printf("# %d CUs; M = MFMA chain on registers (mw waves per SIMD, lds KB per workgroup), G = L2 gather (gw 4-wave workgroups per CU)\n", cus);
  printf("# %3s %4s %3s | %8s %8s | %8s %8s | %s\n", "mw", "lds", "gw", "M alone", "G alone", "M both", "G both", "speed M + speed G");
  {  // Round 5: one MFMA wave per SIMD of each opcode beside the gather (gw = 2 and 6); iteration counts sized to ~2 ms alone
    struct Op { const char* name; int mt; int iters; };
    // cycles per MFMA: f32 32x32x2 64, f32 16x16x4 32, bf16 32x32x16 (8 passes x 4 clk x 2 = measured below; sized from a first timing)
    const Op ops[] = {{"v_mfma_f32_32x32x2_f32", 0, m_total_iters}, {"v_mfma_f32_16x16x4_f32", 1, 2 * m_total_iters},
                      {"v_mfma_f32_32x32x16_bf16", 2, 2 * m_total_iters}};
    printf("# opcode sweep: one MFMA wave per SIMD (4 independent accumulators, registers only) beside the L2 gather\n");
    printf("# %-26s %3s | %8s %8s | %8s %8s | %s\n", "opcode", "gw", "M alone", "G alone", "M both", "G both", "speed M + speed G");
    for (const Op& op : ops) {
      auto run_mt = [&](hipStream_t st) {
        if (op.mt == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(cus), dim3(256), 0, st, out, op.iters);
        else if (op.mt == 1) hipLaunchKernelGGL(mfma_kernel_t<1>, dim3(cus), dim3(256), 0, st, out, op.iters);
        else hipLaunchKernelGGL(mfma_kernel_t<2>, dim3(cus), dim3(256), 0, st, out, op.iters);
      };
      for (int gw : {2, 6}) {
        const float ma = time_alone(run_mt, sm);
        const float ga = time_alone([&](hipStream_t st) { run_g(gw, st); }, sg);
        float mb = 1e9f, gb = 1e9f;
        for (int r = 0; r < 3; ++r) {
          CHECK(hipDeviceSynchronize());
          CHECK(hipEventRecord(e0, sm)); run_mt(sm); CHECK(hipEventRecord(e1, sm));
          CHECK(hipEventRecord(f0, sg)); run_g(gw, sg); CHECK(hipEventRecord(f1, sg));
          CHECK(hipDeviceSynchronize());
          float a, b;
          CHECK(hipEventElapsedTime(&a, e0, e1));
          CHECK(hipEventElapsedTime(&b, f0, f1));
          if (a + b < mb + gb) { mb = a; gb = b; }
        }
        const float t = mb < gb ? mb : gb;
        const float m_done = mb <= gb ? 1.f : (1.f - (mb - t) / ma);
        const float g_done = gb <= mb ? 1.f : (1.f - (gb - t) / ga);
        printf("  %-26s %3d | %8.3f %8.3f | %8.3f %8.3f | %.2f + %.2f = %.2f   (%.1f cycles per MFMA alone at 2.4 GHz)\n", op.name, gw, ma,
               ga, mb, gb, m_done * ma / t, g_done * ga / t, (m_done * ma + g_done * ga) / t, ma * 2.4e6 / ((double)op.iters * 4));
        fflush(stdout);
      }
    }
  }
  const int mws[] = {1, 2};
  const int ldss[] = {0};
  const int gws[] = {2, 6};
  for (int prio = 0; prio < 1; ++prio)
  for (int mw : mws)
    for (int lds : ldss) {
      g_prio = prio;
      if (mw == mws[0] && lds == ldss[0]) printf("# gather waves at s_setprio %d\n", prio ? 3 : 0);
      if (lds && mw > 2) continue;  // 64 KB x 4 would not fit anyway
      for (int gw : gws) {
        const float ma = time_alone([&](hipStream_t st) { run_m(mw, lds, st); }, sm);
        const float ga = time_alone([&](hipStream_t st) { run_g(gw, st); }, sg);
        float mb = 1e9f, gb = 1e9f;
        for (int r = 0; r < 3; ++r) {
          CHECK(hipDeviceSynchronize());
          CHECK(hipEventRecord(e0, sm));
          run_m(mw, lds, sm);
          CHECK(hipEventRecord(e1, sm));
          CHECK(hipEventRecord(f0, sg));
          run_g(gw, sg);
          CHECK(hipEventRecord(f1, sg));
          CHECK(hipDeviceSynchronize());
          float a, b;
          CHECK(hipEventElapsedTime(&a, e0, e1));
          CHECK(hipEventElapsedTime(&b, f0, f1));
          if (a + b < mb + gb) { mb = a; gb = b; }
        }
        // both kernels start together; the shorter one ran beside the other for its whole length, the longer one partly alone:
        // speeds while overlapped, from the shorter kernel's point of view
        const float t = mb < gb ? mb : gb;                       // overlapped interval
        const float m_done = mb <= gb ? 1.f : (1.f - (mb - t) / ma);  // fraction of M's work done in it
        const float g_done = gb <= mb ? 1.f : (1.f - (gb - t) / ga);
        printf("  %3d %4d %3d | %8.3f %8.3f | %8.3f %8.3f | %.2f + %.2f = %.2f\n", mw, lds, gw, ma, ga, mb, gb,
               m_done * ma / t, g_done * ga / t, (m_done * ma + g_done * ga) / t);
        fflush(stdout);
      }
    }
  g_prio = 0;
  for (int gap : {4, 12}) {  // the matrix core busy 1/2 and 1/4 of the time (one MFMA wave per SIMD)
    m_gap = gap;
    const float ma = time_alone([&](hipStream_t st) { run_m(1, 0, st); }, sm);
    const float ga = time_alone([&](hipStream_t st) { run_g(4, st); }, sg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, sm)); run_m(1, 0, sm); CHECK(hipEventRecord(e1, sm));
    CHECK(hipEventRecord(f0, sg)); run_g(4, sg); CHECK(hipEventRecord(f1, sg));
    CHECK(hipDeviceSynchronize());
    float a, b;
    CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, f0, f1));
    printf("# matrix core busy %d/%d of the time: M alone %.3f G alone %.3f | M both %.3f G both %.3f\n", 4, 4 + gap, ma, ga, a, b);
  }
  m_gap = 0;
  {  // one instruction stream: 2 workgroups per CU (2 waves per SIMD), the gather's iteration count, MPI x 4 MFMAs per iteration
    const int it = (int)g_total_iters;
    auto t_fused = [&](auto kernel) {
      return time_alone([&](hipStream_t st) {
        hipLaunchKernelGGL(kernel, dim3(cus * 2), dim3(256), 0, st, buf, lines_per_xcd, it, reinterpret_cast<float4*>(out) + (8u << 20) / 16);
      }, sg);
    };
    const float g_only = t_fused(fused_kernel<0>);
    const float f2 = t_fused(fused_kernel<2>), f4 = t_fused(fused_kernel<4>), f8 = t_fused(fused_kernel<8>);
    // MFMA time of the same counts alone: 2 waves per SIMD x it x MPI x 4 MFMAs x 64 cycles
    printf("# one instruction stream (2 waves per SIMD, %d iterations of 8 lines per lane group): gather only %.3f ms; with 8 / 16 / 32 MFMAs per "
           "iteration %.3f / %.3f / %.3f ms (their MFMAs alone: %.3f / %.3f / %.3f ms at 64 cycles each)\n", it, g_only, f2, f4, f8,
           2.0 * it * 8 * 64 / 2.4e6, 2.0 * it * 16 * 64 / 2.4e6, 2.0 * it * 32 * 64 / 2.4e6);
  }
  // controls: (a) the same pair with G enqueued first; (b) two gathers on the two streams (do two streams' kernels share
  // the chip at all?); (c) two half-occupancy MFMA kernels
  {
    const float ma = time_alone([&](hipStream_t st) { run_m(1, 0, st); }, sm);
    const float ga = time_alone([&](hipStream_t st) { run_g(4, st); }, sg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(f0, sg)); run_g(4, sg); CHECK(hipEventRecord(f1, sg));
    CHECK(hipEventRecord(e0, sm)); run_m(1, 0, sm); CHECK(hipEventRecord(e1, sm));
    CHECK(hipDeviceSynchronize());
    float a, b;
    CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, f0, f1));
    printf("# G enqueued first: M alone %.3f G alone %.3f | M both %.3f G both %.3f\n", ma, ga, a, b);
    const float g2 = time_alone([&](hipStream_t st) { run_g(2, st); }, sg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(f0, sg)); run_g(2, sg); CHECK(hipEventRecord(f1, sg));
    CHECK(hipEventRecord(e0, sm)); run_g(2, sm); CHECK(hipEventRecord(e1, sm));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, f0, f1));
    printf("# G(gw=2) + G(gw=2) on two streams: alone %.3f each | together %.3f and %.3f\n", g2, a, b);
    const float m1 = time_alone([&](hipStream_t st) { run_m(1, 0, st); }, sm);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(f0, sg)); run_m(1, 0, sg); CHECK(hipEventRecord(f1, sg));
    CHECK(hipEventRecord(e0, sm)); run_m(1, 0, sm); CHECK(hipEventRecord(e1, sm));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, f0, f1));
    printf("# M(mw=1) + M(mw=1) on two streams: alone %.3f each | together %.3f and %.3f\n", m1, a, b);
  }
  return 0;
}
