// What does an event cost the stream it is recorded on, and does binding it to a kernel's own completion signal
// (hipExtLaunchKernelGGL's stopEvent) make that cost go away?  (diagnostic, round 5)
//   one stream, pairs of ~20 us kernels: (C) no event, (A) hipEventRecord between the two, (B) the first launched with
//   stopEvent, (W) + a second stream waiting for the event and running a kernel of its own (is the edge honoured?).
//   hipcc -O3 --offload-arch=gfx950 tools/ext_event_probe.hip -o /tmp/ext_event_probe && /tmp/ext_event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(float* p, int iters, float v) {
  float a = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0000001f + 1e-7f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = a + v;
}
__global__ void check(const float* p, float* out, int n) {  // reads what spin wrote: ordering check across streams
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}

int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float* buf; CK(hipMalloc(&buf, 256 * 1024 * sizeof(float))); CK(hipMemset(buf, 0, 256 * 1024 * sizeof(float)));
  float* out; CK(hipMalloc(&out, 4)); 
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int N = 400, iters = 500;
  for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)}) {
    hipEvent_t ev[8]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags));
    auto run = [&](int mode) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s1));
      for (int i = 0; i < N; ++i) {
        hipEvent_t e = ev[i & 7];
        if (mode == 2 || mode == 3) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, nullptr, e, 0, buf, iters, 1.f);
        else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, buf, iters, 1.f);
        if (mode == 1) CK(hipEventRecord(e, s1));
        if (mode == 3 || mode == 4) {
          if (mode == 4) CK(hipEventRecord(e, s1));
          CK(hipStreamWaitEvent(s2, e, 0));
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, buf + 128 * 1024, 200, 0.f);
        }
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, buf, iters, 1.f);
      }
      CK(hipEventRecord(t1, s1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      return ms * 1e3f / N;
    };
    run(0);
    printf("event flags 0x%x: us per pair of kernels: no event %.2f | hipEventRecord between %.2f | stopEvent on the first %.2f | "
           "stopEvent + a waiting stream %.2f | hipEventRecord + a waiting stream %.2f\n", flags, run(0), run(1), run(2), run(3), run(4));
    for (auto& e : ev) CK(hipEventDestroy(e));
  }
  // ordering: s1 writes buf (spin adds 1 per launch to every element), s2 waits for the stopEvent and sums: must see the write
  hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  int bad = 0;
  for (int rep = 0; rep < 50; ++rep) {
    CK(hipMemsetAsync(buf, 0, 65536 * sizeof(float), s1)); CK(hipMemsetAsync(out, 0, 4, s1)); CK(hipStreamSynchronize(s1));
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, nullptr, e, 0, buf, 20000, 1.f);
    CK(hipStreamWaitEvent(s2, e, 0));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, s2, buf, out, 65536);
    CK(hipDeviceSynchronize());
    float h; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    if (h < 65535.5f) ++bad;
  }
  printf("ordering through a stopEvent: %d of 50 runs saw stale data\n", bad);
  return 0;
}
