// K3 (cosine-distance triplet loss fwd+bwd), K4 (Adam) and the split-K slab
// reduction — the bandwidth-trivial kernels of the hot path (gfx950).
#include "tt_common.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * TT_WAVE;

// One wave per triplet row.  Follows the op order of F.cosine_similarity
// (normalise each operand by max(|x|, 1e-8), then dot) so that values agree with
// the reference to rounding; gradient is the closed form checked against
// autograd in oracle/two_towers_oracle.py::triplet_loss_closed_form.
// Optional in-launch loss mean ("last workgroup reduces"): every workgroup publishes the sum of its
// rows' hinges, takes a ticket, and the workgroup that draws the last ticket adds the partials up in
// a fixed order.  Hand-off in the fence-free form of cdna_hip_programming.md Guideline 16 /
// MI355X_MICROARCH.md "Valid forms": ONE lane per workgroup stores its 4-byte partial write-through
// (relaxed agent-scope atomic store = `global_store ... sc1`), drains it (vmcnt(0)), then takes the
// ticket with a relaxed agent-scope fetch_add; the workgroup whose add returned the last ticket
// reads every partial with sc1 loads (relaxed agent-scope atomic loads) after a workgroup barrier.
// No release/acquire fence: a release here would write back the L2 lines of the 9 MB of gradients
// this same kernel is storing (measured +15 us).  The last workgroup re-zeroes the ticket.
struct TripletMean {
  float* partials;       // [gridDim.x]; nullptr = no in-launch mean
  unsigned* ticket;      // zero before the first launch
  float denom;
  float* loss;           // last batch loss
  double* acc_sum;       // running sum of batch losses (may be null)
  long long* acc_count;
};

// NC = ceil(P / 64) columns held per lane (register-resident rows).  MEAN: in-launch loss mean.
template <int NC, bool MEAN>
__global__ __launch_bounds__(kThreads) void triplet_kernel(
    const float* __restrict__ q, const float* __restrict__ p, const float* __restrict__ n, float margin,
    float grad_scale, long long B, int P, float* __restrict__ pos_dist, float* __restrict__ neg_dist,
    float* __restrict__ hinge_rows, float* __restrict__ dq, float* __restrict__ dp, float* __restrict__ dn,
    const int* __restrict__ neg_index, const TripletMean mean) {
  __shared__ float bh[kWavesPerBlock];
  __shared__ int is_last;
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const long long row_raw = (long long)blockIdx.x * kWavesPerBlock + wave;
  const bool active = row_raw < B;
  if (!MEAN && !active) return;
  const long long row = active ? row_raw : B - 1;  // idle waves of the last workgroup recompute row B-1, store nothing
  const float eps = 1e-8f;
  const long long o = row * (long long)P;
  // in-batch negatives (backend/data.py:113-137): negative i is positive neg_index[i]; `n` then
  // points at the positives' embeddings
  const long long on = neg_index ? (long long)min(max(neg_index[row], 0), (int)B - 1) * P : o;

  float xq[NC], xp[NC], xn[NC];
  float sq = 0.f, sp = 0.f, sn = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * TT_WAVE + lane;
    const bool ok = col < P;
    xq[c] = ok ? q[o + col] : 0.f;
    xp[c] = ok ? p[o + col] : 0.f;
    xn[c] = ok ? n[on + col] : 0.f;
    sq = fmaf(xq[c], xq[c], sq);
    sp = fmaf(xp[c], xp[c], sp);
    sn = fmaf(xn[c], xn[c], sn);
  }
  const float nq = sqrtf(tt_wave_sum(sq)), np_ = sqrtf(tt_wave_sum(sp)), nn = sqrtf(tt_wave_sum(sn));
  const float cq = fmaxf(nq, eps), cp = fmaxf(np_, eps), cn = fmaxf(nn, eps);

  float dqp = 0.f, dqn = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    xq[c] = xq[c] / cq;  // unit vectors from here on
    xp[c] = xp[c] / cp;
    xn[c] = xn[c] / cn;
    dqp = fmaf(xq[c], xp[c], dqp);
    dqn = fmaf(xq[c], xn[c], dqn);
  }
  const float cos_p = tt_wave_sum(dqp), cos_n = tt_wave_sum(dqn);
  const float pd = 1.f - cos_p, nd = 1.f - cos_n;  // model.py:135
  const float hinge = pd - nd + margin;            // model.py:143
  if (lane == 0 && active) {
    if (pos_dist) pos_dist[row] = pd;
    if (neg_dist) neg_dist[row] = nd;
    hinge_rows[row] = fmaxf(hinge, 0.f);
  }
  if (MEAN) {
    if (lane == 0) bh[wave] = active ? fmaxf(hinge, 0.f) : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(mean.partials + blockIdx.x, ((bh[0] + bh[1]) + bh[2]) + bh[3], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(mean.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (is_last) {  // workgroup-uniform; the barrier above orders every wave's loads after the last ticket
      float s = 0.f;
      for (unsigned i = threadIdx.x; i < gridDim.x; i += kThreads)
        s += __hip_atomic_load(mean.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s = tt_wave_sum(s);
      if (lane == 0) bh[wave] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float l = (((bh[0] + bh[1]) + bh[2]) + bh[3]) / mean.denom;
        *mean.loss = l;
        if (mean.acc_sum) {
          *mean.acc_sum += static_cast<double>(l);
          *mean.acc_count += 1;
        }
        __hip_atomic_store(mean.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (!dq || !active) return;

  // d cos(a,b)/da = (v - u cos r)/c, r = c/|a| (|a|>0) else 0  (r == 1 unless |a| < eps)
  const float g = hinge > 0.f ? grad_scale : 0.f;
  const float rq = nq > 0.f ? cq / nq : 0.f, rp = np_ > 0.f ? cp / np_ : 0.f, rn = nn > 0.f ? cn / nn : 0.f;
  const float kqp = cos_p * rq, kqn = cos_n * rq, kp = cos_p * rp, kn = cos_n * rn;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * TT_WAVE + lane;
    if (col < P) {
      const float u = xq[c], vp = xp[c], vn = xn[c];
      const float dcp_dq = (vp - u * kqp) / cq;
      const float dcn_dq = (vn - u * kqn) / cq;
      dq[o + col] = g * (dcn_dq - dcp_dq);
      dp[o + col] = -g * ((u - vp * kp) / cp);
      dn[o + col] = g * ((u - vn * kn) / cn);
    }
  }
}

// In-batch negatives: the document-tower output row j takes part in its own triplet (as the
// positive) and in every triplet that drew it as the negative, so its gradient is
// dp[j] + sum_{i: neg_index[i]==j} dn[i].  One wave per destination row j scans neg_index in
// ascending i (fixed order => reproducible), no atomics.
__global__ __launch_bounds__(kThreads) void scatter_neg_grad_kernel(float* __restrict__ dp,
                                                                    const float* __restrict__ dn,
                                                                    const int* __restrict__ neg_index,
                                                                    int B, int P) {
  constexpr int G = 8;  // index chunks requested together (one dependent load latency per G*64 rows)
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int j = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (j >= B) return;
  for (int base = 0; base < B; base += G * TT_WAVE) {
    int idx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int i = base + g * TT_WAVE + lane;
      idx[g] = i < B ? neg_index[i] : -1;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      unsigned long long hit = __ballot(idx[g] == j);
      while (hit) {
        const int src = base + g * TT_WAVE + __builtin_ctzll(hit);
        hit &= hit - 1;
        for (int c = lane; c < P; c += TT_WAVE) dp[(long long)j * P + c] += dn[(long long)src * P + c];
      }
    }
  }
}

// The same scatter for large batches (the scan above reads all B indices per destination row: B^2 index reads, 268 M
// at B = 16384).  Sort by destination instead — a CSR of the sources of every destination row: count (integer
// atomics: the counts do not depend on their order), exclusive scan by one workgroup, fill (the order inside a
// bucket is whatever the atomics gave), then one wave per destination row adds its sources in ASCENDING source order
// (repeated minimum selection inside the bucket, which holds a handful of entries), so the sums are still
// reproducible bit for bit.
__global__ __launch_bounds__(kThreads) void csr_count_kernel(const int* __restrict__ neg_index, int B, int* __restrict__ counts) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < B) atomicAdd(&counts[min(max(neg_index[i], 0), B - 1)], 1);
}

// offsets[j] = sum of counts[0..j) (exclusive), offsets[B] = B; `counts` and `offsets` may not alias.  One workgroup.
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets,
                                                        int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (B + 1023) / 1024;
  const int lo = min(t * per, B), hi = min(lo + per, B);
  int s = 0;
  for (int j = lo; j < hi; ++j) s += counts[j];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the 1024 chunk sums
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;  // exclusive prefix of this thread's chunk
  for (int j = lo; j < hi; ++j) {
    offsets[j] = run;
    cursor[j] = run;
    run += counts[j];
  }
  if (t == 1023) offsets[B] = part[1023];
}

__global__ __launch_bounds__(kThreads) void csr_fill_kernel(const int* __restrict__ neg_index, int B, int* __restrict__ cursor,
                                                            int* __restrict__ list) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < B) list[atomicAdd(&cursor[min(max(neg_index[i], 0), B - 1)], 1)] = i;
}

__global__ __launch_bounds__(kThreads) void scatter_neg_grad_csr_kernel(float* __restrict__ dp, const float* __restrict__ dn,
                                                                        const int* __restrict__ offsets,
                                                                        const int* __restrict__ list, int B, int P) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const int j = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (j >= B) return;
  const int lo = offsets[j], n = offsets[j + 1] - lo;
  int prev = -1;
  for (int k = 0; k < n; ++k) {  // the smallest source index above `prev`
    int cur = 0x7fffffff;
    for (int base = 0; base < n; base += TT_WAVE) {
      const int v = base + lane < n ? list[lo + base + lane] : 0x7fffffff;
      if (v > prev) cur = min(cur, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cur = min(cur, __shfl_xor(cur, off, TT_WAVE));
    for (int c = lane; c < P; c += TT_WAVE) dp[(long long)j * P + c] += dn[(long long)cur * P + c];
    prev = cur;
  }
}

// Single workgroup: loss = (sum_b hinge_b) / denom in a fixed order; optionally adds
// into a running {sum, count, last} accumulator (the engine's deferred loss.item()).
__global__ __launch_bounds__(kThreads) void loss_mean_kernel(const float* __restrict__ hinge_rows,
                                                             long long B, float denom, float* __restrict__ loss,
                                                             double* __restrict__ acc_sum,
                                                             long long* __restrict__ acc_count) {
  __shared__ float red[kWavesPerBlock];
  // The order of triplet_kernel's in-launch mean, so that the two backward schedules (loss mean in the triplet launch /
  // as this launch) report bitwise the same loss: first the sum of each run of kWavesPerBlock rows — one triplet
  // workgroup's partial — then thread t adds partials t, t + 256, ...
  static_assert(kWavesPerBlock == 4, "one triplet workgroup's partial is the sum of four rows");
  float s = 0.f;
  const long long nparts = (B + kWavesPerBlock - 1) / kWavesPerBlock;
  for (long long i = threadIdx.x; i < nparts; i += kThreads) {
    const long long r = i * kWavesPerBlock;
    const float h0 = hinge_rows[r], h1 = r + 1 < B ? hinge_rows[r + 1] : 0.f, h2 = r + 2 < B ? hinge_rows[r + 2] : 0.f,
                h3 = r + 3 < B ? hinge_rows[r + 3] : 0.f;
    s += ((h0 + h1) + h2) + h3;
  }
  s = tt_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = (((red[0] + red[1]) + red[2]) + red[3]) / denom;
    if (loss) *loss = l;
    if (acc_sum) {
      *acc_sum += static_cast<double>(l);
      *acc_count += 1;
    }
  }
}

// torch.optim.Adam single-tensor op order (see oracle adam_step): fp32 throughout,
// un-fused multiplies/adds where ATen's vectorised CPU kernels do not fuse.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float one_minus_b1, float b2,
                                          float one_minus_b2, float step_size, float bc2_sqrt, float eps) {
  m = __fadd_rn(m, __fmul_rn(one_minus_b1, __fsub_rn(g, m)));                        // m.lerp_(g, 1-b1)
  v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(one_minus_b2, g), g));         // v.mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);            // (v.sqrt()/bc2_sqrt).add_(eps)
  p = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(m, denom)));                      // p.addcdiv_(m, denom, -step_size)
}

// Device-side step counter for graph replay: one thread bumps the step and derives the two
// step-dependent scalars in double, like torch does on the host for a Python-float step.
__global__ void adam_prep_kernel(long long* __restrict__ step, float* __restrict__ scal, double lr, double b1,
                                 double b2) {
  const long long s = *step + 1;
  *step = s;
  scal[0] = static_cast<float>(lr / (1.0 - pow(b1, static_cast<double>(s))));   // step_size
  scal[1] = static_cast<float>(sqrt(1.0 - pow(b2, static_cast<double>(s))));    // bc2_sqrt
}

__global__ void set_step_kernel(long long* __restrict__ step, long long value) { *step = value; }

__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        int nslabs, long long slab_stride,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long long n, float one_minus_b1, float b2,
                                                        float one_minus_b2, float step_size,
                                                        float bc2_sqrt, float eps, float grad_scale,
                                                        const float* __restrict__ scal) {
  const long long i4 = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (scal) {  // step-dependent scalars produced on the device (adam_prep_kernel)
    step_size = scal[0];
    bc2_sqrt = scal[1];
  }
  if (i4 + 3 < n) {
    float4 gs = *reinterpret_cast<const float4*>(g + i4);
    for (int z = 1; z < nslabs; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(g + z * slab_stride + i4);
      gs.x += t.x; gs.y += t.y; gs.z += t.z; gs.w += t.w;
    }
    float4 pp = *reinterpret_cast<float4*>(p + i4);
    float4 mm = *reinterpret_cast<float4*>(m + i4);
    float4 vv = *reinterpret_cast<float4*>(v + i4);
    float* ge = &gs.x; float* pe = &pp.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      adam_elem(pe[e], me[e], ve[e], ge[e] * grad_scale, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps);
    }
    *reinterpret_cast<float4*>(p + i4) = pp;
    *reinterpret_cast<float4*>(m + i4) = mm;
    *reinterpret_cast<float4*>(v + i4) = vv;
  } else {
    for (long long i = i4; i < n; ++i) {
      float gg = g[i];
      for (int z = 1; z < nslabs; ++z) gg += g[z * slab_stride + i];
      gg *= grad_scale;
      adam_elem(p[i], m[i], v[i], gg, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps);
    }
  }
}

// Multi-tensor Adam: ONE launch over up to kAdamMaxTensors separate tensors (the reference's optimizer walks the 8
// trainable tensors of model.parameters(), backend/training.py:436).  A workgroup finds its tensor by
// scanning the per-tensor first-block table (kernel argument), then does adam_kernel's work on its 1024 elements.
constexpr int kAdamMaxTensors = 16;
struct AdamMulti {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  long long n[kAdamMaxTensors];
  int first_block[kAdamMaxTensors + 1];  // first_block[count] = grid size
  int count;
};

__global__ __launch_bounds__(kThreads) void adam_multi_kernel(const AdamMulti a, float one_minus_b1, float b2,
                                                              float one_minus_b2, float step_size, float bc2_sqrt,
                                                              float eps) {
  int t = 0;
#pragma unroll
  for (int i = 1; i < kAdamMaxTensors; ++i)
    if (i < a.count && (int)blockIdx.x >= a.first_block[i]) t = i;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const long long n = a.n[t];
  const long long i4 = ((long long)((int)blockIdx.x - a.first_block[t]) * kThreads + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 3 < n) {
    const float4 gs = *reinterpret_cast<const float4*>(g + i4);
    float4 pp = *reinterpret_cast<float4*>(p + i4);
    float4 mm = *reinterpret_cast<float4*>(m + i4);
    float4 vv = *reinterpret_cast<float4*>(v + i4);
    const float* ge = &gs.x; float* pe = &pp.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) adam_elem(pe[e], me[e], ve[e], ge[e], one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps);
    *reinterpret_cast<float4*>(p + i4) = pp;
    *reinterpret_cast<float4*>(m + i4) = mm;
    *reinterpret_cast<float4*>(v + i4) = vv;
  } else {
    for (long long i = i4; i < n; ++i) adam_elem(p[i], m[i], v[i], g[i], one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps);
  }
}

// 1 - F.cosine_similarity(x, y, dim=1) per row and its gradient (backend/model.py:132-135), one wave per row, the
// op order and the closed-form gradient of triplet_kernel (both operands divided by max(|.|, 1e-8), then dotted;
// d cos(a,b)/da = (v - u cos r)/c with r = c/|a| for |a| > 0, else 0).  g == nullptr: forward only.
template <int NC>
__global__ __launch_bounds__(kThreads) void cosine_distance_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ g, long long B, int P,
                                                                   float* __restrict__ dist, float* __restrict__ dx,
                                                                   float* __restrict__ dy) {
  const int lane = threadIdx.x & (TT_WAVE - 1);
  const long long row = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= B) return;
  const float eps = 1e-8f;
  const long long o = row * (long long)P;
  float xa[NC], xb[NC];
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * TT_WAVE + lane;
    const bool ok = col < P;
    xa[c] = ok ? x[o + col] : 0.f;
    xb[c] = ok ? y[o + col] : 0.f;
    sa = fmaf(xa[c], xa[c], sa);
    sb = fmaf(xb[c], xb[c], sb);
  }
  const float na = sqrtf(tt_wave_sum(sa)), nb = sqrtf(tt_wave_sum(sb));
  const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
  float d = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    xa[c] = xa[c] / ca;
    xb[c] = xb[c] / cb;
    d = fmaf(xa[c], xb[c], d);
  }
  const float cs = tt_wave_sum(d);
  if (lane == 0 && dist) dist[row] = 1.f - cs;
  if (!g) return;
  const float gr = -g[row];  // d(1 - cos) = -d cos
  const float ra = na > 0.f ? ca / na : 0.f, rb = nb > 0.f ? cb / nb : 0.f;
  const float ka = cs * ra, kb = cs * rb;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = c * TT_WAVE + lane;
    if (col < P) {
      dx[o + col] = gr * ((xb[c] - xa[c] * ka) / ca);
      dy[o + col] = gr * ((xa[c] - xb[c] * kb) / cb);
    }
  }
}

__global__ __launch_bounds__(kThreads) void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs,
                                                                long long slab_stride,
                                                                float* __restrict__ out, long long n,
                                                                int accumulate) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  float s = slabs[i];
  for (int z = 1; z < nslabs; ++z) s += slabs[z * slab_stride + i];
  out[i] = accumulate ? out[i] + s : s;
}

}  // namespace

// internal (engine.hip): loss mean with the running accumulator
int tt_launch_loss_mean(const float* hinge_rows, long long B, float denom, float* loss, double* acc_sum,
                        long long* acc_count, hipStream_t st) {
  hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(kThreads), 0, st, hinge_rows, B, denom, loss, acc_sum, acc_count);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

// internal: triplet rows only (no mean)
int tt_launch_triplet_rows(const float* q, const float* p, const float* n, float margin, float grad_scale,
                           long long B, int P, float* pos_dist, float* neg_dist, float* hinge_rows,
                           float* dq, float* dp, float* dn, const int* neg_index, hipStream_t st,
                           float* mean_partials = nullptr, unsigned* mean_ticket = nullptr, float mean_denom = 1.f,
                           float* mean_loss = nullptr, double* acc_sum = nullptr, long long* acc_count = nullptr,
                           int* csr_ws = nullptr) {
  const TripletMean mean{mean_partials, mean_ticket, mean_denom, mean_loss, acc_sum, acc_count};
  const dim3 grid((unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock)), block(kThreads);
  const int nc = (P + TT_WAVE - 1) / TT_WAVE;
#define TT_TRIP(NC)                                                                                           \
  do {                                                                                                        \
    if (mean_partials)                                                                                        \
      hipLaunchKernelGGL((triplet_kernel<NC, true>), grid, block, 0, st, q, p, n, margin, grad_scale, B, P,   \
                         pos_dist, neg_dist, hinge_rows, dq, dp, dn, neg_index, mean);                        \
    else                                                                                                      \
      hipLaunchKernelGGL((triplet_kernel<NC, false>), grid, block, 0, st, q, p, n, margin, grad_scale, B, P,  \
                         pos_dist, neg_dist, hinge_rows, dq, dp, dn, neg_index, mean);                        \
  } while (0)
  if (nc <= 1) TT_TRIP(1);
  else if (nc <= 2) TT_TRIP(2);
  else if (nc <= 4) TT_TRIP(4);
  else if (nc <= 6) TT_TRIP(6);
  else if (nc <= 8) TT_TRIP(8);
  else if (nc <= 12) TT_TRIP(12);
  else if (nc <= 16) TT_TRIP(16);
  else return TT_ERR_UNSUPPORTED;  // P > 1024
#undef TT_TRIP
  TT_CHECK_LAUNCH();
  if (neg_index && dp && dn) {
    if (csr_ws && B > 4096) {  // csr_ws: 4 * B + 1 ints — counts [B] | offsets [B + 1] | cursor [B] | list [B]
      int* counts = csr_ws;
      int* offsets = counts + B;
      int* cursor = offsets + B + 1;
      int* list = cursor + B;
      const dim3 gb((unsigned)((B + kThreads - 1) / kThreads));
      TT_HIP(hipMemsetAsync(counts, 0, (size_t)B * sizeof(int), st));
      hipLaunchKernelGGL(csr_count_kernel, gb, block, 0, st, neg_index, (int)B, counts);
      hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, counts, (int)B, offsets, cursor);
      hipLaunchKernelGGL(csr_fill_kernel, gb, block, 0, st, neg_index, (int)B, cursor, list);
      hipLaunchKernelGGL(scatter_neg_grad_csr_kernel, grid, block, 0, st, dp, dn, offsets, list, (int)B, P);
    } else {
      hipLaunchKernelGGL(scatter_neg_grad_kernel, grid, block, 0, st, dp, dn, neg_index, (int)B, P);
    }
    TT_CHECK_LAUNCH();
  }
  return TT_OK;
}

extern "C" int tt_triplet_fwd_bwd_f32(const float* q, const float* p, const float* n, float margin,
                                      float grad_scale, int64_t B, int P, float* loss, float* pos_dist,
                                      float* neg_dist, float* hinge_rows, float* dq, float* dp, float* dn,
                                      tt_stream_t stream) {
  TT_REQUIRE(q && p && n && hinge_rows, TT_ERR_NULL_POINTER);
  TT_REQUIRE((dq && dp && dn) || (!dq && !dp && !dn), TT_ERR_NULL_POINTER);
  TT_REQUIRE(B >= 1 && P >= 1 && P <= 1024, TT_ERR_BAD_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = tt_launch_triplet_rows(q, p, n, margin, grad_scale, B, P, pos_dist, neg_dist, hinge_rows, dq, dp, dn, nullptr, st);
  if (rc != TT_OK) return rc;
  if (loss) return tt_launch_loss_mean(hinge_rows, B, static_cast<float>(B), loss, nullptr, nullptr, st);
  return TT_OK;
}

extern "C" int tt_adam_f32(float* params, const float* grads, int nslabs, int64_t slab_stride,
                           float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                           double eps, int64_t step, float grad_scale, tt_stream_t stream) {
  TT_REQUIRE(params && grads && exp_avg && exp_avg_sq, TT_ERR_NULL_POINTER);
  TT_REQUIRE(n >= 0 && nslabs >= 1 && step >= 1, TT_ERR_BAD_SHAPE);
  TT_REQUIRE(tt_aligned16(params) && tt_aligned16(grads) && tt_aligned16(exp_avg) && tt_aligned16(exp_avg_sq),
             TT_ERR_ALIGNMENT);
  TT_REQUIRE(nslabs == 1 || (slab_stride % 4) == 0, TT_ERR_ALIGNMENT);
  if (n == 0) return TT_OK;
  // bias corrections in double on the host, as torch does for a python-float step
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const long long nthreads = (n + 3) / 4;
  const dim3 grid((unsigned)((nthreads + kThreads - 1) / kThreads)), block(kThreads);
  hipLaunchKernelGGL(adam_kernel, grid, block, 0, static_cast<hipStream_t>(stream), params, grads, nslabs,
                     (long long)slab_stride, exp_avg, exp_avg_sq, (long long)n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps, grad_scale,
                     static_cast<const float*>(nullptr));
  TT_CHECK_LAUNCH();
  return TT_OK;
}

// internal (engine.hip): Adam whose step lives on the device (graph-replayable)
int tt_launch_adam_auto(float* params, const float* grads, int nslabs, long long slab_stride, float* exp_avg,
                        float* exp_avg_sq, long long n, double lr, double beta1, double beta2, double eps,
                        long long* dev_step, float* dev_scal, hipStream_t st) {
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, st, dev_step, dev_scal, lr, beta1, beta2);
  TT_CHECK_LAUNCH();
  const long long nthreads = (n + 3) / 4;
  const dim3 grid((unsigned)((nthreads + kThreads - 1) / kThreads)), block(kThreads);
  hipLaunchKernelGGL(adam_kernel, grid, block, 0, st, params, grads, nslabs, slab_stride, exp_avg, exp_avg_sq, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, 1.f, (float)eps, 1.f,
                     static_cast<const float*>(dev_scal));
  TT_CHECK_LAUNCH();
  return TT_OK;
}

int tt_launch_set_step(long long* dev_step, long long value, hipStream_t st) {
  hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, st, dev_step, value);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

extern "C" int tt_reduce_slabs_f32(const float* slabs, int nslabs, int64_t slab_stride, float* out, int64_t n,
                                   int accumulate, tt_stream_t stream) {
  TT_REQUIRE(slabs && out, TT_ERR_NULL_POINTER);
  TT_REQUIRE(n >= 0 && nslabs >= 1, TT_ERR_BAD_SHAPE);
  if (n == 0) return TT_OK;
  const dim3 grid((unsigned)((n + kThreads - 1) / kThreads)), block(kThreads);
  hipLaunchKernelGGL(reduce_slabs_kernel, grid, block, 0, static_cast<hipStream_t>(stream), slabs, nslabs,
                     (long long)slab_stride, out, (long long)n, accumulate);
  TT_CHECK_LAUNCH();
  return TT_OK;
}

extern "C" int tt_adam_multi_f32(const tt_adam_tensor* tensors, int count, double lr, double beta1, double beta2,
                                 double eps, int64_t step, tt_stream_t stream) {
  TT_REQUIRE(count >= 0 && step >= 1, TT_ERR_BAD_SHAPE);
  if (count == 0) return TT_OK;
  TT_REQUIRE(tensors, TT_ERR_NULL_POINTER);
  for (int i = 0; i < count; ++i) {
    const tt_adam_tensor& t = tensors[i];
    TT_REQUIRE(t.n >= 0, TT_ERR_BAD_SHAPE);
    if (t.n == 0) continue;
    TT_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, TT_ERR_NULL_POINTER);
    TT_REQUIRE(tt_aligned16(t.param) && tt_aligned16(t.grad) && tt_aligned16(t.exp_avg) && tt_aligned16(t.exp_avg_sq),
               TT_ERR_ALIGNMENT);
  }
  const double bc1 = 1.0 - pow(beta1, (double)step);   // in double on the host, as torch does for a python-float step
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  // one launch per kAdamMaxTensors NON-EMPTY tensors; `i` carries on across launches (empty tensors are skipped without
  // taking a place, so a chunk may consume more than kAdamMaxTensors input entries)
  for (int i = 0; i < count;) {
    AdamMulti a{};
    long long blocks = 0;
    for (; i < count && a.count < kAdamMaxTensors; ++i) {
      const tt_adam_tensor& t = tensors[i];
      if (t.n == 0) continue;
      const int k = a.count++;
      a.p[k] = t.param; a.g[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq; a.n[k] = t.n;
      a.first_block[k] = (int)blocks;
      blocks += ((t.n + 3) / 4 + kThreads - 1) / kThreads;
      TT_REQUIRE(blocks <= 0x7fffffffLL, TT_ERR_BAD_SHAPE);
    }
    if (a.count == 0) continue;
    a.first_block[a.count] = (int)blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps);
    TT_CHECK_LAUNCH();
  }
  return TT_OK;
}

static int launch_cosine_distance(const float* x, const float* y, const float* g, int64_t B, int P, float* dist, float* dx,
                                  float* dy, hipStream_t st) {
  const dim3 grid((unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock)), block(kThreads);
  const int nc = (P + TT_WAVE - 1) / TT_WAVE;
#define TT_COSD(NC) hipLaunchKernelGGL((cosine_distance_kernel<NC>), grid, block, 0, st, x, y, g, (long long)B, P, dist, dx, dy)
  if (nc <= 1) TT_COSD(1);
  else if (nc <= 2) TT_COSD(2);
  else if (nc <= 4) TT_COSD(4);
  else if (nc <= 6) TT_COSD(6);
  else if (nc <= 8) TT_COSD(8);
  else if (nc <= 12) TT_COSD(12);
  else if (nc <= 16) TT_COSD(16);
  else return TT_ERR_UNSUPPORTED;  // P > 1024
#undef TT_COSD
  TT_CHECK_LAUNCH();
  return TT_OK;
}

extern "C" int tt_cosine_distance_f32(const float* x, const float* y, int64_t B, int P, float* dist, tt_stream_t stream) {
  TT_REQUIRE(x && y && dist, TT_ERR_NULL_POINTER);
  TT_REQUIRE(B >= 0 && P >= 1 && P <= 1024, TT_ERR_BAD_SHAPE);
  if (B == 0) return TT_OK;
  return launch_cosine_distance(x, y, nullptr, B, P, dist, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int tt_cosine_distance_bwd_f32(const float* x, const float* y, const float* grad_dist, int64_t B, int P,
                                          float* dx, float* dy, tt_stream_t stream) {
  TT_REQUIRE(x && y && grad_dist && dx && dy, TT_ERR_NULL_POINTER);
  TT_REQUIRE(B >= 0 && P >= 1 && P <= 1024, TT_ERR_BAD_SHAPE);
  if (B == 0) return TT_OK;
  return launch_cosine_distance(x, y, grad_dist, B, P, nullptr, dx, dy, static_cast<hipStream_t>(stream));
}
