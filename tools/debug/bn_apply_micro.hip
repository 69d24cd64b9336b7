// bn_apply_micro.hip — standalone timing of PostNet BatchNorm-apply variants at the training shape (6768 x 512 fp32 -> bf16 + keep bytes).
// Diagnostic only (not part of the library): which part of the 15-16 us of bn_apply2_kernel is Philox, the keep-byte stores, the
// partial-row sums, the 64-channel slab layout.   build: hipcc -O3 --offload-arch=gfx950 -o bn_apply_micro.out bn_apply_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return *reinterpret_cast<bf16_t*>(&b); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }
__device__ __forceinline__ uint4 philox(uint2 key, uint4 ctr) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
  }
  return ctr;
}

struct Args {
  const float* x; const float* partials; int nblk; const float* mean; const float* rstd; const float* gamma; const float* beta;
  bf16_t* out; uint8_t* keep; int rows, C;
};

// ---- variant A: the library's slab kernel (64 channels x 64 rows per workgroup, totals re-summed per workgroup in double)
template <bool PHILOX, bool KEEPST, bool TOTALS>
__global__ __launch_bounds__(256) void slab_kernel(const Args a) {
  __shared__ double tot[256 + 1024];
  __shared__ float ms[128], rs[128];
  const int slab = 64, c0 = blockIdx.x * slab, tpr = 16, rpi = 16, r_in = threadIdx.x / tpr, cq = threadIdx.x % tpr;
  const int per = (a.rows + gridDim.y - 1) / gridDim.y, rb = blockIdx.y * per, re = min(rb + per, a.rows);
  const int cl = cq * 4, c4 = c0 + cl, tprC = a.C >> 2;
  float v[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = rb + r_in + u * rpi, rc = max(min(r, re - 1), 0);
    const f32x4 t = *(const f32x4*)(a.x + (int64_t)rc * a.C + c4);
    v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = t[2]; v[u][3] = t[3];
  }
  const f32x4 gm = *(const f32x4*)(a.gamma + c4), bt = *(const f32x4*)(a.beta + c4);
  if (TOTALS) {
    const int nq = slab >> 1, groups = 256 / nq, t = threadIdx.x;
    {
      const int q = t % nq, g0 = t / nq, v0 = q * 4;
      const int col = v0 < slab ? c0 + v0 : a.C + c0 + (v0 - slab);
      double acc[4] = {0, 0, 0, 0};
      for (int b = g0; b < a.nblk; b += 16 * groups) {
        f32x4 f[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int bb = b + u * groups;
          f[u] = *(const f32x4*)(a.partials + (int64_t)min(bb, a.nblk - 1) * 2 * a.C + col);
          if (bb >= a.nblk) f[u] = f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += f[u][e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tot[256 + t * 4 + e] = acc[e];
    }
    __syncthreads();
    if (t < 2 * slab) {
      double sum = 0.0;
      for (int k = 0; k < groups; ++k) sum += tot[256 + (k * nq + (t >> 2)) * 4 + (t & 3)];
      tot[t] = sum;
    }
    __syncthreads();
    if (t < slab) {
      const double m = tot[t] / a.rows;
      double var = tot[slab + t] / a.rows - m * m;
      if (var < 0.0) var = 0.0;
      ms[t] = (float)m; rs[t] = (float)(1.0 / sqrt(var + 1e-5));
    }
  } else {
    if (threadIdx.x < slab) { ms[threadIdx.x] = a.mean[c0 + threadIdx.x]; rs[threadIdx.x] = a.rstd[c0 + threadIdx.x]; }
  }
  __syncthreads();
  const unsigned thr = 0x80000000u;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = rb + r_in + u * rpi;
    if (r >= re) break;
    const int64_t i = (int64_t)r * tprC + (c4 >> 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[u][e] = tanh_fast((v[u][e] - ms[cl + e]) * rs[cl + e] * gm[e] + bt[e]);
    unsigned kb;
    if (PHILOX) {
      const uint4 bb = philox(make_uint2(17u, 3u), make_uint4((unsigned)i, 40u, 5u, 0u));
      kb = (bb.x >= thr) | ((bb.y >= thr) << 1) | ((bb.z >= thr) << 2) | ((bb.w >= thr) << 3);
      if (KEEPST) a.keep[i] = (uint8_t)kb;
    } else {
      kb = a.keep[i];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[u][e] = ((kb >> e) & 1) ? v[u][e] * 2.f : 0.f;
    *(uint2*)(a.out + (int64_t)r * a.C + c4) = make_uint2(pack_bf2(v[u][0], v[u][1]), pack_bf2(v[u][2], v[u][3]));
  }
}

// ---- variant E: whole rows; a thread owns CPT = 16 consecutive channels (64 B in, 32 B out, 4 keep bytes = one dword), RPW rows per
// workgroup iteration = 256 / (C / 16); mean / rstd come finished
template <bool PHILOX, int UNR>
__global__ __launch_bounds__(256) void row_kernel(const Args a) {
  const int tpr = a.C / 16, rpi = 256 / tpr, r_in = threadIdx.x / tpr, c0 = (threadIdx.x % tpr) * 16;
  float sc[16], mn[16], sh[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { mn[e] = a.mean[c0 + e]; sc[e] = a.rstd[c0 + e] * a.gamma[c0 + e]; sh[e] = a.beta[c0 + e]; }
  const unsigned thr = 0x80000000u;
  for (int r0 = blockIdx.x * rpi * UNR; r0 < a.rows; r0 += gridDim.x * rpi * UNR) {
    f32x4 v[UNR][4];
    unsigned kw[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = min(r0 + u * rpi + r_in, a.rows - 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = *(const f32x4*)(a.x + (int64_t)r * a.C + c0 + k * 4);
      if (!PHILOX) kw[u] = *(const unsigned*)(a.keep + (int64_t)r * (a.C >> 2) + (c0 >> 2));
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = r0 + u * rpi + r_in;
      if (r >= a.rows) break;
      if (PHILOX) {
        kw[u] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint4 bb = philox(make_uint2(17u, 3u), make_uint4((unsigned)(r * (a.C >> 2) + (c0 >> 2) + k), 40u, 5u, 0u));
          kw[u] |= ((bb.x >= thr) | ((bb.y >= thr) << 1) | ((bb.z >= thr) << 2) | ((bb.w >= thr) << 3)) << (8 * k);
        }
        *(unsigned*)(a.keep + (int64_t)r * (a.C >> 2) + (c0 >> 2)) = kw[u];
      }
      float o[16];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = tanh_fast((v[u][k][e] - mn[k * 4 + e]) * sc[k * 4 + e] + sh[k * 4 + e]);
          o[k * 4 + e] = ((kw[u] >> (8 * k + e)) & 1) ? t * 2.f : 0.f;
        }
      uint4 w0 = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
      uint4 w1 = make_uint4(pack_bf2(o[8], o[9]), pack_bf2(o[10], o[11]), pack_bf2(o[12], o[13]), pack_bf2(o[14], o[15]));
      *(uint4*)(a.out + (int64_t)r * a.C + c0) = w0;
      *(uint4*)(a.out + (int64_t)r * a.C + c0 + 8) = w1;
    }
  }
}

// pure copy of the same bytes (the streaming floor): fp32 row in, bf16 row out
__global__ __launch_bounds__(256) void copy_kernel(const Args a) {
  const int64_t n8 = (int64_t)a.rows * a.C / 8;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += gridDim.x * 256ll) {
    const f32x4 p = *(const f32x4*)(a.x + i * 8), q = *(const f32x4*)(a.x + i * 8 + 4);
    *(uint4*)(a.out + i * 8) = make_uint4(pack_bf2(p[0], p[1]), pack_bf2(p[2], p[3]), pack_bf2(q[0], q[1]), pack_bf2(q[2], q[3]));
  }
}

template <class F>
float timeit(F&& f, int iters = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main() {
  const int rows = 16 * 423, C = 512, nblk = 112;
  std::vector<float> hx((size_t)rows * C), hp((size_t)nblk * 2 * C), hm(C), hr(C), hg(C), hb(C);
  srand(1);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : hp) v = rand() / (float)RAND_MAX * 60.f;
  for (int c = 0; c < C; ++c) { hm[c] = 0.1f; hr[c] = 0.9f; hg[c] = 1.1f; hb[c] = 0.05f; }
  Args a; a.rows = rows; a.C = C; a.nblk = nblk;
  float *x, *p, *m, *r, *g, *b; bf16_t* out; uint8_t* keep;
  // a second set of buffers so that consecutive launches do not find everything in the Infinity Cache... they will: the step's
  // kernels run right after the producing conv, so warm is the realistic state.  Both are reported (cold = 8 rotating buffers of 21 MB).
  const int NB = 16;
  CK(hipMalloc(&x, (size_t)NB * rows * C * 4)); CK(hipMalloc(&out, (size_t)NB * rows * C * 2)); CK(hipMalloc(&keep, (size_t)NB * rows * C / 4));
  CK(hipMalloc(&p, hp.size() * 4)); CK(hipMalloc(&m, C * 4)); CK(hipMalloc(&r, C * 4)); CK(hipMalloc(&g, C * 4)); CK(hipMalloc(&b, C * 4));
  for (int i = 0; i < NB; ++i) CK(hipMemcpy(x + (size_t)i * rows * C, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(m, hm.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(r, hr.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(keep, 0x5, (size_t)NB * rows * C / 4));
  a.partials = p; a.mean = m; a.rstd = r; a.gamma = g; a.beta = b;
  for (int rot = 0; rot < 2; ++rot) {
    int it = 0;
    auto args = [&]() { Args q = a; const int k = rot ? (it++ % NB) : 0; q.x = x + (size_t)k * rows * C; q.out = out + (size_t)k * rows * C; q.keep = keep + (size_t)k * rows * C / 4; return q; };
    printf("---- %s buffers\n", rot ? "16 rotating (336 MB: past the Infinity Cache)" : "one set of (warm)");
    const dim3 gs(8, (rows + 63) / 64);
    printf("slab: philox+keepstore+totals  %6.2f us\n", timeit([&] { hipLaunchKernelGGL((slab_kernel<true, true, true>), gs, dim3(256), 0, 0, args()); }));
    printf("slab: philox+totals, no keepst %6.2f us\n", timeit([&] { hipLaunchKernelGGL((slab_kernel<true, false, true>), gs, dim3(256), 0, 0, args()); }));
    printf("slab: keep loaded + totals     %6.2f us\n", timeit([&] { hipLaunchKernelGGL((slab_kernel<false, false, true>), gs, dim3(256), 0, 0, args()); }));
    printf("slab: philox+keepst, no totals %6.2f us\n", timeit([&] { hipLaunchKernelGGL((slab_kernel<true, true, false>), gs, dim3(256), 0, 0, args()); }));
    printf("slab: keep loaded, no totals   %6.2f us\n", timeit([&] { hipLaunchKernelGGL((slab_kernel<false, false, false>), gs, dim3(256), 0, 0, args()); }));
    for (int wg : {256, 512, 846, 1024, 2048}) {
      printf("row<philox,1> grid %4d          %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((row_kernel<true, 1>), dim3(wg), dim3(256), 0, 0, args()); }));
      printf("row<philox,2> grid %4d          %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((row_kernel<true, 2>), dim3(wg), dim3(256), 0, 0, args()); }));
      printf("row<keepld,1> grid %4d          %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((row_kernel<false, 1>), dim3(wg), dim3(256), 0, 0, args()); }));
      printf("row<keepld,2> grid %4d          %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((row_kernel<false, 2>), dim3(wg), dim3(256), 0, 0, args()); }));
      printf("row<keepld,4> grid %4d          %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL((row_kernel<false, 4>), dim3(wg), dim3(256), 0, 0, args()); }));
    }
    for (int wg : {512, 1024, 2048, 4096}) printf("copy grid %4d                   %6.2f us\n", wg, timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(wg), dim3(256), 0, 0, args()); }));
  }
  return 0;
}
