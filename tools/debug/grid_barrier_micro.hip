// grid_barrier_micro.hip — diagnostic only: what does an in-kernel grid barrier cost on this chip for a one-round launch (224 workgroups x 512
// threads, one per CU), with the data exchange a fused conv + BatchNorm epilogue would need around it: every workgroup publishes a partial row
// (512 floats), all arrive at a counter, every workgroup then reads all partial rows of its channel group (112 rows x 512 floats).  The spin is
// BOUNDED (a workgroup that waits longer than ~20 ms sets an error flag and goes on): a mis-sized launch ends wrong, not hung.
// build: hipcc -O3 --offload-arch=gfx950 -o grid_barrier_micro.out grid_barrier_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: empty body; 1: publish + barrier; 2: publish + barrier + read the partial rows
__global__ __launch_bounds__(512, 1) void k(float* partials, unsigned* ctr, unsigned target, float* out, unsigned* err, int nrows) {
  __shared__ float red[512];
  const int tid = threadIdx.x, wg = blockIdx.x;
  float v = (float)(wg * 512 + tid);
  if (MODE >= 1) {
    partials[(size_t)wg * 512 + tid] = v;
    __syncthreads();
    if (tid == 0) {
      __threadfence();                     // (one fence per workgroup, behind the barrier: the row is visible device-wide before the arrival)
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 8000000u) { atomicExch(err, 1u); break; }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
  if (MODE >= 2) {
    float s = 0.f;
    // rows of the same channel group: every other workgroup (two groups), 16 loads in flight
    for (int r0 = (wg & 1); r0 < nrows; r0 += 32) {
      float f[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int r = r0 + 2 * u; f[u] = r < nrows ? __builtin_nontemporal_load(partials + (size_t)r * 512 + tid) : 0.f; }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += f[u];
    }
    v = s;
  }
  red[tid] = v;
  __syncthreads();
  if (tid == 0) out[wg] = red[0] + red[511];
}

int main() {
  const int NWG = 224;
  float *partials, *out; unsigned *ctr, *err;
  CK(hipMalloc(&partials, NWG * 512 * 4)); CK(hipMalloc(&out, NWG * 4)); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned epoch = 0;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      const int N = 200;
      CK(hipEventRecord(e0));
      for (int i = 0; i < N; ++i) {
        ++epoch;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NWG), dim3(512), 0, 0, partials, ctr, epoch * NWG, out, err, NWG);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NWG), dim3(512), 0, 0, partials, ctr, epoch * NWG, out, err, NWG);
        else hipLaunchKernelGGL(k<2>, dim3(NWG), dim3(512), 0, 0, partials, ctr, epoch * NWG, out, err, NWG);
        if (mode == 0) { /* keep the counter in step with the epochs */ }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (mode == 0) { /* the empty kernels did not arrive: add what they would have */ unsigned add = (unsigned)N * NWG; unsigned h; CK(hipMemcpy(&h, ctr, 4, hipMemcpyDeviceToHost)); h += add; CK(hipMemcpy(ctr, &h, 4, hipMemcpyHostToDevice)); }
      unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("mode %d (%s): %.2f us per launch back to back, timeout flag %u\n", mode, mode == 0 ? "empty" : mode == 1 ? "publish + grid barrier" : "publish + barrier + read 112 rows", 1e3 * ms / N, herr);
    }
  }
  float h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("out[0..1] = %.1f %.1f\n", h[0], h[1]);
  return 0;
}
