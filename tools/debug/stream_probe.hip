// stream_probe.hip — diagnostic only (not part of the library): an Adam-shaped HBM stream (read p, g, m, v; write p, m, v; nontemporal) with
// a chosen grid, to be run on a second stream beside a phase of the replayed step: what does a deferred optimizer slice cost the
// latency-bound encoder chain?   build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o stream_probe.so stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void probe_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n4) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = __builtin_nontemporal_load((f32x4*)(p + i * 4)), gg = __builtin_nontemporal_load((const f32x4*)(g + i * 4)),
          mm = __builtin_nontemporal_load((f32x4*)(m + i * 4)), vv = __builtin_nontemporal_load((f32x4*)(v + i * 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = 0.9f * mm[e] + 0.1f * gg[e];
      vv[e] = 0.98f * vv[e] + 0.02f * gg[e] * gg[e];
      pp[e] -= 1e-9f * mm[e] / (sqrtf(vv[e]) + 1e-9f);
    }
    __builtin_nontemporal_store(pp, (f32x4*)(p + i * 4));
    __builtin_nontemporal_store(mm, (f32x4*)(m + i * 4));
    __builtin_nontemporal_store(vv, (f32x4*)(v + i * 4));
  }
}

extern "C" int probe_launch(float* p, const float* g, float* m, float* v, long long n, int grid, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (n & 3) || grid <= 0 || grid > 65535) return 1;
  hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (int64_t)(n >> 2));
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
