// Does a kernel that leaves its output dirty in L2 pay for the write-back at its end?  A chain of dependent (producer, consumer) launches:
// the producer's workgroups (one per CU, 256 of them) spin ~8 us, then store 64 KiB each (16 MiB per launch) plain / nt / sc1 (write-
// through); the consumer reads one line per workgroup.  Prints us per producer launch by HIP events over the chain.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/debug/wt_store_micro.hip -o /tmp/wt && /tmp/wt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void producer(u32x4* out, int spin, int per_wg16) {
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
  u32x4* dst = out + (size_t)blockIdx.x * per_wg16;
  const u32x4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
  for (int i = threadIdx.x; i < per_wg16; i += 256) {
    if (MODE == 0) dst[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, dst + i);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + i), "v"(v) : "memory");
  }
}
__global__ __launch_bounds__(256) void consumer(const u32x4* in, unsigned* sink, int per_wg16) {
  const u32x4 v = in[(size_t)blockIdx.x * per_wg16 + threadIdx.x];
  if (v.x == 0xdeadbeefu) sink[0] = v.y;
}
template <int MODE>
float run(u32x4* buf, unsigned* sink, int spin, int per_wg16, int wgs, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) { producer<MODE><<<wgs, 256>>>(buf, spin, per_wg16); consumer<<<wgs, 256>>>(buf, sink, per_wg16); }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) { producer<MODE><<<wgs, 256>>>(buf, spin, per_wg16); consumer<<<wgs, 256>>>(buf, sink, per_wg16); }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / reps;
}
int main() {
  const int wgs = 256;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, (size_t)wgs * (1 << 20));
  hipMalloc(&sink, 4);
  for (int kb : {0, 16, 64, 256}) {
    const int per_wg16 = kb * 1024 / 16;
    for (int spin : {0, 40}) {
      printf("%4d KiB per workgroup (%5.1f MiB per launch), spin %2d: plain %6.2f  nt %6.2f  sc1 %6.2f  sc0sc1 %6.2f us per (producer + consumer)\n", kb,
             wgs * kb / 1024.0, spin, run<0>(buf, sink, spin, per_wg16 ? per_wg16 : 0, wgs, 200), run<1>(buf, sink, spin, per_wg16, wgs, 200),
             run<2>(buf, sink, spin, per_wg16, wgs, 200), run<3>(buf, sink, spin, per_wg16, wgs, 200));
    }
  }
  return 0;
}
