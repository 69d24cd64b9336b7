// mfma_rate_micro.hip — diagnostic only (not part of the library): what rate does v_mfma_f32_16x16x32_f16 sustain on gfx950 from the
// instruction streams the window / pair kernels use?  One workgroup per CU, WAVES waves; every wave runs NIT iterations of a body of 32 MFMAs:
//   variant 0: operands in registers only, NACC accumulators in rotation
//   variant 1: + one conflict-free ds_read_b128 per two MFMAs through a ring of DR registers (the pair kernels' B fragments)
// Prints cycles (s_memtime) per MFMA per SIMD.   build + run: hipcc -O3 --offload-arch=gfx950 -o mfma_rate_micro mfma_rate_micro.hip && ./mfma_rate_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int WAVES, int NACC, int DR, bool LDS, int MPR>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void rate_kernel(const half8* __restrict__ wsrc, float* __restrict__ out, unsigned long long* __restrict__ cyc, int nit) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * 1024 / 16; i += WAVES * 64) ((uint4*)smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  half8 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = wsrc[i * 64 + lane];
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* base = smem + (lane & 15) * 160 + (lane >> 4) * 16 + wave * 4096;
  half8 ring[DR > 0 ? DR : 1];
#pragma unroll
  for (int j = 0; j < DR; ++j) ring[j] = *(const half8*)(base + j * 160);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < nit; ++it) {
#pragma unroll
    for (int f = 0; f < 32 / MPR; ++f) {
      half8 B = LDS ? ring[f % (DR > 0 ? DR : 1)] : w[(f + 5) & 15];
#pragma unroll
      for (int m = 0; m < MPR; ++m) {
        const int idx = f * MPR + m;
        acc[idx % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[idx & 15], B, acc[idx % NACC], 0, 0, 0);
      }
      if (LDS) ring[f % (DR > 0 ? DR : 1)] = *(const half8*)(base + ((f + DR) & 15) * 160);
    }
    if (LDS) {
#pragma unroll
      for (int f = 0; f < 32 / MPR; ++f) {
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[(blockIdx.x * WAVES * 64 + tid)] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

// dst != srcC: each accumulation chain ping-pongs between two register quads (what hipcc's allocator does when it reuses a dead B register)
template <int WAVES, bool TIED>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void pingpong_kernel(const half8* __restrict__ wsrc, float* __restrict__ out, unsigned long long* __restrict__ cyc, int nit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  half8 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = wsrc[i * 64 + lane];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < nit; ++it) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      if (TIED) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(w[f]), "v"(w[f + 8]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(w[f + 1]), "v"(w[f + 8]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(w[f]), "v"(w[f + 7]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(w[f + 1]), "v"(w[f + 7]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(b0) : "v"(a0), "v"(w[f]), "v"(w[f + 8]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(b1) : "v"(a1), "v"(w[f + 1]), "v"(w[f + 8]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(a0) : "v"(b0), "v"(w[f]), "v"(w[f + 7]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(a1) : "v"(b1), "v"(w[f + 1]), "v"(w[f + 7]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = a0 + a1 + b0 + b1;
  out[(blockIdx.x * WAVES * 64 + tid)] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}
template <int WAVES, bool TIED>
void run_pp(const char* name, const half8* w, float* out, unsigned long long* cyc, int nit) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((pingpong_kernel<WAVES, TIED>), dim3(256), dim3(WAVES * 64), 0, 0, w, out, cyc, nit);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * WAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-58s  %6.1f cycles per MFMA per wave, %5.1f per MFMA per SIMD\n", name, med / (nit * 32.0), med / (nit * 32.0) / (WAVES / 4.0));
}

template <int WAVES, int NACC, int DR, bool LDS, int MPR>
void run(const char* name, const half8* w, float* out, unsigned long long* cyc, int nit) {
  const int grid = 256;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((rate_kernel<WAVES, NACC, DR, LDS, MPR>), dim3(grid), dim3(WAVES * 64), 0, 0, w, out, cyc, nit);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<WAVES, NACC, DR, LDS, MPR>), dim3(grid), dim3(WAVES * 64), 0, 0, w, out, cyc, nit);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * WAVES);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2], per_simd = WAVES / 4.0;
  const double mf = (double)nit * 32;
  printf("%-58s  %6.1f cycles per MFMA per wave, %5.1f per MFMA per SIMD   (%.0f us, %.0f TFLOP/s)\n", name, med / mf, med / mf / per_simd, ms * 1e3,
         256.0 * WAVES * mf * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  half8* w; float* out; unsigned long long* cyc;
  hipMalloc(&w, 16 * 64 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<unsigned short> hw(16 * 64 * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x2c00 + (i * 2654435761u >> 22));     // random-ish small fp16 values
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int nit = 2000;
  run<4, 1, 0, false, 2>("regs only, 1 wave/SIMD, 1 accumulator", w, out, cyc, nit);
  run<4, 2, 0, false, 2>("regs only, 1 wave/SIMD, 2 accumulators", w, out, cyc, nit);
  run<4, 4, 0, false, 2>("regs only, 1 wave/SIMD, 4 accumulators", w, out, cyc, nit);
  run<8, 2, 0, false, 2>("regs only, 2 waves/SIMD, 2 accumulators", w, out, cyc, nit);
  run<4, 2, 4, true, 2>("LDS read per 2 MFMAs, ring 4, 1 wave/SIMD, 2 acc", w, out, cyc, nit);
  run<4, 2, 8, true, 2>("LDS read per 2 MFMAs, ring 8, 1 wave/SIMD, 2 acc", w, out, cyc, nit);
  run<4, 4, 8, true, 2>("LDS read per 2 MFMAs, ring 8, 1 wave/SIMD, 4 acc", w, out, cyc, nit);
  run<8, 2, 4, true, 2>("LDS read per 2 MFMAs, ring 4, 2 waves/SIMD, 2 acc", w, out, cyc, nit);
  run<8, 2, 8, true, 2>("LDS read per 2 MFMAs, ring 8, 2 waves/SIMD, 2 acc", w, out, cyc, nit);
  run<8, 4, 8, true, 2>("LDS read per 2 MFMAs, ring 8, 2 waves/SIMD, 4 acc", w, out, cyc, nit);
  run<4, 4, 4, true, 4>("LDS read per 4 MFMAs, ring 4, 1 wave/SIMD, 4 acc", w, out, cyc, nit);
  run<8, 4, 4, true, 4>("LDS read per 4 MFMAs, ring 4, 2 waves/SIMD, 4 acc", w, out, cyc, nit);
  run<4, 2, 4, true, 1>("LDS read per MFMA, ring 4, 1 wave/SIMD, 2 acc", w, out, cyc, nit);
  run<8, 2, 4, true, 1>("LDS read per MFMA, ring 4, 2 waves/SIMD, 2 acc", w, out, cyc, nit);
  run_pp<4, true>("asm, dst == srcC, 2 chains, 1 wave/SIMD", w, out, cyc, nit);
  run_pp<4, false>("asm, dst != srcC (ping-pong), 2 chains, 1 wave/SIMD", w, out, cyc, nit);
  run_pp<8, true>("asm, dst == srcC, 2 chains, 2 waves/SIMD", w, out, cyc, nit);
  run_pp<8, false>("asm, dst != srcC (ping-pong), 2 chains, 2 waves/SIMD", w, out, cyc, nit);
  return 0;
}
