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
#include <type_traits>
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

// The pair kernel's c1 tile: NFR fragments (2 MFMAs each) from a ring of DR, then the epilogue (bias, LeakyReLU, pack, one ds_write_b128), tile after tile.
// EPI: 0 none, 1 as the kernel has it (behind the tile's MFMAs), 2 software-pipelined (the previous tile's epilogue inside this tile's fragment loop)
template <int WAVES, int NFR, int DR, int EPI>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void tile_kernel(const half8* __restrict__ wsrc, float* __restrict__ out, unsigned long long* __restrict__ cyc, int nit, float slope) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 96 * 1024 / 16; i += WAVES * 64) ((uint4*)smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  half8 w[NFR][2];
#pragma unroll
  for (int i = 0; i < NFR; ++i) { w[i][0] = wsrc[((2 * i) & 15) * 64 + lane]; w[i][1] = wsrc[((2 * i + 1) & 15) * 64 + lane]; }
  f32x4 bv[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.5f, 0.6f, 0.7f, 0.8f}};
  __syncthreads();
  const unsigned char* base = smem + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  unsigned char* tw = smem + 48 * 1024 + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  half8 ring[DR];
#pragma unroll
  for (int j = 0; j < DR; ++j) ring[j] = *(const half8*)(base + (j >> 1) * 480 + (j & 1) * 64);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 pacc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < nit; ++it) {
    const unsigned char* b2 = base + (it & 1) * 2560;
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      const half8 B = ring[f % DR];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][0], B, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][1], B, acc[1], 0, 0, 0);
      const int nf = (f + DR) % NFR;
      ring[f % DR] = *(const half8*)(b2 + (nf >> 1) * 480 + (nf & 1) * 64);
    }
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (EPI == 2) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // three VALU of the previous tile's epilogue
    }
    if (EPI == 1 || EPI == 2) {
      f32x4 (&src)[2] = EPI == 2 ? pacc : acc;
      unsigned o[4];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        f32x4 v = src[cc] + bv[cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 a; a[0] = (_Float16)v[0]; a[1] = (_Float16)v[1];
        h2 b; b[0] = (_Float16)v[2]; b[1] = (_Float16)v[3];
        o[cc * 2] = __builtin_bit_cast(unsigned, a); o[cc * 2 + 1] = __builtin_bit_cast(unsigned, b);
      }
      *(uint4*)(tw + (it & 7) * 2560) = make_uint4(o[0], o[1], o[2], o[3]);
      if (EPI == 2) { pacc[0] = acc[0]; pacc[1] = acc[1]; }
    } else {
      pacc[0] += acc[0]; pacc[1] += acc[1];
    }
    if (EPI != 2) __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = pacc[0] + pacc[1];
  out[(blockIdx.x * WAVES * 64 + tid)] = s[0] + s[1] + s[2] + s[3] + (float)smem[48 * 1024 + tid];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}
// The same tile with the MFMAs as asm (accumulator tied: dst == srcC, registers fixed) and the ring's reads volatile: program order IS the schedule
template <int WAVES, int NFR, int DR, int EPI>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void tile_asm_kernel(const half8* __restrict__ wsrc, float* __restrict__ out, unsigned long long* __restrict__ cyc, int nit, float slope) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 96 * 1024 / 16; i += WAVES * 64) ((uint4*)smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  half8 w[NFR][2];
#pragma unroll
  for (int i = 0; i < NFR; ++i) { w[i][0] = wsrc[((2 * i) & 15) * 64 + lane]; w[i][1] = wsrc[((2 * i + 1) & 15) * 64 + lane]; }
  f32x4 bv[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.5f, 0.6f, 0.7f, 0.8f}};
  __syncthreads();
  const unsigned char* base = smem + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  unsigned char* tw = smem + 48 * 1024 + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  half8 ring[DR];
#pragma unroll
  for (int j = 0; j < DR; ++j) ring[j] = *(const volatile half8*)(base + (j >> 1) * 480 + (j & 1) * 64);
  f32x4 sum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  static_assert(NFR % DR == 0 || (2 * NFR) % DR == 0, "");
#pragma unroll 1
  for (int it = 0; it < nit; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int ROT = (half * NFR) % DR;
      const unsigned char* b2 = base + half * 2560;
      const unsigned char* nx = base + (1 - half) * 2560;
      f32x4 acc0, acc1;
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const half8 B = ring[(f + ROT) % DR];
        if (f == 0) {
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc0) : "v"(w[f][0]), "v"(B));
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc1) : "v"(w[f][1]), "v"(B));
        } else {
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(w[f][0]), "v"(B));
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(w[f][1]), "v"(B));
        }
        const int nf = f + DR;
        ring[(f + ROT) % DR] = nf < NFR ? *(const volatile half8*)(b2 + (nf >> 1) * 480 + (nf & 1) * 64)
                                        : *(const volatile half8*)(nx + ((nf - NFR) >> 1) * 480 + ((nf - NFR) & 1) * 64);
      }
      if (EPI == 1) {
        unsigned o[4];
        f32x4 src[2] = {acc0, acc1};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          f32x4 v = src[cc] + bv[cc];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          h2 a; a[0] = (_Float16)v[0]; a[1] = (_Float16)v[1];
          h2 b; b[0] = (_Float16)v[2]; b[1] = (_Float16)v[3];
          o[cc * 2] = __builtin_bit_cast(unsigned, a); o[cc * 2 + 1] = __builtin_bit_cast(unsigned, b);
        }
        *(uint4*)(tw + ((it + half) & 7) * 2560) = make_uint4(o[0], o[1], o[2], o[3]);
      } else {
        sum[0] += acc0; sum[1] += acc1;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = sum[0] + sum[1];
  out[(blockIdx.x * WAVES * 64 + tid)] = s[0] + s[1] + s[2] + s[3] + (float)smem[48 * 1024 + tid];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}
template <int WAVES, int NFR, int DR, int EPI>
void run_tile_asm(const char* name, const half8* w, float* out, unsigned long long* cyc, int nit) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((tile_asm_kernel<WAVES, NFR, DR, EPI>), dim3(256), dim3(WAVES * 64), 0, 0, w, out, cyc, nit, 0.1f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * WAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-64s  %7.0f cycles per tile per wave (%d MFMAs = %d cycles of MFMA), %5.1f per MFMA per SIMD\n", name, med / nit, 2 * NFR, 32 * NFR, med / nit / (2 * NFR) / (WAVES / 4.0));
}

// pairws.hip's conv_tile verbatim (ring with rotation, next tile's fragments requested at the tail, sched_group_barrier pipeline), tile after tile
template <int WAVES, int NFR, int DR, int EPI>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void tile2_kernel(const half8* __restrict__ wsrc, float* __restrict__ out, unsigned long long* __restrict__ cyc, int nit, float slope) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 96 * 1024 / 16; i += WAVES * 64) ((uint4*)smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  half8 w[NFR][2];
#pragma unroll
  for (int i = 0; i < NFR; ++i) { w[i][0] = wsrc[((2 * i) & 15) * 64 + lane]; w[i][1] = wsrc[((2 * i + 1) & 15) * 64 + lane]; }
  f32x4 bv[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.5f, 0.6f, 0.7f, 0.8f}};
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const unsigned char* base = smem + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  unsigned char* tw = smem + 48 * 1024 + (lane & 15) * 160 + (lane >> 4) * 16 + (wave & 3) * 8192;
  half8 ring[DR];
  constexpr int ROT1 = NFR % DR;
  auto conv_tile = [&](auto rotc, f32x4 (&acc)[2], const unsigned char* b, const unsigned char* nxt) __attribute__((always_inline)) {
    constexpr int ROT = decltype(rotc)::value;
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      const half8 Bf = ring[(f + ROT) % DR];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][0], Bf, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f][1], Bf, acc[1], 0, 0, 0);
      const int nf = f + DR;
      ring[(f + ROT) % DR] = nf < NFR ? *(const half8*)(b + (nf >> 1) * 480 + (nf & 1) * 64) : *(const half8*)(nxt + ((nf - NFR) >> 1) * 480 + ((nf - NFR) & 1) * 64);
    }
#pragma unroll
    for (int f = 0; f < NFR; ++f) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  auto epilogue = [&](f32x4 (&acc)[2], int slot) __attribute__((always_inline)) {
    unsigned o[4];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      f32x4 v = acc[cc] + bv[cc];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      h2 a; a[0] = (_Float16)v[0]; a[1] = (_Float16)v[1];
      h2 b; b[0] = (_Float16)v[2]; b[1] = (_Float16)v[3];
      o[cc * 2] = __builtin_bit_cast(unsigned, a); o[cc * 2 + 1] = __builtin_bit_cast(unsigned, b);
    }
    *(uint4*)(tw + slot * 2560) = make_uint4(o[0], o[1], o[2], o[3]);
  };
#pragma unroll
  for (int j = 0; j < DR; ++j) ring[j] = *(const half8*)(base + (j >> 1) * 480 + (j & 1) * 64);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 sum[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < nit; it += 2) {
    f32x4 acc[2];
    conv_tile(std::integral_constant<int, 0>{}, acc, base, base + 2560);
    if (EPI) epilogue(acc, it & 7); else { sum[0] += acc[0]; sum[1] += acc[1]; }
    conv_tile(std::integral_constant<int, ROT1>{}, acc, base + 2560, base);
    if (EPI) epilogue(acc, (it + 1) & 7); else { sum[0] += acc[0]; sum[1] += acc[1]; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = sum[0] + sum[1];
  out[(blockIdx.x * WAVES * 64 + tid)] = s[0] + s[1] + s[2] + s[3] + (float)smem[48 * 1024 + tid];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}
template <int WAVES, int NFR, int DR, int EPI>
void run_tile2(const char* name, const half8* w, float* out, unsigned long long* cyc, int nit) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((tile2_kernel<WAVES, NFR, DR, EPI>), dim3(256), dim3(WAVES * 64), 0, 0, w, out, cyc, nit, 0.1f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * WAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-64s  %7.0f cycles per tile per wave (%d MFMAs = %d cycles of MFMA), %5.1f per MFMA per SIMD\n", name, med / nit, 2 * NFR, 32 * NFR, med / nit / (2 * NFR) / (WAVES / 4.0));
}

template <int WAVES, int NFR, int DR, int EPI>
void run_tile(const char* name, const half8* w, float* out, unsigned long long* cyc, int nit) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((tile_kernel<WAVES, NFR, DR, EPI>), dim3(256), dim3(WAVES * 64), 0, 0, w, out, cyc, nit, 0.1f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * WAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  printf("%-64s  %7.0f cycles per tile per wave (%d MFMAs = %d cycles of MFMA), %5.1f per MFMA per SIMD\n", name, med / nit, 2 * NFR, 32 * NFR, med / nit / (2 * NFR) / (WAVES / 4.0));
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
  run_tile2<4, 14, 4, 0>("pairws tile k=7: no epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile2<4, 14, 4, 1>("pairws tile k=7: epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile2<8, 14, 4, 0>("pairws tile k=7: no epilogue, ring 4, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile2<8, 14, 4, 1>("pairws tile k=7: epilogue, ring 4, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile2<4, 6, 6, 1>("pairws tile k=3: epilogue, ring 6, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile2<8, 6, 6, 1>("pairws tile k=3: epilogue, ring 6, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile2<4, 22, 4, 1>("pairws tile k=11: epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile2<8, 22, 4, 1>("pairws tile k=11: epilogue, ring 4, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile_asm<4, 14, 4, 0>("asm tile k=7: no epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile_asm<4, 14, 4, 1>("asm tile k=7: epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile_asm<4, 14, 7, 1>("asm tile k=7: epilogue, ring 7, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile_asm<8, 14, 4, 1>("asm tile k=7: epilogue, ring 4, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile_asm<8, 14, 7, 1>("asm tile k=7: epilogue, ring 7, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile_asm<4, 6, 6, 1>("asm tile k=3: epilogue, ring 6, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile_asm<8, 6, 6, 1>("asm tile k=3: epilogue, ring 6, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile_asm<4, 22, 4, 1>("asm tile k=11: epilogue, ring 4, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile_asm<8, 22, 4, 1>("asm tile k=11: epilogue, ring 4, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile<4, 14, 4, 0>("tile k=7: no epilogue, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile<4, 14, 4, 1>("tile k=7: epilogue behind the MFMAs, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile<4, 14, 4, 2>("tile k=7: epilogue pipelined into the next tile, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile<8, 14, 4, 1>("tile k=7: epilogue behind the MFMAs, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile<8, 14, 4, 2>("tile k=7: epilogue pipelined, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile<4, 6, 6, 1>("tile k=3: epilogue behind the MFMAs, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile<4, 6, 6, 2>("tile k=3: epilogue pipelined, 1 wave/SIMD", w, out, cyc, 1000);
  run_tile<8, 6, 6, 1>("tile k=3: epilogue behind the MFMAs, 2 waves/SIMD", w, out, cyc, 1000);
  run_tile<8, 6, 6, 2>("tile k=3: epilogue pipelined, 2 waves/SIMD", w, out, cyc, 1000);
  run_pp<4, true>("asm, dst == srcC, 2 chains, 1 wave/SIMD", w, out, cyc, nit);
  run_pp<4, false>("asm, dst != srcC (ping-pong), 2 chains, 1 wave/SIMD", w, out, cyc, nit);
  run_pp<8, true>("asm, dst == srcC, 2 chains, 2 waves/SIMD", w, out, cyc, nit);
  run_pp<8, false>("asm, dst != srcC (ping-pong), 2 chains, 2 waves/SIMD", w, out, cyc, nit);
  return 0;
}
