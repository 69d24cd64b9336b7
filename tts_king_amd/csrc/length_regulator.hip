// length_regulator.hip — LengthRegulator expand as a wavefront prefix scan + coalesced row gather.
// reference: fs_two/model/modules.py:220-252, fs_two/utils/tools.py:369-387 (see include/ttsk.h).
#include "common.h"

namespace {

constexpr int LR_ROWS = 32;     // output frames per workgroup
constexpr int LR_MAXL = 2048;   // phonemes per utterance held in LDS

__device__ __forceinline__ int load_dur(const void* dur, int dtype, int64_t i) {
  if (dtype == 0) { const long long v = ((const long long*)dur)[i]; return v > 0 ? (v > 0x3fffffff ? 0x3fffffff : (int)v) : 0; }
  if (dtype == 2) { const int v = ((const int*)dur)[i]; return v > 0 ? v : 0; }
  const float f = ((const float*)dur)[i];
  return f > 0.f ? (int)truncf(fminf(f, 1.0e9f)) : 0;   // int() truncates toward zero, negatives clamp to 0
}

// inclusive scan of one value per lane across the 64-lane wavefront
__device__ __forceinline__ int wave_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

__global__ __launch_bounds__(256) void lr_fwd_kernel(const bf16_t* __restrict__ x, const void* __restrict__ dur, int dtype,
                                                     const float* __restrict__ pe, bf16_t* __restrict__ out,
                                                     int* __restrict__ idx_out, int* __restrict__ cs_out,
                                                     long long* __restrict__ mel_len, int L, int T, int D) {
  __shared__ int cs[LR_MAXL];
  __shared__ int ridx[LR_ROWS];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {  // wave 0: prefix scan, 64 phonemes per step, running carry
    int carry = 0;
    for (int base = 0; base < L; base += 64) {
      const int i = base + lane;
      const int v = i < L ? load_dur(dur, dtype, (int64_t)b * L + i) : 0;
      const int s = wave_scan(v, lane) + carry;
      if (i < L) cs[i] = s;
      carry = __shfl(s, 63, 64);
    }
  }
  __syncthreads();
  const int total = cs[L - 1];
  if (blockIdx.x == 0) {
    if (tid == 0) mel_len[b] = total;
    if (cs_out) for (int i = tid; i < L; i += 256) cs_out[(int64_t)b * L + i] = cs[i];
  }
  const int t0 = blockIdx.x * LR_ROWS;
  if (tid < LR_ROWS) {
    const int t = t0 + tid;
    int r = -1;
    if (t < T && t < total) {  // r = #{i : cs[i] <= t}  (upper bound)
      int lo = 0, hi = L;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (cs[mid] <= t) lo = mid + 1; else hi = mid; }
      r = lo;
    }
    ridx[tid] = r;
    if (idx_out && t < T) idx_out[(int64_t)b * T + t] = r;
  }
  __syncthreads();
  const int cpr = D >> 3;  // 16-byte chunks per row
  for (int c = tid; c < LR_ROWS * cpr; c += 256) {
    const int row = c / cpr, ch = c - row * cpr;
    const int t = t0 + row;
    if (t >= T) break;
    const int r = ridx[row];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r >= 0) v = *(const uint4*)(x + ((int64_t)b * L + r) * D + ch * 8);
    if (pe) {
      const f32x4 p0 = *(const f32x4*)(pe + (int64_t)t * D + ch * 8);
      const f32x4 p1 = *(const f32x4*)(pe + (int64_t)t * D + ch * 8 + 4);
      unsigned* w = (unsigned*)&v;
      w[0] = pack_bf2(__uint_as_float(w[0] << 16) + p0[0], __uint_as_float(w[0] & 0xFFFF0000u) + p0[1]);
      w[1] = pack_bf2(__uint_as_float(w[1] << 16) + p0[2], __uint_as_float(w[1] & 0xFFFF0000u) + p0[3]);
      w[2] = pack_bf2(__uint_as_float(w[2] << 16) + p1[0], __uint_as_float(w[2] & 0xFFFF0000u) + p1[1]);
      w[3] = pack_bf2(__uint_as_float(w[3] << 16) + p1[2], __uint_as_float(w[3] & 0xFFFF0000u) + p1[3]);
    }
    *(uint4*)(out + ((int64_t)b * T + t) * D + ch * 8) = v;
  }
}

// one wavefront per (utterance, phoneme): sums the frames [cs[i-1], min(cs[i], T)) of dout
__global__ __launch_bounds__(64) void lr_bwd_kernel(const bf16_t* __restrict__ dout, const int* __restrict__ cs,
                                                    bf16_t* __restrict__ dx, int L, int T, int D) {
  const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int beg = i > 0 ? cs[(int64_t)b * L + i - 1] : 0;
  int end = cs[(int64_t)b * L + i];
  if (end > T) end = T;
  for (int c = lane * 4; c < D; c += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int t = beg; t < end; ++t) {
      const uint2 v = *(const uint2*)(dout + ((int64_t)b * T + t) * D + c);
      a0 += __uint_as_float(v.x << 16); a1 += __uint_as_float(v.x & 0xFFFF0000u);
      a2 += __uint_as_float(v.y << 16); a3 += __uint_as_float(v.y & 0xFFFF0000u);
    }
    *(uint2*)(dx + ((int64_t)b * L + i) * D + c) = make_uint2(pack_bf2(a0, a1), pack_bf2(a2, a3));
  }
}

}  // namespace

extern "C" int ttsk_length_regulator_fwd(const void* x, const void* dur, int dur_dtype, const float* pe, void* out,
                                         int32_t* idx_out, int32_t* cumsum_out, int64_t* mel_len, int B, int L, int T,
                                         int D, void* stream) {
  TTSK_REQUIRE(x && dur && out && mel_len, "length_regulator_fwd: null pointer");
  TTSK_REQUIRE(B > 0 && L > 0 && L <= LR_MAXL && T > 0, "length_regulator_fwd: bad sizes B=%d L=%d T=%d", B, L, T);
  TTSK_REQUIRE(D > 0 && (D & 7) == 0, "length_regulator_fwd: D must be a multiple of 8");
  TTSK_REQUIRE(dur_dtype >= 0 && dur_dtype <= 2, "length_regulator_fwd: dur_dtype");
  dim3 grid((T + LR_ROWS - 1) / LR_ROWS, B);
  hipLaunchKernelGGL(lr_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, dur, dur_dtype, pe,
                     (bf16_t*)out, idx_out, cumsum_out, (long long*)mel_len, L, T, D);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_length_regulator_bwd(const void* dout, const int32_t* cumsum, void* dx, int B, int L, int T, int D,
                                         void* stream) {
  TTSK_REQUIRE(dout && cumsum && dx, "length_regulator_bwd: null pointer");
  TTSK_REQUIRE(B > 0 && L > 0 && T > 0 && D > 0 && (D & 3) == 0, "length_regulator_bwd: bad sizes");
  hipLaunchKernelGGL(lr_bwd_kernel, dim3(L, B), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dout, cumsum,
                     (bf16_t*)dx, L, T, D);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
