// hifigan.hip — HiFi-GAN V1 generator support kernels: weight-norm folding, weight repacking for the implicit-GEMM
// conv kernels, and the MRF average.  reference: hifi/models.py:146-210 (Generator), :12-95 (ResBlock1).
#include "common.h"
#include "conv_post.h"

namespace {

// w[r][:] = v[r][:] * g[r] / ||v[r][:]||   — torch.nn.utils.remove_weight_norm with dim=0 (hifi/models.py:203-210):
// the norm runs over every dim but 0; for ConvTranspose1d dim 0 is the IN-channel axis.  One workgroup per row.
__global__ __launch_bounds__(256) void wn_fold_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                      float* __restrict__ w, int cols) {
  __shared__ float part[4];
  const int r = blockIdx.x;
  const float* vr = v + (int64_t)r * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  const float norm = sqrtf(part[0] + part[1] + part[2] + part[3]);
  const float sc = g[r] / norm;
  float* wr = w + (int64_t)r * cols;
  for (int c = threadIdx.x; c < cols; c += 256) wr[c] = vr[c] * sc;
}

// mode 0: Conv1d weight (Cout, Cin, k) fp32 -> (Cout, k, Cin) bf16   (input channels contiguous per tap)
// mode 1: ConvTranspose1d weight (Cin, Cout, k) fp32 -> (k, Cout, Cin) bf16
template <bool F16>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int d0,
                                                          int d1, int d2, int mode, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    // i indexes dst
    int64_t s;
    if (mode == 0) {  // dst (d0, d2, d1)
      const int ci = (int)(i % d1);
      const int j = (int)((i / d1) % d2);
      const int co = (int)(i / ((int64_t)d1 * d2));
      s = ((int64_t)co * d1 + ci) * d2 + j;
    } else {          // dst (d2, d1, d0)
      const int ci = (int)(i % d0);
      const int co = (int)((i / d0) % d1);
      const int j = (int)(i / ((int64_t)d0 * d1));
      s = ((int64_t)ci * d1 + co) * d2 + j;
    }
    dst[i] = pack1<F16>(src[s]);
  }
}

// out = lrelu((a + b + c) * scale, slope), 8 elements per lane — the MRF average (hifi/models.py:190-196) with the
// consumer's LeakyReLU (hifi/models.py:188,197) fused
template <bool F16>
__global__ __launch_bounds__(256) void avg3_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                   const uint4* __restrict__ c, uint4* __restrict__ out, int64_t n8,
                                                   float scale, float slope) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 va = a[i], vb = b[i], vc = c[i];
    const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float al, ah, bl, bh, cl, ch;
      unpack2<F16>(wa[e], al, ah); unpack2<F16>(wb[e], bl, bh); unpack2<F16>(wc[e], cl, ch);
      float lo = (al + bl + cl) * scale, hi = (ah + bh + ch) * scale;
      lo = lo > 0.f ? lo : lo * slope;      // slope 1.0 = identity
      hi = hi > 0.f ? hi : hi * slope;
      o[e] = pack2<F16>(lo, hi);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// conv_post: Conv1d(C -> 1, k, pad (k-1)/2) + tanh on channels-last 16-bit activations (hifi/models.py:198-199; the
// LeakyReLU(0.01) of :197 is produced by the previous stage's epilogue).  One output sample per thread: a GEMM tile
// would spend 127/128 of its MFMA columns on padding (205 us on the 128x128 kernel); this is a 50 MB streaming read.
// A workgroup stages 256 + k - 1 frames in LDS (16-byte rows chunks, zero outside the utterance), weights as fp32.
template <bool F16>
__global__ __launch_bounds__(256) void conv_post_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w /* [k][C] */,
                                                        const float* __restrict__ bias, float* __restrict__ out, int len, int C, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int HK = (K - 1) / 2, CH8 = C >> 3, rows = 256 + K - 1;
  const int RS = C * 2 + 16;                        // padded row stride: consecutive lanes hit different banks
  float* wf = (float*)(sm + rows * RS);             // [K][C] fp32
  const int b = blockIdx.y, t0 = blockIdx.x * 256;
  const bf16_t* xb = x + (int64_t)b * len * C;
  for (int i = threadIdx.x; i < rows * CH8; i += 256) {
    const int r = i / CH8, ch = i - r * CH8;
    const int t = t0 - HK + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < len) v = *(const uint4*)(xb + (int64_t)t * C + ch * 8);
    *(uint4*)(sm + r * RS + ch * 16) = v;
  }
  for (int i = threadIdx.x; i < K * C; i += 256) wf[i] = unpack1<F16>(w[i]);
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= len) return;
  float acc = bias[0];
  for (int j = 0; j < K; ++j) {
    const unsigned char* row = sm + (threadIdx.x + j) * RS;
    const float* wj = wf + j * C;
    for (int ch = 0; ch < CH8; ++ch) {
      const uint4 v = *(const uint4*)(row + ch * 16);
      acc += conv_post_dot8<F16>(v, wj + ch * 8);
    }
  }
  out[(int64_t)b * len + t] = tanhf(acc);
}

}  // namespace

extern "C" int ttsk_weight_norm_fold(const float* v, const float* g, float* w, int rows, int cols, void* stream) {
  TTSK_REQUIRE(v && g && w && rows > 0 && cols > 0, "ttsk_weight_norm_fold: bad arguments");
  hipLaunchKernelGGL(wn_fold_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, cols);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_pack_conv_weight(const float* src, void* dst16, int f16, int d0, int d1, int d2, int mode, void* stream) {
  TTSK_REQUIRE(src && dst16 && d0 > 0 && d1 > 0 && d2 > 0 && (mode == 0 || mode == 1), "ttsk_pack_conv_weight: bad arguments");
  const int64_t n = (int64_t)d0 * d1 * d2;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (f16)
    hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst16, d0, d1, d2, mode, n);
  else
    hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst16, d0, d1, d2, mode, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_avg3(const void* a, const void* b, const void* c, void* out, int f16, int64_t n, float scale, float slope, void* stream) {
  TTSK_REQUIRE(a && b && c && out && n > 0 && (n & 7) == 0, "ttsk_avg3: n must be a positive multiple of 8");
  TTSK_REQUIRE(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)out)) & 15) == 0, "ttsk_avg3: 16-byte alignment");
  const int64_t n8 = n / 8;
  int blocks = (int)((n8 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (f16)
    hipLaunchKernelGGL(avg3_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b,
                       (const uint4*)c, (uint4*)out, n8, scale, slope);
  else
    hipLaunchKernelGGL(avg3_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b,
                       (const uint4*)c, (uint4*)out, n8, scale, slope);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_hifi_conv_post(const void* x16, const void* w16 /* (1, k, C) tap-major */, const float* bias, float* out,
                                   int f16, int B, int len, int C, int K, void* stream) {
  TTSK_REQUIRE(x16 && w16 && bias && out && B > 0 && len > 0 && B <= 65535, "ttsk_hifi_conv_post: bad arguments");
  TTSK_REQUIRE(C >= 8 && C <= 128 && (C & 7) == 0 && (K & 1) == 1 && K <= 15, "ttsk_hifi_conv_post: C in 8..128 step 8, odd K <= 15");
  const size_t shm = (size_t)(256 + K - 1) * (C * 2 + 16) + (size_t)K * C * 4;
  dim3 grid((len + 255) / 256, B);
  if (f16)
    hipLaunchKernelGGL(conv_post_kernel<true>, grid, dim3(256), shm, (hipStream_t)stream, (const bf16_t*)x16, (const bf16_t*)w16, bias, out, len, C, K);
  else
    hipLaunchKernelGGL(conv_post_kernel<false>, grid, dim3(256), shm, (hipStream_t)stream, (const bf16_t*)x16, (const bf16_t*)w16, bias, out, len, C, K);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
