// audio.hip — mel-spectrogram extraction: the two kernels either side of the STFT contraction.
// reference: hifi/meldataset.py:49-74 (mel_spectrogram), fs_two/audio/stft.py:57-90 (STFT.transform), :174-193
// (TacotronSTFT.mel_spectrogram).
//
// The STFT itself is the reference's own formulation — a strided conv with a windowed Fourier basis (stft.py:25-50,
// 77-84) — run on ttsk_gemm as an implicit GEMM: the padded signal is viewed as rows of `hop` samples ("hop blocks"),
// frame t is hop blocks t .. t + n_fft/hop - 1, so the conv is n_fft/hop taps over rows with `hop` channels and
// 2*(n_fft/2+1) output channels.  To keep fp32-level accuracy on 16-bit MFMA operands both the signal and the basis
// are split into fp16 high and low parts (x = hi + lo, 22 significant bits) and the three products hi*hi + hi*lo + lo*hi
// are one contraction over 3*hop channels ([hi | hi | lo] rows against a [hi | lo | hi] basis; lo*lo is below fp32
// rounding), accumulated in fp32.  Operands are pre-scaled by powers of two so that the low parts stay normal fp16.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

// wav [B][len] fp32 -> fp16 rows [B][rows][3*hop] = [hi | hi | lo] of the hop block: reflect padding by `pad` on both
// sides (torch F.pad mode='reflect'), zeros past the padded length, value * scale split into hi + lo.  Against a basis
// laid out [hi | lo | hi] per tap the three products hi*hi + hi*lo + lo*hi are ONE contraction of 3*hop channels.
// 8 samples per thread.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wav, uint4* __restrict__ out, int len, int pad,
                                                          int rows, int hop8, float scale) {
  const int b = blockIdx.y;
  const float* w = wav + (int64_t)b * len;
  const int64_t padded = (int64_t)len + 2 * pad;
  const int64_t n8 = (int64_t)rows * hop8;
  for (int64_t c = blockIdx.x * 256ll + threadIdx.x; c < n8; c += (int64_t)gridDim.x * 256) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t i = c * 8 + e * 2 + q;
        float x = 0.f;
        if (i < padded) {
          int64_t j = i - pad;
          if (j < 0) j = -j;
          if (j >= len) j = 2 * ((int64_t)len - 1) - j;
          x = w[j] * scale;
        }
        v[q] = x;
      }
      const __half h0 = __float2half_rn(v[0]), h1 = __float2half_rn(v[1]);
      const __half l0 = __float2half_rn(v[0] - __half2float(h0)), l1 = __float2half_rn(v[1] - __half2float(h1));
      h[e] = (unsigned)__half_as_ushort(h0) | ((unsigned)__half_as_ushort(h1) << 16);
      l[e] = (unsigned)__half_as_ushort(l0) | ((unsigned)__half_as_ushort(l1) << 16);
    }
    const int64_t row = c / hop8;
    const int col = (int)(c - row * hop8);
    uint4* o = out + ((int64_t)b * rows + row) * (3 * hop8) + col;
    const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]);
    o[0] = hv;
    o[hop8] = hv;
    o[2 * hop8] = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

// spec [B*rows][ld] fp32 (Re bins 0..nbins-1 | Im bins nbins..2*nbins-1) -> log-mel (B, n_mels, T), energy (B, T).
// A workgroup covers FB = 16 consecutive frames of one utterance, 4 per wave.  A wave first issues every load of its
// four frames (2 * NIT per lane per frame, all in flight together), then writes the magnitudes sqrt(re^2 + im^2 + eps) to
// LDS and reduces the energy; then lane m sums mel filter m over its nonzero range for the four frames at once (the
// Slaney triangles are stored packed: vals[off[m] .. off[m+1]) apply to bins start[m] ..).  The filters beyond the
// 64th — the widest ones — are each split over four lanes and combined by two shuffles.  The output tile is staged in
// LDS so that the (B, n_mels, T) result is written in 64-byte runs.
constexpr int FB = 16;          // frames per workgroup
constexpr int FW = 4;           // frames per wave
constexpr int MAXM = 80;        // mel channels: 64 (one lane each) + 16 (four lanes each)
constexpr int MAXNZ = 6144;     // packed filterbank entries

template <int NIT>              // ceil(nbins / 64)
__global__ __launch_bounds__(256) void mel_from_spec_kernel(const float* __restrict__ spec, int ld, int rows, int T, int nbins,
                                                            const float* __restrict__ vals, const int* __restrict__ start,
                                                            const int* __restrict__ off, int n_mels, float eps, float clip,
                                                            float* __restrict__ mel, float* __restrict__ energy) {
  constexpr int MB = NIT * 64;
  __shared__ float mag[4][FW][MB];
  __shared__ float tile[MAXM][FB + 1];
  __shared__ float fvals[MAXNZ];
  __shared__ int fstart[MAXM], foff[MAXM + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * FB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nnz = off[n_mels];
  for (int i = threadIdx.x; i < nnz; i += 256) fvals[i] = vals[i];
  for (int i = threadIdx.x; i < n_mels; i += 256) fstart[i] = start[i];
  for (int i = threadIdx.x; i <= n_mels; i += 256) foff[i] = off[i];

  float re[FW][NIT], im[FW][NIT];
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    const int t = t0 + wave * FW + f;
    const float* row = spec + ((int64_t)b * rows + (t < T ? t : T - 1)) * ld;     // clamp: rows past T are never stored
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = lane + 64 * i;
      re[f][i] = c < nbins ? row[c] : 0.f;
      im[f][i] = c < nbins ? row[nbins + c] : 0.f;
    }
  }
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    float e2 = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const float m2 = re[f][i] * re[f][i] + im[f][i] * im[f][i];
      e2 += m2;
      mag[wave][f][lane + 64 * i] = sqrtf(m2 + eps);
    }
    e2 = wave_sum(e2);
    const int t = t0 + wave * FW + f;
    if (lane == 0 && t < T) energy[(int64_t)b * T + t] = sqrtf(e2);
  }
  __syncthreads();       // filterbank tables and this wave's magnitudes are in LDS
  {
    // filters 0..63: one lane each
    const int m = lane;
    if (m < n_mels) {
      const int q0 = foff[m], q1 = foff[m + 1], s0 = fstart[m];
      float s[FW] = {0.f, 0.f, 0.f, 0.f};
      for (int q = q0; q < q1; ++q) {
        const float w = fvals[q];
#pragma unroll
        for (int f = 0; f < FW; ++f) s[f] += w * mag[wave][f][s0 + q - q0];
      }
#pragma unroll
      for (int f = 0; f < FW; ++f) tile[m][wave * FW + f] = logf(fmaxf(s[f], clip));
    }
  }
  {
    // filters 64..79: four lanes each (lane = part * 16 + (m - 64))
    const int m = 64 + (lane & 15), part = lane >> 4;
    float s[FW] = {0.f, 0.f, 0.f, 0.f};
    if (m < n_mels) {
      const int q0 = foff[m], q1 = foff[m + 1], s0 = fstart[m];
      for (int q = q0 + part; q < q1; q += 4) {
        const float w = fvals[q];
#pragma unroll
        for (int f = 0; f < FW; ++f) s[f] += w * mag[wave][f][s0 + q - q0];
      }
    }
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      s[f] += __shfl_xor(s[f], 16);
      s[f] += __shfl_xor(s[f], 32);
    }
    if (m < n_mels && part == 0) {
#pragma unroll
      for (int f = 0; f < FW; ++f) tile[m][wave * FW + f] = logf(fmaxf(s[f], clip));
    }
  }
  __syncthreads();
  const int nf = (T - t0) < FB ? (T - t0) : FB;
  for (int i = threadIdx.x; i < n_mels * FB; i += 256) {
    const int m = i / FB, f = i - m * FB;
    if (f < nf) mel[((int64_t)b * n_mels + m) * T + t0 + f] = tile[m][f];
  }
}

}  // namespace

extern "C" int ttsk_stft_frames(const float* wav, void* out16, int B, int len, int pad, int rows, int hop, float scale,
                                void* stream) {
  TTSK_REQUIRE(wav && out16 && B > 0 && B <= 65535 && len > 1 && rows > 0 && hop > 0 && (hop & 7) == 0,
               "ttsk_stft_frames: bad arguments (hop must be a multiple of 8)");
  TTSK_REQUIRE(pad >= 0 && pad < len, "ttsk_stft_frames: reflect padding needs pad < len");
  TTSK_REQUIRE((((uintptr_t)out16) & 15) == 0, "ttsk_stft_frames: output must be 16-byte aligned");
  const int64_t n8 = (int64_t)rows * hop / 8;
  int blocks = (int)((n8 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(stft_frames_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, wav, (uint4*)out16, len, pad, rows,
                     hop / 8, scale);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_mel_from_spec(const float* spec, int ld, const float* basis_vals, const int32_t* basis_start,
                                  const int32_t* basis_off, int nnz, float* mel, float* energy, int B, int rows, int T,
                                  int nbins, int n_mels, float eps, float clip, void* stream) {
  TTSK_REQUIRE(spec && basis_vals && basis_start && basis_off && mel && energy, "ttsk_mel_from_spec: null pointer");
  TTSK_REQUIRE(B > 0 && B <= 65535 && T > 0 && rows >= T && nbins > 0 && nbins <= 1088 && ld >= 2 * nbins && n_mels > 0 &&
                   n_mels <= MAXM && nnz >= 0 && nnz <= MAXNZ,
               "ttsk_mel_from_spec: bad sizes (nbins <= 1088, n_mels <= %d, packed filterbank <= %d entries)", MAXM, MAXNZ);
  dim3 grid((T + FB - 1) / FB, B);
  if (nbins <= 576)
    hipLaunchKernelGGL(mel_from_spec_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, spec, ld, rows, T, nbins, basis_vals,
                       basis_start, basis_off, n_mels, eps, clip, mel, energy);
  else
    hipLaunchKernelGGL(mel_from_spec_kernel<17>, grid, dim3(256), 0, (hipStream_t)stream, spec, ld, rows, T, nbins, basis_vals,
                       basis_start, basis_off, n_mels, eps, clip, mel, energy);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
