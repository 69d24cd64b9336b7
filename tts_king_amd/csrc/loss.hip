// loss.hip — FastSpeech2Loss forward AND its gradient in one pass.
// reference: fs_two/model/loss.py:24-134 (use_cwt False):
//   mel terms: predictions and targets are zeroed on PAD rows, then MSE + L1 (mel) + L1 (postnet mel), each a mean
//   over ALL B*T*n_mel elements (PAD rows stay in the denominator);
//   pitch / energy / log-duration: MSE over valid phonemes only; log_d_target = log(d + 1).
// Outputs: losses[8] = {total, mel_total, pitch, energy, duration, 0, 0, n_valid_phonemes} and the gradients of
// `grad_scale * total` wrt every prediction (grad_scale = 1/grad_acc_step, train.py:43).
#include "common.h"

namespace {

struct LossArgs {
  const float* mel;      // [B][T][nm]
  const float* post;     // [B][T][nm]
  const float* mel_t;    // [B][Tt][nm] (Tt >= T; targets are cropped to T, loss.py:57)
  const long long* mel_lens;
  const float* pitch; const float* energy; const float* logd;       // [B][L]
  const float* pitch_t; const float* energy_t; const long long* dur_t;
  const long long* src_lens;
  float* dmel_sum;       // d/dmel of the two mel terms + d/dpost (postnet output = postnet(mel) + mel)
  float* dpost;          // d/dpost
  float* dpitch; float* denergy; float* dlogd;
  float* partials;       // [nblk][6]
  int B, T, Tt, nm, L;
  float grad_scale;
  const int* frame_limit;   // device int32[1] or null: the mel means run over B * frame_limit[0] * nm elements (bucketed T)
};

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// MEL / VAR: the frame-level terms (partials columns 0-2) and the phoneme-level ones (columns 3-5).  Both = the one-launch form; one each =
// the two-stream form (ttsk_fs2_loss_mel on the step's stream, ttsk_fs2_loss_var on the predictors'): the same grid, the same per-thread
// order, so the partial rows — and the losses ttsk_fs2_loss_finalize makes of them — are bit-identical to the one-launch form's.
template <bool MEL, bool VAR>
__global__ __launch_bounds__(256) void loss_kernel(const LossArgs a) {
  __shared__ float red[6][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  const int nm4 = a.nm >> 2;
  const int64_t n4 = (int64_t)a.B * a.T * nm4;
  const float inv_n = 1.f / ((float)a.B * (a.frame_limit ? a.frame_limit[0] : a.T) * a.nm);
  const float gs = a.grad_scale * inv_n;
  if constexpr (MEL)
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t rt = i / nm4;
    const int c = (int)(i - rt * nm4) * 4;
    const int b = (int)(rt / a.T), t = (int)(rt - (int64_t)b * a.T);
    const bool ok = t < a.mel_lens[b];
    f32x4 dm = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
    if (ok) {
      const f32x4 m = *(const f32x4*)(a.mel + rt * a.nm + c), p = *(const f32x4*)(a.post + rt * a.nm + c);
      const f32x4 tg = *(const f32x4*)(a.mel_t + ((int64_t)b * a.Tt + t) * a.nm + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d1 = m[e] - tg[e], d2 = p[e] - tg[e];
        acc[0] += d1 * d1; acc[1] += fabsf(d1); acc[2] += fabsf(d2);
        dp[e] = gs * sgn(d2);
        dm[e] = gs * (2.f * d1 + sgn(d1)) + dp[e];
      }
    }
    *(f32x4*)(a.dmel_sum + rt * a.nm + c) = dm;
    *(f32x4*)(a.dpost + rt * a.nm + c) = dp;
  }
  // phoneme-level terms
  if constexpr (VAR) {
  float nv = 0.f;
  for (int b = 0; b < a.B; ++b) nv += (float)a.src_lens[b];
  const float gv = a.grad_scale * 2.f / nv;
  const int nph = a.B * a.L;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nph; i += gridDim.x * 256) {
    const int b = i / a.L, l = i - b * a.L;
    const bool ok = l < a.src_lens[b];
    float g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (ok) {
      const float d1 = a.pitch[i] - a.pitch_t[i], d2 = a.energy[i] - a.energy_t[i];
      const float d3 = a.logd[i] - logf((float)a.dur_t[i] + 1.f);
      acc[3] += d1 * d1; acc[4] += d2 * d2; acc[5] += d3 * d3;
      g1 = gv * d1; g2 = gv * d2; g3 = gv * d3;
    }
    a.dpitch[i] = g1; a.denergy[i] = g2; a.dlogd[i] = g3;
  }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) { const float s = wave_sum(acc[q]); if (lane == 0) red[q][wave] = s; }
  __syncthreads();
  const int q0 = MEL ? 0 : 3, q1 = VAR ? 6 : 3;
  if ((int)threadIdx.x >= q0 && (int)threadIdx.x < q1)
    a.partials[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(64) void loss_finalize_kernel(const float* __restrict__ partials, const float* __restrict__ partials_var, int nblk,
                                                           const long long* __restrict__ src_lens,
                                                           int B, float n_mel_elems_all, const int* __restrict__ frame_limit, int T,
                                                           float* __restrict__ losses) {
  const float n_mel_elems = frame_limit ? n_mel_elems_all / (float)T * (float)frame_limit[0] : n_mel_elems_all;
  // one wave: lane l adds blocks l, l+64, ... of all six quantities (its loads are independent, so they overlap; six lanes
  // walking 256 blocks one dependent load at a time took 22 us), then a butterfly in fixed order: deterministic
  __shared__ double tot[6];
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblk; b += 64) {
#pragma unroll
    for (int q = 0; q < 6; ++q) s[q] += (q < 3 ? partials : partials_var)[b * 6 + q];      // (two buffers when the halves came from two streams)
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
    if (threadIdx.x == 0) tot[q] = s[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double nv = 0.0;
    for (int b = 0; b < B; ++b) nv += (double)src_lens[b];
    const double mel_total = (tot[0] + tot[1] + tot[2]) / n_mel_elems;
    const double pl = tot[3] / nv, el = tot[4] / nv, dl = tot[5] / nv;
    losses[0] = (float)(mel_total + dl + pl + el);
    losses[1] = (float)mel_total; losses[2] = (float)pl; losses[3] = (float)el; losses[4] = (float)dl;
    losses[5] = 0.f; losses[6] = 0.f; losses[7] = (float)nv;
  }
}

}  // namespace

extern "C" int ttsk_fs2_loss_nblocks(void) { return 256; }

extern "C" int ttsk_fs2_loss(const float* mel, const float* post, const float* mel_target, const int64_t* mel_lens,
                             const float* pitch, const float* energy, const float* logd, const float* pitch_target,
                             const float* energy_target, const int64_t* dur_target, const int64_t* src_lens, int B, int T,
                             int T_target, int n_mel, int L, float grad_scale, float* dmel_sum, float* dpost, float* dpitch,
                             float* denergy, float* dlogd, float* partials, float* losses, const int32_t* frame_limit, void* stream) {
  TTSK_REQUIRE(mel && post && mel_target && mel_lens && pitch && energy && logd && pitch_target && energy_target && dur_target &&
                   src_lens && dmel_sum && dpost && dpitch && denergy && dlogd && partials && losses, "fs2_loss: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && T_target >= T && L > 0 && n_mel > 0 && (n_mel & 3) == 0, "fs2_loss: bad sizes");
  LossArgs a{mel, post, mel_target, (const long long*)mel_lens, pitch, energy, logd, pitch_target, energy_target,
             (const long long*)dur_target, (const long long*)src_lens, dmel_sum, dpost, dpitch, denergy, dlogd, partials,
             B, T, T_target, n_mel, L, grad_scale, frame_limit};
  const int nblk = ttsk_fs2_loss_nblocks();
  hipLaunchKernelGGL((loss_kernel<true, true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, partials, nblk, (const long long*)src_lens, B,
                     (float)B * T * n_mel, frame_limit, T, losses);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ---- the same loss as three launches, for a step whose variance predictors run on a stream of their own (tts_king_amd/graph.py): the frame-level
// terms and their gradients on the step's stream (nothing of the predictors is read), the phoneme-level terms on the predictors' stream,
// the eight loss values wherever both sets of partial rows are visible.  partials: [ttsk_fs2_loss_nblocks()][6] each (the frame-level
// launch writes columns 0-2 of its buffer, the phoneme-level one columns 3-5 of its own — or of the same one).
extern "C" int ttsk_fs2_loss_mel(const float* mel, const float* post, const float* mel_target, const int64_t* mel_lens, int B, int T, int T_target,
                                 int n_mel, float grad_scale, float* dmel_sum, float* dpost, float* partials, const int32_t* frame_limit,
                                 void* stream) {
  TTSK_REQUIRE(mel && post && mel_target && mel_lens && dmel_sum && dpost && partials, "fs2_loss_mel: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && T_target >= T && n_mel > 0 && (n_mel & 3) == 0, "fs2_loss_mel: bad sizes");
  LossArgs a{mel, post, mel_target, (const long long*)mel_lens, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dmel_sum, dpost,
             nullptr, nullptr, nullptr, partials, B, T, T_target, n_mel, 1, grad_scale, frame_limit};
  hipLaunchKernelGGL((loss_kernel<true, false>), dim3(ttsk_fs2_loss_nblocks()), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_fs2_loss_var(const float* pitch, const float* energy, const float* logd, const float* pitch_target, const float* energy_target,
                                 const int64_t* dur_target, const int64_t* src_lens, int B, int L, float grad_scale, float* dpitch, float* denergy,
                                 float* dlogd, float* partials, void* stream) {
  TTSK_REQUIRE(pitch && energy && logd && pitch_target && energy_target && dur_target && src_lens && dpitch && denergy && dlogd && partials,
               "fs2_loss_var: null pointer");
  TTSK_REQUIRE(B > 0 && L > 0, "fs2_loss_var: bad sizes");
  LossArgs a{nullptr, nullptr, nullptr, nullptr, pitch, energy, logd, pitch_target, energy_target, (const long long*)dur_target,
             (const long long*)src_lens, nullptr, nullptr, dpitch, denergy, dlogd, partials, B, 1, 1, 4, L, grad_scale, nullptr};
  hipLaunchKernelGGL((loss_kernel<false, true>), dim3(ttsk_fs2_loss_nblocks()), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_fs2_loss_finalize(const float* partials_mel, const float* partials_var, const int64_t* src_lens, int B, int T, int n_mel,
                                      const int32_t* frame_limit, float* losses, void* stream) {
  TTSK_REQUIRE(partials_mel && partials_var && src_lens && losses, "fs2_loss_finalize: null pointer");
  TTSK_REQUIRE(B > 0 && T > 0 && n_mel > 0, "fs2_loss_finalize: bad sizes");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials_mel, partials_var, ttsk_fs2_loss_nblocks(),
                     (const long long*)src_lens, B, (float)B * T * n_mel, frame_limit, T, losses);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
