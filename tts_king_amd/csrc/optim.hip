// optim.hip — global-norm gradient clipping + Adam + LR schedule + bf16 weight shadow, over ONE flat buffer.
// reference: train.py:47-54 (clip_grad_norm_(1.0) -> step_and_update_lr -> zero_grad),
//            fs_two/model/optimizer.py:35-53 (lr = d^-0.5 * min(s^-0.5, warmup^-1.5 * s) * anneal^{#(s > a_k)}),
//            torch.optim.Adam (betas (0.95, 0.999), eps 1e-5, bias correction, no weight decay).
// The step counters, learning rate and bias corrections live in a small DEVICE state block so that a captured
// hipGraph of the train step replays with fresh values (kernel arguments are frozen at capture).
#include "common.h"

namespace {

struct OptState {          // mirrors tts_king_amd/optimizer.py: 8 x 8 bytes
  long long sched_step;    // ScheduledOptim.current_step (counts optimizer updates)
  long long adam_t;        // Adam's per-parameter step
  unsigned long long rng_seed, rng_step;   // dropout counter state (rng = &rng_seed)
  float lr, bc1, bc2, clip_coef;
  float gnorm, pad0;
  long long pad1;
};

__global__ void optim_advance_kernel(OptState* st, float d_model, float warmup, float a0, float a1, float a2, float a3,
                                     int n_anneal, float anneal_rate, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long s = ++st->sched_step;
  const long long t = ++st->adam_t;
  double lr = fmin(pow((double)s, -0.5), pow((double)warmup, -1.5) * (double)s);
  const float an[4] = {a0, a1, a2, a3};
  for (int i = 0; i < n_anneal; ++i)
    if ((double)s > (double)an[i]) lr *= (double)anneal_rate;
  st->lr = (float)(pow((double)d_model, -0.5) * lr);
  st->bc1 = (float)(1.0 - pow((double)b1, (double)t));
  st->bc2 = (float)(1.0 - pow((double)b2, (double)t));
}

__global__ void rng_advance_kernel(OptState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->rng_step += 1;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// clip_coef = min(1, max_norm / (||g|| + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int nblk, float max_norm, OptState* st) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    st->gnorm = norm;
    const float c = max_norm / (norm + 1e-6f);
    st->clip_coef = c < 1.f ? c : 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n,
                                                   const OptState* __restrict__ st, float b1, float b2, float eps, int zero_grad) {
  const float lr = st->lr, bc1 = st->bc1, coef = st->clip_coef;
  const float isb2 = 1.f / sqrtf(st->bc2);
  const float step = lr / bc1;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = __builtin_nontemporal_load((f32x4*)(p + i * 4)), gg = __builtin_nontemporal_load((const f32x4*)(g + i * 4)),
          mm = __builtin_nontemporal_load((f32x4*)(m + i * 4)), vv = __builtin_nontemporal_load((f32x4*)(v + i * 4));      // streamed once: see adam_pack_kernel
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * coef;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      pp[e] -= step * mm[e] / (sqrtf(vv[e]) * isb2 + eps);
    }
    __builtin_nontemporal_store(pp, (f32x4*)(p + i * 4)); __builtin_nontemporal_store(mm, (f32x4*)(m + i * 4)); __builtin_nontemporal_store(vv, (f32x4*)(v + i * 4));
    if (zero_grad) *(f32x4*)(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (shadow) *(uint2*)(shadow + i * 4) = make_uint2(pack_bf2(pp[0], pp[1]), pack_bf2(pp[2], pp[3]));
  }
}

// ---- the whole optimizer step as TWO launches (was five: rng_advance, optim_advance, sumsq, clip_coef, adam):
//   sumsq_advance_kernel: per-block sums of g^2, and its block 0 also advances the scheduler / Adam / dropout counters
//   (it reads nothing the counters feed; the Adam launch after it reads the advanced state);
//   adam_clip_kernel: every workgroup first folds the 1024 partials in the same fixed order (double) into the global norm
//   and the clip coefficient, then applies Adam; workgroup 0 records gnorm / clip_coef in the state block.
struct SchedArgs { float d_model, warmup, a0, a1, a2, a3, anneal_rate, b1, b2; int n_anneal, advance_rng; };

__global__ __launch_bounds__(256) void sumsq_advance_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials, OptState* st,
                                                            const SchedArgs sc) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long sstep = ++st->sched_step;
    const long long t = ++st->adam_t;
    double lr = fmin(pow((double)sstep, -0.5), pow((double)sc.warmup, -1.5) * (double)sstep);
    const float an[4] = {sc.a0, sc.a1, sc.a2, sc.a3};
    for (int i = 0; i < sc.n_anneal; ++i)
      if ((double)sstep > (double)an[i]) lr *= (double)sc.anneal_rate;
    st->lr = (float)(pow((double)sc.d_model, -0.5) * lr);
    st->bc1 = (float)(1.0 - pow((double)sc.b1, (double)t));
    st->bc2 = (float)(1.0 - pow((double)sc.b2, (double)t));
    if (sc.advance_rng) st->rng_step += 1;
  }
}

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n, OptState* st,
                                                        const float* __restrict__ partials, int nblk, float max_norm, float b1, float b2,
                                                        float eps, int zero_grad) {
  __shared__ double red[256];
  {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  }
  const float norm = (float)sqrt(red[0]);
  const float cc = max_norm / (norm + 1e-6f);
  const float coef = cc < 1.f ? cc : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->gnorm = norm; st->clip_coef = coef; }
  const float lr = st->lr, bc1 = st->bc1;
  const float isb2 = 1.f / sqrtf(st->bc2);
  const float step = lr / bc1;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = *(f32x4*)(p + i * 4), gg = *(const f32x4*)(g + i * 4), mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gg[e] * coef;
      mm[e] = b1 * mm[e] + (1.f - b1) * ge;
      vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
      pp[e] -= step * mm[e] / (sqrtf(vv[e]) * isb2 + eps);
    }
    *(f32x4*)(p + i * 4) = pp; *(f32x4*)(m + i * 4) = mm; *(f32x4*)(v + i * 4) = vv;
    if (zero_grad) *(f32x4*)(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (shadow) *(uint2*)(shadow + i * 4) = make_uint2(pack_bf2(pp[0], pp[1]), pack_bf2(pp[2], pp[3]));
  }
}


// ---- Adam that also writes the window kernels' weight packs (round 3).  After round 2 the optimizer step was followed by
// win_pack_kernel: a pure re-layout of the bf16 shadow Adam had just written (fragment-major copies of every FFT-block / PostNet
// weight as it is and transposed with flipped taps, 140 MB, 55 us on the step's serial tail).  Here the workgroup that updates a tile
// of a packed weight — 32 storage rows x 256 storage columns of one tap, the tile ffn_conv.hip:pack_one transposes through LDS —
// keeps the tile's bf16 values in LDS and stores the 16 fragments of the plain pack and the 16 of the transposed pack itself (1 KiB
// runs each).  Everything outside the packed weights (biases, LayerNorms, embeddings, predictors, the 80-channel ends: 5 % of the
// parameters) goes through the flat loop over the gap ranges between them.  Same arithmetic per element as adam_clip_kernel: the
// parameters, moments, shadow and packs are bit-identical to adam_clip_kernel + win_pack_kernel.
struct AdamTables {
  const ttsk_adam_item* items;   // device, sorted by tile0
  int n_items, n_tiles;
  const long long* gaps;         // device: n_gaps x {start, end, first compact f32x4 index}
  int n_gaps;
  long long gap4;                // f32x4 groups in all gaps
};

__device__ __forceinline__ void adam4(f32x4& pp, const f32x4 gg, f32x4& mm, f32x4& vv, float coef, float b1, float b2, float step, float isb2,
                                      float eps) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ge = gg[e] * coef;
    mm[e] = b1 * mm[e] + (1.f - b1) * ge;
    vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
    pp[e] -= step * mm[e] / (sqrtf(vv[e]) * isb2 + eps);
  }
}

__global__ __launch_bounds__(256) void adam_pack_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ shadow, OptState* st,
                                                        const float* __restrict__ partials, int nblk, float max_norm, float b1, float b2,
                                                        float eps, int zero_grad, const AdamTables tb) {
  __shared__ double red[256];
  __shared__ __attribute__((aligned(16))) unsigned short tile[32][256 + 8];
  {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  }
  const float norm = (float)sqrt(red[0]);
  const float cc = max_norm / (norm + 1e-6f);
  const float coef = cc < 1.f ? cc : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->gnorm = norm; st->clip_coef = coef; }
  const float lr = st->lr, bc1 = st->bc1;
  const float isb2 = 1.f / sqrtf(st->bc2);
  const float step = lr / bc1;
  const int tid = threadIdx.x;
  // ---- the packed weights, tile by tile
  for (int t = blockIdx.x; t < tb.n_tiles; t += gridDim.x) {
    int lo = 0, hi = tb.n_items;               // invariant: items[lo].tile0 <= t < items[hi].tile0
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tb.items[mid].tile0 <= t) lo = mid; else hi = mid;
    }
    const ttsk_adam_item it = tb.items[lo];
    const int Cs = it.Cs, K = it.K, Ds = it.Ds;
    const int ncb = Ds / 256, nks = Cs / 32;
    const int lt = t - it.tile0;
    const int cb = lt % ncb, ks = (lt / ncb) % nks, ts = lt / (ncb * nks);        // ts = storage tap
    __syncthreads();                            // the previous tile's LDS readers are done
#pragma unroll
    for (int u = 0; u < 8; ++u) {               // 32 rows x 64 groups of four floats; a wave takes a whole 1 KiB row
      const int idx = u * 256 + tid, r = idx >> 6, c4 = idx & 63;
      const int64_t e = it.off + ((int64_t)(ks * 32 + r) * K + ts) * Ds + cb * 256 + c4 * 4;
      // p, m, v and g stream through once per step (1.1 GB): nontemporal, so that they do not evict the shadow and the packs the next
      // forward reads (measured -25 us on the kernel)
      f32x4 pp = __builtin_nontemporal_load((f32x4*)(p + e)), mm = __builtin_nontemporal_load((f32x4*)(m + e)), vv = __builtin_nontemporal_load((f32x4*)(v + e));
      const f32x4 gg = __builtin_nontemporal_load((const f32x4*)(g + e));
      adam4(pp, gg, mm, vv, coef, b1, b2, step, isb2, eps);
      __builtin_nontemporal_store(pp, (f32x4*)(p + e)); __builtin_nontemporal_store(mm, (f32x4*)(m + e)); __builtin_nontemporal_store(vv, (f32x4*)(v + e));
      if (zero_grad) *(f32x4*)(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
      const uint2 w = make_uint2(pack_bf2(pp[0], pp[1]), pack_bf2(pp[2], pp[3]));
      *(uint2*)(shadow + e) = w;
      *(uint2*)&tile[r][c4 * 4] = w;
    }
    __syncthreads();
    if (it.pack) {
      // plain pack [K][Ds/32][Cs/16][64][8]: the tile's rows are couts ks*32.., its columns cins cb*256..: 8 k-steps x 2 cout tiles
      bf16_t* dst = (bf16_t*)it.pack;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pi = u * 256 + tid, f = pi >> 6, l = pi & 63;
        const int ksl = f >> 1, cl = f & 1;
        const uint4 val = *(const uint4*)&tile[cl * 16 + (l & 15)][ksl * 32 + (l >> 4) * 8];
        const int64_t piece = ((int64_t)(ts * (Ds / 32) + cb * 8 + ksl) * (Cs / 16) + 2 * ks + cl) * 64 + l;
        *(uint4*)(dst + piece * 8) = val;
      }
    }
    if (it.pack_t) {
      // transposed pack, taps flipped (ffn_conv.hip:pack_one): Cout' = Ds, Cin' = Cs; 16 contiguous fragments
      const int tap = K - 1 - ts;
      bf16_t* out = (bf16_t*)it.pack_t + ((int64_t)(tap * nks + ks) * (Ds / 16) + cb * 16) * 512;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pi = u * 256 + tid, c = pi >> 6, l = pi & 63;
        const int col = c * 16 + (l & 15), r0 = (l >> 4) * 8;
        unsigned short q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = tile[r0 + j][col];
        *(uint4*)(out + (int64_t)pi * 8) = make_uint4(q[0] | ((unsigned)q[1] << 16), q[2] | ((unsigned)q[3] << 16), q[4] | ((unsigned)q[5] << 16),
                                                      q[6] | ((unsigned)q[7] << 16));
      }
    }
  }
  // ---- everything else: the gap ranges (multiples of four floats, 16-byte aligned), as one compact index space
  for (int64_t i = blockIdx.x * 256ll + tid; i < tb.gap4; i += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = tb.n_gaps;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tb.gaps[mid * 3 + 2] <= i) lo = mid; else hi = mid;
    }
    const int64_t e = tb.gaps[lo * 3] + (i - tb.gaps[lo * 3 + 2]) * 4;
    f32x4 pp = __builtin_nontemporal_load((f32x4*)(p + e)), mm = __builtin_nontemporal_load((f32x4*)(m + e)), vv = __builtin_nontemporal_load((f32x4*)(v + e));
    const f32x4 gg = __builtin_nontemporal_load((const f32x4*)(g + e));
    adam4(pp, gg, mm, vv, coef, b1, b2, step, isb2, eps);
    __builtin_nontemporal_store(pp, (f32x4*)(p + e)); __builtin_nontemporal_store(mm, (f32x4*)(m + e)); __builtin_nontemporal_store(vv, (f32x4*)(v + e));
    if (zero_grad) *(f32x4*)(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
    *(uint2*)(shadow + e) = make_uint2(pack_bf2(pp[0], pp[1]), pack_bf2(pp[2], pp[3]));
  }
}

}  // namespace

extern "C" int ttsk_optim_state_bytes(void) { return (int)sizeof(OptState); }

extern "C" int ttsk_optim_advance(void* state, float d_model, float warmup, const float* anneal_steps_host, int n_anneal,
                                  float anneal_rate, float beta1, float beta2, void* stream) {
  TTSK_REQUIRE(state && n_anneal >= 0 && n_anneal <= 4, "optim_advance: at most 4 anneal steps");
  float a[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_anneal; ++i) a[i] = anneal_steps_host[i];
  hipLaunchKernelGGL(optim_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (OptState*)state, d_model, warmup, a[0], a[1],
                     a[2], a[3], n_anneal, anneal_rate, beta1, beta2);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_rng_advance(void* state, void* stream) {
  TTSK_REQUIRE(state, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (OptState*)state);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n,
                                   void* state, float* partials /* >= 1024 floats */, float max_norm, float beta1, float beta2,
                                   float eps, int zero_grad, void* stream) {
  TTSK_REQUIRE(params && grads && exp_avg && exp_avg_sq && state && partials && n > 0, "clip_adam_step: null pointer");
  TTSK_REQUIRE((n & 3) == 0, "clip_adam_step: n must be a multiple of 4 (pad the flat buffer)");
  TTSK_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "clip_adam_step: alignment");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, s, grads, n, partials);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, partials, 1024, max_norm, (OptState*)state);
  hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n,
                     (const OptState*)state, beta1, beta2, eps, zero_grad);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_grad_sumsq(const float* grads, int64_t n, float* partials /* 1024 floats */, void* stream) {
  TTSK_REQUIRE(grads && partials && n > 0, "grad_sumsq: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, grads, n, partials);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_optim_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n, void* state,
                               float* partials /* >= 1024 floats */, float max_norm, float beta1, float beta2, float eps, int zero_grad,
                               float d_model, float warmup, const float* anneal_steps_host, int n_anneal, float anneal_rate,
                               int advance_rng, void* stream) {
  TTSK_REQUIRE(params && grads && exp_avg && exp_avg_sq && state && partials && n > 0, "optim_step: null pointer");
  TTSK_REQUIRE((n & 3) == 0, "optim_step: n must be a multiple of 4 (pad the flat buffer)");
  TTSK_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "optim_step: alignment");
  TTSK_REQUIRE(n_anneal >= 0 && n_anneal <= 4, "optim_step: at most 4 anneal steps");
  float a[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_anneal; ++i) a[i] = anneal_steps_host[i];
  const SchedArgs sc{d_model, warmup, a[0], a[1], a[2], a[3], anneal_rate, beta1, beta2, n_anneal, advance_rng};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_advance_kernel, dim3(1024), dim3(256), 0, s, grads, n, partials, (OptState*)state, sc);
  hipLaunchKernelGGL(adam_clip_kernel, dim3(2048), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n,
                     (OptState*)state, partials, 1024, max_norm, beta1, beta2, eps, zero_grad);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_optim_step_packed(float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n, void* state,
                                      float* partials /* >= 1024 floats */, float max_norm, float beta1, float beta2, float eps, int zero_grad,
                                      float d_model, float warmup, const float* anneal_steps_host, int n_anneal, float anneal_rate,
                                      int advance_rng, const ttsk_adam_item* dev_items, int n_items, int n_tiles, const int64_t* dev_gaps,
                                      int n_gaps, int64_t gap_floats, void* stream) {
  TTSK_REQUIRE(params && grads && exp_avg && exp_avg_sq && shadow_bf16 && state && partials && n > 0, "optim_step_packed: null pointer");
  TTSK_REQUIRE((n & 3) == 0 && (gap_floats & 3) == 0, "optim_step_packed: n and the gap total must be multiples of 4");
  TTSK_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)shadow_bf16) & 15) == 0,
               "optim_step_packed: alignment");
  TTSK_REQUIRE(n_anneal >= 0 && n_anneal <= 4, "optim_step_packed: at most 4 anneal steps");
  TTSK_REQUIRE(dev_items && n_items > 0 && n_tiles > 0 && n_gaps >= 0 && (n_gaps == 0 || dev_gaps) && gap_floats >= 0,
               "optim_step_packed: bad tables");
  TTSK_REQUIRE((int64_t)n_tiles * 8192 + gap_floats == n, "optim_step_packed: tiles (%d x 8192) + gaps (%lld) do not cover n = %lld", n_tiles,
               (long long)gap_floats, (long long)n);
  float a[4] = {0, 0, 0, 0};
  for (int i = 0; i < n_anneal; ++i) a[i] = anneal_steps_host[i];
  const SchedArgs sc{d_model, warmup, a[0], a[1], a[2], a[3], anneal_rate, beta1, beta2, n_anneal, advance_rng};
  const AdamTables tb{dev_items, n_items, n_tiles, (const long long*)dev_gaps, n_gaps, (long long)(gap_floats / 4)};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_advance_kernel, dim3(1024), dim3(256), 0, s, grads, n, partials, (OptState*)state, sc);
  hipLaunchKernelGGL(adam_pack_kernel, dim3(2048), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, (OptState*)state,
                     partials, 1024, max_norm, beta1, beta2, eps, zero_grad, tb);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
