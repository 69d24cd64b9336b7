// batchnorm.hip — PostNet BatchNorm1d (training: two-pass batch statistics over ALL B*T rows, PAD rows included)
// fused with tanh, dropout(0.5) and the final residual add; and its backward.
// reference: fs_two/transformer/Layers.py:133-143 (PostNet.forward), :85-129 (Conv1d + BatchNorm1d stacks),
//            fastspeech2.py:104 (postnet(output) + output).
// Layout: x [rows][C] bf16 (conv output), a thread owns 4 contiguous channels (C % 4 == 0).
#include "common.h"

namespace {

struct BnCommon {
  const void* x;          // conv output [rows][C]: fp32 (x_f32) or bf16
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  const uint64_t* rng;
  int rows, C, use_tanh;
  float p;
  unsigned site;
  int x_f32;
  // Frame limit (shape-bucketed training, tts_king_amd/engine.py): rows are (utterance, frame) pairs, row = u*seg_len + t, and
  // frames t >= frame_limit[0] do not exist in the reference's batch (it pads to the batch's own longest utterance, the
  // bucketed batch to a multiple of 32 frames): they are left out of the statistics, count as zero rows for the next
  // convolution (its zero padding) and carry no gradient.  null = every row exists.
  const int* frame_limit;
  int seg_len;
  // Dropout keep bits of the forward, one byte per (row, channel quad), bit e = channel c4 + e kept (may be null).  The slab
  // forward writes them and the slab backward kernels read them: Philox costs 40 quarter-rate integer multiplies per quad, and
  // regenerating the mask twice in the backward made those kernels VALU-bound.
  uint8_t* keep;
};
// tanh = 1 - 2 / (exp(2x) + 1) on the hardware exp and reciprocal: absolute error ~1e-7 (the result is stored as bf16, or multiplies
// a bf16 gradient); libm's tanhf is ~4x the instructions, and these kernels are VALU-bound
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }
__device__ __forceinline__ bool bn_live(const int* frame_limit, int seg_len, int r) {
  return !frame_limit || (r % seg_len) < frame_limit[0];
}
__device__ __forceinline__ int bn_rows(const int* frame_limit, int seg_len, int rows) {
  return frame_limit ? (rows / seg_len) * frame_limit[0] : rows;
}

__device__ __forceinline__ void ld4(const bf16_t* p, float v[4]) {
  const uint2 u = *(const uint2*)p;
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
}
__device__ __forceinline__ void ldx4(const void* x, int x_f32, int64_t off, float v[4]) {
  if (x_f32) {
    const f32x4 t = *(const f32x4*)((const float*)x + off);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    ld4((const bf16_t*)x + off, v);
  }
}
__device__ __forceinline__ uint4 bits4(const uint64_t* rng, unsigned site, unsigned e4) {
  const uint64_t seed = rng[0], step = rng[1];
  return Philox::gen(make_uint2((unsigned)seed, (unsigned)(seed >> 32)), make_uint4(e4, site, (unsigned)step, (unsigned)(step >> 32)));
}

// ---- pass 1: per-column sum and sum of squares -> partials[blk][2C]
__global__ __launch_bounds__(256) void bn_stats_kernel(const void* __restrict__ x, int x_f32, int rows, int C, float* __restrict__ partials,
                                                       const int* __restrict__ frame_limit, int seg_len) {
  extern __shared__ float red[];  // [rpi][4*tpr] x 2
  const int tpr = C >> 2, rpi = 256 / tpr;
  const int r0 = threadIdx.x / tpr, c4 = (threadIdx.x % tpr) * 4;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int rb = blockIdx.x * per;
  const int re = min(rb + per, rows);
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r0 < rpi)
    for (int r = rb + r0; r < re; r += rpi) {
      if (!bn_live(frame_limit, seg_len, r)) continue;
      float v[4];
      ldx4(x, x_f32, (int64_t)r * C + c4, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
  if (r0 < rpi) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[r0 * C + c4 + e] = s[e]; red[rpi * C + r0 * C + c4 + e] = q[e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < rpi; ++k) { a += red[k * C + c]; b += red[rpi * C + k * C + c]; }
    partials[(int64_t)blockIdx.x * 2 * C + c] = a;
    partials[(int64_t)blockIdx.x * 2 * C + C + c] = b;
  }
}

// mean / rstd from the partials (double accumulation), running statistics update (momentum, unbiased variance)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nblk, int C, int rows_all, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          long long* __restrict__ nbt, const int* __restrict__ frame_limit, int seg_len) {
  const int rows = bn_rows(frame_limit, seg_len, rows_all);
  // 16 channels x 16 row-groups per workgroup; every thread adds its partial rows in ascending order (double), the 16
  // group sums are combined in fixed order: deterministic
  __shared__ double rs[16][17], rq[16][17];
  const int cx = threadIdx.x & 15, gy = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // 8 loads in flight per thread (the partial rows are independent; a one-row-per-iteration loop paid one L2 latency per
    // row: 11 us for 423 rows); the order of additions is fixed
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    int b = gy;
    for (; b + 48 < nblk; b += 64) {
      const float a0 = partials[(int64_t)b * 2 * C + c], b0 = partials[(int64_t)b * 2 * C + C + c];
      const float a1 = partials[(int64_t)(b + 16) * 2 * C + c], b1 = partials[(int64_t)(b + 16) * 2 * C + C + c];
      const float a2 = partials[(int64_t)(b + 32) * 2 * C + c], b2 = partials[(int64_t)(b + 32) * 2 * C + C + c];
      const float a3 = partials[(int64_t)(b + 48) * 2 * C + c], b3 = partials[(int64_t)(b + 48) * 2 * C + C + c];
      s += a0; q += b0; s1 += a1; q1 += b1; s2 += a2; q2 += b2; s3 += a3; q3 += b3;
    }
    for (; b < nblk; b += 16) { s += partials[(int64_t)b * 2 * C + c]; q += partials[(int64_t)b * 2 * C + C + c]; }
    s = (s + s1) + (s2 + s3);
    q = (q + q1) + (q2 + q3);
  }
  rs[gy][cx] = s;
  rq[gy][cx] = q;
  __syncthreads();
  if (gy != 0 || c >= C) return;
  s = 0.0; q = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { s += rs[k][cx]; q += rq[k][cx]; }
  const double m = s / rows;
  double var = q / rows - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * rows / (rows > 1 ? rows - 1 : 1));
  }
}

__global__ __launch_bounds__(256) void rsqrt_eps_kernel(const float* __restrict__ var, float eps, float* __restrict__ rstd, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) rstd[i] = 1.f / sqrtf(var[i] + eps);
}

// ---- pass 2: y = tanh?(gamma * xhat + beta) -> dropout -> (+ residual) -> out
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnCommon a, const float* __restrict__ resid, bf16_t* __restrict__ out16,
                                                       float* __restrict__ out32) {
  const int tpr = a.C >> 2;
  const int64_t n = (int64_t)a.rows * tpr;
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / tpr), c4 = (int)(i - (int64_t)r * tpr) * 4;
    if (!bn_live(a.frame_limit, a.seg_len, r)) {     // a frame past the batch's own length: a zero row (the next conv's padding)
      if (out16) *(uint2*)(out16 + (int64_t)r * a.C + c4) = make_uint2(0u, 0u);
      if (out32) *(f32x4*)(out32 + (int64_t)r * a.C + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    float v[4];
    ldx4(a.x, a.x_f32, (int64_t)r * a.C + c4, v);
    const f32x4 m = *(const f32x4*)(a.mean + c4), rs = *(const f32x4*)(a.rstd + c4);
    const f32x4 g = *(const f32x4*)(a.gamma + c4), b = *(const f32x4*)(a.beta + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (v[e] - m[e]) * rs[e] * g[e] + b[e];
      if (a.use_tanh) v[e] = tanh_fast(v[e]);
    }
    if (a.p > 0.f) {
      const uint4 bb = bits4(a.rng, a.site, (unsigned)i);
      v[0] = bb.x >= thr ? v[0] * scale : 0.f; v[1] = bb.y >= thr ? v[1] * scale : 0.f;
      v[2] = bb.z >= thr ? v[2] * scale : 0.f; v[3] = bb.w >= thr ? v[3] * scale : 0.f;
    }
    if (resid) {
      const f32x4 rr = *(const f32x4*)(resid + (int64_t)r * a.C + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rr[e];
    }
    if (out16) *(uint2*)(out16 + (int64_t)r * a.C + c4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    if (out32) *(f32x4*)(out32 + (int64_t)r * a.C + c4) = f32x4{v[0], v[1], v[2], v[3]};
  }
}

// gradient wrt the BN output y (before tanh/dropout) for 4 channels of one row.  No branch around the loads (a row past the frame
// limit is real memory, its results are zeroed at the end), so that the loads of several rows of an unrolled loop go out together.
__device__ __forceinline__ void bn_dy(const BnCommon& a, const void* dout, int dout_f32, int r, int c4, int64_t i, unsigned thr,
                                      float scale, float xh[4], float dy[4]) {
  float v[4];
  ldx4(a.x, a.x_f32, (int64_t)r * a.C + c4, v);
  const f32x4 m = *(const f32x4*)(a.mean + c4), rs = *(const f32x4*)(a.rstd + c4);
  if (dout_f32) {
    const f32x4 d = *(const f32x4*)((const float*)dout + (int64_t)r * a.C + c4);
    dy[0] = d[0]; dy[1] = d[1]; dy[2] = d[2]; dy[3] = d[3];
  } else {
    ld4((const bf16_t*)dout + (int64_t)r * a.C + c4, dy);
  }
  if (a.p > 0.f) {
    if (a.keep) {
      const unsigned kb = a.keep[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) dy[e] = ((kb >> e) & 1u) ? dy[e] * scale : 0.f;
    } else {
      const uint4 bb = bits4(a.rng, a.site, (unsigned)i);
      dy[0] = bb.x >= thr ? dy[0] * scale : 0.f; dy[1] = bb.y >= thr ? dy[1] * scale : 0.f;
      dy[2] = bb.z >= thr ? dy[2] * scale : 0.f; dy[3] = bb.w >= thr ? dy[3] * scale : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) xh[e] = (v[e] - m[e]) * rs[e];
  if (a.use_tanh) {
    const f32x4 g = *(const f32x4*)(a.gamma + c4), b = *(const f32x4*)(a.beta + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float t = tanh_fast(xh[e] * g[e] + b[e]); dy[e] *= 1.f - t * t; }
  }
  if (!bn_live(a.frame_limit, a.seg_len, r)) {       // no such frame in the reference's batch: no gradient, no statistics
#pragma unroll
    for (int e = 0; e < 4; ++e) { xh[e] = 0.f; dy[e] = 0.f; }
  }
}

// ---- backward pass 1: Σ dy and Σ dy·xhat per column -> partials[blk][2C]
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const BnCommon a, const void* __restrict__ dout, int dout_f32,
                                                           float* __restrict__ partials) {
  extern __shared__ float red[];
  const int C = a.C, tpr = C >> 2, rpi = 256 / tpr;
  const int r0 = threadIdx.x / tpr, c4 = (threadIdx.x % tpr) * 4;
  const int per = (a.rows + gridDim.x - 1) / gridDim.x;
  const int rb = blockIdx.x * per;
  const int re = min(rb + per, a.rows);
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (r0 < rpi)
    for (int r = rb + r0; r < re; r += rpi) {
      float xh[4], dy[4];
      bn_dy(a, dout, dout_f32, r, c4, (int64_t)r * tpr + (c4 >> 2), thr, scale, xh, dy);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += dy[e]; q[e] += dy[e] * xh[e]; }
    }
  if (r0 < rpi) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[r0 * C + c4 + e] = s[e]; red[rpi * C + r0 * C + c4 + e] = q[e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float x1 = 0.f, x2 = 0.f;
    for (int k = 0; k < rpi; ++k) { x1 += red[k * C + c]; x2 += red[rpi * C + k * C + c]; }
    partials[(int64_t)blockIdx.x * 2 * C + c] = x1;
    partials[(int64_t)blockIdx.x * 2 * C + C + c] = x2;
  }
}

// ---- backward pass 2: dx = gamma * rstd * (dy − mean(dy) − xhat * mean(dy·xhat)); block 0 adds dgamma/dbeta
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnCommon a, const void* __restrict__ dout, int dout_f32,
                                                           const float* __restrict__ sums, bf16_t* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  const int tpr = a.C >> 2;
  const int64_t n = (int64_t)a.rows * tpr;
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const float inv_n = 1.f / bn_rows(a.frame_limit, a.seg_len, a.rows);
  if (blockIdx.x == 0 && dgamma)
    for (int c = threadIdx.x; c < a.C; c += 256) {
      dbeta[c] = accumulate ? dbeta[c] + sums[c] : sums[c];
      dgamma[c] = accumulate ? dgamma[c] + sums[a.C + c] : sums[a.C + c];
    }
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / tpr), c4 = (int)(i - (int64_t)r * tpr) * 4;
    float xh[4], dy[4];
    bn_dy(a, dout, dout_f32, r, c4, i, thr, scale, xh, dy);
    const f32x4 g = *(const f32x4*)(a.gamma + c4), rs = *(const f32x4*)(a.rstd + c4);
    const f32x4 s1 = *(const f32x4*)(sums + c4), s2 = *(const f32x4*)(sums + a.C + c4);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] * rs[e] * (dy[e] - s1[e] * inv_n - xh[e] * s2[e] * inv_n);
    if (!bn_live(a.frame_limit, a.seg_len, r)) o[0] = o[1] = o[2] = o[3] = 0.f;
    *(uint2*)(dx + (int64_t)r * a.C + c4) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
  }
}


// ================================================================================================================================
// Training path in two launches per direction (was three): a workgroup owns a SLAB of channels (64, or all of them when C is not a
// multiple of 64) and a chunk of rows, so the number of partial rows (<= 64 chunks) is independent of the number of workgroups, and
// the kernel that applies the normalisation sums those few partial rows for its own slab itself — in a fixed order, in double:
// every workgroup of a slab computes bit-identical statistics — instead of waiting for a finalize launch.
constexpr int BN_MAX_CHUNKS = 64;
__host__ __device__ inline int bn_slab(int C) { return (C % 64 == 0) ? 64 : C; }

struct BnSlab {
  int c0, slab, tpr, rpi, r_in, cq, rb, re;
  __device__ BnSlab(int C, int rows) {
    slab = bn_slab(C);
    c0 = blockIdx.x * slab;
    tpr = slab >> 2;
    rpi = 256 / tpr;
    r_in = threadIdx.x / tpr;
    cq = threadIdx.x % tpr;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    rb = blockIdx.y * per;
    re = min(rb + per, rows);
  }
};

// per-thread sums over the workgroup's rows -> one partial row segment per chunk: partials[chunk][c0 + ..] | partials[chunk][C + c0 + ..]
__device__ __forceinline__ void bn_slab_store(const BnSlab& g, int C, const float s[4], const float q[4], float* red, float* partials) {
  if (g.r_in < g.rpi) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[g.r_in * g.slab + g.cq * 4 + e] = s[e]; red[(g.rpi + g.r_in) * g.slab + g.cq * 4 + e] = q[e]; }
  }
  __syncthreads();
  for (int v = threadIdx.x; v < 2 * g.slab; v += 256) {
    const int half = v >= g.slab, c = v - half * g.slab;
    float acc = 0.f;
    for (int k = 0; k < g.rpi; ++k) acc += red[(half * g.rpi + k) * g.slab + c];
    partials[(int64_t)blockIdx.y * 2 * C + half * C + g.c0 + c] = acc;
  }
}

// tot[v], v < 2*slab: the column sums of the partial rows for this workgroup's slab (sum | second sum), double, fixed order.
// A thread owns four columns and every (256 / (slab/2))-th partial row: with <= 64 partial rows that is at most eight 16-byte loads,
// all in flight at once (the partial rows come from another XCD's kernel: each dependent batch of loads costs ~2 us here).
constexpr int BN_TOT = 256 + 1024;
__device__ __forceinline__ void bn_slab_totals(const float* __restrict__ partials, int nblk, int C, int c0, int slab, double* tot /* [BN_TOT] */) {
  const int nq = slab >> 1, groups = 256 / nq;          // column quads of (sum | second sum); row groups
  const int t = threadIdx.x;
  if (t < groups * nq) {
    const int q = t % nq, g0 = t / nq;
    const int v0 = q * 4;
    const int col = v0 < slab ? c0 + v0 : C + c0 + (v0 - slab);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    // sixteen loads in flight: the 112 partial rows the PostNet's convs emit (16 utterances x 7 tiles) are one batch, not two
    for (int b = g0; b < nblk; b += 16 * groups) {
      f32x4 f[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int bb = b + u * groups;
        f[u] = *(const f32x4*)(partials + (int64_t)min(bb, nblk - 1) * 2 * C + col);
        if (bb >= nblk) f[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += f[u][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tot[256 + t * 4 + e] = a[e];
  }
  __syncthreads();
  if (t < 2 * slab) {
    double sum = 0.0;
    for (int k = 0; k < groups; ++k) sum += tot[256 + (k * nq + (t >> 2)) * 4 + (t & 3)];
    tot[t] = sum;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void bn_stats2_kernel(const void* __restrict__ x, int x_f32, int rows, int C, float* __restrict__ partials,
                                                        const int* __restrict__ frame_limit, int seg_len) {
  extern __shared__ float red[];  // [2][rpi][slab]
  const BnSlab g(C, rows);
  const int c4 = g.c0 + g.cq * 4;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (g.r_in < g.rpi)
    for (int r0 = g.rb + g.r_in; r0 < g.re; r0 += 8 * g.rpi) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // eight independent row loads in flight
        const int r = r0 + u * g.rpi, rc = min(r, g.re - 1);
        ldx4(x, x_f32, (int64_t)rc * C + c4, v[u]);
        if (r >= g.re || !bn_live(frame_limit, seg_len, rc)) v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v[u][e]; q[e] += v[u][e] * v[u][e]; }
    }
  bn_slab_store(g, C, s, q, red, partials);
}

struct BnTrain {
  const float* partials;
  int nblk;
  float eps, momentum;
  float* mean;
  float* rstd;
  float* run_mean;
  float* run_var;
  long long* nbt;
};

// mean / rstd of the slab from the partial rows (the row-chunk-0 workgroups also publish them and update the running statistics),
// then y = tanh?(gamma * xhat + beta) -> dropout -> (+ residual) -> out for the workgroup's rows
// (Held to 128 registers: the training shape's 848 workgroups then fit the chip's 1,024 slots in ONE round — at the 130-136 the compiler took
// unbounded, 768 slots left 80 workgroups for a second round of the same length.  RESID: the residual rows' 16 registers only where there
// is a residual, the PostNet's last layer.)
template <bool RESID>
__global__ __launch_bounds__(256, RESID ? 3 : 4) void bn_apply2_kernel(const BnCommon a, const BnTrain t, const float* __restrict__ resid,
                                                        bf16_t* __restrict__ out16, float* __restrict__ out32) {
  __shared__ double tot[BN_TOT];
  __shared__ float ms[128], rs[128];
  const BnSlab g(a.C, a.rows);
  const int cl = g.cq * 4, c4 = g.c0 + cl, tprC = a.C >> 2;
  const bool act = g.r_in < g.rpi;
  // the workgroup's (at most four) rows are fetched before the partial rows are summed: one memory latency, not two
  float v[4][4], rr[RESID ? 4 : 1][4];
  bool live[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = g.rb + g.r_in + u * g.rpi, rc = max(min(r, g.re - 1), 0);
    live[u] = act && r < g.re && bn_live(a.frame_limit, a.seg_len, r);
    if (act) {
      ldx4(a.x, a.x_f32, (int64_t)rc * a.C + c4, v[u]);
      if constexpr (RESID) { const f32x4 q = *(const f32x4*)(resid + (int64_t)rc * a.C + c4); rr[u][0] = q[0]; rr[u][1] = q[1]; rr[u][2] = q[2]; rr[u][3] = q[3]; }
    }
  }
  f32x4 gm = f32x4{0.f, 0.f, 0.f, 0.f}, bt = gm;
  if (act) { gm = *(const f32x4*)(a.gamma + c4); bt = *(const f32x4*)(a.beta + c4); }
  bn_slab_totals(t.partials, t.nblk, a.C, g.c0, g.slab, tot);
  if ((int)threadIdx.x < g.slab) {
    const int c = threadIdx.x;
    const int n = bn_rows(a.frame_limit, a.seg_len, a.rows);
    const double m = tot[c] / n;
    double var = tot[g.slab + c] / n - m * m;
    if (var < 0.0) var = 0.0;
    ms[c] = (float)m;
    rs[c] = (float)(1.0 / sqrt(var + (double)t.eps));
    if (blockIdx.y == 0) {
      t.mean[g.c0 + c] = ms[c];
      t.rstd[g.c0 + c] = rs[c];
      if (t.run_mean) {
        t.run_mean[g.c0 + c] = (1.f - t.momentum) * t.run_mean[g.c0 + c] + t.momentum * (float)m;
        t.run_var[g.c0 + c] = (1.f - t.momentum) * t.run_var[g.c0 + c] + t.momentum * (float)(var * n / (n > 1 ? n - 1 : 1));
      }
      if (blockIdx.x == 0 && c == 0 && t.nbt) t.nbt[0] += 1;
    }
  }
  __syncthreads();
  if (!act) return;
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = g.rb + g.r_in + u * g.rpi;
    if (r >= g.re) break;
    const int64_t i = (int64_t)r * tprC + (c4 >> 2);
    if (!live[u]) {     // a frame past the batch's own length: a zero row (the next conv's padding)
      if (out16) *(uint2*)(out16 + (int64_t)r * a.C + c4) = make_uint2(0u, 0u);
      if (out32) *(f32x4*)(out32 + (int64_t)r * a.C + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.keep) a.keep[i] = 0;
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[u][e] = (v[u][e] - ms[cl + e]) * rs[cl + e] * gm[e] + bt[e];
      if (a.use_tanh) v[u][e] = tanh_fast(v[u][e]);
    }
    if (a.p > 0.f) {
      const uint4 bb = bits4(a.rng, a.site, (unsigned)i);
      const bool k0 = bb.x >= thr, k1 = bb.y >= thr, k2 = bb.z >= thr, k3 = bb.w >= thr;
      v[u][0] = k0 ? v[u][0] * scale : 0.f; v[u][1] = k1 ? v[u][1] * scale : 0.f;
      v[u][2] = k2 ? v[u][2] * scale : 0.f; v[u][3] = k3 ? v[u][3] * scale : 0.f;
      if (a.keep) a.keep[i] = (uint8_t)((int)k0 | ((int)k1 << 1) | ((int)k2 << 2) | ((int)k3 << 3));
    }
    if constexpr (RESID) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] += rr[u][e];
    }
    if (out16) *(uint2*)(out16 + (int64_t)r * a.C + c4) = make_uint2(pack_bf2(v[u][0], v[u][1]), pack_bf2(v[u][2], v[u][3]));
    if (out32) *(f32x4*)(out32 + (int64_t)r * a.C + c4) = f32x4{v[u][0], v[u][1], v[u][2], v[u][3]};
  }
}

__global__ __launch_bounds__(256) void bn_bwd_stats2_kernel(const BnCommon a, const void* __restrict__ dout, int dout_f32,
                                                            float* __restrict__ partials) {
  extern __shared__ float red[];
  const BnSlab g(a.C, a.rows);
  const int c4 = g.c0 + g.cq * 4, tprC = a.C >> 2;
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (g.r_in < g.rpi)
    for (int r0 = g.rb + g.r_in; r0 < g.re; r0 += 8 * g.rpi) {
      float xh[8][4], dy[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // eight independent rows in flight (a chunk of a full-size batch is seven)
        const int r = r0 + u * g.rpi, rc = min(r, g.re - 1);
        bn_dy(a, dout, dout_f32, rc, c4, (int64_t)rc * tprC + (c4 >> 2), thr, scale, xh[u], dy[u]);
        if (r >= g.re) { xh[u][0] = xh[u][1] = xh[u][2] = xh[u][3] = 0.f; dy[u][0] = dy[u][1] = dy[u][2] = dy[u][3] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += dy[u][e]; q[e] += dy[u][e] * xh[u][e]; }
    }
  bn_slab_store(g, a.C, s, q, red, partials);
}

// dx = gamma * rstd * (dy − mean(dy) − xhat * mean(dy·xhat)) with the two sums taken from the partial rows; the row-chunk-0
// workgroups add them to dbeta / dgamma
__global__ __launch_bounds__(256, 4) void bn_bwd_apply2_kernel(const BnCommon a, const void* __restrict__ dout, int dout_f32,
                                                            const float* __restrict__ partials, int nblk, bf16_t* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  __shared__ double tot[BN_TOT];
  __shared__ float s1s[128], s2s[128];
  const BnSlab g(a.C, a.rows);
  const int cl = g.cq * 4, c4 = g.c0 + cl, tprC = a.C >> 2;
  const bool act = g.r_in < g.rpi;
  const unsigned thr = keep_threshold(a.p);
  const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  // dy and xhat of the workgroup's (at most four) rows need nothing from the partial rows: computed first, so that their loads
  // are in flight while the partial rows are summed
  float xh[4][4], dy[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rc = max(min(g.rb + g.r_in + u * g.rpi, g.re - 1), 0);
    if (act) bn_dy(a, dout, dout_f32, rc, c4, (int64_t)rc * tprC + (c4 >> 2), thr, scale, xh[u], dy[u]);
  }
  bn_slab_totals(partials, nblk, a.C, g.c0, g.slab, tot);
  if ((int)threadIdx.x < g.slab) {
    const int c = threadIdx.x;
    s1s[c] = (float)tot[c];
    s2s[c] = (float)tot[g.slab + c];
    if (blockIdx.y == 0 && dgamma) {
      dbeta[g.c0 + c] = accumulate ? dbeta[g.c0 + c] + s1s[c] : s1s[c];
      dgamma[g.c0 + c] = accumulate ? dgamma[g.c0 + c] + s2s[c] : s2s[c];
    }
  }
  __syncthreads();
  if (!act) return;
  const float inv_n = 1.f / bn_rows(a.frame_limit, a.seg_len, a.rows);
  const f32x4 gm = *(const f32x4*)(a.gamma + c4), rsd = *(const f32x4*)(a.rstd + c4);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = g.rb + g.r_in + u * g.rpi;
    if (r >= g.re) break;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gm[e] * rsd[e] * (dy[u][e] - s1s[cl + e] * inv_n - xh[u][e] * s2s[cl + e] * inv_n);
    if (!bn_live(a.frame_limit, a.seg_len, r)) o[0] = o[1] = o[2] = o[3] = 0.f;
    *(uint2*)(dx + (int64_t)r * a.C + c4) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
  }
}

inline int bn_chunks(int rows) { int n = (rows + 15) / 16; return n > BN_MAX_CHUNKS ? BN_MAX_CHUNKS : (n < 1 ? 1 : n); }
// row chunks of the applying kernels: about four row iterations per workgroup
inline int bn_apply_chunks(int rows, int C) { const int rpi = 256 / (bn_slab(C) >> 2); int n = (rows + 4 * rpi - 1) / (4 * rpi); return n < 1 ? 1 : n; }

// enough workgroups to cover the chip twice with few sequential (latency-bound) row iterations each
inline int bn_blocks(int rows) { int n = (rows + 15) / 16; return n > 1024 ? 1024 : (n < 1 ? 1 : n); }

}  // namespace

extern "C" int ttsk_bn_nblocks(int rows) { return bn_blocks(rows); }

static int bn_check(int rows, int C) {
  TTSK_REQUIRE(rows > 0 && C >= 4 && C <= 1024 && (C & 3) == 0, "batchnorm: C must be a multiple of 4, <= 1024 (got %d)", C);
  return 0;
}

static int limit_check(const int32_t* frame_limit, int seg_len, int rows) {
  TTSK_REQUIRE(!frame_limit || (seg_len > 0 && rows % seg_len == 0), "batchnorm: frame_limit needs rows %% seg_len == 0");
  return 0;
}

extern "C" int ttsk_bn_stats(const void* x, int x_is_f32, int rows, int C, float* partials, const int32_t* frame_limit, int seg_len,
                             void* stream) {
  TTSK_REQUIRE(x && partials, "bn_stats: null pointer");
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  if (int rc = bn_check(rows, C)) return rc;
  const int rpi = 256 / (C >> 2);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(bn_blocks(rows)), dim3(256), 2 * rpi * C * sizeof(float), (hipStream_t)stream,
                     x, x_is_f32, rows, C, partials, frame_limit, seg_len > 0 ? seg_len : 1);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_finalize(const float* partials, int nblk, int C, int rows, float eps, float momentum, float* mean,
                                float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(partials && mean && rstd && nblk > 0, "bn_finalize: null pointer");
  TTSK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nblk, C, rows, eps,
                     momentum, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked, frame_limit, seg_len > 0 ? seg_len : 1);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_rsqrt_eps(const float* var, float eps, float* rstd, int n, void* stream) {
  TTSK_REQUIRE(var && rstd && n > 0, "rsqrt_eps: bad arguments");
  hipLaunchKernelGGL(rsqrt_eps_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, var, eps, rstd, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_apply(const void* x, int x_is_f32, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             int rows, int C, int use_tanh, float p, uint32_t site, const uint64_t* rng, const float* resid_f32,
                             void* out_bf16, float* out_f32, const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(x && mean && rstd && gamma && beta && (out_bf16 || out_f32), "bn_apply: null pointer");
  TTSK_REQUIRE(p == 0.f || rng, "bn_apply: dropout needs rng");
  if (int rc = bn_check(rows, C)) return rc;
  BnCommon a{x, mean, rstd, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1};
  int64_t n = (int64_t)rows * (C >> 2);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, resid_f32, (bf16_t*)out_bf16, out_f32);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_bwd_stats(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                                 const uint64_t* rng, float* partials, const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(dout && x && mean && rstd && gamma && beta && partials, "bn_bwd_stats: null pointer");
  TTSK_REQUIRE(p == 0.f || rng, "bn_bwd_stats: dropout needs rng");
  if (int rc = bn_check(rows, C)) return rc;
  BnCommon a{x, mean, rstd, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1};
  const int rpi = 256 / (C >> 2);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(bn_blocks(rows)), dim3(256), 2 * rpi * C * sizeof(float), (hipStream_t)stream, a,
                     dout, dout_is_f32, partials);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_bwd_apply(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                                 const uint64_t* rng, const float* sums, void* dx_bf16, float* dgamma, float* dbeta, int accumulate,
                                 const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(dout && x && mean && rstd && gamma && beta && sums && dx_bf16, "bn_bwd_apply: null pointer");
  TTSK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_apply: dgamma/dbeta come in pairs");
  if (int rc = bn_check(rows, C)) return rc;
  BnCommon a{x, mean, rstd, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1};
  int64_t n = (int64_t)rows * (C >> 2);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, dout, dout_is_f32, sums,
                     (bf16_t*)dx_bf16, dgamma, dbeta, accumulate);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ---- the two-launch training path
static int bn2_check(int rows, int C) {
  if (int rc = bn_check(rows, C)) return rc;
  TTSK_REQUIRE(C % 64 == 0 || C <= 128, "batchnorm (slab path): C must be a multiple of 64 or <= 128 (got %d)", C);
  return 0;
}

extern "C" int ttsk_bn_nchunks(int rows) { return bn_chunks(rows); }

extern "C" int ttsk_bn_stats_slab(const void* x, int x_is_f32, int rows, int C, float* partials, const int32_t* frame_limit, int seg_len,
                                  void* stream) {
  TTSK_REQUIRE(x && partials, "bn_stats_slab: null pointer");
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  if (int rc = bn2_check(rows, C)) return rc;
  const int slab = bn_slab(C), rpi = 256 / (slab >> 2);
  hipLaunchKernelGGL(bn_stats2_kernel, dim3(C / slab, bn_chunks(rows)), dim3(256), 2 * rpi * slab * sizeof(float), (hipStream_t)stream,
                     x, x_is_f32, rows, C, partials, frame_limit, seg_len > 0 ? seg_len : 1);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_train_apply(const void* x, int x_is_f32, const float* partials, int nblk, float eps, float momentum, float* mean,
                                   float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                                   const uint64_t* rng, const float* resid_f32, void* out_bf16, float* out_f32, uint8_t* keep_out,
                                   const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(x && partials && nblk > 0 && mean && rstd && gamma && beta && (out_bf16 || out_f32), "bn_train_apply: null pointer");
  TTSK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_apply: running stats come in pairs");
  TTSK_REQUIRE(p == 0.f || rng, "bn_train_apply: dropout needs rng");
  if (int rc = bn2_check(rows, C)) return rc;
  BnCommon a{x, nullptr, nullptr, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1, keep_out};
  BnTrain t{partials, nblk, eps, momentum, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked};
  const dim3 grid(C / bn_slab(C), bn_apply_chunks(rows, C));
  if (resid_f32) hipLaunchKernelGGL(bn_apply2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, t, resid_f32, (bf16_t*)out_bf16, out_f32);
  else hipLaunchKernelGGL(bn_apply2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, t, resid_f32, (bf16_t*)out_bf16, out_f32);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_bwd_stats_slab(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                                      const uint64_t* rng, const uint8_t* keep, float* partials, const int32_t* frame_limit, int seg_len,
                                      void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(dout && x && mean && rstd && gamma && beta && partials, "bn_bwd_stats_slab: null pointer");
  TTSK_REQUIRE(p == 0.f || rng || keep, "bn_bwd_stats_slab: dropout needs rng or the forward's keep bits");
  if (int rc = bn2_check(rows, C)) return rc;
  BnCommon a{x, mean, rstd, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1, (uint8_t*)keep};
  const int slab = bn_slab(C), rpi = 256 / (slab >> 2);
  hipLaunchKernelGGL(bn_bwd_stats2_kernel, dim3(C / slab, bn_chunks(rows)), dim3(256), 2 * rpi * slab * sizeof(float), (hipStream_t)stream,
                     a, dout, dout_is_f32, partials);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bn_bwd_apply_slab(const void* dout, int dout_is_f32, const void* x, int x_is_f32, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int rows, int C, int use_tanh, float p, uint32_t site,
                                      const uint64_t* rng, const uint8_t* keep, const float* partials, int nblk, void* dx_bf16, float* dgamma,
                                      float* dbeta, int accumulate, const int32_t* frame_limit, int seg_len, void* stream) {
  if (int rc = limit_check(frame_limit, seg_len, rows)) return rc;
  TTSK_REQUIRE(dout && x && mean && rstd && gamma && beta && partials && nblk > 0 && dx_bf16, "bn_bwd_apply_slab: null pointer");
  TTSK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_apply_slab: dgamma/dbeta come in pairs");
  TTSK_REQUIRE(p == 0.f || rng || keep, "bn_bwd_apply_slab: dropout needs rng or the forward's keep bits");
  if (int rc = bn2_check(rows, C)) return rc;
  BnCommon a{x, mean, rstd, gamma, beta, rng, rows, C, use_tanh, p, site, x_is_f32, frame_limit, seg_len > 0 ? seg_len : 1, (uint8_t*)keep};
  hipLaunchKernelGGL(bn_bwd_apply2_kernel, dim3(C / bn_slab(C), bn_apply_chunks(rows, C)), dim3(256), 0, (hipStream_t)stream, a, dout,
                     dout_is_f32, partials, nblk, (bf16_t*)dx_bf16, dgamma, dbeta, accumulate);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
