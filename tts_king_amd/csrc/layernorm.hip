// layernorm.hip — fused [dropout] + [residual] + LayerNorm + [dropout] + [PAD-row zeroing] + [Linear(D,1) head]
// and its backward.  One wavefront per row, 4 contiguous channels per lane (8-byte bf16 accesses).
// reference: SubLayers.py:62-63 (MHA: LN(dropout(fc)+residual)), SubLayers.py:99-101 (FFN), Layers.py:29-32
// (masked_fill of PAD rows), model/modules.py:270-309 (VariancePredictor: LN -> Dropout, final Linear + mask).
#include "common.h"
#include "proj32.h"

namespace {

constexpr int MAXJ = 4;  // D <= 1024

struct LnArgs {
  const bf16_t* y;        // [rows][D] main input (sub-layer output incl. bias)
  const bf16_t* res;      // [rows][D] residual or null
  const float* gamma;
  const float* beta;
  bf16_t* out;            // [rows][D]
  bf16_t* z_save;         // [rows][D] LN input (after dropout+residual), null = do not save
  float* mean;            // [rows]
  float* rstd;            // [rows]
  const long long* lens;  // [B] valid length per segment, null = no masking
  const uint64_t* rng;    // {seed, step}
  const float* head_w;    // [D] or null
  const float* head_b;    // [1]
  float* head_out;        // [rows]
  int rows, D, seg_len;
  float p_pre, p_post, eps;
  unsigned site_pre, site_post;
  // grouped launch (the three VariancePredictors of a training step as one launch): rows = groups * group_rows; group g
  // uses gamma / beta / head_w / head_b at + g * pstride floats and dropout sites + g * site_stride; PAD masks and dropout
  // element indices are taken inside the group, so a grouped launch equals `groups` single launches bit for bit.
  int group_rows;         // 0 = one group
  long long pstride;
  unsigned site_stride;
};

__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
  const uint2 u = *(const uint2*)p;
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
  *(uint2*)p = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
__device__ __forceinline__ void drop4(float v[4], uint64_t seed, uint64_t step, unsigned site, unsigned e4, unsigned thr, float scale) {
  const uint4 b = Philox::gen(make_uint2((unsigned)seed, (unsigned)(seed >> 32)),
                              make_uint4(e4, site, (unsigned)step, (unsigned)(step >> 32)));
  v[0] = b.x >= thr ? v[0] * scale : 0.f;
  v[1] = b.y >= thr ? v[1] * scale : 0.f;
  v[2] = b.z >= thr ? v[2] * scale : 0.f;
  v[3] = b.w >= thr ? v[3] * scale : 0.f;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int D = a.D, nj = D >> 8;
  const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
  const int grp = a.group_rows > 0 ? row / a.group_rows : 0;
  const int lrow = row - grp * a.group_rows;                       // row inside its group (== row when ungrouped)
  const float* gamma = a.gamma + grp * a.pstride;
  const float* beta = a.beta + grp * a.pstride;
  const unsigned site_pre = a.site_pre + grp * a.site_stride, site_post = a.site_post + grp * a.site_stride;
  float z[MAXJ][4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    if (j < nj) {
      const int c = j * 256 + lane * 4;
      load4(a.y + (int64_t)row * D + c, z[j]);
      if (a.p_pre > 0.f)
        drop4(z[j], seed, step, site_pre, (unsigned)(((int64_t)lrow * D + c) >> 2), keep_threshold(a.p_pre), 1.f / (1.f - a.p_pre));
      if (a.res) {
        float r[4];
        load4(a.res + (int64_t)row * D + c, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[j][e] += r[e];
      }
      if (a.z_save) {  // backward sees exactly the rounded value
        store4(a.z_save + (int64_t)row * D + c, z[j]);
      }
      s += z[j][0] + z[j][1] + z[j][2] + z[j][3];
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
    if (j < nj)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = z[j][e] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / D + a.eps);
  if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
  bool masked = false;
  if (a.lens) { const int b = lrow / a.seg_len, t = lrow - b * a.seg_len; masked = t >= a.lens[b]; }
  float hs = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    if (j < nj) {
      const int c = j * 256 + lane * 4;
      const f32x4 g = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (z[j][e] - mean) * rstd * g[e] + bt[e];
      if (a.p_post > 0.f)
        drop4(o, seed, step, site_post, (unsigned)(((int64_t)lrow * D + c) >> 2), keep_threshold(a.p_post), 1.f / (1.f - a.p_post));
      if (masked) o[0] = o[1] = o[2] = o[3] = 0.f;
      if (a.out) store4(a.out + (int64_t)row * D + c, o);
      if (a.head_w) {
        const f32x4 w = *(const f32x4*)(a.head_w + grp * a.pstride + c);
        hs += o[0] * w[0] + o[1] * w[1] + o[2] * w[2] + o[3] * w[3];
      }
    }
  }
  if (a.head_w) {
    hs = wave_sum(hs);
    if (lane == 0) a.head_out[row] = masked ? 0.f : hs + a.head_b[grp * a.pstride];
  }
}

struct LnBwdArgs {
  const bf16_t* dout;     // [rows][D] grad of the LN output, or null when the head is the only consumer
  const float* dhead;     // [rows] grad of the head output (head mode)
  const float* head_w;    // [D]
  const bf16_t* z;        // [rows][D] saved LN input
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;      // needed only in head mode / post dropout (recompute of the LN output)
  const long long* lens;
  const uint64_t* rng;
  bf16_t* dz;             // [rows][D] grad wrt LN input (the residual branch's grad)
  bf16_t* dy;             // [rows][D] grad wrt the sub-layer output (dz with the pre-dropout mask); null if p_pre == 0
  float* partials;        // [nblk][nq*D (+1)]: dbias | dgamma | dbeta (| dhead_w | dhead_b): the order of the flat gradient buffer
  int rows, D, seg_len, relu_in;
  float p_pre, p_post;
  unsigned site_pre, site_post;
  int group_rows, nblk_group;     // grouped launch (see LnArgs): gridDim.x = groups * nblk_group, partials [group][nblk_group][ncol]
  long long pstride;
  unsigned site_stride;
  // upstream gradient still in split-K form (ttsk_layernorm_bwd_slabs): dout[row] = sum_s slabs[s * slab_stride + row*D ..] + R[row]
  const float* slabs;
  const bf16_t* R;
  long long slab_stride;
  int nsplit;
};

// 8 waves per workgroup, one row per wave at a time: the per-row chain (loads -> two wave reductions -> stores) is pure
// latency, so the rows in flight per CU set the rate (4 waves: 11 us for 6768 x 256; the partial count stays 256 blocks)
// MJ = D / 256 rounded up to the instance (1: the predictors' 256-channel LayerNorms — 80 registers instead of the 231 the D <= 1024
// instance carries in per-lane column sums, so that three of these 8-wave workgroups share a CU with whatever runs beside the predictors'
// backward on the other stream).
constexpr int LNB_WAVES = 8;
template <int MJ>
__global__ __launch_bounds__(LNB_WAVES * 64) void ln_bwd_kernel(const LnBwdArgs a) {
  __shared__ float red[LNB_WAVES][MJ * 256];
  __shared__ float redb[LNB_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D, nj = D >> 8;
  const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
  const bool head = a.dhead != nullptr;
  float dg[MJ][4], db[MJ][4], dbias[MJ][4], dhw[MJ][4];
  float dhb = 0.f;
#pragma unroll
  for (int j = 0; j < MJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[j][e] = db[j][e] = dbias[j][e] = dhw[j][e] = 0.f;

  const int grp = a.group_rows > 0 ? blockIdx.x / a.nblk_group : 0;
  const int lblk = blockIdx.x - grp * a.nblk_group;
  const int grows = a.group_rows > 0 ? a.group_rows : a.rows;
  const float* gamma = a.gamma + grp * a.pstride;
  const float* beta = a.beta ? a.beta + grp * a.pstride : nullptr;
  const float* head_w = a.head_w ? a.head_w + grp * a.pstride : nullptr;
  const unsigned site_pre = a.site_pre + grp * a.site_stride, site_post = a.site_post + grp * a.site_stride;
  // the loads of a wave's NEXT row are issued before the two wave reductions of the current one (the per-row chain
  // loads -> reductions -> stores is latency; one row at a time took 12.5 us for 6768 x 256)
  struct RowIn { uint2 z[MJ]; f32x4 d[MJ]; float mean, rstd, dh; bool masked; };
  auto fetch = [&](int lr, RowIn& r) __attribute__((always_inline)) {
    const int rw = grp * grows + lr;
    r.masked = false;
    if (a.lens) { const int b = lr / a.seg_len, t = lr - b * a.seg_len; r.masked = t >= a.lens[b]; }
    r.mean = a.mean[rw]; r.rstd = a.rstd[rw];
    r.dh = head ? (r.masked ? 0.f : a.dhead[rw]) : 0.f;
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (j < nj) {
        const int c = j * 256 + lane * 4;
        r.z[j] = *(const uint2*)(a.z + (int64_t)rw * D + c);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (!r.masked) {
          if (a.slabs) {
            const float* sp = a.slabs + (int64_t)rw * D + c;
            for (int q = 0; q < a.nsplit; ++q) d += *(const f32x4*)(sp + q * a.slab_stride);       // fixed order: deterministic
            if (a.R) {
              const uint2 u = *(const uint2*)(a.R + (int64_t)rw * D + c);
              d += f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
            }
          } else if (a.dout) {
            const uint2 u = *(const uint2*)(a.dout + (int64_t)rw * D + c);
            d = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
          }
        }
        r.d[j] = d;
      }
  };
  const int rstride = a.nblk_group * LNB_WAVES;
  RowIn cur;
  if (lblk * LNB_WAVES + wave < grows) fetch(lblk * LNB_WAVES + wave, cur);
  for (int lrow = lblk * LNB_WAVES + wave; lrow < grows; lrow += rstride) {
    const int row = grp * grows + lrow;
    RowIn nxt;
    const bool more = lrow + rstride < grows;
    if (more) fetch(lrow + rstride, nxt);
    const bool masked = cur.masked;
    const float mean = cur.mean, rstd = cur.rstd, dh = cur.dh;
    float xh[MJ][4], g[MJ][4];
    float c1 = 0.f, c2 = 0.f;
    unsigned closed = 0;  // bit j*4+e set: the ReLU that produced z was inactive (relu_in mode)
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      if (j < nj) {
        const int c = j * 256 + lane * 4;
        const float zz[4] = {__uint_as_float(cur.z[j].x << 16), __uint_as_float(cur.z[j].x & 0xFFFF0000u),
                             __uint_as_float(cur.z[j].y << 16), __uint_as_float(cur.z[j].y & 0xFFFF0000u)};
        float d[4] = {cur.d[j][0], cur.d[j][1], cur.d[j][2], cur.d[j][3]};
        const f32x4 gm = *(const f32x4*)(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[j][e] = (zz[e] - mean) * rstd;
        if (head) {
          const f32x4 w = *(const f32x4*)(head_w + c), bt = *(const f32x4*)(beta + c);
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = xh[j][e] * gm[e] + bt[e]; d[e] += dh * w[e]; }
          if (a.p_post > 0.f)
            drop4(o, seed, step, site_post, (unsigned)(((int64_t)lrow * D + c) >> 2), keep_threshold(a.p_post), 1.f / (1.f - a.p_post));
#pragma unroll
          for (int e = 0; e < 4; ++e) dhw[j][e] += dh * o[e];
        }
        if (a.p_post > 0.f)
          drop4(d, seed, step, site_post, (unsigned)(((int64_t)lrow * D + c) >> 2), keep_threshold(a.p_post), 1.f / (1.f - a.p_post));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dg[j][e] += d[e] * xh[j][e];
          db[j][e] += d[e];
          g[j][e] = d[e] * gm[e];
          c1 += g[j][e];
          c2 += g[j][e] * xh[j][e];
        }
        if (a.relu_in) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (!(zz[e] > 0.f)) closed |= 1u << (j * 4 + e);
        }
      }
    }
    if (lane == 0) dhb += dh;
    c1 = wave_sum(c1) / D;
    c2 = wave_sum(c2) / D;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      if (j < nj) {
        const int c = j * 256 + lane * 4;
        float dzv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dzv[e] = ((closed >> (j * 4 + e)) & 1u) ? 0.f : rstd * (g[j][e] - c1 - xh[j][e] * c2);
        }
        if (a.dz) store4(a.dz + (int64_t)row * D + c, dzv);
        if (a.p_pre > 0.f) {
          drop4(dzv, seed, step, site_pre, (unsigned)(((int64_t)lrow * D + c) >> 2), keep_threshold(a.p_pre), 1.f / (1.f - a.p_pre));
          if (a.dy) store4(a.dy + (int64_t)row * D + c, dzv);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dbias[j][e] += dzv[e];
      }
    }
    if (more) cur = nxt;
  }
  // cross-wave reduction of the per-lane column sums, one quantity at a time
  const int nq = head ? 4 : 3;
  float* P = a.partials + (int64_t)blockIdx.x * (nq * D + (head ? 1 : 0));
  for (int qn = 0; qn < nq; ++qn) {
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      if (j < nj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = qn == 0 ? dbias[j][e] : qn == 1 ? dg[j][e] : qn == 2 ? db[j][e] : dhw[j][e];
          red[wave][j * 256 + lane * 4 + e] = v;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += LNB_WAVES * 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) t += red[w][c];
      P[qn * D + c] = t;
    }
    __syncthreads();
  }
  if (head) {
    dhb = wave_sum(dhb);
    if (lane == 0) redb[wave] = dhb;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) t += redb[w];
      P[nq * D] = t;
    }
  }
}

// ---- D = 256 fast path of the backward (the FFT blocks' 20 LayerNorms per step: no head, no post-dropout, no ReLU input).
// A wave works on FOUR rows at a time: 16 lanes per row, 16 columns per lane (32-byte bf16 loads, 64-byte fp32 slab loads), the two
// row reductions are 4 shuffle steps inside the 16-lane group.  256 workgroups x 8 waves x 4 rows = 8192 rows per sweep: the
// decoder's 6768 rows are ONE sweep, every wave issues all its loads at once (the one-row-per-wave kernel above walked 3-4 rows
// per wave one latency chain after the other: 14-19 us per launch, x 22 launches per step).
__device__ __forceinline__ void unpack8(uint4 u, float* v) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xFFFF0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8f(const float* v) {
  return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ float group16_sum(float v) { return quad16_sum(v); }     // DPP, no LDS round trips (common.h)

// One row of the D = 256 backward for a 16-lane group (lane l of the group owns columns c0 = 16 l .. + 15): the row's contributions
// to dbias / dgamma / dbeta are ADDED to sbias / sgam / sbeta, dzv returns the gradient of the sub-layer output (dz with the
// pre-dropout mask applied; all zero for a PAD row or a row past the end).  dz and dy go to memory when the row exists.
__device__ __forceinline__ void lnb256_row(const LnBwdArgs& a, int row, int c0, const float gam[16], uint64_t seed, uint64_t step,
                                           unsigned thr, float dscale, float sbias[16], float sgam[16], float sbeta[16], float dzv[16],
                                           const float* dl = nullptr /* this row's upstream gradient, fp32 in LDS (+ R), instead of slabs / dout */) {
  constexpr int D = 256;
  const bool live = row < a.rows;
  bool masked = !live;
  if (live && a.lens) { const int b = row / a.seg_len, t = row - b * a.seg_len; masked = t >= a.lens[b]; }
  float zz[16], d[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { zz[e] = 0.f; d[e] = 0.f; }
  float mean = 0.f, rstd = 0.f;
  if (live) {
    const bf16_t* zp = a.z + (int64_t)row * D + c0;
    unpack8(*(const uint4*)zp, zz); unpack8(*(const uint4*)(zp + 8), zz + 8);
    mean = a.mean[row]; rstd = a.rstd[row];
    if (!masked) {
      if (dl || a.slabs) {
        if (dl) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 t = *(const f32x4*)(dl + c0 + 4 * i);
            d[4 * i] = t[0]; d[4 * i + 1] = t[1]; d[4 * i + 2] = t[2]; d[4 * i + 3] = t[3];
          }
        } else {
          const float* sp = a.slabs + (int64_t)row * D + c0;
          for (int q = 0; q < a.nsplit; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 t = *(const f32x4*)(sp + q * a.slab_stride + 4 * i);
              d[4 * i] += t[0]; d[4 * i + 1] += t[1]; d[4 * i + 2] += t[2]; d[4 * i + 3] += t[3];
            }
          }
        }
        if (a.R) {
          float r[16];
          const bf16_t* rp = a.R + (int64_t)row * D + c0;
          unpack8(*(const uint4*)rp, r); unpack8(*(const uint4*)(rp + 8), r + 8);
#pragma unroll
          for (int e = 0; e < 16; ++e) d[e] += r[e];
        }
      } else {
        const bf16_t* dp = a.dout + (int64_t)row * D + c0;
        unpack8(*(const uint4*)dp, d); unpack8(*(const uint4*)(dp + 8), d + 8);
      }
    }
  }
  // explicit roundings (no contraction left to the compiler): this function is inlined into two kernels, which must agree bit for bit
  float xh[16], g[16];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    xh[e] = __fmul_rn(__fsub_rn(zz[e], mean), rstd);
    g[e] = __fmul_rn(d[e], gam[e]);
    c1 = __fadd_rn(c1, g[e]); c2 = __fmaf_rn(g[e], xh[e], c2);
    sgam[e] = __fmaf_rn(d[e], xh[e], sgam[e]);
    sbeta[e] = __fadd_rn(sbeta[e], d[e]);
  }
  c1 = __fmul_rn(group16_sum(c1), 1.f / D);
  c2 = __fmul_rn(group16_sum(c2), 1.f / D);
#pragma unroll
  for (int e = 0; e < 16; ++e) dzv[e] = __fmul_rn(rstd, __fmaf_rn(-xh[e], c2, __fsub_rn(g[e], c1)));
  if (live && a.dz) {
    bf16_t* op = a.dz + (int64_t)row * D + c0;
    *(uint4*)op = pack8f(dzv); *(uint4*)(op + 8) = pack8f(dzv + 8);
  }
  if (a.p_pre > 0.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) drop4(dzv + 4 * i, seed, step, a.site_pre, (unsigned)(((int64_t)row * D + c0 + 4 * i) >> 2), thr, dscale);
    if (live && a.dy) {
      bf16_t* op = a.dy + (int64_t)row * D + c0;
      *(uint4*)op = pack8f(dzv); *(uint4*)(op + 8) = pack8f(dzv + 8);
    }
  }
  if (live) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sbias[e] += dzv[e];
  }
}

// v summed over the wave's four 16-lane rows (lanes l, l+16, l+32, l+48), the total in every one of them, as (row0 + row1) + (row2 +
// row3): gfx950's v_permlane16_swap / v_permlane32_swap exchange rows between two registers on the VALU.  The 96 ds_bpermute
// (__shfl_xor 16, 32) they replace went through the LDS crossbar of all eight waves at once: 3.1 us of a 23 us workgroup
// (tools/debug/lnb_stamps.py).
__device__ __forceinline__ float rows4_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // r[0] rows (0,0,2,2), r[1] rows (1,1,3,3)
  const unsigned w = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);       // q[0] = lower half twice, q[1] = upper half twice
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// the four row groups of a wave, then the waves, in a fixed order (deterministic) -> partials[blockIdx.x][dbias | dgamma | dbeta]
// LDS image: column c = 16 l + e of each 256-column array sits at e * 17 + l (LNB_RED floats per wave).  Lane l owns 16 CONSECUTIVE
// columns, so in column order the 16 lanes of a store hit two banks (stride 16 floats: 28-38 % of this kernel family's LDS cycles were
// bank conflicts, profiles/r03_mfma_util.json); here a store's lanes are consecutive floats, and the read-back (lane = column) walks
// e * 17 + l with 17 odd: conflict-free both ways.  After rows4_sum every row group holds the totals: group g stores e = 4 g .. 4 g + 3.
constexpr int LNB_RED_LD = 16 * 17, LNB_RED = 3 * LNB_RED_LD;
__device__ __forceinline__ void lnb256_partials(const LnBwdArgs& a, float sbias[16], float sgam[16], float sbeta[16], float (*red)[LNB_RED],
                                                int wave, int grp, int c0, int pblock = -1, bool write = true) {
  constexpr int D = 256;
#pragma unroll
  for (int e = 0; e < 16; ++e) { sbias[e] = rows4_sum(sbias[e]); sgam[e] = rows4_sum(sgam[e]); sbeta[e] = rows4_sum(sbeta[e]); }
  const int l = c0 >> 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (grp == g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * g + j;          // (compile-time register index inside the unrolled g / j loops)
        red[wave][e * 17 + l] = sbias[e]; red[wave][LNB_RED_LD + e * 17 + l] = sgam[e]; red[wave][2 * LNB_RED_LD + e * 17 + l] = sbeta[e];
      }
    }
  }
  __syncthreads();
  if (!write) return;                      // (a workgroup that only replicates the rows for its share of the projection: SPLIT)
  float* P = a.partials + (int64_t)(pblock < 0 ? (int)blockIdx.x : pblock) * 3 * D;
  for (int c = threadIdx.x; c < 3 * D; c += LNB_WAVES * 64) {
    const int pos = (c >> 8) * LNB_RED_LD + (c & 15) * 17 + ((c & 255) >> 4);
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) t += red[w][pos];
    P[c] = t;
  }
}

__global__ __launch_bounds__(LNB_WAVES * 64) void ln_bwd256_kernel(const LnBwdArgs a) {
  constexpr int D = 256;
  __shared__ float red[LNB_WAVES][LNB_RED];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, l = lane & 15, c0 = l * 16;
  const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
  const unsigned thr = keep_threshold(a.p_pre);
  const float dscale = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
  float gam[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const f32x4 t = *(const f32x4*)(a.gamma + c0 + 4 * i); gam[4 * i] = t[0]; gam[4 * i + 1] = t[1]; gam[4 * i + 2] = t[2]; gam[4 * i + 3] = t[3]; }
  float sbias[16], sgam[16], sbeta[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sbias[e] = sgam[e] = sbeta[e] = 0.f;
  const int rows_per_sweep = gridDim.x * LNB_WAVES * 4;
  // (a.rows + 3: lanes of a partly filled wave still reach the shuffles)
  for (int row = (blockIdx.x * LNB_WAVES + wave) * 4 + grp; row < a.rows + 3; row += rows_per_sweep) {
    float dzv[16];
    lnb256_row(a, row, c0, gam, seed, step, thr, dscale, sbias, sgam, sbeta, dzv);
  }
  lnb256_partials(a, sbias, sgam, sbeta, red, wave, grp, c0);
}

// dtile[32][PRE_DT_LD] (fp32, LDS) = x[m0 .. m0+32)[768] · W', W = the ttsk_win_conv pack of a transposed (768, 1, 256) weight: the input
// gradient of a q|k|v projection for 32 rows.  512 threads; `as` = 32 x PRE_RS bytes of LDS for the rows of x (dead on return).
// Ends with a barrier: dtile is complete and `as` may be overwritten.
constexpr int PRE_K = 768, PRE_RS = PRE_K * 2 + 32, PRE_DT_LD = 260;
__device__ __forceinline__ void pre768_gemm(const bf16_t* __restrict__ pre_x, const bf16_t* __restrict__ pre_w, int m0, int rows,
                                            unsigned char* as, float* dtile) {
  constexpr int D = 256, PK = PRE_K, PRS = PRE_RS, DT_LD = PRE_DT_LD;
  constexpr int KH = P32_KH, CT = P32_CT, NS = PK / 128, NF = P32_TT / 16, CH8 = PK / 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const bf16_t* wrow[CT];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc) wrow[cc] = pre_w + ((int64_t)(wave * CT + cc) * 64 + lane) * 8;
  constexpr int64_t kstep_stride = (D / 16) * 512;
  // The rows of x are requested FIRST (loads return in order: behind the 48 KiB of weight fragments a wave asks for they would arrive
  // last, and the LDS image — hence every MFMA — waits for them), then all six steps' fragments up front (192 VGPRs: nothing else is
  // live yet; with three sets in flight the contraction, 96 MFMAs per wave, took three L2 round trips per two steps).
  constexpr int NCH = P32_TT * CH8 / P32_NT;         // 6 chunks of 16 bytes per thread
  uint4 xv[NCH];
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int idx = it * P32_NT + tid, row = idx / CH8, ch = idx - row * CH8;
    xv[it] = make_uint4(0, 0, 0, 0);
    if (m0 + row < rows) xv[it] = *(const uint4*)(pre_x + (int64_t)(m0 + row) * PK + ch * 8);
  }
  bf16x8 wv[NS][KH][CT];
#pragma unroll
  for (int g = 0; g < NS; ++g)
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) wv[g][ks][cc] = *(const bf16x8*)(wrow[cc] + (int64_t)(g * KH + ks) * kstep_stride);
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int idx = it * P32_NT + tid, row = idx / CH8, ch = idx - row * CH8;
    *(uint4*)(as + row * PRS + ch * 16) = xv[it];
  }
  __syncthreads();
  f32x4 acc[CT][NF];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the 48 B fragments (6 steps x 4 k-steps x 2 row tiles) four ahead of their two MFMAs each (FragStream, tapring.h: 240 of the 256
  // registers are the weights, the accumulators and this ring)
  const unsigned char* inl = as + l15 * PRS + q * 16;
  {
    FragStream<NS * KH * NF, 4, 1, CT> bs;
    auto frag = [&](int i) __attribute__((always_inline)) { return *(const bf16x8*)(inl + (i % NF) * 16 * PRS + (i / NF) * 64); };
    bs.prime(frag);
    bs.run(frag, [&](int i, bf16x8 Bf) __attribute__((always_inline)) {
      const int nf = i % NF, ks = (i / NF) % KH, g = i / (NF * KH);
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) acc[cc][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[g][ks][cc], Bf, acc[cc][nf], 0, 0, 0);
    });
  }
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) *(f32x4*)(dtile + (i * 16 + l15) * DT_LD + (wave * CT + cc) * 16 + q * 4) = acc[cc][i];
  __syncthreads();                       // dtile complete; every wave is done with the rows of x
}

// out[rows][256] bf16 = x[rows][768] · W' (+ R): the q|k|v input gradient on its own (the first block of a stack has no LayerNorm
// backward in front of it to host it): 32-row tiles, one pass over the 768-wide contraction, no split-K slabs
__global__ __launch_bounds__(LNB_WAVES * 64, 1) void pre768_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                   const bf16_t* __restrict__ R, bf16_t* __restrict__ out, int rows) {
  __shared__ __attribute__((aligned(16))) unsigned char as[P32_TT * PRE_RS];
  __shared__ __attribute__((aligned(16))) float dtile[P32_TT * PRE_DT_LD];
  const int m0 = blockIdx.x * P32_TT;
  pre768_gemm(x, w, m0, rows, as, dtile);
  const int rl = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 16, row = m0 + rl;
  if (row >= rows) return;
  float v[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 t = *(const f32x4*)(dtile + rl * PRE_DT_LD + c0 + 4 * i);
    v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
  }
  if (R) {
    float r[16];
    const bf16_t* rp = R + (int64_t)row * 256 + c0;
    unpack8(*(const uint4*)rp, r); unpack8(*(const uint4*)(rp + 8), r + 8);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += r[e];
  }
  bf16_t* op = out + (int64_t)row * 256 + c0;
  *(uint4*)op = pack8f(v); *(uint4*)(op + 8) = pack8f(v + 8);
}

// ---- ln_bwd256 + the k = 1 projection that consumes its dy (the sub-layer's input gradient), one kernel.
// The FFT block's backward ran  LN backward -> [dy to memory] -> window conv (w_2's dX with the ReLU gate; fc's dX with the attention
// delta)  as two dependent launches; a dependent launch costs ~4.5 us on this part before any work, x 20 per step.  Here a workgroup
// takes 32 rows (the 8 waves x 4 rows of one ln_bwd256 sweep), leaves their dy in LDS as the B operand, and multiplies by the packed
// transposed weight (ttsk_win_conv's pack: each wave owns 32 output channels of a 256-channel group and streams its fragments from
// L2, the next group's while this one computes), NG groups one after the other.  Same arithmetic in the same order as the two
// kernels: bit-identical outputs.
struct LnbProjArgs {
  LnBwdArgs ln;
  const bf16_t* w;       // [8 k-steps][Cout/16][64][8]
  bf16_t* out;           // [rows][Cout]
  const bf16_t* gate;    // [rows][Cout] or null: out = gate > 0 ? out : 0
  const float* o32;      // [rows][256] or null (Cout = 256): delta[(b*2 + h)*S + t] = sum over head h's 128 columns of out * o32
  float* delta;
  int Cout;
  // PRE: the upstream gradient is itself a k = 1 projection that nobody else reads — dout = pre_x · pre_w' (+ R): the input gradient of
  // the NEXT block's q|k|v projection (pre_x = dqkv [rows][768], pre_w = the pack of the transposed (768, 1, 256) weight).  Computed
  // here for the workgroup's 32 rows (fp32, straight into LDS) instead of by a launch of its own that leaves fp32 slabs in memory.
  const bf16_t* pre_x;
  const bf16_t* pre_w;
#ifdef TTSK_STAMPS
  unsigned long long* stamps;   // diagnostic build only (make stamps; ttsk_layernorm_bwd_proj_set_stamps): 8 x s_memrealtime per workgroup
#endif
};
#ifdef TTSK_STAMPS
#define LNB_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (SPLIT == 1 && p.stamps && threadIdx.x == 0) p.stamps[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); /* (the tool's buffer is sized for whole tiles) */ \
  } while (0)
#else
#define LNB_STAMP(i) do {} while (0)       // the product library carries no stamp code and no global state for it
#endif

// SPLIT (round 6, NG = 4 on the phoneme side: 32 tiles of 32 rows leave 7/8 of the chip idle and each workgroup streams all of w_2's 0.5 MB): the
// tile's four 256-channel output groups go to SPLIT workgroups (blockIdx = tile * SPLIT + part), NG / SPLIT groups each.  Each repeats the tile's
// upstream projection and LayerNorm backward (cheap beside its share of the weight stream); part 0 alone writes the row outputs (dz, dy) and the
// partial sums.  No seam, no reduction: every output element is computed by exactly one workgroup in the unsplit order — bit-identical.
// SPLIT = 2: in the train step these launches run beside the decoder's weight gradients, which hold 192 of the 256 CUs — 64 workgroups fit the
// rest in one round (four per tile ran two rounds there: 30-33 us against the unsplit 26-28).
constexpr int LNB_SPLIT_MAX_TILES = 63, LNB_SPLIT = 2;
template <int NG, bool PRE, int SPLIT = 1>
__global__ __launch_bounds__(LNB_WAVES * 64, 1) void ln_bwd256_proj_kernel(const LnbProjArgs p) {
  constexpr int D = 256;
  static_assert(LNB_WAVES * 64 == P32_NT && P32_D == D, "proj32.h is built for 8 waves and 256 channels");
  constexpr int PRS = PRE_RS, DT_LD = PRE_DT_LD;                          // PRE: LDS row stride of the dqkv rows, fp32 tile stride
  constexpr int MAIN_BYTES = LNB_WAVES * LNB_RED * 4 + 2 * P32_TT * P32_RS;  // red | xs | os
  constexpr int PRE_BYTES = P32_TT * PRS;                                  // dqkv rows: dead before red / xs / os are written
  __shared__ __attribute__((aligned(16))) unsigned char smem[(PRE && PRE_BYTES > MAIN_BYTES) ? PRE_BYTES : MAIN_BYTES];
  __shared__ __attribute__((aligned(16))) float dtile[PRE ? P32_TT * DT_LD : 4];
  float (*red)[LNB_RED] = (float (*)[LNB_RED])smem;
  unsigned char* xs = smem + LNB_WAVES * LNB_RED * 4;     // dy rows (bf16): the GEMM's B operand
  unsigned char* os = xs + P32_TT * P32_RS;               // one channel group's output rows (bf16)
  static_assert(NG % SPLIT == 0, "whole groups per workgroup");
  const int part = (int)blockIdx.x % SPLIT, tile = (int)blockIdx.x / SPLIT;
  LnBwdArgs a = p.ln;
  if (SPLIT > 1 && part != 0) a.dz = a.dy = nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, l = lane & 15, c0 = l * 16;
  const int m0 = tile * P32_TT;
  Proj32W W;
  LNB_STAMP(0);
  if (PRE) pre768_gemm(p.pre_x, p.pre_w, m0, a.rows, smem, dtile);
  LNB_STAMP(1);
  {
    const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
    const unsigned thr = keep_threshold(a.p_pre);
    const float dscale = a.p_pre > 0.f ? 1.f / (1.f - a.p_pre) : 1.f;
    float gam[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const f32x4 t = *(const f32x4*)(a.gamma + c0 + 4 * i); gam[4 * i] = t[0]; gam[4 * i + 1] = t[1]; gam[4 * i + 2] = t[2]; gam[4 * i + 3] = t[3]; }
    float sbias[16], sgam[16], sbeta[16], dzv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sbias[e] = sgam[e] = sbeta[e] = 0.f;
    const int rl = wave * 4 + grp;
    lnb256_row(a, m0 + rl, c0, gam, seed, step, thr, dscale, sbias, sgam, sbeta, dzv, PRE ? dtile + rl * DT_LD : nullptr);
    {
      // a lane's 32 bytes as two 16-byte stores: lanes l and l + 4 of a store's 8-lane group are 128 bytes apart — the same banks
      // (a two-way conflict on every store: 16 % of this kernel's LDS cycles at NG = 1).  Lanes 4-7 of a group store their upper half
      // first: the eight lanes then cover all 32 banks
      const uint4 lo = pack8f(dzv), hi = pack8f(dzv + 8);
      const int up = (l >> 2) & 1;
      unsigned char* dst = xs + rl * P32_RS + c0 * 2;
      *(uint4*)(dst + up * 16) = up ? hi : lo;
      *(uint4*)(dst + (up ^ 1) * 16) = up ? lo : hi;
    }
    LNB_STAMP(2);
    // group 0's weight fragments: requested behind the rows' own loads (loads return in order: in front of them they delayed every
    // row), with the partial sums and their barrier (~3 us) to arrive
    proj32_prefetch(p.w, p.Cout, wave, lane, W, part * (NG / SPLIT));
    lnb256_partials(a, sbias, sgam, sbeta, red, wave, grp, c0, tile, part == 0);        // (its barrier also publishes xs)
    LNB_STAMP(3);
  }
  uint4 gpre[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};       // the group's gate chunks (w_2's saved ReLU output, cold in HBM)
  f32x4 ypre[2][2];                                                         // fc's dX: the o32 chunks of the delta sums, likewise
#pragma unroll
  for (int it = 0; it < 2; ++it) { ypre[it][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ypre[it][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto pre = [&](int cg) __attribute__((always_inline)) {                   // requested when the group starts (proj32_run)
    if (NG == 1 && p.delta) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = it * P32_NT + tid, rr = idx >> 5, ch = idx & 31;
        if (m0 + rr < a.rows) {
          const float* op = p.o32 + (int64_t)(m0 + rr) * D + ch * 8;
          ypre[it][0] = *(const f32x4*)op; ypre[it][1] = *(const f32x4*)(op + 4);
        }
      }
    }
    if (p.gate) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = it * P32_NT + tid, rr = idx >> 5, ch = idx & 31;
        if (m0 + rr < a.rows) gpre[it] = *(const uint4*)(p.gate + (int64_t)(m0 + rr) * p.Cout + cg * D + ch * 8);
      }
    }
  };
  proj32_run<NG / SPLIT>(xs, os, p.w, p.Cout, nullptr, W, tid, [&](int cg, int rr, int ch, uint4 v, int it) __attribute__((always_inline)) {
    const int row = m0 + rr;
    float dacc = 0.f;
    if (row < a.rows) {
      if (p.gate) {
        const uint4 gt = gpre[it];
        auto keep = [](unsigned w) {      // 0xFFFF per bf16 half that is > 0 (sign clear, not zero)
          const unsigned lo = w & 0xFFFFu, hi = w >> 16;
          return ((lo - 1u) < 0x7FFFu ? 0xFFFFu : 0u) | ((hi - 1u) < 0x7FFFu ? 0xFFFF0000u : 0u);
        };
        v.x &= keep(gt.x); v.y &= keep(gt.y); v.z &= keep(gt.z); v.w &= keep(gt.w);
      }
      *(uint4*)(p.out + (int64_t)row * p.Cout + cg * D + ch * 8) = v;
      if (NG == 1 && p.delta) {
        const f32x4 y0 = ypre[it][0], y1 = ypre[it][1];
        dacc = __uint_as_float(v.x << 16) * y0[0] + __uint_as_float(v.x & 0xFFFF0000u) * y0[1] + __uint_as_float(v.y << 16) * y0[2] +
               __uint_as_float(v.y & 0xFFFF0000u) * y0[3] + __uint_as_float(v.z << 16) * y1[0] + __uint_as_float(v.z & 0xFFFF0000u) * y1[1] +
               __uint_as_float(v.w << 16) * y1[2] + __uint_as_float(v.w & 0xFFFF0000u) * y1[3];
      }
    }
    if (NG == 1 && p.delta) {          // 16 consecutive lanes hold one (row, head): 16 chunks of 8 columns
      dacc = quad16_sum(dacc);
      if (row < a.rows && (ch & 15) == 0) {
        const int b = row / a.seg_len, t = row - b * a.seg_len;
        p.delta[((int64_t)b * 2 + (ch >> 4)) * a.seg_len + t] = dacc;
      }
    }
  }, pre, part * (NG / SPLIT));
  LNB_STAMP(4);
}

// dst[c] (+)= scale * sum_b partials[b][c].  32 columns x 8 row-groups per workgroup; every thread adds its rows in
// ascending order and the 8 group sums are combined in a fixed order: deterministic.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partials, int nblk, int ncols,
                                                              int ld, float* __restrict__ dst, int accumulate, float scale) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, gy = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < ncols) {
    int b = gy;
    for (; b + 48 < nblk; b += 64) {       // 4 loads in flight per thread; summation order is fixed
      s0 += partials[(int64_t)b * ld + c];
      s1 += partials[(int64_t)(b + 16) * ld + c];
      s2 += partials[(int64_t)(b + 32) * ld + c];
      s3 += partials[(int64_t)(b + 48) * ld + c];
    }
    for (; b < nblk; b += 16) s0 += partials[(int64_t)b * ld + c];
  }
  red[gy][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gy == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    t *= scale;
    dst[c] = accumulate ? dst[c] + t : t;
  }
}

// the same reduction for a list of up to 64 (partials, destination) pairs in one launch: the gradient column sums of a
// whole backward pass are only needed by the optimizer, so their finalisation is deferred and batched
struct FinalizeBatch {
  ttsk_finalize_item it[64];
};
__global__ __launch_bounds__(256) void colsum_finalize_batch_kernel(const FinalizeBatch fb) {
  __shared__ float red[16][17];
  const ttsk_finalize_item& it = fb.it[blockIdx.y];
  const int cx = threadIdx.x & 15, gy = threadIdx.x >> 4;
  for (int c0 = blockIdx.x * 16; c0 < it.ncols; c0 += gridDim.x * 16) {
    const int c = c0 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < it.ncols) {
      int b = gy;
      for (; b + 48 < it.nblk; b += 64) {
        s0 += it.partials[(int64_t)b * it.ld + c];
        s1 += it.partials[(int64_t)(b + 16) * it.ld + c];
        s2 += it.partials[(int64_t)(b + 32) * it.ld + c];
        s3 += it.partials[(int64_t)(b + 48) * it.ld + c];
      }
      for (; b < it.nblk; b += 16) s0 += it.partials[(int64_t)b * it.ld + c];
    }
    red[gy][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy == 0 && c < it.ncols) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][cx];
      t *= it.scale;
      it.dst[c] = it.accumulate ? it.dst[c] + t : t;
    }
    __syncthreads();
  }
}

// column sums of a [rows][C] matrix (bf16 or fp32) -> partials[nblk][C]; a thread owns 16 bytes of a row (8 bf16 / 4 fp32 columns; 4
// bf16 columns when C or ld is not a multiple of 8) and keeps four rows' loads in flight (nontemporal: the matrix is read for the last
// time), added in row order.  red: [256 / threads-per-row][C] floats.
__device__ __forceinline__ int colsum_vec(int C, int ld, const void* x, bool f32) {
  return (!f32 && (C & 7) == 0 && (ld & 7) == 0 && (((uintptr_t)x) & 15) == 0) ? 8 : 4;
}
template <typename T>
__device__ __forceinline__ void colsum_body(const T* __restrict__ x, int rows, int C, int ld, float* __restrict__ partials, int blk, int nblk,
                                            float* red) {
  const int V = colsum_vec(C, ld, x, sizeof(T) == 4);
  const int tpr = (C + V - 1) / V;              // threads per row (C % 4 == 0 or C < 4 handled by the host)
  const int rpi = 256 / tpr;                    // rows per iteration
  const int r0 = threadIdx.x / tpr, c0 = (threadIdx.x % tpr) * V;
  const int per = (rows + nblk - 1) / nblk;
  const int rb = blk * per;
  const int re = min(rb + per, rows);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r0 < rpi) {
    if constexpr (sizeof(T) == 2) {
      if (V == 8) {
        auto add = [&](const uint4 u) __attribute__((always_inline)) {
          s[0] += __uint_as_float(u.x << 16); s[1] += __uint_as_float(u.x & 0xFFFF0000u);
          s[2] += __uint_as_float(u.y << 16); s[3] += __uint_as_float(u.y & 0xFFFF0000u);
          s[4] += __uint_as_float(u.z << 16); s[5] += __uint_as_float(u.z & 0xFFFF0000u);
          s[6] += __uint_as_float(u.w << 16); s[7] += __uint_as_float(u.w & 0xFFFF0000u);
        };
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
        const bf16_t* xp = (const bf16_t*)x + c0;
        int r = rb + r0;
        for (; r + 3 * rpi < re; r += 4 * rpi) {
          const u32x4_ a0 = __builtin_nontemporal_load((const u32x4_*)(xp + (int64_t)r * ld));
          const u32x4_ a1 = __builtin_nontemporal_load((const u32x4_*)(xp + (int64_t)(r + rpi) * ld));
          const u32x4_ a2 = __builtin_nontemporal_load((const u32x4_*)(xp + (int64_t)(r + 2 * rpi) * ld));
          const u32x4_ a3 = __builtin_nontemporal_load((const u32x4_*)(xp + (int64_t)(r + 3 * rpi) * ld));
          add(make_uint4(a0[0], a0[1], a0[2], a0[3])); add(make_uint4(a1[0], a1[1], a1[2], a1[3]));
          add(make_uint4(a2[0], a2[1], a2[2], a2[3])); add(make_uint4(a3[0], a3[1], a3[2], a3[3]));
        }
        for (; r < re; r += rpi) {
          const u32x4_ a0 = __builtin_nontemporal_load((const u32x4_*)(xp + (int64_t)r * ld));
          add(make_uint4(a0[0], a0[1], a0[2], a0[3]));
        }
      } else {
        for (int r = rb + r0; r < re; r += rpi) {
          const uint2 u = *(const uint2*)((const bf16_t*)x + (int64_t)r * ld + c0);
          s[0] += __uint_as_float(u.x << 16); s[1] += __uint_as_float(u.x & 0xFFFF0000u);
          s[2] += __uint_as_float(u.y << 16); s[3] += __uint_as_float(u.y & 0xFFFF0000u);
        }
      }
    } else {
      const float* xp = (const float*)x + c0;
      int r = rb + r0;
      for (; r + 3 * rpi < re; r += 4 * rpi) {
        const f32x4 a0 = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)r * ld));
        const f32x4 a1 = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)(r + rpi) * ld));
        const f32x4 a2 = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)(r + 2 * rpi) * ld));
        const f32x4 a3 = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)(r + 3 * rpi) * ld));
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += a0[e]; s[e] += a1[e]; s[e] += a2[e]; s[e] += a3[e]; }
      }
      for (; r < re; r += rpi) {
        const f32x4 u = __builtin_nontemporal_load((const f32x4*)(xp + (int64_t)r * ld));
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += u[e];
      }
    }
    for (int e = 0; e < V; ++e) red[r0 * C + c0 + e] = s[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int k = 0; k < rpi; ++k) t += red[k * C + c];
    partials[(int64_t)blk * C + c] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int rows, int C, int ld, float* __restrict__ partials) {
  extern __shared__ float red[];                // [rpi][C]
  colsum_body<T>(x, rows, C, ld, partials, blockIdx.x, gridDim.x, red);
}

// column sums of up to 64 matrices in one launch (bias gradients of a whole backward pass, deferred)
struct ColsumBatch {
  ttsk_colsum_item it[64];
};
__global__ __launch_bounds__(256) void colsum_batch_kernel(const ColsumBatch cb) {
  extern __shared__ float red[];
  const ttsk_colsum_item& it = cb.it[blockIdx.y];
  if ((int)blockIdx.x >= it.nblk) return;
  if (it.is_f32) colsum_body<float>((const float*)it.x, it.rows, it.C, it.ld, it.partials, blockIdx.x, it.nblk, red);
  else colsum_body<bf16_t>((const bf16_t*)it.x, it.rows, it.C, it.ld, it.partials, blockIdx.x, it.nblk, red);
}

}  // namespace

extern "C" int ttsk_layernorm_fwd(const void* y, const void* res, const float* gamma, const float* beta, void* out,
                                  void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int rows, int D,
                                  float eps, float p_pre, uint32_t site_pre, float p_post, uint32_t site_post,
                                  const uint64_t* rng, const float* head_w, const float* head_b, float* head_out,
                                  void* stream) {
  TTSK_REQUIRE(y && gamma && beta && mean && rstd, "layernorm_fwd: null pointer");
  TTSK_REQUIRE(rows > 0 && D >= 256 && D <= 1024 && (D & 255) == 0, "layernorm_fwd: D must be 256..1024 step 256 (got %d)", D);
  TTSK_REQUIRE(!lens || seg_len > 0, "layernorm_fwd: lens needs seg_len");
  TTSK_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "layernorm_fwd: dropout needs rng state");
  TTSK_REQUIRE(!head_w || (head_b && head_out), "layernorm_fwd: head needs bias and output");
  LnArgs a{(const bf16_t*)y, (const bf16_t*)res, gamma, beta, (bf16_t*)out, (bf16_t*)z_save, mean, rstd,
           (const long long*)lens, rng, head_w, head_b, head_out, rows, D, seg_len, p_pre, p_post, eps, site_pre, site_post, 0, 0, 0};
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_layernorm_fwd_grouped(const void* y, const void* res, const float* gamma, const float* beta, void* out,
                                          void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int groups,
                                          int group_rows, int64_t param_stride, uint32_t site_stride, int D, float eps, float p_pre,
                                          uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng,
                                          const float* head_w, const float* head_b, float* head_out, void* stream) {
  TTSK_REQUIRE(y && gamma && beta && mean && rstd, "layernorm_fwd_grouped: null pointer");
  TTSK_REQUIRE(groups > 0 && group_rows > 0 && D >= 256 && D <= 1024 && (D & 255) == 0, "layernorm_fwd_grouped: bad sizes");
  TTSK_REQUIRE(!lens || (seg_len > 0 && group_rows % seg_len == 0), "layernorm_fwd_grouped: lens needs group_rows %% seg_len == 0");
  TTSK_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "layernorm_fwd_grouped: dropout needs rng state");
  TTSK_REQUIRE(!head_w || (head_b && head_out), "layernorm_fwd_grouped: head needs bias and output");
  TTSK_REQUIRE((param_stride & 3) == 0, "layernorm_fwd_grouped: param_stride must keep 16-byte alignment");
  const int rows = groups * group_rows;
  LnArgs a{(const bf16_t*)y, (const bf16_t*)res, gamma, beta, (bf16_t*)out, (bf16_t*)z_save, mean, rstd,
           (const long long*)lens, rng, head_w, head_b, head_out, rows, D, seg_len, p_pre, p_post, eps, site_pre, site_post,
           group_rows, (long long)param_stride, site_stride};
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_layernorm_bwd_nblocks(int rows) {
  int n = (rows + LNB_WAVES - 1) / LNB_WAVES;
  return n > 256 ? 256 : n;
}

extern "C" int ttsk_layernorm_bwd(const void* dout, const float* dhead, const float* head_w, const void* z,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const int64_t* lens, int seg_len, int rows, int D, int relu_in, float p_pre,
                                  uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng, void* dz, void* dy,
                                  float* partials, void* stream) {
  TTSK_REQUIRE(z && mean && rstd && gamma && partials, "layernorm_bwd: null pointer");
  TTSK_REQUIRE(dout || dhead, "layernorm_bwd: need dout or dhead");
  TTSK_REQUIRE(!dhead || (head_w && beta), "layernorm_bwd: head mode needs head_w and beta");
  TTSK_REQUIRE(rows > 0 && D >= 256 && D <= 1024 && (D & 255) == 0, "layernorm_bwd: bad D %d", D);
  TTSK_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "layernorm_bwd: dropout needs rng state");
  TTSK_REQUIRE(p_post == 0.f || beta, "layernorm_bwd: post dropout needs beta");
  const int nblk = ttsk_layernorm_bwd_nblocks(rows);
  LnBwdArgs a{(const bf16_t*)dout, dhead, head_w, (const bf16_t*)z, mean, rstd, gamma, beta, (const long long*)lens, rng,
              (bf16_t*)dz, (bf16_t*)dy, partials, rows, D, seg_len, relu_in, p_pre, p_post, site_pre, site_post, 0, nblk, 0, 0, nullptr, nullptr, 0, 0};
  if (D == 256 && !dhead && !relu_in && p_post == 0.f && dout)
    hipLaunchKernelGGL(ln_bwd256_kernel, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (a.D <= 256)
    hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<MAXJ>, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_layernorm_bwd_grouped(const void* dout, const float* dhead, const float* head_w, const void* z,
                                          const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          const int64_t* lens, int seg_len, int groups, int group_rows, int64_t param_stride,
                                          uint32_t site_stride, int D, int relu_in, float p_pre, uint32_t site_pre, float p_post,
                                          uint32_t site_post, const uint64_t* rng, void* dz, void* dy, float* partials, void* stream) {
  TTSK_REQUIRE(z && mean && rstd && gamma && partials, "layernorm_bwd_grouped: null pointer");
  TTSK_REQUIRE(dout || dhead, "layernorm_bwd_grouped: need dout or dhead");
  TTSK_REQUIRE(!dhead || (head_w && beta), "layernorm_bwd_grouped: head mode needs head_w and beta");
  TTSK_REQUIRE(groups > 0 && group_rows > 0 && D >= 256 && D <= 1024 && (D & 255) == 0, "layernorm_bwd_grouped: bad sizes");
  TTSK_REQUIRE(!lens || (seg_len > 0 && group_rows % seg_len == 0), "layernorm_bwd_grouped: lens needs group_rows %% seg_len == 0");
  TTSK_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "layernorm_bwd_grouped: dropout needs rng state");
  TTSK_REQUIRE(p_post == 0.f || beta, "layernorm_bwd_grouped: post dropout needs beta");
  const int nblk = ttsk_layernorm_bwd_nblocks(group_rows);
  LnBwdArgs a{(const bf16_t*)dout, dhead, head_w, (const bf16_t*)z, mean, rstd, gamma, beta, (const long long*)lens, rng,
              (bf16_t*)dz, (bf16_t*)dy, partials, groups * group_rows, D, seg_len, relu_in, p_pre, p_post, site_pre, site_post,
              group_rows, nblk, (long long)param_stride, site_stride, nullptr, nullptr, 0, 0};
  if (a.D <= 256) hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(groups * nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(ln_bwd_kernel<MAXJ>, dim3(groups * nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_layernorm_bwd_slabs(const float* slabs, int nsplit, int64_t slab_stride, const void* R, const void* z,
                                        const float* mean, const float* rstd, const float* gamma, const float* beta,
                                        const int64_t* lens, int seg_len, int rows, int D, int relu_in, float p_pre,
                                        uint32_t site_pre, float p_post, uint32_t site_post, const uint64_t* rng, void* dz, void* dy,
                                        float* partials, void* stream) {
  TTSK_REQUIRE(slabs && nsplit > 0 && z && mean && rstd && gamma && partials, "layernorm_bwd_slabs: null pointer");
  TTSK_REQUIRE(rows > 0 && D >= 256 && D <= 1024 && (D & 255) == 0, "layernorm_bwd_slabs: bad D %d", D);
  TTSK_REQUIRE((slab_stride & 3) == 0 && slab_stride >= (int64_t)rows * D && (((uintptr_t)slabs) & 15) == 0, "layernorm_bwd_slabs: slabs must be 16-byte aligned [nsplit][rows][D]");
  TTSK_REQUIRE((p_pre == 0.f && p_post == 0.f) || rng, "layernorm_bwd_slabs: dropout needs rng state");
  TTSK_REQUIRE(p_post == 0.f || beta, "layernorm_bwd_slabs: post dropout needs beta");
  const int nblk = ttsk_layernorm_bwd_nblocks(rows);
  LnBwdArgs a{nullptr, nullptr, nullptr, (const bf16_t*)z, mean, rstd, gamma, beta, (const long long*)lens, rng,
              (bf16_t*)dz, (bf16_t*)dy, partials, rows, D, seg_len, relu_in, p_pre, p_post, site_pre, site_post, 0, nblk, 0, 0,
              slabs, (const bf16_t*)R, (long long)slab_stride, nsplit};
  if (D == 256 && !relu_in && p_post == 0.f)
    hipLaunchKernelGGL(ln_bwd256_kernel, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (a.D <= 256)
    hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<MAXJ>, dim3(nblk), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

#ifdef TTSK_STAMPS
static unsigned long long* g_lnb_stamps = nullptr;
// diagnostic build only (`make stamps`, tools/debug/lnb_stamps.py; not declared in ttsk.h, not in the product library): device buffer
// of 8 x uint64 per workgroup for the launches that follow; null switches the stamps off again
extern "C" int ttsk_layernorm_bwd_proj_set_stamps(void* dev_buffer) {
  g_lnb_stamps = (unsigned long long*)dev_buffer;
  return TTSK_OK;
}
#endif

extern "C" int ttsk_layernorm_bwd_proj_nblocks(int rows) { return (rows + 31) / 32; }

extern "C" int ttsk_layernorm_bwd_proj(const void* dout, const float* slabs, int nsplit, int64_t slab_stride, const void* R, const void* z,
                                       const float* mean, const float* rstd, const float* gamma, const int64_t* lens, int seg_len,
                                       int rows, int D, float p_pre, uint32_t site_pre, const uint64_t* rng, void* dz, void* dy,
                                       float* partials, const void* w_packed, int Cout, const void* gate, const float* delta_o32,
                                       float* delta_out, void* out, const void* pre_x, const void* pre_w_packed, int pre_K, void* stream) {
  TTSK_REQUIRE(z && mean && rstd && gamma && partials && w_packed && out, "layernorm_bwd_proj: null pointer");
  TTSK_REQUIRE((dout != nullptr) + (slabs != nullptr) + (pre_x != nullptr) == 1, "layernorm_bwd_proj: exactly one of dout / slabs / pre_x");
  TTSK_REQUIRE(!pre_x || (pre_w_packed && pre_K == 768 && (((uintptr_t)pre_x | (uintptr_t)pre_w_packed) & 15) == 0),
               "layernorm_bwd_proj: the upstream projection is built for a 768-wide contraction (q|k|v), 16-byte aligned operands");
  TTSK_REQUIRE(rows > 0 && D == 256 && (Cout == 256 || Cout == 1024), "layernorm_bwd_proj: D = 256, Cout = 256 or 1024 (got %d, %d)", D, Cout);
  TTSK_REQUIRE(!slabs || (nsplit > 0 && (slab_stride & 3) == 0 && slab_stride >= (int64_t)rows * D && (((uintptr_t)slabs) & 15) == 0),
               "layernorm_bwd_proj: slabs must be 16-byte aligned [nsplit][rows][D]");
  TTSK_REQUIRE(p_pre == 0.f || (rng && dy), "layernorm_bwd_proj: dropout needs the rng state and dy");
  TTSK_REQUIRE(!lens || seg_len > 0, "layernorm_bwd_proj: lens needs seg_len");
  TTSK_REQUIRE(!delta_out || (delta_o32 && Cout == 256 && !gate && seg_len > 0 && rows % seg_len == 0 && (((uintptr_t)delta_o32) & 15) == 0),
               "layernorm_bwd_proj: delta needs o32 (16-byte aligned), Cout = 256 = 2 heads x 128 and rows = B * seg_len");
  TTSK_REQUIRE((((uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)gate | (uintptr_t)z | (uintptr_t)dout | (uintptr_t)R | (uintptr_t)dz | (uintptr_t)dy) & 15) == 0,
               "layernorm_bwd_proj: operands must be 16-byte aligned");
  const int nblk = (rows + 31) / 32;
  LnbProjArgs p{{(const bf16_t*)dout, nullptr, nullptr, (const bf16_t*)z, mean, rstd, gamma, nullptr, (const long long*)lens, rng,
                 (bf16_t*)dz, (bf16_t*)dy, partials, rows, D, seg_len > 0 ? seg_len : 1, 0, p_pre, 0.f, site_pre, 0, 0, nblk, 0, 0,
                 slabs, (const bf16_t*)R, (long long)slab_stride, nsplit},
                (const bf16_t*)w_packed, (bf16_t*)out, (const bf16_t*)gate, delta_o32, delta_out, Cout, (const bf16_t*)pre_x,
                (const bf16_t*)pre_w_packed};
#ifdef TTSK_STAMPS
  p.stamps = g_lnb_stamps;
#endif
  const dim3 grid(nblk), block(LNB_WAVES * 64);
  // Few tiles (the phoneme side: 1,024 rows = 32 tiles on 256 CUs) and four output groups: one workgroup per (tile, group) — a quarter of
  // the weight stream each, four times the workgroups (SPLIT above).  From 64 tiles on the chip is better filled by whole tiles.
  const bool split = Cout == 1024 && nblk <= LNB_SPLIT_MAX_TILES;
  const dim3 grid4(nblk * LNB_SPLIT);
  if (pre_x) {
    if (Cout == 256) hipLaunchKernelGGL((ln_bwd256_proj_kernel<1, true>), grid, block, 0, (hipStream_t)stream, p);
    else if (split) hipLaunchKernelGGL((ln_bwd256_proj_kernel<4, true, LNB_SPLIT>), grid4, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((ln_bwd256_proj_kernel<4, true>), grid, block, 0, (hipStream_t)stream, p);
  } else {
    if (Cout == 256) hipLaunchKernelGGL((ln_bwd256_proj_kernel<1, false>), grid, block, 0, (hipStream_t)stream, p);
    else if (split) hipLaunchKernelGGL((ln_bwd256_proj_kernel<4, false, LNB_SPLIT>), grid4, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((ln_bwd256_proj_kernel<4, false>), grid, block, 0, (hipStream_t)stream, p);
  }
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_qkv_dx(const void* dqkv_bf16, const void* w_packed, const void* R_bf16, void* out_bf16, int rows, int K, int D,
                           void* stream) {
  TTSK_REQUIRE(dqkv_bf16 && w_packed && out_bf16 && rows > 0, "ttsk_qkv_dx: null pointer");
  TTSK_REQUIRE(K == 768 && D == 256, "ttsk_qkv_dx: built for a 768-wide contraction and 256 output channels (got %d, %d)", K, D);
  TTSK_REQUIRE((((uintptr_t)dqkv_bf16 | (uintptr_t)w_packed | (uintptr_t)R_bf16 | (uintptr_t)out_bf16) & 15) == 0, "ttsk_qkv_dx: 16-byte alignment");
  hipLaunchKernelGGL(pre768_kernel, dim3((rows + 31) / 32), dim3(LNB_WAVES * 64), 0, (hipStream_t)stream, (const bf16_t*)dqkv_bf16,
                     (const bf16_t*)w_packed, (const bf16_t*)R_bf16, (bf16_t*)out_bf16, rows);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_colsum_finalize(const float* partials, int nblk, int ncols, int ld, float* dst, int accumulate,
                                    float scale, void* stream) {
  TTSK_REQUIRE(partials && dst && nblk > 0 && ncols > 0 && ld >= ncols, "colsum_finalize: bad arguments");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nblk,
                     ncols, ld, dst, accumulate, scale);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_colsum_nblocks(int rows) {
  int n = (rows + 15) / 16;
  return n > 256 ? 256 : (n < 1 ? 1 : n);
}

extern "C" int ttsk_colsum(const void* x, int is_f32, int rows, int C, int ld, float* partials, void* stream) {
  TTSK_REQUIRE(x && partials && rows > 0 && C > 0 && ld >= C, "colsum: bad arguments");
  TTSK_REQUIRE((C & 3) == 0 && C <= 1024 && (ld & 3) == 0, "colsum: C must be a multiple of 4 and <= 1024 (got %d)", C);
  const int nblk = ttsk_colsum_nblocks(rows);
  const size_t shm = 256 * 8 * sizeof(float);          // [256 / threads-per-row][C]: 4 or 8 columns per thread
  if (is_f32)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(nblk), dim3(256), shm, (hipStream_t)stream, (const float*)x, rows, C, ld, partials);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(nblk), dim3(256), shm, (hipStream_t)stream, (const bf16_t*)x, rows, C, ld, partials);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_colsum_finalize_batch(const ttsk_finalize_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0, "colsum_finalize_batch: bad arguments");
  for (int base = 0; base < n; base += 64) {
    FinalizeBatch fb;
    const int m = n - base < 64 ? n - base : 64;
    for (int i = 0; i < m; ++i) {
      fb.it[i] = items[base + i];
      TTSK_REQUIRE(fb.it[i].partials && fb.it[i].dst && fb.it[i].nblk > 0 && fb.it[i].ncols > 0 && fb.it[i].ld >= fb.it[i].ncols,
                   "colsum_finalize_batch: bad item %d", base + i);
    }
    hipLaunchKernelGGL(colsum_finalize_batch_kernel, dim3(64, m), dim3(256), 0, (hipStream_t)stream, fb);      // 64 x 16 columns: one pass for up to 1,024 columns
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}

extern "C" int ttsk_colsum_batch(const ttsk_colsum_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0, "colsum_batch: bad arguments");
  for (int base = 0; base < n; base += 64) {
    ColsumBatch cb;
    const int m = n - base < 64 ? n - base : 64;
    int max_blk = 1;
    size_t shm = 0;
    for (int i = 0; i < m; ++i) {
      cb.it[i] = items[base + i];
      const ttsk_colsum_item& it = cb.it[i];
      TTSK_REQUIRE(it.x && it.partials && it.rows > 0 && it.C > 0 && (it.C & 3) == 0 && it.C <= 1024 && it.ld >= it.C && (it.ld & 3) == 0 &&
                       it.nblk == ttsk_colsum_nblocks(it.rows), "colsum_batch: bad item %d", base + i);
      if (it.nblk > max_blk) max_blk = it.nblk;
      const size_t need = 256 * 8 * sizeof(float);       // [256 / threads-per-row][C]: 4 or 8 columns per thread
      if (need > shm) shm = need;
    }
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(max_blk, m), dim3(256), shm, (hipStream_t)stream, cb);
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}
