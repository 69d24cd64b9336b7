// rowops.hip — memory-bound row kernels of the FS2 path: masked softmax (+backward), embedding gathers and their
// deterministic scatter-sums, bucketize, dtype/layout conversions.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- softmax
// reference: fs_two/transformer/Modules.py:15-22 — scores were already scaled by 1/sqrt(d_k) in the Q·Kᵀ GEMM
// epilogue; keys at or past the utterance length get -inf.  One wavefront per (head-batch, query) row.
constexpr int SM_MAXE = 16;  // keys per lane: S <= 1024

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                          const long long* __restrict__ lens, int nrows, int S, int Sp,
                                                          int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int z = row / S;              // z = b*H + h
  const int len = lens ? (int)lens[z / H] : S;
  const float* sr = s + (int64_t)row * Sp;
  float v[SM_MAXE];
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    const int k = e * 64 + lane;
    v[e] = (k < S && k < len) ? sr[k] : -INFINITY;
    m = fmaxf(m, v[e]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) { v[e] = __expf(v[e] - m); sum += v[e]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  bf16_t* pr = p + (int64_t)row * Sp;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    const int k = e * 64 + lane;
    if (k < Sp) pr[k] = f2bf(k < S ? v[e] * inv : 0.f);
  }
}

// dS = alpha * P ∘ (dP − Σ_k dP∘P)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p, const float* __restrict__ dp,
                                                          bf16_t* __restrict__ ds, int nrows, int S, int Sp, float alpha) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const bf16_t* pr = p + (int64_t)row * Sp;
  const float* dr = dp + (int64_t)row * Sp;
  float pv[SM_MAXE], dv[SM_MAXE];
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    const int k = e * 64 + lane;
    pv[e] = k < S ? bf2f(pr[k]) : 0.f;
    dv[e] = k < S ? dr[k] : 0.f;
    dot += pv[e] * dv[e];
  }
  dot = wave_sum(dot);
  bf16_t* o = ds + (int64_t)row * Sp;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    const int k = e * 64 + lane;
    if (k < Sp) o[k] = f2bf(k < S ? alpha * pv[e] * (dv[e] - dot) : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------- bucketize
// reference: torch.bucketize(v, bins) (right=False) at model/modules.py:95-100,134-139: index = #{bins < v}
__global__ __launch_bounds__(256) void bucketize_kernel(const float* __restrict__ v, const float* __restrict__ bins, int nb,
                                                        float scale, int* __restrict__ idx, float* __restrict__ scaled, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = v[i] * scale;
  if (scaled) scaled[i] = x;
  int lo = 0, hi = nb;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] < x) lo = mid + 1; else hi = mid; }
  idx[i] = (x != x) ? nb : lo;   // NaN sorts last, as in torch
}

// ---------------------------------------------------------------------------------------------- gather + add
// out[row] = (in ? in[row] : 0) + table[idx[row / idx_div]] + (pe ? pe[row % pe_mod] : 0)
// reference: Models.py:101-103 (word embedding + position table), fastspeech2.py:72-75 + modules.py:159 (speaker
// embedding broadcast over phonemes), modules.py:95-100,134-139 (pitch / energy embeddings).
__global__ __launch_bounds__(256) void gather_add_kernel(const bf16_t* __restrict__ in, const float* __restrict__ table,
                                                         const void* __restrict__ idx, int idx_i64, int idx_div,
                                                         const float* __restrict__ pe, int pe_mod,
                                                         bf16_t* __restrict__ out, int rows, int D) {
  const int cpr = D >> 2;
  const int64_t n = (int64_t)rows * cpr;
  for (int64_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
    const int row = (int)(c / cpr), ch = (int)(c - (int64_t)row * cpr) * 4;
    const int64_t ir = row / idx_div;
    const int64_t t = idx_i64 ? ((const long long*)idx)[ir] : ((const int*)idx)[ir];
    f32x4 v = *(const f32x4*)(table + t * D + ch);
    if (pe) v += *(const f32x4*)(pe + (int64_t)(row % pe_mod) * D + ch);
    if (in) {
      const uint2 u = *(const uint2*)(in + (int64_t)row * D + ch);
      v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xFFFF0000u);
      v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xFFFF0000u);
    }
    *(uint2*)(out + (int64_t)row * D + ch) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}

// dtable[v] (+)= Σ_{rows with idx[row/idx_div] == v} dx[row]   — one workgroup per table row, rows visited in
// ascending order (deterministic; no atomics).  `skip` = padding_idx that receives no gradient (-1: none).
__device__ __forceinline__ void scatter_sum_body(const bf16_t* __restrict__ dx, const void* __restrict__ idx, int idx_i64, int idx_div,
                                                 int nidx, float* __restrict__ dtable, int D, int skip, int accumulate, int v, int* hits,
                                                 int& nh) {
  float wacc[4][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // D <= 1024: lane owns columns j*256 + 4*lane ..
  if (v != skip) {
    for (int base = 0; base < nidx; base += 1024) {
      // every thread tests 4 indices of the chunk (one round trip for the whole chunk; a single wave walking the chunk 64
      // indices per dependent load took most of the kernel's 59 us), then one wave compacts the flags in ascending order
      unsigned char* flag = (unsigned char*)(hits + 1024);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + threadIdx.x + 256 * u;
        flag[threadIdx.x + 256 * u] = (i < nidx && (idx_i64 ? ((const long long*)idx)[i] : ((const int*)idx)[i]) == v) ? 1 : 0;
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        int count = 0;
        for (int j0 = 0; j0 < 1024 && base + j0 < nidx; j0 += 64) {
          const bool hit = flag[j0 + threadIdx.x] != 0;
          const unsigned long long bal = __ballot(hit);
          if (hit) hits[count + __popcll(bal & ((1ull << threadIdx.x) - 1ull))] = base + j0 + threadIdx.x;
          count += __popcll(bal);
        }
        if (threadIdx.x == 0) nh = count;
      }
      __syncthreads();
      // the rows a table entry collects: hit h contributes rows hits[h]*idx_div .. +idx_div-1 (q = h*idx_div + r, ascending).
      // Wave w takes rows q = w, w+4, ...; a lane owns 4 columns (8-byte loads), eight row loads in flight per wave.  One
      // thread per column walking the rows one dependent 2-byte load at a time took 34 us for the speaker table (64 rows
      // per hit).  Fixed association: per-wave sums in ascending q, then (w0 + w1) + (w2 + w3) — deterministic.
      const int nrow = nh * idx_div;
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < D)
        for (int q0 = wave; q0 < nrow; q0 += 32) {
          uint2 u[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int q = q0 + 4 * t;
            u[t] = make_uint2(0u, 0u);
            if (q < nrow) {
              const int64_t row = (int64_t)hits[q / idx_div] * idx_div + q % idx_div;
              u[t] = *(const uint2*)(dx + row * D + c);
            }
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            wacc[j][0] += __uint_as_float(u[t].x << 16); wacc[j][1] += __uint_as_float(u[t].x & 0xFFFF0000u);
            wacc[j][2] += __uint_as_float(u[t].y << 16); wacc[j][3] += __uint_as_float(u[t].y & 0xFFFF0000u);
          }
        }
      }
      __syncthreads();
    }
  }
  // cross-wave combine through LDS (the hit list is dead now): red[wave][column]
  float* red = (float*)hits;
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j * 256 >= D) continue;          // uniform over the workgroup
      __syncthreads();
      const int c = lane * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave * 256 + c + e] = wacc[j][e];
      __syncthreads();
      const int col = j * 256 + threadIdx.x;
      if (col < D) {
        const float t = (red[threadIdx.x] + red[256 + threadIdx.x]) + (red[512 + threadIdx.x] + red[768 + threadIdx.x]);
        dtable[(int64_t)v * D + col] = accumulate ? dtable[(int64_t)v * D + col] + t : t;
      }
    }
  }
}

__global__ __launch_bounds__(256) void scatter_sum_kernel(const bf16_t* __restrict__ dx, const void* __restrict__ idx,
                                                          int idx_i64, int idx_div, int nidx, float* __restrict__ dtable,
                                                          int D, int skip, int accumulate) {
  __shared__ int hits[1024 + 256];      // hit list + 1024 flag bytes
  __shared__ int nh;
  scatter_sum_body(dx, idx, idx_i64, idx_div, nidx, dtable, D, skip, accumulate, blockIdx.x, hits, nh);
}

// up to 8 independent scatter-sums (the embedding-table gradients of one backward pass) in one launch: grid.y = item
struct ScatterBatch {
  ttsk_scatter_item it[8];
};
__global__ __launch_bounds__(256) void scatter_sum_batch_kernel(const ScatterBatch sb) {
  __shared__ int hits[1024 + 256];      // hit list + 1024 flag bytes
  __shared__ int nh;
  const ttsk_scatter_item& it = sb.it[blockIdx.y];
  if ((int)blockIdx.x >= it.n_table_rows) return;
  scatter_sum_body((const bf16_t*)it.dx, it.idx, it.idx_is_i64, it.idx_div, it.n_idx, it.dtable, it.D, it.skip_row, it.accumulate,
                   blockIdx.x, hits, nh);
}

// ---------------------------------------------------------------------------------------------- variance adaptor, training
// reference: model/modules.py:158-193 with targets given: x1 = x + speaker_emb; x2 = x1 + pitch_emb[bucketize(pitch_target)];
// x3 = x2 + energy_emb[bucketize(energy_target)] — the three gather_add launches and the two bucketize launches of the
// step-by-step path as ONE pass over the rows (each of x1, x2, x3 rounded to bf16 as there).  x1 / x2 are the inputs of the
// pitch / energy predictors: together with x they form the [3][rows][D] operand of the grouped predictor launches.
__device__ __forceinline__ int bucket_of(const float* __restrict__ bins, int nb, float x) {
  int lo = 0, hi = nb;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] < x) lo = mid + 1; else hi = mid; }
  return (x != x) ? nb : lo;
}
__device__ __forceinline__ f32x4 round_bf4(f32x4 v) {
  return f32x4{bf2f(f2bf(v[0])), bf2f(f2bf(v[1])), bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
}
__global__ __launch_bounds__(256) void va_embed_kernel(const bf16_t* __restrict__ x, const float* __restrict__ spk_table,
                                                       const long long* __restrict__ speakers, int L, const float* __restrict__ pitch_t,
                                                       const float* __restrict__ pitch_bins, const float* __restrict__ pitch_table,
                                                       const float* __restrict__ energy_t, const float* __restrict__ energy_bins,
                                                       const float* __restrict__ energy_table, int nb, bf16_t* __restrict__ x1,
                                                       bf16_t* __restrict__ x2, bf16_t* __restrict__ x3, int* __restrict__ pidx,
                                                       int* __restrict__ eidx, int rows, int D, const long long* __restrict__ row_limit) {
  const int cpr = D >> 2;
  const int64_t n = (int64_t)rows * cpr;
  for (int64_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
    const int row = (int)(c / cpr), ch = (int)(c - (int64_t)row * cpr) * 4;
    const int pi = bucket_of(pitch_bins, nb, pitch_t[row]), ei = bucket_of(energy_bins, nb, energy_t[row]);
    if (ch == 0) { pidx[row] = pi; eidx[row] = ei; }
    if (row_limit && (row % L) >= row_limit[row / L]) {   // a phoneme position past the batch's own longest text (bucketed L): zero rows
      const uint2 zz = make_uint2(0u, 0u);
      *(uint2*)(x1 + (int64_t)row * D + ch) = zz; *(uint2*)(x2 + (int64_t)row * D + ch) = zz; *(uint2*)(x3 + (int64_t)row * D + ch) = zz;
      continue;
    }
    const uint2 u = *(const uint2*)(x + (int64_t)row * D + ch);
    f32x4 v = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
    v = round_bf4(v + *(const f32x4*)(spk_table + speakers[row / L] * D + ch));
    *(uint2*)(x1 + (int64_t)row * D + ch) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    v = round_bf4(v + *(const f32x4*)(pitch_table + (int64_t)pi * D + ch));
    *(uint2*)(x2 + (int64_t)row * D + ch) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    v = v + *(const f32x4*)(energy_table + (int64_t)ei * D + ch);
    *(uint2*)(x3 + (int64_t)row * D + ch) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}

// backward of that chain around the grouped predictor backward: dxin [3][rows][D] fp32 = the gradients the duration / pitch /
// energy predictors send to their inputs x / x1 / x2; dx3 = gradient of x3 (from the LengthRegulator).
//   dx2 = dx3 + dxin[2]  (gradient of x2: also what pitch_embedding collects)
//   dx1 = dx2 + dxin[1]  (gradient of x1: also what speaker_emb collects)
//   dx  = dx1 + dxin[0]  (gradient of the encoder output)
// each rounded to bf16 — the values the step-by-step path's conv dX epilogues (fp32 accumulator + bf16 residual) produce.
__global__ __launch_bounds__(256) void va_combine_kernel(const bf16_t* __restrict__ dx3, const float* __restrict__ dxin, bf16_t* __restrict__ dx2,
                                                         bf16_t* __restrict__ dx1, bf16_t* __restrict__ dx, int64_t n4, int64_t gstride,
                                                         int D4, int L, const long long* __restrict__ row_limit) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    if (row_limit) {
      const int row = (int)(i / D4);
      if ((row % L) >= row_limit[row / L]) {              // no such phoneme position in the reference's batch: no gradient
        const uint2 zz = make_uint2(0u, 0u);
        *(uint2*)(dx2 + i * 4) = zz; *(uint2*)(dx1 + i * 4) = zz; *(uint2*)(dx + i * 4) = zz;
        continue;
      }
    }
    const uint2 u = *(const uint2*)(dx3 + i * 4);
    f32x4 v = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
    v = round_bf4(v + *(const f32x4*)(dxin + 2 * gstride + i * 4));
    *(uint2*)(dx2 + i * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    v = round_bf4(v + *(const f32x4*)(dxin + gstride + i * 4));
    *(uint2*)(dx1 + i * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    v = v + *(const f32x4*)(dxin + i * 4);
    *(uint2*)(dx + i * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}

// ---------------------------------------------------------------------------------------------- conversions
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(src + i * 4);
    *(uint2*)(dst + i * 4) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x]);
}

// (B,C,T) fp32 -> (B,T,C) bf16 through a 32x33 LDS tile (HiFi-GAN takes mel as (B,80,T); kernels are channels-last)
template <bool F16>
__global__ __launch_bounds__(256) void nct_to_ntc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    tile[r][tx] = (c < C && t < T) ? src[((int64_t)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < T && c < C) dst[((int64_t)b * T + t) * C + c] = pack1<F16>(tile[tx][r]);
  }
}

// audio fp32 -> int16: (x * max_wav).astype(int16), C truncation toward zero (reference: hifiapi.py:50-51)
__global__ __launch_bounds__(256) void to_int16_kernel(const float* __restrict__ src, short* __restrict__ dst, int64_t n, float scale) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int v = (int)(src[i] * scale);            // trunc toward zero
    dst[i] = (short)v;                              // wraps like numpy's int32->int16 cast path on overflow
  }
}

// rows (u, t) with t >= frame_limit[0] := 0 (16-byte pieces; bytes per row a multiple of 16)
__global__ __launch_bounds__(256) void zero_frames_kernel(uint4* __restrict__ x, int rows, int q_per_row, int seg_len, const int* __restrict__ frame_limit) {
  const int lim = frame_limit[0];
  const int64_t n = (int64_t)rows * q_per_row;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / q_per_row);
    if (r % seg_len >= lim) x[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// out = a + sb * b (fp32)
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float sb,
                                                      float* __restrict__ o, int64_t n) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = a[i] + sb * b[i];
}

// d = clamp(round(exp(logd) - 1) * d_control, min 0)   (reference: model/modules.py:199-203; round = half-to-even)
__global__ __launch_bounds__(256) void duration_round_kernel(const float* __restrict__ logd, float d_control, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = fmaxf(rintf(expf(logd[i]) - 1.f) * d_control, 0.f);
}

// mask[b][t] = t >= lens[b]   (True = PAD; reference: fs_two/utils/tools.py:121-131)
__global__ __launch_bounds__(256) void length_mask_kernel(const long long* __restrict__ lens, unsigned char* __restrict__ mask, int B, int T) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B * T) { const int b = i / T, t = i - b * T; mask[i] = (float)t >= (float)lens[b] ? 1 : 0; }
}

}  // namespace

static inline int grid_for(int64_t n, int cap = 2048) {
  int64_t b = (n + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

extern "C" int ttsk_softmax_fwd(const float* scores, void* probs_bf16, const int64_t* lens, int nz, int H, int S, int Sp,
                                void* stream) {
  TTSK_REQUIRE(scores && probs_bf16 && nz > 0 && H > 0, "softmax_fwd: bad arguments");
  TTSK_REQUIRE(S > 0 && S <= 1024 && Sp >= S && Sp <= 1024, "softmax_fwd: S=%d Sp=%d out of range (<=1024)", S, Sp);
  const int nrows = nz * S;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores,
                     (bf16_t*)probs_bf16, (const long long*)lens, nrows, S, Sp, H);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_softmax_bwd(const void* probs_bf16, const float* dprobs, void* dscores_bf16, int nz, int S, int Sp,
                                float alpha, void* stream) {
  TTSK_REQUIRE(probs_bf16 && dprobs && dscores_bf16 && nz > 0, "softmax_bwd: bad arguments");
  TTSK_REQUIRE(S > 0 && S <= 1024 && Sp >= S && Sp <= 1024, "softmax_bwd: S out of range");
  const int nrows = nz * S;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)probs_bf16,
                     dprobs, (bf16_t*)dscores_bf16, nrows, S, Sp, alpha);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_bucketize(const float* values, const float* bins, int n_bins, float scale, int32_t* idx,
                              float* scaled_out, int n, void* stream) {
  TTSK_REQUIRE(values && bins && idx && n > 0 && n_bins > 0, "bucketize: bad arguments");
  hipLaunchKernelGGL(bucketize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, values, bins, n_bins, scale,
                     idx, scaled_out, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_gather_add(const void* in_bf16, const float* table, const void* idx, int idx_is_i64, int idx_div,
                               const float* pe, int pe_mod, void* out_bf16, int rows, int D, void* stream) {
  TTSK_REQUIRE(table && idx && out_bf16 && rows > 0 && D > 0 && (D & 3) == 0, "gather_add: bad arguments");
  TTSK_REQUIRE(idx_div > 0 && (!pe || pe_mod > 0), "gather_add: idx_div/pe_mod");
  hipLaunchKernelGGL(gather_add_kernel, dim3(grid_for((int64_t)rows * (D >> 2))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in_bf16, table, idx, idx_is_i64, idx_div, pe, pe_mod, (bf16_t*)out_bf16, rows, D);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_scatter_sum(const void* dx_bf16, const void* idx, int idx_is_i64, int idx_div, int n_idx, float* dtable,
                                int n_table_rows, int D, int skip_row, int accumulate, void* stream) {
  TTSK_REQUIRE(dx_bf16 && idx && dtable && n_idx > 0 && n_table_rows > 0, "scatter_sum: bad arguments");
  TTSK_REQUIRE(D > 0 && D <= 1024 && (D & 3) == 0 && idx_div > 0, "scatter_sum: D must be a multiple of 4, <= 1024");
  hipLaunchKernelGGL(scatter_sum_kernel, dim3(n_table_rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx_bf16, idx,
                     idx_is_i64, idx_div, n_idx, dtable, D, skip_row, accumulate);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_scatter_sum_batch(const ttsk_scatter_item* items, int n, void* stream) {
  TTSK_REQUIRE(items && n > 0, "scatter_sum_batch: bad arguments");
  for (int base = 0; base < n; base += 8) {
    ScatterBatch sb;
    const int m = n - base < 8 ? n - base : 8;
    int max_rows = 1;
    for (int i = 0; i < m; ++i) {
      sb.it[i] = items[base + i];
      const ttsk_scatter_item& it = sb.it[i];
      TTSK_REQUIRE(it.dx && it.idx && it.dtable && it.n_idx > 0 && it.n_table_rows > 0 && it.D > 0 && it.D <= 1024 && (it.D & 3) == 0 && it.idx_div > 0,
                   "scatter_sum_batch: bad item %d", base + i);
      if (it.n_table_rows > max_rows) max_rows = it.n_table_rows;
    }
    hipLaunchKernelGGL(scatter_sum_batch_kernel, dim3(max_rows, m), dim3(256), 0, (hipStream_t)stream, sb);
    TTSK_CHECK_LAUNCH();
  }
  return TTSK_OK;
}

extern "C" int ttsk_cast_bf16(const float* src, void* dst_bf16, int64_t n, void* stream) {
  TTSK_REQUIRE(src && dst_bf16 && n > 0, "cast_bf16: bad arguments");
  TTSK_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst_bf16) & 7) == 0, "cast_bf16: alignment");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n >> 2, 4096)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst_bf16, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_nct_to_ntc(const float* src, void* dst16, int f16, int B, int C, int T, void* stream) {
  TTSK_REQUIRE(src && dst16 && B > 0 && C > 0 && T > 0 && B <= 65535, "nct_to_ntc: bad arguments");
  if (f16)
    hipLaunchKernelGGL(nct_to_ntc_kernel<true>, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst16, C, T);
  else
    hipLaunchKernelGGL(nct_to_ntc_kernel<false>, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst16, C, T);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_to_int16(const float* src, int16_t* dst, int64_t n, float scale, void* stream) {
  TTSK_REQUIRE(src && dst && n > 0, "to_int16: bad arguments");
  hipLaunchKernelGGL(to_int16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (short*)dst, n, scale);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_add_f32(const float* a, const float* b, float scale_b, float* out, int64_t n, void* stream) {
  TTSK_REQUIRE(a && b && out && n > 0, "add_f32: bad arguments");
  hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, scale_b, out, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_duration_round(const float* logd, float d_control, float* out, int n, void* stream) {
  TTSK_REQUIRE(logd && out && n > 0, "duration_round: bad arguments");
  hipLaunchKernelGGL(duration_round_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, logd, d_control, out, n);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_length_mask(const int64_t* lens, uint8_t* mask, int B, int T, void* stream) {
  TTSK_REQUIRE(lens && mask && B > 0 && T > 0, "length_mask: bad arguments");
  hipLaunchKernelGGL(length_mask_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)lens, mask, B, T);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_va_embed(const void* x_bf16, const float* speaker_table, const int64_t* speakers, int L, const float* pitch_target,
                             const float* pitch_bins, const float* pitch_table, const float* energy_target, const float* energy_bins,
                             const float* energy_table, int n_bins_minus_1, void* x1_bf16, void* x2_bf16, void* x3_bf16, int32_t* pitch_idx,
                             int32_t* energy_idx, int rows, int D, const int64_t* row_limit, void* stream) {
  TTSK_REQUIRE(x_bf16 && speaker_table && speakers && pitch_target && pitch_bins && pitch_table && energy_target && energy_bins &&
                   energy_table && x1_bf16 && x2_bf16 && x3_bf16 && pitch_idx && energy_idx, "va_embed: null pointer");
  TTSK_REQUIRE(rows > 0 && L > 0 && rows % L == 0 && D > 0 && (D & 3) == 0 && n_bins_minus_1 > 0, "va_embed: bad sizes");
  hipLaunchKernelGGL(va_embed_kernel, dim3(grid_for((int64_t)rows * (D >> 2))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                     speaker_table, (const long long*)speakers, L, pitch_target, pitch_bins, pitch_table, energy_target, energy_bins,
                     energy_table, n_bins_minus_1, (bf16_t*)x1_bf16, (bf16_t*)x2_bf16, (bf16_t*)x3_bf16, pitch_idx, energy_idx, rows, D,
                     (const long long*)row_limit);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_va_combine(const void* dx3_bf16, const float* dxin, void* dx2_bf16, void* dx1_bf16, void* dx_bf16, int rows, int D,
                               int L, const int64_t* row_limit, void* stream) {
  TTSK_REQUIRE(dx3_bf16 && dxin && dx2_bf16 && dx1_bf16 && dx_bf16 && rows > 0 && D > 0 && (D & 3) == 0, "va_combine: bad arguments");
  TTSK_REQUIRE(!row_limit || (L > 0 && rows % L == 0), "va_combine: row_limit needs rows %% L == 0");
  const int64_t n4 = (int64_t)rows * D / 4;
  hipLaunchKernelGGL(va_combine_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx3_bf16, dxin,
                     (bf16_t*)dx2_bf16, (bf16_t*)dx1_bf16, (bf16_t*)dx_bf16, n4, (int64_t)rows * D, D / 4, L > 0 ? L : 1, (const long long*)row_limit);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_zero_frames_from(void* x, int elem_bytes, int rows, int C, int seg_len, const int32_t* frame_limit, void* stream) {
  TTSK_REQUIRE(x && frame_limit && rows > 0 && C > 0 && seg_len > 0 && rows % seg_len == 0, "zero_frames_from: bad arguments");
  TTSK_REQUIRE((elem_bytes == 2 || elem_bytes == 4) && ((int64_t)C * elem_bytes) % 16 == 0 && (((uintptr_t)x) & 15) == 0,
               "zero_frames_from: rows must be whole 16-byte pieces");
  const int q = C * elem_bytes / 16;
  hipLaunchKernelGGL(zero_frames_kernel, dim3(grid_for((int64_t)rows * q)), dim3(256), 0, (hipStream_t)stream, (uint4*)x, rows, q, seg_len,
                     frame_limit);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

// ---- the dropout keep-mask of one site, as every kernel of the step draws it: element e of the site's [rows][C] tensor (row-major,
// e = row * C + c) is kept iff word (e & 3) of Philox4x32-10(key = seed, counter = (e >> 2, site, step)) >= keep_threshold(p).
// Parity tests hand these masks to the oracle (tests/test_parity_gpu.py: one full-size step with dropout ON on both sides).
namespace {
__global__ __launch_bounds__(256) void dropout_keep_mask_kernel(const uint64_t* __restrict__ rng, unsigned site, int64_t n4, float p,
                                                                uint8_t* __restrict__ keep) {
  const uint64_t seed = rng[0], step = rng[1];
  const unsigned thr = keep_threshold(p);
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const uint4 b = Philox::gen(make_uint2((unsigned)seed, (unsigned)(seed >> 32)), make_uint4((unsigned)i, site, (unsigned)step, (unsigned)(step >> 32)));
    *(uchar4*)(keep + i * 4) = make_uchar4(b.x >= thr, b.y >= thr, b.z >= thr, b.w >= thr);
  }
}
}  // namespace

extern "C" int ttsk_dropout_keep_mask(const uint64_t* rng, uint32_t site, int64_t n, float p, uint8_t* keep, void* stream) {
  TTSK_REQUIRE(rng && keep && n > 0 && (n & 3) == 0 && p >= 0.f && p < 1.f, "dropout_keep_mask: n must be a positive multiple of 4, 0 <= p < 1");
  TTSK_REQUIRE(n / 4 <= 0xFFFFFFFFll && (((uintptr_t)keep) & 3) == 0, "dropout_keep_mask: element index exceeds the 32-bit counter word / alignment");
  hipLaunchKernelGGL(dropout_keep_mask_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, rng, site, n / 4, p, keep);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
