// gemm_common.h — shared by the two tile configurations of the MFMA GEMM (gemm.hip: 128x128x64 / 4 waves, register
// staging; gemm2.hip: 256x128x64 / 8 waves, 3-stage LDS-DMA ring): launch arguments and the fused epilogue.
#pragma once
#include "common.h"

struct GemmArgs {
  ttsk_gemm_desc d;
  int tiles_m, tiles_n, kchunks, chunks_per_split;
};

template <bool F16> __device__ __forceinline__ void add_h8(float v[8], uint4 r) {
  float lo, hi;
  unpack2<F16>(r.x, lo, hi); v[0] += lo; v[1] += hi;
  unpack2<F16>(r.y, lo, hi); v[2] += lo; v[3] += hi;
  unpack2<F16>(r.z, lo, hi); v[4] += lo; v[5] += hi;
  unpack2<F16>(r.w, lo, hi); v[6] += lo; v[7] += hi;
}
template <bool F16> __device__ __forceinline__ uint4 pack8(const float v[8]) {
  return make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7]));
}

// Epilogue of one output row segment of 8 columns: v = alpha*acc + bias, residual, ReLU / gate / LeakyReLU / tanh, then the
// store(s).  Shared by the GEMM kernels (splits == 1) and by the split-K reducer.  In three steps so that a tile epilogue can
// issue the residual / gate loads of ALL its row segments before the first store (one thread handles 4-8 segments; with the
// loads inside the per-segment code every segment paid its own L2 round trip: the ReLU-gated w_2 dX GEMM, 4 K steps, took 26 us).

// output row of GEMM row gm (polyphase remap), or -1 when the row falls outside the output segment
__device__ __forceinline__ int64_t epilogue_row(const ttsk_gemm_desc& d, int gm, int z2) {
  if (d.out_mul == 0) return gm;
  const int s = gm / d.seg_len, t = gm - s * d.seg_len;
  const int o = t * d.out_mul + d.out_add + z2 * d.out_add_dz;
  if (o < 0 || o >= d.out_seg) return -1;
  return (int64_t)s * d.out_seg + o;
}

// the residual / gate operands of one segment, fetched with 16-byte loads when the layout allows (q0/q1: fp32 residual = 8 floats,
// or q0 = 8 x 16-bit residual and q1 = 8 x 16-bit gate); `fast` false = the apply step loads element by element
struct EpiOperands {
  uint4 q0, q1;
  bool fast;
};
__device__ __forceinline__ EpiOperands epilogue_load(const ttsk_gemm_desc& d, int64_t roff, int64_t orow, int gn, int nvalid) {
  EpiOperands o;
  o.q0 = o.q1 = make_uint4(0u, 0u, 0u, 0u);
  const int flags = d.flags;
  const bool has_r = flags & TTSK_GEMM_ADD_R, r32 = flags & TTSK_GEMM_R_F32, has_g = flags & TTSK_GEMM_MASK_G;
  o.fast = orow >= 0 && nvalid == 8 && ((((uintptr_t)d.R | (uintptr_t)d.G) & 15) == 0) && (!has_r || (r32 ? ((d.ldr & 3) == 0 && (roff & 3) == 0) : ((d.ldr & 7) == 0 && (roff & 7) == 0))) &&
           (!has_g || ((d.ldg & 7) == 0 && (roff & 7) == 0 && !(has_r && r32)));
  if (!o.fast) return o;
  if (has_r) {
    if (r32) {
      const float* rp = (const float*)d.R + roff + orow * d.ldr + gn;
      o.q0 = *(const uint4*)rp;
      o.q1 = *(const uint4*)(rp + 4);
    } else {
      o.q0 = *(const uint4*)((const bf16_t*)d.R + roff + orow * d.ldr + gn);
    }
  }
  if (has_g) o.q1 = *(const uint4*)((const bf16_t*)d.G + roff + orow * d.ldg + gn);
  return o;
}

template <bool F16>
__device__ __forceinline__ void epilogue_apply(const ttsk_gemm_desc& d, int64_t coff, int64_t roff, int64_t orow, int gn, int nvalid,
                                               float v[8], const float bias[8], const EpiOperands& op) {
  const int flags = d.flags;
  if (orow < 0) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + bias[e];
  if (flags & TTSK_GEMM_ADD_R) {
    if (op.fast) {
      if (flags & TTSK_GEMM_R_F32) {
        v[0] += __uint_as_float(op.q0.x); v[1] += __uint_as_float(op.q0.y); v[2] += __uint_as_float(op.q0.z); v[3] += __uint_as_float(op.q0.w);
        v[4] += __uint_as_float(op.q1.x); v[5] += __uint_as_float(op.q1.y); v[6] += __uint_as_float(op.q1.z); v[7] += __uint_as_float(op.q1.w);
      } else {
        add_h8<F16>(v, op.q0);
      }
    } else if (flags & TTSK_GEMM_R_F32) {
      const float* rp = (const float*)d.R + roff + orow * d.ldr + gn;
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += rp[e];
    } else {
      const bf16_t* rp = (const bf16_t*)d.R + roff + orow * d.ldr + gn;
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += unpack1<F16>(rp[e]);
    }
  }
  if (flags & TTSK_GEMM_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (flags & TTSK_GEMM_MASK_G) {
    if (op.fast) {
      float gv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      add_h8<F16>(gv, op.q1);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gv[e] > 0.f ? v[e] : 0.f;
    } else {
      const bf16_t* gp = (const bf16_t*)d.G + roff + orow * d.ldg + gn;
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] = unpack1<F16>(gp[e]) > 0.f ? v[e] : 0.f;
    }
  }
  if (flags & TTSK_GEMM_LRELU_OUT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.out_slope;
  }
  if (flags & TTSK_GEMM_TANH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
  }
  if (flags & TTSK_GEMM_C_F32) {
    float* cp = (float*)d.C + coff + orow * d.ldc + gn;
    if ((flags & TTSK_GEMM_ACCUM_C) && nvalid == 8 && ((d.ldc & 3) == 0) && ((coff & 3) == 0)) {
      const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);      // 16-byte read-modify-write of the gradient row
      *(f32x4*)cp = f32x4{v[0] + c0[0], v[1] + c0[1], v[2] + c0[2], v[3] + c0[3]};
      *(f32x4*)(cp + 4) = f32x4{v[4] + c1[0], v[5] + c1[1], v[6] + c1[2], v[7] + c1[3]};
    } else if (flags & TTSK_GEMM_ACCUM_C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] += v[e];
    } else if (nvalid == 8 && ((d.ldc & 3) == 0) && ((coff & 3) == 0)) {
      *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = v[e];
    }
  } else {
    bf16_t* cp = (bf16_t*)d.C + coff + orow * d.ldc + gn;
    if (nvalid == 8 && ((d.ldc & 7) == 0) && ((coff & 7) == 0)) {
      *(uint4*)cp = pack8<F16>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = pack1<F16>(v[e]);
    }
  }
  if (d.C2) {
    if (flags & TTSK_GEMM_C2_LRELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * d.out_slope;
    }
    bf16_t* cp = (bf16_t*)d.C2 + coff + orow * d.ldc + gn;
    if (nvalid == 8 && ((d.ldc & 7) == 0) && ((coff & 7) == 0)) {
      *(uint4*)cp = pack8<F16>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) cp[e] = pack1<F16>(v[e]);
    }
  }
}

// one segment start to finish (the split-K reducer, and anything that handles a single segment per thread)
template <bool F16>
__device__ __forceinline__ void epilogue_store(const ttsk_gemm_desc& d, int64_t coff, int64_t roff, int gm, int gn, int nvalid,
                                               float v[8], const float bias[8], int z2 = 0) {
  const int64_t orow = epilogue_row(d, gm, z2);
  if (orow < 0) return;
  const EpiOperands op = epilogue_load(d, roff, orow, gn, nvalid);
  epilogue_apply<F16>(d, coff, roff, orow, gn, nvalid, v, bias, op);
}

__device__ __forceinline__ int tr_sw(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// ---- LDS-DMA helper and tile epilogue of gemm2.hip
typedef __attribute__((address_space(3))) void* lds_ptr;

// one LDS-DMA piece: 64 lanes x 16 B from per-lane buffer offsets `voff` to the 1 KiB at LDS byte address `lds`
// (wave-uniform, in an SGPR).  Issued from inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write
// and puts `s_waitcnt vmcnt(0)` before the first ds_read of every K step (measured: DMA, LDS reads and MFMA then ran
// back to back, 1.0 us per K step for a lone workgroup whose DMA alone takes 0.6).  An asm load is invisible to that
// bookkeeping; landing is ordered by the counted s_waitcnt vmcnt(N) + s_barrier in the K loop.  M0 (the DMA's LDS
// destination) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rs, unsigned lds, int voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds), "s"(rs)
               : "memory");
}


// Epilogue of a BM x 128 tile held as 4x4 accumulator fragments per wave (waves as BM/64 x 2, each 64x64): 128 rows per
// pass through an fp32 LDS tile so that C (and the residual / gate operands) move as full 16-byte rows; split-K launches
// write raw partial sums into their workspace slab instead.  Every wave must have finished reading `smem` (barrier).
template <int BM, int NTHREADS, bool F16>
__device__ __forceinline__ void tile_epilogue(const ttsk_gemm_desc& d, unsigned char* smem, f32x4 (&acc)[4][4], int m0, int n0,
                                              int z, int z1, int z2, int split, int nzgrid, int wm, int wn, int lane, int tid) {
  constexpr int CS_LD = 132;
  const int M = d.M, N = d.N;
  const int l15 = lane & 15, lg = lane >> 4;
  float* cs = (float*)smem;
  const int64_t coff = z1 * d.sC1 + z2 * d.sC2;
  const int64_t roff = z1 * d.sR1 + z2 * d.sR2;
  const int cg = tid & 15;
  const int gn = n0 + cg * 8;
  const int nvalid = (N - gn) < 8 ? (N - gn) : 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = (d.bias && gn < N && e < nvalid && d.splits <= 1) ? d.bias[z1 * d.s_bias1 + gn + e] : 0.f;
  float* ws = (d.splits > 1 || (d.flags & TTSK_GEMM_RAW_SLABS)) ? (float*)d.workspace + ((int64_t)split * nzgrid + z) * ((int64_t)M * N) : nullptr;

#pragma unroll 1
  for (int pass = 0; pass < BM / 128; ++pass) {
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cs[((wm & 1) * 64 + i * 16 + lg * 4 + r) * CS_LD + wn * 64 + j * 16 + l15] = acc[i][j][r];
    }
    __syncthreads();
    if (gn < N) {
      constexpr int NP = 128 / (NTHREADS / 16);
      if (ws) {  // split-K: raw partial sums; the reducer applies the epilogue
        for (int p = 0; p < NP; ++p) {
          const int row = p * (NTHREADS / 16) + (tid >> 4);
          const int gm = m0 + pass * 128 + row;
          if (gm >= M) continue;
          const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
          const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
          float* wp = ws + (int64_t)gm * N + gn;
          if (nvalid == 8 && (N & 3) == 0) {
            *(f32x4*)wp = lo;
            *(f32x4*)(wp + 4) = hi;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (e < nvalid) wp[e] = lo[e]; if (e + 4 < nvalid) wp[e + 4] = hi[e]; }
          }
        }
      } else {
        // residual / gate operands of all this thread's segments first (independent loads in flight together), then the stores
        EpiOperands ops[NP];
        int64_t orow[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int gm = m0 + pass * 128 + p * (NTHREADS / 16) + (tid >> 4);
          orow[p] = gm < M ? epilogue_row(d, gm, z2) : -1;
          ops[p] = epilogue_load(d, roff, orow[p], gn, nvalid);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int row = p * (NTHREADS / 16) + (tid >> 4);
          if (orow[p] < 0) continue;
          const f32x4 lo = *(const f32x4*)(cs + row * CS_LD + cg * 8);
          const f32x4 hi = *(const f32x4*)(cs + row * CS_LD + cg * 8 + 4);
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[e + 4] = hi[e]; }
          epilogue_apply<F16>(d, coff, roff, orow[p], gn, nvalid, v, bias, ops[p]);
        }
      }
    }
    __syncthreads();
  }
}

// gemm2.hip
int ttsk_launch_gemm2(const GemmArgs& g, bool atr, bool btr, bool f16, hipStream_t s);
// grouped launch of the 256x128 configuration: prefix / args live in device memory (see gemm.hip: ttsk_gemm_group_*)
int ttsk_launch_gemm2_group(const int* prefix, const GemmArgs* args, int n, int total_wgs, int max_wgs, bool atr, bool btr, bool f16,
                            hipStream_t s);
