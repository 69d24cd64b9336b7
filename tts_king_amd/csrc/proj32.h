// proj32.h — the GEMM tail shared by the kernels that hand a 32-row tile straight to the next k = 1 projection instead of writing it
// out for another launch to read (layernorm.hip: LayerNorm backward -> dX projection; gemm_ln.hip: FFTBlock tail -> the next block's
// q|k|v projection).  A dependent launch costs ~4.5 us on MI355X before any work; these tiles are already in the workgroup's LDS.
//
// Operands: 32 rows x 256 channels bf16 in LDS (`xs`, row stride P32_RS: the B operand, D[cout][row] orientation) and the packed
// weight of ttsk_win_conv for K = 1, [8 k-steps][Cout/16][64 lanes][8], streamed L2 -> registers: each of the 8 waves owns 32 output
// channels of a 256-channel group, NG groups one after the other, the next group's fragments requested when this one starts.
// Same k-step order as win_conv_kernel<256, ...>: bit-identical results.
#pragma once
#include "common.h"
#include "tapring.h"

constexpr int P32_TT = 32, P32_D = 256, P32_RS = P32_D * 2 + 32, P32_KH = 4, P32_CT = 2, P32_NT = 512;

struct Proj32W {
  bf16x8 a[P32_KH][P32_CT], b[P32_KH][P32_CT];      // the two 128-channel steps of a group
};

// (buffer loads: one per-lane byte offset, the group's and step's distance in a scalar register — tapring.h)
__device__ __forceinline__ void proj32_load(const bf16_t* __restrict__ w, int Cout, int cg, int g, int wave, int lane, bf16x8 (&dst)[P32_KH][P32_CT]) {
  const int kstep_bytes = (Cout / 16) * 1024;
  frags_load<P32_KH, P32_CT>(dst, weights_rsrc(w, (P32_D / 32) * kstep_bytes), (wave * P32_CT) * 1024 + lane * 16, cg * 16 * 1024 + g * P32_KH * kstep_bytes,
                             kstep_bytes);
}
__device__ __forceinline__ void proj32_prefetch(const bf16_t* __restrict__ w, int Cout, int wave, int lane, Proj32W& W, int cg0 = 0) {
  proj32_load(w, Cout, cg0, 0, wave, lane, W.a);
  proj32_load(w, Cout, cg0, 1, wave, lane, W.b);
}

// xs must be visible to the workgroup (a barrier behind its writers) and W prefetched.  bias: fp32 [Cout] or null.  Every thread calls
// epi(cg, rr, ch, v, it) for its 16-byte chunks (row rr of the tile, chunk ch of the group's 32; v = 8 bf16 outputs; it = 0, 1: the
// thread's first / second chunk of the group: idx = it * 512 + tid, rr = idx / 32, ch = idx % 32) — twice per group —
// including rows that do not exist: the caller checks and stores.  `os`: 32 x P32_RS bytes of LDS for the staging tile.
struct Proj32NoPre { __device__ __forceinline__ void operator()(int) const {} };

// `pre(cg)` runs at the start of group cg, before anything else of the group is requested: the place to ask for what epi(cg, ...)
// will need from memory (a gate) — loads return in order, so it must go out before the next group's weight fragments, and it then has
// the group's contraction and staging to arrive instead of standing between the staging barrier and the stores.
// `cg0`: the first of the NG groups this workgroup runs (round 6: the groups of one tile split over workgroups, layernorm.hip SPLIT).
template <int NG, class Epi, class Pre = Proj32NoPre>
__device__ __forceinline__ void proj32_run(const unsigned char* xs, unsigned char* os, const bf16_t* __restrict__ w, int Cout,
                                           const float* __restrict__ bias, Proj32W& W, int tid, Epi&& epi, Pre&& pre = Pre(), int cg0 = 0) {
  constexpr int NF = P32_TT / 16;
  const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const unsigned char* inl = xs + l15 * P32_RS + q * 16;
  Proj32W V;                                     // the other register pair: group cg + 1 is requested when group cg STARTS, so it has
                                                 // the whole group (contraction, staging, barrier, stores: ~2 us) to arrive, not just its tail
  // the B fragments of a step (4 k-steps x 2 row tiles) run a step ahead of their MFMAs (tapring.h); every group reads the same rows,
  // so the ring is carried from one group into the next
  bf16x8 ring[P32_KH][NF];
  ring_prime_step<NF, P32_RS, P32_KH>(ring, inl, 0);
#pragma unroll
  for (int cg = 0; cg < NG; ++cg) {
    Proj32W& cur = (cg & 1) ? V : W;
    Proj32W& nxt = (cg & 1) ? W : V;
    pre(cg0 + cg);
    if (cg + 1 < NG) {
      proj32_load(w, Cout, cg0 + cg + 1, 0, wave, lane, nxt.a);
      proj32_load(w, Cout, cg0 + cg + 1, 1, wave, lane, nxt.b);
    }
    __builtin_amdgcn_sched_barrier(0);             // the requests stay ahead of the contraction
    f32x4 acc[P32_CT][NF];
#pragma unroll
    for (int cc = 0; cc < P32_CT; ++cc)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](int g, const bf16x8 (&wf)[P32_KH][P32_CT]) __attribute__((always_inline)) {
      tap_ring_step<false, P32_KH, P32_CT, NF, P32_RS>(acc, ring, wf, inl, (g ^ 1) * (P32_KH * 64));      // step 1 refills with step 0's fragments
    };
    step(0, cur.a);
    step(1, cur.b);
    // this group's 32 x 256 outputs through LDS: 16-byte stores of whole rows
#pragma unroll
    for (int cc = 0; cc < P32_CT; ++cc) {
      const int col = (wave * P32_CT + cc) * 16 + q * 4;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias) bv = *(const f32x4*)(bias + (cg0 + cg) * P32_D + col);
#pragma unroll
      // 8-byte units XOR-swizzled by (row >> 2) & 3: the 16 lanes of a store (16 rows, one unit column; row stride 136 dwords = 8
      // mod 32) would hit 4 bank pairs four deep; with the swizzle they cover all 32 banks once
      for (int i = 0; i < NF; ++i) {
        const f32x4 v = acc[cc][i] + bv;
        *(uint2*)(os + (i * 16 + l15) * P32_RS + ((col * 2) ^ (((l15 >> 2) & 3) << 3))) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      }
    }
    __syncthreads();
    constexpr int OCH = P32_D / 8;                 // 16-byte chunks per row of the group
#pragma unroll
    for (int it = 0; it < P32_TT * OCH / P32_NT; ++it) {
      const int idx = it * P32_NT + tid;
      const int rr = idx / OCH, ch = idx - rr * OCH;
      const int sw = (rr >> 2) & 3;                // the row's unit swizzle: bit 1 moves the chunk, bit 0 swaps its halves
      uint4 v = *(const uint4*)(os + rr * P32_RS + ((ch ^ (sw >> 1)) << 4));
      if (sw & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      epi(cg0 + cg, rr, ch, v, it);
    }
    if (cg + 1 < NG) __syncthreads();          // the next group overwrites the staging rows
  }
}

