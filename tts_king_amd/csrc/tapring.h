// tapring.h — the inner step of the FS2 window kernel (ffn_conv.hip): one weight step held in registers (KS k-steps x CT channel tiles)
// against NF frame tiles whose activation fragments are read from LDS.
//
// What the compiler does with the plain loop nest (measured on gfx950, tools/debug/wc_stamps.py): it sinks every ds_read_b128 to just
// before the two MFMAs that use it on one k-step in four, so the LDS latency is exposed NF times per step; and wherever a weight
// request sits under an `if`, its s_waitcnt vmcnt(N) assumes no younger request exists and every step starts by waiting for the
// request issued just before it.  A wave that has its SIMD to itself kept the MFMA pipe 60 % busy, two waves 84 %.  Here:
//   * the fragments go through a ring of NF registers, each refilled right after its MFMAs with the same frame tile's fragment of the
//     next k-step (of the next step's first k-step at the end), and sched_group_barrier holds the scheduler to that order: every read
//     is NF * CT MFMAs ahead of its use;
//   * the weights come by buffer loads — one resource descriptor, one per-lane byte offset for the whole kernel, the step's distance
//     in a scalar register: no per-lane 64-bit pointer arithmetic between the MFMAs;
//   * callers issue their weight requests unconditionally (clamped past the end) so that the wait counts are exact.
// Same-box A/B (tools/ab_libs.sh): FS2 step 2.670 -> 2.644 ms.  In cycles the tap loop of w_1 went 41.1k -> 36.5k (32.3k = its MFMAs
// alone), in time less: the loop runs at the clock the chip holds under this load (1.66-1.94 GHz across boxes, never 2.4), and the
// wave in slot 0 of a SIMD finishes 7 us before the one in slot 1 (HW_ID stamps: waves w and w + 4 share a SIMD; slot 0 issues whenever it
// can, whatever s_setprio says), which then has the SIMD to itself.  Splitting the pair's four channel tiles 3 : 1 made both end
// together — at the same 36 k cycles: the loop is at the rate the pipe sustains with an LDS read per two MFMAs, not losing a tail.  HiFi-GAN's pair kernels
// (convwin.hip: two 4-wave workgroups per CU, unconditional requests in c1 already) measured the same with the buffer loads and 1 %
// slower with the ring — they keep their plain loops.
#pragma once
#include "common.h"

// a read-only buffer over `bytes` bytes at p (reads past the end return 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weights_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 frag_load(__amdgpu_buffer_rsrc_t r, int lane_off, int uniform_off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 0));
}
// fragment-major packs ([tap][k-step][cout tile][64 lanes][8]): the KS x CT fragments of one step; consecutive cout tiles are 1 KiB apart
template <int KS, int CT>
__device__ __forceinline__ void frags_load(bf16x8 (&w)[KS][CT], __amdgpu_buffer_rsrc_t r, int lane_off, int step_off, int kstep_bytes) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) w[ks][cc] = frag_load(r, lane_off + cc * 1024, step_off + ks * kstep_bytes);
}

// ... with one byte offset per cout tile (a kernel that permutes the rows of its A operand: ffn_conv.hip's direct-store epilogue)
template <int KS, int CT>
__device__ __forceinline__ void frags_load(bf16x8 (&w)[KS][CT], __amdgpu_buffer_rsrc_t r, const int (&lane_off)[CT], int step_off, int kstep_bytes) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) w[ks][cc] = frag_load(r, lane_off[cc], step_off + ks * kstep_bytes);
}

// A window position is a byte offset into the workgroup's LDS array (this lane's row and 16-byte column of the window, rows RS bytes
// apart).  LdsPos turns it into address registers that go through an empty asm, so that they stay THE address registers of the step's
// reads and every (frame tile, k-step) distance lands in the instruction's 16-bit offset field (one register per TPB frame tiles: what
// the field reaches).  Left alone, the compiler re-bases the reads on some other register and spends a v_add_u32 per ds_read_b128 —
// one more instruction between every two MFMAs of a wave that already issues five there.
template <int NF, int RS, int KS>
struct LdsPos {
  static constexpr int TPB = (65536 - KS * 64) / (16 * RS) < NF ? (65536 - KS * 64) / (16 * RS) : NF, NB = (NF + TPB - 1) / TPB;
  unsigned base[NB];
  __device__ __forceinline__ explicit LdsPos(unsigned off) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      base[j] = off + j * TPB * 16 * RS;
      asm volatile("" : "+v"(base[j]));
    }
  }
  __device__ __forceinline__ bf16x8 frag(const unsigned char* lds, int i, int ks) const {
    return *(const bf16x8*)(lds + base[i / TPB] + (i % TPB) * 16 * RS + ks * 64);
  }
};
// ring[i] = frame tile i's fragment of the first k-step at position `inp`
template <int NF, int RS>
__device__ __forceinline__ void ring_prime(bf16x8 (&ring)[NF], const unsigned char* lds, unsigned inp) {
  const LdsPos<NF, RS, 1> p(inp);
#pragma unroll
  for (int i = 0; i < NF; ++i) ring[i] = p.frag(lds, i, 0);
  __builtin_amdgcn_sched_barrier(0);      // all of them in flight before the first MFMA (the scheduler otherwise sinks each to its use)
}
// acc += w (x) window at `inp`; on return the ring holds the first k-step's fragments at `nxt` (the next step's window position; the
// last step passes any readable position)
// FIRST: this step starts the sums — its first k-step multiplies onto the constant 0 instead of onto accumulators somebody had to zero (CT * NF * 4
// v_mov per conv, and a VALU instruction costs a quarter of an MFMA: DESIGN.md 8.2); same sums, bit for bit.
template <bool F16, int KS, int CT, int NF, int RS, bool FIRST = false>
__device__ __forceinline__ void tap_ring(f32x4 (&acc)[CT][NF], bf16x8 (&ring)[NF], const bf16x8 (&w)[KS][CT], const unsigned char* lds, unsigned inp,
                                         unsigned nxt) {
  const LdsPos<NF, RS, KS> pi(inp), pn(nxt);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int idx = 0; idx < KS * NF; ++idx) {
    const int ks = idx / NF, i = idx % NF;
    const bf16x8 Bf = ring[i];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<F16>(w[ks][cc], Bf, (FIRST && ks == 0) ? zero4 : acc[cc][i]);
    ring[i] = ks + 1 < KS ? pi.frag(lds, i, ks + 1) : pn.frag(lds, i, 0);
  }
#pragma unroll
  for (int idx = 0; idx < KS * NF; ++idx) {
    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);      // CT MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
  }
}

// The same with a whole step of fragments in flight (KS x NF registers): for short tiles (NF = 2: the 32-row tiles of gemm_ln.hip) one
// k-step ahead would be 2 * CT MFMAs, less than the LDS latency.  On return the ring holds the step at `nxt`.
template <int NF, int RS, int KS>
__device__ __forceinline__ void ring_prime_step(bf16x8 (&ring)[KS][NF], const unsigned char* lds, unsigned inp) {
  const LdsPos<NF, RS, KS> p(inp);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int i = 0; i < NF; ++i) ring[ks][i] = p.frag(lds, i, ks);
  __builtin_amdgcn_sched_barrier(0);
}
template <bool F16, int KS, int CT, int NF, int RS>
__device__ __forceinline__ void tap_ring_step(f32x4 (&acc)[CT][NF], bf16x8 (&ring)[KS][NF], const bf16x8 (&w)[KS][CT], const unsigned char* lds,
                                              unsigned nxt) {
  const LdsPos<NF, RS, KS> pn(nxt);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const bf16x8 Bf = ring[ks][i];
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) acc[cc][i] = mfma16<F16>(w[ks][cc], Bf, acc[cc][i]);
      ring[ks][i] = pn.frag(lds, i, ks);
    }
#pragma unroll
  for (int idx = 0; idx < KS * NF; ++idx) {
    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
}

// A run of N LDS operands `frag(i)`, each feeding MPF MFMAs (`mma(i, operand)` issues them): the operands go through a ring of D
// registers, each refilled right after its MFMAs with operand i + D, and sched_group_barrier holds the scheduler to that order (READS =
// LDS instructions per operand: 1 for a ds_read_b128, 2 for a pair of transposing reads).  Left to the compiler every operand is read just before its MFMA and waited for
// with lgkmcnt(0): an LDS round trip per MFMA.  prime() may be called early (before a stretch of VALU work the first D reads can fly under).
template <int N, int D, int READS, int MPF = 1>
struct FragStream {
  bf16x8 ring[D];
  template <class Frag>
  __device__ __forceinline__ void prime(Frag&& frag) {
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = frag(d);
    __builtin_amdgcn_sched_barrier(0);      // (the reads of the run must not be mistaken for these by the scheduler's groups, nor these for theirs)
  }
  template <class Frag, class Mma>
  __device__ __forceinline__ void run(Frag&& frag, Mma&& mma) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      mma(i, ring[i % D]);
      if (i + D < N) ring[i % D] = frag(i + D);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, MPF, 0);
      if (i + D < N) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
};
