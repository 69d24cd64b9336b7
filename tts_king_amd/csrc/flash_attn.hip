// flash_attn.hip — FFT-block self-attention (d_k = 128) without the S x S matrices: forward with an online softmax, backward
// by recomputation from Q, K and the per-row log-sum-exp.
// reference: fs_two/transformer/Modules.py:14-24 (softmax(Q K^T / sqrt(d_k), PAD keys masked) V), SubLayers.py:44-60 (head
// split / merge); backward = autograd of those lines.
//
// Why (round 2): the first fused kernels (attn.hip) kept P (bf16, [B*H][S][S]) and the fp32 output for the backward, wrote dS,
// and left dK = dS^T Q / dV = P^T dO to two batched GEMMs: 23 MB of S x S traffic per decoder block and three launches for
// the backward (22 + 26 us), 24 us for the forward — 15 % of the train step for 5 % of its FLOPs.  Here:
//   forward   one workgroup per (utterance, head, 64 queries), K / V tiles of 64 keys double-buffered in LDS (next tile's loads
//             in flight during the MFMAs), running row max / sum, O rescaled per tile; only O and LSE = m + log(l) leave the chip.
//             Key tiles entirely past the utterance's length are never visited.
//   backward  delta = rowsum(dO o O); P = exp(S * scale - LSE) recomputed per tile;
//             bwd_q  (per 64 queries, loop over key tiles):   dS = P o (dP - delta) * scale,  dQ += dS K
//             bwd_kv (per 64 keys,    loop over query tiles): S^T, dP^T computed transposed (keys on the MFMA rows), so
//                                                              dV += P^T dO and dK += dS^T Q are plain A B products
//             — no atomics, each output tile is written once: deterministic.
// Fragment / LDS idioms (XOR-swizzled row-major images, ds_read_b64_tr_b16 images for operands whose contraction index is the
// memory row) are those of attn.hip (attn_common.h).  Two workgroups fit a CU (<= 80 KB of LDS each).
#include "attn_common.h"

namespace {

struct FlashArgs {
  const bf16_t* qkv;      // [B*S][3*d]: q | k | v, head h = columns h*128 .. of each part
  bf16_t* o;              // fwd out / bwd in: [B*S][d]
  float* o32;             // the same in fp32 (fwd out, bwd in; may be null): delta = rowsum(dO o O) is the term dP is cancelled
                          // against, and O rounded to bf16 leaves a 2^-9 error in it that dominates small dQ / dK gradients
  float* lse;             // [B*H][S]: fwd out (may be null), bwd in
  const bf16_t* dout;     // bwd: dO [B*S][d]
  float* delta;           // bwd: [B*H][S] workspace, written by bwd_q, read by bwd_kv
  bf16_t* dqkv;           // bwd out [B*S][3*d]
  const long long* lens;  // [B] valid keys per utterance (null = S)
  int S, H, d;
  float scale;
};

constexpr int STAGE_KV = KS_BYTES + VS_BYTES;            // one K tile (row-major image) + one V tile (contraction-major image)

// ------------------------------------------------------------------------------------------------------------ forward
// Workgroup -> (tile, batch*head) with the tiles of one (batch, head) on ONE XCD: consecutive workgroup ids go round the 8 XCDs, each
// with an L2 of its own, so the plain (blockIdx.x, blockIdx.y) order spreads the query tiles that share a head's K / V (or the key
// tiles that share its Q / dO) over all eight and every one of them fetches its own copy from memory.  Bijective when the number of
// (batch, head) pairs is a multiple of 8; otherwise the plain order.
__device__ __forceinline__ void xcd_tile(int& tile, int& z) {
  const int nx = gridDim.x, ny = gridDim.y;
  tile = blockIdx.x;
  z = blockIdx.y;
  if ((ny & 7) == 0) {
    const int L = blockIdx.x + nx * blockIdx.y;
    const int xcd = L & 7, slot = L >> 3;
    z = (slot / nx) * 8 + xcd;
    tile = slot - (slot / nx) * nx;
  }
}

// ------------------------------------------------------------------------------------------------ forward, P in registers
// Round 3.  The same tiling as flash_fwd_kernel (64 queries per workgroup, 16 per wave, key tiles of 64), with both products
// transposed so that P never leaves the registers:
//   S^T[key][query] = K Q^T        MFMA(A = K rows, B = Q rows): lane (lg, l15) holds keys 16*nt + 4*lg + r of ONE query, l15
//   O^T[dv][query] += V^T P^T      MFMA(A = V^T from the contraction-major V image, B = P^T): the B operand wants, for query l15, the 8
//                                  contraction slots 8*lg .. 8*lg+7 — and a contraction may be summed in any order, so slot (lg, e) is
//                                  DEFINED as key 16*(2u) + 4*lg + e (e < 4) / 16*(2u+1) + 4*lg + e - 4 (e >= 4) of 32-key step u:
//                                  exactly the eight values the lane already holds (two accumulators of S^T, packed to bf16).  The V^T
//                                  fragment follows the same slot -> key map through the rows its transposing read addresses.
// What goes: the 16 two-byte LDS stores of P per tile, the wait for them, their read-back, and 4 x 2 sixteen-lane DPP reductions per
// tile (a query's keys now sit in one lane's registers and in the three lanes 16, 32, 48 further on: two cross-row exchanges);
// the epilogue writes O^T as 8-byte (bf16 staging) / 16-byte (fp32 copy) pieces instead of 2- / 4-byte ones.
// V image for this kernel: [64 keys][128 dv], 32-byte block b of row k at block b ^ (k & 7): one transposing read touches keys
// 4*lg + (l15>>2) for lg = 0, 1 — eight consecutive rows — conflict-free.
__device__ __forceinline__ void store_tr8(unsigned char* dst, const uint4 (&r)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int kr = c >> 4, ts = c & 15;
    *(uint4*)(dst + kr * 256 + (((ts >> 1) ^ (kr & 7)) << 5) + ((ts & 1) << 4)) = r[i];
  }
}
// A fragment of V^T: 16 dv rows (block nblk), 32-key step u, slots as above: lo = keys 32u + 4lg + 0..3, hi = keys 32u + 16 + 4lg + 0..3
__device__ __forceinline__ bf16x8 frag_tr8(const unsigned char* base, int nblk, int u, int l15, int lg) {
  const int k0 = u * 32 + 4 * lg + (l15 >> 2), k1 = k0 + 16;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(base + k0 * 256 + ((nblk ^ (k0 & 7)) << 5) + ((l15 & 3) << 3)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(base + k1 * 256 + ((nblk ^ (k1 & 7)) << 5) + ((l15 & 3) << 3)));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// A row-major fragment (rows r0 + l15, r0 a multiple of 16; k step ks = 32 contraction elements) read from a store_tr8 IMAGE: the
// 16-byte chunk c = 4 ks + lg of row k sits in 32-byte block (c >> 1) ^ (k & 7), half c & 1 — a group of 16 lanes (8 rows at chunk
// c, the 8 other rows at c + 1) covers 16 distinct 16-byte units: conflict-free (tools/debug/lds_bank_model.py).  So an operand that
// is multiplied BOTH ways (K on the query side of the backward, Q and dO on its key side) needs one LDS image, not two.
__device__ __forceinline__ bf16x8 frag_rows8(const unsigned char* base, int r0, int ks, int l15, int lg) {
  const int row = r0 + l15, c = ks * 4 + lg;
  return *(const bf16x8*)(base + row * 256 + ((((c >> 1) ^ (row & 7)) << 5) | ((c & 1) << 4)));
}
// max / sum over the four lanes l15, l15 + 16, l15 + 32, l15 + 48 (the lane groups that share a query)
__device__ __forceinline__ float cross16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float cross16_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__global__ __launch_bounds__(256, 2) void flash_fwd_t_kernel(const FlashArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_KV];      // 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  int tile_x, z;
  xcd_tile(tile_x, z);
  const int b = z / a.H, h = z - b * a.H;
  const int q0 = tile_x * TQ;
  const int S = a.S, ld = 3 * a.d;
  const int len = a.lens ? min((int)a.lens[b], S) : S;
  const int ntk = (len + TK - 1) / TK;
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;

  uint4 rk[4], rv[4];
  {  // Q goes through the K slot of stage 1 (free until tile 1 is stored); tile 0 of K / V into stage 0
    uint4 rq[4];
    load_tile(rq, base, ld, q0, S, tid);
    if (ntk > 0) { load_tile(rk, base + a.d, ld, 0, S, tid); load_tile(rv, base + 2 * a.d, ld, 0, S, tid); }
    store_rows(smem + STAGE_KV, rq, tid);
    if (ntk > 0) { store_rows(smem, rk, tid); store_tr8(smem + KS_BYTES, rv, tid); }
  }
  __syncthreads();
  bf16x8 qa[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qa[ks] = frag_rows(smem + STAGE_KV, wave * 16, ks, l15, lg);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                       // every wave holds its Q fragments: stage 1 may be refilled

  float m = -INFINITY, l = 0.f;                          // running maximum / sum of THIS lane's query (column l15 of the wave's 16)
  f32x4 oacc[8];                                         // O^T: rows dv = 16*nb + 4*lg + r, column = the query
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < ntk; ++j) {
    const unsigned char* Ks = smem + (j & 1) * STAGE_KV;
    const unsigned char* Vs = Ks + KS_BYTES;
    if (j + 1 < ntk) {     // next tile's loads fly during this tile's arithmetic
      load_tile(rk, base + a.d, ld, (j + 1) * TK, S, tid);
      load_tile(rv, base + 2 * a.d, ld, (j + 1) * TK, S, tid);
    }
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {  // S^T: operand i = K rows of key block i & 3, k-step i >> 2, eight reads ahead of their MFMAs (FragStream, attn_common.h)
      FragStream<16, 8, 1> ks_;
      auto kfrag = [&](int i) __attribute__((always_inline)) { return frag_rows(Ks, (i & 3) * 16, i >> 2, l15, lg); };
      ks_.prime(kfrag);
      ks_.run(kfrag, [&](int i, bf16x8 f) __attribute__((always_inline)) { s[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, qa[i >> 2], s[i & 3], 0, 0, 0); });
    }
    // the first eight V^T operands are requested here: they arrive under the softmax arithmetic
    FragStream<16, 8, 2> vs_;
    auto vfrag = [&](int i) __attribute__((always_inline)) { return frag_tr8(Vs, i & 7, i >> 3, l15, lg); };
    vs_.prime(vfrag);
    __builtin_amdgcn_sched_barrier(0);
    float tm = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j * TK + nt * 16 + lg * 4 + r;
        s[nt][r] = key < len ? s[nt][r] * a.scale : -INFINITY;
        tm = fmaxf(tm, s[nt][r]);
      }
    // online softmax: every tile j < ntk holds a valid key, so the new maximum is finite
    tm = cross16_max(tm);
    const float mn = fmaxf(m, tm);
    const float corr = __expf(m - mn);                   // exp(-inf) = 0 on the first tile
    float ts = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float p = __expf(s[nt][r] - mn); s[nt][r] = p; ts += p; }
    ts = cross16_sum(ts);
    l = l * corr + ts;
    m = mn;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) oacc[nb] *= corr;
    bf16x8 pb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // P^T fragment of 32-key step u, straight from the accumulators (slot order = the order frag_tr8 reads V^T's keys in)
      const unsigned p0 = pack_bf2(s[2 * u][0], s[2 * u][1]), p1 = pack_bf2(s[2 * u][2], s[2 * u][3]);
      const unsigned p2 = pack_bf2(s[2 * u + 1][0], s[2 * u + 1][1]), p3 = pack_bf2(s[2 * u + 1][2], s[2 * u + 1][3]);
      pb[u] = __builtin_bit_cast(bf16x8, make_uint4(p0, p1, p2, p3));
    }
    __builtin_amdgcn_sched_barrier(0);
    vs_.run(vfrag, [&](int i, bf16x8 f) __attribute__((always_inline)) { oacc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, pb[i >> 3], oacc[i & 7], 0, 0, 0); });
    if (j + 1 < ntk) {
      unsigned char* nx = smem + ((j + 1) & 1) * STAGE_KV;   // last read in iteration j-1: every wave passed that iteration's barrier
      store_rows(nx, rk, tid);
      store_tr8(nx + KS_BYTES, rv, tid);
    }
    __syncthreads();
  }

  // ---- O^T / l: lane holds dv = 16*nb + 4*lg .. +3 of query l15 -> 8-byte bf16 pieces into the staging tile, 16-byte fp32 pieces
  // straight to memory; LSE from the lanes of group 0
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const int q = q0 + wave * 16 + l15;
  if (a.lse && lg == 0 && q < S) a.lse[(int64_t)z * S + q] = l > 0.f ? m + __logf(l) : 0.f;
  if (a.o32 && q < S) {
    float* dst = a.o32 + ((int64_t)b * S + q) * a.d + h * DK + 4 * lg;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) *(f32x4*)(dst + nb * 16) = oacc[nb] * inv;
  }
  unsigned char* Os = smem;   // 64 x 272 B = 17 KiB over stage 0 (dead: the loop ended with a barrier)
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
    *(uint2*)(Os + (wave * 16 + l15) * OS_RS + (nb * 16 + 4 * lg) * 2) =
        make_uint2(pack_bf2(oacc[nb][0] * inv, oacc[nb][1] * inv), pack_bf2(oacc[nb][2] * inv, oacc[nb][3] * inv));
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (q0 + row < S) *(uint4*)(a.o + ((int64_t)b * S + q0 + row) * a.d + h * DK + ch * 8) = *(const uint4*)(Os + row * OS_RS + ch * 16);
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Round 3: flash_fwd_t_kernel's idiom on both sides of the backward.  Every product is taken transposed so that the tile that is both
// an MFMA result and the next MFMA's operand (dS for dQ; P and dS for dV and dK) is packed from the accumulators straight into a B
// operand — under the slot order frag_tr8 reads its transposed A operand in — instead of going through 16 (32 on the key side)
// two-byte LDS stores, a wait and a read-back per tile:
//   query side, per (utterance, head, 64 queries), queries on the MFMA COLUMNS (one query per lane column l15):
//     S^T = K Q^T, dP^T = V dO^T  (A = K / V rows of the tile, B = this wave's Q / dO fragments, held for the whole loop)
//     dS^T = P^T o (dP^T - delta[q]) * scale           dQ^T[d][q] += K^T dS^T   (A = K^T: transposing reads of the K image)
//   key side, per (utterance, head, 64 keys), keys on the columns:
//     S = Q K^T, dP = dO V^T      (A = Q / dO rows of the query tile, B = this wave's K / V fragments)
//     dV^T[d][k] += dO^T P,   dK^T[d][k] += Q^T dS     (A = dO^T / Q^T: transposing reads)
// Outputs leave as in flash_fwd_t_kernel: 8-byte bf16 pieces into a staging tile, full rows to memory.  Same sums in a different
// order (a contraction's slots are permuted, fp32 accumulation): results differ from flash_bwd_kernel's in the last bits only.
__device__ __forceinline__ void flash_bwd_q_t_body(const FlashArgs& a, unsigned char* smem) {      // uses 49,152 B of smem
  unsigned char* Kr = smem;                          // staging of the workgroup's Q rows (before the loop)
  unsigned char* Vr = smem + KS_BYTES;               // V tile, row-major image            (A operand of dP^T = V dO^T); dO staging before the loop
  unsigned char* Kt = Vr + KS_BYTES;                 // K tile, store_tr8 image            (A operand of S^T = K Q^T as rows, of dQ^T += K^T dS^T transposed)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  int tile_x, z;
  xcd_tile(tile_x, z);
  const int b = z / a.H, h = z - b * a.H;
  const int q0 = tile_x * TQ;
  const int S = a.S, ld = 3 * a.d;
  const int len = a.lens ? min((int)a.lens[b], S) : S;
  const int ntk = (len + TK - 1) / TK;
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;
  const bf16_t* dob = a.dout + (int64_t)b * S * a.d + h * DK;

  uint4 rk[4], rv[4];
  {
    uint4 rq[4], rd[4];
    load_tile(rq, base, ld, q0, S, tid);
    load_tile(rd, dob, a.d, q0, S, tid);
    if (ntk > 0) { load_tile(rk, base + a.d, ld, 0, S, tid); load_tile(rv, base + 2 * a.d, ld, 0, S, tid); }
    store_rows(Kr, rq, tid);
    store_rows(Vr, rd, tid);
  }
  const int qme = q0 + wave * 16 + l15;               // this lane's query
  const float lse = qme < S ? a.lse[(int64_t)z * S + qme] : 0.f;
  const float delta = qme < S ? a.delta[(int64_t)z * S + qme] : 0.f;
  __syncthreads();
  bf16x8 qb[4], db[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { qb[ks] = frag_rows(Kr, wave * 16, ks, l15, lg); db[ks] = frag_rows(Vr, wave * 16, ks, l15, lg); }

  f32x4 dq[8];                                        // dQ^T: rows d = 16*nb + 4*lg + r, column = the query
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) dq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < ntk; ++j) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                  // the previous tile's (or the Q / dO staging's) LDS reads are done
    store_tr8(Kt, rk, tid);                           // ONE image of K: read as rows for S^T (frag_rows8) and transposed for dQ^T (frag_tr8)
    store_rows(Vr, rv, tid);
    if (j + 1 < ntk) { load_tile(rk, base + a.d, ld, (j + 1) * TK, S, tid); load_tile(rv, base + 2 * a.d, ld, (j + 1) * TK, S, tid); }
    __syncthreads();
    f32x4 st[4], dpt[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { st[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dpt[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    {  // operand i: bit 0 = S^T (K image) | dP^T (V rows), bits 1-2 = key block, bits 3-4 = k-step; eight reads ahead (FragStream)
      FragStream<32, 8, 1> ab;
      auto frag = [&](int i) __attribute__((always_inline)) {
        return (i & 1) ? frag_rows(Vr, ((i >> 1) & 3) * 16, i >> 3, l15, lg) : frag_rows8(Kt, ((i >> 1) & 3) * 16, i >> 3, l15, lg);
      };
      ab.prime(frag);
      ab.run(frag, [&](int i, bf16x8 f) __attribute__((always_inline)) {
        const int nt = (i >> 1) & 3, ks = i >> 3;
        if (i & 1) dpt[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, db[ks], dpt[nt], 0, 0, 0);
        else st[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, qb[ks], st[nt], 0, 0, 0);
      });
    }
    // the first eight K^T operands of dQ^T are requested here: they arrive under the exponentials
    FragStream<16, 8, 2> kt;
    auto ktfrag = [&](int i) __attribute__((always_inline)) { return frag_tr8(Kt, i & 7, i >> 3, l15, lg); };
    kt.prime(ktfrag);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j * TK + nt * 16 + lg * 4 + r;
        const float p = key < len ? __expf(st[nt][r] * a.scale - lse) : 0.f;
        st[nt][r] = a.scale * p * (dpt[nt][r] - delta);
      }
    bf16x8 sb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      sb[u] = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(st[2 * u][0], st[2 * u][1]), pack_bf2(st[2 * u][2], st[2 * u][3]),
                                                    pack_bf2(st[2 * u + 1][0], st[2 * u + 1][1]), pack_bf2(st[2 * u + 1][2], st[2 * u + 1][3])));
    __builtin_amdgcn_sched_barrier(0);
    kt.run(ktfrag, [&](int i, bf16x8 f) __attribute__((always_inline)) { dq[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, sb[i >> 3], dq[i & 7], 0, 0, 0); });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned char* Os = smem;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
    *(uint2*)(Os + (wave * 16 + l15) * OS_RS + (nb * 16 + 4 * lg) * 2) = make_uint2(pack_bf2(dq[nb][0], dq[nb][1]), pack_bf2(dq[nb][2], dq[nb][3]));
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (q0 + row < S) *(uint4*)(a.dqkv + ((int64_t)b * S + q0 + row) * ld + h * DK + ch * 8) = *(const uint4*)(Os + row * OS_RS + ch * 16);
  }
}

__device__ __forceinline__ void flash_bwd_kv_t_body(const FlashArgs& a, unsigned char* smem) {     // uses 66,048 B of smem
  unsigned char* Qr = smem;                          // staging of the workgroup's K rows (before the loop)
  unsigned char* Dr = smem + KS_BYTES;               // staging of its V rows
  unsigned char* Qt = Dr + KS_BYTES;                 // Q tile, store_tr8 image            (A operand of S = Q K^T as rows, of dK^T += Q^T dS transposed)
  unsigned char* Dt = Qt + VS_BYTES;                 // dO tile, store_tr8 image           (A operand of dP = dO V^T as rows, of dV^T += dO^T P transposed)
  float* LD = (float*)(Dt + VS_BYTES);               // [64] LSE | [64] delta of the query tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  int tile_x, z;
  xcd_tile(tile_x, z);
  const int b = z / a.H, h = z - b * a.H;
  const int k0 = tile_x * TK;
  const int S = a.S, ld = 3 * a.d;
  const int len = a.lens ? min((int)a.lens[b], S) : S;
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;
  const bf16_t* dob = a.dout + (int64_t)b * S * a.d + h * DK;
  const int ntq = (S + TQ - 1) / TQ;
  const bool key_live = k0 + wave * 16 + l15 < len;   // this lane's key (column l15 of the wave's 16)

  f32x4 dk[8], dv[8];                                 // dK^T / dV^T: rows d = 16*nb + 4*lg + r, column = the key
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) { dk[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  if (k0 < len) {        // (a key tile entirely past the utterance gets zero gradients: nothing to compute)
    uint4 rq[4], rd[4];
    load_tile(rq, base, ld, 0, S, tid);
    load_tile(rd, dob, a.d, 0, S, tid);
    {  // K and V fragments of this wave's 16 keys: through LDS once (Qr / Dr slots), then registers
      uint4 rk[4], rv[4];
      load_tile(rk, base + a.d, ld, k0, S, tid);
      load_tile(rv, base + 2 * a.d, ld, k0, S, tid);
      store_rows(Qr, rk, tid);
      store_rows(Dr, rv, tid);
    }
    __syncthreads();
    bf16x8 kb[4], vb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kb[ks] = frag_rows(Qr, wave * 16, ks, l15, lg); vb[ks] = frag_rows(Dr, wave * 16, ks, l15, lg); }

    for (int i = 0; i < ntq; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();                            // the previous tile's (or the K / V staging's) LDS reads are done
      store_tr8(Qt, rq, tid);                       // one image each of Q and dO, read both ways (frag_rows8 / frag_tr8)
      store_tr8(Dt, rd, tid);
      if (tid < 128) {
        const int q = i * TQ + (tid & 63);
        const float* src = tid < 64 ? a.lse : a.delta;
        LD[tid] = q < S ? src[(int64_t)z * S + q] : 0.f;
      }
      if (i + 1 < ntq) { load_tile(rq, base, ld, (i + 1) * TQ, S, tid); load_tile(rd, dob, a.d, (i + 1) * TQ, S, tid); }
      __syncthreads();
      f32x4 s[4], dp[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { s[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      {  // operand i: bit 0 = S (Q image) | dP (dO image), bits 1-2 = query block, bits 3-4 = k-step; eight reads ahead (FragStream)
        FragStream<32, 8, 1> ab;
        auto frag = [&](int i) __attribute__((always_inline)) { return frag_rows8((i & 1) ? Dt : Qt, ((i >> 1) & 3) * 16, i >> 3, l15, lg); };
        ab.prime(frag);
        ab.run(frag, [&](int i, bf16x8 f) __attribute__((always_inline)) {
          const int nt = (i >> 1) & 3, ks = i >> 3;
          if (i & 1) dp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, vb[ks], dp[nt], 0, 0, 0);
          else s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, kb[ks], s[nt], 0, 0, 0);
        });
      }
      // the first eight transposed operands of dV^T / dK^T are requested here: they arrive under the exponentials
      FragStream<32, 8, 2> tq;
      auto tfrag = [&](int i) __attribute__((always_inline)) { return frag_tr8((i & 1) ? Qt : Dt, (i >> 1) & 7, i >> 4, l15, lg); };
      tq.prime(tfrag);
      // rows = queries 16*nt + 4*lg + r of the tile, column = this lane's key
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 lq = *(const f32x4*)(LD + nt * 16 + 4 * lg), dq_ = *(const f32x4*)(LD + 64 + nt * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = i * TQ + nt * 16 + 4 * lg + r;
          const float p = (key_live && q < S) ? __expf(s[nt][r] * a.scale - lq[r]) : 0.f;
          s[nt][r] = p;
          dp[nt][r] = a.scale * p * (dp[nt][r] - dq_[r]);
        }
      }
      bf16x8 pb[2], sb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pb[u] = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(s[2 * u][0], s[2 * u][1]), pack_bf2(s[2 * u][2], s[2 * u][3]),
                                                      pack_bf2(s[2 * u + 1][0], s[2 * u + 1][1]), pack_bf2(s[2 * u + 1][2], s[2 * u + 1][3])));
        sb[u] = __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(dp[2 * u][0], dp[2 * u][1]), pack_bf2(dp[2 * u][2], dp[2 * u][3]),
                                                      pack_bf2(dp[2 * u + 1][0], dp[2 * u + 1][1]), pack_bf2(dp[2 * u + 1][2], dp[2 * u + 1][3])));
      }
      __builtin_amdgcn_sched_barrier(0);
      tq.run(tfrag, [&](int i, bf16x8 f) __attribute__((always_inline)) {        // operand i: bit 0 = dV^T (dO^T) | dK^T (Q^T), bits 1-3 = d block, bit 4 = 32-query step
        const int nb = (i >> 1) & 7, u = i >> 4;
        if (i & 1) dk[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, sb[u], dk[nb], 0, 0, 0);
        else dv[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, pb[u], dv[nb], 0, 0, 0);
      });
    }
  }
  // ---- dK^T, dV^T -> bf16 -> LDS (8-byte pieces) -> full-row stores into the k / v columns of dqkv
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned char* Os = smem;                   // two 64 x 272 B tiles = 34 KiB
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    *(uint2*)(Os + (wave * 16 + l15) * OS_RS + (nb * 16 + 4 * lg) * 2) = make_uint2(pack_bf2(dk[nb][0], dk[nb][1]), pack_bf2(dk[nb][2], dk[nb][3]));
    *(uint2*)(Os + 64 * OS_RS + (wave * 16 + l15) * OS_RS + (nb * 16 + 4 * lg) * 2) =
        make_uint2(pack_bf2(dv[nb][0], dv[nb][1]), pack_bf2(dv[nb][2], dv[nb][3]));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (k0 + row < S) {
      bf16_t* dst = a.dqkv + ((int64_t)b * S + k0 + row) * ld + h * DK + ch * 8;
      *(uint4*)(dst + a.d) = *(const uint4*)(Os + row * OS_RS + ch * 16);
      *(uint4*)(dst + 2 * a.d) = *(const uint4*)(Os + 64 * OS_RS + row * OS_RS + ch * 16);
    }
  }
}

// delta[z][q] = sum_d dO[q][h*128 + d] * O[q][h*128 + d] (O in fp32 when kept): a 16-lane group per (row, head), 16 rows per workgroup
__global__ __launch_bounds__(256) void flash_delta_kernel(const FlashArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
  const int z = blockIdx.y, b = z / a.H, h = z - b * a.H;
  const int q = blockIdx.x * 16 + wave * 4 + lg;
  const int S = a.S;
  float acc = 0.f;
  if (q < S) {
    const uint4 x = *(const uint4*)(a.dout + ((int64_t)b * S + q) * a.d + h * DK + l15 * 8);
    if (a.o32) {
      const float* op = a.o32 + ((int64_t)b * S + q) * a.d + h * DK + l15 * 8;
      const f32x4 y0 = *(const f32x4*)op, y1 = *(const f32x4*)(op + 4);
      acc = __uint_as_float(x.x << 16) * y0[0] + __uint_as_float(x.x & 0xFFFF0000u) * y0[1] + __uint_as_float(x.y << 16) * y0[2] +
            __uint_as_float(x.y & 0xFFFF0000u) * y0[3] + __uint_as_float(x.z << 16) * y1[0] + __uint_as_float(x.z & 0xFFFF0000u) * y1[1] +
            __uint_as_float(x.w << 16) * y1[2] + __uint_as_float(x.w & 0xFFFF0000u) * y1[3];
    } else {
      const uint4 y = *(const uint4*)(a.o + ((int64_t)b * S + q) * a.d + h * DK + l15 * 8);
      acc = __uint_as_float(x.x << 16) * __uint_as_float(y.x << 16) + __uint_as_float(x.x & 0xFFFF0000u) * __uint_as_float(y.x & 0xFFFF0000u) +
            __uint_as_float(x.y << 16) * __uint_as_float(y.y << 16) + __uint_as_float(x.y & 0xFFFF0000u) * __uint_as_float(y.y & 0xFFFF0000u) +
            __uint_as_float(x.z << 16) * __uint_as_float(y.z << 16) + __uint_as_float(x.z & 0xFFFF0000u) * __uint_as_float(y.z & 0xFFFF0000u) +
            __uint_as_float(x.w << 16) * __uint_as_float(y.w << 16) + __uint_as_float(x.w & 0xFFFF0000u) * __uint_as_float(y.w & 0xFFFF0000u);
    }
  }
  acc = quad16_sum(acc);
  if (q < S && l15 == 0) a.delta[(int64_t)z * S + q] = acc;
}

// The query side (dQ) and the key side (dK, dV) as ONE grid (blockIdx.z = side): with delta precomputed neither needs the other, and a
// side alone is 7 x 32 workgroups for the decoder — one per CU, a single wave per SIMD on a chain of dependent loads, LDS round trips
// and VALU work; together two workgroups share a CU and fill each other's stalls.
__global__ __launch_bounds__(256, 2) void flash_bwd_t_kernel(const FlashArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KS_BYTES + 2 * VS_BYTES + 512];           // 66,048 B (the key side's)
  if (blockIdx.z == 0) flash_bwd_q_t_body(a, smem);
  else flash_bwd_kv_t_body(a, smem);
}

}  // namespace

extern "C" int ttsk_flash_attention_fwd(const void* qkv_bf16, void* o_bf16, float* o_f32, float* lse, const int64_t* lens, int B, int H,
                                        int S, int d, float scale, void* stream) {
  TTSK_REQUIRE(qkv_bf16 && o_bf16, "flash_attention_fwd: null pointer");
  TTSK_REQUIRE(B > 0 && H > 0 && S > 0 && d == H * DK, "flash_attention_fwd: head size must be 128 (d = %d, H = %d)", d, H);
  TTSK_REQUIRE(B * H <= 65535 && (int64_t)S * 3 * d * 2 < ((int64_t)1 << 31), "flash_attention_fwd: sizes out of range");
  FlashArgs a{(const bf16_t*)qkv_bf16, (bf16_t*)o_bf16, o_f32, lse, nullptr, nullptr, nullptr, (const long long*)lens, S, H, d, scale};
  hipLaunchKernelGGL(flash_fwd_t_kernel, dim3((S + TQ - 1) / TQ, B * H), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_flash_attention_bwd(const void* qkv_bf16, const void* o_bf16, const float* o_f32, const void* dout_bf16, const float* lse,
                                        float* delta_ws, int delta_ready, void* dqkv_bf16, const int64_t* lens, int B, int H, int S, int d,
                                        float scale, void* stream) {
  TTSK_REQUIRE(qkv_bf16 && o_bf16 && dout_bf16 && lse && delta_ws && dqkv_bf16, "flash_attention_bwd: null pointer");
  TTSK_REQUIRE(B > 0 && H > 0 && S > 0 && d == H * DK, "flash_attention_bwd: head size must be 128");
  TTSK_REQUIRE(B * H <= 65535 && (int64_t)S * 3 * d * 2 < ((int64_t)1 << 31), "flash_attention_bwd: sizes out of range");
  FlashArgs a{(const bf16_t*)qkv_bf16, (bf16_t*)o_bf16, (float*)o_f32, (float*)lse, (const bf16_t*)dout_bf16, delta_ws, (bf16_t*)dqkv_bf16,
              (const long long*)lens, S, H, d, scale};
  if (!delta_ready) hipLaunchKernelGGL(flash_delta_kernel, dim3((S + 15) / 16, B * H), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(flash_bwd_t_kernel, dim3((S + TQ - 1) / TQ, B * H, 2), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
