// attn.hip — fused multi-head self-attention for the FFT blocks (d_k = 128), forward and the query-side backward.
// reference: fs_two/transformer/Modules.py:14-24 (softmax(Q K^T / sqrt(d_k) masked on PAD keys) V) and
// fs_two/transformer/SubLayers.py:44-60 (head split / merge), autograd for the backward.
//
// Forward: one 256-thread workgroup per (utterance, head, 64 queries).  Q stays in LDS; K and V stream through LDS in
// 64-key tiles.  Each wave owns 16 query rows: S = Q K^T on v_mfma_f32_16x16x32_bf16 (keys on the MFMA columns), a
// first sweep over the keys accumulates the row maxima and sums, a second sweep recomputes S, normalises, rounds P to
// bf16 (the same rounding point as the unfused path), optionally stores P for the backward, and multiplies P V with V
// read through the transposing LDS read.  The S x S score matrix never exists in HBM in fp32 and three launches
// (scores GEMM, softmax, P.V GEMM) become one.
//
// Backward (query side): dP = dO V^T per key tile, dS = P o (dP - rowsum(dO o O)) / sqrt(d_k) with P from the forward,
// dS stored (bf16) for the key-side products dK = dS^T Q, dV = P^T dO (batched GEMMs), and dQ = dS K accumulated over the
// key tiles — one launch instead of dP GEMM + softmax backward + dQ GEMM.
#include "attn_common.h"

namespace {

struct AttnArgs {
  const bf16_t* qkv;     // [B*S][3*d]: q | k | v, head h = columns h*128 .. of each part
  bf16_t* o;             // [B*S][d]
  float* o32;            // optional fp32 copy of o (un-rounded sum_k P V): lets the backward form rowsum(P o dP) = dO . o32
  bf16_t* probs;         // [B*H][S][Sp] or null
  const long long* lens; // [B] valid keys per utterance (null = S)
  int S, Sp, H, d;
  float scale;
};

constexpr int KBLK = 7;                         // key tiles resident in LDS at once (448 keys = 112 KiB)

__global__ __launch_bounds__(256, 1) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[QS_BYTES + KBLK * KS_BYTES + VS_BYTES + PS_BYTES];
  unsigned char* Qs = smem;
  unsigned char* Kb = smem + QS_BYTES;                 // KBLK key tiles, row-major images
  unsigned char* Vs = Kb + KBLK * KS_BYTES;
  unsigned char* Ps = Vs + VS_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int z = blockIdx.y, b = z / a.H, h = z - b * a.H;
  const int q0 = blockIdx.x * TQ;
  const int S = a.S, ld = 3 * a.d;
  const int len = a.lens ? min((int)a.lens[b], S) : S;     // (train-mode truncation: mel_len may exceed the S = 1000 rows kept)
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;
  const int ntile = (S + TK - 1) / TK;
  const int nblk = (ntile + KBLK - 1) / KBLK;
  unsigned char* Pw = Ps + wave * 16 * PS_RS;

  // scores of this wave's 16 queries against key tile `jt` of the resident block (already scaled, PAD keys -inf)
  auto scores = [&](int j, int jt, f32x4 (&s)[4]) __attribute__((always_inline)) {
    const unsigned char* Ks = Kb + jt * KS_BYTES;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 qa = frag_rows(Qs, wave * 16, ks, l15, lg);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, frag_rows(Ks, nt * 16, ks, l15, lg), s[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int key = j * TK + nt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[nt][r] = key < len ? s[nt][r] * a.scale : -INFINITY;
    }
  };
  // all tiles of key block kb: loads issued together (one memory latency per block, not per tile)
  auto stage_kblock = [&](int kb) __attribute__((always_inline)) {
    uint4 r[KBLK][4];
#pragma unroll
    for (int jt = 0; jt < KBLK; ++jt)
      if (kb * KBLK + jt < ntile) load_tile(r[jt], base + a.d, ld, (kb * KBLK + jt) * TK, S, tid);
#pragma unroll
    for (int jt = 0; jt < KBLK; ++jt)
      if (kb * KBLK + jt < ntile) store_rows(Kb + jt * KS_BYTES, r[jt], tid);
  };

  stage_rows(Qs, base, ld, q0, S, tid);
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }

  // ---- sweep 1: row maxima and sums
  for (int kb = 0; kb < nblk; ++kb) {
    if (kb > 0) __syncthreads();
    stage_kblock(kb);
    __syncthreads();
    for (int jt = 0; jt < KBLK && kb * KBLK + jt < ntile; ++jt) {
      f32x4 s[4];
      scores(kb * KBLK + jt, jt, s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float tm = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
        tm = quad16_max(tm);
        const float mn = fmaxf(m[r], tm);
        float ts = 0.f;
        if (mn > -INFINITY) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) ts += __expf(s[nt][r] - mn);
        }
        ts = quad16_sum(ts);
        l[r] = (m[r] > -INFINITY ? l[r] * __expf(m[r] - mn) : 0.f) + ts;
        m[r] = mn;
      }
    }
  }
  float inv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv[r] = 1.f / l[r];

  // ---- sweep 2: P = exp(S - m) / l  ->  bf16  ->  O += P V, V tiles one ahead in registers
  f32x4 oacc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 vr[4];
  load_tile(vr, base + 2 * a.d, ld, 0, S, tid);
  for (int kb = 0; kb < nblk; ++kb) {
    if (nblk > 1) {                          // more than 448 keys: the block has to be staged again
      __syncthreads();
      stage_kblock(kb);
    }
    for (int jt = 0; jt < KBLK && kb * KBLK + jt < ntile; ++jt) {
      const int j = kb * KBLK + jt;
      __syncthreads();                       // previous tile's V reads are done
      store_tr(Vs, vr, tid);
      if (j + 1 < ntile) load_tile(vr, base + 2 * a.d, ld, (j + 1) * TK, S, tid);
      __syncthreads();
      f32x4 s[4];
      scores(j, jt, s);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[nt][r] - m[r]) * inv[r];                  // exp(-inf) = 0 on PAD keys
          *(bf16_t*)(Pw + (lg * 4 + r) * PS_RS + (nt * 16 + l15) * 2) = f2bf(p);
        }
      // the P tile is private to the wave; the wait orders its LDS writes before its reads (and pins the compiler)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (a.probs) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cid = i * 64 + lane, row = cid >> 3, ch = cid & 7;
          const int q = q0 + wave * 16 + row, key = j * TK + ch * 8;
          if (q < S && key < a.Sp) *(uint4*)(a.probs + ((int64_t)z * S + q) * a.Sp + key) = *(const uint4*)(Pw + row * PS_RS + ch * 16);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pa = *(const bf16x8*)(Pw + l15 * PS_RS + (ks * 32 + lg * 8) * 2);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, frag_tr(Vs, nb, ks, l15, lg), oacc[nb], 0, 0, 0);
      }
    }
  }

  // ---- O tile -> LDS -> full-row stores (heads merged: column h*128); fp32 copy straight from the accumulators
  if (a.o32) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + wave * 16 + lg * 4 + r;
        if (q < S) a.o32[((int64_t)b * S + q) * a.d + h * DK + nb * 16 + l15] = oacc[nb][r];
      }
  }
  __syncthreads();
  unsigned char* Os = smem;   // 64 x 272 B = 17 KiB over Qs / the first key tile
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(bf16_t*)(Os + (wave * 16 + lg * 4 + r) * OS_RS + (nb * 16 + l15) * 2) = f2bf(oacc[nb][r]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (q0 + row < S) *(uint4*)(a.o + ((int64_t)b * S + q0 + row) * a.d + h * DK + ch * 8) = *(const uint4*)(Os + row * OS_RS + ch * 16);
  }
}

// Forward for S <= KBLK * 64 = 448 keys (every utterance of the training batch): ONE sweep.  The scores of a wave's 16
// queries against all keys stay in registers (7 tiles x 4 accumulators), so the row maxima and sums need no recompute, and
// once every wave is past the scores the key tiles in LDS are dead: V is staged into their place (4 + 3 tiles, one memory
// latency each instead of one per tile) and P V runs over resident tiles without a barrier per key tile.
__global__ __launch_bounds__(256, 1) void attn_fwd1_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[QS_BYTES + KBLK * KS_BYTES + PS_BYTES];
  unsigned char* Qs = smem;
  unsigned char* Kb = smem + QS_BYTES;                 // KBLK key tiles (row-major), later the value tiles (contraction-major)
  unsigned char* Ps = Kb + KBLK * KS_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int z = blockIdx.y, b = z / a.H, h = z - b * a.H;
  const int q0 = blockIdx.x * TQ;
  const int S = a.S, ld = 3 * a.d;
  const int len = a.lens ? min((int)a.lens[b], S) : S;     // (train-mode truncation: mel_len may exceed the S = 1000 rows kept)
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;
  const int ntile = (S + TK - 1) / TK;                 // <= KBLK
  unsigned char* Pw = Ps + wave * 16 * PS_RS;

  {  // Q and all key tiles: every load in flight before the first LDS store
    uint4 rq[4], rk[KBLK][4];
    load_tile(rq, base, ld, q0, S, tid);
#pragma unroll
    for (int jt = 0; jt < KBLK; ++jt)
      if (jt < ntile) load_tile(rk[jt], base + a.d, ld, jt * TK, S, tid);
    store_rows(Qs, rq, tid);
#pragma unroll
    for (int jt = 0; jt < KBLK; ++jt)
      if (jt < ntile) store_rows(Kb + jt * KS_BYTES, rk[jt], tid);
  }
  __syncthreads();

  // ---- scores of the wave's 16 queries against every key (scaled; PAD keys and absent tiles -inf)
  f32x4 sc[KBLK][4];
  bf16x8 qa[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qa[ks] = frag_rows(Qs, wave * 16, ks, l15, lg);
#pragma unroll
  for (int jt = 0; jt < KBLK; ++jt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) sc[jt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (jt < ntile) {
      const unsigned char* Ks = Kb + jt * KS_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          sc[jt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], frag_rows(Ks, nt * 16, ks, l15, lg), sc[jt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int key = jt * TK + nt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[jt][nt][r] = key < len ? sc[jt][nt][r] * a.scale : -INFINITY;
    }
  }
  // value tiles 0..3 are requested now; their latency runs under the softmax arithmetic
  uint4 vr[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    if (jt < ntile) load_tile(vr[jt], base + 2 * a.d, ld, jt * TK, S, tid);

  float m[4], inv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float tm = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < KBLK; ++jt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) tm = fmaxf(tm, sc[jt][nt][r]);
    tm = quad16_max(tm);
    float ts = 0.f;
    if (tm > -INFINITY) {
#pragma unroll
      for (int jt = 0; jt < KBLK; ++jt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float e = __expf(sc[jt][nt][r] - tm);      // exp(-inf) = 0 on PAD keys
          sc[jt][nt][r] = e;
          ts += e;
        }
    } else {
#pragma unroll
      for (int jt = 0; jt < KBLK; ++jt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) sc[jt][nt][r] = 0.f;
    }
    ts = quad16_sum(ts);
    m[r] = tm;
    inv[r] = 1.f / ts;
  }

  f32x4 oacc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // P tile j (normalised, rounded to bf16: the same rounding point as the unfused path) -> LDS -> probs, P V
  auto pv_tile = [&](int j, const f32x4 (&e)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) *(bf16_t*)(Pw + (lg * 4 + r) * PS_RS + (nt * 16 + l15) * 2) = f2bf(e[nt][r] * inv[r]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the P tile is private to the wave
    if (a.probs) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cid = i * 64 + lane, row = cid >> 3, ch = cid & 7;
        const int q = q0 + wave * 16 + row, key = j * TK + ch * 8;
        if (q < S && key < a.Sp) *(uint4*)(a.probs + ((int64_t)z * S + q) * a.Sp + key) = *(const uint4*)(Pw + row * PS_RS + ch * 16);
      }
    }
    const unsigned char* Vs = Kb + j * VS_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 pa = *(const bf16x8*)(Pw + l15 * PS_RS + (ks * 32 + lg * 8) * 2);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) oacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, frag_tr(Vs, nb, ks, l15, lg), oacc[nb], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // P reads done before the next tile overwrites Pw
  };

  __syncthreads();                                   // every wave is past its score MFMAs: the key tiles are dead
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    if (jt < ntile) store_tr(Kb + jt * VS_BYTES, vr[jt], tid);
#pragma unroll
  for (int jt = 4; jt < KBLK; ++jt)
    if (jt < ntile) load_tile(vr[jt - 4], base + 2 * a.d, ld, jt * TK, S, tid);
  __syncthreads();
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    if (jt < ntile) pv_tile(jt, sc[jt]);
#pragma unroll
  for (int jt = 4; jt < KBLK; ++jt)
    if (jt < ntile) store_tr(Kb + jt * VS_BYTES, vr[jt - 4], tid);     // slots 4..6: nobody reads them before this barrier
  __syncthreads();
#pragma unroll
  for (int jt = 4; jt < KBLK; ++jt)
    if (jt < ntile) pv_tile(jt, sc[jt]);

  // ---- O tile -> LDS -> full-row stores (heads merged: column h*128); fp32 copy straight from the accumulators
  if (a.o32) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + wave * 16 + lg * 4 + r;
        if (q < S) a.o32[((int64_t)b * S + q) * a.d + h * DK + nb * 16 + l15] = oacc[nb][r];
      }
  }
  __syncthreads();
  unsigned char* Os = smem;   // 64 x 272 B = 17 KiB over Qs / the first tile
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(bf16_t*)(Os + (wave * 16 + lg * 4 + r) * OS_RS + (nb * 16 + l15) * 2) = f2bf(oacc[nb][r]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (q0 + row < S) *(uint4*)(a.o + ((int64_t)b * S + q0 + row) * a.d + h * DK + ch * 8) = *(const uint4*)(Os + row * OS_RS + ch * 16);
  }
}

struct AttnBwdArgs {
  const bf16_t* qkv;     // forward input
  const float* o32;      // forward output, fp32 (sum_k P V before rounding)
  const bf16_t* dout;    // dO [B*S][d]
  const bf16_t* probs;   // P [B*H][S][Sp]
  bf16_t* ds;            // dS [B*H][S][Sp] out
  bf16_t* dq;            // dqkv base: dQ written at columns h*128 of a [B*S][3*d] buffer
  int S, Sp, H, d;
  float scale;
};

__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const AttnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[QS_BYTES + KS_BYTES + VS_BYTES + 2 * PS_BYTES];
  unsigned char* Ds = smem;                 // dO tile (row-major, like Q)
  unsigned char* Vs = smem + QS_BYTES;      // V tile row-major [key][d]   (B operand of dP = dO V^T)
  unsigned char* Kt = Vs + KS_BYTES;        // K tile contraction-major [key k][d n] (B operand of dQ = dS K)
  unsigned char* Ps = Kt + VS_BYTES;        // per-wave P tile
  unsigned char* Ss = Ps + PS_BYTES;        // per-wave dS tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int z = blockIdx.y, b = z / a.H, h = z - b * a.H;
  const int q0 = blockIdx.x * TQ;
  const int S = a.S, ld = 3 * a.d;
  const bf16_t* base = a.qkv + (int64_t)b * S * ld + h * DK;
  const bf16_t* dob = a.dout + (int64_t)b * S * a.d + h * DK;
  const float* ob = a.o32 + (int64_t)b * S * a.d + h * DK;
  const int ntile = (S + TK - 1) / TK;
  unsigned char* Pw = Ps + wave * 16 * PS_RS;
  unsigned char* Sw = Ss + wave * 16 * PS_RS;

  // tile j = {V tile, K tile, this wave's 16 x 64 slice of P}, requested one tile ahead
  uint4 vr[4], kr[4], pr[2];
  auto load_j = [&](int j) __attribute__((always_inline)) {
    load_tile(vr, base + 2 * a.d, ld, j * TK, S, tid);
    load_tile(kr, base + a.d, ld, j * TK, S, tid);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cid = i * 64 + lane, row = cid >> 3, ch = cid & 7;
      const int q = q0 + wave * 16 + row, key = j * TK + ch * 8;
      pr[i] = make_uint4(0, 0, 0, 0);
      if (q < S && key < a.Sp) pr[i] = *(const uint4*)(a.probs + ((int64_t)z * S + q) * a.Sp + key);
    }
  };
  load_j(0);
  stage_rows(Ds, dob, a.d, q0, S, tid);
  // delta[q] = sum_k P dP = sum_d dO[q][d] * (sum_k P V)[q][d]: rows lg*4 + r of the wave, 16 lanes x 8 columns each
  float delta[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + wave * 16 + lg * 4 + r;
    float acc = 0.f;
    if (q < S) {
      const uint4 x = *(const uint4*)(dob + (int64_t)q * a.d + l15 * 8);
      const f32x4 y0 = *(const f32x4*)(ob + (int64_t)q * a.d + l15 * 8), y1 = *(const f32x4*)(ob + (int64_t)q * a.d + l15 * 8 + 4);
      acc = __uint_as_float(x.x << 16) * y0[0] + __uint_as_float(x.x & 0xFFFF0000u) * y0[1] + __uint_as_float(x.y << 16) * y0[2] +
            __uint_as_float(x.y & 0xFFFF0000u) * y0[3] + __uint_as_float(x.z << 16) * y1[0] + __uint_as_float(x.z & 0xFFFF0000u) * y1[1] +
            __uint_as_float(x.w << 16) * y1[2] + __uint_as_float(x.w & 0xFFFF0000u) * y1[3];
    }
    delta[r] = quad16_sum(acc);
  }
  f32x4 dq[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) dq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int j = 0; j < ntile; ++j) {
    __syncthreads();                              // previous tile's LDS reads are done
    store_rows(Vs, vr, tid);
    store_tr(Kt, kr, tid);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cid = i * 64 + lane, row = cid >> 3, ch = cid & 7;
      *(uint4*)(Pw + row * PS_RS + ch * 16) = pr[i];
    }
    if (j + 1 < ntile) load_j(j + 1);
    __syncthreads();
    f32x4 dp[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dp[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 da = frag_rows(Ds, wave * 16, ks, l15, lg);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, frag_rows(Vs, nt * 16, ks, l15, lg), dp[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j * TK + nt * 16 + l15;
        const float p = bf2f(*(const bf16_t*)(Pw + (lg * 4 + r) * PS_RS + (nt * 16 + l15) * 2));
        const float v = key < S ? a.scale * p * (dp[nt][r] - delta[r]) : 0.f;
        *(bf16_t*)(Sw + (lg * 4 + r) * PS_RS + (nt * 16 + l15) * 2) = f2bf(v);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cid = i * 64 + lane, row = cid >> 3, ch = cid & 7;
      const int q = q0 + wave * 16 + row, key = j * TK + ch * 8;
      if (q < S && key < a.Sp) *(uint4*)(a.ds + ((int64_t)z * S + q) * a.Sp + key) = *(const uint4*)(Sw + row * PS_RS + ch * 16);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 sa = *(const bf16x8*)(Sw + l15 * PS_RS + (ks * 32 + lg * 8) * 2);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) dq[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, frag_tr(Kt, nb, ks, l15, lg), dq[nb], 0, 0, 0);
    }
  }
  __syncthreads();
  unsigned char* Os = smem;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(bf16_t*)(Os + (wave * 16 + lg * 4 + r) * OS_RS + (nb * 16 + l15) * 2) = f2bf(dq[nb][r]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid, row = c >> 4, ch = c & 15;
    if (q0 + row < S) *(uint4*)(a.dq + ((int64_t)b * S + q0 + row) * ld + h * DK + ch * 8) = *(const uint4*)(Os + row * OS_RS + ch * 16);
  }
}

}  // namespace

extern "C" int ttsk_attention_fwd(const void* qkv_bf16, void* o_bf16, float* o_f32, void* probs_bf16, const int64_t* lens, int B, int H,
                                  int S, int Sp, int d, float scale, void* stream) {
  TTSK_REQUIRE(qkv_bf16 && o_bf16, "attention_fwd: null pointer");
  TTSK_REQUIRE(B > 0 && H > 0 && S > 0 && d == H * DK, "attention_fwd: head size must be 128 (d = %d, H = %d)", d, H);
  TTSK_REQUIRE(!probs_bf16 || (Sp >= S && (Sp & 7) == 0), "attention_fwd: Sp must be a multiple of 8 >= S");
  TTSK_REQUIRE(B * H <= 65535, "attention_fwd: too many (utterance, head) pairs");
  AttnArgs a{(const bf16_t*)qkv_bf16, (bf16_t*)o_bf16, o_f32, (bf16_t*)probs_bf16, (const long long*)lens, S, Sp, H, d, scale};
  if ((S + TK - 1) / TK <= KBLK)
    hipLaunchKernelGGL(attn_fwd1_kernel, dim3((S + TQ - 1) / TQ, B * H), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((S + TQ - 1) / TQ, B * H), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_attention_bwd_q(const void* qkv_bf16, const float* o_f32, const void* dout_bf16, const void* probs_bf16,
                                    void* ds_bf16, void* dqkv_bf16, int B, int H, int S, int Sp, int d, float scale, void* stream) {
  TTSK_REQUIRE(qkv_bf16 && o_f32 && dout_bf16 && probs_bf16 && ds_bf16 && dqkv_bf16, "attention_bwd_q: null pointer");
  TTSK_REQUIRE(B > 0 && H > 0 && S > 0 && d == H * DK, "attention_bwd_q: head size must be 128");
  TTSK_REQUIRE(Sp >= S && (Sp & 7) == 0 && B * H <= 65535, "attention_bwd_q: bad sizes");
  AttnBwdArgs a{(const bf16_t*)qkv_bf16, o_f32, (const bf16_t*)dout_bf16, (const bf16_t*)probs_bf16, (bf16_t*)ds_bf16,
                (bf16_t*)dqkv_bf16, S, Sp, H, d, scale};
  hipLaunchKernelGGL(attn_bwd_q_kernel, dim3((S + TQ - 1) / TQ, B * H), dim3(256), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}
