// attn_common.h — tile staging / fragment helpers shared by the attention kernels (attn.hip, flash_attn.hip), d_k = 128.
#pragma once
#include "common.h"
#include "tapring.h"

namespace {

constexpr int DK = 128;
constexpr int TQ = 64, TK = 64;
constexpr int QS_BYTES = TQ * DK * 2;      // 16 KiB: two [64][64] row-major sub-tiles (k chunks of 64), chunk ^= row & 7
constexpr int KS_BYTES = TK * DK * 2;      // same layout, rows = keys
constexpr int VS_BYTES = TK * DK * 2;      // [64 k][128 n] contraction-major tile (ds_read_b64_tr_b16 layout)
constexpr int PS_RS = 144;                 // per-wave P / dS tile [16][64] bf16, padded rows
constexpr int PS_BYTES = 4 * 16 * PS_RS;   // 9 KiB
constexpr int OS_RS = 272;                 // output staging [64][128] bf16, padded rows

__device__ __forceinline__ int tr_sw(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// A 64 x 128 bf16 tile travels global -> 4 x uint4 per thread -> LDS; loads and stores are separate so that the next
// tile's loads are in flight while the current tile is multiplied.
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16_t* __restrict__ src, int ld, int row0, int nrows_valid, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 4, kc = c & 15;
    r[i] = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows_valid) r[i] = *(const uint4*)(src + (int64_t)(row0 + row) * ld + kc * 8);
  }
}
// row-major image: two swizzled [64][64] sub-tiles (k chunks of 64), 16-byte chunk ^= row & 7
__device__ __forceinline__ void store_rows(unsigned char* dst, const uint4 (&r)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int row = c >> 4, kc = c & 15;
    *(uint4*)(dst + (kc >> 3) * 8192 + row * 128 + (((kc & 7) ^ (row & 7)) << 4)) = r[i];
  }
}
// contraction-major image [64 k][128 n] for ds_read_b64_tr_b16
__device__ __forceinline__ void store_tr(unsigned char* dst, const uint4 (&r)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int kr = c >> 4, ts = c & 15;
    *(uint4*)(dst + kr * 256 + (((ts >> 1) ^ tr_sw(kr)) << 5) + ((ts & 1) << 4)) = r[i];
  }
}
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* __restrict__ src, int ld, int row0, int nrows_valid,
                                           int tid) {
  uint4 r[4];
  load_tile(r, src, ld, row0, nrows_valid, tid);
  store_rows(dst, r, tid);
}
// fragment of a row-major tile: rows r0 + l15, k step (sub-tile ks >> 1, half ks & 1)
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* base, int r0, int ks, int l15, int lg) {
  const int row = r0 + l15;
  return *(const bf16x8*)(base + (ks >> 1) * 8192 + row * 128 + ((((ks & 1) * 4 + lg) ^ (row & 7)) << 4));
}
// fragment of a contraction-major tile: n tile nblk, k step ks (32 k rows)
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* base, int nblk, int ks, int l15, int lg) {
  const int k0 = ks * 32 + 8 * lg + (l15 >> 2), k1 = k0 + 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(base + k0 * 256 + ((nblk ^ tr_sw(k0)) << 5) + ((l15 & 3) << 3)));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(base + k1 * 256 + ((nblk ^ tr_sw(k1)) << 5) + ((l15 & 3) << 3)));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

}  // namespace
