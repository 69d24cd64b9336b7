// gemm_ln.hip — the tail of both FFT-block sub-layers as ONE kernel:
//     out = zero_PAD_rows( LayerNorm( dropout(A @ W^T + bias) + residual ) )
// reference: fs_two/transformer/SubLayers.py:62-63 (MHA: layer_norm(dropout(fc(o)) + residual)), :96-99 (FFN:
// layer_norm(dropout(w_2(h)) + residual), w_2 = Conv1d(k=1) = a Linear over channels), Layers.py:29,32 (masked_fill).
//
// Why a kernel of its own: the output has D = 256 columns, i.e. ONE 256-wide tile row holds whole LayerNorm rows, so
// the normalisation can run in the epilogue.  Before, each of these was GEMM (106 workgroups of 128x128 for 6768 rows, or
// split-K + a reducer launch for the encoder's 1024 rows) -> bf16 y -> a LayerNorm launch: 2-3 dependent launches of
// 5-25 us for 0.9 / 3.5 GFLOP.  Here a workgroup owns 32 rows x 256 columns (212 workgroups for 6768 rows, 32 for 1024):
// 4 waves, wave w computes columns [64w, 64w+64) of all 32 rows as 2x4 v_mfma_f32_16x16x32_bf16 accumulators.
// The kernel is bound by streaming W (256 x K bf16, L2-resident, 32 KiB per K step per workgroup) through LDS, not by MFMA:
// LDS-DMA into a 4-stage ring (see the comment at the kernel), XOR-swizzled 128-byte rows (conflict-free fragment reads),
// one barrier per K step.  Epilogue: accumulators -> fp32 LDS tile -> one wave per row (4 columns per
// lane, exactly ln_fwd_kernel's row code: same Philox indexing, so ln_bwd_kernel regenerates the same dropout mask).
#include "gemm_common.h"
#include "proj32.h"
#include "tapring.h"

namespace {

constexpr int BM = 32, BN = 256, BK = 64, NSTAGE = 4;
constexpr int A_BYTES = BM * BK * 2;              // 4 KiB  = 4 LDS-DMA pieces (one per wave)
constexpr int B_BYTES = BN * BK * 2;              // 32 KiB = 32 pieces (eight per wave)
constexpr int STAGE = A_BYTES + B_BYTES;          // 36 KiB
constexpr int CS_LD = 260;                        // fp32 epilogue tile leading dimension
constexpr int SMEM = NSTAGE * STAGE;              // 144 KiB ring (the epilogue tile 32 x 260 x 4 = 33,280 B reuses it)
constexpr int OOB = 0x7FFFFFFF;
constexpr int WL_SPLIT_MAX_TILES = 64;            // win_ln_kernel<1024, PROJ>: up to this many 32-row tiles, a tile's three projection groups go to three workgroups

struct GemmLnArgs {
  const bf16_t* A;        // [M][lda] bf16
  const bf16_t* W;        // [256][ldw] bf16 (nn.Linear / Conv1d(k=1) weight: out x in)
  const float* bias;      // [256]
  const bf16_t* res;      // [M][256] residual
  const float* gamma;
  const float* beta;
  bf16_t* out;            // [M][256]
  bf16_t* z_save;         // [M][256] LayerNorm input (after dropout + residual), or null
  float* mean;            // [M]
  float* rstd;            // [M]
  const long long* lens;  // [M / seg_len] valid rows per segment, or null
  const uint64_t* rng;    // {seed, step}
  int M, K, lda, ldw, seg_len;
  float p_pre, eps;
  unsigned site_pre;
  // win_ln_kernel<CIN, true>: the block's output rows go straight into the NEXT block's q|k|v projection (proj32.h) as well:
  // pout[M][pCout] = out · pw' + pbias, pw = ttsk_win_conv's pack of that (pCout = 768, 1, 256) weight
  const bf16_t* pw;
  const float* pbias;
  bf16_t* pout;
  int pCout;
#ifdef TTSK_STAMPS
  unsigned long long* stamps;   // diagnostic build only (make stamps; ttsk_win_ln_set_stamps): 8 x s_memrealtime per workgroup
#endif
};
#ifdef TTSK_STAMPS
#define WL_STAMP(i)                                                                                              \
  do {                                                                                                           \
    if (a.stamps && threadIdx.x == 0) a.stamps[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define WL_STAMP(i) do {} while (0)        // the product library carries no stamp code and no global state for it
#endif

__device__ __forceinline__ void drop4(float v[4], uint64_t seed, uint64_t step, unsigned site, unsigned e4, unsigned thr, float scale) {
  const uint4 b = Philox::gen(make_uint2((unsigned)seed, (unsigned)(seed >> 32)),
                              make_uint4(e4, site, (unsigned)step, (unsigned)(step >> 32)));
  v[0] = b.x >= thr ? v[0] * scale : 0.f;
  v[1] = b.y >= thr ? v[1] * scale : 0.f;
  v[2] = b.z >= thr ? v[2] * scale : 0.f;
  v[3] = b.w >= thr ? v[3] * scale : 0.f;
}

// The LayerNorm rows of a 32 x 256 fp32 tile `cs` (leading dimension CS_LD): one wave per row, RW rows per wave, lane owns columns
// 4*lane .. 4*lane+3 (ln_fwd_kernel's row code: same Philox indexing, so ln_bwd_kernel regenerates the same dropout mask).
// xs2 (may be null): LDS tile [32][P32_RS] that receives the output rows as well (zero rows past M), proj32.h's B operand.
template <int RW>
__device__ __forceinline__ void ln_rows_epilogue(const GemmLnArgs& a, const float* cs, int m0, int wave, int lane, const uint2 (&resv)[RW],
                                                 unsigned char* xs2 = nullptr, bool write = true /* false: the rows go to xs2 only (win_ln_kernel's SPLIT parts > 0) */) {
  const int M = a.M, c = lane * 4;
  const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
  const f32x4 bi = *(const f32x4*)(a.bias + c);
  const f32x4 g = *(const f32x4*)(a.gamma + c), bt = *(const f32x4*)(a.beta + c);
  const unsigned thr = keep_threshold(a.p_pre);
  const float scale = 1.f / (1.f - a.p_pre);
  float z[RW][4];
  float s[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int lr = wave * RW + rr;
    const int row = m0 + lr;
    const bool live = row < M;
    const f32x4 v = *(const f32x4*)(cs + lr * CS_LD + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) z[rr][e] = v[e] + bi[e];
    if (a.p_pre > 0.f) drop4(z[rr], seed, step, a.site_pre, (unsigned)(((int64_t)row * BN + c) >> 2), thr, scale);
    z[rr][0] += __uint_as_float(resv[rr].x << 16); z[rr][1] += __uint_as_float(resv[rr].x & 0xFFFF0000u);
    z[rr][2] += __uint_as_float(resv[rr].y << 16); z[rr][3] += __uint_as_float(resv[rr].y & 0xFFFF0000u);
    if (a.z_save && live && write) *(uint2*)(a.z_save + (int64_t)row * BN + c) = make_uint2(pack_bf2(z[rr][0], z[rr][1]), pack_bf2(z[rr][2], z[rr][3]));
    s[rr] = z[rr][0] + z[rr][1] + z[rr][2] + z[rr][3];
  }
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) s[rr] = wave_sum(s[rr]);       // independent chains: DPP inside the 16-lane rows, two permutes across
  float q[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    s[rr] *= (1.f / BN);            // mean
    q[rr] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = z[rr][e] - s[rr]; q[rr] += d * d; }
  }
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) q[rr] = wave_sum(q[rr]);
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = m0 + wave * RW + rr;
    if (row >= M) {
      if (xs2) *(uint2*)(xs2 + (wave * RW + rr) * P32_RS + c * 2) = make_uint2(0u, 0u);
      continue;
    }
    const float mean = s[rr], rstd = rsqrtf(q[rr] * (1.f / BN) + a.eps);
    if (lane == 0 && write) { a.mean[row] = mean; a.rstd[row] = rstd; }
    bool masked = false;
    if (a.lens) { const int b = row / a.seg_len, t = row - b * a.seg_len; masked = t >= a.lens[b]; }
    float o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = masked ? 0.f : (z[rr][e] - mean) * rstd * g[e] + bt[e];
    const uint2 ov = make_uint2(pack_bf2(o4[0], o4[1]), pack_bf2(o4[2], o4[3]));
    if (write) *(uint2*)(a.out + (int64_t)row * BN + c) = ov;
    if (xs2) *(uint2*)(xs2 + (wave * RW + rr) * P32_RS + c * 2) = ov;
  }
}

// The kernel streams W (256 x K, the same for every workgroup: L2-resident) and its 32 rows of A through LDS; one CU pulls
// about 70-90 GB/s from L2 into LDS, and only with ~100 KB of requests in flight (measured with the register-staged first
// version: two 36 KiB tiles in flight gave 1.7 us per K step = 21 GB/s).  So: LDS-DMA (buffer_load ... lds, no VGPR round
// trip, swizzle applied to the SOURCE address) into a 4-stage ring with three tiles in flight, counted vmcnt, one raw
// s_barrier per K step (the structure of gemm2.hip).  The residual rows are fetched into registers before the K loop.
// NW waves: wave w owns columns [256/NW * w, +256/NW) of all 32 rows and 32/NW of the workgroup's B pieces per K tile (waves 0-3 also
// one A piece).  STAGGER: workgroup i starts its K walk at tile (i * 5) % nk and wraps — the workgroups all stream the SAME W, and
// in lockstep they would all ask L2 for the same 32 KiB at the same time.
template <int NW, bool STAGGER>
__global__ __launch_bounds__(NW * 64, 1) void gemm_ln_kernel(const GemmLnArgs a) {
  constexpr int WC = BN / NW;            // columns per wave: 64 / 32
  constexpr int NJ = WC / 16;            // MFMA column tiles per wave: 4 / 2
  constexpr int NBP = 32 / NW;           // B pieces per wave per K tile: 8 / 4
  constexpr int RW = BM / NW;            // epilogue rows per wave: 8 / 4
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int M = a.M, K = a.K;
  const int K8 = (K + 7) & ~7;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 0x7FFFFFF0, 0x00020000);
  const int nk = (K + BK - 1) / BK;
  const int c = lane * 4;

  // ---- residual rows of this wave's eight epilogue rows: in flight during the whole K loop
  uint2 resv[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = m0 + wave * RW + rr;
    resv[rr] = (a.res && row < M) ? *(const uint2*)(a.res + (int64_t)row * BN + c) : make_uint2(0u, 0u);
  }

  // ---- LDS-DMA source coordinates.  A piece = 1 KiB of the LDS image = 8 tile rows x 128 B; lane -> tile row 8p + lane/8,
  // physical 16-byte chunk lane%8, which must hold logical chunk (lane%8) ^ (row & 7): the XOR swizzle of the fragment reads
  const int prow = lane >> 3, kch = ((lane & 7) ^ (lane >> 3)) * 8;     // row inside the piece, first k of this lane's 16 bytes
  const int a_row = m0 + (wave & 3) * 8 + prow;
  const int a_off = a_row < M ? (a_row * a.lda + kch) * 2 : OOB;
  int w_off[NBP];
#pragma unroll
  for (int i = 0; i < NBP; ++i) w_off[i] = (((i * NW + wave) * 8 + prow) * a.ldw + kch) * 2;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_ptr)smem + wave * 1024);
  const bool has_a = wave < 4;           // wave-uniform
  const int kshift = STAGGER ? (int)((blockIdx.x * 5u) % (unsigned)nk) : 0;

  // tile kt of THIS workgroup's walk is K tile (kt + kshift) % nk; it lives in ring stage kt % NSTAGE
  auto issue_tile = [&](int kt) __attribute__((always_inline)) {
    const unsigned st = lds0 + (kt % NSTAGE) * STAGE;
    int kk = kt + kshift;
    if (kk >= nk) kk -= nk;
    const int kbase = kk * BK;
    const bool kok = kbase + kch < K8;
    if (has_a) dma16(rsA, st, (kok && a_off != OOB) ? a_off + kbase * 2 : OOB);
#pragma unroll
    for (int i = 0; i < NBP; ++i) dma16(rsW, st + A_BYTES + i * (NW * 1024), kok ? w_off[i] + kbase * 2 : OOB);
  };

  f32x4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nk) issue_tile(t);

  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of tile kt have landed when at most the pieces of the (up to two) younger tiles are outstanding
    const int younger = nk - 1 - kt;
    if (younger >= 2) {
      if (has_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NBP + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBP) : "memory");
    } else if (younger == 1) {
      if (has_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // everybody's pieces of tile kt landed; everybody is done reading tile kt-1's stage
    if (kt + NSTAGE - 1 < nk) issue_tile(kt + NSTAGE - 1);      // refills the stage tile kt-1 occupied
    const unsigned char* sa = smem + (kt % NSTAGE) * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 16 + l15;
        af[i] = *(const bf16x8*)(sa + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wave * WC + j * 16 + l15;
        bfr[j] = *(const bf16x8*)(sb + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                         // the last stage has been read by every wave: the ring becomes the epilogue tile

  // ---- epilogue: accumulators -> fp32 tile in LDS
  float* cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(i * 16 + lg * 4 + r) * CS_LD + wave * WC + j * 16 + l15] = acc[i][j][r];
  __syncthreads();

  ln_rows_epilogue<RW>(a, cs, m0, wave, lane, resv);
}


// ---- the same fused op with the window-conv data path (csrc/ffn_conv.hip): the 32 rows of A (all K channels, K = 256 or 1024) go into
// LDS once, each of the eight waves owns 32 of the 256 output columns and streams its weights L2 -> registers from the MFMA-fragment-
// major pack (ttsk_win_conv_pack_*: [k-step][cout tile][lane][8]), one step = 128 input channels = 8 fragments, three register sets, no
// barrier in the loop; then the fp32 tile and ln_rows_epilogue as above.  W still crosses L2 -> CU once per workgroup (128 / 512 KiB), but
// as 1 KiB contiguous fragments into registers instead of through the LDS ring with a barrier per 64 channels.
// SPLIT (round 6; the phoneme side: 32 tiles on 256 CUs): the three 256-channel groups of the projection behind the LayerNorm go to SPLIT workgroups per
// tile, each repeating the tile's contraction and LayerNorm and streaming ITS group of the projection's weights; part 0 alone writes the rows, the statistics
// and the saved LayerNorm input.  No seam, bit-identical.
template <int CIN, bool PROJ, int SPLIT = 1>
__global__ __launch_bounds__(512, (CIN == 256 && !PROJ) ? 2 : 1) void win_ln_kernel(const GemmLnArgs a) {
  constexpr int TT = BM, RS = CIN * 2 + 32, NT = 512, CH8 = CIN / 8, KH = 4, CT = 2, NF = TT / 16, NS = CIN / 128, RW = BM / 8;
  constexpr int XBYTES = TT * RS, CBYTES = BM * CS_LD * 4;
  static_assert(BM == P32_TT && BN == P32_D && NT == P32_NT, "proj32.h tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem[XBYTES > CBYTES ? XBYTES : CBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ptile[PROJ ? 2 * P32_TT * P32_RS : 16];      // output rows (B operand) | staging rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int part = SPLIT > 1 ? (int)blockIdx.x % SPLIT : 0;
  const int m0 = (SPLIT > 1 ? (int)blockIdx.x / SPLIT : (int)blockIdx.x) * BM;
  const int M = a.M;
  WL_STAMP(0);
  // weight fragments by buffer loads, activation fragments a step ahead of their MFMAs (tapring.h)
  const __amdgpu_buffer_rsrc_t wres = weights_rsrc(a.W, CIN * BN * 2);
  const int wlane = (wave * CT) * 1024 + lane * 16;
  constexpr int kstep_bytes = (BN / 16) * 1024;                // bytes per k-step of the pack
  bf16x8 wa[KH][CT], wb[KH][CT], wc[KH][CT];
  auto load_w = [&](int g, bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) { frags_load<KH, CT>(w, wres, wlane, g * KH * kstep_bytes, kstep_bytes); };
  uint2 resv[RW];
  {
    // request order = arrival order: the rows of A (nothing starts before they are in LDS), the first three steps' weight fragments,
    // the residual rows (read in the epilogue)
    constexpr int NCH = (TT * CH8 + NT - 1) / NT;
    uint4 xv[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (idx < TT * CH8 && m0 + row < M) xv[it] = *(const uint4*)(a.A + (int64_t)(m0 + row) * a.lda + ch * 8);
    }
    load_w(0, wa);
    if (1 < NS) load_w(1, wb);
    if (2 < NS) load_w(2, wc);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int row = m0 + wave * RW + rr;
      resv[rr] = (a.res && row < M) ? *(const uint2*)(a.res + (int64_t)row * BN + lane * 4) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int idx = it * NT + tid;
      const int row = idx / CH8, ch = idx - row * CH8;
      if (idx < TT * CH8) *(uint4*)(smem + row * RS + ch * 16) = xv[it];
    }
  }
  WL_STAMP(1);
  __syncthreads();
  WL_STAMP(2);
  f32x4 acc[CT][NF];
#pragma unroll
  for (int cc = 0; cc < CT; ++cc)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[cc][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int inl = l15 * RS + q * 16;
    bf16x8 ring[KH][NF];
    ring_prime_step<NF, RS, KH>(ring, smem, inl);
    auto step = [&](int g, const bf16x8 (&w)[KH][CT]) __attribute__((always_inline)) {
      tap_ring_step<false, KH, CT, NF, RS>(acc, ring, w, smem, inl + (g + 1 < NS ? g + 1 : g) * (KH * 64));
    };
    // NS = 2 or 8 steps, unrolled: every `if` below is decided at compile time, so the wait counts of the weight requests are exact
    // (the scheduling barriers keep each request where it is written: right behind the step that frees its registers, two steps ahead of its use)
#pragma unroll
    for (int g = 0; g < NS; g += 3) {
      step(g, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 3 < NS) load_w(g + 3, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < NS) {
        step(g + 1, wb);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 4 < NS) load_w(g + 4, wb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (g + 2 < NS) {
        step(g + 2, wc);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 5 < NS) load_w(g + 5, wc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  WL_STAMP(3);
  Proj32W PW;
  if (PROJ) proj32_prefetch(a.pw, a.pCout, wave, lane, PW, part * (3 / SPLIT));      // the next projection's first fragments arrive behind the LayerNorm rows
  __syncthreads();                         // every wave is done with the rows of A: they become the fp32 tile
  WL_STAMP(4);
  float* cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) *(f32x4*)(cs + (i * 16 + l15) * CS_LD + (wave * CT + cc) * 16 + q * 4) = acc[cc][i];
  __syncthreads();
  ln_rows_epilogue<RW>(a, cs, m0, wave, lane, resv, PROJ ? ptile : nullptr, part == 0);
  WL_STAMP(5);
  if (PROJ) {
    __syncthreads();
    proj32_run<3 / SPLIT>(ptile, ptile + P32_TT * P32_RS, a.pw, a.pCout, a.pbias, PW, tid, [&](int cg, int rr, int ch, uint4 v, int) __attribute__((always_inline)) {
      if (m0 + rr < M) *(uint4*)(a.pout + (int64_t)(m0 + rr) * a.pCout + cg * BN + ch * 8) = v;
    }, Proj32NoPre(), part * (3 / SPLIT));
  }
  WL_STAMP(6);
}

}  // namespace

#ifdef TTSK_STAMPS
static unsigned long long* g_wl_stamps = nullptr;
// diagnostic build only (`make stamps`, tools/debug/wl_stamps.py; not declared in ttsk.h, not in the product library)
extern "C" int ttsk_win_ln_set_stamps(void* dev_buffer) {
  g_wl_stamps = (unsigned long long*)dev_buffer;
  return TTSK_OK;
}
#endif

extern "C" int ttsk_gemm_ln_fwd(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res,
                                const float* gamma, const float* beta, void* out, void* z_save, float* mean, float* rstd,
                                const int64_t* lens, int seg_len, int M, int K, int D, float eps, float p_pre, uint32_t site_pre,
                                const void* rng, void* stream) {
  TTSK_REQUIRE(A && W && bias && gamma && beta && out && mean && rstd, "ttsk_gemm_ln_fwd: null pointer");
  TTSK_REQUIRE(D == BN, "ttsk_gemm_ln_fwd: the fused kernel is built for D = 256 (got %d); use ttsk_gemm + ttsk_layernorm_fwd", D);
  TTSK_REQUIRE(M > 0 && K > 0 && (K & 7) == 0 && (lda & 7) == 0 && (ldw & 7) == 0, "ttsk_gemm_ln_fwd: K, lda, ldw must be multiples of 8");
  TTSK_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)res | (uintptr_t)z_save) & 15) == 0, "ttsk_gemm_ln_fwd: operands must be 16-byte aligned");
  TTSK_REQUIRE(((int64_t)M * lda + K) * 2 < ((int64_t)1 << 31) && ((int64_t)BN * ldw + K) * 2 < ((int64_t)1 << 31), "ttsk_gemm_ln_fwd: operand extent exceeds 2 GiB");
  TTSK_REQUIRE(!(p_pre > 0.f) || rng, "ttsk_gemm_ln_fwd: dropout needs the rng state");
  TTSK_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "ttsk_gemm_ln_fwd: p_pre must be in [0, 1)");
  TTSK_REQUIRE(!lens || (seg_len > 0 && M % seg_len == 0), "ttsk_gemm_ln_fwd: lens needs M %% seg_len == 0");
  GemmLnArgs a{(const bf16_t*)A, (const bf16_t*)W, bias, (const bf16_t*)res, gamma, beta, (bf16_t*)out, (bf16_t*)z_save, mean, rstd,
               (const long long*)lens, (const uint64_t*)rng, M, K, lda, ldw, seg_len > 0 ? seg_len : 1, p_pre, eps, site_pre};
  // 8 waves, staggered weight stream: the measured optimum of (4 | 8 | 16 waves) x (plain | staggered) on the step's shapes
  hipLaunchKernelGGL((gemm_ln_kernel<8, true>), dim3((M + BM - 1) / BM), dim3(512), 0, (hipStream_t)stream, a);
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_win_ln_supported(int K, int D) { return D == BN && (K == 256 || K == 1024); }

static int win_ln_launch(const void* A, int lda, const void* W_packed, const float* bias, const void* res, const float* gamma,
                         const float* beta, void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int M,
                         int K, int D, float eps, float p_pre, uint32_t site_pre, const void* rng, const void* proj_w, const float* proj_bias,
                         int proj_Cout, void* proj_out, void* stream) {
  TTSK_REQUIRE(A && W_packed && bias && gamma && beta && out && mean && rstd, "ttsk_win_ln_fwd: null pointer");
  TTSK_REQUIRE(ttsk_win_ln_supported(K, D), "ttsk_win_ln_fwd: built for D = 256, K = 256 or 1024 (got K=%d D=%d)", K, D);
  TTSK_REQUIRE(M > 0 && (lda & 7) == 0 && lda >= K, "ttsk_win_ln_fwd: bad M / lda");
  TTSK_REQUIRE((((uintptr_t)A | (uintptr_t)W_packed | (uintptr_t)out | (uintptr_t)res | (uintptr_t)z_save) & 15) == 0, "ttsk_win_ln_fwd: operands must be 16-byte aligned");
  TTSK_REQUIRE(!(p_pre > 0.f) || rng, "ttsk_win_ln_fwd: dropout needs the rng state");
  TTSK_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "ttsk_win_ln_fwd: p_pre must be in [0, 1)");
  TTSK_REQUIRE(!lens || (seg_len > 0 && M % seg_len == 0), "ttsk_win_ln_fwd: lens needs M %% seg_len == 0");
  TTSK_REQUIRE(!proj_w || (proj_out && proj_Cout == 768 && (((uintptr_t)proj_w | (uintptr_t)proj_out | (uintptr_t)proj_bias) & 15) == 0),
               "ttsk_win_ln_proj_fwd: the projection is built for 768 output channels (q|k|v), 16-byte aligned operands");
  GemmLnArgs a{(const bf16_t*)A, (const bf16_t*)W_packed, bias, (const bf16_t*)res, gamma, beta, (bf16_t*)out, (bf16_t*)z_save, mean, rstd,
               (const long long*)lens, (const uint64_t*)rng, M, K, lda, 0, seg_len > 0 ? seg_len : 1, p_pre, eps, site_pre,
               (const bf16_t*)proj_w, proj_bias, (bf16_t*)proj_out, proj_Cout};
#ifdef TTSK_STAMPS
  a.stamps = g_wl_stamps;
#endif
  const dim3 grid((M + BM - 1) / BM);
  if (proj_w) {
    if (K == 256) hipLaunchKernelGGL((win_ln_kernel<256, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (grid.x <= WL_SPLIT_MAX_TILES) hipLaunchKernelGGL((win_ln_kernel<1024, true, 3>), dim3(grid.x * 3), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((win_ln_kernel<1024, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
  } else {
    if (K == 256) hipLaunchKernelGGL((win_ln_kernel<256, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((win_ln_kernel<1024, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
  }
  TTSK_CHECK_LAUNCH();
  return TTSK_OK;
}

extern "C" int ttsk_win_ln_fwd(const void* A, int lda, const void* W_packed, const float* bias, const void* res, const float* gamma,
                               const float* beta, void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len, int M,
                               int K, int D, float eps, float p_pre, uint32_t site_pre, const void* rng, void* stream) {
  return win_ln_launch(A, lda, W_packed, bias, res, gamma, beta, out, z_save, mean, rstd, lens, seg_len, M, K, D, eps, p_pre, site_pre, rng,
                       nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int ttsk_win_ln_proj_fwd(const void* A, int lda, const void* W_packed, const float* bias, const void* res, const float* gamma,
                                    const float* beta, void* out, void* z_save, float* mean, float* rstd, const int64_t* lens, int seg_len,
                                    int M, int K, int D, float eps, float p_pre, uint32_t site_pre, const void* rng, const void* proj_w_packed,
                                    const float* proj_bias, int proj_Cout, void* proj_out, void* stream) {
  TTSK_REQUIRE(proj_w_packed && proj_out, "ttsk_win_ln_proj_fwd: null pointer");
  return win_ln_launch(A, lda, W_packed, bias, res, gamma, beta, out, z_save, mean, rstd, lens, seg_len, M, K, D, eps, p_pre, site_pre, rng,
                       proj_w_packed, proj_bias, proj_Cout, proj_out, stream);
}
