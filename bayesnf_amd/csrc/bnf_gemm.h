// bnf_gemm.h -- the dense-contraction cores of a BayesNF layer (reference models.py:263-268
// forward and its autodiff, inference.py:602), batched over ensemble members.
//
//   gemm_nt   C[m][n] = sum_k A[m][k] Bt[n][k]        both operands K-contiguous
//     forward   Z  = H_l (rows x n_l) . K_l^T stored (W x n_l)          EPI_FWD / EPI_LAST
//     dgrad     dH = dZ_{l+1} (rows x W) . K_{l+1} stored (n x W)       EPI_DGRAD / EPI_DGRAD0
//   gemm_tn   C[i][j] = sum_r A[r][i] B[r][j]          both operands row-major (r slow)
//     wgrad     dK = H_l^T dZ_l                        (transpose reads in LDS; no transposed
//                                                       copy of any activation exists in HBM)
//   gemm_tn_ring / gemm_tn_skinny: the same contraction for the row-panel pipeline's bf16 shapes
//     (256 x 256 tile; 64 x 512 row stream of layer 0) with the K loop as a four-stage LDS-DMA ring
//     and the transpose reads as inline asm -- an HBM stream first
//
// Tiling (gfx950): a workgroup is a WGM x WGN grid of waves, each wave a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators (64 f32 VGPRs); 2x2 waves (128x128) by default, 4x4 (256x256) and
// 1xN / 2xN full-width panels where they pay (table in DESIGN.md section 4).  K advances through a
// ring of XOR-swizzled LDS stages filled by LDS-DMA (global_load_lds_dwordx4, swizzle on the
// source address, scalar base + 32-bit lane offset): bf16 64-byte rows (32 k), 3 stages, counted
// s_waitcnt vmcnt + one raw s_barrier per K tile; f32 128-byte rows, 2 stages.  bf16 multiplies
// with v_mfma_f32_32x32x16_bf16, f32 with the exact v_mfma_f32_32x32x2_f32.  Workgroup ids are
// remapped so that each XCD (private 4 MiB L2) owns a contiguous range of (member, tile) work.
#pragma once

#include <type_traits>

#include "bnf_device.h"

namespace bnf {

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_DGRAD0 = 2, EPI_WGRAD = 3, EPI_PLAIN = 4, EPI_LAST = 5 };

constexpr int kBM = 128, kBN = 128, kThreads = 256;
// gemm_tn: 2 x 32 KiB stages; gemm_nt<float>: the f32 epilogue tile (128 x 132 x 4 B) needs 66 KiB
constexpr int kGemmLds = 128 * (kBN + 4) * 4;

struct GemmArgs {
  const void* A;  // (M, K) elements of T, leading dim a_ld, member stride a_batch
  const void* B;  // (N, K)
  int64_t a_batch, b_batch;
  int32_t a_ld, b_ld;
  int32_t M, N, K;  // K: multiple of the tile depth, zero padded
  int32_t tiles_m, tiles_n, splitk, members;
  // gemm_tn_ring only: n_multi > 1 = that many contractions of the SAME shape in one launch (the W x W weight
  // gradients of every layer above layer 0: one launch fills the chip's last round once instead of once per layer)
  int32_t n_multi;
  const void* A_multi[BNF_MAX_LAYERS]; const void* B_multi[BNF_MAX_LAYERS];
  int32_t off_out_multi[BNF_MAX_LAYERS];
};

struct EpiArgs {
  const float* theta;  // (members, P) f32 compute parameters
  int64_t theta_stride;
  float scale;  // 1/sqrt(fan_in) folded into the accumulator
  int32_t off_bias, off_layer_scale, off_act_weight;
  const float* scal;   // (members, scal_stride): softplus(layer scales), [MAX_LAYERS] sigmoid(activation
  int32_t scal_stride, layer;   // weight), [MAX_LAYERS + 1] softplus(output scale)  -- k_member_scalars
  // activations: row-major (rows, ld) and transposed (ld, ldt) copies
  void* out_a;         // FWD: pre-activation A_l^T (transposed only)
  void* out_h;         // FWD: H_{l+1} row-major (null for the last layer) ; DGRAD: dZ_l row-major
  const void* in_a;    // DGRAD: A_l^T  (TAG 1)
  // DGRAD TAG 2 (layer below = layer 0): A_0 is recomputed from the layer-0 operands instead
  const void* aux_a;   //   H0 (rows, aux_ld)
  const void* aux_b;   //   packed K_0^T (W, aux_ld)
  int64_t aux_a_batch, aux_b_batch;
  int32_t aux_ld;      //   Fp
  float aux_scale;     //   1/sqrt(F)
  float* vdot;         // FWD (last layer): (members, vdot_batch) += H . k_o (un-normalised)
  int64_t vdot_batch;
  int32_t off_ko;      // offset of the output-layer kernel
  float inv_sw;        // EPI_LAST: 1 / sqrt(model width)
  int64_t act_batch;   // elements between members, row-major buffers
  int64_t actt_batch;  // elements between members, transposed buffers
  int32_t ld, ldt;
  // f32 outputs
  float* grad;         // (members, P): bias / scale grads (DGRAD), kernel grads (WGRAD)
  int64_t grad_stride;
  int32_t off_out;     // WGRAD: offset of Dense_l/kernel ; PLAIN/DGRAD0: unused
  float* out_f32;      // DGRAD0 / PLAIN: (members, M, ld_f32)
  int64_t f32_batch;
  int32_t ld_f32;
  // EPI_LAST (last hidden layer + output layer + likelihood + its backward, one kernel)
  const float* ybat;   // (members, row_batch) targets of the batch rows
  float* out;          // (members, out_batch) network output
  int64_t row_batch, out_batch;
  float* loss;         // loss[(e / S) * loss_stride] += loss_scale * step loss
  float* loss_raw;     // optional (members,) raw step loss
  const StepState* st; // graph replay: loss column offset
  int64_t loss_stride;
  int32_t S;
  float loss_scale, lik_c;   // lik_c = (N / B) * likelihood scale
  int32_t off_os, off_bias_out, off_lns, off_shape, off_infl, obs;
  // gemm_tn / gemm_tn_skinny, layer 0 of the panel kernel's F0 forms: row M of the (padded) output -- the ones column of
  // the feature operand -- is d bias0 (un-scaled), accumulated at grad[off_bias_row + n]
  int32_t bias_row, off_bias_row;
  // fp8 operand copies (bnf_gemm8.h): per-member product of the operands' storage scales, folded into the output
  const float* qscale;
  unsigned long long* prof;   // -DBNF_ENABLE_ABLATE builds: per-workgroup phase clocks (8 marks)
  int32_t ablate;      // perf experiments only (env BNF_ABLATE): 1 no transposed stores,
                       // 2 no row-major stores, 4 no activation math, 8 no row dot,
                       // 16 no K loop, 32 no transposed loads
};

template <typename T>
struct Mma;

// K-tile geometry per element type.  An LDS operand tile is 128 rows of kRowBytes; one
// LDS-DMA wave instruction (1 KiB) fills 1024 / kRowBytes consecutive rows, lane l ->
// (row l / kChunks, physical 16-byte chunk l % kChunks), and physical chunk c' of a row holds
// logical chunk c' ^ swz(row).  The swizzles make every ds_read_b128 lane group (16 lanes,
// MI355X_MICROARCH.md LDS table) cover all 64 banks:
//   128-byte rows (8 chunks): swz = (row >> 1) & 7      64-byte rows (4 chunks): swz = (row >> 2) & 3
// bf16 uses the SHORT rows: a ring of three 16 KiB stages (48 KiB, also the epilogue tile), so
// three workgroups share a CU and each keeps two K tiles of loads in flight while it
// multiplies a third.
template <>
struct Mma<bf16_t> {
  static constexpr int kRowBytes = 64;
  static constexpr int kTileK = 32;  // elements per row
  static constexpr int kSteps = 2;   // MFMA k-steps (16 elements) per tile
  // ring of three K tiles (two for the 64-row full-width panel, so two workgroups fit a CU)
  __host__ __device__ static constexpr int stages(int wgm, int wgn) { return (wgm == 1 && wgn == 8) ? 2 : 3; }
  // LDS of a (64 WG x 64 WG) tile: the larger of the ring and the epilogue tile (pitch + 16 B)
  __host__ __device__ static constexpr int lds_bytes(int wgm, int wgn, int extra) {
    const int ring = stages(wgm, wgn) * 64 * (wgm + wgn) * kRowBytes, tile = 64 * wgm * (64 * wgn + 8) * 2 + extra;
    return ring > tile ? ring : tile;
  }
  // registers: 3 waves / SIMD for the 4-wave grid (LDS allows 3 workgroups per CU), 4 for
  // the larger grids (16 waves: one workgroup per CU)
  __host__ __device__ static constexpr int min_waves(int waves) { return waves <= 4 ? 3 : 4; }
  __device__ static __forceinline__ int swz(int row) { return (row >> 2) & 3; }
  struct Frag {
    bf16x8 v;
  };
  // lane (row, kg = lane >> 5) takes 8 consecutive k at chunk 2*ks + kg
  __device__ static __forceinline__ Frag load(const char* row, int swz, int ks, int kg) {
    Frag f;
    f.v = *reinterpret_cast<const bf16x8*>(row + (((ks * 2 + kg) ^ swz) << 4));
    return f;
  }
  __device__ static __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
  }
};

template <>
struct Mma<float> {
  static constexpr int kRowBytes = 128;
  static constexpr int kTileK = 32;
  static constexpr int kSteps = 2;  // k-groups of 16 floats (64 bytes)
  __host__ __device__ static constexpr int stages(int wgm, int wgn) { return 2; }
  __host__ __device__ static constexpr int lds_bytes(int wgm, int wgn, int extra) {
    const int ring = stages(wgm, wgn) * 64 * (wgm + wgn) * kRowBytes, tile = 64 * wgm * (64 * wgn + 4) * 4 + extra;
    return ring > tile ? ring : tile;
  }
  __host__ __device__ static constexpr int min_waves(int waves) { return 2; }
  __device__ static __forceinline__ int swz(int row) { return (row >> 1) & 7; }
  struct Frag {
    f32x4 lo, hi;
  };
  // lane (row, kg) takes 8 consecutive k at chunks 4*ks + 2*kg, +1.  MFMA j then
  // contracts k = base + j (lanes 0-31) and base + 8 + j (lanes 32-63): a
  // permutation of k shared by both operands, so the sum is unchanged.
  __device__ static __forceinline__ Frag load(const char* row, int swz, int ks, int kg) {
    Frag f;
    const int c = ks * 4 + kg * 2;
    f.lo = *reinterpret_cast<const f32x4*>(row + (((c) ^ swz) << 4));
    f.hi = *reinterpret_cast<const f32x4*>(row + (((c + 1) ^ swz) << 4));
    return f;
  }
  __device__ static __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
  }
  // BNF_DTYPE_F32S (round 5): the same contraction on SPLIT-bf16 MFMAs.  Each f32 operand is split IN REGISTERS into two
  // bf16 pieces, x = hi + lo with hi = bf16(x) and lo = bf16(x - hi) (round to nearest even: 16 operand bits, |x - hi - lo| <=
  // 2^-18 |x|), and the three products that matter -- lo*hi, hi*lo, hi*hi, smallest first -- are summed by three
  // v_mfma_f32_32x32x16_bf16 (32 cycles each, f32 accumulate) where the exact chain issues eight v_mfma_f32_32x32x2_f32 of 64
  // cycles: 96 against 512 matrix-pipe cycles per fragment pair, + 20 VALU per fragment for the split.  The lane layout needs
  // no change: a lane's 8 consecutive k ARE the bf16 32x32x16 fragment.  Measured (profiles/r05_panel_ab.md r05l): a
  // contraction is good to 5e-6 of its largest output (exact chain: 1e-7), the f32 storage, accumulation and epilogues are
  // unchanged; every fp32 parity bar and all three golden reproductions (< 1e-4) hold, the C2 step goes from 11.1 to 6.5 ms.
  // (A third piece -- six MFMAs, 24 operand bits -- passes even the 2e-6 exactness tests but is only 6 % faster than f32.)
  struct Split {
    bf16x8 hi, lo;
  };
  __device__ static __forceinline__ Split split(const Frag& f) {
    // five instructions per pair of elements: v_cvt_pk_bf16_f32 (hi), shift / mask back to two floats, ONE packed
    // subtraction, v_cvt_pk_bf16_f32 (lo).  (Left to itself hipcc converts the first element of every pair twice.)
    Split s;
    const f32x2 x[4] = {{f.lo[0], f.lo[1]}, {f.lo[2], f.lo[3]}, {f.hi[0], f.hi[1]}, {f.hi[2], f.hi[3]}};
    uint32_t wh[4], wl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wh[q] = pack_bf16x2(x[q][0], x[q][1]);
      asm("" : "+v"(wh[q]));
      const f32x2 hf = {__builtin_bit_cast(float, wh[q] << 16), __builtin_bit_cast(float, wh[q] & 0xffff0000u)};
      const f32x2 d = x[q] - hf;
      wl[q] = pack_bf16x2(d[0], d[1]);
    }
    s.hi = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    s.lo = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
    return s;
  }
  // the same split as hipcc schedules it on its own (seven instructions per pair, no opaque value in the chain): what
  // gemm_tn's gathered fragments run fastest with (r05t: 1374 against 1532 us with the form above)
  __device__ static __forceinline__ Split split_plain(const Frag& f) {
    Split s;
    const float x[8] = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
    uint32_t wh[4], wl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wh[q] = pack_bf16x2(x[2 * q], x[2 * q + 1]);
      wl[q] = pack_bf16x2(x[2 * q] - __builtin_bit_cast(float, wh[q] << 16), x[2 * q + 1] - __builtin_bit_cast(float, wh[q] & 0xffff0000u));
    }
    s.hi = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    s.lo = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
    return s;
  }
  // an operand that was split when it was PACKED (the Dense kernels: k_pack_weights writes, in place of the 8 floats of
  // every aligned group of 8 consecutive k, their 8 hi parts and then their 8 lo parts -- the same 32 bytes): the
  // fragment's first 16 bytes are the hi operand, the second 16 the lo operand, no arithmetic
  __device__ static __forceinline__ Split presplit(const Frag& f) {
    Split s;
    s.hi = __builtin_bit_cast(bf16x8, f.lo);
    s.lo = __builtin_bit_cast(bf16x8, f.hi);
    return s;
  }
  __device__ static __forceinline__ void mma_split(f32x16& acc, const Split& a, const Split& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
  }
};

// XCD-aware bijective remap of the linear workgroup id (8 XCDs, block b runs on
// XCD b % 8): XCD x gets the contiguous logical range starting at first(x).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t total) {
  const uint32_t q = total >> 3, r = total & 7u, x = b & 7u, slot = b >> 3;
  const uint32_t first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return first + slot;
}

// Bytes of LDS an epilogue needs beyond the (rows x cols) output tile of `esz`-byte elements.
// EPI_LAST: row-dot partials (rows x wgn), dv (rows), column sums 2 x (wgm x cols), scalars;
// they sit behind max(output tile, row-dot scratch = 64 rows x 36 floats per wave).
#ifndef BNF_TN_AUX
#define BNF_TN_AUX 0   // cache-policy bits of gemm_tn's operand loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef BNF_SK_AUX
#define BNF_SK_AUX 2   // gemm_tn_skinny reads every byte once: non-temporal
#endif
constexpr int kRowDotPitch = 36;   // floats: 16 lanes x ds_read_b128 at this pitch hit 64 distinct banks
__host__ __device__ constexpr int epi_extra_lds(int epi, int wgm, int wgn, int esz) {
  if (epi != EPI_LAST) return 0;
  const int xs = (64 * wgm * wgn + 64 * wgm + 2 * wgm * 64 * wgn + 16 + 3 * wgm * wgn) * 4;
  const int tile = 64 * wgm * (64 * wgn + 16 / esz) * esz;
  const int scratch = wgm * wgn * 64 * kRowDotPitch * 4;
  return xs + (scratch > tile ? scratch - tile : 0);
}

// WGM x WGN waves per workgroup, each wave a 64 x 64 sub-tile:
//   2 x 2  128 x 128 tiles, any width (the default);
//   4 x 4  256 x 256 tiles, 16 waves: bf16 forward layers whose width is a multiple of 256.
//          The K loop of the small tile is bound by the L2 -> LDS operand stream (64 flop /
//          byte; it does not speed up with occupancy or prefetch depth); the large tile
//          halves that stream and quarters the number of workgroups and reduction atomics;
//   2 x N  128-row panels spanning the WHOLE layer width (EPI_LAST): the workgroup owns
//          complete rows, so the output-layer dot, the likelihood and the backward pass of
//          the last activation all happen on the accumulators (no A_L^T round trip).
// SPLIT (float only): the contraction on split-bf16 MFMAs (BNF_DTYPE_F32S, Mma<float>::mma_split) -- its own instantiation:
// with both arithmetic paths behind a run-time flag in one kernel every f32 contraction lost 2 - 4x to register pressure
// (gpurun_out/r05m: C2 step 12.7 / 18.3 ms against 6.5 / 11.1 as separate instantiations).  SPLIT 1: both operands split in
// registers; 2: B is a Dense kernel that k_pack_weights split when it packed it (every launch but the test entry's plain
// product) -- the K loop's VALU goes from 424 to 174 instructions per 48 MFMAs, the layer contractions get 9 - 12 % shorter
// (they are then bound by the f32 operand stream through LDS-DMA, ~13 B/clk/CU: profiles/r05_panel_ab.md r05t)
template <typename T, int EPI, int TAG, int WGM, int WGN, int SPLIT = 0>
__global__ __launch_bounds__(64 * WGM * WGN, Mma<T>::min_waves(WGM * WGN)) void gemm_nt(const GemmArgs g, const EpiArgs ep) {
  using M_ = Mma<T>;
  constexpr int kBM = 64 * WGM, kBN = 64 * WGN, kThreads = 64 * WGM * WGN, kWaves = WGM * WGN;
  constexpr int kRowBytes = M_::kRowBytes;
  constexpr int kStageBytes = (kBM + kBN) * kRowBytes;
  constexpr int kChunks = kRowBytes / 16;           // 16-byte chunks per row
  constexpr int kRowsPerInstr = 64 / kChunks;       // rows one LDS-DMA wave instruction fills
  constexpr int kInstr = (kBM + kBN) / kRowsPerInstr;   // LDS-DMA instructions per stage
  constexpr int kPerWave = (kInstr + kWaves - 1) / kWaves, kRem = kInstr % kWaves;
  static_assert(kBM % 16 == 0, "the swizzle period must divide the A tile");
  constexpr bool FAST = Elem<T>::kFast;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index is uniform: keeping it in an SGPR makes the stage bookkeeping scalar code
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WGN, wc = wave % WGN;
  BNF_MARK(ep, 0);

  // ---- which (member, k-split, tile) ----------------------------------------
  const uint32_t per_member = (uint32_t)(g.tiles_m * g.tiles_n * g.splitk);
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = (int)(w / (uint32_t)tiles);
  w -= (uint32_t)split * tiles;
  const int tm = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int m0 = tm * kBM, n0 = tn * kBN;

  const int nk_total = g.K / M_::kTileK;
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)e * g.a_batch * Elem<T>::kBytes;
  const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)e * g.b_batch * Elem<T>::kBytes;

  // ---- staging map: global -> LDS by LDS-DMA (global_load_lds, 16 B per lane) ---
  // A stage is kBM rows of A followed by kBN rows of B.  One wave instruction fills
  // kRowsPerInstr consecutive rows; instruction q of a stage belongs to wave q % kWaves.  The
  // XOR swizzle lives on the SOURCE side, so there are no staging VGPRs and no ds_write
  // traffic.  When kInstr is not a multiple of kWaves the first kRem waves issue one more.
  // Address of a piece = (operand base of this tile and K tile: 64-bit, SCALAR) + (row and
  // swizzled chunk of this lane inside the tile: 32-bit, constant over the K loop).
  struct StageMap {
    const char* sbase[kPerWave];   // uniform
    uint32_t voff[kPerWave];       // per lane
  };
  int lds_base[kPerWave];          // uniform
  const bool last_slot = (kRem == 0) || (wave < kRem);   // does slot kPerWave-1 exist for this wave
  auto make_map = [&](const char* a_base, int a_ld, const char* b_base, int b_ld) {
    StageMap sm;
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
      const int q = min(wave + i * kWaves, kInstr - 1);
      const int r0 = q * kRowsPerInstr;                 // uniform; a piece never straddles A | B
      const int row = r0 + lane / kChunks, cp = lane % kChunks;
      const int c = cp ^ M_::swz(row);
      if (r0 < kBM) {
        const int lim = g.M - 1 - m0;                   // rows past M re-read the last valid row
        sm.sbase[i] = a_base + (int64_t)m0 * a_ld * Elem<T>::kBytes;
        sm.voff[i] = (uint32_t)(min(row, lim) * a_ld * Elem<T>::kBytes + c * 16);
      } else {
        const int lim = g.N - 1 - n0;
        sm.sbase[i] = b_base + (int64_t)n0 * b_ld * Elem<T>::kBytes;
        sm.voff[i] = (uint32_t)(min(row - kBM, lim) * b_ld * Elem<T>::kBytes + c * 16);
      }
      lds_base[i] = r0 * kRowBytes;
    }
    return sm;
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DGRAD: this lane's 2x2x4 groups of four consecutive rows of A_l^T, fetched
  // (undecoded) before the K loop so the scattered, HBM-latency loads hide under
  // the MFMAs.  A_l^T has ldt >= tiles_m * 128 zero-padded columns, so the
  // 4-row vector is always in bounds.
  typename Raw<T>::R4 apre[(EPI == EPI_DGRAD) ? 2 : 1][(EPI == EPI_DGRAD) ? 2 : 1][(EPI == EPI_DGRAD) ? 4 : 1];
  if constexpr (EPI == EPI_DGRAD && TAG != 2) {
    const T* iat = reinterpret_cast<const T*>(ep.in_a) + (int64_t)e * ep.actt_batch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = min(n0 + wc * 64 + j * 32 + (lane & 31), g.N - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int mb = m0 + wr * 64 + 4 * (lane >> 5) + i * 32 + 8 * rg;
          apre[j][i][rg] = load_raw4(iat + (int64_t)n * ep.ldt + mb);
        }
    }
  }

  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void glb_void_t;
  auto stage = [&](const StageMap& sm, int buf, int kt) {
    const int64_t koff = (int64_t)kt * kRowBytes;
    char* sS = smem + buf * kStageBytes;
#pragma unroll
    for (int i = 0; i < kPerWave; ++i)
      if (i < kPerWave - 1 || last_slot) {
        // pin (tile base + K offset) in SGPRs: left alone the compiler re-associates the sum into
        // a loop-invariant 64-bit per-lane pointer per piece (two VGPRs each, spilled in DGRAD)
        const uint64_t b = (uint64_t)(sm.sbase[i] + koff);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const char* sp = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(sp + sm.voff[i]), (lds_void_t*)(sS + lds_base[i]), 16, 0, 0);
      }
  };

  // fragment rows of this lane
  const int frow = lane & 31, kg = lane >> 5;
  int a_row[2], b_row[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_row[i] = wr * 64 + i * 32 + frow;
    b_row[i] = wc * 64 + i * 32 + frow;
    a_swz[i] = M_::swz(a_row[i]);
    b_swz[i] = M_::swz(b_row[i]);
  }

  // K loop: a ring of kStages LDS stages, filled kStages-1 tiles ahead by LDS-DMA.  One
  // s_barrier per K tile: passing it means (a) every wave's share of tile kt has landed
  // (each wave first waits on its own vmcnt, allowing only the younger prefetches to stay in
  // flight) and (b) every wave is done reading tile kt-1, whose buffer the next prefetch
  // overwrites.  (__syncthreads() would drain vmcnt to 0 and serialise load and compute.)
  constexpr int kStages = M_::stages(WGM, WGN);
  // s_waitcnt immediates (gfx9 encoding): vmcnt in [3:0] + [15:14], expcnt 7 / lgkmcnt 15 = no wait
  constexpr int kAheadHi = (kStages - 2) * kPerWave;             // this wave's younger DMA instructions
  constexpr int kAheadLo = (kStages - 2) * (kPerWave - 1);       // ... for waves without the last slot
  static_assert(kAheadHi < 64, "vmcnt range");
  constexpr int kWaitHi = (kAheadHi & 15) | ((kAheadHi >> 4) << 14) | 0x0F70;
  constexpr int kWaitLo = (kAheadLo & 15) | ((kAheadLo >> 4) << 14) | 0x0F70;
  constexpr int kWaitAll = 0x0F70;
  // acc += A[:, k0..k1) . B[:, k0..k1)^T over K tiles [k0, k1) of the operands behind `sm`
  auto k_loop = [&](const StageMap& sm, int k0, int k1) {
#pragma unroll
    for (int s = 0; s < kStages - 1; ++s)
      if (k0 + s < k1) stage(sm, s, k0 + s);
    // one trip = kStages K tiles, so every stage index below is a compile-time constant
    // (LDS offsets become instruction immediates)
    for (int ktb = k0; ktb < k1; ktb += kStages) {
#pragma unroll
      for (int sb = 0; sb < kStages; ++sb) {
        const int kt = ktb + sb;
        if (kt >= k1) break;
        if (kt + kStages - 2 >= k1) __builtin_amdgcn_s_waitcnt(kWaitAll);
        else if (last_slot) __builtin_amdgcn_s_waitcnt(kWaitHi);
        else __builtin_amdgcn_s_waitcnt(kWaitLo);
        __builtin_amdgcn_s_barrier();
        if (kt + kStages - 1 < k1) stage(sm, (sb + kStages - 1) % kStages, kt + kStages - 1);
        const char* sA = smem + sb * kStageBytes;
        const char* sB = sA + kBM * kRowBytes;
#pragma unroll
        for (int ks = 0; ks < M_::kSteps; ++ks) {
          typename M_::Frag fa[2], fb[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[i] = M_::load(sA + a_row[i] * kRowBytes, a_swz[i], ks, kg);
            fb[i] = M_::load(sB + b_row[i] * kRowBytes, b_swz[i], ks, kg);
          }
          if constexpr (std::is_same<T, float>::value && SPLIT) {   // BNF_DTYPE_F32S: three bf16 MFMAs per fragment pair
            typename Mma<float>::Split sa[2], sb2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              sa[i] = Mma<float>::split(fa[i]);
              if constexpr (SPLIT == 2) sb2[i] = Mma<float>::presplit(fb[i]);   // B = Dense kernels packed as hi | lo
              else sb2[i] = Mma<float>::split(fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) Mma<float>::mma_split(acc[i][j], sa[i], sb2[j]);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) M_::mma(acc[i][j], fa[i], fb[j]);
          }
        }
      }
    }
    __syncthreads();   // the next user of the stage buffers (K loop or epilogue) may overwrite them
  };
  if constexpr (EPI == EPI_DGRAD && TAG == 2) {
    // The layer below is layer 0, whose contraction depth is only Fp: recompute its
    // pre-activation A_0 = gamma_0 (H0 K_0 / sqrt F + b_0) for this tile (the same MFMA sequence
    // and the same rounding as the forward kernel -> bit-identical to what it would have
    // stored) instead of writing A_0^T in the forward pass and gathering it here.
    const char* Xa = reinterpret_cast<const char*>(ep.aux_a) + (int64_t)e * ep.aux_a_batch * Elem<T>::kBytes;
    const char* Xb = reinterpret_cast<const char*>(ep.aux_b) + (int64_t)e * ep.aux_b_batch * Elem<T>::kBytes;
    const StageMap saux = make_map(Xa, ep.aux_ld, Xb, ep.aux_ld);
    k_loop(saux, 0, ep.aux_ld / M_::kTileK);
    const float* th0 = ep.theta + (int64_t)e * ep.theta_stride;
    const float gs0 = ep.scal[(int64_t)e * ep.scal_stride + ep.layer] * ep.aux_scale;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = min(n0 + wc * 64 + j * 32 + (lane & 31), g.N - 1);
      const float gb0 = ep.scal[(int64_t)e * ep.scal_stride + ep.layer] * th0[ep.off_bias + n];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          apre[j][i][rg] = pack_raw4((const T*)nullptr, acc[i][j][rg * 4] * gs0 + gb0,
                                     acc[i][j][rg * 4 + 1] * gs0 + gb0, acc[i][j][rg * 4 + 2] * gs0 + gb0,
                                     acc[i][j][rg * 4 + 3] * gs0 + gb0);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i][j][rg * 4 + q] = 0.f;
        }
    }
  }
  BNF_MARK(ep, 1);
  if (kt0 < kt1 && !BNF_ABL(ep, 16)) {
    const StageMap smain = make_map(Ab, g.a_ld, Bb, g.b_ld);
    k_loop(smain, kt0, kt1);
  }
  BNF_MARK(ep, 2);

  // ---- epilogues --------------------------------------------------------------
  // accumulator element (i, j, r): row m = m0 + wr*64 + i*32 + 8*(r>>2) + 4*kg + (r&3)
  //                                col n = n0 + wc*64 + j*32 + frow
  // A lane therefore owns, per (i, j, r>>2), FOUR CONSECUTIVE ROWS of one column:
  // transposed (column-major) activations are written / read as 8- or 16-byte
  // vectors straight from the registers, while row-major tiles are staged through
  // LDS (free after the K loop) and leave as coalesced 16-byte stores.
  const int mw = m0 + wr * 64 + 4 * kg;
  const int nw = n0 + wc * 64 + frow;
  constexpr int kEpc = 16 / Elem<T>::kBytes;   // elements per 16-byte chunk
  constexpr int kPitch = kBN + kEpc;           // LDS tile pitch (elements), 1 chunk of padding

  // cooperative copy of the staged (kBM x kBN) tile of T to a row-major array
  auto tile_to_global = [&](T* dst, int ld) {
    const T* tile = reinterpret_cast<const T*>(smem);
    constexpr int kCpr = kBN / kEpc;  // chunks per row
#pragma unroll
    for (int c = 0; c < (kBM * kCpr) / kThreads; ++c) {
      const int q = tid + c * kThreads;
      const int row = q / kCpr, cc = q % kCpr;
      const int m = m0 + row, n = n0 + cc * kEpc;
      if (m < g.M && n < g.N)
        *reinterpret_cast<u32x4*>(dst + (int64_t)m * ld + n) =
            *reinterpret_cast<const u32x4*>(tile + row * kPitch + cc * kEpc);
    }
  };

  if constexpr (EPI == EPI_PLAIN) {
    float* out = ep.out_f32 + (int64_t)e * ep.f32_batch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32;
      if (n >= g.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
          if (m < g.M) out[(int64_t)m * ep.ld_f32 + n] = acc[i][j][r] * ep.scale;
        }
    }
  } else if constexpr (EPI == EPI_DGRAD0) {
    // dH0^T (Fp, ldt) f32: column n of the tile is a row of the output
    float* out = ep.out_f32 + (int64_t)e * ep.f32_batch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32;
      if (n >= g.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int mb = mw + i * 32 + 8 * rg;
          float* p = out + (int64_t)n * ep.ld_f32 + mb;
          const float s = ep.scale;
          if (mb + 3 < g.M) {
            store4(p, acc[i][j][rg * 4] * s, acc[i][j][rg * 4 + 1] * s, acc[i][j][rg * 4 + 2] * s,
                   acc[i][j][rg * 4 + 3] * s);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (mb + q < g.M) p[q] = acc[i][j][rg * 4 + q] * s;
          }
        }
    }
  } else if constexpr (EPI == EPI_WGRAD) {
    float* out = ep.grad + (int64_t)e * ep.grad_stride + ep.off_out;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32;
      if (n >= g.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
          if (m < g.M) {
            const float v = acc[i][j][r] * ep.scale;
            if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
            else out[(int64_t)m * ep.ld_f32 + n] = v;
          }
        }
    }
  } else if constexpr (EPI == EPI_FWD) {
    const float* th = ep.theta + (int64_t)e * ep.theta_stride;
    // per-member transforms of the scalar leaves, precomputed by k_member_scalars (every lane
    // would otherwise spend ~200 VALU instructions on two log1p / exp expansions)
    const float gamma = ep.scal[(int64_t)e * ep.scal_stride + ep.layer];
    const float alpha = ep.scal[(int64_t)e * ep.scal_stride + BNF_MAX_LAYERS];
    // A_l^T (W, ldt); null when the backward pass recomputes it (layer 0, DGRAD TAG 2)
    T* oat = ep.out_a ? reinterpret_cast<T*>(ep.out_a) + (int64_t)e * ep.actt_batch : nullptr;
    T* oh = ep.out_h ? reinterpret_cast<T*>(ep.out_h) + (int64_t)e * ep.act_batch : nullptr;
    float* vd = ep.vdot ? ep.vdot + (int64_t)e * ep.vdot_batch : nullptr;
    T* tile = reinterpret_cast<T*>(smem);
    float pdot[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) pdot[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32;
      if (n >= g.N) continue;
      const float gs = gamma * ep.scale, gb = gamma * th[ep.off_bias + n];   // A = acc gs + gb
      const float kov = vd ? th[ep.off_ko + n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int mb = mw + i * 32 + 8 * rg;
          float av[4], hv[4];
          if constexpr (FAST) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 a2 = f32x2{acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]} * gs + gb;
              const f32x2 h2 = BNF_ABL(ep, 4) ? a2 : act_fwd2(a2, alpha);
              av[q] = a2.x; av[q + 1] = a2.y;
              hv[q] = h2.x; hv[q + 1] = h2.y;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              av[q] = acc[i][j][rg * 4 + q] * gs + gb;
              hv[q] = BNF_ABL(ep, 4) ? av[q] : act_fwd<FAST>(av[q], alpha);
            }
          }
          if (vd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pdot[i][rg * 4 + q] += hv[q] * kov;
          }
          if (oh && !BNF_ABL(ep, 2)) {
            const int lr = wr * 64 + 4 * kg + i * 32 + 8 * rg, lc = wc * 64 + j * 32 + frow;
#pragma unroll
            for (int q = 0; q < 4; q += 2)
              store_pair(tile + (lr + q) * kPitch + lc, tile + (lr + q + 1) * kPitch + lc, hv[q], hv[q + 1]);
          }
          // A_l^T has ldt >= tiles_m * 128 columns: the four-row vector is always in bounds
          // (rows past M hold copies of the last row -- the operand loader clamps -- which
          // the backward pass masks out)
          if (oat && !BNF_ABL(ep, 1)) store4(oat + (int64_t)n * ep.ldt + mb, av[0], av[1], av[2], av[3]);
          // keep the 32 four-row groups sequential: without the fence the scheduler
          // interleaves all of them and the live ranges spill
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (vd && !BNF_ABL(ep, 8)) {
      // output-layer row dot (models.py:269-273): sum over this tile's columns,
      // then one atomic per row and wave.
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sacc = half_wave_sum_dpp(pdot[i][r]);
          const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
          if (frow == 16 && m < g.M) atomicAdd(&vd[m], sacc);
        }
    }
    if (oh && !BNF_ABL(ep, 2)) {
      __syncthreads();
      tile_to_global(oh, ep.ld);
    }
  } else if constexpr (EPI == EPI_DGRAD) {
    const float* th = ep.theta + (int64_t)e * ep.theta_stride;
    // per-member transforms of the scalar leaves, precomputed by k_member_scalars (every lane
    // would otherwise spend ~200 VALU instructions on two log1p / exp expansions)
    const float gamma = ep.scal[(int64_t)e * ep.scal_stride + ep.layer];
    const float alpha = ep.scal[(int64_t)e * ep.scal_stride + BNF_MAX_LAYERS];
    T* oz = reinterpret_cast<T*>(ep.out_h) + (int64_t)e * ep.act_batch;
    T* tile = reinterpret_cast<T*>(smem);
    const bool full_tile = m0 + kBM <= g.M;
    // running sums per element parity (folded after the loop)
    f32x2 sa2 = {0.f, 0.f}, sg2 = {0.f, 0.f}, cs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    // bf16 (FAST): the lean activation core (bnf_device.h) with gamma * scale folded into the constants of
    // act' -- dZ = z = raw act'_folded, sa2 = sum raw (elu - tanh), sg2 = sum z t (d gamma ~ (ln 2 / gamma)
    // sum z t), cs2 = sum z
    [[maybe_unused]] const ActConst ak = act_const(alpha);
    [[maybe_unused]] const float gza = gamma * ep.scale * ak.alpha, gzc = gamma * ep.scale * ak.c2;
    // two instances of the element loop: only the last row tile needs the row masks
    auto dgrad_tile = [&](auto full_c) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = nw + j * 32;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int mb = mw + i * 32 + 8 * rg;
            float av[4], zv[4];
            unpack(apre[j][i][rg], av);
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 a2 = {av[q], av[q + 1]};
              if constexpr (FAST) {
                f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
                // rows past M carry copies of the last row: masked out of the accumulator
                if constexpr (!decltype(full_c)::value)
                  raw = raw * f32x2{mb + q < g.M ? 1.f : 0.f, mb + q + 1 < g.M ? 1.f : 0.f};
                const f32x2 tv = a2 * kLog2e;
                f32x2 z = raw;
                if (!BNF_ABL(ep, 4)) {
                  const ActCore2 c = act_core2(tv);
                  const f32x2 s = kLn2 * c.mxt + c.dl;
                  const f32x2 dg = gzc * (c.r - c.r * c.r) + gza * c.dl;
                  z = raw * dg;
                  sa2 += raw * ((2.f * c.r + s) - 2.f);   // (in place: a running sum of raw would cost two more registers -> spills)
                }
                sg2 += z * tv;
                cs2[j] += z;
                zv[q] = z.x; zv[q + 1] = z.y;
              } else {
                // rows past M carry copies of the last row: masked through the scale
                f32x2 ms = {ep.scale, ep.scale};
                if constexpr (!decltype(full_c)::value)
                  ms = f32x2{mb + q < g.M ? ep.scale : 0.f, mb + q + 1 < g.M ? ep.scale : 0.f};
                const f32x2 dh = f32x2{acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]} * ms;
                ActOut2 o;
                if (BNF_ABL(ep, 4)) { o.h = a2; o.dact = f32x2{1.f, 1.f}; o.ediff = a2; }
                else {
                  const ActOut o0 = act_eval<FAST>(a2.x, alpha), o1 = act_eval<FAST>(a2.y, alpha);
                  o.h = f32x2{o0.h, o1.h}; o.dact = f32x2{o0.dact, o1.dact}; o.ediff = f32x2{o0.ediff, o1.ediff};
                }
                sa2 += dh * o.ediff;
                const f32x2 da = dh * o.dact;
                sg2 += da * a2;
                const f32x2 z = gamma * da;
                cs2[j] += z;
                zv[q] = z.x; zv[q + 1] = z.y;
              }
            }
            const int lr = wr * 64 + 4 * kg + i * 32 + 8 * rg, lc = wc * 64 + j * 32 + frow;
            if (!BNF_ABL(ep, 2)) {
#pragma unroll
              for (int q = 0; q < 4; q += 2)
                store_pair_pk(tile + (lr + q) * kPitch + lc, tile + (lr + q + 1) * kPitch + lc, zv[q], zv[q + 1]);
            }
            // pin the running sums (see EPI_LAST): keeps the add chains from sinking to their use
            asm volatile("" : "+v"(sa2), "+v"(sg2), "+v"(cs2[0]), "+v"(cs2[1]));
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    };
    if (full_tile) dgrad_tile(std::true_type{});
    else dgrad_tile(std::false_type{});
    float s_alpha = sa2.x + sa2.y, s_gamma = sg2.x + sg2.y;
    if constexpr (FAST) {
      s_alpha *= ep.scale;
      s_gamma *= kLn2 / gamma;
    }
    const float colsum[2] = {cs2[0].x + cs2[0].y, cs2[1].x + cs2[1].y};
    __syncthreads();
    if (!BNF_ABL(ep, 2)) tile_to_global(oz, ep.ld);
    __syncthreads();
    // block reduction of bias / scale gradients through LDS
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float c = colsum[j] + __shfl_xor(colsum[j], 32, 64);
      if (lane < 32) red[wr * kBN + wc * 64 + j * 32 + lane] = c;
    }
    const float sa = wave_sum(s_alpha), sg = wave_sum(s_gamma);
    if (lane == 0) {
      red[WGM * kBN + wave * 2] = sa;
      red[WGM * kBN + wave * 2 + 1] = sg;
    }
    __syncthreads();
    float* gr = ep.grad + (int64_t)e * ep.grad_stride;
    if (tid < kBN) {
      const int n = n0 + tid;
      float c = 0.f;
#pragma unroll
      for (int r = 0; r < WGM; ++r) c += red[r * kBN + tid];
      if (n < g.N) atomicAdd(&gr[ep.off_bias + n], c);
    }
    if (tid == 0) {
      float ta = 0.f, tg = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) {
        ta += red[WGM * kBN + w2 * 2];
        tg += red[WGM * kBN + w2 * 2 + 1];
      }
      atomicAdd(&gr[ep.off_act_weight], alpha * (1.f - alpha) * ta);
      atomicAdd(&gr[ep.off_layer_scale], sigmoidf(th[ep.off_layer_scale]) * tg / gamma);
    }
  } else if constexpr (EPI == EPI_LAST) {
    // The workgroup owns kBM complete rows (kBN == N == layer width).
    //   1. A = gamma (acc s + b) kept in the accumulators; H = act(A); row dots H . k_o
    //   2. one thread per row: output, likelihood, d out  (row_loss_eval, models.py:157-191)
    //   3. dZ = gamma (dv k_o / sqrt W) act'(A) -> LDS tile -> coalesced rows; column sums
    // Rows past M carry copies of the last row (clamped operand loads) and dv = 0.
    const float* th = ep.theta + (int64_t)e * ep.theta_stride;
    // per-member transforms of the scalar leaves, precomputed by k_member_scalars (every lane
    // would otherwise spend ~200 VALU instructions on two log1p / exp expansions)
    const float gamma = ep.scal[(int64_t)e * ep.scal_stride + ep.layer];
    const float alpha = ep.scal[(int64_t)e * ep.scal_stride + BNF_MAX_LAYERS];
    const float inv_sw = ep.inv_sw;   // 1 / sqrt(model width): fan-in of the output layer
    T* tile = reinterpret_cast<T*>(smem);
    constexpr int kTileBytes = kBM * kPitch * (int)sizeof(T), kDotBytes = kWaves * 64 * kRowDotPitch * 4;
    float* xs = reinterpret_cast<float*>(smem + (kTileBytes > kDotBytes ? kTileBytes : kDotBytes));
    float* s_dot = reinterpret_cast<float*>(smem) + wave * (64 * kRowDotPitch);   // this wave's [64 rows][32 lanes]
    float* s_part = xs;                       // [kBM][WGN]
    float* s_dv = s_part + kBM * WGN;         // [kBM]
    float* s_col = s_dv + kBM;                // [2][WGM][kBN]
    float* s_sc = s_col + 2 * WGM * kBN;      // scalars
    [[maybe_unused]] const ActConst ak = act_const(alpha);
    float ksum = 0.f;    // FAST: c0 * (sum of k_o over this wave's columns), the constant term of the row dot
    {
      // FAST (bf16): the accumulators keep t = A log2(e) from here on (lean activation core, bnf_device.h);
      // act(A) = c0 + c1 r + alpha s, so the row dot with k_o accumulates (k_o alpha) s + (k_o c1) r
      const float gs = gamma * ep.scale * (FAST ? kLog2e : 1.f);   // A = acc gs + gb
      float gb[2], kov[2];
      [[maybe_unused]] float ka[2], kc1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gb[j] = gamma * th[ep.off_bias + nw + j * 32] * (FAST ? kLog2e : 1.f);
        kov[j] = th[ep.off_ko + nw + j * 32];
        ka[j] = kov[j] * ak.alpha; kc1[j] = kov[j] * ak.c1;
      }
      if constexpr (FAST) {
        ksum = wave_sum(kg == 0 ? kov[0] + kov[1] : 0.f);
        if (lane == 0) s_sc[16 + 2 * kWaves + wave] = ksum;
        ksum *= ak.c0;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (FAST) {
#pragma unroll
              for (int q = 0; q < 4; q += 2) {
                const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
                const f32x2 tv = raw * gs + gb[j];
                acc[i][j][rg * 4 + q] = tv.x;
                acc[i][j][rg * 4 + q + 1] = tv.y;
                const ActCore2 c = act_core2(tv);
                const f32x2 s = kLn2 * c.mxt + c.dl;
                f32x2 pq = {pd[q], pd[q + 1]};
                pq = ka[j] * s + pq;
                pq = kc1[j] * c.r + pq;
                pd[q] = pq.x;
                pd[q + 1] = pq.y;
              }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float av = acc[i][j][rg * 4 + q] * gs + gb[j];
                acc[i][j][rg * 4 + q] = av;
                pd[q] += act_fwd<FAST>(av, alpha) * kov[j];
              }
            }
          }
          // row (i, rg, kg, q) of this wave's 64 x 32 partial-dot image (summed below by the
          // lane that owns the row: 4 ds_write_b32 here instead of 20 DPP adds + hazard nops)
          float* dst = s_dot + (i * 32 + 8 * rg + 4 * kg) * kRowDotPitch + frow;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q * kRowDotPitch] = pd[q];
          __builtin_amdgcn_sched_barrier(0);
        }
      // lane = row: sum its 32 partials (8 x ds_read_b128; the wave's own writes are in order)
      __builtin_amdgcn_wave_barrier();
      const f32x4* rp = reinterpret_cast<const f32x4*>(s_dot + lane * kRowDotPitch);
      f32x4 t4 = rp[0];
#pragma unroll
      for (int c = 1; c < 8; ++c) t4 += rp[c];
      s_part[(wr * 64 + lane) * WGN + wc] = ((t4.x + t4.y) + (t4.z + t4.w)) + ksum;
    }
    __syncthreads();
    BNF_MARK(ep, 3);
    {
      // waves 0 .. kBM/64-1: one row per lane
      float ll = 0.f, s_doutv = 0.f, s_dvsum = 0.f, s_par = 0.f, s_infl = 0.f;
      if (tid < kBM) {
        const int m = m0 + tid;
        float vsum = 0.f;
#pragma unroll
        for (int c = 0; c < WGN; ++c) vsum += s_part[tid * WGN + c];
        float dvv = 0.f;
        if (m < g.M) {
          const float gam_o = ep.scal[(int64_t)e * ep.scal_stride + BNF_MAX_LAYERS + 1];
          const float v = vsum * inv_sw + th[ep.off_bias_out];
          const float outv = gam_o * v;
          ep.out[(int64_t)e * ep.out_batch + m] = outv;
          // TAG 3: NORMAL only (the count-model code, lgamma / digamma, is compiled out)
          const RowLoss rl = row_loss_eval(TAG == 3 ? 0 : ep.obs, th, ep.off_lns, ep.off_shape, ep.off_infl,
                                           ep.ybat[(int64_t)e * ep.row_batch + m], outv, ep.lik_c);
          ll = rl.ll; s_par = rl.d_par; s_infl = rl.d_infl;
          s_doutv = rl.dout * v;
          dvv = gam_o * rl.dout;
          s_dvsum = dvv;
        }
        s_dv[tid] = dvv;
        const float t0 = wave_sum(ll), t1 = wave_sum(s_doutv), t2 = wave_sum(s_dvsum), t3 = wave_sum(s_par),
                    t4 = wave_sum(s_infl);
        if (lane == 0) {
          float* q = s_sc + wave * 5;
          q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; q[4] = t4;
        }
      }
    }
    __syncthreads();
    BNF_MARK(ep, 4);
    float* gr = ep.grad + (int64_t)e * ep.grad_stride;
    float dv_all = 0.f;   // thread 0: sum of d loss / d v over the tile's rows
    if (tid == 0) {
      float u[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w2 = 0; w2 < kBM / 64; ++w2)
#pragma unroll
        for (int i = 0; i < 5; ++i) u[i] += s_sc[w2 * 5 + i];
      dv_all = u[2];
      const float step_loss = -ep.lik_c * u[0];
      atomicAdd(&ep.loss[(int64_t)(e / ep.S) * ep.loss_stride + (ep.st ? ep.st->col : 0)], ep.loss_scale * step_loss);
      if (ep.loss_raw) atomicAdd(&ep.loss_raw[e], step_loss);
      atomicAdd(&gr[ep.off_os], sigmoidf(th[ep.off_os]) * u[1]);
      atomicAdd(&gr[ep.off_bias_out], u[2]);
      atomicAdd(&gr[ep.obs == BNF_OBS_NORMAL ? ep.off_lns : ep.off_shape], u[3]);
      if (ep.obs == BNF_OBS_ZINB) atomicAdd(&gr[ep.off_infl], u[4]);
    }
    // running sums, kept per column half j and per element parity (folded after the loop):
    //   sa = sum dv ediff, sg = sum p A, cp = sum p, ck = sum H dv   with p = dv act'(A)
    // so that  d alpha ~ kvn sa,  d gamma ~ kvn sg,  d bias = gamma kvn cp,  d k_o = ck / sqrt W
    f32x2 sa[2], sg[2], cp[2], ck[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) sa[j] = sg[j] = cp[j] = ck[j] = f32x2{0.f, 0.f};
    const float kvn[2] = {th[ep.off_ko + nw] * inv_sw, th[ep.off_ko + nw + 32] * inv_sw};
    const float gk[2] = {gamma * kvn[0], gamma * kvn[1]};
    // FAST: dZ = z = dv (gamma k_o / sqrt W) act'(A) formed directly (column factor folded into act''s
    // constants); sa = sum dv (elu - tanh + 2) (the "+ 2" leaves through thread 0 below), sg = sum z t,
    // cp = sum z (= d bias), ck = sum act(A) dv -- the row-panel kernel's scheme (bnf_panel.h)
    [[maybe_unused]] const float gka[2] = {gk[0] * ak.alpha, gk[1] * ak.alpha};
    [[maybe_unused]] const float gkc[2] = {gk[0] * ak.c2, gk[1] * ak.c2};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int lr = wr * 64 + 4 * kg + i * 32 + 8 * rg;
        const f32x4 dv4 = *reinterpret_cast<const f32x4*>(s_dv + lr);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int lc = wc * 64 + j * 32 + frow;
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            f32x2 av = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
            // (volatile asm statements keep their order: this ties the group's arithmetic to its
            // place between the fences; pure arithmetic would otherwise be emitted before them)
            asm volatile("" : "+v"(av));
            const f32x2 dv2 = {dv4[q], dv4[q + 1]};
            if constexpr (FAST) {
              const ActCore2 c = act_core2(av);     // av holds t = A log2(e)
              const f32x2 s = kLn2 * c.mxt + c.dl;
              const f32x2 h = ak.c1 * c.r + (ak.alpha * s + ak.c0);
              const f32x2 dg = gkc[j] * (c.r - c.r * c.r) + gka[j] * c.dl;
              const f32x2 z = dv2 * dg;
              sa[j] += dv2 * (2.f * c.r + s);
              sg[j] += z * av;
              cp[j] += z;
              ck[j] += h * dv2;
              store_pair_pk(tile + (lr + q) * kPitch + lc, tile + (lr + q + 1) * kPitch + lc, z.x, z.y);
            } else {
              const ActOut o0 = act_eval<FAST>(av.x, alpha), o1 = act_eval<FAST>(av.y, alpha);
              ActOut2 o;
              o.h = f32x2{o0.h, o1.h}; o.dact = f32x2{o0.dact, o1.dact}; o.ediff = f32x2{o0.ediff, o1.ediff};
              const f32x2 p = dv2 * o.dact;
              sa[j] += dv2 * o.ediff;
              sg[j] += p * av;
              cp[j] += p;
              ck[j] += o.h * dv2;
              const f32x2 z = gk[j] * p;
              store_pair(tile + (lr + q) * kPitch + lc, tile + (lr + q + 1) * kPitch + lc, z.x, z.y);
            }
          }
        }
        // pin the running sums here: otherwise the add chains (and everything feeding them)
        // are sunk below the barrier to their first use and every term is spilled
        asm volatile("" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sg[0]), "+v"(sg[1]), "+v"(cp[0]), "+v"(cp[1]),
                     "+v"(ck[0]), "+v"(ck[1]));
        __builtin_amdgcn_sched_barrier(0);
      }
    const float s_alpha = kvn[0] * (sa[0].x + sa[0].y) + kvn[1] * (sa[1].x + sa[1].y);
    const float s_gamma = FAST ? (kLn2 / gamma) * ((sg[0].x + sg[0].y) + (sg[1].x + sg[1].y))
                               : kvn[0] * (sg[0].x + sg[0].y) + kvn[1] * (sg[1].x + sg[1].y);
    const float cs_b[2] = {FAST ? cp[0].x + cp[0].y : gk[0] * (cp[0].x + cp[0].y),
                           FAST ? cp[1].x + cp[1].y : gk[1] * (cp[1].x + cp[1].y)};
    const float cs_k[2] = {ck[0].x + ck[0].y, ck[1].x + ck[1].y};
    __syncthreads();
    BNF_MARK(ep, 5);
    tile_to_global(reinterpret_cast<T*>(ep.out_h) + (int64_t)e * ep.act_batch, ep.ld);
    BNF_MARK(ep, 6);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float b = cs_b[j] + __shfl_xor(cs_b[j], 32, 64);
      const float k = cs_k[j] + __shfl_xor(cs_k[j], 32, 64);
      if (lane < 32) {
        s_col[wr * kBN + wc * 64 + j * 32 + lane] = b;
        s_col[(WGM + wr) * kBN + wc * 64 + j * 32 + lane] = k;
      }
    }
    const float wsa = wave_sum(s_alpha), wsg = wave_sum(s_gamma);
    if (lane == 0) {
      s_sc[16 + wave * 2] = wsa;
      s_sc[17 + wave * 2] = wsg;
    }
    __syncthreads();
    for (int c = tid; c < kBN; c += kThreads) {
      float b = 0.f, k = 0.f;
#pragma unroll
      for (int r = 0; r < WGM; ++r) {
        b += s_col[r * kBN + c];
        k += s_col[(WGM + r) * kBN + c];
      }
      atomicAdd(&gr[ep.off_bias + n0 + c], b);
      atomicAdd(&gr[ep.off_ko + n0 + c], k * inv_sw);
    }
    if (tid == 0) {
      float ta = 0.f, tg = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) {
        ta += s_sc[16 + w2 * 2];
        tg += s_sc[17 + w2 * 2];
      }
      if constexpr (FAST) {   // the "+ 2" of the sa sums: 2 (sum_c k_o / sqrt W) (sum_r dv); wave row 0 holds one column slab each
        float ko_all = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WGN; ++w2) ko_all += s_sc[16 + 2 * kWaves + w2];
        ta -= 2.f * inv_sw * ko_all * dv_all;
      }
      atomicAdd(&gr[ep.off_act_weight], alpha * (1.f - alpha) * ta);
      atomicAdd(&gr[ep.off_layer_scale], sigmoidf(th[ep.off_layer_scale]) * tg / gamma);
    }
  }
  BNF_MARK(ep, 7);
}

// ===========================================================================
// gemm_tn -- weight-gradient contraction on ROW-MAJOR operands:
//     C[i][j] = scale * sum_r A[r][i] * B[r][j]        (dK_l = H_l^T dZ_l / sqrt n_l)
// A (rows, a_ld) and B (rows, b_ld) are the activations exactly as the forward /
// backward kernels leave them (batch row = slow axis), so no transposed copies
// exist anywhere.  The contraction index r is the slow axis of both LDS tiles; the
// MFMA fragments (8 consecutive r for one column) come from the gfx950 transpose
// read ds_read_b64_tr_b16 (semantics measured with scripts/probes/tr_read_probe.hip:
// within each 16-lane group, lane i receives element (i % 4) of the 8-byte datum
// addressed by lane 4j + i/4, j = 0..3 -- i.e. a 4 x 16 block read by rows comes
// back by columns).  f32 operands use plain 4-byte reads (the f32 MFMA takes one
// value per lane).
// Tile: 128 (i) x 128 (j) per block, 4 waves (2 x 2), K tile = 128-byte... rows:
// kTnRows batch rows per stage, row pitch 256 B (bf16) / 512 B (f32), 64-byte
// segments XOR-swizzled with (row & 3) so the four rows of a transpose read fall
// on disjoint banks; filled by LDS-DMA with the swizzle on the source address.
// ===========================================================================
template <typename T>
struct TnCfg;
template <>
struct TnCfg<bf16_t> {
  static constexpr int kRows = 64;        // batch rows per K tile
};
template <>
struct TnCfg<float> {
  static constexpr int kRows = 32;
};
// LDS of a (64 WG)^2 tile: two stages of two operands, kRows rows of 64 WG columns each
template <typename T>
__host__ __device__ constexpr int gemm_tn_lds(int wg) {
  return 2 * 2 * TnCfg<T>::kRows * 64 * wg * (int)sizeof(T);
}

// WG x WG waves, each a 64 x 64 sub-tile: WG = 2 -> 128 x 128 (default), WG = 4 -> 256 x 256
// with 16 waves for the W x W weight gradients (W a multiple of 256, >= 512): this kernel is
// almost pure K loop and bound by the operand stream into LDS, which the large tile halves.
template <typename T, int TAG, int WG, bool SPLIT = false>
__global__ __launch_bounds__(64 * WG * WG, WG == 2 ? 2 : 4) void gemm_tn(const GemmArgs g, const EpiArgs ep) {
  using C_ = TnCfg<T>;
  constexpr int kBM = 64 * WG, kBN = 64 * WG, kWaves = WG * WG;
  constexpr int kRows = C_::kRows, kRB = kBM * (int)sizeof(T);   // row bytes of one operand tile
  constexpr int kOpBytes = kRows * kRB;       // per operand per stage
  constexpr int kStage = 2 * kOpBytes;
  constexpr int kChunksPerRow = kRB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WG, wc = wave % WG;

  const uint32_t per_member = (uint32_t)(g.tiles_m * g.tiles_n * g.splitk);
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = (int)(w / (uint32_t)tiles);
  w -= (uint32_t)split * tiles;
  const int tm = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int m0 = tm * kBM, n0 = tn * kBN;   // m: columns of A (i), n: columns of B (j)

  const int nk_total = g.K / kRows;           // g.K = padded batch rows, multiple of 64
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)e * g.a_batch * Elem<T>::kBytes;
  const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)e * g.b_batch * Elem<T>::kBytes;

  // ---- LDS-DMA staging: one wave instruction = 1 KiB = 1024/kRB rows ---------------
  constexpr int kInstrPerOp = kOpBytes / 1024;          // per operand per tile
  constexpr int kPerWave = kInstrPerOp / kWaves;
  constexpr int kRowsPerInstr = 1024 / kRB;
  static_assert(kPerWave >= 1 && kPerWave * kWaves == kInstrPerOp && kRowsPerInstr >= 1, "staging map");
  int src_off_a[kPerWave], src_off_b[kPerWave], lds_base[kPerWave];
  // valid bytes of a tile row, from the (16-byte aligned, zero padded) leading dimensions
  const int a_cols_bytes = min(g.a_ld - m0, kBM) * Elem<T>::kBytes;
  const int b_cols_bytes = min(g.b_ld - n0, kBN) * Elem<T>::kBytes;
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int q = wave * kPerWave + i;                  // instruction index
    const int row = q * kRowsPerInstr + lane / kChunksPerRow;
    const int cp = lane % kChunksPerRow;                // physical 16-byte chunk
    const int c = cp ^ ((row & 3) << 2);                // logical chunk: 64-byte segment ^= row & 3
    // columns beyond the matrix edge are clamped to the last valid 16-byte chunk
    // (their products land in masked output rows / columns)
    const int ca = min(c * 16, max(a_cols_bytes - 16, 0));
    const int cb = min(c * 16, max(b_cols_bytes - 16, 0));
    src_off_a[i] = row * g.a_ld * Elem<T>::kBytes + m0 * Elem<T>::kBytes + ca;
    src_off_b[i] = row * g.b_ld * Elem<T>::kBytes + n0 * Elem<T>::kBytes + cb;
    lds_base[i] = q * 1024;
  }
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void glb_void_t;
  auto stage = [&](int buf, int kt) {
    const int64_t ra = (int64_t)kt * kRows * g.a_ld * Elem<T>::kBytes;
    const int64_t rb = (int64_t)kt * kRows * g.b_ld * Elem<T>::kBytes;
    char* sA = smem + buf * kStage;
    char* sB = sA + kOpBytes;
    // scalar base (pinned in SGPRs, see gemm_nt) + 32-bit lane offset
    auto pin = [](const char* p) {
      const uint64_t b = (uint64_t)p;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
      return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* pa = pin(Ab + ra);
    const char* pb = pin(Bb + rb);
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(pa + (uint32_t)src_off_a[i]), (lds_void_t*)(sA + lds_base[i]), 16, 0, BNF_TN_AUX);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(pb + (uint32_t)src_off_b[i]), (lds_void_t*)(sB + lds_base[i]), 16, 0, BNF_TN_AUX);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, kg = lane >> 5;

  if (kt0 < kt1) {
    stage(0, kt0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      if (kt + 1 < kt1) stage(buf ^ 1, kt + 1);
      const char* sA = smem + buf * kStage;
      const char* sB = sA + kOpBytes;
      if constexpr (Elem<T>::kBytes == 2) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4 lds_v4;
        const int p = lane & 15, half = (lane >> 4) & 1;
        // per lane: row offset within a 4-row group and column (elements) inside a 32-column tile
        const int prow = p >> 2, pcol = half * 16 + (p & 3) * 4;
#pragma unroll
        for (int ks = 0; ks < kRows / 16; ++ks) {
          // fragments are assembled as raw dwords (two 64-bit transpose reads each) and
          // bit-cast once: element-wise inserts into a __bf16 vector miscompile here
          uint2 ra_[2][2], rb_[2][2];   // [t][i]
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = ks * 16 + kg * 8 + t * 4 + prow;
            const int rsw = (row & 3) << 6;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int ba = (wr * 64 + i * 32 + pcol) * 2, bb = (wc * 64 + i * 32 + pcol) * 2;
              const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (lds_v4*)(const_cast<char*>(sA) + row * kRB + ((ba & ~63) ^ rsw) + (ba & 63)));
              const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (lds_v4*)(const_cast<char*>(sB) + row * kRB + ((bb & ~63) ^ rsw) + (bb & 63)));
              ra_[t][i] = __builtin_bit_cast(uint2, va);
              rb_[t][i] = __builtin_bit_cast(uint2, vb);
            }
          }
          bf16x8 fa[2], fb[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const u32x4 wa = {ra_[0][i].x, ra_[0][i].y, ra_[1][i].x, ra_[1][i].y};
            const u32x4 wb = {rb_[0][i].x, rb_[0][i].y, rb_[1][i].x, rb_[1][i].y};
            fa[i] = __builtin_bit_cast(bf16x8, wa);
            fb[i] = __builtin_bit_cast(bf16x8, wb);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      } else {
        // f32: MFMA step s of group ks contracts rows ks*16 + s (lanes 0-31) and ks*16 + 8 + s (32-63)
#pragma unroll
        for (int ks = 0; ks < kRows / 16; ++ks) {
          if constexpr (SPLIT) {
            // BNF_DTYPE_F32S: the lane's 8 batch rows of one column = the bf16 32x32x16 fragment: gather, split,
            // three bf16 MFMAs per fragment pair (Mma<float>::mma_split)
            typename Mma<float>::Frag ga[2], gb[2];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
              const int row = ks * 16 + kg * 8 + s8;
              const int rsw = (row & 3) << 6;
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int ba = (wr * 64 + i * 32 + frow) * 4, bb = (wc * 64 + i * 32 + frow) * 4;
                const float va = *reinterpret_cast<const float*>(sA + row * kRB + ((ba & ~63) ^ rsw) + (ba & 63));
                const float vb = *reinterpret_cast<const float*>(sB + row * kRB + ((bb & ~63) ^ rsw) + (bb & 63));
                if (s8 < 4) { ga[i].lo[s8] = va; gb[i].lo[s8] = vb; } else { ga[i].hi[s8 - 4] = va; gb[i].hi[s8 - 4] = vb; }
              }
            }
            typename Mma<float>::Split sa[2], sb2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              sa[i] = Mma<float>::split_plain(ga[i]);
              sb2[i] = Mma<float>::split_plain(gb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) Mma<float>::mma_split(acc[i][j], sa[i], sb2[j]);
            continue;
          }
#pragma unroll
          for (int s8 = 0; s8 < 8; ++s8) {
            const int row = ks * 16 + kg * 8 + s8;
            const int rsw = (row & 3) << 6;
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int ba = (wr * 64 + i * 32 + frow) * 4, bb = (wc * 64 + i * 32 + frow) * 4;
              fa[i] = *reinterpret_cast<const float*>(sA + row * kRB + ((ba & ~63) ^ rsw) + (ba & 63));
              fb[i] = *reinterpret_cast<const float*>(sB + row * kRB + ((bb & ~63) ^ rsw) + (bb & 63));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: scaled f32 store / atomic accumulate into the gradient vector ----
  const int mw = m0 + wr * 64 + 4 * kg;
  const int nw = n0 + wc * 64 + frow;
  float* out = ep.out_f32 ? ep.out_f32 + (int64_t)e * ep.f32_batch
                          : ep.grad + (int64_t)e * ep.grad_stride + ep.off_out;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nw + j * 32;
    if (n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
        if (m < g.M) {
          const float v = acc[i][j][r] * ep.scale;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        } else if (ep.bias_row && m == g.M) {
          atomicAdd(&ep.grad[(int64_t)e * ep.grad_stride + ep.off_bias_row + n], acc[i][j][r]);
        }
      }
  }
}

// ===========================================================================
// gemm_tn_skinny -- the layer-0 weight gradient dK_0 = H0^T dZ_0 / sqrt F when the feature
// panel is 64 columns wide (Fp = 64) and W is a multiple of 512: 2 flops per byte of dZ_0, i.e. a
// pure HBM stream.  gemm_tn's 128 x 128 tile reads quarter rows of dZ_0 (256 bytes) one stage
// ahead -- ~40 KiB in flight per CU in bursts, 3.9 TB/s at C2.  Here one 8-wave workgroup per CU
// owns 64 x 512 outputs (whole 1 KiB rows of dZ_0, H0 read once), K advances through a ring of
// FOUR 32-row stages (36 KiB each) filled three stages ahead by LDS-DMA with counted
// `s_waitcnt vmcnt` and one raw s_barrier per stage (gemm_nt's scheme): ~108 KiB in flight per
// CU all the time.  Fragments by ds_read_b64_tr_b16 exactly as in gemm_tn (64-byte segments of a
// row XOR-swizzled with the row index: two segments for H0's 128-byte rows, four for dZ_0's).
// ===========================================================================
// Transpose reads issued as inline asm: hipcc treats the ds_read_tr builtin as an LDS load that may
// alias every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of it -- which turns a ring of
// prefetched stages into load -> wait -> compute (seen in the ISA; plain ds_read_b128 loads in gemm_nt
// do not get that wait).  The caller orders them itself: counted vmcnt + s_barrier before, lds_tr_fence after.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#ifndef BNF_TN_BUFDMA
#define BNF_TN_BUFDMA 1   // LDS-DMA of the ring / skinny weight-gradient kernels as raw BUFFER loads (round 4, r04f)
#endif
// One 1 KiB wave load global -> LDS (16 bytes per lane) from the wave-uniform address `p` + the lane's 32-bit offset.
// BNF_TN_BUFDMA: `buffer_load_dwordx4 v_off, s[rsrc] ... lds` -- the uniform pointer becomes the resource's base (SALU), the
// lane offset the VGPR offset: no VALU instruction per load (the global_load_lds form took a 64-bit v_lshl_add_u64 each;
// VALU issue slots between MFMAs are what these loops are short of, see profiles/r04_panel_ab.md r04d).
template <int AUX>
__device__ __forceinline__ void dma_1k(const char* p, uint32_t lane_off, char* lds_dst) {
  typedef __attribute__((address_space(3))) void lds_void_t;
#if BNF_TN_BUFDMA
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_dst, 16, lane_off, 0, 0, AUX);
#else
  typedef __attribute__((address_space(1))) const void glb_void_t;
  __builtin_amdgcn_global_load_lds((glb_void_t*)(p + lane_off), (lds_void_t*)lds_dst, 16, 0, AUX);
#endif
}
template <int OFF>
__device__ __forceinline__ u32x2_t lds_tr16_b64(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_tr_fence(u32x2_t (&a)[2][2], u32x2_t (&b)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]),
                 "+v"(b[1][0]), "+v"(b[1][1]));
}
constexpr int kSkRows = 32, kSkStages = 4, kSkA = kSkRows * 128, kSkB = kSkRows * 1024, kSkStage = kSkA + kSkB;
constexpr int kSkLds = kSkStages * kSkStage;
__global__ __launch_bounds__(512, 2) void gemm_tn_skinny(const GemmArgs g, const EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const uint32_t per_member = (uint32_t)(g.tiles_n * g.splitk);
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int split = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int n0 = tn * 512;

  const int nk_total = g.K / kSkRows;         // g.K = padded batch rows, a multiple of 64
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)e * g.a_batch * 2;
  const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)e * g.b_batch * 2 + n0 * 2;

  // ---- LDS-DMA staging: per stage and wave 1 instruction of H0 (8 rows of 128 bytes; waves 4-7
  // repeat those of waves 0-3, which keeps the vmcnt bookkeeping uniform) + 4 rows of dZ_0
  const int qa = wave & 3;
  const int a_row = qa * 8 + (lane >> 3);
  const uint32_t src_a = (uint32_t)a_row * 128u + (uint32_t)(((lane & 7) ^ ((a_row & 1) << 2)) * 16);
  uint32_t src_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)   // row wave * 4 + i: (row & 3) == i
    src_b[i] = (uint32_t)(wave * 4 + i) * (uint32_t)g.b_ld * 2u + (uint32_t)((lane ^ (i << 2)) * 16);
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void glb_void_t;
  auto pin = [](const char* p) {
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * kSkStage;
    char* sB = sA + kSkA;
    const char* pa = pin(Ab + (int64_t)kt * kSkRows * 128);
    const char* pb = pin(Bb + (int64_t)kt * kSkRows * g.b_ld * 2);
    dma_1k<BNF_SK_AUX>(pa, src_a, sA + qa * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_1k<BNF_SK_AUX>(pb, src_b[i], sB + (wave * 4 + i) * 1024);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, kg = lane >> 5;
  const int p = lane & 15, half = (lane >> 4) & 1;
  const int prow = p >> 2, pcol = half * 16 + (p & 3) * 4;
  // per-lane byte offsets of the transpose reads inside a stage, k step 0: [t][i]
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
  uint32_t off_a[2][2], off_b[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = kg * 8 + t * 4 + prow;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ba = (i * 32 + pcol) * 2, bb = (wave * 64 + i * 32 + pcol) * 2;
      off_a[t][i] = lds0 + (uint32_t)(row * 128 + ((ba & ~63) ^ ((row & 1) << 6)) + (ba & 63));
      off_b[t][i] = lds0 + (uint32_t)(kSkA + row * 1024 + ((bb & ~63) ^ ((row & 3) << 6)) + (bb & 63));
    }
  }

  constexpr int kPerWave = 5;
  constexpr int kAhead = (kSkStages - 2) * kPerWave;            // this wave's younger DMA instructions
  constexpr int kWait = (kAhead & 15) | ((kAhead >> 4) << 14) | 0x0F70;
  constexpr int kWaitAll = 0x0F70;
#pragma unroll
  for (int s = 0; s < kSkStages - 1; ++s)
    if (kt0 + s < kt1) stage(s, kt0 + s);
  for (int ktb = kt0; ktb < kt1; ktb += kSkStages) {
#pragma unroll
    for (int sb = 0; sb < kSkStages; ++sb) {
      const int kt = ktb + sb;
      if (kt >= kt1) break;
      if (kt + kSkStages - 2 >= kt1) __builtin_amdgcn_s_waitcnt(kWaitAll);
      else __builtin_amdgcn_s_waitcnt(kWait);
      __builtin_amdgcn_s_barrier();
      if (kt + kSkStages - 1 < kt1) stage((sb + kSkStages - 1) % kSkStages, kt + kSkStages - 1);
      u32x2_t ra_[2][2][2], rb_[2][2][2];   // [ks][t][i]
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t aa = off_a[t][i] + sb * kSkStage, ab = off_b[t][i] + sb * kSkStage;
          ra_[0][t][i] = lds_tr16_b64<0>(aa);
          rb_[0][t][i] = lds_tr16_b64<0>(ab);
          ra_[1][t][i] = lds_tr16_b64<16 * 128>(aa);
          rb_[1][t][i] = lds_tr16_b64<16 * 1024>(ab);
        }
#pragma unroll
      for (int ks = 0; ks < kSkRows / 16; ++ks) {
        if (ks == 0) lds_tr_fence(ra_[0], rb_[0]);     // (both k steps were issued above: one wait covers them)
        else asm volatile("" : "+v"(ra_[1][0][0]), "+v"(ra_[1][0][1]), "+v"(ra_[1][1][0]), "+v"(ra_[1][1][1]),
                               "+v"(rb_[1][0][0]), "+v"(rb_[1][0][1]), "+v"(rb_[1][1][0]), "+v"(rb_[1][1][1]));
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 wa = {ra_[ks][0][i].x, ra_[ks][0][i].y, ra_[ks][1][i].x, ra_[ks][1][i].y};
          const u32x4 wb = {rb_[ks][0][i].x, rb_[ks][0][i].y, rb_[ks][1][i].x, rb_[ks][1][i].y};
          fa[i] = __builtin_bit_cast(bf16x8, wa);
          fb[i] = __builtin_bit_cast(bf16x8, wb);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: scaled f32 store / atomic accumulate into the gradient vector ----
  const int mw = 4 * kg;
  const int nw = n0 + wave * 64 + frow;
  float* out = ep.out_f32 ? ep.out_f32 + (int64_t)e * ep.f32_batch
                          : ep.grad + (int64_t)e * ep.grad_stride + ep.off_out;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nw + j * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
        if (m < g.M) {
          const float v = acc[i][j][r] * ep.scale;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        } else if (ep.bias_row && m == g.M) {
          atomicAdd(&ep.grad[(int64_t)e * ep.grad_stride + ep.off_bias_row + n], acc[i][j][r]);
        }
      }
  }
}


// ===========================================================================
// gemm_tn_ring -- gemm_tn's 256 x 256 bf16 tile with the K loop as a RING: four 32-row stages
// (32 KiB each), filled three stages ahead by LDS-DMA, counted `s_waitcnt vmcnt` and one raw
// s_barrier per stage, transpose reads as inline asm (lds_tr16_b64: the builtin gets a
// compiler-inserted vmcnt(0) that serialises any ring).  gemm_tn's two 64-row stages under
// __syncthreads leave the memory pipe idle between "stage landed" and "next stage issued": at C2 a
// stage took 5.1k cycles for 2.0k cycles of MFMA work.
// EIGHT waves of 128 x 64 (4 x 2 accumulators, two waves per SIMD, 212 registers): 12 transpose reads
// per 8 MFMAs (sixteen waves of 64 x 64 need 8 per 4) and a barrier among 8 waves -- 350 -> 340 us at C2.
// Software pipeline: the barrier of iteration kt certifies stage kt + 1 (landed for every wave) and
// frees the buffer of stage kt - 1 for the prefetch of stage kt + 3.  The fragments of stage kt were
// read during iteration kt - 1; those of stage kt + 1 are requested BETWEEN the two MFMA groups of
// stage kt, each fragment set refilled right after the MFMAs that consumed it were issued, so that
// the LDS serves them while the matrix pipe works (ablation of the read-then-multiply version:
// 172 us of a 295 us compute-only run were the reads and the barrier; MFMA floor 164 us).
// ===========================================================================
constexpr int kRgRows = 32, kRgStages = 4, kRgOp = kRgRows * 512, kRgStage = 2 * kRgOp;
constexpr int kRgLds = kRgStages * kRgStage;
// SWAP (split-K = 1, plain stores): the MFMA operands in swapped roles, see mma_k
// NW = 4 (not instantiated: measured slower): FOUR waves of 128 x 128 (4 x 4 accumulators pinned in the 256 accumulation
// registers, one wave per SIMD): 16 transpose reads per 16 MFMAs instead of 12 per 8 -- a third less LDS traffic per MFMA,
// clean loop (32 MFMAs, 32 reads, one barrier, no moves, no scratch), parity green, and 419 us against 376 - 384 us at C2,
// C3/8 6,380 against 6,460, C4/8 677 against 692 (gpurun_out/r03af): with one wave per SIMD nothing hides a wave's own
// LDS and barrier latencies.
template <int TAG, bool SWAP, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void gemm_tn_ring(const GemmArgs g, const EpiArgs ep) {
  constexpr int TJ = NW == 8 ? 2 : 4;        // 32-column accumulator tiles per wave (x 4 row tiles of 32)
  constexpr int QN = 16 / NW;                // 1 KiB wave-loads per operand, stage and wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / (NW / 2), wc = wave % (NW / 2);

  const int n_multi = g.n_multi > 1 ? g.n_multi : 1;
  const uint32_t per_layer = (uint32_t)(g.tiles_m * g.tiles_n * g.splitk);
  const uint32_t per_member = per_layer * (uint32_t)n_multi;
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int which = (int)(w / per_layer);      // which of the n_multi contractions
  w -= (uint32_t)which * per_layer;
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = (int)(w / (uint32_t)tiles);
  w -= (uint32_t)split * tiles;
  const int tm = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int m0 = tm * 256, n0 = tn * 256;
  // (selected by uniform compares: a dynamically indexed by-value argument array is copied to scratch)
  const void* gA = g.A;
  const void* gB = g.B;
  int32_t off_out = ep.off_out;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (g.n_multi > 1 && which == k) { gA = g.A_multi[k]; gB = g.B_multi[k]; off_out = g.off_out_multi[k]; }

  const int nk_total = g.K / kRgRows;
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(gA) + (int64_t)e * g.a_batch * 2 + m0 * 2;
  const char* Bb = reinterpret_cast<const char*>(gB) + (int64_t)e * g.b_batch * 2 + n0 * 2;

  uint32_t src_a[QN], src_b[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int srow = (wave * QN + q) * 2 + (lane >> 5);
    const uint32_t schunk = (uint32_t)(((lane & 31) ^ ((srow & 3) << 2)) * 16);
    src_a[q] = (uint32_t)srow * (uint32_t)g.a_ld * 2u + schunk;
    src_b[q] = (uint32_t)srow * (uint32_t)g.b_ld * 2u + schunk;
  }
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef __attribute__((address_space(1))) const void glb_void_t;
  auto pin = [](const char* p) {
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * kRgStage;
    char* sB = sA + kRgOp;
    const char* pa = pin(Ab + (int64_t)kt * kRgRows * g.a_ld * 2);
    const char* pb = pin(Bb + (int64_t)kt * kRgRows * g.b_ld * 2);
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      dma_1k<BNF_TN_AUX>(pa, src_a[q], sA + (wave * QN + q) * 1024);
      dma_1k<BNF_TN_AUX>(pb, src_b[q], sB + (wave * QN + q) * 1024);
    }
  };

  f32x16 acc[4][TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, kg = lane >> 5;
  const int p = lane & 15, half = (lane >> 4) & 1;
  const int prow = p >> 2, pcol = half * 16 + (p & 3) * 4;
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
  // byte offsets of the transpose reads in a stage for t = 0, k step 0 (t = 1: + 4 rows, k step 1: + 16 rows)
  uint32_t off_a[4], off_b[TJ];
  {
    const int row = kg * 8 + prow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ba = (wr * 128 + i * 32 + pcol) * 2;
      off_a[i] = lds0 + (uint32_t)(row * 512 + ((ba & ~63) ^ ((row & 3) << 6)) + (ba & 63));
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int bb = (wc * (32 * TJ) + j * 32 + pcol) * 2;
      off_b[j] = lds0 + (uint32_t)(kRgOp + row * 512 + ((bb & ~63) ^ ((row & 3) << 6)) + (bb & 63));
    }
  }
  struct Frags {
    u32x2_t a[2][4], b[2][TJ];   // [t][tile]
  };
  auto read_k = [&](Frags& f, int sb, auto ks_tag) {
    constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.a[0][i] = lds_tr16_b64<ks * 16 * 512>(off_a[i] + sb * kRgStage);
      f.a[1][i] = lds_tr16_b64<ks * 16 * 512 + 4 * 512>(off_a[i] + sb * kRgStage);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      f.b[0][j] = lds_tr16_b64<ks * 16 * 512>(off_b[j] + sb * kRgStage);
      f.b[1][j] = lds_tr16_b64<ks * 16 * 512 + 4 * 512>(off_b[j] + sb * kRgStage);
    }
  };
  auto fence = [&](Frags& f) {
    if constexpr (TJ == 4)
      asm volatile("" : "+v"(f.b[0][2]), "+v"(f.b[0][3]), "+v"(f.b[1][2]), "+v"(f.b[1][3]));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[0][3]), "+v"(f.a[1][0]), "+v"(f.a[1][1]),
                   "+v"(f.a[1][2]), "+v"(f.a[1][3]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]));
    if constexpr (TJ == 4)
      asm volatile("" : "+v"(f.b[0][2]), "+v"(f.b[0][3]), "+v"(f.b[1][2]), "+v"(f.b[1][3]));
  };
  auto touch = [&](Frags& f) {
    asm volatile("" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[0][3]), "+v"(f.a[1][0]), "+v"(f.a[1][1]),
                      "+v"(f.a[1][2]), "+v"(f.a[1][3]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]));
    if constexpr (TJ == 4)
      asm volatile("" : "+v"(f.b[0][2]), "+v"(f.b[0][3]), "+v"(f.b[1][2]), "+v"(f.b[1][3]));
  };
  auto mma_k = [&](const Frags& f) {
    bf16x8 fa[4], fb[TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 wa = {f.a[0][i].x, f.a[0][i].y, f.a[1][i].x, f.a[1][i].y};
      fa[i] = __builtin_bit_cast(bf16x8, wa);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const u32x4 wb = {f.b[0][j].x, f.b[0][j].y, f.b[1][j].x, f.b[1][j].y};
      fb[j] = __builtin_bit_cast(bf16x8, wb);
    }
    // SWAP: operands in swapped roles -- the accumulator tile is (n x m), i.e. a lane holds four consecutive n of ONE
    // m = 16 contiguous bytes of the row-major (m, n) output: the epilogue stores 32 dwordx4 per wave instead of 128
    // dwords (C3/8 +0.8 %, C2 +0.4 % on the step).  Not with split K: a lane's four atomics would go to consecutive
    // addresses and a wave instruction to 64 different cache lines (C5/8 -8 %: measured, gpurun_out/r03ac).
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if constexpr (NW == 4) {   // all sixteen accumulators in the accumulation registers here: nothing else may live there
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

  constexpr int kPerWave = 2 * QN;          // LDS-DMA loads of one stage per wave
  auto vm = [](int n) constexpr { return (n & 15) | ((n >> 4) << 14) | 0x0F70; };   // s_waitcnt vmcnt(n) only
  constexpr int kWait1 = vm(kPerWave);
  constexpr int kWaitAll = 0x0F70;
#pragma unroll
  for (int s = 0; s < kRgStages - 1; ++s)
    if (kt0 + s < kt1) stage(s, kt0 + s);
  if (kt0 + 1 < kt1) {
    if (kt0 + 2 < kt1) __builtin_amdgcn_s_waitcnt(vm(2 * kPerWave));
    else __builtin_amdgcn_s_waitcnt(kWait1);
  } else {
    __builtin_amdgcn_s_waitcnt(kWaitAll);
  }
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
  if (kt0 < kt1) {
    read_k(f0, 0, K0{});
    read_k(f1, 0, K1{});
  }
  for (int ktb = kt0; ktb < kt1; ktb += kRgStages) {
#pragma unroll(NW == 8 ? kRgStages : 1)
    for (int sb = 0; sb < kRgStages; ++sb) {
      const int kt = ktb + sb;
      if (kt >= kt1) break;
      if (kt + 2 < kt1) __builtin_amdgcn_s_waitcnt(kWait1);
      else __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_s_barrier();
      if (kt + kRgStages - 1 < kt1) stage((sb + kRgStages - 1) % kRgStages, kt + kRgStages - 1);
      fence(f0);
      touch(f1);
      const bool more = kt + 1 < kt1;
      const int sn = (sb + 1) % kRgStages;
      mma_k(f0);
      if (more) read_k(f0, sn, K0{});
      mma_k(f1);
      if (more) read_k(f1, sn, K1{});
    }
  }

  float* out = ep.out_f32 ? ep.out_f32 + (int64_t)e * ep.f32_batch
                          : ep.grad + (int64_t)e * ep.grad_stride + off_out;
  if constexpr (SWAP) {
    const int mw = m0 + wr * 128 + frow;
    const int nw = n0 + wc * (32 * TJ) + 4 * kg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* orow = out + (int64_t)(mw + i * 32) * ep.ld_f32 + nw;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float v[4] = {acc[i][j][rg * 4] * ep.scale, acc[i][j][rg * 4 + 1] * ep.scale,
                              acc[i][j][rg * 4 + 2] * ep.scale, acc[i][j][rg * 4 + 3] * ep.scale};
          store4u(orow + j * 32 + 8 * rg, 4, v);     // (the gradient leaf starts at any 4-byte aligned offset)
        }
    }
  } else {
    const int mw = m0 + wr * 128 + 4 * kg;
    const int nw = n0 + wc * (32 * TJ) + frow;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int n = nw + j * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
          const float v = acc[i][j][r] * ep.scale;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        }
    }
  }
}


}  // namespace bnf
