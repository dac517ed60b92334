// bnf_gemm8.h -- the weight-gradient contractions on FP8 operand copies (round 5: compute_dtype 'fp8' = bf16
// forward / backward-data contractions + fp8 OPERAND STORAGE for the weight-gradient streams).
//
//     dK_l[i][j] = s_H s_dZ / sqrt n_l * sum_r Hq_l[r][i] dZq_l[r][j]
//
// Hq_l: the activations H_l as OCP e4m3 (what the panel kernel / k_featurize wrote next to -- instead of -- the bf16
// copies), dZq_l: the backward signals as OCP e5m2 divided by a per-member power of two (EpiArgs.qscale: the product
// s_H s_dZ the epilogue folds back in).  Why: these kernels are HBM streams of exactly those copies (C5/8: 5.96 and
// 6.24 TB/s, 22 % of the step; C2: the ring kernel at 4.5 TB/s bound by neither pipe) -- half the bytes, and HALF THE LDS
// READS PER MFMA: ds_read_b64_tr_b8 delivers a whole 8-deep fragment (one instruction where the bf16 kernels issue two
// ds_read_b64_tr_b16), which is what bounds gemm_tn_ring (profiles/r02w_wgrad_streams.md: reads + barriers do not
// overlap its MFMAs).  The matrix instruction is the non-scaled v_mfma_f32_32x32x16_fp8_bf8 (bf16 rate, K = 16, f32
// accumulate): same tile shapes, same accumulator layout, same epilogues as the bf16 kernels of bnf_gemm.h -- and, in
// gemm_tn_ring8 (the one of the three that was bound by the matrix pipe), the block-scaled K = 64 form at unit scales,
// v_mfma_scale_f32_32x32x64_f8f6f4: twice the rate, the same products (mfma8x64 below; r05v: 226 -> 175 us at C2).
//
// Measured semantics (scripts/probes/fp8_probe.hip, gpurun_out/r05h/fp8_probe.txt):
//   * ds_read_b64_tr_b8: lane i of a 16-lane group receives, for j = 0 .. 7, byte (i % 8) of the 8-byte datum addressed
//     by lane 2 j + i / 8 of the group -- with lane p addressing row base + p / 2, bytes 8 (p % 2) .. + 7, lane i ends up
//     with rows base .. base + 7 of byte column i: an 8 x 16 block read by rows comes back by columns;
//   * the MFMA operand of lane (m = l % 32, kg = l / 32) is k = 8 kg + byte index (64-bit operand), like bf16's.
// Staging: K advances through LDS-DMA stages of 64 batch rows; rows are split in 32-byte chunks XOR-swizzled with the
// row index on the SOURCE address so that the 8 rows x 32 bytes a half wave reads with one transpose read fall on 64
// distinct banks (256-byte rows: chunk ^ (row % 8); 128-byte rows: chunk ^ (row / 2 % 4); 64-byte rows: chunk ^ (row / 4 % 2)).
#pragma once

#include "bnf_gemm.h"

namespace bnf {

template <int OFF>
__device__ __forceinline__ u32x2_t lds_tr8_b64(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ long frag8(const u32x2_t& v) { return __builtin_bit_cast(long, v); }
// A = e4m3 (activations), B = e5m2 (backward signals); SWAP: operand roles exchanged (transposed accumulator tile)
template <bool SWAP>
__device__ __forceinline__ f32x16 mfma8(const u32x2_t& h, const u32x2_t& dz, const f32x16& c) {
  if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(frag8(dz), frag8(h), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_fp8_bf8(frag8(h), frag8(dz), c, 0, 0, 0);
}
// The block-scaled form at unit scales: v_mfma_scale_f32_32x32x64_f8f6f4 contracts 64 k per instruction at TWICE the
// rate of the non-scaled K = 16 form (64 against 4 x 32 matrix-pipe cycles; MI355X_MICROARCH.md: the only fp8 MFMA above
// the bf16 rate), formats per operand (0 = e4m3, 1 = e5m2), scales E8M0 127 = 2^0.  A lane's 32 operand bytes are the four
// transpose-read results of a 64-row stage back to back: lane (m, kg) byte 8 ks + b  <->  k = 16 ks + 8 kg + b for BOTH
// operands, and the instruction pairs byte (kg, idx) of A with byte (kg, idx) of B, so the sum is the same set of products.
#ifndef BNF_RING8_X64
#define BNF_RING8_X64 1
#endif
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <bool SWAP>
__device__ __forceinline__ f32x16 mfma8x64(const i32x8& h, const i32x8& dz, const f32x16& c) {
  if constexpr (SWAP) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(dz, h, c, 1, 0, 0, 127, 0, 127);
  else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(h, dz, c, 0, 1, 0, 127, 0, 127);
}
// LDS-DMA of one KiB per wave as inline assembly: the x64 loop below reads LDS through the compiler's own
// ds_read_b64_tr_b8 builtin, and behind a builtin buffer_load ... lds the compiler puts s_waitcnt vmcnt(0) in front of
// every LDS read that MAY alias its destination -- i.e. it drains the ring each stage.  The stage buffers a read and a
// DMA in flight touch are different by construction (what the barriers order); assembly keeps the DMA out of that model.
// lds_addr: wave-uniform LDS byte address (M0 is a reserved register: set here, used by nothing else in these kernels).
__device__ __forceinline__ void dma_1k_asm(const char* p, uint32_t lane_off, uint32_t lds_addr) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, 0x7fffffff, 0x00020000);
  // (clobbers: M0 -- so the compiler never keeps a value of its own there across this statement -- and memory: the DMA
  // writes LDS behind the compiler's back; the explicit vmcnt waits and barriers of the callers stay what orders it)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(lane_off), "s"(r), "s"(lds_addr) : "m0", "memory");
}
// per-lane constants of a transpose read: lane = 32 kg + 16 half + p, p = 2 j + q
struct Tr8Lane {
  int j, q, half, kg;
};
__device__ __forceinline__ Tr8Lane tr8_lane(int lane) {
  const int p = lane & 15;
  return Tr8Lane{p >> 1, p & 1, (lane >> 4) & 1, lane >> 5};
}

// f32 -> fp8 for the debug entry points (saturating): rows x cols, leading dimension ld (bytes = elements)
template <bool BF8>
__global__ void k_from_f32_q8(const float* __restrict__ src, int64_t rows, int32_t cols, uint8_t* __restrict__ dst, int32_t ld) {
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);   // MODE.FP16_OVFL: conversions clamp to the largest finite value
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const int64_t r = i / ld;
  const int c = (int)(i % ld);
  const float v = c < cols ? src[r * cols + c] : 0.f;
  const int pk = BF8 ? __builtin_amdgcn_cvt_pk_bf8_f32(v, 0.f, 0, false) : __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
  dst[i] = (uint8_t)(pk & 0xff);
}

// fp8 copy (rows, ld bytes) of one member per blockIdx.y -> f32 (members, rows, cols): bnf_debug_activation
__global__ void k_q8_to_f32(const uint8_t* __restrict__ src, int64_t batch, int32_t ld, int64_t rows, int32_t cols,
                            float* __restrict__ out, int32_t bf8, const float* __restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int e = blockIdx.y;
  const int64_t r = i / cols;
  const int c = (int)(i % cols);
  const uint32_t b = src[(int64_t)e * batch + r * ld + c];
  const float v = bf8 ? __builtin_amdgcn_cvt_f32_bf8((int)b, 0) : __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
  out[(int64_t)e * rows * cols + i] = v * (scale ? scale[e] : 1.f);
}

// ===========================================================================
// gemm_tn8 -- the generic 128 x 128 tile (4 waves of 64 x 64, two 64-row stages under __syncthreads): every shape
// the two stream kernels below do not take (C5's layer 0: F = 105 -> 128 padded features x W = 256)
// ===========================================================================
constexpr int kTn8Rows = 64, kTn8Op = kTn8Rows * 128, kTn8Stage = 2 * kTn8Op, kTn8Lds = 2 * kTn8Stage;
template <int TAG>
__global__ __launch_bounds__(256, 2) void gemm_tn8(const GemmArgs g, const EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const uint32_t per_member = (uint32_t)(g.tiles_m * g.tiles_n * g.splitk);
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = (int)(w / (uint32_t)tiles);
  w -= (uint32_t)split * tiles;
  const int tm = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int m0 = tm * 128, n0 = tn * 128;

  const int nk_total = g.K / kTn8Rows;
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)e * g.a_batch;
  const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)e * g.b_batch;

  // staging: one wave instruction = 1 KiB = 8 rows of 128 bytes; 8 instructions per operand and stage, 2 per wave
  int src_a[2], src_b[2];
  const int a_cols = min(g.a_ld - m0, 128), b_cols = min(g.b_ld - n0, 128);     // valid bytes of a tile row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3);
    const int pp = lane & 7;                                       // physical 16-byte piece of the 128-byte row
    const int col = (((pp >> 1) ^ ((row >> 1) & 3)) << 5) + (pp & 1) * 16;   // logical byte column
    src_a[i] = row * g.a_ld + m0 + min(col, max(a_cols - 16, 0));   // (beyond the edge: the last valid piece; masked outputs)
    src_b[i] = row * g.b_ld + n0 + min(col, max(b_cols - 16, 0));
  }
  auto pin = [](const char* p) {
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * kTn8Stage;
    char* sB = sA + kTn8Op;
    const char* pa = pin(Ab + (int64_t)kt * kTn8Rows * g.a_ld);
    const char* pb = pin(Bb + (int64_t)kt * kTn8Rows * g.b_ld);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma_1k<BNF_TN_AUX>(pa, (uint32_t)src_a[i], sA + (wave * 2 + i) * 1024);
      dma_1k<BNF_TN_AUX>(pb, (uint32_t)src_b[i], sB + (wave * 2 + i) * 1024);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const Tr8Lane tl = tr8_lane(lane);
  const int frow = lane & 31, kg = lane >> 5;
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
  uint32_t off_a[2], off_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = tl.kg * 8 + tl.j, in = tl.half * 16 + tl.q * 8;
    off_a[i] = lds0 + (uint32_t)(row * 128 + (((wr * 2 + i) ^ (tl.j >> 1)) << 5) + in);
    off_b[i] = lds0 + (uint32_t)(kTn8Op + row * 128 + (((wc * 2 + i) ^ (tl.j >> 1)) << 5) + in);
  }

  if (kt0 < kt1) {
    stage(0, kt0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      if (kt + 1 < kt1) stage(buf ^ 1, kt + 1);
      u32x2_t ra[4][2], rb[4][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t aa = off_a[i] + buf * kTn8Stage, ab = off_b[i] + buf * kTn8Stage;
        ra[0][i] = lds_tr8_b64<0>(aa);           rb[0][i] = lds_tr8_b64<0>(ab);
        ra[1][i] = lds_tr8_b64<16 * 128>(aa);    rb[1][i] = lds_tr8_b64<16 * 128>(ab);
        ra[2][i] = lds_tr8_b64<32 * 128>(aa);    rb[2][i] = lds_tr8_b64<32 * 128>(ab);
        ra[3][i] = lds_tr8_b64<48 * 128>(aa);    rb[3][i] = lds_tr8_b64<48 * 128>(ab);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        asm volatile("" : "+v"(ra[ks][0]), "+v"(ra[ks][1]), "+v"(rb[ks][0]), "+v"(rb[ks][1]));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma8<false>(ra[ks][i], rb[ks][j], acc[i][j]);
      }
      __syncthreads();
    }
  }

  const float qs = ep.scale * (ep.qscale ? ep.qscale[e] : 1.f);
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
          const float v = acc[i][j][r] * qs;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        } else if (ep.bias_row && m == g.M) {     // the ones column of the features (1.0 exactly in e4m3): d bias0
          atomicAdd(&ep.grad[(int64_t)e * ep.grad_stride + ep.off_bias_row + n], acc[i][j][r] * (ep.qscale ? ep.qscale[e] : 1.f));
        }
      }
  }
}

// ===========================================================================
// gemm_tn_skinny8 -- layer 0 (Fp = 64, W a multiple of 512) as a 64 x 512 row stream: gemm_tn_skinny's ring of four
// stages, 64 rows each (H0q 4 KiB + dZq_0 32 KiB per stage: the same 36 KiB), FOUR k steps per barrier, one transpose
// read per fragment
// ===========================================================================
constexpr int kSk8Rows = 64, kSk8A = kSk8Rows * 64, kSk8B = kSk8Rows * 512, kSk8Stage = kSk8A + kSk8B;
constexpr int kSk8Lds = kSkStages * kSk8Stage;
__global__ __launch_bounds__(512, 2) void gemm_tn_skinny8(const GemmArgs g, const EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const uint32_t per_member = (uint32_t)(g.tiles_n * g.splitk);
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int split = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int n0 = tn * 512;

  const int nk_total = g.K / kSk8Rows;
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)e * g.a_batch;
  const char* Bb = reinterpret_cast<const char*>(g.B) + (int64_t)e * g.b_batch + n0;

  // staging per stage and wave: 1 instruction of H0q (16 rows of 64 bytes; waves 4-7 repeat those of waves 0-3, which
  // keeps the vmcnt bookkeeping uniform) + 4 instructions of dZq_0 (2 rows of 512 bytes each)
  const int qa = wave & 3;
  uint32_t src_a;
  {
    const int row = qa * 16 + (lane >> 2), pp = lane & 3;
    src_a = (uint32_t)row * 64u + (uint32_t)((((pp >> 1) ^ ((row >> 2) & 1)) << 5) + (pp & 1) * 16);
  }
  uint32_t src_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 2 + (lane >> 5), pp = lane & 31;
    src_b[i] = (uint32_t)row * (uint32_t)g.b_ld + (uint32_t)((((pp >> 1) ^ (row & 7)) << 5) + (pp & 1) * 16);
  }
  auto pin = [](const char* p) {
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * kSk8Stage;
    char* sB = sA + kSk8A;
    const char* pa = pin(Ab + (int64_t)kt * kSk8Rows * 64);
    const char* pb = pin(Bb + (int64_t)kt * kSk8Rows * g.b_ld);
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

  const Tr8Lane tl = tr8_lane(lane);
  const int frow = lane & 31, kg = lane >> 5;
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
  uint32_t off_a[2], off_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = tl.kg * 8 + tl.j, in = tl.half * 16 + tl.q * 8;
    off_a[i] = lds0 + (uint32_t)(row * 64 + ((i ^ (tl.j >> 2)) << 5) + in);
    off_b[i] = lds0 + (uint32_t)(kSk8A + row * 512 + (((wave * 2 + i) ^ tl.j) << 5) + in);
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
      u32x2_t ra[4][2], rb[4][2];   // [ks][tile]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t aa = off_a[i] + sb * kSk8Stage, ab = off_b[i] + sb * kSk8Stage;
        ra[0][i] = lds_tr8_b64<0>(aa);          rb[0][i] = lds_tr8_b64<0>(ab);
        ra[1][i] = lds_tr8_b64<16 * 64>(aa);    rb[1][i] = lds_tr8_b64<16 * 512>(ab);
        ra[2][i] = lds_tr8_b64<32 * 64>(aa);    rb[2][i] = lds_tr8_b64<32 * 512>(ab);
        ra[3][i] = lds_tr8_b64<48 * 64>(aa);    rb[3][i] = lds_tr8_b64<48 * 512>(ab);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        asm volatile("" : "+v"(ra[ks][0]), "+v"(ra[ks][1]), "+v"(rb[ks][0]), "+v"(rb[ks][1]));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma8<false>(ra[ks][i], rb[ks][j], acc[i][j]);
      }
    }
  }

  const float qm = ep.qscale ? ep.qscale[e] : 1.f, qs = ep.scale * qm;
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
          const float v = acc[i][j][r] * qs;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        } else if (ep.bias_row && m == g.M) {
          atomicAdd(&ep.grad[(int64_t)e * ep.grad_stride + ep.off_bias_row + n], acc[i][j][r] * qm);
        }
      }
  }
}

// ===========================================================================
// gemm_tn_ring8 -- the W x W layers: gemm_tn_ring's 256 x 256 tile, eight waves of 128 x 64, four LDS-DMA stages of the
// same 32 KiB -- but 64 batch rows each.  BNF_RING8_X64 (default): ONE K = 64 MFMA per fragment pair and stage (8 per wave
// and barrier; a fragment = the stage's four transpose reads in an 8-register operand).  -DBNF_RING8_X64=0, the first
// form: four K = 16 steps (32 MFMAs per wave) per barrier, 6 transpose reads per 8 MFMAs (bf16: 12), fragment sets
// double-buffered by k step -- the set of step ks + 2 requested right after the MFMAs of step ks were issued (steps 2, 3
// request steps 0, 1 of the NEXT stage, which the barrier at the top of this iteration certified), counted waits.
// ===========================================================================
template <int TAG, bool SWAP>
__global__ __launch_bounds__(512, 2) void gemm_tn_ring8(const GemmArgs g, const EpiArgs ep) {
  constexpr int kRows = 64, kOp = kRows * 256, kStage = 2 * kOp;     // = kRgOp, kRgStage
  static_assert(kStage == kRgStage, "same ring size as the bf16 kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int n_multi = g.n_multi > 1 ? g.n_multi : 1;
  const uint32_t per_layer = (uint32_t)(g.tiles_m * g.tiles_n * g.splitk);
  const uint32_t per_member = per_layer * (uint32_t)n_multi;
  uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(w / per_member);
  w -= (uint32_t)e * per_member;
  const int which = (int)(w / per_layer);
  w -= (uint32_t)which * per_layer;
  const int tiles = g.tiles_m * g.tiles_n;
  const int split = (int)(w / (uint32_t)tiles);
  w -= (uint32_t)split * tiles;
  const int tm = (int)(w / (uint32_t)g.tiles_n), tn = (int)(w % (uint32_t)g.tiles_n);
  const int m0 = tm * 256, n0 = tn * 256;
  const void* gA = g.A;
  const void* gB = g.B;
  int32_t off_out = ep.off_out;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (g.n_multi > 1 && which == k) { gA = g.A_multi[k]; gB = g.B_multi[k]; off_out = g.off_out_multi[k]; }

  const int nk_total = g.K / kRows;
  const int nk_per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = split * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  const char* Ab = reinterpret_cast<const char*>(gA) + (int64_t)e * g.a_batch + m0;
  const char* Bb = reinterpret_cast<const char*>(gB) + (int64_t)e * g.b_batch + n0;

  // staging: one wave instruction = 1 KiB = 4 rows of 256 bytes; 16 per operand and stage, 2 per wave
  uint32_t src_a[2], src_b[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wave * 2 + q) * 4 + (lane >> 4), pp = lane & 15;
    const uint32_t col = (uint32_t)((((pp >> 1) ^ (row & 7)) << 5) + (pp & 1) * 16);
    src_a[q] = (uint32_t)row * (uint32_t)g.a_ld + col;
    src_b[q] = (uint32_t)row * (uint32_t)g.b_ld + col;
  }
  auto pin = [](const char* p) {
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  typedef __attribute__((address_space(3))) char lds_c_t;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_c_t*)smem);
  auto stage = [&](int buf, int kt) {
    const char* pa = pin(Ab + (int64_t)kt * kRows * g.a_ld);
    const char* pb = pin(Bb + (int64_t)kt * kRows * g.b_ld);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#if BNF_RING8_X64
      const uint32_t la = lds_base + (uint32_t)(buf * kStage + (wave * 2 + q) * 1024);
      dma_1k_asm(pa, src_a[q], la);
      dma_1k_asm(pb, src_b[q], la + (uint32_t)kOp);
#else
      char* sA = smem + buf * kStage;
      dma_1k<BNF_TN_AUX>(pa, src_a[q], sA + (wave * 2 + q) * 1024);
      dma_1k<BNF_TN_AUX>(pb, src_b[q], sA + kOp + (wave * 2 + q) * 1024);
#endif
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const Tr8Lane tl = tr8_lane(lane);
  const int frow = lane & 31, kg = lane >> 5;
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
  uint32_t off_a[4], off_b[2];
  {
    const int row = tl.kg * 8 + tl.j, in = tl.half * 16 + tl.q * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) off_a[i] = lds0 + (uint32_t)(row * 256 + (((wr * 4 + i) ^ tl.j) << 5) + in);
#pragma unroll
    for (int j = 0; j < 2; ++j) off_b[j] = lds0 + (uint32_t)(kOp + row * 256 + (((wc * 2 + j) ^ tl.j) << 5) + in);
  }
  constexpr int kPerWave = 4;               // LDS-DMA loads of one stage per wave
  auto vm = [](int n) constexpr { return (n & 15) | ((n >> 4) << 14) | 0x0F70; };   // s_waitcnt vmcnt(n) only
  constexpr int kWait1 = vm(kPerWave);
  constexpr int kWaitAll = 0x0F70;
#if BNF_RING8_X64
  // One K = 64 MFMA per fragment pair and stage (8 per wave and barrier instead of 32).  A fragment = the stage's four
  // transpose reads (k steps 0 .. 3) in one 8-register operand -- through the BUILTIN read here, so that the register
  // allocator places the four results in the operand's registers itself (assembled from inline-asm results they were
  // copied: 394 v_mov_b64 and 588 bytes of scratch per wave) and counts its own lgkmcnt waits; what pins the order against
  // the barriers and the LDS-DMA is a compiler-level memory barrier on either side.  The next stage's fragments are
  // requested under this stage's MFMAs (the barrier at the top of this trip certified that stage).
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2i_t lds_v2i_t;
  auto read4 = [&](uint32_t rel) {             // rel: byte offset from smem of this lane's k-step-0 datum
    lds_v2i_t* p = (lds_v2i_t*)(smem + rel);
    const v2i_t r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p), r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p + 16 * 256 / 8);
    const v2i_t r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p + 2 * 16 * 256 / 8), r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p + 3 * 16 * 256 / 8);
    return i32x8{r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
  };
  uint32_t rel_a[4], rel_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) rel_a[i] = off_a[i] - lds0;
#pragma unroll
  for (int j = 0; j < 2; ++j) rel_b[j] = off_b[j] - lds0;
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
  asm volatile("" ::: "memory");
  i32x8 fa[4], fb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 2; ++j) fb[j] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
  if (kt0 < kt1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = read4(rel_b[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = read4(rel_a[i]);
  }
  for (int ktb = kt0; ktb < kt1; ktb += kRgStages) {
#pragma unroll
    for (int sb = 0; sb < kRgStages; ++sb) {
      const int kt = ktb + sb;
      if (kt >= kt1) break;
      if (kt + 2 < kt1) __builtin_amdgcn_s_waitcnt(kWait1);
      else __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + kRgStages - 1 < kt1) stage((sb + kRgStages - 1) % kRgStages, kt + kRgStages - 1);
      asm volatile("" ::: "memory");
      const uint32_t nxt = (uint32_t)(((sb + 1) % kRgStages) * kStage);
      // ONE fragment set: a[i] of the next stage is requested right after the two MFMAs that read a[i], b[0 .. 1] after
      // the stage's last MFMA (sched_barrier: the scheduler would hoist the reads and hold two sets -- 96 registers on top
      // of 128 accumulators spilled)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma8x64<SWAP>(fa[i], fb[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        fa[i] = read4(rel_a[i] + nxt);     // (past the last stage: stale bytes of the ring, never multiplied)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read4(rel_b[j] + nxt);
      asm volatile("" ::: "memory");
    }
  }
#else
  struct Frags {
    u32x2_t a[4], b[2];
  };
  auto read_k = [&](Frags& f, int sb, auto ks_tag) {
    constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = lds_tr8_b64<ks * 16 * 256>(off_a[i] + sb * kStage);
#pragma unroll
    for (int j = 0; j < 2; ++j) f.b[j] = lds_tr8_b64<ks * 16 * 256>(off_b[j] + sb * kStage);
  };
  // this set has landed (the OTHER set's six reads, issued after it, may still be in flight); `all`: nothing behind it
  auto wait_set = [&](Frags& f, bool all) {
    if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]));
    else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]));
  };
  auto mma_k = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma8<SWAP>(f.a[i], f.b[j], acc[i][j]);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

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
#pragma unroll
    for (int sb = 0; sb < kRgStages; ++sb) {
      const int kt = ktb + sb;
      if (kt >= kt1) break;
      if (kt + 2 < kt1) __builtin_amdgcn_s_waitcnt(kWait1);
      else __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_s_barrier();
      if (kt + kRgStages - 1 < kt1) stage((sb + kRgStages - 1) % kRgStages, kt + kRgStages - 1);
      const bool more = kt + 1 < kt1;
      const int sn = (sb + 1) % kRgStages;
      wait_set(f0, false);
      mma_k(f0);
      read_k(f0, sb, K2{});
      wait_set(f1, false);
      mma_k(f1);
      read_k(f1, sb, K3{});
      wait_set(f0, false);
      mma_k(f0);
      if (more) read_k(f0, sn, K0{});
      wait_set(f1, !more);
      mma_k(f1);
      if (more) read_k(f1, sn, K1{});
    }
  }

#endif

  const float qs = ep.scale * (ep.qscale ? ep.qscale[e] : 1.f);
  float* out = ep.out_f32 ? ep.out_f32 + (int64_t)e * ep.f32_batch
                          : ep.grad + (int64_t)e * ep.grad_stride + off_out;
  if constexpr (SWAP) {
    const int mw = m0 + wr * 128 + frow;
    const int nw = n0 + wc * 64 + 4 * kg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* orow = out + (int64_t)(mw + i * 32) * ep.ld_f32 + nw;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float v[4] = {acc[i][j][rg * 4] * qs, acc[i][j][rg * 4 + 1] * qs, acc[i][j][rg * 4 + 2] * qs,
                              acc[i][j][rg * 4 + 3] * qs};
          store4u(orow + j * 32 + 8 * rg, 4, v);     // (the gradient leaf starts at any 4-byte aligned offset)
        }
    }
  } else {
    const int mw = m0 + wr * 128 + 4 * kg;
    const int nw = n0 + wc * 64 + frow;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
          const float v = acc[i][j][r] * qs;
          if (g.splitk > 1) atomicAdd(&out[(int64_t)m * ep.ld_f32 + n], v);
          else out[(int64_t)m * ep.ld_f32 + n] = v;
        }
    }
  }
}

}  // namespace bnf
