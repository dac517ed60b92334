// bnf_kernels.h -- everything of a BayesNF train / predict step that is not a
// dense contraction: featurisation (fwd + bwd), output layer + likelihood +
// last-layer activation backward, prior + Adam, weight packing, VI sampling,
// mixture quantiles.  All kernels are batched over (virtual) ensemble members
// with blockIdx.y = member.
#pragma once

#include "bnf_device.h"
#include "../../include/bnf.h"

namespace bnf {

// Device copy of the static network description (passed by value, ~700 B).
// Per-member table of transformed scalar leaves (member_scalars_row, once per step, by the kernel that packs the weights):
//   [l]                 softplus(layer scale l)          [BNF_MAX_LAYERS]     sigmoid(activation weight)
//   [BNF_MAX_LAYERS+1]  softplus(output scale)
//   [kScalGroup + g]    softplus(feature-group scale g)  [kScalInput + d]     in_scale[d] * exp(log_scale_adjustment[d])
constexpr int kScalGroup = BNF_MAX_LAYERS + 2;
constexpr int kScalInput = kScalGroup + BNF_MAX_GROUPS;
constexpr int kScalGfac = kScalInput + BNF_MAX_INPUTS;      // [kScalGfac + g] sigmoid(scale_g) / softplus(scale_g): d softplus(x) / dx over the value
constexpr int kScalStride = kScalGfac + BNF_MAX_GROUPS;

// sin / cos of 2 pi x (x in revolutions), evaluated like the reference: ocml sincosf of the
// float32 product float32(2 pi) * x.  (The hardware v_sin_f32 / v_cos_f32 were tried for the
// bf16 pipeline: no measurable gain -- the feature kernels are not VALU-bound -- so the
// accurate path is used everywhere.)
__device__ __forceinline__ void sincos_rev(float x, float* s, float* c) {
  sincosf(6.28318530717958647692f * x, s, c);
}

struct NetDev {
  // W: width the kernels run at (a multiple of 64); Wt: the model's width (fan-in of the layers above
  // layer 0).  Wt < W: the kernels see zero-padded copies of the width-dependent leaves (k_pad_params).
  int32_t D, F, Fp, W, Wt, depth, P, n_groups, n_freqs, n_interact, obs;
  int32_t group_kind[BNF_MAX_GROUPS], group_arg[BNF_MAX_GROUPS], group_ncols[BNF_MAX_GROUPS],
      group_col0[BNF_MAX_GROUPS], group_scale_off[BNF_MAX_GROUPS];
  int32_t fdeg[BNF_MAX_INPUTS];
  float in_scale[BNF_MAX_INPUTS];
  int32_t interact[BNF_MAX_INTERACT][2];
  int32_t off_lns, off_shape, off_infl;
  int32_t off_bias[BNF_MAX_LAYERS + 1], off_kernel[BNF_MAX_LAYERS + 1], off_ls[BNF_MAX_LAYERS];
  int32_t off_os, off_lsa, off_law;
};

// Which data row does batch position r of (virtual) member e read?
struct RowSrc {
  int32_t mode;          // 0 identity, 1 per-member epoch shuffle, 2 shared random batch, 3 caller's row table
  int32_t S;             // virtual members per member (VI samples), >= 1
  uint64_t seed;
  uint64_t epoch;        // mode 1: epoch ; mode 2: step
  int64_t pos0;          // mode 1 / 3: step * B
  int64_t n_rows;        // N
  int64_t member_offset; // global id of local member 0
  const int32_t* table;  // mode 3: this epoch's (members, table_ld) shuffled row ids (bnf_row_tables)
  int64_t table_ld;
};

__device__ __forceinline__ int64_t row_of(const RowSrc& rs, int e, int64_t r) {
  if (rs.mode == 0) return r;
  if (rs.mode == 3) return (int64_t)rs.table[(int64_t)(e / rs.S) * rs.table_ld + rs.pos0 + r];
  if (rs.mode == 1) {
    const FeistelKey fk = feistel_key(rs.seed, (uint32_t)(rs.member_offset + e / rs.S), rs.epoch,
                                      STREAM_SHUFFLE, (uint64_t)rs.n_rows);
    return (int64_t)feistel_perm(fk, (uint64_t)(rs.pos0 + r));
  }
  const FeistelKey fk = feistel_key(rs.seed, 0u, rs.epoch, STREAM_VI_BATCH, (uint64_t)rs.n_rows);
  return (int64_t)feistel_perm(fk, (uint64_t)r);
}

constexpr float kTwoPiF = 6.2831854820251465f;  // float32(2*pi), as jnp evaluates 2*jnp.pi in f32

// ---------------------------------------------------------------------------
// seasonal feature table (data-constant):  S[n][j]      = cos(y)/h_j
//                                          S[n][nf + j] = sin(y)/h_j
// y = fl32(fl32(fl32(2 pi) f_j) t_n), exactly the float32 argument of the
// reference (models.py:73-74); cos/sin of that float are taken in double and
// rounded once.  Time is column 0 of X.
// ---------------------------------------------------------------------------
struct FreqTab {
  int32_t n;
  float f[BNF_MAX_FREQS], h[BNF_MAX_FREQS];
};

__global__ void k_seasonal_table(const float* __restrict__ X, int64_t n_rows, int D, FreqTab ft,
                                 float* __restrict__ S) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float t = X[r * D];
  for (int j = 0; j < ft.n; ++j) {
    const float cf = kTwoPiF * ft.f[j];
    const float y = cf * t;
    double s, c;
    sincos((double)y, &s, &c);
    S[r * (2 * ft.n) + j] = (float)c / ft.h[j];
    S[r * (2 * ft.n) + ft.n + j] = (float)s / ft.h[j];
  }
}

// ---------------------------------------------------------------------------
// featurise forward (models.py:218-252): one thread per batch row, 128 rows per
// block.  The row-major tile (rows, Fp) is staged in LDS and leaves as coalesced
// 16-byte stores (H0t is a legacy optional transposed copy, null in every pipeline).
// Loads are issued in batches: a load under a per-group branch costs a full memory latency.
// Also gathers the target of the row.
// ---------------------------------------------------------------------------
constexpr int kFeatRows = 128;

template <typename T>
__global__ __launch_bounds__(kFeatRows) void k_featurize(
    NetDev nd, RowSrc rs, const float* __restrict__ X, const float* __restrict__ Stab,
    const float* __restrict__ y, const float* __restrict__ scal, int64_t B,
    T* __restrict__ H0, int64_t h0_batch, T* __restrict__ H0f, int64_t h0f_batch, int32_t ldt,
    float* __restrict__ ybat, int64_t ybat_batch, int32_t n_members, int32_t n_ones,
    uint8_t* __restrict__ H0q = nullptr) {
  // H0q (optional, 2-byte T; compute_dtype 'fp8'): a third copy, row-major (Bp, Fp) OCP e4m3 (un-scaled: features are O(1),
  // the ones columns exact) -- the A operand of the layer-0 weight gradient on fp8 operands (bnf_gemm8.h)
  // n_ones (0 or 2; the panel kernel's F0 forms): that many feature columns behind the F real ones hold 1.0 -- the
  // contraction partners of the bias rows in the forward-packed layer-0 weights, and the row of the layer-0 weight
  // gradient that IS the bias gradient
  // H0f (optional, 2-byte T): a second copy in MFMA A-fragment-major order for the row-panel
  // kernel: element (row r, k) at ((r / 32 * Fp / 16 + k / 16) * 64 + (k % 16) / 8 * 32 + r % 32) * 8 + k % 8,
  // so that the 32-row x 16-deep fragment a wave multiplies is ONE contiguous 1 KiB load (a lane
  // reading 16 bytes of its own row-major row touches 64 different 64-byte segments per wave).
  extern __shared__ __attribute__((aligned(16))) char fsm[];
  constexpr int kEpc = 16 / Elem<T>::kBytes;
  T* tile = reinterpret_cast<T*>(fsm);
  const int pitch = nd.Fp + kEpc;
  // one-dimensional grid, MEMBER FASTEST: consecutive workgroups featurise the same rows for different members, so
  // the rows of X and of the seasonal table come out of L2 for all but the first (row block outermost, they were
  // re-read from HBM once per member: 0.9 of the kernel's 3.3 GB at C5/8)
  const int e = (int)(blockIdx.x % (uint32_t)n_members);
  const int64_t r0 = (int64_t)(blockIdx.x / (uint32_t)n_members) * kFeatRows;
  const int64_t r = r0 + threadIdx.x;
  const float* sc = scal + (int64_t)e * kScalStride;   // transformed scalar leaves of this member
  if (r < B) {
    const int64_t row = row_of(rs, e, r);
    const float* x = X + row * nd.D;
    // all inputs in flight together (clamped index instead of a branch per input: a load under
    // a branch is followed by its own s_waitcnt, i.e. one full memory latency per input)
    float xr[BNF_MAX_INPUTS], u[BNF_MAX_INPUTS];
#pragma unroll
    for (int d = 0; d < BNF_MAX_INPUTS; ++d) xr[d] = x[min(d, nd.D - 1)];
#pragma unroll
    for (int d = 0; d < BNF_MAX_INPUTS; ++d) u[d] = xr[d] / sc[kScalInput + min(d, nd.D - 1)];
    T* trow = tile + threadIdx.x * pitch;
    auto put = [&](int col, float v) { Elem<T>::store(trow + col, v); };
    for (int g = 0; g < nd.n_groups; ++g) {
      const float sp = sc[kScalGroup + g];
      const int c0 = nd.group_col0[g], nc = nd.group_ncols[g];
      const int kind = nd.group_kind[g];
      if (kind == BNF_GROUP_INPUT) {
        for (int d = 0; d < nd.D; ++d) put(c0 + d, u[d] * sp);
      } else if (kind == BNF_GROUP_FOURIER) {
        const int deg = nc >> 1;
        float ud = 0.f;
#pragma unroll
        for (int d = 0; d < BNF_MAX_INPUTS; ++d)
          if (d == nd.group_arg[g]) ud = u[d];
        for (int k = 0; k < deg; ++k) {
          float sn, cs;
          if constexpr (sizeof(T) == 2) {
            // bf16 features (8 significant bits): the hardware v_sin / v_cos (argument in revolutions,
            // ~1e-6 absolute) and v_rcp instead of ocml sincosf + IEEE divides -- the kernel was
            // VALU-bound on them (r02z counters: VALU busy 85 %); the fp32 pipeline keeps the
            // reference-accurate path
            const float x = ud * (float)(1u << k);
            const float fx = x - floorf(x);
            sn = __builtin_amdgcn_sinf(fx);
            cs = __builtin_amdgcn_cosf(fx);
            const float q = __builtin_amdgcn_rcpf((float)(k + 1)) * sp;
            put(c0 + k, cs * q);
            put(c0 + deg + k, sn * q);
            continue;
          }
          sincos_rev(ud * (float)(1u << k), &sn, &cs);
          const float den = (float)(k + 1);
          put(c0 + k, (cs / den) * sp);
          put(c0 + deg + k, (sn / den) * sp);
        }
      } else if (kind == BNF_GROUP_SEASONAL) {
        const float* srow = Stab + row * nc;
        int j = 0;
        for (; j + 8 <= nc; j += 8) {   // eight table entries in flight per wait
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = srow[j + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) put(c0 + j + q, v[q] * sp);
        }
        for (; j < nc; ++j) put(c0 + j, srow[j] * sp);
      } else {
        for (int k = 0; k < nc; ++k) {
          float up = 0.f, uq = 0.f;
#pragma unroll
          for (int d = 0; d < BNF_MAX_INPUTS; ++d) {
            if (d == nd.interact[k][0]) up = u[d];
            if (d == nd.interact[k][1]) uq = u[d];
          }
          put(c0 + k, (up * uq) * sp);
        }
      }
    }
    for (int c = nd.F; c < nd.Fp; ++c) Elem<T>::store(trow + c, c < nd.F + n_ones ? 1.f : 0.f);  // K padding of the contraction
    if (ybat) ybat[(int64_t)e * ybat_batch + r] = y ? y[row] : 0.f;
  }
  __syncthreads();
  // Copy-out.  The tile's rows are consecutive rows of the row-major destination, so 16-byte piece q of the tile (row
  // q / cpr, chunk q % cpr) goes to byte 16 q of ONE contiguous block; only the LDS address needs the row, and that
  // is carried along (a thread's pieces are kFeatRows apart) instead of divided out per piece: the division and the
  // 64-bit index arithmetic were ~35 instructions, seven of them quarter-rate multiplies, per 16 bytes.  Four pieces in
  // flight per wait.
  const int cpr = nd.Fp / kEpc;  // 16-byte chunks per row
  const int n_live = (int)min((int64_t)kFeatRows, B - r0);   // rows of this block that exist
  const int pitch_b = pitch * Elem<T>::kBytes;
  auto copy_out = [&](char* dst_block, int ppr, int sb, auto conv) {   // ppr pieces per row, sb source bytes per piece
    const int total = n_live * ppr;
    const int dc = kFeatRows % ppr, d_off = (kFeatRows / ppr) * pitch_b + dc * sb, wrap = pitch_b - ppr * sb;
    int cc = (int)threadIdx.x % ppr, off = ((int)threadIdx.x / ppr) * pitch_b + cc * sb;
    for (int q = threadIdx.x; q < total; q += 4 * kFeatRows) {
      u32x4 v[4];
      int offs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        offs[i] = off;
        cc += dc; off += d_off;
        if (cc >= ppr) { cc -= ppr; off += wrap; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q + i * kFeatRows < total) v[i] = conv(fsm + offs[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q + i * kFeatRows < total) *reinterpret_cast<u32x4*>(dst_block + (size_t)(q + i * kFeatRows) * 16) = v[i];
    }
  };
  if (n_live > 0) {
    copy_out(reinterpret_cast<char*>(H0 + (int64_t)e * h0_batch + r0 * nd.Fp), cpr, 16,
             [&](const char* src) { return *reinterpret_cast<const u32x4*>(src); });
  }
  if constexpr (sizeof(T) == 2) {
    if (H0q && n_live > 0) {
      __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);   // MODE.FP16_OVFL: conversions clamp to the largest finite value
      // (same element count per member, one byte each; 16-element pieces: 32 bytes of bf16 -> 16 bytes of e4m3)
      copy_out(reinterpret_cast<char*>(H0q + (int64_t)e * h0_batch + r0 * nd.Fp), nd.Fp / 16, 32, [&](const char* src) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(src);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(src + 16);
        return u32x4{bf16x4_to_fp8(lo[0], lo[1], 1.f), bf16x4_to_fp8(lo[2], lo[3], 1.f), bf16x4_to_fp8(hi[0], hi[1], 1.f),
                     bf16x4_to_fp8(hi[2], hi[3], 1.f)};
      });
    }
    if (H0f) {
      T* df = H0f + (int64_t)e * h0f_batch;
      const int ks0 = nd.Fp / 16;
      for (int q = threadIdx.x; q < kFeatRows * cpr; q += kFeatRows) {
        const int lr = q % kFeatRows, cc = q / kFeatRows;   // rows fastest: 16-byte stores of a fragment are contiguous
        const int64_t rr = r0 + lr;
        if (rr < B)
          *reinterpret_cast<u32x4*>(df + ((((rr >> 5) * ks0 + (cc >> 1)) * 64 + (cc & 1) * 32 + (rr & 31)) << 3)) =
              *reinterpret_cast<const u32x4*>(tile + lr * pitch + cc * kEpc);
      }
    }
  }
  (void)ldt;
}

// ---------------------------------------------------------------------------
// The same features with a lane per ROW and a wave per block of consecutive COLUMNS (round 4): which column holds what is
// uniform across the wave -- column descriptors and group scales by scalar loads, scalar branches, no divergence -- where
// k_featurize's thread walks all F columns of its row through per-group loops.  Value for value what k_featurize<bf16_t>
// computes (bit-identical on the GPU: profiles/r04_panel_ab.md r04l / r04m).  Used by the panel kernel's in-kernel
// featurisation experiment (-DBNF_PANEL_FIN=1) only: as a standalone kernel (64-row workgroups, eight per CU) it measured
// SLOWER than k_featurize -- 58 against 37 us at C2, 921 against 436 us at C5/8: a scalar load + branch per column costs
// more than the row walk's per-group loops with their batched loads -- and was removed again.
//   fcol: per padded feature column {kind | group << 8, a, b, 0} (bnf_api.hip builds it from the config's groups):
//         input a; Fourier cos / sin of input a, degree index b; seasonal table column a; interaction a x b; one; zero
// ---------------------------------------------------------------------------
constexpr int kFcZero = 0, kFcInput = 1, kFcCos = 2, kFcSin = 3, kFcSeasonal = 4, kFcInter = 5, kFcOne = 6;
struct FeatIn {
  const float* X; const float* stab; const float* sc;   // inputs (N, D); seasonal table (N, n_seas); the member's scalar table
  const int32_t* fcol;
  int32_t n_in, n_seas;
};
// columns [col0, col0 + CPT) of data row `row` for this lane; live = false: zeros
template <int CPT>
__device__ __forceinline__ void featurize_cols(const FeatIn& f, int64_t row, bool live, int col0, float (&vals)[CPT]) {
  const float* x = f.X + row * f.n_in;
  float u[BNF_MAX_INPUTS];
#pragma unroll
  for (int d = 0; d < BNF_MAX_INPUTS; ++d) u[d] = (d < f.n_in) ? x[d] : 0.f;     // all in flight (uniform bound)
#pragma unroll
  for (int d = 0; d < BNF_MAX_INPUTS; ++d)
    if (d < f.n_in) u[d] = u[d] / f.sc[kScalInput + d];
  static_assert(BNF_MAX_INPUTS == 8, "pick");
  auto pick = [&](int idx) {     // idx is wave-uniform: one scalar branch, one move (a select chain costs 8 VALU slots)
    switch (__builtin_amdgcn_readfirstlane(idx)) {
      case 0: return u[0]; case 1: return u[1]; case 2: return u[2]; case 3: return u[3];
      case 4: return u[4]; case 5: return u[5]; case 6: return u[6]; default: return u[7];
    }
  };
  const float* srow = f.stab + row * f.n_seas;
  const int4* fc = reinterpret_cast<const int4*>(f.fcol) + col0;   // uniform
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int4 md = fc[k];
    const int kind = md.x & 0xff;
    const float sp = f.sc[kScalGroup + ((md.x >> 8) & 0xff)];
    float v = 0.f;
    if (kind == kFcInput) {
      v = pick(md.y) * sp;
    } else if (kind == kFcCos || kind == kFcSin) {
      const float xk = pick(md.y) * (float)(1u << md.z);
      const float fx = xk - floorf(xk);
      const float q = __builtin_amdgcn_rcpf((float)(md.z + 1)) * sp;
      v = (kind == kFcCos ? __builtin_amdgcn_cosf(fx) : __builtin_amdgcn_sinf(fx)) * q;
    } else if (kind == kFcSeasonal) {
      v = srow[md.y] * sp;
    } else if (kind == kFcInter) {
      v = (pick(md.y) * pick(md.z)) * sp;
    } else if (kind == kFcOne) {
      v = 1.f;
    }
    vals[k] = live ? v : 0.f;
  }
}

// ---------------------------------------------------------------------------
// featurise backward: d feature scales, d log_scale_adjustment.
// dH0^T (Fp, ldt) f32 comes from the layer-0 dgrad contraction (transposed so
// that a thread-per-row read is coalesced).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_feat_bwd(
    NetDev nd, RowSrc rs, const float* __restrict__ X, const float* __restrict__ Stab,
    const float* __restrict__ theta, int64_t theta_stride, const float* __restrict__ scal, int64_t B,
    const float* __restrict__ dH0t, int64_t dh0_batch, int32_t ldt, float* __restrict__ grad,
    int64_t grad_stride) {
  __shared__ float red[4][BNF_MAX_GROUPS + BNF_MAX_INPUTS];
  const int e = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float* th = theta + (int64_t)e * theta_stride;
  const float* sc = scal + (int64_t)e * kScalStride;
  float dfs[BNF_MAX_GROUPS], dlsa[BNF_MAX_INPUTS];
#pragma unroll
  for (int g = 0; g < BNF_MAX_GROUPS; ++g) dfs[g] = 0.f;
#pragma unroll
  for (int d = 0; d < BNF_MAX_INPUTS; ++d) dlsa[d] = 0.f;
  if (r < B) {
    const int64_t row = row_of(rs, e, r);
    const float* x = X + row * nd.D;
    float xr[BNF_MAX_INPUTS], u[BNF_MAX_INPUTS], du[BNF_MAX_INPUTS];
#pragma unroll
    for (int d = 0; d < BNF_MAX_INPUTS; ++d) xr[d] = x[min(d, nd.D - 1)];   // all in flight (see k_featurize)
#pragma unroll
    for (int d = 0; d < BNF_MAX_INPUTS; ++d) {
      du[d] = 0.f;
      u[d] = d < nd.D ? xr[d] / sc[kScalInput + min(d, nd.D - 1)] : 0.f;
    }
    const float* dhp = dH0t + (int64_t)e * dh0_batch + r;
    auto dh = [&](int col) { return dhp[(int64_t)col * ldt]; };
#pragma unroll
    for (int g = 0; g < BNF_MAX_GROUPS; ++g) {
      if (g >= nd.n_groups) continue;
      const float sp = sc[kScalGroup + g];
      const int c0 = nd.group_col0[g], nc = nd.group_ncols[g];
      const int kind = nd.group_kind[g];
      float acc = 0.f;
      if (kind == BNF_GROUP_INPUT) {
#pragma unroll
        for (int d = 0; d < BNF_MAX_INPUTS; ++d)
          if (d < nd.D) {
            const float dhv = dh(c0 + d);
            acc += dhv * u[d];
            du[d] += sp * dhv;
          }
      } else if (kind == BNF_GROUP_FOURIER) {
        const int deg = nc >> 1;
        float ud = 0.f;
#pragma unroll
        for (int d = 0; d < BNF_MAX_INPUTS; ++d)
          if (d == nd.group_arg[g]) ud = u[d];
        float dud = 0.f;
        for (int k0 = 0; k0 < deg; k0 += 4) {
          float dcv[4], dsv[4];   // the gradient entries of four degrees in flight together
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = min(k0 + q, deg - 1);
            dcv[q] = dh(c0 + k);
            dsv[q] = dh(c0 + deg + k);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = k0 + q;
            if (k < deg) {
              float s, c;
              const float p2 = (float)(1u << k);
              sincos_rev(ud * p2, &s, &c);
              const float den = (float)(k + 1);
              acc += dcv[q] * (c / den) + dsv[q] * (s / den);
              dud += (kTwoPiF * p2) * (-s * dcv[q] + c * dsv[q]) / den;
            }
          }
        }
#pragma unroll
        for (int d = 0; d < BNF_MAX_INPUTS; ++d)
          if (d == nd.group_arg[g]) du[d] += sp * dud;
      } else if (kind == BNF_GROUP_SEASONAL) {
        const float* srow = Stab + row * nc;
        int j = 0;
        for (; j + 8 <= nc; j += 8) {   // sixteen loads in flight per wait
          float v[8], w[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { v[q] = srow[j + q]; w[q] = dh(c0 + j + q); }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc += w[q] * v[q];
        }
        for (; j < nc; ++j) acc += dh(c0 + j) * srow[j];
      } else {
        for (int k = 0; k < nc; ++k) {
          const int p = nd.interact[k][0], q = nd.interact[k][1];
          float up = 0.f, uq = 0.f;
#pragma unroll
          for (int d = 0; d < BNF_MAX_INPUTS; ++d) {
            if (d == p) up = u[d];
            if (d == q) uq = u[d];
          }
          const float dhv = dh(c0 + k);
          acc += dhv * up * uq;
#pragma unroll
          for (int d = 0; d < BNF_MAX_INPUTS; ++d) {
            if (d == p) du[d] += sp * dhv * uq;
            if (d == q) du[d] += sp * dhv * up;
          }
        }
      }
      dfs[g] = acc;
    }
#pragma unroll
    for (int d = 0; d < BNF_MAX_INPUTS; ++d) dlsa[d] = -du[d] * u[d];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < BNF_MAX_GROUPS; ++g) {
    const float s = wave_sum(dfs[g]);
    if (lane == 0) red[wave][g] = s;
  }
#pragma unroll
  for (int d = 0; d < BNF_MAX_INPUTS; ++d) {
    const float s = wave_sum(dlsa[d]);
    if (lane == 0) red[wave][BNF_MAX_GROUPS + d] = s;
  }
  __syncthreads();
  float* gr = grad + (int64_t)e * grad_stride;
  const int t = threadIdx.x;
  if (t < nd.n_groups) {
    const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    atomicAdd(&gr[nd.group_scale_off[t]], sigmoidf(th[nd.group_scale_off[t]]) * s);
  } else if (t >= BNF_MAX_GROUPS && t < BNF_MAX_GROUPS + nd.D) {
    const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    atomicAdd(&gr[nd.off_lsa + (t - BNF_MAX_GROUPS)], s);
  }
}

// ---------------------------------------------------------------------------
// transforms of the scalar leaves, once per member and step (the contraction epilogues read
// them instead of re-deriving them in every lane): softplus(layer scale_l), sigmoid(activation
// weight), softplus(output scale)         models.py:256-273
// ---------------------------------------------------------------------------
__device__ __forceinline__ void member_scalars_row(const NetDev& nd, const float* __restrict__ th, float* __restrict__ o) {
  for (int l = 0; l < nd.depth; ++l) o[l] = softplusf(th[nd.off_ls[l]]);
  o[BNF_MAX_LAYERS] = sigmoidf(th[nd.off_law]);
  o[BNF_MAX_LAYERS + 1] = softplusf(th[nd.off_os]);
  for (int g = 0; g < nd.n_groups; ++g) o[kScalGroup + g] = softplusf(th[nd.group_scale_off[g]]);
  for (int d = 0; d < nd.D; ++d) o[kScalInput + d] = nd.in_scale[d] * expf(th[nd.off_lsa + d]);
  for (int g = 0; g < nd.n_groups; ++g) o[kScalGfac + g] = sigmoidf(th[nd.group_scale_off[g]]) / o[kScalGroup + g];
}
// ---------------------------------------------------------------------------
// output layer + likelihood, one thread per row.  The last forward contraction
// has already accumulated vacc[row] = sum_j H_L[row][j] k_o[j] (EPI_FWD vdot).
//   forward  (models.py:269-273, 157-164): v = vacc/sqrt W + b_o, out = gamma_o v
//   loss     (inference.py:558-569):       -(N/B) * lik_scale * loglik
//   backward: d out -> dv = gamma_o * dout, d log_noise_scale,
//            d output scale / bias
// vacc is cleared for the next step.  TRAIN=false: forward only (predict path).
// ---------------------------------------------------------------------------
struct RowLossArgs {
  const float* theta;
  int64_t theta_stride;
  int64_t B;
  float* vacc;           // (members, vacc_batch)
  int64_t vacc_batch;
  const float* ybat;     // (members, vacc_batch)
  float* out;            // (members, out_batch) network output
  int64_t out_batch;
  float* dv;             // (members, vacc_batch)
  float* grad;
  int64_t grad_stride;
  float* loss;           // loss[(e / S) * loss_stride] += loss_scale * step loss
  int64_t loss_stride;
  int32_t S;
  float loss_scale;
  float c;               // (N/B) * lik_scale
  float* loss_raw;
  const StepState* st;   // graph replay: loss column offset
};

template <bool TRAIN>
__global__ __launch_bounds__(256) void k_row_loss(NetDev nd, RowLossArgs a) {
  __shared__ float s_red[4][5];
  const int e = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float* th = a.theta + (int64_t)e * a.theta_stride;
  const int L = nd.depth;
  const float gam_o = softplusf(th[nd.off_os]);
  // s_par: d loss / d (log_noise_scale | shape) ; s_infl: d loss / d inflated_loc_probs
  float ll = 0.f, s_doutv = 0.f, s_dvsum = 0.f, s_par = 0.f, s_infl = 0.f;
  if (r < a.B) {
    const int64_t vi = (int64_t)e * a.vacc_batch + r;
    const float v = a.vacc[vi] * (1.0f / sqrtf((float)nd.Wt)) + th[nd.off_bias[L]];
    a.vacc[vi] = 0.f;
    const float out = gam_o * v;
    a.out[(int64_t)e * a.out_batch + r] = out;
    if constexpr (TRAIN) {
      const RowLoss rl = row_loss_eval(nd.obs, th, nd.off_lns, nd.off_shape, nd.off_infl, a.ybat[vi],
                                       out, a.c);
      ll = rl.ll; s_par = rl.d_par; s_infl = rl.d_infl;
      const float dout = rl.dout;
      s_doutv = dout * v;
      const float dvv = gam_o * dout;
      a.dv[vi] = dvv;
      s_dvsum = dvv;
    }
  }
  if constexpr (!TRAIN) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float t0 = wave_sum(ll), t1 = wave_sum(s_doutv), t2 = wave_sum(s_dvsum), t3 = wave_sum(s_par);
  const float t4 = nd.obs == BNF_OBS_ZINB ? wave_sum(s_infl) : 0.f;
  if (lane == 0) {
    s_red[wave][0] = t0; s_red[wave][1] = t1; s_red[wave][2] = t2; s_red[wave][3] = t3;
    s_red[wave][4] = t4;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* gr = a.grad + (int64_t)e * a.grad_stride;
    float u[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) u[i] = s_red[0][i] + s_red[1][i] + s_red[2][i] + s_red[3][i];
    const float step_loss = -a.c * u[0];
    atomicAdd(&a.loss[(int64_t)(e / a.S) * a.loss_stride + (a.st ? a.st->col : 0)], a.loss_scale * step_loss);
    if (a.loss_raw) atomicAdd(&a.loss_raw[e], step_loss);
    atomicAdd(&gr[nd.off_os], sigmoidf(th[nd.off_os]) * u[1]);
    atomicAdd(&gr[nd.off_bias[L]], u[2]);
    atomicAdd(&gr[nd.obs == BNF_OBS_NORMAL ? nd.off_lns : nd.off_shape], u[3]);
    if (nd.obs == BNF_OBS_ZINB) atomicAdd(&gr[nd.off_infl], u[4]);
  }
}

// ---------------------------------------------------------------------------
// backward of the last hidden activation (layer l = L-1):
//   dH = dv k_o^T / sqrt W ;  dZ = gamma_l (dH . act'(A)) ; d bias_l, d gamma_l,
//   d alpha ; d k_o = H^T dv / sqrt W with H = act(A) recomputed (the last hidden
//   output is never stored).
// One wave owns a 64-column strip and walks `row_tiles` 64-row tiles; lane
// (rg = lane & 7, cg = lane >> 3) holds an 8 x 8 block, so A^T (8 rows contiguous)
// and dZ (8 columns contiguous) both move as 16-byte vectors and every wave
// instruction covers full 128-byte lines.
// ---------------------------------------------------------------------------
struct LastBwdArgs {
  const float* theta;
  int64_t theta_stride;
  const void* At;        // (W, ldt) last pre-activation, transposed
  void* dZ;              // (rows, W)
  int64_t act_batch, actt_batch;
  int32_t ldt;
  const float* dv;       // (members, dv_batch)
  int64_t dv_batch;
  float* grad;
  int64_t grad_stride;
  int32_t n_row_tiles;   // 64-row tiles in the (padded) batch
  int32_t tiles_per_task;
};

// (f32: 64 registers of tile + 64 of loads in flight -- two waves per SIMD would spill 198 of them)
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void k_last_bwd(NetDev nd, const LastBwdArgs a) {
  constexpr bool FAST = Elem<T>::kFast;
  const int e = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = nd.W, L = nd.depth, l = L - 1;
  const int strips = W / 64;
  const int task = blockIdx.x * 4 + wave;
  const int n_chunks = (a.n_row_tiles + a.tiles_per_task - 1) / a.tiles_per_task;
  if (task >= strips * n_chunks) return;
  const int strip = task % strips, chunk = task / strips;
  const int rg = lane & 7, cg = lane >> 3;
  const int j0 = strip * 64 + cg * 8;
  const float* th = a.theta + (int64_t)e * a.theta_stride;
  const float inv_sw = 1.0f / sqrtf((float)nd.Wt);
  const float gamma = softplusf(th[nd.off_ls[l]]);
  const float alpha = sigmoidf(th[nd.off_law]);
  const T* __restrict__ At = reinterpret_cast<const T*>(a.At) + (int64_t)e * a.actt_batch;
  T* __restrict__ dZ = reinterpret_cast<T*>(a.dZ) + (int64_t)e * a.act_batch;
  const float* __restrict__ dv = a.dv + (int64_t)e * a.dv_batch;
  float kv[8], cs_b[8], cs_k[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    kv[c] = th[nd.off_kernel[L] + j0 + c] * inv_sw;
    cs_b[c] = cs_k[c] = 0.f;
  }
  float s_alpha = 0.f, s_gamma = 0.f;
  const int t_end = min(a.n_row_tiles, (chunk + 1) * a.tiles_per_task);
  for (int t = chunk * a.tiles_per_task; t < t_end; ++t) {
    const int64_t r0 = (int64_t)t * 64 + rg * 8;
    float dvr[8];
    load8(dv + r0, dvr);
    float dz[8][8];  // [row i][col c]
    typename Raw<T>::R8 raws[8];  // all eight column loads of the tile in flight together
#pragma unroll
    for (int c = 0; c < 8; ++c) raws[c] = load_raw8(At + (int64_t)(j0 + c) * a.ldt + r0);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float avc[8];
      unpack(raws[c], avc);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float avv = avc[i];
        const float dh = dvr[i] * kv[c];
        const ActOut o = act_eval<FAST>(avv, alpha);
        s_alpha += dh * o.ediff;
        const float da = dh * o.dact;
        s_gamma += da * avv;
        const float z = gamma * da;
        dz[i][c] = z;
        cs_b[c] += z;
        cs_k[c] += o.h * dvr[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) store8(dZ + (r0 + i) * W + j0, dz[i]);
  }
  // column sums: reduce over the 8 row groups (lane bits 0..2), then one atomic per column
  float* gr = a.grad + (int64_t)e * a.grad_stride;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float b = cs_b[c], k = cs_k[c];
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      b += __shfl_xor(b, off, 64);
      k += __shfl_xor(k, off, 64);
    }
    if (rg == 0) {
      atomicAdd(&gr[nd.off_bias[l] + j0 + c], b);
      atomicAdd(&gr[nd.off_kernel[L] + j0 + c], k * inv_sw);
    }
  }
  const float ta = wave_sum(s_alpha), tg = wave_sum(s_gamma);
  if (lane == 0) {
    atomicAdd(&gr[nd.off_law], alpha * (1.f - alpha) * ta);
    atomicAdd(&gr[nd.off_ls[l]], sigmoidf(th[nd.off_ls[l]]) * tg / gamma);
  }
}

// ---------------------------------------------------------------------------
// packed operand copies of the Dense kernels (both layouts, type T):
//   Kn (n_pad, W)  = K            -> Bt of the dgrad contraction
//   Kt (W, n_pad)  = K^T          -> Bt of the forward contraction
// 32x32 tiles through LDS so both writes are coalesced.
// ---------------------------------------------------------------------------
// split != 0 (T = float, BNF_DTYPE_F32S): every aligned group of 8 consecutive elements of a packed row -- 8 consecutive
// k of the contraction the copy is the B operand of -- holds, in its 32 bytes, the 8 bf16 hi parts and then the 8 bf16 lo
// parts of its floats (hi = bf16(x), lo = bf16(x - hi)): the split-bf16 contraction reads its B fragments ready-made
// (Mma<float>::presplit) instead of splitting every weight once per workgroup that uses it.
template <typename T>
__device__ __forceinline__ void store_packed(T* row, int p, float v, int split) {
  if constexpr (sizeof(T) == 4) {
    if (split) {
      uint16_t* q = reinterpret_cast<uint16_t*>(row + (p & ~7));
      const uint16_t hi = f32_to_bf16_bits(v);
      q[p & 7] = hi;
      q[8 + (p & 7)] = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
      return;
    }
  }
  Elem<T>::store(row + p, v);
}

// ONE launch packs every layer (a job per layer: 32 x 32 tiles [tile0[l], tile0[l + 1]) of grid.x) and fills the member's
// row of the scalar table (block 0 of the member, thread 0: what k_member_scalars does) -- the per-layer launches + the
// scalar kernel were three dependent launches of a twelve-launch step at depth 2, i.e. a sixth of the step at the sizes
// the reference's own fixtures have (100 rows: the step is launch latency).
struct PackWJobs {
  int32_t n_layers;
  int32_t off_kernel[BNF_MAX_LAYERS], n_in[BNF_MAX_LAYERS], n_pad[BNF_MAX_LAYERS], tile0[BNF_MAX_LAYERS + 1];
  void* Kn[BNF_MAX_LAYERS];
  void* Kt[BNF_MAX_LAYERS];
  int64_t pack_batch[BNF_MAX_LAYERS];
};

template <typename T>
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ theta, int64_t theta_stride, PackWJobs jb,
                                                      int32_t W, int32_t split, NetDev nd, float* __restrict__ scal) {
  __shared__ float tile[32][33];
  const int e = blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0 && scal)
    member_scalars_row(nd, theta + (int64_t)e * theta_stride, scal + (int64_t)e * kScalStride);
  int l = 0;
  while (l + 1 < jb.n_layers && (int)blockIdx.x >= jb.tile0[l + 1]) ++l;
  const int bx = (int)blockIdx.x - jb.tile0[l];
  const int n_in = jb.n_in[l], n_pad = jb.n_pad[l];
  const int tiles_j = W / 32;
  const int ti = bx / tiles_j, tj = bx % tiles_j;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* K = theta + (int64_t)e * theta_stride + jb.off_kernel[l];
  T* kn = reinterpret_cast<T*>(jb.Kn[l]) + (int64_t)e * jb.pack_batch[l];
  T* kt = reinterpret_cast<T*>(jb.Kt[l]) + (int64_t)e * jb.pack_batch[l];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = ti * 32 + ty + s * 8, j = tj * 32 + tx;
    const float v = (i < n_in) ? K[(int64_t)i * W + j] : 0.f;
    tile[ty + s * 8][tx] = v;
    if (i < n_pad) store_packed(kn + (int64_t)i * W, j, v, split);
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int j = tj * 32 + ty + s * 8, i = ti * 32 + tx;
    if (i < n_pad) store_packed(kt + (int64_t)j * n_pad, i, tile[tx][ty + s * 8], split);
  }
}

// ---------------------------------------------------------------------------
// Widths that are not a multiple of 64 (the reference accepts any: spatiotemporal.py:217-232).
// The contraction kernels run at the padded width W on a padded COPY of the parameters:
//   theta_pad[e] = [ theta[e] (P floats, verbatim) | padded Dense biases / kernels (zeros in the pad) ]
// NetDev's bias / kernel offsets point into the second part, every other leaf is read from the
// first.  Zero pad => the extra hidden units are exactly 0 in the forward pass (act(0) = 0) and
// carry exactly 0 gradient, so the model is the width-Wt model; fan-in scales use Wt.  The
// gradient comes back through the inverse map.  Optimiser, prior, initialisation and the VI
// sampler only ever see the true layout.
// ---------------------------------------------------------------------------
__global__ void k_pad_params(const float* __restrict__ theta, int64_t P, float* __restrict__ theta_pad,
                             int64_t Pp, const int32_t* __restrict__ pad_src) {
  const int e = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Pp) return;
  const float* th = theta + (int64_t)e * P;
  float v;
  if (j < P) v = th[j];
  else {
    const int32_t src = pad_src[j - P];
    v = src >= 0 ? th[src] : 0.f;
  }
  theta_pad[(int64_t)e * Pp + j] = v;
}
// grad[e][i] = grad_pad[e][fold_src[i]]; the padded gradient is cleared for the next step's atomics
__global__ void k_fold_grad(float* __restrict__ grad_pad, int64_t Pp, const int32_t* __restrict__ fold_src,
                            float* __restrict__ grad, int64_t P) {
  const int e = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  grad[(int64_t)e * P + i] = grad_pad[(int64_t)e * Pp + fold_src[i]];
}

// ---------------------------------------------------------------------------
// Logistic prior + Adam (models.py:94-103, inference.py:569,580,605-606).
//   g_total = g_lik + prior_weight * tanh((theta - loc)/2)
//   loss   += loss_scale * (-prior_weight * sum log Logistic(theta; loc, 1))
// Also clears the gradient buffer for the next step's atomics.
// ---------------------------------------------------------------------------
struct AdamArgs {
  float* theta; float* m; float* v; float* grad;
  int64_t stride;         // P
  int32_t P;
  int32_t off_shape;      // the one leaf with prior loc -1.5
  float lr, bc1, bc2;     // bias corrections 1 - b1^t, 1 - b2^t
  float prior_weight;
  float* loss; int64_t loss_stride; float loss_scale;
  int32_t apply;          // 0: only add the prior gradient into grad (debug path)
  float* loss_raw;
  const StepState* st;    // graph replay: bias corrections and loss column from device memory
  int32_t keep_lo, keep_hi;   // [lo, hi): gradient entries the next step OVERWRITES (weight-gradient stores,
                              // split-K = 1): not cleared here (4-aligned bounds; 67 of 600 MB per step at C2)
};

// end of a replayed step: next Adam step, next loss column
__global__ void k_step_advance(StepState* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long t = st->t + 1;
    st->t = t;
    st->col += 1;
    st->bc1 = (float)(1.0 - pow(0.9, (double)t));
    st->bc2 = (float)(1.0 - pow(0.999, (double)t));
  }
}

#ifndef BNF_ADAM_NT
#define BNF_ADAM_NT 1
#endif
#ifndef BNF_ADAM_NTLOAD
#define BNF_ADAM_NTLOAD 0   // the moments also READ non-temporally (touched once per step)
#endif
// A thread owns four consecutive parameters (16-byte accesses at any 4-byte aligned address: load4u);
// the last thread of a member takes the P % 4 tail element by element.
#ifndef BNF_ADAM_QUADS
#define BNF_ADAM_QUADS 4   // quads of parameters per thread, one after the other (a block covers 1024 x this many consecutive
                           // parameters): C2 105.6 -> 85.5 us (1: 105.6, 2: 86.6, 3: 85.2, 4: 85.5, 8: 91.5), C5/8 41 -> 34
#endif
__global__ __launch_bounds__(256) void k_adam_map(AdamArgs a) {
  __shared__ float red[4];
  const int e = blockIdx.y;
  const float bc1 = a.st ? a.st->bc1 : a.bc1, bc2 = a.st ? a.st->bc2 : a.bc2;
  float lp = 0.f;
#pragma unroll
  for (int qd = 0; qd < BNF_ADAM_QUADS; ++qd) {
  const int p0 = ((blockIdx.x * BNF_ADAM_QUADS + qd) * blockDim.x + threadIdx.x) * 4;
  if (p0 < a.P) {
    const int nv = min(4, a.P - p0);
    const int64_t i0 = (int64_t)e * a.stride + p0;
    float th[4], g[4], m[4], v[4];
    load4u(a.theta + i0, nv, th); load4u(a.grad + i0, nv, g);
    if (a.apply) { load4u<BNF_ADAM_NTLOAD != 0>(a.m + i0, nv, m); load4u<BNF_ADAM_NTLOAD != 0>(a.v + i0, nv, v); }
    // One hardware exp2 / log2 / rcp / sqrt each (1 ulp) instead of the libm tanhf, expf + log1pf and
    // the IEEE divide / sqrt sequences: 246 VALU instructions per element made this kernel VALU-bound
    // (profiles/r02z_mfma_valu_counters.md: VALU busy ~100 %, 4.2 TB/s); with e = exp(-|z|):
    //   tanh(z / 2) = sign(z) (1 - e) / (1 + e),   log Logistic(z) = -z - 2 softplus(-z) = -|z| - 2 log1p(e)
    const float ibc1 = 1.0f / bc1, ibc2 = 1.0f / bc2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (a.prior_weight != 0.f && k < nv) {
        const float z = th[k] - ((p0 + k) == a.off_shape ? -1.5f : 0.f);
        const float az = fabsf(z);
        const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * az);
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        g[k] += a.prior_weight * copysignf((1.0f - e) * r, z);
        lp += -az - 1.38629436111989061883f * __builtin_amdgcn_logf(1.0f + e);   // 2 ln 2 log2(1 + e)
      }
      if (a.apply) {
        m[k] = 0.9f * m[k] + 0.1f * g[k];
        v[k] = 0.999f * v[k] + 0.001f * g[k] * g[k];
        th[k] = th[k] - a.lr * (m[k] * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[k] * ibc2) + 1e-8f);
        g[k] = 0.f;  // ready for the next step's atomics
      }
    }
    if (!(a.apply && p0 >= a.keep_lo && p0 < a.keep_hi)) store4u(a.grad + i0, nv, g);
    if (a.apply) {
      store4u(a.theta + i0, nv, th);
      // the moments are touched once per step: written through (non-temporal), they do not sit dirty in the
      // memory-side cache while the next kernels stream
      store4u<BNF_ADAM_NT != 0>(a.m + i0, nv, m);
      store4u<BNF_ADAM_NT != 0>(a.v + i0, nv, v);
    }
  }
  }
  const float s = wave_sum(lp);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && a.prior_weight != 0.f) {
    const float t = -(a.prior_weight) * (red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&a.loss[(int64_t)e * a.loss_stride + (a.st ? a.st->col : 0)], a.loss_scale * t);
    if (a.loss_raw) atomicAdd(&a.loss_raw[e], t);
  }
}

// ---------------------------------------------------------------------------
// mean-field VI (inference.py:687-720): sample, then combine S sample gradients.
// ---------------------------------------------------------------------------
// The optimiser-side kernels below evaluate a handful of transcendentals per parameter (and per VI
// sample); with libm they were VALU-bound (k_adam_map: ~246 instructions per parameter), so they use
// the hardware exp2 / log2 / rcp / sqrt (1 ulp each) -- parity bars unchanged (tests/test_gpu_parity.py).
__device__ __forceinline__ float hw_exp_neg_abs(float x) { return __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x)); }
__device__ __forceinline__ float hw_log1p_of_exp(float t) { return 0.69314718055994530942f * __builtin_amdgcn_logf(1.0f + t); }  // log(1 + t), t in (0, 1]
__device__ __forceinline__ float hw_softplus(float x) { return fmaxf(x, 0.f) + hw_log1p_of_exp(hw_exp_neg_abs(x)); }
__device__ __forceinline__ float hw_sigmoid(float x) {
  const float t = hw_exp_neg_abs(x), r = __builtin_amdgcn_rcpf(1.0f + t);
  return x >= 0.f ? r : t * r;
}
__device__ __forceinline__ float vi_sigma(float rho) { return 1e-4f + hw_softplus(rho); }

// parameter ranges [lo, hi) a launch of k_vi_sample covers (one range = everything; the step's sampler leaves the
// hidden Dense kernels to k_vi_sample_pack and covers what lies between them)
struct ViSegs {
  int32_t n;
  int32_t lo[BNF_MAX_LAYERS + 2], hi[BNF_MAX_LAYERS + 2];
};
__global__ __launch_bounds__(256) void k_vi_sample(const float* __restrict__ mu,
                                                   const float* __restrict__ rho, int32_t P,
                                                   int32_t S, uint64_t seed, int64_t member_offset,
                                                   uint64_t step, uint32_t stream,
                                                   float* __restrict__ z, int64_t z_member_stride,
                                                   int64_t z_sample_stride, ViSegs segs,
                                                   const float* __restrict__ ext_eps = nullptr,
                                                   JaxNoise jn = JaxNoise{}) {
  // grid: (blocks over the quads of the longest range, members, ranges): a thread serves one quad of parameters
  // (vi_eps_quad: p0 = 4 quad - 1 .. + 3; 16-byte accesses, load4w / store4w) for every sample -- one Philox call per
  // sample
  const int e = blockIdx.y;
  const int lo = segs.lo[blockIdx.z], hi = segs.hi[blockIdx.z];
  const int u0 = ((lo + kEpsQuadPhase) & ~3) + (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int p0 = u0 - kEpsQuadPhase;
  const int klo = max(0, lo - p0), khi = min(4, hi - p0);
  if (khi <= klo) return;
  const int64_t i = (int64_t)e * P + p0;
  float m[4], sg[4];
  load4w(mu + i, klo, khi, m);
  load4w(rho + i, klo, khi, sg);
#pragma unroll
  for (int k = 0; k < 4; ++k) sg[k] = vi_sigma(sg[k]);
  for (int s = 0; s < S; ++s) {
    Normal4 n;
    if (ext_eps) {          // bnf_debug_vi_noise: the caller's standard normals, (members, S, P)
      load4w(ext_eps + ((int64_t)e * S + s) * P + p0, klo, khi, n.v);
    } else if (jn.keys) {   // the reference's stream (jaxseed.vi_noise_keys)
#pragma unroll
      for (int k = 0; k < 4; ++k) n.v[k] = (k >= klo && k < khi) ? jax_normal(jn, e, s, p0 + k) : 0.f;
    } else {
      n = vi_eps_quad(seed, (uint32_t)(member_offset + e), (uint32_t)s, (uint32_t)(u0 >> 2), step, stream);
    }
    float zv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) zv[k] = m[k] + sg[k] * n.v[k];
    store4w(z + (int64_t)e * z_member_stride + (int64_t)s * z_sample_stride + p0, klo, khi, zv);
  }
}

struct ViAdamArgs {
  float* mu; float* rho; float* m_mu; float* v_mu; float* m_rho; float* v_rho;
  float* grad;            // (members*S, P) likelihood gradients wrt z (scaled by 1/kl)
  int32_t P, S, off_shape;
  uint64_t seed; int64_t member_offset; uint64_t step;
  float lr, bc1, bc2, kl_weight;
  float* loss; int64_t loss_stride;
  int32_t apply;
  float* gmu_out; float* grho_out;  // debug: (members, P) each
  const float* ext_eps;             // bnf_debug_vi_noise: (members, S, P) standard normals instead of the generator's
  JaxNoise jn;                      // the reference's stream when jn.keys != null
  const float* z;                   // null: the device generator's noise, made again here.  Else (members*S, P), the samples
                                    // k_vi_sample wrote for this step (theta_c): the noise is recovered as (z - mu) / sigma
  int32_t n_keep;                   // parameter ranges whose sample gradients the next step STORES (weight-gradient
  int32_t keep_lo[BNF_MAX_LAYERS], keep_hi[BNF_MAX_LAYERS];   // kernels without split-K): not cleared here
};

#ifndef BNF_VIADAM_INFLIGHT
#define BNF_VIADAM_INFLIGHT 1   // samples whose gradient loads are in flight together: 1 -> 78 registers, 6 waves per SIMD (C3/8: 179 us; 4 -> 90 registers, 208 us)
#endif
// (several quads per thread, one after the other -- what took k_adam_map from 106 to 86 us -- cost this kernel registers
// and time: 179 -> 191 (2) -> 199 us (4) at C3/8, profiles/r04_panel_ab.md r04ab)
#ifndef BNF_VIADAM_OCC
#define BNF_VIADAM_OCC 4
#endif
template <bool REGEN>
__global__ __launch_bounds__(256, BNF_VIADAM_OCC) void k_vi_adam(ViAdamArgs a) {
  constexpr int NF = BNF_VIADAM_INFLIGHT;
  // grid: (ceil(ceil((P + 1)/4)/256), members): a thread owns one quad of the noise stream -- parameters 4 t - 1 .. 4 t + 2
  // (vi_eps_quad); every access is 16 bytes per lane (load4w), which is what this kernel is bound by: at C3/8 it reads
  // 267 MB of sample gradients and reads + writes 6 x 53 MB of optimiser state
  __shared__ float red[4];
  const int e = blockIdx.y;
  const int u0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int p0 = u0 - kEpsQuadPhase;
  const int klo = max(0, -p0), khi = min(4, a.P - p0);
  float lterm = 0.f;
  if (khi > klo) {
    const int64_t i = (int64_t)e * a.P + p0;
    float mu[4], rho[4], sig[4], inv_sig[4], loc[4];
    float gmu[4] = {0.f, 0.f, 0.f, 0.f}, grho[4] = {0.f, 0.f, 0.f, 0.f}, e2[4] = {0.f, 0.f, 0.f, 0.f}, lpr[4] = {0.f, 0.f, 0.f, 0.f};
    load4w(a.mu + i, klo, khi, mu);
    load4w(a.rho + i, klo, khi, rho);
    // The step's noise: with the device generator (a.z == null) the quad's S Philox calls are simply made again -- one
    // call per sample and four parameters hides under this kernel's memory time, and the S x P samples need not exist
    // in memory at all for the Dense kernels (k_vi_sample_pack writes only their bf16 fragments).  With the caller's
    // noise or the reference's stream (threefry + erfinv: too dear to run twice) k_vi_sample left z_s = mu + sigma eps_s
    // in theta_c and nothing has written it since, so eps_s = (z_s - mu) / sigma -- exact up to the rounding of z_s
    // (|z| 6e-8 / sigma absolute on a unit normal).
    constexpr bool regen = REGEN;     // (a.z == null)
    bool clear[4], all_clear = true, any_clear = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sig[k] = vi_sigma(rho[k]);
      inv_sig[k] = 1.0f / sig[k];
      loc[k] = (p0 + k == a.off_shape) ? -1.5f : 0.f;
      clear[k] = a.apply != 0;
      for (int r = 0; r < a.n_keep; ++r) clear[k] = clear[k] && !(p0 + k >= a.keep_lo[r] && p0 + k < a.keep_hi[r]);
      all_clear = all_clear && clear[k];
      any_clear = any_clear || clear[k];
    }
    const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < a.S; s0 += NF) {
      float gl[NF][4], zs[NF][4];   // the likelihood gradients and samples of NF samples in flight together
#pragma unroll
      for (int ks = 0; ks < NF; ++ks) {
        const int64_t gi = ((int64_t)e * a.S + min(s0 + ks, a.S - 1)) * a.P + p0;
        load4w(a.grad + gi, klo, khi, gl[ks]);
        if constexpr (!regen) load4w(a.z + gi, klo, khi, zs[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < NF; ++ks) {
        const int s = s0 + ks;
        if (s >= a.S) break;
        Normal4 n;
        if constexpr (regen) n = vi_eps_quad(a.seed, (uint32_t)(a.member_offset + e), (uint32_t)s, (uint32_t)(u0 >> 2), a.step, STREAM_VI_EPS);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float eps = regen ? n.v[k] : (zs[ks][k] - mu[k]) * inv_sig[k];
          const float z = (regen ? mu[k] + sig[k] * eps : zs[ks][k]) - loc[k];
          // Logistic(loc, 1) prior: d(-log p)/dz = tanh(z/2), log p = -z - 2 softplus(-z);
          // both from u = exp(-|z|)
          const float u = hw_exp_neg_abs(z);
          const float g = gl[ks][k] + copysignf((1.f - u) * __builtin_amdgcn_rcpf(1.f + u), z);
          gmu[k] += g;
          grho[k] += g * eps;
          e2[k] += eps * eps;
          lpr[k] += -fabsf(z) - 2.f * hw_log1p_of_exp(u);     // -z - 2 softplus(-z)
        }
        if (any_clear) {
          float* gp = a.grad + ((int64_t)e * a.S + s) * a.P + p0;
          if (all_clear) {
            store4w(gp, klo, khi, zero4);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k >= klo && k < khi && clear[k]) gp[k] = 0.f;
          }
        }
      }
    }
    const float invS = 1.f / (float)a.S;
    float m1[4], v1[4], m2[4], v2[4];
    if (a.apply) {
      load4w(a.m_mu + i, klo, khi, m1); load4w(a.v_mu + i, klo, khi, v1);
      load4w(a.m_rho + i, klo, khi, m2); load4w(a.v_rho + i, klo, khi, v2);
    }
    const float ibc1 = 1.0f / a.bc1, ibc2 = 1.0f / a.bc2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gmu[k] *= invS;
      grho[k] = hw_sigmoid(rho[k]) * (grho[k] * invS - __builtin_amdgcn_rcpf(sig[k]));
      // mean_s [ log q(z_s) - log p(z_s) ] for this coordinate
      if (k >= klo && k < khi)
        lterm += (-0.5f * e2[k] * invS - 0.69314718055994530942f * __builtin_amdgcn_logf(sig[k]) - 0.918938533204672742f) - lpr[k] * invS;
      if (a.apply) {
        m1[k] = 0.9f * m1[k] + 0.1f * gmu[k]; v1[k] = 0.999f * v1[k] + 0.001f * gmu[k] * gmu[k];
        mu[k] = mu[k] - a.lr * (m1[k] * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v1[k] * ibc2) + 1e-8f);
        m2[k] = 0.9f * m2[k] + 0.1f * grho[k]; v2[k] = 0.999f * v2[k] + 0.001f * grho[k] * grho[k];
        rho[k] = rho[k] - a.lr * (m2[k] * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v2[k] * ibc2) + 1e-8f);
      }
    }
    if (a.apply) {
      store4w(a.mu + i, klo, khi, mu);
      store4w(a.rho + i, klo, khi, rho);
      store4w<true>(a.m_mu + i, klo, khi, m1); store4w<true>(a.v_mu + i, klo, khi, v1);     // touched once per step: written through
      store4w<true>(a.m_rho + i, klo, khi, m2); store4w<true>(a.v_rho + i, klo, khi, v2);
    } else {
      store4w(a.gmu_out + i, klo, khi, gmu);
      store4w(a.grho_out + i, klo, khi, grho);
    }
  }
  const float s = wave_sum(lterm);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(&a.loss[(int64_t)e * a.loss_stride], a.kl_weight * (red[0] + red[1] + red[2] + red[3]));
}

// ---------------------------------------------------------------------------
// initial values (inference.py:399-427, 203-231)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_params(float* __restrict__ theta, int32_t P,
                                                     const uint8_t* __restrict__ is_matrix,
                                                     int32_t off_lns, float lns_init, uint64_t seed,
                                                     int64_t member_offset) {
  const int e = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float v = 0.f;
  if (is_matrix[p]) {
    const Philox r = philox4x32((uint32_t)p, (uint32_t)(member_offset + e), 0u, STREAM_INIT,
                                (uint32_t)seed, (uint32_t)(seed >> 32));
    v = trunc_normal_m2p2(r.v[0]);
  } else if (p == off_lns) {
    v = lns_init;
  }
  theta[(int64_t)e * P + p] = v;
}

// The reference's OWN initial Dense kernels for a member key chain (bnf_init_params_keys): element i of leaf lf of
// member e = sqrt2 erfinv(u) clipped to (-2, 2), u uniform on [erf(-sqrt2), erf(sqrt2)) from
// random_bits(leaf_keys[e][lf], (size,))[i] -- jax.random.truncated_normal as TFP's TruncatedNormal(0, 1, -2, 2) sampler
// calls it (inference.py:399-427, 203-231); every other leaf 0, log_noise_scale = lns_init.  The f32 erfinv is XLA's
// polynomial (erfinv_f32); the host restatement (jaxseed.truncated_normal_std: scipy f64 erfinv rounded to f32)
// differs from it by at most one ulp in a few elements.
struct LeafTable {
  int32_t n;
  int32_t off[65];        // n + 1 offsets of the packed leaves
};
__global__ __launch_bounds__(256) void k_init_params_keys(float* __restrict__ theta, int32_t P,
                                                          const uint8_t* __restrict__ is_matrix, LeafTable lt,
                                                          const uint32_t* __restrict__ leaf_keys, int32_t off_lns,
                                                          float lns_init, float ua, float ub) {
  const int e = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float v = 0.f;
  if (is_matrix[p]) {
    int lf = 0;
    for (int k = 1; k < lt.n; ++k) lf = (p >= lt.off[k]) ? k : lf;
    const uint32_t n = (uint32_t)(lt.off[lf + 1] - lt.off[lf]), idx = (uint32_t)(p - lt.off[lf]);
    const uint32_t half = (n + 1u) >> 1;
    const uint32_t* k = leaf_keys + ((int64_t)e * lt.n + lf) * 2;
    uint32_t y0, y1;
    if (idx < half) threefry2x32(k[0], k[1], idx, idx + half < n ? idx + half : 0u, &y0, &y1);
    else { threefry2x32(k[0], k[1], idx - half, idx, &y0, &y1); y0 = y1; }
    const float f = __builtin_bit_cast(float, (y0 >> 9) | 0x3F800000u) - 1.0f;
    const float u = fmaxf(ua, f * (ub - ua) + ua);
    const float x = 1.41421356237309504880f * erfinv_f32(u);
    v = fminf(fmaxf(x, -1.99999988f), 1.99999988f);      // nextafter(-+2, 0)
  } else if (p == off_lns) {
    v = lns_init;
  }
  theta[(int64_t)e * P + p] = v;
}

__global__ void k_fill(float* p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// sort keys of one round of jax.random.permutation: bits[e][i] = random_bits(keys[e][round], (n,))[i] -- threefry2x32
// over iota(n) split in halves (an odd n pads the second half with counter 0), first outputs then second outputs;
// vals (round 0): the identity order.   grid (ceil(ceil(n/2)/256), members)
__global__ __launch_bounds__(256) void k_jax_perm_bits(const uint32_t* __restrict__ keys, int32_t rounds, int32_t round,
                                                       uint32_t n, uint32_t* __restrict__ bits, int32_t* __restrict__ vals) {
  const int e = blockIdx.y;
  const uint32_t half = (n + 1u) >> 1;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half) return;
  const uint32_t* k = keys + ((int64_t)e * rounds + round) * 2;
  uint32_t y0, y1;
  threefry2x32(k[0], k[1], j, j + half < n ? j + half : 0u, &y0, &y1);
  uint32_t* b = bits + (int64_t)e * n;
  b[j] = y0;
  if (j + half < n) b[j + half] = y1;
  if (vals) {
    int32_t* v = vals + (int64_t)e * n;
    v[j] = (int32_t)j;
    if (j + half < n) v[j + half] = (int32_t)(j + half);
  }
}

__global__ void k_row_index(RowSrc rs, int64_t B, int32_t* out) {
  const int e = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < B) out[(int64_t)e * B + r] = (int32_t)row_of(rs, e, r);
}

__global__ void k_vi_eps_dump(int32_t P, uint64_t seed, int64_t member_offset, uint64_t step,
                              float* out, JaxNoise jn) {
  const int e = blockIdx.y, s = blockIdx.z, S = gridDim.z;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P)
    out[((int64_t)e * S + s) * P + p] =
        jn.keys ? jax_normal(jn, e, s, p)
                : vi_eps(seed, (uint32_t)(member_offset + e), (uint32_t)s, (uint32_t)p, step, STREAM_VI_EPS);
}

// predict: aux[e] = {0.01 + exp(lns), softplus(shape), sigmoid(infl)}
__global__ void k_forecast_aux(const float* theta, int64_t stride, int32_t n, int32_t off_lns,
                               int32_t off_shape, int32_t off_infl, float* aux) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float* th = theta + (int64_t)e * stride;
  aux[e * 3 + 0] = 0.01f + expf(th[off_lns]);
  aux[e * 3 + 1] = softplusf(th[off_shape]);
  aux[e * 3 + 2] = sigmoidf(th[off_infl]);
}

// ---------------------------------------------------------------------------
// conversions / copies used by the debug entry points
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_to_f32(const T* __restrict__ src, int64_t src_batch, int32_t src_ld,
                         int64_t rows, int32_t cols, float* __restrict__ dst, int transposed) {
  const int e = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols;
  const int c = (int)(i % cols);
  const int64_t off = transposed ? (int64_t)c * src_ld + r : r * src_ld + c;
  dst[(int64_t)e * rows * cols + i] = Elem<T>::load(src + (int64_t)e * src_batch + off);
}

template <typename T>
__global__ void k_from_f32(const float* __restrict__ src, int64_t rows, int32_t cols,
                           T* __restrict__ dst, int32_t dst_ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols;
  const int c = (int)(i % cols);
  Elem<T>::store(dst + r * dst_ld + c, src[i]);
}

// ---------------------------------------------------------------------------
// mixture-of-Normals quantiles (inference.py:42-84)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ x, int64_t n,
                                                        float* __restrict__ part) {
  __shared__ float smin[4], smax[4];
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

// bracket[0..1] = [min mu - 5 max sigma, max mu + 5 max sigma]
__global__ void k_bracket(const float* mean_part, int n_mean_part, const float* scale_part,
                          int n_scale_part, float* bracket) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lo = INFINITY, hi = -INFINITY, smax = -INFINITY;
  for (int i = 0; i < n_mean_part; ++i) {
    lo = fminf(lo, mean_part[2 * i]);
    hi = fmaxf(hi, mean_part[2 * i + 1]);
  }
  for (int i = 0; i < n_scale_part; ++i) smax = fmaxf(smax, scale_part[2 * i + 1]);
  bracket[0] = lo - 5.f * smax;
  bracket[1] = hi + 5.f * smax;
}

__device__ __forceinline__ float ndtrf(float z) { return 0.5f * erfcf(-z * 0.70710678118654752440f); }

__device__ __forceinline__ float mix_cdf(const float* __restrict__ means,
                                         const float* __restrict__ scales, int64_t n_members,
                                         int64_t n_rows, int64_t r, float x) {
  float acc = 0.f;
  for (int64_t m = 0; m < n_members; ++m) acc += ndtrf((x - means[m * n_rows + r]) / scales[m]);
  return acc / (float)n_members;
}

// Chandrupatla's bracketing root finder (the algorithm behind
// tfp.math.find_root_chandrupatla), one thread per row.
__global__ __launch_bounds__(256) void k_quantile_root(const float* __restrict__ means,
                                                       const float* __restrict__ scales,
                                                       int64_t n_members, int64_t n_rows,
                                                       const float* __restrict__ bracket, float q,
                                                       float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float vtol = 1e-5f, ptol = 1e-8f;
  float a = bracket[0], b = bracket[1];
  float fa = mix_cdf(means, scales, n_members, n_rows, r, a) - q;
  float fb = mix_cdf(means, scales, n_members, n_rows, r, b) - q;
  float c = a, fc = fa, t = 0.5f;
  float best = fabsf(fa) < fabsf(fb) ? a : b;
  float fbest = fabsf(fa) < fabsf(fb) ? fa : fb;
  for (int it = 0; it < 60 && fabsf(fbest) > vtol; ++it) {
    const float xn = a + t * (b - a);
    const float fn = mix_cdf(means, scales, n_members, n_rows, r, xn) - q;
    const bool same = (fn > 0.f) == (fa > 0.f) && (fn < 0.f) == (fa < 0.f);
    if (same) { c = a; fc = fa; }
    else { c = b; fc = fb; b = a; fb = fa; }
    a = xn; fa = fn;
    if (fabsf(fa) < fabsf(fb)) { best = a; fbest = fa; } else { best = b; fbest = fb; }
    const float tol = ptol / fabsf(b - c);
    if (tol > 0.5f || fbest == 0.f) break;
    const float xi = (a - b) / (c - b), phi = (fa - fb) / (fc - fb);
    if (phi * phi < xi && (1.f - phi) * (1.f - phi) < 1.f - xi) {
      t = (fa / (fb - fa)) * (fc / (fb - fc)) + ((c - a) / (b - a)) * (fa / (fc - fa)) * (fb / (fc - fb));
    } else {
      t = 0.5f;
    }
    t = fminf(fmaxf(t, tol), 1.f - tol);
    if (!(t == t)) t = 0.5f;
  }
  out[r] = best;
}

// moment-matched Normal quantile (inference.py:55-84)
__global__ __launch_bounds__(256) void k_quantile_approx(const float* __restrict__ means,
                                                         const float* __restrict__ scales,
                                                         int64_t n_members, int64_t n_rows, float q,
                                                         float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t m = 0; m < n_members; ++m) {
    const float mu = means[m * n_rows + r], sd = scales[m];
    s1 += mu;
    s2 += sd * sd + mu * mu;
  }
  const float mm = s1 / (float)n_members;
  const float var = s2 / (float)n_members - mm * mm;
  out[r] = mm + sqrtf(fmaxf(var, 0.f)) * normcdfinvf(q);
}

// ---------------------------------------------------------------------------
// count observation models: NB / ZINB forecast moments and mixture quantiles
// (inference.py:271-333, 497-502; TFP 0.24 NegativeBinomial / Mixture).
//   s = softplus(theta_shape) = aux[e][1], m = softplus(loc), tc = 1/s,
//   logits = -log s - log m  =>  e^logits = 1/(s m), sigmoid(-logits) = s m/(1+s m)
//   NB mean = tc e^logits = 1/(s^2 m), var = mean / sigmoid(-logits)
//   ZINB = Mixture([1-pi, pi], [NB, delta_0])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_moments(const float* __restrict__ loc,
                                                       const float* __restrict__ aux,
                                                       int64_t n_members, int64_t n_rows, int32_t obs,
                                                       float* __restrict__ means,
                                                       float* __restrict__ part) {
  __shared__ float smean[4], ssd[4];
  float mx_mean = 0.f, mx_sd = 0.f;
  const int64_t n = n_members * n_rows;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / n_rows;
    const float s = aux[e * 3 + 1];
    const float m = softplusf(loc[i]);
    const float sm = s * m;
    float mean = 1.0f / (s * sm);
    float var = mean * (1.0f + sm) / sm;
    if (obs == BNF_OBS_ZINB) {
      const float pi = aux[e * 3 + 2];
      const float zmean = (1.0f - pi) * mean;
      var = (1.0f - pi) * (var + mean * mean) - zmean * zmean;
      mean = zmean;
    }
    means[i] = mean;
    mx_mean = fmaxf(mx_mean, mean);
    mx_sd = fmaxf(mx_sd, sqrtf(fmaxf(var, 0.f)));
  }
  mx_mean = wave_max(mx_mean);
  mx_sd = wave_max(mx_sd);
  if ((threadIdx.x & 63) == 0) { smean[threadIdx.x >> 6] = mx_mean; ssd[threadIdx.x >> 6] = mx_sd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = fmaxf(fmaxf(smean[0], smean[1]), fmaxf(smean[2], smean[3]));
    part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(ssd[0], ssd[1]), fmaxf(ssd[2], ssd[3]));
  }
}

// bracket[2 + 0..1] = {max mean, max stddev}
__global__ void k_count_bracket(const float* part, int n_part, float* bracket) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mm = 0.f, ms = 0.f;
  for (int i = 0; i < n_part; ++i) { mm = fmaxf(mm, part[2 * i]); ms = fmaxf(ms, part[2 * i + 1]); }
  bracket[2] = mm;
  bracket[3] = ms;
}

// continued fraction of the incomplete beta function (modified Lentz), f64: the
// prefactor a log x + b log(1-x) - lbeta(a,b) cancels catastrophically in f32 for the
// b ~ 1e3..1e5 the bracket reaches.
__device__ inline double beta_cf(double a, double b, double x) {
  const double tiny = 1e-300, eps = 1e-13;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double hh = d;
#pragma unroll 1
  for (int m = 1; m <= 2000; ++m) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    hh *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    hh *= del;
    if (fabs(del - 1.0) < eps) break;
  }
  return hh;
}

// regularised incomplete beta I_x(a, b) with xc = 1 - x supplied separately
__device__ inline double betainc_xc(double a, double b, double x, double xc) {
  if (x <= 0.0) return 0.0;
  if (xc <= 0.0) return 1.0;
  const double lnpre = lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(xc);
  if (x < (a + 1.0) / (a + b + 2.0)) return exp(lnpre) * beta_cf(a, b, x) / a;
  return 1.0 - exp(lnpre) * beta_cf(b, a, xc) / b;
}

// mean over members of the (ZI)NB cdf at x >= 0, row r
__device__ inline float count_mix_cdf(const float* __restrict__ loc, const float* __restrict__ aux,
                                      int64_t n_members, int64_t n_rows, int32_t obs, int64_t r,
                                      float x) {
  double acc = 0.0;
  for (int64_t e = 0; e < n_members; ++e) {
    const double s = aux[e * 3 + 1];
    const double sm = s * (double)softplusf(loc[e * n_rows + r]);
    double F = betainc_xc(1.0 / s, 1.0 + (double)x, sm / (1.0 + sm), 1.0 / (1.0 + sm));
    if (obs == BNF_OBS_ZINB) {
      const double pi = aux[e * 3 + 2];
      F = pi + (1.0 - pi) * F;
    }
    acc += F;
  }
  return (float)(acc / (double)n_members);
}

// one thread per row: Chandrupatla on [0, max mean + 1.1 rsqrt(1-q) max sd], then ceil;
// rows whose mixture pmf(0) already exceeds q are 0 (inference.py:319-333).
__global__ __launch_bounds__(64) void k_count_quantile_root(const float* __restrict__ loc,
                                                            const float* __restrict__ aux,
                                                            int64_t n_members, int64_t n_rows,
                                                            int32_t obs,
                                                            const float* __restrict__ bracket, float q,
                                                            float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float vtol = 1e-5f, ptol = 1e-8f;
  float a = 0.f, b = bracket[2] + 1.1f * rsqrtf(1.0f - q) * bracket[3];
  float fa = count_mix_cdf(loc, aux, n_members, n_rows, obs, r, a) - q;
  if (fa > 0.f) { out[r] = 0.f; return; }
  float fb = count_mix_cdf(loc, aux, n_members, n_rows, obs, r, b) - q;
  float c = a, fc = fa, t = 0.5f;
  float best = fabsf(fa) < fabsf(fb) ? a : b;
  float fbest = fabsf(fa) < fabsf(fb) ? fa : fb;
  for (int it = 0; it < 60 && fabsf(fbest) > vtol; ++it) {
    const float xn = a + t * (b - a);
    const float fn = count_mix_cdf(loc, aux, n_members, n_rows, obs, r, xn) - q;
    const bool same = (fn > 0.f) == (fa > 0.f) && (fn < 0.f) == (fa < 0.f);
    if (same) { c = a; fc = fa; }
    else { c = b; fc = fb; b = a; fb = fa; }
    a = xn; fa = fn;
    if (fabsf(fa) < fabsf(fb)) { best = a; fbest = fa; } else { best = b; fbest = fb; }
    const float tol = ptol / fabsf(b - c);
    if (tol > 0.5f || fbest == 0.f) break;
    const float xi = (a - b) / (c - b), phi = (fa - fb) / (fc - fb);
    if (phi * phi < xi && (1.f - phi) * (1.f - phi) < 1.f - xi) {
      t = (fa / (fb - fa)) * (fc / (fb - fc)) + ((c - a) / (b - a)) * (fa / (fc - fa)) * (fb / (fc - fb));
    } else {
      t = 0.5f;
    }
    t = fminf(fmaxf(t, tol), 1.f - tol);
    if (!(t == t)) t = 0.5f;
  }
  out[r] = ceilf(best);
}

}  // namespace bnf
