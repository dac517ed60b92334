// bnf_fused.h -- row-panel fused forward + backward of one BayesNF train step.
//
// One persistent workgroup (4 waves) owns a panel of BM = 64 batch rows of one
// ensemble member at a time and carries it through the WHOLE network and back:
//
//   featurise -> [contraction -> scale/bias/activation] x L -> output dot -> likelihood
//   -> d(last activation) -> [contraction -> d activation] x (L-1) -> dH0 -> d features
//
// The panel's activations live in LDS (the A operand of the next contraction) and
// in the MFMA accumulators; nothing but what the weight-gradient contraction
// (gemm_tn) needs -- H_l and dZ_l, row-major -- is written to HBM.  Pre-activations
// of the intermediate layers are parked in a per-workgroup scratch slot (reused for
// every panel, so it stays in L2 / Infinity Cache) in the lane-private accumulator
// order and read back by the same lanes.  This replaces featurize, gemm_fwd*,
// row_loss, last_bwd, gemm_dgrad*, feat_bwd of the unfused pipeline (same maths:
// reference models.py:212-273, inference.py:558-569 and their autodiff).
//
// Geometry: wave w owns the 128*NT/4... columns [w*32*NT, (w+1)*32*NT) of the layer
// width W = 128*NT (NT = 1, 2, 4), all 64 rows: 2 x NT accumulators of 32x32.
// B operands (weights) are read straight from L2 in a fragment-major packing
// (one 1 KiB wave load per 32x16 fragment, no LDS); A operands from the LDS panel
// with a 16-byte-chunk XOR swizzle.
#pragma once

#include "bnf_gemm.h"
#include "bnf_kernels.h"

namespace bnf {

constexpr int kFusedBM = 64;
constexpr int kFusedThreads = 256;
constexpr int kFusedSmallBytes = (4 * kFusedBM + kFusedBM) * 4 + 256;   // vpart[4][64] + sdv (+ pad)

// dynamic LDS of k_fused_fwd_bwd: small arrays + max(activation panel + feature panel, f32 dH0 panel)
inline size_t fused_lds_bytes(int W, int Fp, int es) {
  return kFusedSmallBytes + (size_t)kFusedBM * (W * es + 16) + (size_t)kFusedBM * (Fp * es + 16);
}

struct FusedArgs {
  // static network facts (only what the MLP part needs)
  int32_t Fp, F, L;
  int32_t off_bias[BNF_MAX_LAYERS + 1], off_ls[BNF_MAX_LAYERS];
  int32_t off_ko, off_os, off_lns, off_law;
  const float* theta;
  int64_t theta_stride;
  int64_t B;
  int32_t n_tiles;          // row panels per member
  int32_t members;
  const void* Wfwd[BNF_MAX_LAYERS];   // fragment-major Bt[n][k] = K_l[k][n]
  const void* Wbwd[BNF_MAX_LAYERS];   // fragment-major Bt[n][k] = K_l[n][k]
  int64_t wfwd_batch[BNF_MAX_LAYERS]; // elements between members
  int64_t wbwd_batch[BNF_MAX_LAYERS];
  void* H[BNF_MAX_LAYERS];            // H[0] = features (Bp, Fp), written by k_featurize;
                                      // H[l] (Bp, W), l >= 1: inputs of layer l, written here
  void* dZ[BNF_MAX_LAYERS];           // (Bp, W), written here
  int64_t h0_batch, act_batch;
  const float* ybat;                  // (members, yb_batch) targets of the batch rows
  int64_t yb_batch;
  float* dH0t;                        // (members, Fp, ldt) f32, transposed (for k_feat_bwd)
  int64_t dh0_batch;
  int32_t ldt;
  void* spill;                        // (slots, L-1, 64*W) elements of T
  float* out;                         // (members, out_batch) network output
  int64_t out_batch;
  float* grad;
  int64_t grad_stride;
  float* loss;
  int64_t loss_stride;
  int32_t S;
  float loss_scale;
  float c;                            // (N/B) * lik_scale
  float* loss_raw;
  int32_t ablate;   // perf experiments (env BNF_ABLATE): 1 no contractions, 2 no tile epilogues,
                    // 8 no panel copies to HBM, 32 no dH0 contraction
};

// ---- fragment-major weight packing -------------------------------------------------
// Wp[nt][ks][lane][8] : lane l of fragment (nt, ks) holds Bt[nt*32 + (l&31)][ks*16 + (l>>5)*8 + 0..7]
//   which = 0 (forward):  Bt[n][k] = K[k][n], k < n_in, n < W          (n_tiles = W/32,  KS = n_pad/16)
//   which = 1 (backward): Bt[n][k] = K[n][k], n < n_in, k < W          (n_tiles = n_pad/32, KS = W/16)
template <typename T>
__global__ __launch_bounds__(256) void k_pack_fragments(const float* __restrict__ theta,
                                                        int64_t theta_stride, int32_t off_kernel,
                                                        int32_t n_in, int32_t n_pad, int32_t W,
                                                        int32_t which, T* __restrict__ out,
                                                        int64_t out_batch) {
  const int e = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (nt, ks, lane) per thread
  const int KS = (which == 0 ? n_pad : W) / 16;
  const int NTt = (which == 0 ? W : n_pad) / 32;
  if (idx >= (int64_t)NTt * KS * 64) return;
  const int lane = (int)(idx & 63);
  const int ks = (int)((idx >> 6) % KS);
  const int nt = (int)((idx >> 6) / KS);
  const int n = nt * 32 + (lane & 31);
  const int k0 = ks * 16 + (lane >> 5) * 8;
  const float* K = theta + (int64_t)e * theta_stride + off_kernel;  // (n_in, W) row-major
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    if (which == 0) v[j] = (k < n_in && n < W) ? K[(int64_t)k * W + n] : 0.f;
    else v[j] = (n < n_in && k < W) ? K[(int64_t)n * W + k] : 0.f;
  }
  store8(out + (int64_t)e * out_batch + idx * 8, v);
}

// ---- LDS panel addressing ------------------------------------------------------------
// Row pitch = cols * sizeof(T) + 16 bytes: one 16-byte chunk of padding per row rotates
// consecutive rows by one 16-byte slot of the 256-byte bank row, so the 16 lanes of a
// ds_read_b128 group (16 different rows, same chunk) hit 16 distinct slots -- and every
// element address stays lane_base + compile-time constant (an XOR swizzle here made hipcc
// hoist and spill hundreds of precomputed addresses).
__device__ __forceinline__ int panel_pitch(int cols, int es) { return cols * es + 16; }
// Makes a lane-invariant value opaque to the optimiser at this point: the (XOR-swizzled,
// hence non-affine) LDS addresses derived from it are recomputed where they are used
// instead of being hoisted out of the persistent panel loop and spilled by the hundred.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// Same for a float: placed right after a barrier it pins the arithmetic that depends on it
// BELOW the barrier (otherwise hipcc hoists a whole 128-element epilogue above the barrier
// and keeps -- i.e. spills -- every result until the stores are allowed to happen).
__device__ __forceinline__ float opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ int panel_off(int pitch, int row, int byte_in_row) {
  return row * pitch + byte_in_row;
}

template <typename T>
struct FusedOps;
template <>
struct FusedOps<bf16_t> {
  using Frag = Mma<bf16_t>::Frag;
  __device__ static __forceinline__ Frag lds_frag(const char* panel, int pitch, int row, int ks, int kg) {
    Frag f;
    f.v = *reinterpret_cast<const bf16x8*>(panel + row * pitch + ((ks * 2 + kg) << 4));
    return f;
  }
  __device__ static __forceinline__ Frag glb_frag(const bf16_t* wp, int64_t frag_index, int lane) {
    Frag f;
    f.v = *reinterpret_cast<const bf16x8*>(wp + (frag_index * 64 + lane) * 8);
    return f;
  }
};
template <>
struct FusedOps<float> {
  using Frag = Mma<float>::Frag;
  __device__ static __forceinline__ Frag lds_frag(const char* panel, int pitch, int row, int ks, int kg) {
    Frag f;
    const int c = ks * 4 + kg * 2;
    f.lo = *reinterpret_cast<const f32x4*>(panel + row * pitch + (c << 4));
    f.hi = *reinterpret_cast<const f32x4*>(panel + row * pitch + ((c + 1) << 4));
    return f;
  }
  __device__ static __forceinline__ Frag glb_frag(const float* wp, int64_t frag_index, int lane) {
    Frag f;
    const float* p = wp + (frag_index * 64 + lane) * 8;
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    return f;
  }
};

// acc[MT][NT] += panel (64 x K) . Wp  for this wave's NT column tiles (first = nt0)
template <typename T, int NT>
__device__ __forceinline__ void fused_gemm(f32x16 (&acc)[2][NT], const char* panel, int pitch,
                                           const T* wp, int KS, int nt0, int lane) {
  using O = FusedOps<T>;
  const int frow = lane & 31, kg = lane >> 5;
  typename O::Frag fb0[NT], fb1[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) fb0[j] = O::glb_frag(wp, (int64_t)(nt0 + j) * KS + 0, lane);
  for (int ks = 0; ks < KS; ks += 2) {
#pragma unroll
    for (int j = 0; j < NT; ++j) fb1[j] = O::glb_frag(wp, (int64_t)(nt0 + j) * KS + ks + 1, lane);
    {
      typename O::Frag fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = O::lds_frag(panel, pitch, i * 32 + frow, ks, kg);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::mma(acc[i][j], fa[i], fb0[j]);
    }
    if (ks + 2 < KS) {
#pragma unroll
      for (int j = 0; j < NT; ++j) fb0[j] = O::glb_frag(wp, (int64_t)(nt0 + j) * KS + ks + 2, lane);
    }
    {
      typename O::Frag fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = O::lds_frag(panel, pitch, i * 32 + frow, ks + 1, kg);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::mma(acc[i][j], fa[i], fb1[j]);
    }
  }
}

// copy the LDS panel (64 x cols of T, swizzled) to a row-major global array, rows < n_valid
template <typename T>
__device__ __forceinline__ void panel_to_global(const char* panel, int pitch, T* dst, int ld,
                                                int n_valid, int tid) {
  const int cpr = (pitch - 16) >> 4;   // data chunks per row (the last 16 bytes are padding)
  if (n_valid < 0) return;
  for (int q = tid; q < kFusedBM * cpr; q += kFusedThreads) {
    const int row = q / cpr, cc = q % cpr;
    if (row < n_valid)
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dst + (int64_t)row * ld) + cc * 16) =
          *reinterpret_cast<const u32x4*>(panel + row * pitch + (cc << 4));
  }
}


// ---- per-accumulator-tile epilogues as real (non-inlined) functions -------------------
// hipcc interleaves and hoists a fully unrolled 128-element epilogue across barriers and
// tiles until hundreds of values are live (thousands of spill bytes per lane); a call
// boundary per 32x32 tile bounds the live state to one tile (16 accumulator values).
struct TileSums {
  float a, b, c, d;
};
typedef __attribute__((address_space(3))) char lds_char;
template <typename T>
__device__ __forceinline__ T* lds_ptr(uint32_t off) {   // LDS byte offset -> pointer (address space inferred)
  return (T*)((lds_char*)(uintptr_t)off);
}

// intermediate forward layer: A = gamma (acc*scale + bias); H = act(A) -> LDS panel; A -> scratch slot
template <typename T, int PITCH>
__device__ __attribute__((noinline)) void tile_fwd_mid(f32x16 acc, float gamma, float scale, float bias,
                                                       float alpha, uint32_t lds_off, T* sp_lo, T* sp_hi) {
  constexpr bool FAST = Elem<T>::kFast;
  float av[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    av[r] = gamma * (acc[r] * scale + bias);
    const float h = act_fwd<FAST>(av[r], alpha);
    // rows 8*(r>>2) + (r&3) below the lane's base row
    Elem<T>::store(lds_ptr<T>(lds_off + (8 * (r >> 2) + (r & 3)) * PITCH), h);
  }
  float lo[8], hi[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { lo[r] = av[r]; hi[r] = av[8 + r]; }
  store8(sp_lo, lo);
  store8(sp_hi, hi);
}

// last layer, pass 2: returns {col sum dZ, col sum h*dv, sum dH*ediff, sum dA*a}
template <typename T, int PITCH>
__device__ __attribute__((noinline)) TileSums tile_last_bwd(f32x16 a_pre, f32x16 dvr, float kov, float gamma,
                                                            float alpha, uint32_t lds_off) {
  constexpr bool FAST = Elem<T>::kFast;
  TileSums s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float av = a_pre[r];
    const ActOut o = act_eval<FAST>(av, alpha);
    const float dh = dvr[r] * kov;
    s.c += dh * o.ediff;
    const float da = dh * o.dact;
    s.d += da * av;
    const float dz = gamma * da;
    s.a += dz;
    s.b += o.h * dvr[r];
    Elem<T>::store(lds_ptr<T>(lds_off + (8 * (r >> 2) + (r & 3)) * PITCH), dz);
  }
  return s;
}

// hidden layer backward: dH = acc/sqrt(W); reads the parked pre-activations; returns
// {col sum dZ, 0, sum dH*ediff, sum dA*a}
template <typename T, int PITCH>
__device__ __attribute__((noinline)) TileSums tile_bwd_mid(f32x16 acc, float inv_sw, float gamma, float alpha,
                                                           uint32_t lds_off, const T* sp_lo, const T* sp_hi) {
  constexpr bool FAST = Elem<T>::kFast;
  float lo[8], hi[8];
  load8(sp_lo, lo);
  load8(sp_hi, hi);
  TileSums s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float av = r < 8 ? lo[r & 7] : hi[r & 7];
    const float dh = acc[r] * inv_sw;
    const ActOut o = act_eval<FAST>(av, alpha);
    s.c += dh * o.ediff;
    const float da = dh * o.dact;
    s.d += da * av;
    const float dz = gamma * da;
    s.a += dz;
    Elem<T>::store(lds_ptr<T>(lds_off + (8 * (r >> 2) + (r & 3)) * PITCH), dz);
  }
  return s;
}

template <typename T, int NT>
__global__ __launch_bounds__(kFusedThreads, 2) void k_fused_fwd_bwd(const FusedArgs a) {
  constexpr int BM = kFusedBM, W = 128 * NT, ES = Elem<T>::kBytes;
  constexpr bool FAST = Elem<T>::kFast;
  constexpr int kXPitch = W * ES + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Fp = a.Fp, F = a.F, L = a.L;
  const int h0_pitch = Fp * ES + 16;
  float* vpart = reinterpret_cast<float*>(smem);     // [4 waves][BM] output-dot partials
  float* sdv = vpart + 4 * BM;                       // [BM]
  char* Xs = smem + kFusedSmallBytes;                // BM x W of T
  char* H0s = Xs + BM * kXPitch;                     // BM x Fp of T

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow_ = lane & 31, kg_ = lane >> 5;
  const int nt0 = wave * NT;                         // first column tile of this wave
  const int items = a.members * a.n_tiles;
  const float inv_sw = 1.0f / sqrtf((float)W);

  // XCD-aware walk (workgroup b runs on XCD b % 8, private 4 MiB L2): XCD x owns a contiguous
  // range of panels, i.e. of members, so that the weights its workgroups stream stay in its L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q8 = items >> 3, r8 = items & 7;
  const int first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int count = q8 + (xcd < r8 ? 1 : 0);
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;   // workgroups of this launch on this XCD
  for (int it = slot; it < count; it += nslot) {
    const int item = first + it;
    const int e = item / a.n_tiles, tile = item - e * a.n_tiles;
    const int64_t r0 = (int64_t)tile * BM;
    const int n_valid_true = (int)min((int64_t)BM, a.B - r0);
    const int n_valid = BNF_ABL(a, 8) ? 0 : n_valid_true;   // ablation: nothing copied, nothing "valid"
    const float* th = a.theta + (int64_t)e * a.theta_stride;
    float* gr = a.grad + (int64_t)e * a.grad_stride;
    const float alpha = sigmoidf(th[a.off_law]);

    // ---- feature panel (written by k_featurize, zero padded to Fp) -> LDS -------------
    {
      const T* src = reinterpret_cast<const T*>(a.H[0]) + (int64_t)e * a.h0_batch + r0 * Fp;
      const int cpr = (Fp * ES) >> 4;
      for (int q = tid; q < BM * cpr; q += kFusedThreads) {
        const int row = q / cpr, cc = q - row * cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < n_valid) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src + (int64_t)row * Fp) + cc * 16);
        *reinterpret_cast<u32x4*>(H0s + row * h0_pitch + cc * 16) = v;
      }
    }
    __syncthreads();

    f32x16 acc[2][NT];
    // =========================== forward layers ====================================
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      const int K = (l == 0) ? Fp : W;
      if (!BNF_ABL(a, 1))
        fused_gemm<T, NT>(acc, (l == 0) ? H0s : Xs, K * ES + 16,
                          reinterpret_cast<const T*>(a.Wfwd[l]) + (int64_t)e * a.wfwd_batch[l], K / 16, nt0,
                          lane);
      float gamma = softplusf(th[a.off_ls[l]]);
      float scale = 1.0f / sqrtf((float)((l == 0) ? F : W));
      const int frow = opaque(frow_), kg = opaque(kg_);
      if (l < L - 1) {
        if (l > 0) __syncthreads();  // every wave is done reading Xs as the A operand
        gamma = opaque(gamma);
        scale = opaque(scale);
        T* sp_base = reinterpret_cast<T*>(a.spill) + ((int64_t)blockIdx.x * (L - 1) + l) * (BM * W);
        const uint32_t xs_off = (uint32_t)(uintptr_t)Xs;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int col = (nt0 + j) * 32 + frow;
            const float bias = th[a.off_bias[l] + col];
            const int f = (i * NT + j) * 2;
            if (!BNF_ABL(a, 2)) tile_fwd_mid<T, kXPitch>(acc[i][j], gamma, scale, bias, alpha,
                                     xs_off + (i * 32 + 4 * kg) * kXPitch + col * ES,
                                     sp_base + ((int64_t)(wave * 4 * NT + f) * 64 + lane) * 8,
                                     sp_base + ((int64_t)(wave * 4 * NT + f + 1) * 64 + lane) * 8);
          }
        __syncthreads();
        panel_to_global<T>(Xs, kXPitch,
                           reinterpret_cast<T*>(a.H[l + 1]) + (int64_t)e * a.act_batch + r0 * W, W,
                           n_valid, tid);
      } else {
        // ---- last hidden layer: output dot, likelihood, d(last activation) ----------
        // pass 1: per-row partial dot products h . k_o over this wave's columns; the
        // accumulator keeps the pre-activation for pass 2
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float pd[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) pd[r] = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int col = (nt0 + j) * 32 + frow;
            const float bias = th[a.off_bias[l] + col];
            const float kov = th[a.off_ko + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float av = gamma * (acc[i][j][r] * scale + bias);
              pd[r] += act_fwd<FAST>(av, alpha) * kov;
              acc[i][j][r] = av;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float s0 = half_wave_sum_dpp(pd[r]);
            if (frow == 16) vpart[wave * BM + i * 32 + 8 * (r >> 2) + 4 * kg + (r & 3)] = s0;
          }
        }
        __syncthreads();
        // one thread per row: output, likelihood, d out  (models.py:269-273,157-164)
        if (tid < BM) {   // exactly wave 0
          const int row = tid;
          float t_ll = 0.f, t_dov = 0.f, t_dv = 0.f, t_lns = 0.f, dvv = 0.f;
          if (row < n_valid) {
            const float dot = vpart[0 * BM + row] + vpart[1 * BM + row] + vpart[2 * BM + row] +
                              vpart[3 * BM + row];
            const float gam_o = softplusf(th[a.off_os]);
            const float v = dot * inv_sw + th[a.off_bias[L]];
            const float out = gam_o * v;
            a.out[(int64_t)e * a.out_batch + r0 + row] = out;
            const float lns = th[a.off_lns];
            const float sigma = 0.01f + expf(lns);
            const float res = a.ybat[(int64_t)e * a.yb_batch + r0 + row] - out;
            const float z = res / sigma;
            t_ll = -0.5f * z * z - logf(sigma) - 0.918938533204672742f;
            const float dout = -a.c * res / (sigma * sigma);
            t_dov = dout * v;
            dvv = gam_o * dout;
            t_dv = dvv;
            t_lns = -a.c * (res * res / (sigma * sigma * sigma) - 1.0f / sigma) * expf(lns);
          }
          sdv[row] = dvv;
          t_ll = wave_sum(t_ll); t_dov = wave_sum(t_dov); t_dv = wave_sum(t_dv); t_lns = wave_sum(t_lns);
          if (tid == 0) {
            const float step_loss = -a.c * t_ll;
            atomicAdd(&a.loss[(int64_t)(e / a.S) * a.loss_stride], a.loss_scale * step_loss);
            if (a.loss_raw) atomicAdd(&a.loss_raw[e], step_loss);
            atomicAdd(&gr[a.off_os], sigmoidf(th[a.off_os]) * t_dov);
            atomicAdd(&gr[a.off_bias[L]], t_dv);
            atomicAdd(&gr[a.off_lns], t_lns);
          }
        }
        __syncthreads();
        // pass 2: dH = dv k_o / sqrt W ; dZ = gamma dH act'(A) ; d bias, d k_o (column sums),
        //         d alpha, d gamma (scalars)
        {
          float s_alpha = 0.f, s_gamma = 0.f;
          const uint32_t xs_off = (uint32_t)(uintptr_t)Xs;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x16 dvr;
#pragma unroll
            for (int r = 0; r < 16; ++r) dvr[r] = sdv[i * 32 + 8 * (r >> 2) + 4 * kg + (r & 3)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int col = (nt0 + j) * 32 + frow;
              const float kov = th[a.off_ko + col] * inv_sw;
              TileSums ts{0.f, 0.f, 0.f, 0.f};
              if (!BNF_ABL(a, 2)) ts = tile_last_bwd<T, kXPitch>(acc[i][j], dvr, kov, gamma, alpha,
                                                                  xs_off + (i * 32 + 4 * kg) * kXPitch + col * ES);
              s_alpha += ts.c;
              s_gamma += ts.d;
              float cs_b = ts.a, cs_k = ts.b;
              cs_b += __shfl_xor(cs_b, 32, 64);
              cs_k += __shfl_xor(cs_k, 32, 64);
              if (lane < 32) {
                atomicAdd(&gr[a.off_bias[l] + col], cs_b);
                atomicAdd(&gr[a.off_ko + col], cs_k * inv_sw);
              }
            }
          }
          s_alpha = wave_sum(s_alpha);
          s_gamma = wave_sum(s_gamma);
          if (lane == 0) {
            atomicAdd(&gr[a.off_law], alpha * (1.f - alpha) * s_alpha);
            atomicAdd(&gr[a.off_ls[l]], sigmoidf(th[a.off_ls[l]]) * s_gamma / gamma);
          }
        }
        __syncthreads();
        panel_to_global<T>(Xs, kXPitch, reinterpret_cast<T*>(a.dZ[l]) + (int64_t)e * a.act_batch + r0 * W,
                           W, n_valid, tid);
      }
    }

    // =========================== backward through the hidden layers ================
    for (int l = L - 1; l >= 1; --l) {
      // dH_{l-1} = dZ_l . K_l^T / sqrt W   (A operand = dZ_l panel in Xs)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      if (!BNF_ABL(a, 1))
        fused_gemm<T, NT>(acc, Xs, kXPitch,
                          reinterpret_cast<const T*>(a.Wbwd[l]) + (int64_t)e * a.wbwd_batch[l], W / 16, nt0,
                          lane);
      __syncthreads();  // all waves done with the dZ_l panel; it is overwritten with dZ_{l-1}
      const int frow = opaque(frow_), kg = opaque(kg_);
      const int lm = l - 1;
      const float gamma = softplusf(th[a.off_ls[lm]]);
      const float inv_sw_b = opaque(inv_sw);
      const T* sp_base = reinterpret_cast<const T*>(a.spill) + ((int64_t)blockIdx.x * (L - 1) + lm) * (BM * W);
      float s_alpha = 0.f, s_gamma = 0.f;
      const uint32_t xs_off = (uint32_t)(uintptr_t)Xs;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = (nt0 + j) * 32 + frow;
          const int f = (i * NT + j) * 2;
          TileSums ts{0.f, 0.f, 0.f, 0.f};
          if (!BNF_ABL(a, 2)) ts = tile_bwd_mid<T, kXPitch>(
              acc[i][j], inv_sw_b, gamma, alpha, xs_off + (i * 32 + 4 * kg) * kXPitch + col * ES,
              sp_base + ((int64_t)(wave * 4 * NT + f) * 64 + lane) * 8,
              sp_base + ((int64_t)(wave * 4 * NT + f + 1) * 64 + lane) * 8);
          s_alpha += ts.c;
          s_gamma += ts.d;
          float cs = ts.a;
          cs += __shfl_xor(cs, 32, 64);
          if (lane < 32) atomicAdd(&gr[a.off_bias[lm] + col], cs);
        }
      s_alpha = wave_sum(s_alpha);
      s_gamma = wave_sum(s_gamma);
      if (lane == 0) {
        atomicAdd(&gr[a.off_law], alpha * (1.f - alpha) * s_alpha);
        atomicAdd(&gr[a.off_ls[lm]], sigmoidf(th[a.off_ls[lm]]) * s_gamma / gamma);
      }
      __syncthreads();
      panel_to_global<T>(Xs, kXPitch, reinterpret_cast<T*>(a.dZ[lm]) + (int64_t)e * a.act_batch + r0 * W,
                         W, n_valid, tid);
    }

    // =========================== dH0^T = (dZ_0 . K_0^T / sqrt F)^T -> k_feat_bwd ========
    {
      const int frow = opaque(frow_), kg = opaque(kg_);
      const int ct = Fp / 32;                       // column tiles of dH0
      const int n_t = 2 * ct;                       // (row tile, column tile) pairs
      const float scale0 = 1.0f / sqrtf((float)F);
      const T* wp = reinterpret_cast<const T*>(a.Wbwd[0]) + (int64_t)e * a.wbwd_batch[0];
      const int KS = W / 16;
      float* dh0 = a.dH0t + (int64_t)e * a.dh0_batch;
      for (int t = wave; t < n_t && !BNF_ABL(a, 32); t += 4) {
        const int mi = t / ct, ni = t - mi * ct;
        f32x16 acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        for (int ks = 0; ks < KS; ++ks) {
          const typename FusedOps<T>::Frag fa = FusedOps<T>::lds_frag(Xs, kXPitch, mi * 32 + frow, ks, kg);
          const typename FusedOps<T>::Frag fb = FusedOps<T>::glb_frag(wp, (int64_t)ni * KS + ks, lane);
          Mma<T>::mma(acc0, fa, fb);
        }
        // lane: column n = ni*32 + frow of dH0, rows r0 + mi*32 + 8*rg + 4*kg + (0..3)
        float* col_ptr = dh0 + (int64_t)(ni * 32 + frow) * a.ldt + r0 + mi * 32 + 4 * kg;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          store4(col_ptr + 8 * rg, acc0[rg * 4] * scale0, acc0[rg * 4 + 1] * scale0,
                 acc0[rg * 4 + 2] * scale0, acc0[rg * 4 + 3] * scale0);
      }
    }
    __syncthreads();  // LDS is reused by the next panel
  }
}

}  // namespace bnf
