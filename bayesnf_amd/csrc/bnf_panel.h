// bnf_panel.h -- row-panel forward + backward of a BayesNF member in ONE kernel (bf16; the two-layer network is
// described here, DEEP adds the middle layers in the same pattern).  Instances (template <WN, RT, H0L, DEEP, CH, FP>):
//   <8, 4, H0L, ., 1, 64>   W = 512: 128-row panels                                   (C2, C3)
//   <8, 2, H0L, ., 2, 64>   W = 1024: 64-row panels, two 64-column slabs per wave      (C4)
//   <4, 2, true, ., 1, FP>  W = 256: 128-row panels of two 64-row blocks, FP = 64 | 128 (C1, C5)
//   <4, 4, false, ., 1, 64> W = 256 without the LDS feature panel: 256-row panels
// (+ a last flag F0 on the H0L forms: layer 0 folded into its contraction, see k_panel_fwd_bwd)
//
// A workgroup (8 waves, one per CU: 256 registers per lane) owns a panel of BM batch rows of one
// ensemble member and carries it through the network and back (reference models.py:212-273,
// inference.py:558-569 and their autodiff):
//
//   A0 = H0 K0        -> H1 = act(gamma0 (A0/sqrt F + b0))                  -> LDS panel + HBM
//   A1 = H1 K1        -> out = gamma_o (act(A1) k_o/sqrt W + b_o), likelihood, d out
//                     -> dZ1 = gamma1 (dv k_o/sqrt W) act'(A1)              -> LDS panel + HBM
//   dH1 = dZ1 K1^T    -> dZ0 = gamma0 dH1/sqrt W act'(A0)  (A0 recomputed)   -> LDS panel + HBM
//   dH0 = dZ0 K0^T /sqrt F                                                   -> HBM (f32, transposed)
//
// What leaves the chip is exactly what the weight-gradient contractions (gemm_tn) and the feature
// backward kernel read afterwards: H1, dZ1, dZ0 (bf16, row-major) and dH0^T -- the latter only for
// the variants that still run k_feat_bwd: the H0L forms finish the featurisation backward
// themselves from the dH0 tiles in registers and the feature panel in LDS (phase 9).  It replaces
// gemm_fwd_l0 + gemm_fwd_last + gemm_dgrad + gemm_dgrad0 of the layer pipeline (and their HBM
// round trips of H1 / dZ1 / dZ0 as contraction operands: 2.0 GB of the 8.0 GB per C2 step).
//
// Round 4 (profiles/r04_panel_ab.md): the phases of a panel do not overlap with one another and, inside a phase, every
// instruction of whatever kind costs its issue time (ablation clocks add up to the cycle) -- so the kernel got faster by
// ISSUING LESS: the weight fragments of the W x W contractions arrive by raw buffer loads (no VALU address arithmetic
// between the MFMAs: the first wave of a SIMD finishes a contraction in 10.5k cycles instead of 17k), contractions start
// from a zero C operand, the forward epilogues take elu + 1 as one median, the backward ones min(e, 1) of a pair by one
// packed clamp, dZ1's column quantities come from two accumulations instead of five instructions, wave sums are DPP, panel
// copies leave by buffer stores, and layer 0 is FOLDED (template flag F0): scale and bias inside its contraction, both of
// its epilogues on transposed tiles with 8-byte panel stores, d bias0 from the weight-gradient kernel.  C2, same box:
// panel kernel 1288 -> 1145 us, 131.8k -> 110k cycles per panel.
//
// Geometry (W = 64 WN, WN = 8 or 4): wave w = (row block rb = w / WN, column slab cs = w % WN) owns
// rows [128 rb, +128) x columns [64 cs, +64) of every BM x W activation of the panel
// (BM = 128 * 8 / WN): 4 x 2 MFMA 32x32 accumulators = 128 registers.  The A operand of every
// W-deep contraction is the panel in LDS (row pitch W*2 + 16 bytes: the 16 lanes of a ds_read_b128
// group hit 16 different 16-byte slots); the B operand -- the member's weights in fragment-major
// packing, one contiguous 1 KiB wave load per 32 x 16 fragment -- streams L2 -> VGPR through a
// ring of PD fragments per stream, never touching LDS: each weight fragment is consumed by exactly
// one wave, and a panel of BM rows re-reads the weights BM/64 times less often than the 64-row
// panels of gemm_nt<.., EPI_LAST, 1, 8>.  Measured ceilings this is designed against
// (profiles/r02a_panel_probe.txt): L2-resident streams reach 31 TB/s into VGPRs or LDS alike
// (51 B/clk/CU), and this contraction loop alone sustains 1.25-1.34 PFLOP/s (50-54 % of the
// bf16 MFMA peak) against 0.41-0.50 for the LDS-staged K loops of gemm_nt.
#pragma once

#include <type_traits>

#include "bnf_gemm.h"
#include "bnf_kernels.h"

namespace bnf {

struct PanelArgs {
  int32_t F, Fp, B, panels, members;
  int32_t Wt;                     // model width (fan-in of layer 1 and of the output layer); <= the kernel's W
  const float* theta;
  int64_t theta_stride;
  const float* scal;              // k_member_scalars table (kScalStride per member)
  int32_t n_layers;               // hidden layers L >= 2: layer 0 (features -> W), middle layers 1 .. L-2, last layer L-1
  int32_t off_bias[BNF_MAX_LAYERS], off_ls[BNF_MAX_LAYERS];   // Dense_l/bias, inv_sp_layer_scale_l of the hidden layers
  int32_t off_bias_out, off_ko, off_os, off_law;
  int32_t off_lns, off_shape, off_infl, obs;
  const bf16_t* H0;               // (members, Bp x Fp) features in A-fragment-major order (k_featurize H0f), rows >= B zero
  const bf16_t* H0rm;             // the same, row-major (Bp, Fp): staged into LDS by the H0L variant
  int64_t h0_batch;
  const bf16_t* Wf[BNF_MAX_LAYERS];   // fragment-major Bt[n][k] = K_l[k][n]   (W/32 x n_l/16 fragments; n_0 = Fp, else W)
  const bf16_t* Wb[BNF_MAX_LAYERS];   //                 Bt[n][k] = K_l[n][k]   (n_l/32 x W/16)
  int64_t w0_batch, w1_batch;     // elements between members: layer 0, every other layer
  bf16_t* Hout[BNF_MAX_LAYERS];   // Hout[l] = H_{l+1} = act(A_l), l = 0 .. L-2: (members, Bp, W) row-major (weight gradients)
  bf16_t* dZ[BNF_MAX_LAYERS];     // dZ[l], l = 0 .. L-1, likewise
  bf16_t* park[BNF_MAX_LAYERS];   // middle layers 1 .. L-2: t_l = A_l log2(e) as bf16 in the OWNING WAVE's accumulator order
                                  // ((member, panel, wave) x 16 chunks x 64 lanes x 8 values): written by the forward
                                  // epilogue, read back by the same lanes in the backward one -- 1 KiB per wave instruction
  int64_t act_batch;
  float* dH0t;                    // (members, Fp, ldt) f32
  int64_t dh0_batch;
  int32_t ldt;
  const float* ybat;              // (members, row_batch) targets of the batch rows
  int64_t row_batch;
  float* out;                     // (members, out_batch) network output
  int64_t out_batch;
  float* grad;
  int64_t grad_stride;
  float* loss;                    // loss[(e / S) * loss_stride] += loss_scale * step loss
  float* loss_raw;
  int64_t loss_stride;
  int32_t S;
  float loss_scale, lik_c;
  const StepState* st;            // graph replay: loss column offset
  unsigned long long* prof;       // -DBNF_ENABLE_ABLATE builds: per-workgroup phase clocks
  int32_t ablate;                 // perf experiments only (env BNF_ABLATE)
  // H0L variant: the featurisation backward (d feature scales, d log_scale_adjustment: SURVEY A.3,
  // k_feat_bwd) is finished here from the dH0 tiles in registers and the feature panel in LDS;
  // dH0^T is then not written at all.  fbmeta: per feature column 4 words {kind | group << 8 |
  // d1 << 16 | d2 << 24, partner | ucol << 8, coefficient (float bits), 0} followed by the
  // theta offsets of the BNF_MAX_GROUPS group scales (build_featbwd_meta in bnf_api.hip).
  const int32_t* fbmeta;
  int32_t off_lsa, n_groups, n_inputs, fb_in_group;   // fb_in_group: feature group of the raw inputs
  // -DBNF_PANEL_DK0=1 builds (experiment, profiles/r04_panel_ab.md): layer 0's weight gradient contracted here from the
  // dZ0 and feature panels in LDS and accumulated with f32 atomics; dZ0 is then not written and gemm_tn_skinny not run
  int32_t off_k0, dk0_fused;
  // fin (-DBNF_PANEL_FIN=1 builds + env BNF_PANEL_FIN=1; experiment, profiles/r04_panel_ab.md r04l): the H0L forms FEATURISE
  // their own rows (models.py:218-252: what k_featurize does, value for value) straight into the LDS feature panel and write
  // the row-major copy the layer-0 weight gradient reads; no k_featurize launch, no re-read of H0.  Measured: bit-identical
  // features, and 45 - 85 us per C2 step SLOWER than the separate kernel (one workgroup per CU has nothing to hide the
  // panel's serial start-up behind).  fcol: per padded feature column {kind | group << 8, a, b, 0}.
  int32_t fin, n_in, n_seas;
  const float* X; const float* stab; const float* y;
  const int32_t* fcol;
  bf16_t* H0out;
  RowSrc rs;
  // q8 (compute_dtype 'fp8'): the copies this kernel leaves for the weight-gradient contractions are FP8 -- Hout[l] as OCP
  // e4m3 (un-scaled), dZ[l] as OCP e5m2 divided by a per-member power of two s_dZ, written to qscale[member] for the
  // consumers (bnf_gemm8.h) -- row-major (Bp, W) BYTES in the same buffers.  The panels in LDS and every contraction
  // of this kernel stay bf16: the conversion happens where a block leaves for HBM (v_cvt_scalef32_pk_{fp8,bf8}_bf16:
  // one instruction per element pair, half the store instructions and bytes).
  int32_t q8;
  float* qscale;
  // c8 (round 6; compute_dtype 'fp8', the folded two-layer forms): the two W x W contractions of the panel run on the
  // block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4: K = 64 per instruction at twice the bf16 rate) -- weights
  // pre-packed as OCP e4m3 x 2^5 in K = 64 fragments (Wf8 / Wb8, k_pack_layers; the 2^-5 rides in the instruction's E8M0
  // scale), and the panels those contractions read live in LDS AS FP8 (row pitch W + 16 bytes): H_1 as e4m3, written by the
  // layer-0 forward epilogue (one v_cvt_pk_fp8_f32 per element pair where the bf16 form spends one v_cvt_pk_bf16_f32), dZ_L
  // as e5m2 / s_dZ.  The copies for the weight-gradient kernels are then plain byte copies of those panels.  Layer 0, its
  // backward-data and everything that is not a W x W contraction stay bf16 (the dZ_0 panel, the column-sum MFMAs).
  // (A first form converted the bf16 panel in registers on its way to the instruction -- every wave the same rows, 16
  // v_cvt_scalef32_pk per fragment: as slow as the bf16 contraction, profiles/r06_fp8_contract_ab.txt.)
  int32_t c8;
  const uint8_t* Wf8[BNF_MAX_LAYERS];   // [n / 32][k / 64][2][64 lanes][16 bytes]: lane (n % 32, kg) holds k = 64 ks + 32 kg + 0 .. 31
  const uint8_t* Wb8[BNF_MAX_LAYERS];
  int64_t w8_batch;
};
constexpr float kW8Scale = 32.f;          // packed fp8 weights are K x 2^5 (|K| < 14 stays finite; e4m3 is normal down to 2^-6 / 32)
constexpr int kW8ScaleE8M0 = 127 - 5;
constexpr int kFbNone = 0, kFbInput = 1, kFbFourier = 2, kFbInter = 3;

// ---- fragment-major weight packing -------------------------------------------------
// Wp[nt][ks][lane][8] : lane l of fragment (nt, ks) holds Bt[nt*32 + (l&31)][ks*16 + (l>>5)*8 + 0..7]
//   which = 0 (forward):  Bt[n][k] = K[k][n], k < n_in, n < W          (n_tiles = W/32,  KS = n_pad/16)
//   which = 1 (backward): Bt[n][k] = K[n][k], n < n_in, k < W          (n_tiles = n_pad/32, KS = W/16)
// Both layouts of every hidden layer's kernel in ONE launch: a workgroup stages a 64 x 64 tile of
// K (f32, coalesced 256-byte rows) in LDS and writes its 8 forward and 8 backward fragments (1 KiB
// each, 16 bytes per lane).  K is read once (one launch per layer and layout, each reading K --
// the backward one in 32-byte pieces: 4 x 18.6 us at C2 against 50 us for this launch).
struct PackJobs {
  int32_t n_layers, W;
  int32_t off_kernel[BNF_MAX_LAYERS], n_in[BNF_MAX_LAYERS], n_pad[BNF_MAX_LAYERS];
  int32_t tile0[BNF_MAX_LAYERS + 1];   // tile0[l]: first blockIdx.x of layer l
  void* wf[BNF_MAX_LAYERS]; void* wb[BNF_MAX_LAYERS];
  int64_t batch[BNF_MAX_LAYERS];
  // fold0 (the panel kernel's F0 forms): layer 0's FORWARD fragments carry gamma0 log2(e) / sqrt F, and the K rows F and
  // F + 1 hold gamma0 log2(e) b0 split into a bf16 hi and lo part (against the two ones columns of k_featurize)
  int32_t fold0, F0n, off_bias0, off_ls0;
  // c8 (PanelArgs): layers l >= 1 also leave as e4m3 x 2^5 fragments of K = 64 (one 64 x 64 tile = 2 forward + 2 backward
  // fragments of 2 KiB: one 16-byte piece per thread)
  void* wf8[BNF_MAX_LAYERS]; void* wb8[BNF_MAX_LAYERS];
  int64_t batch8;
};
// Workgroup 0 of a member also fills the member's row of the transformed-scalar table
// (k_member_scalars folded in: one launch and one dependent-launch gap less per step).
// the 8 forward and 8 backward fragments of the 64 x 64 tile of layer l's kernel staged in `tile` (origin k0, n0)
template <typename T, int NT = 256>
__device__ __forceinline__ void pack_tile_fragments(const float (&tile)[64][65], const PackJobs& jb, int l, int64_t e,
                                                    int k0, int n0, int tid, const float* th) {
  const int W = jb.W;
  const bool fold = jb.fold0 && l == 0;
  float fs = 1.f, fgb = 0.f;       // forward scale of layer 0's kernel, scale of its bias
  if (fold) {
    const float g0 = softplusf(th[jb.off_ls0]);
    fs = g0 * kLog2e / sqrtf((float)jb.F0n);
    fgb = g0 * kLog2e;
  }
  T* wf = (T*)jb.wf[l] + e * jb.batch[l];
  T* wb = (T*)jb.wb[l] + e * jb.batch[l];
  const int KSf = jb.n_pad[l] / 16, KSb = W / 16;
#pragma unroll
  for (int h = 0; h < 512 / NT; ++h) {
    const int q = tid + NT * h;             // 512 lane vectors per layout
    const int frag = q >> 6, fl = q & 63;
    float v[8];
    {  // forward: Bt[n][k] = K[k][n]; fragment (n/32, k/16), lane = n%32 + 32*((k%16)/8)
      const int nl = (frag >> 2) * 32 + (fl & 31), kl = (frag & 3) * 16 + (fl >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[kl + j][nl] * fs;
      if (fold && k0 + kl + 8 > jb.F0n && k0 + kl <= jb.F0n + 1) {   // this lane's 8 K rows include a bias row
        const float gbv = fgb * th[jb.off_bias0 + n0 + nl];
        const float hi = Elem<bf16_t>::round(gbv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (k0 + kl + j == jb.F0n) v[j] = hi;
          if (k0 + kl + j == jb.F0n + 1) v[j] = gbv - hi;
        }
      }
      const int64_t f = (int64_t)((n0 + nl) / 32) * KSf + (k0 + kl) / 16;
      store8(wf + (f * 64 + fl) * 8, v);
    }
    {  // backward: Bt[i][c] = K[i][c]; fragment (i/32, c/16), lane = i%32 + 32*((c%16)/8)
      const int il = (frag >> 2) * 32 + (fl & 31), cl = (frag & 3) * 16 + (fl >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[il][cl + j];
      const int64_t f = (int64_t)((k0 + il) / 32) * KSb + (n0 + cl) / 16;
      store8(wb + (f * 64 + fl) * 8, v);
    }
  }
  if (jb.wf8[l] && l >= 1) {
    // K = 64 fp8 fragments (PanelArgs.c8): item q = (layout, fragment of the tile, 16-byte piece, lane); W / 64 k steps per
    // 32-wide tile row in both layouts
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);   // MODE.FP16_OVFL: saturate
    const int KS64 = W / 64;
    for (int q = tid; q < 512; q += NT) {
      const int lay = q >> 8, fr = (q >> 7) & 1, pc = (q >> 6) & 1, fl = q & 63;
      const int ml = fr * 32 + (fl & 31), kb = (fl >> 5) * 32 + pc * 16;      // tile-local m (n or i) and first k (k or c)
      uint32_t w4[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float x[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) x[b] = (lay == 0 ? tile[kb + 4 * d + b][ml] : tile[ml][kb + 4 * d + b]) * kW8Scale;
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], pk, true);
        w4[d] = (uint32_t)pk;
      }
      uint8_t* base = (uint8_t*)(lay == 0 ? jb.wf8[l] : jb.wb8[l]) + e * jb.batch8;
      const int64_t f = lay == 0 ? (int64_t)((n0 + ml) / 32) * KS64 + k0 / 64 : (int64_t)((k0 + ml) / 32) * KS64 + n0 / 64;
      *reinterpret_cast<u32x4*>(base + f * 2048 + pc * 1024 + fl * 16) = u32x4{w4[0], w4[1], w4[2], w4[3]};
    }
  }
}
// grid of the two packing kernels: members x tiles in one dimension, XCD-aware -- the tiles next to each other along a
// kernel row run on the SAME XCD back to back, so the 128-byte lines two tiles share (a tile row is 256 bytes starting
// 12 bytes into a line: kernels start at p = 3 mod 4) meet in one L2 instead of being fetched by, or leaving, two XCDs
// as partial lines (C3/8: sample + pack 194 -> 152 us, pack alone 116 -> 110 us)
struct PackItem { int e, l, k0, n0; bool first; };
__device__ __forceinline__ PackItem pack_item_of(const PackJobs& jb) {
  const int n_tiles = jb.tile0[jb.n_layers];
  const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = (int)(w % (uint32_t)n_tiles);
  PackItem it;
  it.e = (int)(w / (uint32_t)n_tiles);
  int l = 0;
  while (l + 1 < jb.n_layers && bx >= jb.tile0[l + 1]) ++l;
  const int t = bx - jb.tile0[l], tn = jb.W / 64;
  it.l = l; it.k0 = (t / tn) * 64; it.n0 = (t % tn) * 64;      // tile origin: K rows (fan-in), K columns
  it.first = bx == 0 && threadIdx.x == 0;
  return it;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pack_layers(const float* __restrict__ theta, int64_t theta_stride,
                                                     PackJobs jb, NetDev nd, float* __restrict__ scal) {
  __shared__ float tile[64][65];
  const PackItem it = pack_item_of(jb);
  const float* th = theta + (int64_t)it.e * theta_stride;
  if (scal && it.first) member_scalars_row(nd, th, scal + (int64_t)it.e * kScalStride);
  const float* K = th + jb.off_kernel[it.l];
  const int tid = threadIdx.x;
  // 16 bytes per lane (K starts at an arbitrary 4-byte aligned offset of the member's parameters: load4u);
  // sixteen lanes cover a 256-byte row of the tile, a wave four rows per access
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (it.k0 + r < jb.n_in[it.l]) load4u(K + (int64_t)(it.k0 + r) * jb.W + it.n0 + c, 4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][c + j] = v[j];
  }
  __syncthreads();
  pack_tile_fragments<T>(tile, jb, it.l, it.e, it.k0, it.n0, tid, th);
}

// VI: sample AND pack in one pass over (mu, rho) of the hidden Dense kernels (inference.py:687-720 draws
// z = mu + sigma eps leaf by leaf; the forward contraction then wants bf16 fragments of it).  A workgroup owns a
// 64 x 64 tile of one member's kernel for ALL S samples: mu and sigma stay in registers (16 + 16 per lane), every
// sample costs one Philox call per quad (vi_eps_quad; the tile's quads are whole: kEpsQuadPhase), goes out as f32
// (theta_c: k_vi_adam recovers eps from it) and through an LDS tile (two in turn: one barrier per sample) into the
// 8 + 8 fragments of network e S + s.  Against k_vi_sample + k_pack_layers this never re-reads the S x P samples
// (C3/8: 260 MB of 880 MB) and is one launch less.  Everything else of the parameter vector (biases, scalars, the
// output layer) is k_vi_sample's, launched BEFORE this kernel: the layer-0 fold and the member-scalar table read those
// samples here.  Measured at C3/8 (profiles/r04_panel_ab.md r04p): 152 us against 82 + 110; neither the generator
// (7 Philox rounds: same time) nor HBM (4.1 TB/s) nor the barriers (a barrier-free form on wave-private 32 x 32
// sub-tiles: 157 us) bounds it; more waves per SIMD spill (104 registers).
#ifndef BNF_VI_Z_NT
#define BNF_VI_Z_NT 0   // 1: the f32 samples leave non-temporally (measured: 190 vs 152 us at C3/8 -- partial lines)
#endif
struct ViSampleArgs {
  const float* mu; const float* rho;
  int32_t P, S;
  uint64_t seed; int64_t member_offset; uint64_t step;
  float* z;                 // (members * S, P): the samples of the other leaves (k_vi_sample's); this kernel's own go there
  int32_t write_z;          // only when something reads them back (k_vi_adam does not with the device generator)
};
#ifndef BNF_VI_SP_THREADS
#define BNF_VI_SP_THREADS 512   // 512: two quads per lane, 62 registers (256: four, 104; C3/8 115 vs 108 us)
#endif
template <typename T, int NT>
__global__ __launch_bounds__(NT) void k_vi_sample_pack(ViSampleArgs a, PackJobs jb, NetDev nd, float* __restrict__ scal) {
  constexpr int NI = 1024 / NT;   // quads per lane
  __shared__ float tile[2][64][65];
  const PackItem it = pack_item_of(jb);
  const int e = it.e, l = it.l, tid = threadIdx.x;
  const int c = (tid & 15) * 4;
  float m[NI][4], sg[NI][4];
  int32_t pq[NI];            // parameter index of this lane's quad in row i (-1: a pad row of layer 0)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (tid >> 4) + (NT / 16) * i;
    pq[i] = (it.k0 + r < jb.n_in[l]) ? jb.off_kernel[l] + (it.k0 + r) * jb.W + it.n0 + c : -1;
    if (pq[i] >= 0) {
      load4u(a.mu + (int64_t)e * a.P + pq[i], 4, m[i]);
      load4u(a.rho + (int64_t)e * a.P + pq[i], 4, sg[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) sg[i][j] = vi_sigma(sg[i][j]);
    }
  }
  for (int s = 0; s < a.S; ++s) {
    const int64_t en = (int64_t)e * a.S + s;
    float* zn = a.z + en * a.P;
    float (&tl)[64][65] = tile[s & 1];
    if (scal && it.first) member_scalars_row(nd, zn, scal + en * kScalStride);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (tid >> 4) + (NT / 16) * i;
      float zv[4] = {0.f, 0.f, 0.f, 0.f};
      if (pq[i] >= 0) {
        const Normal4 n = vi_eps_quad(a.seed, (uint32_t)(a.member_offset + e), (uint32_t)s,
                                      (uint32_t)(pq[i] + kEpsQuadPhase) >> 2, a.step, STREAM_VI_EPS);
#pragma unroll
        for (int j = 0; j < 4; ++j) zv[j] = m[i][j] + sg[i][j] * n.v[j];
        if (a.write_z) store4u<BNF_VI_Z_NT != 0>(zn + pq[i], 4, zv);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tl[r][c + j] = zv[j];
    }
    __syncthreads();
    pack_tile_fragments<T, NT>(tl, jb, l, en, it.k0, it.n0, tid, zn);
  }
}

#ifndef BNF_EPI_FENCE_EVERY
#define BNF_EPI_FENCE_EVERY 1  // scheduling fence after every 4-row group of an epilogue (2: after every second one)
#endif
#ifndef BNF_PANEL_PRIO
#define BNF_PANEL_PRIO 0
#endif
#ifndef BNF_PANEL_NT
#define BNF_PANEL_NT 1   // H1 / dZ1 / dZ0 leave with non-temporal stores (they are read once, by the weight-gradient kernels)
#endif
#ifndef BNF_PANEL_PD
#define BNF_PANEL_PD 4
#endif
// round-4 steps of the panel kernel, each behind its own compile-time switch for same-box A/B builds
// (profiles/r04_panel_ab.md):
#ifndef BNF_PANEL_FWDS
#define BNF_PANEL_FWDS 1     // forward epilogues: elu + 1 as ONE median per element (act_fwd_core2)
#endif
#ifndef BNF_PANEL_DPPSUM
#define BNF_PANEL_DPPSUM 1   // wave-wide sums by DPP + v_readlane instead of six ds_bpermute round trips
#endif
#ifndef BNF_PANEL_ZPEEL
#define BNF_PANEL_ZPEEL 1    // contractions start from a zero C operand (no 128 v_mov per contraction and wave)
#endif
#ifndef BNF_PANEL_PRE0
#define BNF_PANEL_PRE0 0     // (measured: no gain, r04a) layer-0 weights and biases requested BEFORE the barrier that completes the staged feature panel
#endif
#ifndef BNF_PANEL_SADDR
#define BNF_PANEL_SADDR 1    // weight fragments of the W x W contractions: uniform base in SGPRs + the lane's 32-bit offset (no 64-bit VALU address per load)
#endif
#ifndef BNF_PANEL_CPRIO
#define BNF_PANEL_CPRIO 1    // (r04d: -0.4 %) 1: the second-dispatched waves run the contractions at priority 1; 2: the first-dispatched ones do
#endif
#ifndef BNF_PANEL_FIN
#define BNF_PANEL_FIN 0      // experiment: the H0L forms featurise their own rows; -DBNF_PANEL_FIN=1 + env BNF_PANEL_FIN=1
#endif
#ifndef BNF_PANEL_DK0
#define BNF_PANEL_DK0 0      // experiment: dK_0 = H0^T dZ0 inside the panel kernel (f32 atomics), env BNF_PANEL_DK0=1 at run time
#endif
#ifndef BNF_PANEL_PKCLAMP
#define BNF_PANEL_PKCLAMP 1  // backward epilogues: min(e, 1) of an element pair by one packed multiply with the clamp modifier
#endif
#ifndef BNF_PANEL_L1T
#define BNF_PANEL_L1T 1      // round 5: the LAST hidden layer on transposed tiles with its activation evaluated ONCE (see k_panel_fwd_bwd)
#endif
#ifndef BNF_PANEL_FAIR
#define BNF_PANEL_FAIR 0     // round 6 experiment (measured: no gain, profiles/r06_panel_ab.md): the two waves of a SIMD take turns at priority through the VALU-only epilogues (see fair_prio)
#endif
constexpr int kPanelPD = BNF_PANEL_PD;       // weight fragments in flight per stream; must divide W / 16 (2: +2 % panel time, 8: equal -- gpurun_out/r03ar)

// RT = 32-row tiles per wave (4: one workgroup per CU, 256 registers; 2: two workgroups per CU, 128)
__host__ __device__ constexpr int panel_rows(int wn, int rt) { return 32 * rt * (8 / wn); }
// h0l: the feature panel (Fp = 64: BM x 144 bytes) is staged in LDS as well -- fits for W = 512
// ch: 64-column slabs per wave (1: W = 64 wn; 2: W = 128 wn -- the width-1024 variant)
// fp: padded feature count of the LDS feature panel (64, or 128 with its own scratch for the fused featurisation
// backward); the panel image is also the waves' row-dot scratch (8 waves x 64 rows x 36 floats)
__host__ __device__ constexpr int panel_lds_bytes(int wn, int rt, bool h0l, int ch = 1, int fp = 64) {
  const int W = 64 * wn * ch, BM = panel_rows(wn, rt), RB = 8 / wn;
  const int image = BM * (W * 2 + 16), dots = 8 * 64 * 36 * 4;
  const bool own_fb = fp > 64 || RB > 1 || W < 512;   // (else the sums of the fused featurisation backward fit in s_col)
  return (image > dots ? image : dots) + (BM * wn * ch + BM + 2 * RB * W + 128) * 4 +
         (h0l ? BM * (fp * 2 + 16) + (own_fb ? 2 * (BM / 32) * fp * 4 : 0) : 0);
}

// Makes a lane value opaque to the optimiser at this point: everything derived from it (fragment
// rows, LDS / global addresses) is recomputed where a phase starts instead of being computed once,
// kept alive across the contractions (which need all 256 registers) and spilled -- a spill
// reload is a scratch load, and the s_waitcnt vmcnt(0) it needs also drains the prefetched
// operands and the panel stores in flight (measured: 16k of an epilogue's 42k cycles).
__device__ __forceinline__ int opaque_lane(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ ActCore2 panel_act_core2(f32x2 t) {   // the backward epilogues' activation core
#if BNF_PANEL_PKCLAMP
  return act_core2_pkclamp(t);
#else
  return act_core2(t);
#endif
}
__device__ __forceinline__ float panel_wave_sum(float v) {
#if BNF_PANEL_DPPSUM
  return wave_sum_dpp(v);
#else
  return wave_sum(v);
#endif
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains
// vmcnt, i.e. would wait for the panel copies to HBM (128 KiB per workgroup) at every phase change
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// acc[4][2] += P[rows of this wave][0 .. 16 KS) . Bt-fragments (two streams: column tiles nt0, nt0 + 1)
// A fragments from the LDS panel (next step's issued before this step's MFMAs), B fragments from
// global memory through a PD-deep register ring per stream.
// `side(it)` runs once per outer iteration (KS / PD of them) between MFMA groups: the caller
// spreads the copy of the (static) panel to HBM over the contraction, so that its store issue
// (~13 B/clk/CU, ~10k cycles per 128 KiB panel when done in one burst) hides under the MFMAs.
// The first PD weight fragments of both streams: issued by the caller BEFORE the barrier that completes the
// panel (the weights do not depend on it), so that their L2 latency runs under the tail of the previous
// phase and the barrier wait instead of in front of the first MFMA.
struct PanelRing {
  bf16x8 fb[kPanelPD][2];
};
// One weight fragment (16 bytes per lane) at the wave-uniform address wp + soff, lane offset loff.  BNF_PANEL_SADDR: as a
// raw buffer load -- the resource (base, in SGPRs) is made once per contraction, the fragment index goes into the scalar
// offset and the lane offset into the 32-bit VGPR offset: no VALU address arithmetic per load (the flat-pointer form
// costs a 64-bit v_lshl_add_u64 per load, four VALU slots per k step and wave next to eight MFMAs).
struct PanelW {
#if BNF_PANEL_SADDR
  __amdgpu_buffer_rsrc_t rsrc;
#endif
  const char* base;
};
__device__ __forceinline__ PanelW panel_wbase(const char* wp) {
  PanelW w;
  w.base = wp;
#if BNF_PANEL_SADDR
  // raw buffer (stride 0), 2 GiB window, gfx9 DATA_FORMAT = 32 (word 3 = 0x00020000)
  w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wp), 0, 0x7fffffff, 0x00020000);
#endif
  return w;
}
__device__ __forceinline__ bf16x8 panel_wload(const PanelW& w, uint32_t soff, uint32_t loff) {
#if BNF_PANEL_SADDR
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, loff, soff, 0));
#else
  return *reinterpret_cast<const bf16x8*>(w.base + soff + loff);
#endif
}
__device__ __forceinline__ void panel_prefetch(PanelRing& ring, const char* wp, int nt0, int KS, int lane) {
  const PanelW w = panel_wbase(wp);
  const uint32_t o0 = (uint32_t)nt0 * (uint32_t)KS * 1024u, o1 = o0 + (uint32_t)KS * 1024u;   // uniform
  const uint32_t loff = (uint32_t)lane * 16u;
#pragma unroll
  for (int p = 0; p < kPanelPD; ++p) {
    ring.fb[p][0] = panel_wload(w, o0 + p * 1024u, loff);
    ring.fb[p][1] = panel_wload(w, o1 + p * 1024u, loff);
  }
}
// ZERO: the accumulators are written, not accumulated into -- the first k step's MFMAs take a literal zero C operand
// (an inline constant), which saves the 16 RT v_mov per slab a zeroed accumulator costs before the first MFMA can issue.
// SWAP: the operands in swapped roles (weights as A, panel as B): acc[i][j] holds the TRANSPOSED tile -- lane <-> panel
// row i * 32 + lane % 32, register r <-> column j * 32 + 8 (r / 4) + 4 (lane / 32) + r % 4.
template <int KPITCH_B, int RT, bool ZERO, bool SWAP, typename Side>
__device__ __forceinline__ void panel_contract(f32x16 (&acc)[RT][2], const char* prow, const char* wp, int nt0, int KS,
                                               int lane, PanelRing& ring, Side side) {
  constexpr int PD = kPanelPD;
  const PanelW w = panel_wbase(wp);
  const uint32_t o0 = (uint32_t)nt0 * (uint32_t)KS * 1024u, o1 = o0 + (uint32_t)KS * 1024u;   // uniform
  const uint32_t loff = (uint32_t)lane * 16u;
  bf16x8 (&fb)[PD][2] = ring.fb;
  // rows frow + 32 i: TWO base registers keep every offset inside the 16-bit immediate of ds_read_b128 (the second one made
  // opaque, or hipcc re-derives it with a v_add_u32 per read -- VALU issue slots next to the MFMAs are what this loop lacks)
  int hi_off = 64 * KPITCH_B;
  asm volatile("" : "+v"(hi_off));          // (an opaque OFFSET: an opaque pointer would lose the LDS address space)
  const char* prow_hi = prow + hi_off;
  auto load_a = [&](bf16x8 (&fa)[RT], int ks) {
    const int ko = ks * 32;
#pragma unroll
    for (int i = 0; i < RT; ++i)
      fa[i] = *reinterpret_cast<const bf16x8*>(((i >> 1) ? prow_hi : prow) + (i & 1) * 32 * KPITCH_B + ko);
  };
  bf16x8 fa[2][RT];
  load_a(fa[0], 0);
  auto group = [&](int ks0, auto first_tag) {
    constexpr bool kFirst = decltype(first_tag)::value;
    side(ks0 / PD);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int cur = p & 1;
      load_a(fa[cur ^ 1], min(ks0 + p + 1, KS - 1));
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        if constexpr (kFirst) {
          if (p == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][0] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[p][0], fa[cur][i], z, 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][0], z, 0, 0, 0);
            acc[i][1] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[p][1], fa[cur][i], z, 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][1], z, 0, 0, 0);
            continue;
          }
        }
        acc[i][0] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[p][0], fa[cur][i], acc[i][0], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][0], acc[i][0], 0, 0, 0);
        acc[i][1] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[p][1], fa[cur][i], acc[i][1], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][1], acc[i][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int kn = min(ks0 + PD + p, KS - 1);    // (the tail re-reads the last fragment: unused)
      fb[p][0] = panel_wload(w, o0 + (uint32_t)kn * 1024u, loff);
      fb[p][1] = panel_wload(w, o1 + (uint32_t)kn * 1024u, loff);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (ZERO) group(0, std::true_type{});
#pragma unroll 1
  for (int ks0 = ZERO ? PD : 0; ks0 < KS; ks0 += PD) group(ks0, std::false_type{});
}

// The same contraction on the block-scaled fp8 MFMA (PanelArgs.c8): acc[RT][2] = P8[rows of this wave][0 .. 64 KS64) . W8
// fragments, K = 64 per instruction.  The panel is FP8 in LDS (row pitch KPITCH8 = W + 16 bytes: the 16 lanes of a
// ds_read_b128 group -- 16 rows -- fall on 64 different banks): a lane's 32 operand bytes are two ds_read_b128 of its own
// row.  Weight fragments (e4m3 x 2^5, scale byte 127 - 5): two 16-byte pieces per lane, one k step ahead in registers.
// Byte idx of lane (m, kg) <-> k = 32 kg + idx in BOTH operands (the instruction pairs equal (kg, idx): bnf_gemm8.h).
// BF8: the panel holds backward signals, e5m2 of dZ / s_dZ, and `pscale` (the E8M0 byte of s_dZ) puts the factor back.
// Accumulator layout = the bf16 instruction's.
typedef int pc8_i32x8 __attribute__((ext_vector_type(8)));
template <int KPITCH8, int RT, bool SWAP, bool BF8>
__device__ __forceinline__ void panel_contract8(f32x16 (&acc)[RT][2], const char* prow8, const uint8_t* wp8, int nt0, int KS64,
                                                int lane, int pscale) {
  // prow8: this lane's row of the panel + kg * 32 bytes; rows frow + 32 i
  const PanelW w = panel_wbase(reinterpret_cast<const char*>(wp8));
  const uint32_t o0 = (uint32_t)nt0 * (uint32_t)KS64 * 2048u, o1 = o0 + (uint32_t)KS64 * 2048u;   // uniform
  const uint32_t loff = (uint32_t)lane * 16u;
  auto wload = [&](uint32_t o) {
    const u32x4 a = __builtin_bit_cast(u32x4, panel_wload(w, o, loff));
    const u32x4 b = __builtin_bit_cast(u32x4, panel_wload(w, o + 1024u, loff));
    return pc8_i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  int hi_off = 64 * KPITCH8;
  asm volatile("" : "+v"(hi_off));
  const char* prow_hi = prow8 + hi_off;
  auto load_a = [&](int i, int ks) {
    const char* p = ((i >> 1) ? prow_hi : prow8) + (i & 1) * 32 * KPITCH8 + ks * 64;
    const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 16);
    return pc8_i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
  pc8_i32x8 fb[2][2], fa[2][RT];
  fb[0][0] = wload(o0); fb[0][1] = wload(o1);
#pragma unroll
  for (int i = 0; i < RT; ++i) fa[0][i] = load_a(i, 0);
  auto step = [&](int ks, auto first_tag, auto cur_tag) {
    constexpr bool kFirst = decltype(first_tag)::value;
    constexpr int cur = decltype(cur_tag)::value;
    const int kn = min(ks + 1, KS64 - 1);                         // (the tail re-reads the last fragment: unused)
    fb[cur ^ 1][0] = wload(o0 + (uint32_t)kn * 2048u);
    fb[cur ^ 1][1] = wload(o1 + (uint32_t)kn * 2048u);
#pragma unroll
    for (int i = 0; i < RT; ++i) fa[cur ^ 1][i] = load_a(i, kn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // formats: 0 = e4m3, 1 = e5m2; weights carry 2^5 (scale byte 127 - 5), the panel operand s_dZ (pscale)
        if constexpr (kFirst) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[cur][j], fa[cur][i], z, 0, BF8 ? 1 : 0, 0, kW8ScaleE8M0, 0, pscale)
                           : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[cur][i], fb[cur][j], z, BF8 ? 1 : 0, 0, 0, pscale, 0, kW8ScaleE8M0);
        } else {
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[cur][j], fa[cur][i], acc[i][j], 0, BF8 ? 1 : 0, 0, kW8ScaleE8M0, 0, pscale)
                           : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[cur][i], fb[cur][j], acc[i][j], BF8 ? 1 : 0, 0, 0, pscale, 0, kW8ScaleE8M0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (KS64 is even: W = 256 / 512) two k steps per trip so that the fragment buffers alternate without copies
  step(0, std::true_type{}, std::integral_constant<int, 0>{});
  step(1, std::false_type{}, std::integral_constant<int, 1>{});
#pragma unroll 1
  for (int ks = 2; ks < KS64; ks += 2) {
    step(ks, std::false_type{}, std::integral_constant<int, 0>{});
    step(ks + 1, std::false_type{}, std::integral_constant<int, 1>{});
  }
}

// Layer-0 contraction of one 32-row block x this wave's 64 columns, both operands from global
// memory, both fragment-major (one contiguous 1 KiB wave load per 32 x 16 fragment).
// One 32 x 32 output tile at a time (16 accumulator registers: the backward epilogue runs with the
// 128 dH accumulators live); the operands of FOUR k steps travel together (8 fragments, 32
// registers): the caller issues the
// loads of the next group right after the MFMAs that consumed the current one, so that they fly
// during the block's (long, VALU-only) epilogue instead of being waited for one k step at a time
// (measured: 16-20k of a panel's 170k cycles were exposed L2 latency with a one-step prefetch).
struct L0Blk {
  bf16x8 a[4], b[4];
};
__device__ __forceinline__ void l0_load(L0Blk& bk, const char* h0row, const char* w, int ks) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    bk.a[u] = *reinterpret_cast<const bf16x8*>(h0row + (size_t)(ks + u) * 1024);
    bk.b[u] = *reinterpret_cast<const bf16x8*>(w + (size_t)(ks + u) * 1024);
  }
}
__device__ __forceinline__ void l0_mma(f32x16& a0, const L0Blk& bk) {
#pragma unroll
  for (int u = 0; u < 4; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bk.a[u], bk.b[u], a0, 0, 0, 0);
}

// H0L (needs Fp = 64 and the LDS room, i.e. W = 512): the layer-0 contractions (forward, and the
// recomputation of A0 in the backward epilogue) read their A fragments from a feature panel staged
// in LDS and keep the wave's eight weight fragments in registers for the whole phase.  Without it
// every 32 x 32 tile re-fetches 8 KiB of operands through the vector memory path -- 768 KiB per
// panel over both passes, as much as the two W x W contractions stream -- and those phases were
// bound by that stream (~10k of a panel's 160k cycles each).
// DEEP = false: exactly two hidden layers (the benchmark shape): the middle-layer loops are compiled out and the
// last layer's index is a constant (the run-time-depth form cost it 0.8 % at C2, same box).
// CH = 2: every wave owns TWO 64-column slabs, carried one after the other through every phase (the width-1024
// variant: 8 waves x 2 x 64 columns, 64-row panels so that the panel still fits in LDS; twice the weight stream per
// MFMA of the 128-row form).
// FP: padded feature count the H0L code is compiled for (64; 128 = the W = 256 form: 128-row panels, two row blocks).
// F0 (needs H0L; round 4): layer 0's scale and bias are FOLDED into its contraction -- the forward-packed weights carry
// gamma0 log2(e) / sqrt F and two extra K rows hold the bias (bf16 hi + lo) against two columns of ones that
// k_featurize writes behind the F features (needs F + 2 <= FP) -- so t0 = A0 log2(e) IS the accumulator, in the forward
// pass and in the backward recomputation, and d bias0 comes out of the layer-0 weight-gradient kernel as the row of
// the ones column.  With no per-column constant left in the two layer-0 epilogues, both run with the MFMA operand
// roles SWAPPED (weights as A, panel rows as B): a lane then owns ONE row and four consecutive hidden units per
// register group, and the bf16 panel store is one ds_write_b64 per four elements instead of four ds_write_b16.
template <int WN, int RT, bool H0L, bool DEEP = false, int CH = 1, int FP = 64, bool F0 = false, bool C8 = false>
__global__ __launch_bounds__(512, (WN == 8 && RT * CH == 2) ? 4 : 2) void k_panel_fwd_bwd(const PanelArgs a) {   // (4: the r04s experiment form only)
  static_assert(!F0 || H0L, "the folded layer 0 needs the LDS feature panel");
  constexpr int W = 64 * WN * CH, RB = 8 / WN, WR = 32 * RT, BM = WR * RB;   // WR = rows per wave
  constexpr int kSlabs = WN * CH;           // 64-column slabs of the layer
  constexpr int kPitchE = W + 8;            // panel row pitch, elements (16 bytes of padding)
  constexpr int kPitchB = kPitchE * 2;
  constexpr int kPitch8 = W + 16;           // C8: row pitch of an fp8 panel image, bytes
  constexpr int KS1 = W / 16;
  constexpr int kHalves = (RT + 1) / 2;     // row dots go through a 64-row scratch image per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef BNF_MARK0_AT
#define BNF_MARK0_AT 0     // ABLATE builds: where phase clock 0 is read (0: feature staging issued, 1: kernel entry, 2: behind the staging barrier)
#endif
  if (BNF_MARK0_AT == 1) BNF_MARK(a, 0);
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
  constexpr int kImageB = BM * kPitchB > 8 * 64 * kRowDotPitch * 4 ? BM * kPitchB : 8 * 64 * kRowDotPitch * 4;
  float* xs = reinterpret_cast<float*>(smem + kImageB);
  float* s_part = xs;                       // [BM][kSlabs] row-dot partials
  float* s_dv = s_part + BM * kSlabs;       // [BM]
  float* s_col = s_dv + BM;                 // [2][RB][W] column sums
  float* s_sc = s_col + 2 * RB * W;         // scalars
  constexpr int kH0Pitch = FP * 2 + 16;     // a feature row + 16 bytes of padding
  constexpr int KS0c = FP / 16;             // k steps of the layer-0 contraction (H0L)
  const char* h0s = reinterpret_cast<const char*>(s_sc + 128);   // [BM][kH0Pitch] feature panel (H0L)
  // per (row tile, feature column) sums of the fused featurisation backward, two arrays of [BM / 32][FP]
  float* s_fb = (FP > 64 || RB > 1 || W < 512) ? reinterpret_cast<float*>(const_cast<char*>(h0s) + BM * kH0Pitch) : s_col + W;

  const int tid_k = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid_k >> 6);
  const int rb = wave / WN, cs = wave % WN;
  const int rbase = rb * WR;
  // FAIR (round 6 experiment, -DBNF_PANEL_FAIR=1; profiles/r06_panel_ab.md): in a VALU-only epilogue the SIMD arbitrates its
  // two waves by age -- the first-dispatched wave takes every slot it can use and reaches the barrier first (phase clocks:
  // layer-1 forward 10.9k cycles for wave 0, 16.9k for wave 7 from the same barrier).  Would taking turns at priority, tile
  // by tile (chunk c at priority (c + half) % 2), end both earlier?  Measured: no -- wave 0 slows to 13.2k, wave 7 stays at
  // 16.9k, the step is +0.4 %: the phase is bound by the SUM of the two waves' issue, however it is shared.
  auto fair_prio = [&](int chunk) {
#if BNF_PANEL_FAIR
    if (((chunk & 1) != 0) != (wave >= 4)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
    (void)chunk;
#endif
  };
  auto fair_prio_end = [&]() {
#if BNF_PANEL_FAIR
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto slab = [&](int hc) { return cs * CH + hc; };          // this wave's hc-th 64-column slab
  const int KS0 = a.Fp / 16;
#if BNF_PANEL_PRIO
  // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per
  // SIMD, item 4): the younger wave of every SIMD loses the issue arbitration in every phase
  // (measured: 1313-1323 us against 1309-1315 us without, same box -- not kept, profiles/r03_panel_ab.md)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // (A persistent variant -- `ppw` consecutive panels per workgroup, the next panel's features requested a
  // trip ahead -- was built and measured: 1390-1413 us for every ppw in {1, 2, 5, 10, 20} against 1355-1363 us
  // for this one-panel-per-workgroup form on the same box: the loop costs registers (hipcc hoists the
  // lane- and member-invariant values of all phases out of it: 145 SGPR spills, reloads inside the dZ1
  // epilogue) and the per-panel start-up it was meant to hide is not the feature load.  profiles/r03_panel_ab.md)
  const int tid = tid_k;
  const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(item / (uint32_t)a.panels), pn = (int)(item % (uint32_t)a.panels);
  const int m0 = pn * BM;                   // first batch row of the panel

  const float* th = a.theta + (int64_t)e * a.theta_stride;
  const float* sc = a.scal + (int64_t)e * kScalStride;
  const int LL = DEEP ? a.n_layers - 1 : 1; // index of the last hidden layer (1 for the two-layer networks)
  const float gamma0 = sc[0], gamma1 = sc[LL], alpha = sc[BNF_MAX_LAYERS];   // gamma1: the LAST hidden layer's scale
  const ActConst ak = act_const(alpha);
  const float sp_in_u = (H0L && a.fbmeta) ? sc[kScalGroup + a.fb_in_group] : 1.f;   // softplus(scale of the raw-input group)
  const float inv_sw = 1.0f / sqrtf((float)a.Wt), inv_sf = 1.0f / sqrtf((float)a.F);
  float* gr = a.grad + (int64_t)e * a.grad_stride;
  // per-member scalars of the row phase and of the serial tails, fetched and transformed now (uniform):
  // a lone thread reaching for them later pays a full memory latency with the workgroup waiting
  const float gam_o = sc[BNF_MAX_LAYERS + 1], bias_o = th[a.off_bias_out];
  const float dgam_o = sigmoidf(th[a.off_os]);
  const float dgam1 = sigmoidf(th[a.off_ls[LL]]) / gamma1, dgam0 = sigmoidf(th[a.off_ls[0]]) / gamma0;
  const float lns = th[a.off_lns];
  const float e_lns = expf(lns), sigma = 0.01f + e_lns, inv_sigma = 1.0f / sigma;
  const float ll_const = -logf(sigma) - 0.918938533204672742f;
  // q8: s_dZ = 2^(round(log2(c gamma_o / sigma)) - 6): d out is ~ c (y - out) / sigma^2 = (c / sigma) x a residual of order
  // one, and dZ_l is that times gamma_l k / sqrt W act' ~ a few 1e-2 -- stored values land around 2^0 .. 2^4 of e5m2's
  // 2^-14 .. 2^15 normal range, with ten binades of head room either way (count models: no sigma, c gamma_o alone)
  static_assert(!C8 || F0, "fp8 contractions: the folded forms (every BASELINE layout)");
  constexpr bool c8 = C8;     // the W x W contractions on the fp8 MFMA (PanelArgs.c8): its own instantiation -- as a run-time branch
                              // both contraction bodies were live in one kernel (256 registers + 112 bytes of scratch for bf16, too)
  float q_dz = 1.f;
  if (a.q8) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);   // MODE.FP16_OVFL: fp8 conversions clamp to the largest finite value
    const float ref = a.lik_c * gam_o * (a.obs == BNF_OBS_NORMAL ? inv_sigma : 1.f);
    q_dz = exp2f(rintf(log2f(fmaxf(ref, 1e-30f))) - 6.f);
    if (pn == 0 && tid_k == 0) a.qscale[e] = q_dz;
  }
  const float inv_qdz = 1.0f / q_dz;     // (a power of two: exact)
  // the row phase's target value, fetched now (one thread per row; rows >= B read nothing)
  const float y_pre = (!(BNF_PANEL_FIN && H0L && a.fin) && tid < BM && m0 + tid < a.B) ? a.ybat[(int64_t)e * a.row_batch + m0 + tid] : 0.f;

  float* s_grp = s_sc + 64;                 // [BNF_MAX_GROUPS + BNF_MAX_INPUTS] sums of the fused featurisation backward
  float* s_gfac = s_sc + 88;                // [BNF_MAX_GROUPS] sigmoid(scale_g) / softplus(scale_g) ...
  int32_t* s_goff = reinterpret_cast<int32_t*>(s_sc + 100);   // [BNF_MAX_GROUPS] ... and the theta offset of scale_g
  // what the serial tail of the fused featurisation backward needs from global memory is fetched NOW (two
  // dependent loads per group) and parked in LDS after the first phase: at the end of the panel the other
  // seven waves have nothing left to do while wave 0 finishes, so nothing there may wait for memory
  float g_fac = 0.f; int32_t g_off = 0;
  if constexpr (H0L) {
    if (tid < BNF_MAX_GROUPS + BNF_MAX_INPUTS) s_grp[tid] = 0.f;   // (ordered by the barriers of the phases below)
    if (a.fbmeta && tid < a.n_groups) {   // two INDEPENDENT loads (the factor from the member's scalar table: a dependent
      g_off = a.fbmeta[4 * FP + tid];     // th[g_off] here cost wave 0 a memory latency in front of its feature staging)
      g_fac = sc[kScalGfac + tid];
    }
  }
  // fragment-major features: the fragment of (32-row block, k step) is 1 KiB, blocks are Fp/16 KiB apart
  const char* h0p = reinterpret_cast<const char*>(a.H0 + (int64_t)e * a.h0_batch + (int64_t)m0 * a.Fp) +
                    (size_t)(rbase / 32) * KS0 * 1024;                     // uniform; + i blocks, + lane * 16
  const size_t h0blk = (size_t)KS0 * 1024;
  const char* wf0 = reinterpret_cast<const char*>(a.Wf[0] + (int64_t)e * a.w0_batch);
  const char* wb0 = reinterpret_cast<const char*>(a.Wb[0] + (int64_t)e * a.w0_batch);
  auto wfl = [&](int l) { return reinterpret_cast<const char*>(a.Wf[l] + (int64_t)e * a.w1_batch); };   // l >= 1
  auto wbl = [&](int l) { return reinterpret_cast<const char*>(a.Wb[l] + (int64_t)e * a.w1_batch); };
  // lane-dependent values, re-derived at the start of every phase (see opaque_lane)
  struct LaneCtx {
    int lane, frow, kg;
    const char *h0row, *w00, *w01, *prow;
    uint32_t w0off;        // byte offset of slab hc's first layer-0 weight fragment in wf0 (uniform)
  };
  auto lane_ctx = [&](int hc = 0) {
    LaneCtx c;
    c.lane = opaque_lane(tid) & 63;
    c.frow = c.lane & 31;
    c.kg = c.lane >> 5;
    c.h0row = h0p + c.lane * 16;
    c.w00 = wf0 + (size_t)(2 * slab(hc)) * KS0 * 1024 + c.lane * 16;      // layer-0 fragment streams of slab hc
    c.w01 = c.w00 + (size_t)KS0 * 1024;
    c.w0off = (uint32_t)(2 * slab(hc)) * (uint32_t)KS0 * 1024u;
    c.prow = smem + (rbase + c.frow) * kPitchB + c.kg * 16;   // A fragments of this lane
    return c;
  };

  // Copy of this wave's own 32-row x 64-column block (row block i) of the panel to the row-major
  // (Bp, W) array `dst`, issued right after the wave has written the block: only its own LDS
  // writes must have landed (lgkmcnt), no workgroup barrier, and the stores (128-byte runs, 1 KiB
  // per wave instruction) issue underneath the VALU work of the next block instead of in a
  // 10k-cycle burst per panel (a CU issues stores at ~13 B/clk).
  // q8: the block leaves as fp8 -- 32 rows x 64 BYTES = two 1 KiB wave stores; a lane converts two 16-element pieces
  // (2 x 32 bytes of the bf16 panel -> 2 x 16 bytes); `is_dz`: e5m2 / s_dZ, else e4m3 un-scaled
  auto block_to_global_q8 = [&](int lane, bf16_t* dst, int prb, int pcb, int i, bool is_dz) {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst) + (int64_t)e * a.act_batch + (int64_t)(m0 + prb + i * 32) * W + pcb;   // uniform
    const bf16_t* sp = tile + (prb + i * 32) * kPitchE + pcb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d, 0, 0x7fffffff, 0x00020000);
    u32x4 v[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = lane + 64 * u, row = pc >> 2, p4 = pc & 3;
      v[u][0] = *reinterpret_cast<const u32x4*>(sp + row * kPitchE + p4 * 16);
      v[u][1] = *reinterpret_cast<const u32x4*>(sp + row * kPitchE + p4 * 16 + 8);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = lane + 64 * u, row = pc >> 2, p4 = pc & 3;
      u32x4 o;
      if (is_dz)
        o = u32x4{bf16x4_to_bf8(v[u][0][0], v[u][0][1], q_dz), bf16x4_to_bf8(v[u][0][2], v[u][0][3], q_dz),
                  bf16x4_to_bf8(v[u][1][0], v[u][1][1], q_dz), bf16x4_to_bf8(v[u][1][2], v[u][1][3], q_dz)};
      else
        o = u32x4{bf16x4_to_fp8(v[u][0][0], v[u][0][1], 1.f), bf16x4_to_fp8(v[u][0][2], v[u][0][3], 1.f),
                  bf16x4_to_fp8(v[u][1][0], v[u][1][1], 1.f), bf16x4_to_fp8(v[u][1][2], v[u][1][3], 1.f)};
      __builtin_amdgcn_raw_buffer_store_b128(o, rs, (uint32_t)(row * W + p4 * 16), 0, BNF_PANEL_NT ? 2 : 0);
    }
  };
  // C8: a 32-row x 64-column block of an FP8 panel image (H_1 as e4m3, dZ_L as e5m2 / s_dZ) is already what the
  // weight-gradient kernels read: 32 rows x 64 bytes leave as two 1 KiB stores, no conversion
  auto block_to_global_p8 = [&](int lane, bf16_t* dst, int prb, int pcb, int i) {
    uint8_t* d = reinterpret_cast<uint8_t*>(dst) + (int64_t)e * a.act_batch + (int64_t)(m0 + prb + i * 32) * W + pcb;   // uniform
    const char* sp = smem + (prb + i * 32) * kPitch8 + pcb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d, 0, 0x7fffffff, 0x00020000);
    u32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = lane + 64 * u, row = pc >> 2, p4 = pc & 3;
      v[u] = *reinterpret_cast<const u32x4*>(sp + row * kPitch8 + p4 * 16);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = lane + 64 * u, row = pc >> 2, p4 = pc & 3;
      __builtin_amdgcn_raw_buffer_store_b128(v[u], rs, (uint32_t)(row * W + p4 * 16), 0, BNF_PANEL_NT ? 2 : 0);
    }
  };
  // C8, lane <-> column layouts (store_pair_pk's counterpart): the elements (row r, column c) and (row r + 1, column c) of
  // an fp8 panel image; BF8: e5m2 (backward signals, already divided by s_dZ), else e4m3
  auto store_pair_p8 = [&](int r, int c, float v0, float v1, auto bf8_tag) {
    const int pk = decltype(bf8_tag)::value ? __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, 0, false)
                                            : __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false);
    char* p8 = smem + r * kPitch8 + c;
    *reinterpret_cast<uint8_t*>(p8) = (uint8_t)pk;
    *reinterpret_cast<uint8_t*>(p8 + kPitch8) = (uint8_t)(pk >> 8);
  };
  auto is_dz_array = [&](const bf16_t* dst) {     // (uniform) one of the dZ arrays? else an activation copy
    bool dz = false;
#pragma unroll
    for (int l = 0; l < BNF_MAX_LAYERS; ++l) dz = dz || dst == a.dZ[l];
    return dz;
  };
  auto block_to_global = [&](const LaneCtx& L, bf16_t* dst, int i, int cbase) {
    if (BNF_ABL(a, 8)) return;
    if constexpr (C8) {
      if (dst != a.dZ[0]) {     // the panels the fp8 contractions read -- every H_l copy, every dZ_l but layer 0's -- are fp8 images
        block_to_global_p8(L.lane, dst, rbase, cbase, i);
        return;
      }
    }
    if (a.q8) {
      block_to_global_q8(L.lane, dst, rbase, cbase, i, is_dz_array(dst));
      return;
    }
    bf16_t* d = dst + (int64_t)e * a.act_batch + (int64_t)(m0 + rbase + i * 32) * W + cbase;   // uniform
    const bf16_t* sp = tile + (rbase + i * 32) * kPitchE + cbase;
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = L.lane + 64 * u;
      v[u] = *reinterpret_cast<const u32x4*>(sp + (idx >> 3) * kPitchE + (idx & 7) * 8);
    }
#if BNF_PANEL_SADDR
    // raw buffer stores: the block's (uniform) address is the resource base, a lane's place in the block its 32-bit
    // offset -- no 64-bit VALU address per store (aux 2 = non-temporal)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = L.lane + 64 * u;
      __builtin_amdgcn_raw_buffer_store_b128(v[u], rs, (uint32_t)((idx >> 3) * W * 2 + (idx & 7) * 16), 0, BNF_PANEL_NT ? 2 : 0);
    }
#else
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = L.lane + 64 * u;
#if BNF_PANEL_NT
      __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(d + (int64_t)(idx >> 3) * W + (idx & 7) * 8));
#else
      *reinterpret_cast<u32x4*>(d + (int64_t)(idx >> 3) * W + (idx & 7) * 8) = v[u];
#endif
    }
#endif
  };

  f32x16 accs[CH][RT][2];   // [slab][row tile][column tile]; the phases below see one slab at a time as `acc`
  auto zero_acc = [&]() {
#pragma unroll
    for (int hc = 0; hc < CH; ++hc)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[hc][i][j][r] = 0.f;
  };
  // Layer-0 pre-activations of the 32 x 32 tile (row block i, column half j) of this wave,
  // a0 = H0 K0 (un-scaled).  `blk` enters holding the first four k steps of tile t = 2 i + j and
  // leaves holding those of tile t + 1 (in flight during the caller's epilogue).
  L0Blk blk;
  auto l0_tile_src = [&](const LaneCtx& L, int t, const char** hr, const char** w) {
    const int tt = min(t, 2 * RT - 1);
    *hr = L.h0row + (size_t)(tt >> 1) * h0blk;
    *w = (tt & 1) ? L.w01 : L.w00;
  };
  bf16x8 bres[H0L ? 2 : 1][H0L ? KS0c : 4];     // H0L: this wave's layer-0 weight fragments [column half][k step]
  auto l0_weights = [&](const LaneCtx& L) {
    if constexpr (H0L) {
      const PanelW w0w = panel_wbase(wf0);
      const uint32_t o0 = L.w0off;
#pragma unroll
      for (int u = 0; u < KS0c; ++u) {
        bres[0][u] = panel_wload(w0w, o0 + (uint32_t)u * 1024u, (uint32_t)L.lane * 16u);
        bres[1][u] = panel_wload(w0w, o0 + (uint32_t)(KS0 + u) * 1024u, (uint32_t)L.lane * 16u);
      }
    } else {
      l0_load(blk, L.h0row, L.w00, 0);
    }
  };
  auto l0_tile = [&](const LaneCtx& L, f32x16& a0, int i, int j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = 0.f;
    if constexpr (H0L) {
      if (BNF_ABL(a, 16)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = 0.25f;
        return;
      }
      const char* ap = h0s + (rbase + i * 32 + L.frow) * kH0Pitch + L.kg * 16;
#pragma unroll
      for (int u0 = 0; u0 < KS0c; u0 += 4) {
        bf16x8 fa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(ap + (u0 + u) * 32);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bf16x8 fw = bres[j ? (H0L ? 1 : 0) : 0][H0L ? u0 + u : 0];
          a0 = F0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fa[u], a0, 0, 0, 0)      // transposed tile (see F0)
                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u], fw, a0, 0, 0, 0);
        }
      }
      return;
    }
    const int t = 2 * i + j;
    const char *hr, *w, *hn, *wn;
    l0_tile_src(L, t, &hr, &w);
    l0_tile_src(L, t + 1, &hn, &wn);
#pragma unroll 1
    for (int ks = 0; ks < KS0; ks += 4) {
      l0_mma(a0, blk);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 4 < KS0) l0_load(blk, hr, w, ks + 4);
      else l0_load(blk, hn, wn, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // =============================== layer 0 forward -> H1 panel ===============================
  // (BNF_PANEL_PRE0) slab 0's layer-0 weight fragments and biases do not depend on the staged features: requested before
  // the barrier, their L2 latency runs under the feature panel's HBM latency instead of behind it
  constexpr bool kPre0 = BNF_PANEL_PRE0 != 0 && H0L;
  const LaneCtx Lpre = lane_ctx(0);
  float gb_pre[2] = {0.f, 0.f};
  if constexpr (kPre0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) gb_pre[j] = th[a.off_bias[0] + slab(0) * 64 + j * 32 + Lpre.frow];
    l0_weights(Lpre);
  }
  if constexpr (H0L) {
   if (BNF_PANEL_FIN && a.fin) {
    // ---- featurise this panel's rows into LDS (what k_featurize<bf16_t> computes, value for value) -----------------
    // A lane owns a ROW, a wave a block of CPT consecutive columns of 64 rows: which column holds what is then uniform
    // across the wave -- metadata and group scales by scalar loads, no divergence (a first version with four lanes per
    // row and 16 columns each diverged four ways per column and cost 105 us per launch against the 37 us of the kernel
    // it replaced: profiles/r04_panel_ab.md r04l).  Inputs u_d = x_d / (input scale e^lsa_d) (the member's scalar table),
    // Fourier columns cos / sin(2 pi 2^k u_d) / (k + 1) by the hardware v_cos / v_sin on the fractional part, seasonal
    // columns from the data-constant table, interactions u_p u_q, every group times softplus(scale_g); the ones columns
    // of the folded layer 0; zero padding.  The row's target goes to s_dv (read by the row phase).
    constexpr int RG = BM / 64, PARTS = 8 / RG, CPT = FP / PARTS;    // row groups of 64, waves per row group, columns per wave
    static_assert(BM % 64 == 0 && CPT % 8 == 0 && CPT * PARTS == FP, "feature panel geometry");
    const int part = wave % PARTS;                                   // uniform
    const int r = (wave / PARTS) * 64 + (tid & 63);
    const int m = m0 + r;
    float vals[CPT];
    const bool live = m < a.B;
    const int64_t row = live ? row_of(a.rs, e, m) : 0;            // (dead rows compute on row 0 and store zeros)
    const float y_here = (a.y && part == 0 && live) ? a.y[row] : 0.f;
    FeatIn fi;
    fi.X = a.X; fi.stab = a.stab; fi.sc = sc; fi.fcol = a.fcol; fi.n_in = a.n_in; fi.n_seas = a.n_seas;
    featurize_cols<CPT>(fi, row, live, part * CPT, vals);
    if (part == 0) s_dv[r] = y_here;
#pragma unroll
    for (int ch = 0; ch < CPT / 8; ++ch) {
      const u32x4 w = {pack_bf16x2(vals[8 * ch], vals[8 * ch + 1]), pack_bf16x2(vals[8 * ch + 2], vals[8 * ch + 3]),
                       pack_bf16x2(vals[8 * ch + 4], vals[8 * ch + 5]), pack_bf16x2(vals[8 * ch + 6], vals[8 * ch + 7])};
      *reinterpret_cast<u32x4*>(const_cast<char*>(h0s) + r * kH0Pitch + (part * CPT + 8 * ch) * 2) = w;
    }
   } else {
    // feature panel -> LDS (row-major source written by k_featurize, 16-byte chunks, 8 per row)
    const bf16_t* src = a.H0rm + (int64_t)e * a.h0_batch + (int64_t)m0 * FP;
    constexpr int kCpr = FP / 8;              // 16-byte chunks per feature row
#pragma unroll
    for (int c = 0; c < (BNF_ABL(a, 32) ? 0 : (BM * kCpr) / 512); ++c) {
      const int q = tid + c * 512;
      *reinterpret_cast<u32x4*>(const_cast<char*>(h0s) + (q / kCpr) * kH0Pitch + (q % kCpr) * 16) =
          *reinterpret_cast<const u32x4*>(src + (int64_t)(q / kCpr) * FP + (q % kCpr) * 8);
    }
   }
  }
  if (BNF_MARK0_AT == 0) BNF_MARK(a, 0);
  if constexpr (H0L) lds_barrier();     // the staged feature panel is complete
  if (BNF_MARK0_AT == 2) BNF_MARK(a, 0);
  if constexpr (H0L) {
    if (BNF_PANEL_FIN && a.fin) {   // the row-major (Bp, Fp) copy the layer-0 weight gradient reads: whole 16-byte chunks, rows >= B are zero
      constexpr int kCpr = FP / 8;
      bf16_t* dst = a.H0out + (int64_t)e * a.h0_batch + (int64_t)m0 * FP;
#pragma unroll
      for (int c = 0; c < (BM * kCpr) / 512; ++c) {
        const int q = tid + c * 512;
        *reinterpret_cast<u32x4*>(dst + (int64_t)(q / kCpr) * FP + (q % kCpr) * 8) =
            *reinterpret_cast<const u32x4*>(h0s + (q / kCpr) * kH0Pitch + (q % kCpr) * 16);
      }
    }
  }
#pragma unroll
  for (int hc = 0; hc < CH; ++hc) {
    const int cbase = slab(hc) * 64;
    const LaneCtx L = (kPre0 && hc == 0) ? Lpre : lane_ctx(hc);
    const int frow = L.frow, kg = L.kg;
    const float gs = gamma0 * inv_sf * kLog2e;     // t = A0 log2(e): the activation core works on it (act_core2)
    float gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      gb[j] = BNF_ABL(a, 64) ? 0.1f : gamma0 * kLog2e * ((kPre0 && hc == 0) ? gb_pre[j] : th[a.off_bias[0] + cbase + j * 32 + frow]);
    if (!BNF_ABL(a, 64) && !(kPre0 && hc == 0)) l0_weights(L);
    // epilogue of one 32 x 32 tile: t = A0 log2(e) -> H1 = act(A0) -> LDS panel
    auto l0_epilogue = [&](const f32x16& a0, int i, int j) {
      if constexpr (F0) {
        // transposed tile, scale and bias already inside: lane <-> row, registers 4 rg .. 4 rg + 3 <-> four consecutive
        // hidden units: one 8-byte panel store per register group
        bf16_t* rowp = tile + (rbase + i * 32 + frow) * kPitchE + cbase + j * 32 + 4 * kg;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          f32x2 hq[2];
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            const f32x2 tv = {a0[rg * 4 + q], a0[rg * 4 + q + 1]};
#if BNF_PANEL_FWDS
            const ActFwd2 c = act_fwd_core2(tv);
            hq[q >> 1] = ak.c1 * c.r + (ak.alpha * c.s + ak.c0);
#else
            const ActCore2 c = act_core2(tv);
            const f32x2 s = kLn2 * c.mxt + c.dl;
            hq[q >> 1] = ak.c1 * c.r + (ak.alpha * s + ak.c0);
#endif
          }
          if constexpr (C8) {      // H_1 leaves as e4m3 into the fp8 panel image: four consecutive hidden units = one dword
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(hq[0].x, hq[0].y, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(hq[1].x, hq[1].y, pk, true);
            *reinterpret_cast<int*>(smem + (rbase + i * 32 + frow) * kPitch8 + cbase + j * 32 + 4 * kg + 8 * rg) = pk;
          } else {
            store_quad_pk(rowp + 8 * rg, hq[0].x, hq[0].y, hq[1].x, hq[1].y);
          }
        }
        return;
      }
      const int lc = cbase + j * 32 + frow;
      const float gbj = j ? gb[1] : gb[0];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int lr = rbase + i * 32 + 8 * rg + 4 * kg;
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
          const f32x2 tv = f32x2{a0[rg * 4 + q], a0[rg * 4 + q + 1]} * gs + gbj;
          f32x2 h = tv;
          if (!BNF_ABL(a, 2)) {
#if BNF_PANEL_FWDS
            const ActFwd2 c = act_fwd_core2(tv);
            h = ak.c1 * c.r + (ak.alpha * c.s + ak.c0);
#else
            const ActCore2 c = act_core2(tv);
            const f32x2 s = kLn2 * c.mxt + c.dl;
            h = ak.c1 * c.r + (ak.alpha * s + ak.c0);
#endif
          }
          if (!BNF_ABL(a, 4)) store_pair_pk(tile + (lr + q) * kPitchE + lc, tile + (lr + q + 1) * kPitchE + lc, h.x, h.y);
        }
      }
    };
    if constexpr (H0L) {
      // software pipeline over the 2 RT tiles: the MFMA chain of tile t + 1 is issued before the epilogue of
      // tile t, and the copy of row block i - 1 to HBM leaves in the middle of row block i -- its LDS writes
      // landed long ago, so the copy's lgkmcnt wait costs nothing (right behind its own block's last store it
      // stalled both waves of a SIMD at the same point)
      f32x16 a0b[2];
      l0_tile(L, a0b[0], 0, 0);
#pragma unroll 1
      for (int i = 0; i < RT; ++i) {
        l0_tile(L, a0b[1], i, 1);
        fair_prio(0);
        l0_epilogue(a0b[0], i, 0);
        if (i > 0) block_to_global(L, a.Hout[0], i - 1, cbase);
        if (i + 1 < RT) l0_tile(L, a0b[0], i + 1, 0);
        fair_prio(1);
        l0_epilogue(a0b[1], i, 1);
      }
      fair_prio_end();
      block_to_global(L, a.Hout[0], RT - 1, cbase);
    } else {
#pragma unroll 1
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 a0;
          if (!BNF_ABL(a, 1)) l0_tile(L, a0, i, j);
          else {
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = 0.25f;
          }
          l0_epilogue(a0, i, j);
        }
        block_to_global(L, a.Hout[0], i, cbase);
      }
    }
  }
  BNF_MARK(a, 1);
  if constexpr (H0L) {
    if (a.fbmeta && tid < a.n_groups) { s_gfac[tid] = g_fac; s_goff[tid] = g_off; }
  }
  PanelRing ring[CH];
  auto ring_prefetch = [&](const char* wp, int lane) {   // the first weight fragments of every slab of this wave
#pragma unroll
    for (int hc = 0; hc < CH; ++hc) panel_prefetch(ring[hc], wp, 2 * slab(hc), KS1, lane);
  };
  // c8: the W x W contraction on the fp8 MFMA (panel_contract8) out of the fp8 panel image; bf8: backward signals (e5m2 / s_dZ)
  auto contract_all8 = [&](const uint8_t* wp8, auto swap_tag, auto bf8_tag) {
    constexpr bool kBf8 = decltype(bf8_tag)::value;
    const int lane = opaque_lane(tid) & 63;
    const char* prow8 = smem + (rbase + (lane & 31)) * kPitch8 + (lane >> 5) * 32;
    // s_dZ = 2^k exactly: its E8M0 byte is the float's own exponent field
    const int pscale = kBf8 ? (int)((__float_as_uint(q_dz) >> 23) & 0xffu) : 127;
#if BNF_PANEL_CPRIO
    if ((BNF_PANEL_CPRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int hc = 0; hc < CH; ++hc)
      panel_contract8<kPitch8, RT, decltype(swap_tag)::value, kBf8>(accs[hc], prow8, wp8, 2 * slab(hc), W / 64, lane, pscale);
#if BNF_PANEL_CPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto contract_all_t = [&](const char* wp, auto swap_tag) {   // accs[hc] (+)= panel . W[:, slab hc] for every slab
    const LaneCtx L = lane_ctx();
#if BNF_PANEL_CPRIO
    if ((BNF_PANEL_CPRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int hc = 0; hc < CH; ++hc)
      panel_contract<kPitchB, RT, BNF_PANEL_ZPEEL != 0, decltype(swap_tag)::value>(accs[hc], L.prow, wp, 2 * slab(hc), KS1, L.lane,
                                                                                  ring[hc], [](int) {});
#if BNF_PANEL_CPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // L1T: the copy of the dZ_L panel to HBM (for the weight gradient).  The backward epilogue of the last layer has no
  // arithmetic left to hide 128 KiB of stores behind (HBM takes ~13 B/clk/CU with every CU storing: the epilogue measured
  // 10.4k cycles with the copy inside it, 7.6k without), and stores issued DURING the next contraction sit in front of its
  // weight-ring loads in the wave's in-order vmcnt (+2.7k cycles on that contraction: profiles/r05_panel_ab.md).  But the
  // contraction that reads this panel is finished early by the four waves that run it at priority (BNF_PANEL_CPRIO: the
  // second-dispatched half; 11k against 17.5k cycles), which then sit at the barrier: THEY copy the whole panel -- their own
  // 64 columns and those of the wave they share a SIMD with -- while the other four are still multiplying.  Only LDS
  // reads (complete before the barrier: lds_barrier) and fire-and-forget stores.
  auto copy_panel_by_early_waves = [&](bf16_t* dst) {
    if (BNF_ABL(a, 8)) return;
    if ((BNF_PANEL_CPRIO == 2) == (wave >= 4)) return;       // (the waves that ran the contraction WITHOUT priority: busy)
    const int lane = opaque_lane(tid) & 63;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int pw = b == 0 ? wave : (wave ^ 4);              // this wave's slab, then its SIMD partner's
      const int prb = (pw / WN) * WR, pcb = (pw % WN) * CH * 64;
      if constexpr (C8) {          // (only ever called for dZ_L: the e5m2 panel image as it stands)
#pragma unroll
        for (int i = 0; i < RT; ++i) block_to_global_p8(lane, dst, prb, pcb, i);
        continue;
      }
      if (a.q8) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int hc = 0; hc < CH; ++hc) block_to_global_q8(lane, dst, prb, pcb + hc * 64, i, true);
        continue;
      }
      bf16_t* d = dst + (int64_t)e * a.act_batch + (int64_t)(m0 + prb) * W + pcb;   // uniform
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(d, 0, 0x7fffffff, 0x00020000);
      const bf16_t* sp = tile + prb * kPitchE + pcb;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        u32x4 v[4 * CH];
#pragma unroll
        for (int u = 0; u < 4 * CH; ++u) {
          const int idx = lane + 64 * u, row = i * 32 + (idx >> (3 + (CH - 1))), pc = idx & (8 * CH - 1);   // 16-byte pieces: 8 CH per row
          v[u] = *reinterpret_cast<const u32x4*>(sp + row * kPitchE + pc * 8);
        }
#pragma unroll
        for (int u = 0; u < 4 * CH; ++u) {
          const int idx = lane + 64 * u, row = i * 32 + (idx >> (3 + (CH - 1))), pc = idx & (8 * CH - 1);
          __builtin_amdgcn_raw_buffer_store_b128(v[u], rs, (uint32_t)(row * W * 2 + pc * 16), 0, BNF_PANEL_NT ? 2 : 0);
        }
      }
    }
  };
  auto contract_all = [&](const char* wp) { contract_all_t(wp, std::false_type{}); };
  if (!c8) ring_prefetch(wfl(1), opaque_lane(tid) & 63);   // in flight across the barrier
  lds_barrier();
  BNF_MARK(a, 2);

  // this wave's slot of a middle layer's parked pre-activations: per slab 4 RT chunks x 64 lanes x 8 bf16
  auto park_ptr = [&](int l, int hc) {
    return a.park[l] + (((((int64_t)e * a.panels + pn) * 8 + wave) * CH + hc) * (4 * RT)) * (64 * 8);   // 4 RT chunks of 1 KiB per slab
  };
  // =============================== middle layers forward (depth > 2) ==========================
  // H_{l+1} = act(gamma_l (H_l K_l / sqrt W + b_l)): contraction out of the LDS panel exactly like the last
  // layer's; the epilogue overwrites the panel with H_{l+1} (after the barrier: every wave has read H_l), copies it to
  // HBM for the weight gradient, and PARKS t_l = A_l log2(e) for the backward pass as bf16 in this wave's own
  // accumulator order -- the lanes that will need a value are the ones that hold it now, so the layout is free and
  // both directions are whole 1 KiB wave accesses (the layer pipeline stores A_l^T, same rounding).
#pragma unroll 1
  for (int l = 1; DEEP && l < LL; ++l) {
    if (!BNF_PANEL_ZPEEL) zero_acc();
    if constexpr (c8) contract_all8(a.Wf8[l] + (int64_t)e * a.w8_batch, std::false_type{}, std::false_type{});
    else contract_all(wfl(l));
    lds_barrier();
#pragma unroll
    for (int hc = 0; hc < CH; ++hc) {
      const int cbase = slab(hc) * 64;
      f32x16 (&acc)[RT][2] = accs[hc];
      const LaneCtx L = lane_ctx();
      const int lane = L.lane, frow = L.frow, kg = L.kg;
      const float gl = sc[l];
      const float gs = gl * inv_sw * kLog2e;
      float gb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) gb[j] = gl * kLog2e * th[a.off_bias[l] + cbase + j * 32 + frow];
      bf16_t* pk = park_ptr(l, hc);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        int tile_off = (rbase + i * 32 + 4 * kg) * kPitchE;     // (one LDS base per 32-row tile: see the dZ1 epilogue)
        asm volatile("" : "+v"(tile_off));
        bf16_t* tile_i = tile + tile_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int lc = cbase + j * 32 + frow;
          uint32_t pw[8];
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int lr = rbase + i * 32 + 8 * rg + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 tv = f32x2{acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]} * gs + gb[j];
              pw[rg * 2 + (q >> 1)] = pack_bf16x2(tv.x, tv.y);
#if BNF_PANEL_FWDS
              const ActFwd2 c = act_fwd_core2(tv);
              const f32x2 s = c.s;
#else
              const ActCore2 c = act_core2(tv);
              const f32x2 s = kLn2 * c.mxt + c.dl;
#endif
              const f32x2 h = ak.c1 * c.r + (ak.alpha * s + ak.c0);
              if constexpr (C8) store_pair_p8(lr + q, lc, h.x, h.y, std::false_type{});
              else store_pair_pk(tile_i + (8 * rg + q) * kPitchE + lc, tile_i + (8 * rg + q + 1) * kPitchE + lc, h.x, h.y);
            }
          }
          u32x4* dst = reinterpret_cast<u32x4*>(pk + ((i * 2 + j) * 2 * 64 + lane) * 8);
          __builtin_nontemporal_store(u32x4{pw[0], pw[1], pw[2], pw[3]}, dst);
          __builtin_nontemporal_store(u32x4{pw[4], pw[5], pw[6], pw[7]}, dst + 64);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i > 0) block_to_global(L, a.Hout[l], i - 1, cbase);
      }
      block_to_global(L, a.Hout[l], RT - 1, cbase);
    }
    if (!c8) ring_prefetch(wfl(l + 1), opaque_lane(tid) & 63);
    lds_barrier();
  }

  // =============================== last hidden layer forward ==================================
  // L1T (round 5): the last hidden layer runs on TRANSPOSED tiles (MFMA operand roles swapped like the F0 layer-0 forms:
  // lane <-> row, registers 4 rg .. 4 rg + 3 <-> four consecutive hidden units) and its activation is evaluated ONCE:
  //   * the forward epilogue (here) leaves  dg = (gamma_L k_o / sqrt W) act'(A_L)  in the accumulators -- dZ_L is then one
  //     multiply with the row's dv --, writes H_{L+1} = act(A_L) as bf16 into the (dead) panel image with 8-byte stores,
  //     and forms the per-ROW dots in-lane (no LDS round trip: a lane owns its row): sum_c k' s, sum_c k' r -- the output
  //     dot AND the d alpha dot follow from these two -- and sum_c dg t (d gamma_L);
  //   * the backward epilogue evaluates NO activation: z = dv dg, convert, store;
  //   * the two per-COLUMN sums that are left go to the idle matrix pipe: d k_o = dv^T H_{L+1} and d b_L = 1^T dZ_L as
  //     bf16 MFMAs over the wave's own 64 columns of the panel (ds_read_b64_tr_b16 fragments; a wave reads only what it
  //     wrote itself, so no barrier is added).
  // Per element pair: 17 VALU + 4 transcendentals forward, 2 VALU backward, where the r04 kernel spent 9 + 4 and 18 + 4
  // (+ the 64-row LDS scratch of the row dots, + 2-byte panel stores): profiles/r05_panel_ab.md.
  // (not CH = 2, the W = 1024 form: both slabs' accumulators + the column constants spill; not WN = 4, the W = 256 forms: a
  // wave holds half as many elements, the column-sum MFMAs and their atomics -- two row blocks each -- weigh twice as much:
  // C5/8 10.89k -> 10.79k member-steps/s same box, three alternations, profiles/r05_panel_ab.md)
  constexpr bool kL1T = BNF_PANEL_L1T != 0 && CH == 1 && WN == 8;
  if (!BNF_PANEL_ZPEEL) zero_acc();
  if constexpr (c8)
    contract_all8(a.Wf8[LL] + (int64_t)e * a.w8_batch, std::integral_constant<bool, kL1T>{}, std::false_type{});
  else
    contract_all_t(wfl(LL), std::integral_constant<bool, kL1T>{});
  BNF_MARK(a, 3);
  // L1T: the per-register column constants of slab 0 (bias and output kernel of the wave's 64 columns: a lane holds the
  // 32 columns of its half) are requested before the barrier -- the ring and fragment registers are dead -- and arrive
  // under the wait
  f32x4 cst_b[kL1T ? 8 : 1], cst_k[kL1T ? 8 : 1];
  auto l1_consts = [&](int hc) {
    const int kg = (opaque_lane(tid) & 63) >> 5;
    const float* bp = th + a.off_bias[LL] + slab(hc) * 64 + 4 * kg;
    const float* kp = th + a.off_ko + slab(hc) * 64 + 4 * kg;
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // u = 4 j + rg: columns j * 32 + 8 rg + 4 kg + 0..3
      cst_b[u] = *reinterpret_cast<const f32x4u*>(bp + (u >> 2) * 32 + (u & 3) * 8);
      cst_k[u] = *reinterpret_cast<const f32x4u*>(kp + (u >> 2) * 32 + (u & 3) * 8);
    }
  };
  if constexpr (kL1T) l1_consts(0);
  lds_barrier();     // every wave is done reading H1: the panel doubles as row-dot scratch below (L1T: receives H_{L+1})

  // L1T: per-row partial dots of this wave's slab(s), kept in registers from the forward to the backward epilogue
  // (the row's dv is known only after the row phase): wrow = sum_c k' (s + 2 r) / gamma_L (d alpha), urow = sum_c dg t
  float wrow[kL1T ? CH : 1][kL1T ? RT : 1], urow[kL1T ? CH : 1][kL1T ? RT : 1];
  if constexpr (kL1T) {
    float ksum_wave = 0.f;   // sum of k_o over all of this wave's columns
#pragma unroll
    for (int hc = 0; hc < CH; ++hc) {
      const int cbase = slab(hc) * 64;
      f32x16 (&acc)[RT][2] = accs[hc];
      const LaneCtx L = lane_ctx();
      const int lane = L.lane, frow = L.frow, kg = L.kg;
      if (hc > 0) l1_consts(hc);
      const float gs = gamma1 * inv_sw * kLog2e;
      const float gl2 = gamma1 * kLog2e, gkw = gamma1 * inv_sw;
      // t = A log2(e) = acc gs + gbp;  k' = gamma_L k_o / sqrt W  (per register pair: [j][2 rg + q / 2])
      f32x2 gbp[2][8], kvp[2][8];
      f32x2 ks2 = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        gbp[u >> 2][(u & 3) * 2] = f32x2{cst_b[u][0], cst_b[u][1]} * gl2;
        gbp[u >> 2][(u & 3) * 2 + 1] = f32x2{cst_b[u][2], cst_b[u][3]} * gl2;
        kvp[u >> 2][(u & 3) * 2] = f32x2{cst_k[u][0], cst_k[u][1]} * gkw;
        kvp[u >> 2][(u & 3) * 2 + 1] = f32x2{cst_k[u][2], cst_k[u][3]} * gkw;
        ks2 += f32x2{cst_k[u][0], cst_k[u][1]} + f32x2{cst_k[u][2], cst_k[u][3]};
      }
      float ksum = ks2.x + ks2.y;                    // sum of k_o over this lane's 32 columns ...
      ksum += __shfl_xor(ksum, 32, 64);              // ... and over the slab's 64
      ksum_wave += ksum;
      if (hc == CH - 1 && lane == 0) s_sc[48 + wave] = ksum_wave;   // (read by thread 0 after the barriers below)
      const float inv_g1 = 1.0f / gamma1;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        f32x2 ds2 = {0.f, 0.f}, dr2 = {0.f, 0.f}, du2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          fair_prio(2 * i + j);
          bf16_t* rowp = tile + (rbase + i * 32 + frow) * kPitchE + cbase + j * 32 + 4 * kg;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            f32x2 hq[2];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
              const f32x2 kp = kvp[j][rg * 2 + (q >> 1)];
              const f32x2 tv = raw * gs + gbp[j][rg * 2 + (q >> 1)];
              const ActCore2 c = panel_act_core2(tv);
              const f32x2 sv = kLn2 * c.mxt + c.dl;                                   // elu + 1
              hq[q >> 1] = ak.c1 * c.r + (ak.alpha * sv + ak.c0);                   // act(A)
              ds2 += kp * sv;
              dr2 += kp * c.r;
              const f32x2 dg = kp * (ak.c2 * (c.r - c.r * c.r) + ak.alpha * c.dl);  // k' act'(A)
              du2 += dg * tv;
              acc[i][j][rg * 4 + q] = dg.x;
              acc[i][j][rg * 4 + q + 1] = dg.y;
            }
            store_quad_pk(rowp + 8 * rg, hq[0].x, hq[0].y, hq[1].x, hq[1].y);
            asm volatile("" : "+v"(ds2), "+v"(dr2), "+v"(du2));
            if (BNF_EPI_FENCE_EVERY == 1 || (rg & 1)) __builtin_amdgcn_sched_barrier(0);
          }
        }
        // this lane's half of the row (32 of the slab's 64 columns) + the other half's
        float dsv = ds2.x + ds2.y, drv = dr2.x + dr2.y, duv = du2.x + du2.y;
        dsv += __shfl_xor(dsv, 32, 64);
        drv += __shfl_xor(drv, 32, 64);
        duv += __shfl_xor(duv, 32, 64);
        // sum_c k_o act = (c0 sum k' + c1 sum k' r + alpha sum k' s) sqrt W / gamma_L: what the row phase adds up over the slabs
        if (kg == 0)
          s_part[(rbase + i * 32 + frow) * kSlabs + slab(hc)] = ak.c0 * ksum + (ak.c1 * drv + ak.alpha * dsv) * (inv_g1 / inv_sw);
        wrow[hc][i] = (dsv + 2.f * drv) * inv_g1;     // sum_c (k_o / sqrt W) (elu - tanh + 2)
        urow[hc][i] = duv;
      }
      fair_prio_end();
    }
  } else {
  // ---- A1 = gamma1 (acc / sqrt W + b1) kept in the accumulators; row dots act(A1) . k_o ----
  float ksum_wave = 0.f;   // sum of k_o over all of this wave's columns
#pragma unroll
  for (int hc = 0; hc < CH; ++hc) {
    const int cbase = slab(hc) * 64;
    f32x16 (&acc)[RT][2] = accs[hc];
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    // the accumulators keep t1 = A1 log2(e) from here on; row dot of act(A1) = c0 + c1 r + alpha s with k_o:
    // the constant term c0 sum_c k_o[c] is added once per row below
    const float gs = gamma1 * inv_sw * kLog2e;
    float gb[2], ka[2], kc1[2];
    float ksum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gb[j] = gamma1 * kLog2e * th[a.off_bias[LL] + cbase + j * 32 + frow];
      const float kov = th[a.off_ko + cbase + j * 32 + frow];
      ka[j] = kov * ak.alpha; kc1[j] = kov * ak.c1;
      ksum += kov;
    }
    ksum = panel_wave_sum(kg == 0 ? ksum : 0.f);           // sum of k_o over this slab's 64 columns
    ksum_wave += ksum;
    if (hc == CH - 1 && lane == 0) s_sc[48 + wave] = ksum_wave;   // (read by thread 0 after the barriers below)
    ksum *= ak.c0;
    float* s_dot = reinterpret_cast<float*>(smem) + wave * (64 * kRowDotPitch);   // [64 rows][32 lanes]
#pragma unroll
    for (int half = 0; half < kHalves; ++half) {   // (unrolled: a runtime index would push acc to scratch)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = half * 2 + ii;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          float pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
              const f32x2 tv = raw * gs + gb[j];
              acc[i][j][rg * 4 + q] = tv.x;
              acc[i][j][rg * 4 + q + 1] = tv.y;
#if BNF_PANEL_FWDS
              const ActFwd2 c = act_fwd_core2(tv);
              const f32x2 s = c.s;
#else
              const ActCore2 c = act_core2(tv);
              const f32x2 s = kLn2 * c.mxt + c.dl;
#endif
              f32x2 pq = {pd[q], pd[q + 1]};
              pq = ka[j] * s + pq;
              pq = kc1[j] * c.r + pq;
              pd[q] = pq.x;
              pd[q + 1] = pq.y;
            }
          }
          float* dst = s_dot + (ii * 32 + 8 * rg + 4 * kg) * kRowDotPitch + frow;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q * kRowDotPitch] = pd[q];
          if (BNF_EPI_FENCE_EVERY == 1 || (rg & 1)) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const f32x4* rp = reinterpret_cast<const f32x4*>(s_dot + lane * kRowDotPitch);
      f32x4 t4 = rp[0];
#pragma unroll
      for (int c = 1; c < 8; ++c) t4 += rp[c];
      s_part[(rbase + half * 64 + lane) * kSlabs + slab(hc)] = ((t4.x + t4.y) + (t4.z + t4.w)) + ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  }
  BNF_MARK(a, 4);
  lds_barrier();
  // the panel's scalar sums of the row phase (thread 0): loss and the gradients of the output-layer scalars
  float dv_all = 0.f;   // thread 0: sum of d loss / d v over the panel's rows
  auto row_scalars_out = [&]() {
    float u[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < (BM + 63) / 64; ++w2)
#pragma unroll
      for (int i = 0; i < 5; ++i) u[i] += s_sc[w2 * 5 + i];
    dv_all = u[2];
    const float step_loss = -a.lik_c * u[0];
    atomicAdd(&a.loss[(int64_t)(e / a.S) * a.loss_stride + (a.st ? a.st->col : 0)], a.loss_scale * step_loss);
    if (a.loss_raw) atomicAdd(&a.loss_raw[e], step_loss);
    atomicAdd(&gr[a.off_os], dgam_o * u[1]);
    atomicAdd(&gr[a.off_bias_out], u[2]);
    atomicAdd(&gr[a.obs == BNF_OBS_NORMAL ? a.off_lns : a.off_shape], u[3]);
    if (a.obs == BNF_OBS_ZINB) atomicAdd(&gr[a.off_infl], u[4]);
  };
  // ---- one thread per row: output, likelihood, d out (models.py:269-273,157-191) -----------
  {
    float ll = 0.f, s_doutv = 0.f, s_dvsum = 0.f, s_par = 0.f, s_infl = 0.f;
    if (tid < BM) {
      const int m = m0 + tid;
      const float y_row = (BNF_PANEL_FIN && H0L && a.fin) ? s_dv[tid] : y_pre;   // (fin: the featurisation left the row's target here)
      float vsum = 0.f;
#pragma unroll
      for (int c = 0; c < kSlabs; ++c) vsum += s_part[tid * kSlabs + c];
      float dvv = 0.f;
      if (m < a.B) {
        const float v = vsum * inv_sw + bias_o;
        const float outv = gam_o * v;
        a.out[(int64_t)e * a.out_batch + m] = outv;
        RowLoss rl;
        if (a.obs == BNF_OBS_NORMAL) {   // row_loss_eval's NORMAL branch on the precomputed member scalars
          const float z = (y_row - outv) * inv_sigma;
          rl.ll = -0.5f * z * z + ll_const;
          rl.dout = -a.lik_c * z * inv_sigma;
          rl.d_par = -a.lik_c * (z * z - 1.0f) * inv_sigma * e_lns;
          rl.d_infl = 0.f;
        } else {
          rl = row_loss_eval(a.obs, th, a.off_lns, a.off_shape, a.off_infl, y_row, outv, a.lik_c);
        }
        ll = rl.ll; s_par = rl.d_par; s_infl = rl.d_infl;
        s_doutv = rl.dout * v;
        dvv = gam_o * rl.dout;
        s_dvsum = dvv;
      }
      s_dv[tid] = dvv;
      if constexpr (kL1T) reinterpret_cast<bf16_t*>(s_col)[tid].bits = f32_to_bf16_bits(dvv);   // A fragments of d k_o = dv^T H (s_col is idle until the next layer's sums)
      const float t0 = panel_wave_sum(ll), t1 = panel_wave_sum(s_doutv), t2 = panel_wave_sum(s_dvsum), t3 = panel_wave_sum(s_par),
                  t4 = panel_wave_sum(s_infl);
      if ((tid & 63) == 0) {
        float* q = s_sc + wave * 5;
        q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; q[4] = t4;
      }
    }
  }
  lds_barrier();
  BNF_MARK(a, 5);
  if (tid == 0) row_scalars_out();
  if constexpr (kL1T) {
    // ---- L1T: dZ_L = dv dg (no activation left to evaluate); d k_o and d b_L on the matrix pipe --------------------
    float wsa_all = 0.f, wsg_all = 0.f;
#pragma unroll
    for (int hc = 0; hc < CH; ++hc) {
      const int cbase = slab(hc) * 64;
      f32x16 (&acc)[RT][2] = accs[hc];
      const LaneCtx L = lane_ctx();
      const int lane = L.lane, frow = L.frow, kg = L.kg;
      typedef __attribute__((address_space(3))) char lds_char_t;
      const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
      // Column sums over the wave's WR rows x 64 columns of the panel as v_mfma_f32_16x16x32_bf16 (K = 32 rows per
      // instruction, four 16-column tiles = four independent accumulator chains).  B fragments by ds_read_b64_tr_b16: within
      // a 16-lane group lane i receives element i % 4 of the 8-byte datum addressed by lane 4 j + i / 4 (j = 0 .. 3; measured
      // with scripts/probes/tr_read_probe.hip) -- so with lane p of the group addressing row base + p / 4, columns
      // 4 (p % 4) .. + 3, lane i ends up with rows base .. base + 3 of column i; the group's base is 8 (lane / 16) + 4 t:
      // lane (n = lane % 16, kg4 = lane / 16) holds rows 32 ks + 8 kg4 + 0 .. 7 of column n -- a plain B fragment.
      // All reads of a sum are issued at once (inline asm: hipcc must not put its own waits between them), one wait.
      const uint32_t tr0 = lds0 + (uint32_t)((rbase + (lane >> 4) * 8 + ((lane & 15) >> 2)) * kPitchB + (cbase + (lane & 3) * 4) * 2);
      const uint32_t dv0 = lds0 + (uint32_t)(reinterpret_cast<const char*>(s_col) - smem) + (uint32_t)((rbase + (lane >> 4) * 8) * 2);
      constexpr int KS32 = WR / 32;
      auto colsum = [&](float (&out)[4], auto with_dv) {
        constexpr bool kDv = decltype(with_dv)::value;
        u32x2_t rr[KS32][8];        // [k step][2 t + 4 ... ]: index 2 nt + t
        u32x4 av[KS32];
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          const uint32_t ad = tr0 + (uint32_t)(ks * 32 * kPitchB);
          rr[ks][0] = lds_tr16_b64<0>(ad);
          rr[ks][1] = lds_tr16_b64<4 * kPitchB>(ad);
          rr[ks][2] = lds_tr16_b64<32>(ad);
          rr[ks][3] = lds_tr16_b64<4 * kPitchB + 32>(ad);
          rr[ks][4] = lds_tr16_b64<64>(ad);
          rr[ks][5] = lds_tr16_b64<4 * kPitchB + 64>(ad);
          rr[ks][6] = lds_tr16_b64<96>(ad);
          rr[ks][7] = lds_tr16_b64<4 * kPitchB + 96>(ad);
          if constexpr (kDv) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(av[ks]) : "v"(dv0), "n"(ks * 64) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 d[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) d[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          asm volatile("" : "+v"(rr[ks][0]), "+v"(rr[ks][1]), "+v"(rr[ks][2]), "+v"(rr[ks][3]), "+v"(rr[ks][4]),
                            "+v"(rr[ks][5]), "+v"(rr[ks][6]), "+v"(rr[ks][7]));   // (ordered behind the wait)
          u32x4 aw = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};    // ones
          if constexpr (kDv) {
            asm volatile("" : "+v"(av[ks]));
            aw = av[ks];
          }
          const bf16x8 fa = __builtin_bit_cast(bf16x8, aw);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const u32x4 wv = {rr[ks][2 * nt].x, rr[ks][2 * nt].y, rr[ks][2 * nt + 1].x, rr[ks][2 * nt + 1].y};
            d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, __builtin_bit_cast(bf16x8, wv), d[nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) out[nt] = d[nt][0];   // every output row holds the sums: lanes 0 .. 15 <-> column 16 nt + lane
      };
      float dvr[RT];
      float wsa = 0.f, wsg = 0.f;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        dvr[i] = s_dv[rbase + i * 32 + frow];
        wsa += dvr[i] * wrow[hc][i];
        wsg += dvr[i] * urow[hc][i];
      }
      // d k_o sqrt W = dv^T H_{L+1} over this wave's rows: A = dv (bf16) replicated over the 16 output rows
      float dks[4];
      colsum(dks, std::true_type{});
      asm volatile("" : "+v"(dks[0]), "+v"(dks[1]), "+v"(dks[2]), "+v"(dks[3]));
      BNF_MARK(a, 6);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16_t* rowp = tile + (rbase + i * 32 + frow) * kPitchE + cbase + j * 32 + 4 * kg;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const f32x2 z0 = f32x2{acc[i][j][rg * 4], acc[i][j][rg * 4 + 1]} * dvr[i];
            const f32x2 z1 = f32x2{acc[i][j][rg * 4 + 2], acc[i][j][rg * 4 + 3]} * dvr[i];
            store_quad_pk(rowp + 8 * rg, z0.x, z0.y, z1.x, z1.y);
          }
        }
      }
      // (the copy of dZ_L to HBM: copy_panel_by_early_waves, behind the next contraction, which reads this panel)
      BNF_MARK(a, 7);
      if (hc == CH - 1 && !c8) ring_prefetch(wbl(LL), lane);   // the accumulators are dead: weights of dH = dZ K^T on their way
      // d b_L = 1^T dZ_L over the slab this wave has just written (its own LDS writes: in order behind them)
      float dbs[4];
      colsum(dbs, std::false_type{});
      if (lane < 16) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          atomicAdd(&gr[a.off_ko + cbase + nt * 16 + lane], dks[nt] * inv_sw);
          atomicAdd(&gr[a.off_bias[LL] + cbase + nt * 16 + lane], dbs[nt]);
        }
      }
      // both halves of the wave hold every row's dots: count them once
      wsa_all += panel_wave_sum(kg == 0 ? wsa : 0.f);
      wsg_all += panel_wave_sum(kg == 0 ? (kLn2 / gamma1) * wsg : 0.f);
      if (hc == CH - 1 && lane == 0) {
        s_sc[32 + wave * 2] = wsa_all;
        s_sc[33 + wave * 2] = wsg_all;
      }
    }
  } else {
  // ---- dZ1 = gamma1 (dv k_o / sqrt W) act'(A1) -> panel; column sums and scalar gradients ----
  float wsa_all = 0.f, wsg_all = 0.f;
#pragma unroll
  for (int hc = 0; hc < CH; ++hc) {
    const int cbase = slab(hc) * 64;
    f32x16 (&acc)[RT][2] = accs[hc];
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    f32x2 cr[2], sg[2], cp[2], cs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) cr[j] = sg[j] = cp[j] = cs[j] = f32x2{0.f, 0.f};
    // With dZ1 = z = dv (gamma1 k_o / sqrt W) act'(A1) formed directly (the column factor folded into the
    // constants of act'), the sums kept per column half are
    //   cr = sum dv r,  cs = sum dv s   (r = 1 / (1 + e^2), s = elu + 1): BOTH other column quantities follow from them --
    //        sum dv (elu - tanh + 2) = 2 cr + cs   (the "+ 2" leaves as  -2 (sum_c k_o / sqrt W)(sum_r dv)  by thread 0)
    //        d k_o sqrt W = sum act(A1) dv = c0 sum dv + c1 cr + alpha cs   (act = c0 + c1 r + alpha s)
    //      -- two accumulations per element pair where forming act and 2 r + s first took five (round 4)
    //   sg = sum z t1                  (d gamma1 ~ sum dA A = (ln 2 / gamma1) sum z t1)
    //   cp = sum z  (= d bias1)
    const float kvn[2] = {th[a.off_ko + cbase + frow] * inv_sw, th[a.off_ko + cbase + 32 + frow] * inv_sw};
    const float gka[2] = {gamma1 * kvn[0] * ak.alpha, gamma1 * kvn[1] * ak.alpha};
    const float gkc[2] = {gamma1 * kvn[0] * ak.c2, gamma1 * kvn[1] * ak.c2};
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      // one LDS base register per 32-row tile (an opaque offset): every store of the tile then fits the 16-bit immediate --
      // hipcc otherwise re-derives the address of each store beyond 64 KiB with a v_add_u32 (83 of them in this phase)
      int tile_off = (rbase + i * 32 + 4 * kg) * kPitchE;
      asm volatile("" : "+v"(tile_off));
      bf16_t* tile_i = tile + tile_off;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int lr = rbase + i * 32 + 8 * rg + 4 * kg;
        const f32x4 dv4 = *reinterpret_cast<const f32x4*>(s_dv + lr);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int lc = cbase + j * 32 + frow;
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            f32x2 tv = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
            asm volatile("" : "+v"(tv));
            const f32x2 dv2 = {dv4[q], dv4[q + 1]};
            const ActCore2 c = panel_act_core2(tv);
            const f32x2 s = kLn2 * c.mxt + c.dl;
            const f32x2 dg = gkc[j] * (c.r - c.r * c.r) + gka[j] * c.dl;
            const f32x2 z = dv2 * dg;
            cr[j] += dv2 * c.r;
            cs[j] += dv2 * s;
            sg[j] += z * tv;
            cp[j] += z;
            if constexpr (C8) {      // dZ_L leaves as e5m2 / s_dZ into the fp8 panel image (rows lr + q, lr + q + 1, column lc)
              store_pair_p8(lr + q, lc, z.x * inv_qdz, z.y * inv_qdz, std::true_type{});
            } else {
              store_pair_pk(tile_i + (8 * rg + q) * kPitchE + lc, tile_i + (8 * rg + q + 1) * kPitchE + lc, z.x, z.y);
            }
          }
        }
        asm volatile("" : "+v"(cr[0]), "+v"(cr[1]), "+v"(sg[0]), "+v"(sg[1]), "+v"(cp[0]), "+v"(cp[1]),
                     "+v"(cs[0]), "+v"(cs[1]));
        if (BNF_EPI_FENCE_EVERY == 1 || (rg & 1)) __builtin_amdgcn_sched_barrier(0);
        if (rg == 1 && i > 0) block_to_global(L, a.dZ[LL], i - 1, cbase);   // (deferred: see the layer-0 forward)
      }
    }
    block_to_global(L, a.dZ[LL], RT - 1, cbase);
    if (hc == CH - 1 && !c8) ring_prefetch(wbl(LL), lane);   // the accumulators are dead: weights of dH = dZ K^T on their way
    const float crj[2] = {cr[0].x + cr[0].y, cr[1].x + cr[1].y}, csj[2] = {cs[0].x + cs[0].y, cs[1].x + cs[1].y};
    float wsa = kvn[0] * (2.f * crj[0] + csj[0]) + kvn[1] * (2.f * crj[1] + csj[1]);
    float wsg = (kLn2 / gamma1) * ((sg[0].x + sg[0].y) + (sg[1].x + sg[1].y));
    float dv_rows = 0.f;     // sum of dv over this wave's WR rows: the row phase's per-wave sums (64 rows each)
#pragma unroll
    for (int w2 = 0; w2 < (WR + 63) / 64; ++w2) dv_rows += s_sc[(rbase / 64 + w2) * 5 + 2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float b = cp[j].x + cp[j].y, k = ak.c1 * crj[j] + ak.alpha * csj[j];
      b += __shfl_xor(b, 32, 64);
      k += __shfl_xor(k, 32, 64);
      k += ak.c0 * dv_rows;
      if (lane < 32) {
        s_col[rb * W + cbase + j * 32 + lane] = b;
        s_col[(RB + rb) * W + cbase + j * 32 + lane] = k;
      }
    }
    wsa_all += panel_wave_sum(wsa);
    wsg_all += panel_wave_sum(wsg);
    if (hc == CH - 1 && lane == 0) {
      s_sc[32 + wave * 2] = wsa_all;
      s_sc[33 + wave * 2] = wsg_all;
    }
  }
  }
  if constexpr (C8 && kL1T) {
    // The column-sum MFMAs above read the bf16 dZ_L panel (each wave its own region); the contraction that follows wants
    // the e5m2 image, whose rows lie elsewhere in the same buffer: one more barrier, then dZ_L = dv dg again (dg is still in
    // the accumulators), scaled by 1 / s_dZ, one dword per four hidden units.
    lds_barrier();
    const LaneCtx L = lane_ctx();
    const int frow = L.frow, kg = L.kg;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const float dvs = s_dv[rbase + i * 32 + frow] * inv_qdz;
      char* rowp8 = smem + (rbase + i * 32 + frow) * kPitch8 + slab(0) * 64 + 4 * kg;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f32x2 z0 = f32x2{accs[0][i][j][rg * 4], accs[0][i][j][rg * 4 + 1]} * dvs;
          const f32x2 z1 = f32x2{accs[0][i][j][rg * 4 + 2], accs[0][i][j][rg * 4 + 3]} * dvs;
          int pk = __builtin_amdgcn_cvt_pk_bf8_f32(z0.x, z0.y, 0, false);
          pk = __builtin_amdgcn_cvt_pk_bf8_f32(z1.x, z1.y, pk, true);
          *reinterpret_cast<int*>(rowp8 + j * 32 + 8 * rg) = pk;
        }
    }
  }
  BNF_MARK(a, 8);
  lds_barrier();
  for (int c = tid; !kL1T && c < W; c += 512) {   // (L1T: the waves added their column sums themselves)
    float b = 0.f, k = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      b += s_col[r * W + c];
      k += s_col[(RB + r) * W + c];
    }
    atomicAdd(&gr[a.off_bias[LL] + c], b);
    atomicAdd(&gr[a.off_ko + c], k * inv_sw);
  }
  float ta1 = 0.f;   // d alpha (layer 1 share), kept by thread 0 until layer 0's share is known
  if (tid == 0) {
    float tg = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) {
      ta1 += s_sc[32 + w2 * 2];
      tg += s_sc[33 + w2 * 2];
    }
    float ko_all = 0.f;     // sum of k_o over all columns: the waves of row block 0 hold one slab each
#pragma unroll
    for (int w2 = 0; w2 < WN; ++w2) ko_all += s_sc[48 + w2];
    ta1 -= 2.f * inv_sw * ko_all * dv_all;   // the "+ 2" of the sa sums
    atomicAdd(&gr[a.off_ls[LL]], dgam1 * tg);
  }

  // =============================== middle layers backward (depth > 2) =========================
  // for l = L-2 .. 1:  dH_{l+1} = dZ_{l+1} K_{l+1}^T (panel contraction), then
  // dZ_l = gamma_l (dH_{l+1} / sqrt W) act'(A_l) with t_l read back from where this wave parked it
#pragma unroll 1
  for (int l = LL - 1; DEEP && l >= 1; --l) {
    if (!BNF_PANEL_ZPEEL) zero_acc();
    if constexpr (c8) contract_all8(a.Wb8[l + 1] + (int64_t)e * a.w8_batch, std::false_type{}, std::true_type{});
    else contract_all(wbl(l + 1));
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    u32x4 pv[2][2];       // parked t of tile (i, j): requested one tile ahead
    auto park_load = [&](const bf16_t* pk, int t2, u32x4 (&dst)[2]) {
      const u32x4* src = reinterpret_cast<const u32x4*>(pk + (min(t2, 2 * RT - 1) * 2 * 64 + lane) * 8);
      dst[0] = __builtin_nontemporal_load(src);
      dst[1] = __builtin_nontemporal_load(src + 64);
    };
    park_load(park_ptr(l, 0), 0, pv[0]);
    if (kL1T && l == LL - 1) copy_panel_by_early_waves(a.dZ[LL]);   // (behind that load: its wait does not cover these stores)
    lds_barrier();     // every wave is done reading dZ_{l+1}: the panel is overwritten with dZ_l
    float sa_all = 0.f, sg_all = 0.f;
#pragma unroll
    for (int hc = 0; hc < CH; ++hc) {
      const int cbase = slab(hc) * 64;
      f32x16 (&acc)[RT][2] = accs[hc];
      const bf16_t* pk = park_ptr(l, hc);
      if (hc > 0) park_load(pk, 0, pv[0]);
      const float gl = sc[l];
      const float gza = gl * inv_sw * ak.alpha, gzc = gl * inv_sw * ak.c2;
      f32x2 sa2 = {0.f, 0.f}, sg2 = {0.f, 0.f}, sacc = {0.f, 0.f}, cs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        int tile_off = (rbase + i * 32 + 4 * kg) * kPitchE;
        asm volatile("" : "+v"(tile_off));
        bf16_t* tile_i = tile + tile_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int t2 = 2 * i + j;
          park_load(pk, t2 + 1, pv[(t2 + 1) & 1]);
          const u32x4 (&cur)[2] = pv[t2 & 1];
          const int lc = cbase + j * 32 + frow;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int lr = rbase + i * 32 + 8 * rg + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const uint32_t w = cur[rg >> 1][(rg & 1) * 2 + (q >> 1)];
              const f32x2 tv = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
              const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
              const ActCore2 c = panel_act_core2(tv);
              const f32x2 s = kLn2 * c.mxt + c.dl;
              const f32x2 dg = gzc * (c.r - c.r * c.r) + gza * c.dl;
              const f32x2 z = raw * dg;
              sa2 += raw * (2.f * c.r + s);
              sacc += raw;
              sg2 += z * tv;
              cs2[j] += z;
              if constexpr (C8) store_pair_p8(lr + q, lc, z.x * inv_qdz, z.y * inv_qdz, std::true_type{});
              else store_pair_pk(tile_i + (8 * rg + q) * kPitchE + lc, tile_i + (8 * rg + q + 1) * kPitchE + lc, z.x, z.y);
            }
            asm volatile("" : "+v"(sa2), "+v"(sg2), "+v"(sacc), "+v"(cs2[0]), "+v"(cs2[1]));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (j == 0 && i > 0) block_to_global(L, a.dZ[l], i - 1, cbase);
        }
      }
      block_to_global(L, a.dZ[l], RT - 1, cbase);
      if (hc == CH - 1 && !c8) ring_prefetch(wbl(l), lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float c = cs2[j].x + cs2[j].y;
        c += __shfl_xor(c, 32, 64);
        if (lane < 32) s_col[rb * W + cbase + j * 32 + lane] = c;
      }
      sa_all += panel_wave_sum(inv_sw * ((sa2.x + sa2.y) - 2.f * (sacc.x + sacc.y)));
      sg_all += panel_wave_sum((kLn2 / gl) * (sg2.x + sg2.y));
      if (hc == CH - 1 && lane == 0) {
        s_sc[32 + wave * 2] = sa_all;
        s_sc[33 + wave * 2] = sg_all;
      }
    }
    lds_barrier();
    for (int c = tid; c < W; c += 512) {
      float b = 0.f;
#pragma unroll
      for (int r = 0; r < RB; ++r) b += s_col[r * W + c];
      atomicAdd(&gr[a.off_bias[l] + c], b);
    }
    if (tid == 0) {
      float tg = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) {
        ta1 += s_sc[32 + w2 * 2];
        tg += s_sc[33 + w2 * 2];
      }
      atomicAdd(&gr[a.off_ls[l]], sigmoidf(th[a.off_ls[l]]) / sc[l] * tg);
    }
    // (the next contraction's barrier-free start is safe: s_col / s_sc are next written after another barrier)
  }

  // =============================== dH1 = dZ1 K1^T ============================================
  BNF_MARK(a, 9);
  if (!BNF_PANEL_ZPEEL) zero_acc();
  if constexpr (c8)
    contract_all8(a.Wb8[1] + (int64_t)e * a.w8_batch, std::integral_constant<bool, F0>{}, std::true_type{});
  else
    contract_all_t(wbl(1), std::integral_constant<bool, F0>{});   // (F0: transposed accumulators for the swapped dZ0 epilogue)
  BNF_MARK(a, 10);
  const LaneCtx L2 = lane_ctx(0);
  l0_weights(L2);                         // first operands of the A0 recomputation, in flight across the barrier
  if (kL1T && LL == 1) copy_panel_by_early_waves(a.dZ[1]);   // (behind those loads: their wait does not cover these stores)
  lds_barrier();     // every wave is done reading dZ1: the panel is overwritten with dZ0 (and s_col / s_sc reused)

  // ---- dZ0 = gamma0 (dH1 / sqrt W) act'(A0), A0 recomputed per 32-row block ----------------
  float sa0_all = 0.f, sg0_all = 0.f;
#pragma unroll
  for (int hc = 0; hc < CH; ++hc) {
    const int cbase = slab(hc) * 64;
    f32x16 (&acc)[RT][2] = accs[hc];
    const LaneCtx L = (hc == 0) ? L2 : lane_ctx(hc);
    if (hc > 0) l0_weights(L);
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    // as in the dZ1 epilogue: dZ0 = z = dH1 (gamma0 / sqrt W) act'(A0) formed directly from the raw
    // accumulator; sa2 = sum raw (elu - tanh + 2) with sacc = sum raw taking the "+ 2" back out,
    // sg2 = sum z t0 (d gamma0 ~ (ln 2 / gamma0) sum z t0), cs2 = sum z (= d bias0)
    const float gs0 = gamma0 * inv_sf * kLog2e;
    float gb0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) gb0[j] = gamma0 * kLog2e * th[a.off_bias[0] + cbase + j * 32 + frow];
    const float gza = gamma0 * inv_sw * ak.alpha, gzc = gamma0 * inv_sw * ak.c2;
    f32x2 sa2 = {0.f, 0.f}, sg2 = {0.f, 0.f}, sacc = {0.f, 0.f}, cs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    f32x16 a0b[2];
    if constexpr (H0L) l0_tile(L, a0b[0], 0, 0);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        constexpr int kDummy = 0;
        (void)kDummy;
        const int t = 2 * i + j;
        f32x16& a0 = a0b[t & 1];
        fair_prio(t);
        if constexpr (H0L) {   // the next tile's MFMA chain runs under this tile's epilogue
          if (t + 1 < 2 * RT) l0_tile(L, a0b[(t + 1) & 1], (t + 1) >> 1, (t + 1) & 1);
        } else {
          if (!BNF_ABL(a, 1)) l0_tile(L, a0, i, j);
          else {
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = 0.25f;
          }
        }
        if constexpr (F0) {
          // transposed tiles (dH1 from the swapped contraction, t0 from the swapped recomputation, scale and bias
          // inside): lane <-> row, four consecutive hidden units per register group -> one 8-byte store; no column sums
          // (d bias0 is the ones row of the layer-0 weight gradient)
          bf16_t* rowp = tile + (rbase + i * 32 + frow) * kPitchE + cbase + j * 32 + 4 * kg;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            f32x2 zq[2];
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
              const f32x2 tv = {a0[rg * 4 + q], a0[rg * 4 + q + 1]};
              f32x2 z = raw;
              if (!BNF_ABL(a, 2)) {
                const ActCore2 c = panel_act_core2(tv);
                const f32x2 s = kLn2 * c.mxt + c.dl;
                const f32x2 dg = gzc * (c.r - c.r * c.r) + gza * c.dl;
                z = raw * dg;
                sa2 += raw * (2.f * c.r + s);
              }
              sacc += raw;
              sg2 += z * tv;
              zq[q >> 1] = z;
            }
            if (!BNF_ABL(a, 4)) store_quad_pk(rowp + 8 * rg, zq[0].x, zq[0].y, zq[1].x, zq[1].y);
            asm volatile("" : "+v"(sa2), "+v"(sg2), "+v"(sacc));
            if (BNF_EPI_FENCE_EVERY == 1 || (rg & 1)) __builtin_amdgcn_sched_barrier(0);
          }
          if (j == 0 && i > 0 && !(BNF_PANEL_DK0 && a.dk0_fused)) block_to_global(L, a.dZ[0], i - 1, cbase);
          continue;
        }
        const int lc = cbase + j * 32 + frow;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int lr = rbase + i * 32 + 8 * rg + 4 * kg;
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            const f32x2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
            const f32x2 tv = f32x2{a0[rg * 4 + q], a0[rg * 4 + q + 1]} * gs0 + gb0[j];
            f32x2 z = raw;
            if (!BNF_ABL(a, 2)) {
              const ActCore2 c = panel_act_core2(tv);
              const f32x2 s = kLn2 * c.mxt + c.dl;
              const f32x2 dg = gzc * (c.r - c.r * c.r) + gza * c.dl;
              z = raw * dg;
              sa2 += raw * (2.f * c.r + s);
            }
            sacc += raw;
            sg2 += z * tv;
            cs2[j] += z;
            if (!BNF_ABL(a, 4)) store_pair_pk(tile + (lr + q) * kPitchE + lc, tile + (lr + q + 1) * kPitchE + lc, z.x, z.y);
          }
          asm volatile("" : "+v"(sa2), "+v"(sg2), "+v"(sacc), "+v"(cs2[0]), "+v"(cs2[1]));
          if (BNF_EPI_FENCE_EVERY == 1 || (rg & 1)) __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 0 && i > 0 && !(BNF_PANEL_DK0 && a.dk0_fused)) block_to_global(L, a.dZ[0], i - 1, cbase);   // (deferred: see the layer-0 forward)
      }
    }
    fair_prio_end();
    if (!(BNF_PANEL_DK0 && a.dk0_fused)) block_to_global(L, a.dZ[0], RT - 1, cbase);
    BNF_MARK(a, 11);
    if constexpr (!F0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float c = cs2[j].x + cs2[j].y;
        c += __shfl_xor(c, 32, 64);
        if (lane < 32) s_col[rb * W + cbase + j * 32 + lane] = c;
      }
    }
    sa0_all += panel_wave_sum(inv_sw * ((sa2.x + sa2.y) - 2.f * (sacc.x + sacc.y)));
    sg0_all += panel_wave_sum((kLn2 / gamma0) * (sg2.x + sg2.y));
    if (hc == CH - 1 && lane == 0) {
      s_sc[32 + wave * 2] = sa0_all;
      s_sc[33 + wave * 2] = sg0_all;
    }
  }
  int4 md_red = {0, 0, 0, 0};   // fused featurisation backward: column table entry of the final reduction (wave 0)
  // =============================== dH0^T = (dZ0 K0^T / sqrt F)^T ==============================
  // Output tiles of 32 x 32 over the waves.  All W/16 weight fragments of a tile are requested
  // at once -- those of the wave's first tile BEFORE the barrier that completes the dZ0 panel --
  // and the contraction runs as two independent accumulator chains.
  {
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    const int ct = a.Fp / 32;                     // column tiles of dH0
    const int n_t = (BM / 32) * ct;
    float* dh0 = a.dH0t + (int64_t)e * a.dh0_batch;
    constexpr int KC = KS1 < 32 ? KS1 : 32;       // weight fragments in registers at a time (128 registers)
    bf16x8 fb[KC];
    const PanelW wb0w = panel_wbase(wb0);       // (raw buffer loads: no 64-bit VALU address per fragment)
    auto load_b = [&](int t, int k0) {
      const uint32_t o = (uint32_t)((t % ct) * KS1 + k0) * 1024u;   // uniform
#pragma unroll
      for (int u = 0; u < KC; ++u) fb[u] = panel_wload(wb0w, o + (uint32_t)u * 1024u, (uint32_t)lane * 16u);
    };
    load_b(wave, 0);   // (unconditional: a wave without a first tile -- 64-row panels -- loads a tile it never uses)
    // fused featurisation backward: the column table entries of this wave's first tile and of the final
    // reduction (wave 0: one column per lane), in flight across the barrier like the weights
    int4 md_first = {0, 0, 0, 0};
    if constexpr (H0L) {
      if (a.fbmeta) {
        md_first = *reinterpret_cast<const int4*>(a.fbmeta + 4 * ((wave % ct) * 32 + frow));
        if (wave < FP / 64) md_red = *reinterpret_cast<const int4*>(a.fbmeta + 4 * (wave * 64 + lane));
      }
    }
    BNF_MARK(a, 12);
    lds_barrier();
    // the column sums and scalars of the dZ0 epilogue are complete: their atomics leave now, under the
    // contraction below, instead of in a serial tail after it
    if constexpr (!F0) {
      for (int c = tid; c < W; c += 512) {
        float b = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) b += s_col[r * W + c];
        atomicAdd(&gr[a.off_bias[0] + c], b);
      }
    }
    if (tid == 0) {
      float ta = ta1, tg = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) {
        ta += s_sc[32 + w2 * 2];
        tg += s_sc[33 + w2 * 2];
      }
      atomicAdd(&gr[a.off_law], alpha * (1.f - alpha) * ta);
      atomicAdd(&gr[a.off_ls[0]], dgam0 * tg);
    }
    for (int t = wave; t < n_t; t += 8) {
      const int mi = t / ct, ni = t - mi * ct;
      const char* ap = smem + (mi * 32 + frow) * kPitchB + kg * 16;
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
#pragma unroll
      for (int k0 = 0; k0 < KS1; k0 += KC) {
        if (t != wave || k0 != 0) load_b(t, k0);
#pragma unroll
        for (int u = 0; u < KC; u += 2) {
          const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(ap + (k0 + u) * 32);
          const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(ap + (k0 + u + 1) * 32);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[u], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb[u + 1], c1, 0, 0, 0);
        }
      }
      if constexpr (H0L) {
        if (a.fbmeta) {
          // featurisation backward on the tile: with H0 = softplus(scale_g) G,
          //   d scale_g            ~ sum_r dH0[r][f] H0[r][f]                       (every column of group g)
          //   d lsa_d (input / interaction columns)  ~ the same products
          //   d lsa_d (Fourier cos / sin column k)   ~ -+ 2 pi 2^k sum_r dH0[r][f] H0[r][partner] u_d[r],
          //                                            u_d = H0[r][input column d] / softplus(scale_in)
          const int f = ni * 32 + frow;
          const int4 md = (t == wave) ? md_first : *reinterpret_cast<const int4*>(a.fbmeta + 4 * f);
          const char* hc = h0s + (mi * 32 + 4 * kg) * kH0Pitch;
          const int o_f = f * 2, o_p = (md.y & 0xff) * 2, o_u = ((md.y >> 8) & 0xff) * 2;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const char* hr = hc + (8 * rg + j) * kH0Pitch;
              const float d = c0[rg * 4 + j] + c1[rg * 4 + j];
              const float hf = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(hr + o_f));
              const float hp = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(hr + o_p));
              const float hu = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(hr + o_u));
              s1 += d * hf;
              s2 += d * hp * hu;
            }
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (lane < 32) {   // this wave is the only writer of (mi, f)
            s_fb[mi * FP + f] = s1 * inv_sf;
            s_fb[(BM / 32) * FP + mi * FP + f] = s2 * inv_sf;
          }
          continue;          // dH0^T itself is not needed any more
        }
      }
      float* col_ptr = dh0 + (int64_t)(ni * 32 + frow) * a.ldt + m0 + mi * 32 + 4 * kg;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        store4(col_ptr + 8 * rg, (c0[rg * 4] + c1[rg * 4]) * inv_sf, (c0[rg * 4 + 1] + c1[rg * 4 + 1]) * inv_sf,
               (c0[rg * 4 + 2] + c1[rg * 4 + 2]) * inv_sf, (c0[rg * 4 + 3] + c1[rg * 4 + 3]) * inv_sf);
    }
  }
#if BNF_PANEL_DK0
  // ---- experiment: dK_0[f][c] += sum_r H0[r][f] dZ0[r][c] / sqrt F for this wave's 64 columns (both panels in LDS) ----
  // 2 x 2 MFMA tiles (features x columns), K = the panel's BM rows; fragments by ds_read_b64_tr_b16 as in gemm_tn
  // (within a 16-lane group lane i receives element i % 4 of the 8-byte datum addressed by lane 4 j + i / 4).
  if constexpr (H0L && CH == 1 && FP == 64 && RB == 1) {
    if (a.dk0_fused) {
      const int lane = opaque_lane(tid) & 63;
      const int kg = lane >> 5, frow = lane & 31, p = lane & 15, half = (lane >> 4) & 1;
      const int prow = p >> 2, pcol = half * 16 + (p & 3) * 4;
      const int cbase = slab(0) * 64;
      typedef __attribute__((address_space(3))) char lds_char_t;
      const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)smem;
      const uint32_t h0o = lds0 + (uint32_t)(h0s - smem);
      f32x16 dk[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dk[i][j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < BM / 16; ++ks) {
        u32x2_t ra[2][2], rb[2][2];   // [t][tile]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = ks * 16 + kg * 8 + t * 4 + prow;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ra[t][i] = lds_tr16_b64<0>(h0o + (uint32_t)(row * kH0Pitch + (i * 32 + pcol) * 2));
            rb[t][i] = lds_tr16_b64<0>(lds0 + (uint32_t)(row * kPitchB + (cbase + i * 32 + pcol) * 2));
          }
        }
        lds_tr_fence(ra, rb);
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 wa = {ra[0][i].x, ra[0][i].y, ra[1][i].x, ra[1][i].y};
          const u32x4 wb = {rb[0][i].x, rb[0][i].y, rb[1][i].x, rb[1][i].y};
          fa[i] = __builtin_bit_cast(bf16x8, wa);
          fb[i] = __builtin_bit_cast(bf16x8, wb);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) dk[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], dk[i][j], 0, 0, 0);
      }
      float* out = gr + a.off_k0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = i * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
            if (f < a.F) atomicAdd(&out[(int64_t)f * W + cbase + j * 32 + frow], dk[i][j][r] * inv_sf);
          }
    }
  }
#endif
  BNF_MARK(a, 13);
  if constexpr (H0L) {
    if (a.fbmeta) {
      lds_barrier();
      if (wave < FP / 64) {                 // one feature column per lane
        const int f = wave * 64 + (opaque_lane(tid) & 63);
        const int4 md = md_red;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < BM / 32; ++mi) {
          t1 += s_fb[mi * FP + f];
          t2 += s_fb[(BM / 32) * FP + mi * FP + f];
        }
        const int kind = md.x & 0xff, g = (md.x >> 8) & 0xff, d1 = (md.x >> 16) & 0xff, d2 = (md.x >> 24) & 0xff;
        const float sp_in = sp_in_u;
        if (g < BNF_MAX_GROUPS) atomicAdd(&s_grp[g], t1);                       // LDS atomics
        float v = 0.f;
        if (kind == kFbInput || kind == kFbInter) v = t1;
        else if (kind == kFbFourier) v = __int_as_float(md.z) * t2 / sp_in;
        if (d1 < BNF_MAX_INPUTS) atomicAdd(&s_grp[BNF_MAX_GROUPS + d1], v);
        if (d2 < BNF_MAX_INPUTS) atomicAdd(&s_grp[BNF_MAX_GROUPS + d2], v);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS atomics have landed
        __builtin_amdgcn_wave_barrier();
      }
      if constexpr (FP > 64) lds_barrier();   // the columns are spread over two waves
      if (wave == 0) {
        const int f = opaque_lane(tid) & 63;
        if (f < a.n_groups) {
          atomicAdd(&gr[s_goff[f]], s_gfac[f] * s_grp[f]);
        } else if (f >= BNF_MAX_GROUPS && f < BNF_MAX_GROUPS + a.n_inputs) {
          atomicAdd(&gr[a.off_lsa + f - BNF_MAX_GROUPS], -s_grp[f]);
        }
      }
    }
  }
  BNF_MARK(a, 14);
  BNF_MARK(a, 15);
}

}  // namespace bnf
