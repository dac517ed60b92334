// bnf_panel2.h -- the row-panel kernel of bnf_panel.h with the workgroup split into TWO
// independent groups of four waves, one wave of each group per SIMD (W = 512, Fp = 64, bf16).
//
// Why: in k_panel_fwd_bwd all eight waves walk through the same phase at the same time, so a SIMD
// either runs two waves of MFMA (contractions: 26 % of a panel) or two waves of VALU (activation
// epilogues: 50 %) -- the matrix pipe idles through the epilogues and the VALU through the
// contractions (profiles/r02_panel_phase_clocks.md; SQ_VALU_MFMA_COEXEC 11 % of MFMA-busy).  Here
// each group owns a 64-row half of the 128-row panel (own LDS half-panel, own scratch, own feature
// stage) and synchronises only with itself through a counter in LDS -- the two groups never meet
// at a barrier, drift out of phase, and a SIMD's two waves mostly want different pipes.
//
// Geometry: group g = wave / 4 owns panel rows [64 g, 64 g + 64); wave gw = wave % 4 of a group owns
// columns [128 gw, 128 gw + 128) of every 64 x 512 activation: 2 x 4 MFMA 32x32 accumulators = 128
// registers (as before), four weight-fragment streams per wave.  A 64-row half re-reads the
// weights twice as often as the 128-row panel did (1 MiB per contraction and CU instead of 512 KiB):
// 64 B/clk/CU at full MFMA rate against the 51 B/clk/CU an L2-resident stream delivers
// (profiles/r02a_panel_probe.txt) -- the contraction of a group running alone is stream-bound at
// ~80 % of the MFMA rate, which is the price of the overlap.
#pragma once

#include "bnf_panel.h"

namespace bnf {

// Element pairs of the epilogues.  The packed f32 instructions (v_pk_fma_f32 ...) a float2 vector
// type compiles to cost 4 cycles per wave on the SIMD-32 -- no faster than two scalar ops -- and
// are documented as an anti-lever next to MFMAs (MI355X_MICROARCH.md, price list); with
// -DBNF_SCALAR_EPI the same formulas run on plain v_fma_f32 / v_mul_f32 (perf experiment).
#ifdef BNF_SCALAR_EPI
struct P2 {
  float x, y;
};
__device__ __forceinline__ P2 operator+(P2 a, P2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ P2 operator-(P2 a, P2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ P2 operator*(P2 a, P2 b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ P2 operator*(P2 a, float b) { return {a.x * b, a.y * b}; }
__device__ __forceinline__ P2 operator*(float b, P2 a) { return {a.x * b, a.y * b}; }
__device__ __forceinline__ P2 operator+(P2 a, float b) { return {a.x + b, a.y + b}; }
__device__ __forceinline__ P2 operator-(P2 a, float b) { return {a.x - b, a.y - b}; }
__device__ __forceinline__ P2& operator+=(P2& a, P2 b) { a.x += b.x; a.y += b.y; return a; }
#else
using P2 = f32x2;
#endif
#ifdef BNF_SCALAR_EPI
__device__ __forceinline__ void pin(P2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
#else
__device__ __forceinline__ void pin(P2& v) { asm volatile("" : "+v"(v)); }
#endif
struct ActP2 {
  P2 h, dact, ediff;
};
__device__ __forceinline__ P2 p2_parts(P2 a, P2& r_out, P2& el_out) {
  const P2 t = a * 1.44269504088896340736f;
  const P2 e1 = {BNF_EXP2(-fabsf(t.x)), BNF_EXP2(-fabsf(t.y))};
  const P2 den = e1 * e1 + 1.f;
  float rx = BNF_RCP(den.x), ry = BNF_RCP(den.y);
  asm volatile("" : "+v"(rx), "+v"(ry));
  const P2 r = {rx, ry};
  const P2 tha = 2.f * r - 1.f;
  const P2 em1 = e1 - 1.f;
  el_out = P2{vmaxf(a.x, em1.x), vmaxf(a.y, em1.y)};
  r_out = r;
  return P2{copysignf(tha.x, a.x), copysignf(tha.y, a.y)};
}
__device__ __forceinline__ P2 p2_fwd(P2 a, float alpha) {
  P2 r, el;
  const P2 th = p2_parts(a, r, el);
  return th + alpha * (el - th);
}
__device__ __forceinline__ ActP2 p2_eval(P2 a, float alpha) {
  ActP2 o;
  P2 r, el;
  const P2 th = p2_parts(a, r, el);
  const P2 mn = {vminf(el.x, 0.f), vminf(el.y, 0.f)};
  o.ediff = el - th;
  o.h = th + alpha * o.ediff;
  o.dact = (4.f * (1.f - alpha)) * (r - r * r) + (alpha * mn + alpha);
  return o;
}

constexpr int kP2W = 512, kP2Rows = 64, kP2PitchE = kP2W + 8, kP2PitchB = 2 * kP2PitchE;
constexpr int kP2XsFloats = 64 * 4 + 64 + 2 * kP2W + 64;          // s_part, s_dv, s_col, s_sc per group
constexpr int kP2H0Pitch = 144;
// GROUPS = 2: one 512-thread workgroup per CU holding both groups (group barriers through LDS);
// GROUPS = 1: a workgroup IS one group (256 threads, 64-row panel, plain s_barrier), two resident
// per CU -- the hardware starts the next panel the moment a workgroup retires, so the two
// co-resident workgroups de-phase by themselves and no ramp-up / ramp-down is paid per pair.
__host__ __device__ constexpr int panel2_lds_bytes(int groups) {
  return groups * (kP2Rows * kP2PitchB + kP2XsFloats * 4 + kP2Rows * kP2H0Pitch) + 64;
}

// acc[2][4] += P[64 rows of the group][0 .. 16 KS) . Bt-fragments of four column tiles (nt0 .. nt0 + 3)
template <int PD>
__device__ __forceinline__ void panel2_contract(f32x16 (&acc)[2][4], const char* prow, const char* wp, int nt0, int KS,
                                                int lane) {
  const char* w0 = wp + (size_t)nt0 * KS * 1024;   // uniform; stream j at + j * KS KiB
  const uint32_t loff = (uint32_t)lane * 16u;
  bf16x8 fb[PD][4];
#pragma unroll
  for (int p = 0; p < PD; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(w0 + ((size_t)j * KS + p) * 1024 + loff);
  auto load_a = [&](bf16x8 (&fa)[2], int ks) {
    fa[0] = *reinterpret_cast<const bf16x8*>(prow + ks * 32);
    fa[1] = *reinterpret_cast<const bf16x8*>(prow + 32 * kP2PitchB + ks * 32);
  };
  bf16x8 fa[2][2];
  load_a(fa[0], 0);
#pragma unroll 1
  for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int cur = p & 1;
      load_a(fa[cur ^ 1], min(ks0 + p + 1, KS - 1));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int kn = min(ks0 + PD + p, KS - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(w0 + ((size_t)j * KS + kn) * 1024 + loff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int GROUPS>
__global__ __launch_bounds__(256 * GROUPS, 2) void k_panel2_fwd_bwd(const PanelArgs a) {
  constexpr int W = kP2W, BMg = kP2Rows, BM = GROUPS * BMg, kPitchE = kP2PitchE, kPitchB = kP2PitchB;
  constexpr int KS1 = W / 16, KS0 = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, gw = wave & 3;      // group, wave within the group (= 128-column slab)
  const int gt = tid & 255;                    // thread within the group
  char* Pg = smem + g * (BMg * kPitchB);
  bf16_t* tile = reinterpret_cast<bf16_t*>(Pg);
  float* xs = reinterpret_cast<float*>(smem + GROUPS * BMg * kPitchB) + g * kP2XsFloats;
  float* s_part = xs;                       // [64][4] row-dot partials
  float* s_dv = s_part + 64 * 4;            // [64]
  float* s_col = s_dv + 64;                 // [2][W] column sums
  float* s_sc = s_col + 2 * W;              // scalars
  char* h0s = smem + GROUPS * (BMg * kPitchB + kP2XsFloats * 4) + g * (BMg * kP2H0Pitch);
  uint32_t* bars = reinterpret_cast<uint32_t*>(smem + GROUPS * (BMg * kPitchB + kP2XsFloats * 4 + BMg * kP2H0Pitch));
  uint32_t* bar = bars + g * 4;

  const uint32_t item = xcd_remap(blockIdx.x, gridDim.x);
  const int e = (int)(item / (uint32_t)a.panels), pn = (int)(item % (uint32_t)a.panels);
  const int m0 = pn * BM + g * BMg;         // first batch row of this group's half panel
  const int cbase = gw * 128;

  const float* th = a.theta + (int64_t)e * a.theta_stride;
  const float* sc = a.scal + (int64_t)e * kScalStride;
  const float gamma0 = sc[0], gamma1 = sc[1], alpha = sc[BNF_MAX_LAYERS];
  const float inv_sw = 1.0f / sqrtf((float)W), inv_sf = 1.0f / sqrtf((float)a.F);
  float* gr = a.grad + (int64_t)e * a.grad_stride;
  const float gam_o = sc[BNF_MAX_LAYERS + 1], bias_o = th[a.off_bias_out];
  const float dgam_o = sigmoidf(th[a.off_os]);
  const float dgam1 = sigmoidf(th[a.off_ls1]) / gamma1, dgam0 = sigmoidf(th[a.off_ls0]) / gamma0;
  const float lns = th[a.off_lns];
  const float e_lns = expf(lns), sigma = 0.01f + e_lns, inv_sigma = 1.0f / sigma;
  const float ll_const = -logf(sigma) - 0.918938533204672742f;
  const float y_row = (gt < BMg && m0 + gt < a.B) ? a.ybat[(int64_t)e * a.row_batch + m0 + gt] : 0.f;

  const char* wf0 = reinterpret_cast<const char*>(a.Wf0 + (int64_t)e * a.w0_batch);
  const char* wf1 = reinterpret_cast<const char*>(a.Wf1 + (int64_t)e * a.w1_batch);
  const char* wb1 = reinterpret_cast<const char*>(a.Wb1 + (int64_t)e * a.w1_batch);
  const char* wb0 = reinterpret_cast<const char*>(a.Wb0 + (int64_t)e * a.w0_batch);

  // ---- group barrier: a counter in LDS, four arrivals per generation --------------------------
  if (tid == 0) { bars[0] = 0u; bars[4] = 0u; }
  {   // feature half-panel -> LDS (row-major source, 64 rows x 8 chunks of 16 bytes = 2 per thread)
    const bf16_t* src = a.H0rm + (int64_t)e * a.h0_batch + (int64_t)m0 * 64;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = gt + c * 256;
      *reinterpret_cast<u32x4*>(h0s + (q >> 3) * kP2H0Pitch + (q & 7) * 16) =
          *reinterpret_cast<const u32x4*>(src + (int64_t)(q >> 3) * 64 + (q & 7) * 8);
    }
  }
  __syncthreads();                              // the only workgroup-wide barrier
  uint32_t bar_target = 0;
  auto group_barrier = [&]() {
    if constexpr (GROUPS == 1) {
      lds_barrier();
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar_target += 4;
    if ((tid & 63) == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  if (GROUPS == 2 && g == 1) {   // seed the phase offset between the groups (a.ablate >> 8 = units of 64 cycles; perf knob)
    for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(1);
  }

  struct LaneCtx {
    int lane, frow, kg;
    const char* prow;
  };
  auto lane_ctx = [&]() {
    LaneCtx c;
    c.lane = opaque_lane(tid) & 63;
    c.frow = c.lane & 31;
    c.kg = c.lane >> 5;
    c.prow = Pg + c.frow * kPitchB + c.kg * 16;
    return c;
  };
  // this wave's own 32-row x 128-column block (row block i) of the half panel -> row-major (Bp, W)
  auto block_to_global = [&](const LaneCtx& L, bf16_t* dst, int i) {
    bf16_t* d = dst + (int64_t)e * a.act_batch + (int64_t)(m0 + i * 32) * W + cbase;
    const bf16_t* sp = tile + (i * 32) * kPitchE + cbase;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = L.lane + 64 * u;           // 32 rows x 16 chunks
      v[u] = *reinterpret_cast<const u32x4*>(sp + (idx >> 4) * kPitchE + (idx & 15) * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = L.lane + 64 * u;
      *reinterpret_cast<u32x4*>(d + (int64_t)(idx >> 4) * W + (idx & 15) * 8) = v[u];
    }
  };

  BNF_MARK(a, 0);
  f32x16 acc[2][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // layer-0 tile (row block i, column tile j of this wave): A fragments from the staged feature
  // panel, weights of the column-tile PAIR jp resident in registers
  bf16x8 bres[2][4];
  auto l0_weights = [&](const LaneCtx& L, int jp) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        bres[jj][u] = *reinterpret_cast<const bf16x8*>(wf0 + ((size_t)(4 * gw + 2 * jp + jj) * KS0 + u) * 1024 + L.lane * 16);
  };
  auto l0_tile = [&](const LaneCtx& L, f32x16& a0, int i, int jj) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = 0.f;
    const char* ap = h0s + (i * 32 + L.frow) * kP2H0Pitch + L.kg * 16;
    bf16x8 fa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(ap + u * 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u], bres[jj][u], a0, 0, 0, 0);
  };

  // =============================== layer 0 forward -> H1 half panel ===========================
  {
    const LaneCtx L = lane_ctx();
    const int frow = L.frow, kg = L.kg;
    const float gs = gamma0 * inv_sf;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      l0_weights(L, jp);
      float gb[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) gb[jj] = gamma0 * th[a.off_bias0 + cbase + (2 * jp + jj) * 32 + frow];
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {
        f32x16 a0b[2];
        l0_tile(L, a0b[0], i, 0);
        l0_tile(L, a0b[1], i, 1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int lc = cbase + (2 * jp + jj) * 32 + frow;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int lr = i * 32 + 8 * rg + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const P2 av = P2{a0b[jj][rg * 4 + q], a0b[jj][rg * 4 + q + 1]} * gs + gb[jj];
              const P2 h = p2_fwd(av, alpha);
              store_pair(tile + (lr + q) * kPitchE + lc, tile + (lr + q + 1) * kPitchE + lc, h.x, h.y);
            }
          }
        }
      }
    }
    block_to_global(L, a.H1, 0);
    block_to_global(L, a.H1, 1);
  }
  BNF_MARK(a, 1);
  group_barrier();
  BNF_MARK(a, 2);

  // =============================== layer 1 forward ============================================
  zero_acc();
  {
    const LaneCtx L = lane_ctx();
    panel2_contract<kPanelPD>(acc, L.prow, wf1, 4 * gw, KS1, L.lane);
  }
  BNF_MARK(a, 3);
  group_barrier();     // every wave of the group is done reading H1: the half panel doubles as row-dot scratch

  // ---- A1 = gamma1 (acc / sqrt W + b1) kept in the accumulators; row dots act(A1) . k_o ----
  {
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    const float gs = gamma1 * inv_sw;
    float gb[4], kov[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gb[j] = gamma1 * th[a.off_bias1 + cbase + j * 32 + frow];
      kov[j] = th[a.off_ko + cbase + j * 32 + frow];
    }
    float* s_dot = reinterpret_cast<float*>(Pg) + gw * (64 * kRowDotPitch);   // [64 rows][32 lanes]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            const P2 raw = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
            const P2 av = raw * gs + gb[j];
            acc[i][j][rg * 4 + q] = av.x;
            acc[i][j][rg * 4 + q + 1] = av.y;
            const P2 hk = p2_fwd(av, alpha) * kov[j];
            pd[q] += hk.x;
            pd[q + 1] += hk.y;
          }
        }
        float* dst = s_dot + (i * 32 + 8 * rg + 4 * kg) * kRowDotPitch + frow;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * kRowDotPitch] = pd[q];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const f32x4* rp = reinterpret_cast<const f32x4*>(s_dot + lane * kRowDotPitch);
    f32x4 t4 = rp[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) t4 += rp[c];
    s_part[lane * 4 + gw] = (t4.x + t4.y) + (t4.z + t4.w);
  }
  BNF_MARK(a, 4);
  group_barrier();
  // ---- one thread per row (the group's first wave): output, likelihood, d out ---------------
  {
    if (gt < BMg) {
      const int m = m0 + gt;
      float ll = 0.f, s_doutv = 0.f, s_dvsum = 0.f, s_par = 0.f, s_infl = 0.f;
      const float vsum = (s_part[gt * 4] + s_part[gt * 4 + 1]) + (s_part[gt * 4 + 2] + s_part[gt * 4 + 3]);
      float dvv = 0.f;
      if (m < a.B) {
        const float v = vsum * inv_sw + bias_o;
        const float outv = gam_o * v;
        a.out[(int64_t)e * a.out_batch + m] = outv;
        RowLoss rl;
        if (a.obs == BNF_OBS_NORMAL) {
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
      s_dv[gt] = dvv;
      const float t0 = wave_sum(ll), t1 = wave_sum(s_doutv), t2 = wave_sum(s_dvsum), t3 = wave_sum(s_par),
                  t4 = wave_sum(s_infl);
      if (gt == 0) {
        const float step_loss = -a.lik_c * t0;
        atomicAdd(&a.loss[(int64_t)(e / a.S) * a.loss_stride + (a.st ? a.st->col : 0)], a.loss_scale * step_loss);
        if (a.loss_raw) atomicAdd(&a.loss_raw[e], step_loss);
        atomicAdd(&gr[a.off_os], dgam_o * t1);
        atomicAdd(&gr[a.off_bias_out], t2);
        atomicAdd(&gr[a.obs == BNF_OBS_NORMAL ? a.off_lns : a.off_shape], t3);
        if (a.obs == BNF_OBS_ZINB) atomicAdd(&gr[a.off_infl], t4);
      }
    }
  }
  group_barrier();
  BNF_MARK(a, 5);
  // ---- dZ1 = gamma1 (dv k_o / sqrt W) act'(A1) -> half panel; column sums, scalar gradients ----
  float ta1 = 0.f;
  {
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    P2 sa = {0.f, 0.f}, sg = {0.f, 0.f};          // already weighted with k_o / sqrt W per column
    float kvn[4], gk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kvn[j] = th[a.off_ko + cbase + j * 32 + frow] * inv_sw;
      gk[j] = gamma1 * kvn[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P2 cp = {0.f, 0.f}, ck = {0.f, 0.f};
      const int lc = cbase + j * 32 + frow;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int lr = i * 32 + 8 * rg + 4 * kg;
          const f32x4 dv4 = *reinterpret_cast<const f32x4*>(s_dv + lr);
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            P2 av = {acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]};
            pin(av);
            const P2 dv2 = {dv4[q], dv4[q + 1]};
            const ActP2 o = p2_eval(av, alpha);
            const P2 p = dv2 * o.dact;
            sa += (dv2 * o.ediff) * kvn[j];
            sg += (p * av) * kvn[j];
            cp += p;
            ck += o.h * dv2;
            const P2 z = gk[j] * p;
            store_pair(tile + (lr + q) * kPitchE + lc, tile + (lr + q + 1) * kPitchE + lc, z.x, z.y);
          }
          pin(sa); pin(sg); pin(cp); pin(ck);
          __builtin_amdgcn_sched_barrier(0);
        }
      float b = gk[j] * (cp.x + cp.y), k = ck.x + ck.y;
      b += __shfl_xor(b, 32, 64);
      k += __shfl_xor(k, 32, 64);
      if (lane < 32) {
        s_col[cbase + j * 32 + lane] = b;
        s_col[W + cbase + j * 32 + lane] = k;
      }
    }
    block_to_global(L, a.dZ1, 0);
    block_to_global(L, a.dZ1, 1);
    const float wsa = wave_sum(sa.x + sa.y), wsg = wave_sum(sg.x + sg.y);
    if (lane == 0) {
      s_sc[gw * 2] = wsa;
      s_sc[gw * 2 + 1] = wsg;
    }
  }
  BNF_MARK(a, 6);
  group_barrier();
  for (int c = gt; c < W; c += 256) {
    atomicAdd(&gr[a.off_bias1 + c], s_col[c]);
    atomicAdd(&gr[a.off_ko + c], s_col[W + c] * inv_sw);
  }
  if (gt == 0) {
    float tg = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      ta1 += s_sc[w2 * 2];
      tg += s_sc[w2 * 2 + 1];
    }
    atomicAdd(&gr[a.off_ls1], dgam1 * tg);
  }

  // =============================== dH1 = dZ1 K1^T ============================================
  BNF_MARK(a, 7);
  zero_acc();
  {
    const LaneCtx L = lane_ctx();
    panel2_contract<kPanelPD>(acc, L.prow, wb1, 4 * gw, KS1, L.lane);
  }
  BNF_MARK(a, 8);
  group_barrier();     // the group is done reading dZ1: the half panel is overwritten with dZ0 (s_col / s_sc reused)

  // ---- dZ0 = gamma0 (dH1 / sqrt W) act'(A0), A0 recomputed per 32 x 32 tile ------------------
  {
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    const float gs0 = gamma0 * inv_sf;
    P2 sa2 = {0.f, 0.f}, sg2 = {0.f, 0.f};
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      l0_weights(L, jp);
      float gb0[2];
      P2 cs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) gb0[jj] = gamma0 * th[a.off_bias0 + cbase + (2 * jp + jj) * 32 + frow];
      f32x16 a0b[2];
      l0_tile(L, a0b[0], 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int t = 2 * i + jj, j = 2 * jp + jj;
          f32x16& a0 = a0b[t & 1];
          if (t + 1 < 4) l0_tile(L, a0b[(t + 1) & 1], (t + 1) >> 1, (t + 1) & 1);   // next tile's MFMAs under this epilogue
          const int lc = cbase + j * 32 + frow;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int lr = i * 32 + 8 * rg + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const P2 dh = P2{acc[i][j][rg * 4 + q], acc[i][j][rg * 4 + q + 1]} * inv_sw;
              const P2 a2 = P2{a0[rg * 4 + q], a0[rg * 4 + q + 1]} * gs0 + gb0[jj];
              const ActP2 o = p2_eval(a2, alpha);
              sa2 += dh * o.ediff;
              const P2 da = dh * o.dact;
              sg2 += da * a2;
              const P2 z = gamma0 * da;
              cs2[jj] += z;
              store_pair(tile + (lr + q) * kPitchE + lc, tile + (lr + q + 1) * kPitchE + lc, z.x, z.y);
            }
            pin(sa2); pin(sg2); pin(cs2[0]); pin(cs2[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        float c = cs2[jj].x + cs2[jj].y;
        c += __shfl_xor(c, 32, 64);
        if (lane < 32) s_col[cbase + (2 * jp + jj) * 32 + lane] = c;
      }
    }
    block_to_global(L, a.dZ0, 0);
    block_to_global(L, a.dZ0, 1);
    const float sa = wave_sum(sa2.x + sa2.y), sg = wave_sum(sg2.x + sg2.y);
    if (lane == 0) {
      s_sc[gw * 2] = sa;
      s_sc[gw * 2 + 1] = sg;
    }
  }
  // =============================== dH0^T = (dZ0 K0^T / sqrt F)^T ==============================
  {
    const LaneCtx L = lane_ctx();
    const int lane = L.lane, frow = L.frow, kg = L.kg;
    const int mi = gw >> 1, ni = gw & 1;          // 2 x 2 output tiles of 32 x 32, one per wave (Fp = 64)
    float* dh0 = a.dH0t + (int64_t)e * a.dh0_batch;
    bf16x8 fb[KS1];
    const char* bp = wb0 + (size_t)ni * KS1 * 1024 + lane * 16;
#pragma unroll
    for (int u = 0; u < KS1; ++u) fb[u] = *reinterpret_cast<const bf16x8*>(bp + (size_t)u * 1024);
    BNF_MARK(a, 9);
    group_barrier();
    const char* ap = Pg + (mi * 32 + frow) * kPitchB + kg * 16;
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KS1; u += 2) {
      const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(ap + u * 32);
      const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(ap + (u + 1) * 32);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[u], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb[u + 1], c1, 0, 0, 0);
    }
    float* col_ptr = dh0 + (int64_t)(ni * 32 + frow) * a.ldt + m0 + mi * 32 + 4 * kg;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      store4(col_ptr + 8 * rg, (c0[rg * 4] + c1[rg * 4]) * inv_sf, (c0[rg * 4 + 1] + c1[rg * 4 + 1]) * inv_sf,
             (c0[rg * 4 + 2] + c1[rg * 4 + 2]) * inv_sf, (c0[rg * 4 + 3] + c1[rg * 4 + 3]) * inv_sf);
  }
  BNF_MARK(a, 10);
  for (int c = gt; c < W; c += 256) atomicAdd(&gr[a.off_bias0 + c], s_col[c]);
  if (gt == 0) {
    float ta = ta1, tg = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      ta += s_sc[w2 * 2];
      tg += s_sc[w2 * 2 + 1];
    }
    atomicAdd(&gr[a.off_law], alpha * (1.f - alpha) * ta);
    atomicAdd(&gr[a.off_ls0], dgam0 * tg);
  }
  BNF_MARK(a, 11);
}

}  // namespace bnf
