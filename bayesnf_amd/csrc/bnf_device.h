// bnf_device.h -- device-side building blocks (gfx950 / CDNA4 only).
//
//   * storage-type traits for the two arithmetic modes (f32 / bf16 operands,
//     f32 accumulation in both)
//   * the BayesNF scalar math: softplus, sigmoid, the mixed elu/tanh activation
//     (reference models.py:258-262) in an accurate and a fast formulation
//   * counter-based random numbers: Philox4x32-10, truncated normal, Box-Muller
//   * the keyed Feistel bijection used as a memory-free row shuffle
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Perf-experiment switches (env BNF_ABLATE) exist only in builds made with
// -DBNF_ENABLE_ABLATE (make ABLATE=1); production kernels carry no such branches.
#ifdef BNF_ENABLE_ABLATE
#define BNF_ABL(args, bit) ((args).ablate & (bit))
#define BNF_MARK(args, k) do { if ((args).prof && threadIdx.x == (((args).ablate >> 8) & 0x3ff)) (args).prof[(size_t)blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define BNF_ABL(args, bit) false
#define BNF_MARK(args, k) do { } while (0)
#endif

namespace bnf {

// Per-step quantities of a train loop replayed from a hipGraph (bnf_train, launch-bound sizes): a
// captured launch freezes its by-value arguments, so what changes from step to step lives in
// device memory and is advanced by k_step_advance at the end of every step.  Kernels take a
// `const StepState*` (null in the eager path, which passes the same quantities by value).
struct StepState {
  long long t;        // 1-based Adam step about to be taken
  long long col;      // loss column of this step: loss pointers are offset by it
  float bc1, bc2;     // Adam bias corrections 1 - 0.9^t, 1 - 0.999^t
  float pad[2];
};

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ---------------------------------------------------------------------------
// storage types
// ---------------------------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// round-to-nearest-even conversions; hipcc lowers these casts to v_cvt_pk_bf16_f32
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kBytes = 4;
  // The activation of the f32 pipeline through the hardware exp2 / rcp too (1 ulp each; tanh = 1 - 2 r and elu + 1 =
  // ln2 max(t, 0) + min(e, 1) are absolute-accurate to ~2e-7, which is what sums of O(1) activations need).  Against
  // the float64 oracle it is as accurate as ocml's tanhf / expm1f / expf (loss 2.2e-7 vs 2.0e-7, worst gradient leaf
  // 3.7e-6 vs 7.6e-7 at W = 512 and 4.6e-6 both at depth 3, 30 Adam steps identical: profiles/r03_fp32_activation_accuracy.txt;
  // every fp32 parity bar and golden passes either way) and the C2 step is 13 % shorter (the libm-class activation
  // was 150 instructions per element in the contraction epilogues).  -DBNF_FP32_FAST=0 builds the ocml form.
#ifndef BNF_FP32_FAST
#define BNF_FP32_FAST 1
#endif
  static constexpr bool kFast = BNF_FP32_FAST != 0;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float round(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr bool kFast = true;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_bits_to_f32(p->bits); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { p->bits = f32_to_bf16_bits(v); }
  __device__ static __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

// store two values to two (unrelated) addresses with ONE conversion instruction for bf16
// (v_cvt_pk_bf16_f32 rounds both; the halves leave as ds_write_b16 / ds_write_b16_d16_hi)
__device__ __forceinline__ void store_pair(float* p0, float* p1, float a, float b) { *p0 = a; *p1 = b; }
__device__ __forceinline__ void store_pair(bf16_t* p0, bf16_t* p1, float a, float b) {
  const uint32_t pk = pack_bf16x2(a, b);
  p0->bits = (uint16_t)(pk & 0xffffu);
  p1->bits = (uint16_t)(pk >> 16);
}

// store 4 consecutive elements (used for the transposed copies: 4 consecutive
// rows of one column).  p must be aligned to 4 elements.
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = pack_bf16x2(a, b);
  v.y = pack_bf16x2(c, d);
  *reinterpret_cast<uint2*>(p) = v;
}

// vector loads / stores of 4 or 8 consecutive elements (p aligned to the vector)
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
  o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
}
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __builtin_bit_cast(float, v[i] << 16);
    o[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  store4(p, v[0], v[1], v[2], v[3]);
  store4(p + 4, v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = w;
}

// 4 consecutive floats at ANY 4-byte aligned address (gfx950 global memory takes dword-aligned
// dwordx4 accesses; hipcc emits global_load/store_dwordx4 for this type).  The row stride of the
// parameter-shaped arrays (P floats per member) is not a multiple of 4 in general, and a 16-byte access
// per lane streams at ~2x the rate of a 4-byte one (MI355X_MICROARCH.md: 8-byte accesses reach
// 0.54 - 0.70 of the 16-byte rate).  nv < 4: the row's tail, element by element.
typedef f32x4 __attribute__((aligned(4))) f32x4u;
template <bool NT = false>
__device__ __forceinline__ void load4u(const float* p, int nv, float (&o)[4]) {
  if (nv >= 4) {
    f32x4 v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4u*>(p));
    else v = *reinterpret_cast<const f32x4u*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (k < nv) ? p[k] : 0.f;
  }
}
template <bool NT = false>
__device__ __forceinline__ void store4u(float* p, int nv, const float (&v)[4]) {
  if (nv >= 4) {
    const f32x4 w = {v[0], v[1], v[2], v[3]};
    if constexpr (NT) __builtin_nontemporal_store(w, reinterpret_cast<f32x4u*>(p));
    else *reinterpret_cast<f32x4u*>(p) = w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv) p[k] = v[k];
  }
}

// the same with a window: elements [klo, khi) of the four are inside the array (a quad that straddles its start or end)
template <bool NT = false>
__device__ __forceinline__ void load4w(const float* p, int klo, int khi, float (&o)[4]) {
  if (klo <= 0) { load4u<NT>(p, khi, o); return; }
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (k >= klo && k < khi) ? p[k] : 0.f;
}
template <bool NT = false>
__device__ __forceinline__ void store4w(float* p, int klo, int khi, const float (&v)[4]) {
  if (klo <= 0) { store4u<NT>(p, khi, v); return; }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k >= klo && k < khi) p[k] = v[k];
}

// raw (undecoded) vectors: issue the load early, decode at the point of use so
// the compiler does not have to wait for the data right after the load
struct RawF4 { f32x4 v; };
struct RawF8 { f32x4 a, b; };
struct RawB4 { uint2 v; };
struct RawB8 { u32x4 v; };
template <typename T> struct Raw;
template <> struct Raw<float> { using R4 = RawF4; using R8 = RawF8; };
template <> struct Raw<bf16_t> { using R4 = RawB4; using R8 = RawB8; };
__device__ __forceinline__ RawF4 load_raw4(const float* p) { return {*reinterpret_cast<const f32x4*>(p)}; }
__device__ __forceinline__ RawB4 load_raw4(const bf16_t* p) { return {*reinterpret_cast<const uint2*>(p)}; }
__device__ __forceinline__ RawF8 load_raw8(const float* p) {
  return {*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
}
__device__ __forceinline__ RawB8 load_raw8(const bf16_t* p) { return {*reinterpret_cast<const u32x4*>(p)}; }
__device__ __forceinline__ RawF4 pack_raw4(const float*, float a, float b, float c, float d) { return {f32x4{a, b, c, d}}; }
__device__ __forceinline__ RawB4 pack_raw4(const bf16_t*, float a, float b, float c, float d) {
  return {uint2{pack_bf16x2(a, b), pack_bf16x2(c, d)}};   // same rounding as store4
}
__device__ __forceinline__ void unpack(const RawF4& r, float (&o)[4]) {
  o[0] = r.v[0]; o[1] = r.v[1]; o[2] = r.v[2]; o[3] = r.v[3];
}
__device__ __forceinline__ void unpack(const RawB4& r, float (&o)[4]) {
  o[0] = __builtin_bit_cast(float, r.v.x << 16); o[1] = __builtin_bit_cast(float, r.v.x & 0xffff0000u);
  o[2] = __builtin_bit_cast(float, r.v.y << 16); o[3] = __builtin_bit_cast(float, r.v.y & 0xffff0000u);
}
__device__ __forceinline__ void unpack(const RawF8& r, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = r.a[i]; o[4 + i] = r.b[i]; }
}
__device__ __forceinline__ void unpack(const RawB8& r, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __builtin_bit_cast(float, r.v[i] << 16);
    o[2 * i + 1] = __builtin_bit_cast(float, r.v[i] & 0xffff0000u);
  }
}

// ---------------------------------------------------------------------------
// scalar math
// ---------------------------------------------------------------------------
__device__ __forceinline__ float softplusf(float x) {  // jax.nn.softplus = logaddexp(x, 0)
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) {
  return x >= 0.f ? 1.f / (1.f + expf(-x)) : expf(x) / (1.f + expf(x));
}

// act(a) = alpha * elu(a) + (1 - alpha) * tanh(a)            models.py:258-262
// Everything the backward needs in one evaluation:
//   h     = act(a)
//   dact  = alpha * (a > 0 ? 1 : exp(a)) + (1 - alpha) * (1 - tanh(a)^2)
//   ediff = elu(a) - tanh(a)            (d act / d alpha)
// digamma for x > 0: shift to x >= 6 by the recurrence, then the asymptotic series
// (|err| ~ 1e-7 relative in fp32 for the shifted argument).
__device__ __forceinline__ float digammaf(float x) {
  float acc = 0.f;
#pragma unroll 1
  while (x < 6.f) { acc -= 1.0f / x; x += 1.0f; }
  const float r = 1.0f / x, r2 = r * r;
  const float tail = r2 * (1.f / 12.f - r2 * (1.f / 120.f - r2 * (1.f / 252.f - r2 * (1.f / 240.f))));
  return acc + logf(x) - 0.5f * r - tail;
}

// log(1 + e^x) / log sigmoid without overflow
__device__ __forceinline__ float log_sigmoidf(float x) { return -softplusf(-x); }


// ---------------------------------------------------------------------------
// likelihood of one row and its derivatives (models.py:157-191).
//   ll      log p(y | out, theta)
//   dout    d loss / d out            with loss = -c * ll
//   d_par   d loss / d theta_par      par = log_noise_scale (NORMAL) or shape (NB / ZINB)
//   d_infl  d loss / d inflated_loc_probs (ZINB)
// ---------------------------------------------------------------------------
struct RowLoss {
  float ll, dout, d_par, d_infl;
};

__device__ __forceinline__ RowLoss row_loss_eval(int obs, const float* __restrict__ th, int off_lns,
                                                 int off_shape, int off_infl, float yv, float out,
                                                 float c) {
  RowLoss o;
  o.d_infl = 0.f;
  if (obs == 0 /* BNF_OBS_NORMAL */) {
    // sigma = 0.01 + exp(lns)
    const float lns = th[off_lns];
    const float sigma = 0.01f + expf(lns);
    const float res = yv - out;
    const float z = res / sigma;
    o.ll = -0.5f * z * z - logf(sigma) - 0.918938533204672742f;
    o.dout = -c * res / (sigma * sigma);
    o.d_par = -c * (res * res / (sigma * sigma * sigma) - 1.0f / sigma) * expf(lns);
    return o;
  }
  // NB / ZINB: mean = softplus(out), shape = softplus(theta_shape), total_count = 1/shape,
  // logits = -log shape - log mean;  TFP 0.24 log_prob:
  //   tc logsig(-logits) + y logsig(logits) + lgamma(tc+y) - lgamma(1+y) - lgamma(tc)
  const float ths = th[off_shape];
  const float shape = softplusf(ths);
  const float tc = 1.0f / shape;
  const float mean = softplusf(out);
  const float logits = -logf(shape) - logf(mean);
  const float sg = sigmoidf(logits);
  const float lsn = log_sigmoidf(-logits);
  float lp = tc * lsn + yv * log_sigmoidf(logits) + lgammaf(tc + yv) - lgammaf(1.0f + yv) - lgammaf(tc);
  float dl_dlogits = yv * (1.0f - sg) - tc * sg;
  float dl_dtc = lsn + digammaf(tc + yv) - digammaf(tc);
  if (obs == 2 /* BNF_OBS_ZINB */) {
    // Mixture(cat = [1 - pi, pi], [NB, delta_0])
    const float thp = th[off_infl];
    const float pi = sigmoidf(thp);
    float dlp_dpi;
    if (yv == 0.f) {
      const float p0 = expf(lp);
      const float den = (1.0f - pi) * p0 + pi;
      const float w = (1.0f - pi) * p0 / den;
      dlp_dpi = (1.0f - p0) / den;
      lp = logf(den);
      dl_dlogits *= w; dl_dtc *= w;
    } else {
      dlp_dpi = -1.0f / (1.0f - pi);
      lp += log_sigmoidf(-thp);                 // log(1 - pi)
    }
    o.d_infl = -c * dlp_dpi * pi * (1.0f - pi);
  }
  o.ll = lp;
  o.dout = -c * (-dl_dlogits / mean) * sigmoidf(out);
  o.d_par = -c * (-dl_dlogits / shape - dl_dtc / (shape * shape)) * sigmoidf(ths);
  return o;
}

// min(e, 1) for e >= 0 and max(t, 0) as v_med3_f32: one instruction each, no canonicalising self-max in
// front (which __builtin_fmaxf gets in IEEE mode when the compiler cannot prove its operand quiet, e.g.
// an accumulator made opaque) -- and, unlike an inline-asm v_min / v_max, VISIBLE to hipcc's hazard
// recogniser: an asm statement gets no wait states (cdna_hip_programming.md 5.7), and a VALU instruction
// that reads the result of the v_exp_f32 / v_rcp_f32 issued just before it needs one (measured: the asm
// form of these two gave wrong activation gradients in the kernels where hipcc scheduled it right behind
// the v_exp_f32, and right ones where other instructions happened to sit in between).
__device__ __forceinline__ float min_with_one(float e) { return __builtin_amdgcn_fmed3f(e, 1.0f, -1.0f); }
__device__ __forceinline__ float max_with_zero(float t) { return __builtin_amdgcn_fmed3f(t, 0.0f, 3.0e38f); }

struct ActOut {
  float h, dact, ediff;
};

#ifdef BNF_EXP_LIBM
#define BNF_EXP2(x) exp2f(x)
#else
#define BNF_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#ifdef BNF_RCP_DIV
#define BNF_RCP(x) (1.0f / (x))
#else
#define BNF_RCP(x) __builtin_amdgcn_rcpf(x)
#endif

// ---------------------------------------------------------------------------
// Lean form of the same activation for the VALU-bound epilogues of the bf16 kernels (the epilogues
// cost one issue slot per instruction, two per transcendental: the instruction count IS the time).
// Everything is expressed in t = a log2(e), the pre-activation already scaled for v_exp_f32 -- the
// caller folds log2(e) into the scale / bias it applies to the accumulator anyway -- and in
//     e = 2^t = e^a,   r = 1 / (1 + e^2)   (tanh a = 1 - 2 r,  1 - tanh^2 a = 4 (r - r^2)),
//     dl = min(e, 1) = d elu / da,   mxt = max(t, 0),   s = ln2 mxt + dl = elu(a) + 1:
//     act(a)            = c0 + c1 r + alpha s          c0 = 1 - 2 alpha,  c1 = -2 (1 - alpha)
//     act'(a)           = alpha dl + c2 (r - r^2)      c2 = 4 (1 - alpha)
//     elu(a) - tanh(a)  = s + 2 r - 2
// No |a|, no sign transfer, no selects; callers fold their own row / column factors into the
// constants and take constant terms out of the sums (the "- 2" above, "c0" in a row dot).
// e^2 overflows to inf for a > 44: r = 0, tanh = 1, act' = alpha -- the right limits; a -> -inf gives
// e = 0, r = 1.  Same accuracy class as act_parts2 (1 - 2 r near 0 is absolute-accurate to ~2e-7).
// ---------------------------------------------------------------------------
constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
struct ActCore2 {
  f32x2 r, dl, mxt;
};
__device__ __forceinline__ ActCore2 act_core2(f32x2 t) {
  ActCore2 c;
  const f32x2 e = {BNF_EXP2(t.x), BNF_EXP2(t.y)};
  const f32x2 den = e * e + 1.f;
  f32x2 r = {BNF_RCP(den.x), BNF_RCP(den.y)};
  asm volatile("" : "+v"(r));   // see act_eval: keeps hipcc 7.2 from mis-optimising the consumers of both builtins
  c.r = r;
  c.dl = f32x2{min_with_one(e.x), min_with_one(e.y)};
  c.mxt = f32x2{max_with_zero(t.x), max_with_zero(t.y)};
  return c;
}
// act_core2 with min(e, 1) of BOTH elements from one packed multiply carrying the clamp modifier (e >= 0, so
// clamp(e * 1) to [0, 1] = min(e, 1)): hipcc emits one v_max_f32 ... clamp per element for the builtin form, and there is
// no packed min.  The asm takes `den` as a dummy input: it is then scheduled after den = e e + 1 was issued, i.e. at least
// one instruction behind the v_exp_f32 that produced e (an asm statement gets no trans-use wait state from hipcc).
__device__ __forceinline__ ActCore2 act_core2_pkclamp(f32x2 t) {
  ActCore2 c;
  const f32x2 e = {BNF_EXP2(t.x), BNF_EXP2(t.y)};
  f32x2 den = e * e + 1.f;
  f32x2 dl;
  asm("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0] clamp" : "=v"(dl) : "v"(e), "v"(den));
  f32x2 r = {BNF_RCP(den.x), BNF_RCP(den.y)};
  asm volatile("" : "+v"(r));
  c.r = r;
  c.dl = dl;
  c.mxt = f32x2{max_with_zero(t.x), max_with_zero(t.y)};
  return c;
}
// Forward-only core: r and s = elu(a) + 1 with ONE median per element.  For a > 0: a + 1 <= e^a and a + 1 > 1, for
// a <= 0: a + 1 <= e^a <= 1 -- so elu(a) + 1 = median(a + 1, e^a, 1) on both sides (the backward epilogues also need
// dl = min(e, 1) on its own and keep act_core2).  Same values as ln2 max(t, 0) + min(e, 1) up to the 1-ulp rounding of
// v_exp_f32 next to a = 0; the right limits at +-inf (e = inf: a + 1; e = 0: 0).  Two VALU slots per element pair less.
struct ActFwd2 {
  f32x2 r, s;
};
__device__ __forceinline__ ActFwd2 act_fwd_core2(f32x2 t) {
  ActFwd2 c;
  const f32x2 e = {BNF_EXP2(t.x), BNF_EXP2(t.y)};
  const f32x2 den = e * e + 1.f;
  f32x2 r = {BNF_RCP(den.x), BNF_RCP(den.y)};
  asm volatile("" : "+v"(r));   // see act_core2
  c.r = r;
  const f32x2 a1 = kLn2 * t + 1.f;
  c.s = f32x2{__builtin_amdgcn_fmed3f(a1.x, e.x, 1.0f), __builtin_amdgcn_fmed3f(a1.y, e.y, 1.0f)};
  return c;
}
struct ActConst {
  float alpha, c0, c1, c2;
};
__device__ __forceinline__ ActConst act_const(float alpha) {
  return ActConst{alpha, 1.f - 2.f * alpha, -2.f * (1.f - alpha), 4.f * (1.f - alpha)};
}
// Four bf16 values (two packed dwords) -> four fp8 bytes (one dword) by two v_cvt_scalef32_pk_{fp8,bf8}_bf16: the
// gfx950 conversions take a bf16 PAIR straight from its packed dword, DIVIDE by `scale` (a power of two; measured:
// scripts/probes/fp8_probe.hip) and write one half of the destination dword each.  Round to nearest even; with
// MODE.FP16_OVFL set (the callers do: s_setreg) overflow clamps to the largest finite value (448 / 57344) instead of
// giving NaN / inf.  fp8 = OCP e4m3 (activations), bf8 = OCP e5m2 (backward signals).
typedef short v2s_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16x4_to_fp8(uint32_t lo, uint32_t hi, float scale) {
  v2s_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(bf16x2_t, lo), scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, __builtin_bit_cast(bf16x2_t, hi), scale, true);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t bf16x4_to_bf8(uint32_t lo, uint32_t hi, float scale) {
  v2s_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(r, __builtin_bit_cast(bf16x2_t, lo), scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(r, __builtin_bit_cast(bf16x2_t, hi), scale, true);
  return __builtin_bit_cast(uint32_t, r);
}
// two bf16 values rounded by ONE v_cvt_pk_bf16_f32: hipcc splits a bf16x2 whose halves are stored
// separately into two conversions (each with a zero partner); made opaque as ONE dword, the pair is
// converted together and the halves leave as ds_write_b16 / ds_write_b16_d16_hi.  (The empty asm holds no
// instruction, so there is no wait state hipcc could miss.)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r = pack_bf16x2(lo, hi);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ void store_pair_pk(float* p0, float* p1, float a, float b) { *p0 = a; *p1 = b; }
__device__ __forceinline__ void store_pair_pk(bf16_t* p0, bf16_t* p1, float a, float b) {
  const uint32_t pk = cvt_pk_bf16(a, b);
  p0->bits = (uint16_t)(pk & 0xffffu);
  p1->bits = (uint16_t)(pk >> 16);
}

// four bf16 values (consecutive elements of one row) rounded by two v_cvt_pk_bf16_f32 and stored by ONE 8-byte write
__device__ __forceinline__ void store_quad_pk(bf16_t* p, float a, float b, float c, float d) {
  typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2v*>(p) = u32x2v{cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)};
}
__device__ __forceinline__ void store_quad_pk(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }

template <bool FAST>
__device__ __forceinline__ ActOut act_eval(float a, float alpha) {
  ActOut o;
  float th, el, dexp;
  if constexpr (FAST) {
    // one v_exp_f32 + one v_rcp_f32 (bf16 pipeline only): the lean formulas above, one element
    const float t = a * kLog2e;
    const float e = BNF_EXP2(t);
    float r = BNF_RCP(__builtin_fmaf(e, e, 1.f));
    // Keep the reciprocal opaque: with both raw transcendental builtins visible, hipcc 7.2
    // mis-optimises the column reductions that consume this value inside the fused kernel
    // (wrong d bias / d k_o; every variant that hides either builtin is correct -- measured).
    asm volatile("" : "+v"(r));
    const float dl = min_with_one(e);
    const float s1 = __builtin_fmaf(kLn2, max_with_zero(t), dl);          // elu + 1
    o.ediff = __builtin_fmaf(2.f, r, s1) - 2.f;
    o.h = __builtin_fmaf(-2.f * (1.f - alpha), r, __builtin_fmaf(alpha, s1, 1.f - 2.f * alpha));
    o.dact = __builtin_fmaf(4.f * (1.f - alpha), __builtin_fmaf(-r, r, r), alpha * dl);
    return o;
  } else {
    th = tanhf(a);
    el = a > 0.f ? a : expm1f(a);
    dexp = a > 0.f ? 1.f : expf(a);
  }
  o.h = th + alpha * (el - th);
  o.dact = alpha * dexp + (1.f - alpha) * (1.f - th * th);
  o.ediff = el - th;
  return o;
}

template <bool FAST>
__device__ __forceinline__ float act_fwd(float a, float alpha) {
  float th, el;
  if constexpr (FAST) {
    const float t = a * kLog2e;
    const float e = BNF_EXP2(t);
    float r = BNF_RCP(__builtin_fmaf(e, e, 1.f));
    asm volatile("" : "+v"(r));
    const float s1 = __builtin_fmaf(kLn2, max_with_zero(t), min_with_one(e));
    return __builtin_fmaf(-2.f * (1.f - alpha), r, __builtin_fmaf(alpha, s1, 1.f - 2.f * alpha));
  } else {
    th = tanhf(a);
    el = a > 0.f ? a : expm1f(a);
  }
  return th + alpha * (el - th);
}

// Two-element versions for the epilogues: identical formulas, written on float2 so that the
// adds / multiplies / fmas issue as packed v_pk_*_f32 (two elements per VALU slot; the
// epilogues are VALU-bound).  exp2 / rcp / sign transfer / selects stay per element.
struct ActOut2 {
  f32x2 h, dact, ediff;
};
__device__ __forceinline__ f32x2 act_fwd2(f32x2 a, float alpha) {
  const ActCore2 c = act_core2(a * kLog2e);
  const f32x2 s = kLn2 * c.mxt + c.dl;
  return (-2.f * (1.f - alpha)) * c.r + (alpha * s + (1.f - 2.f * alpha));
}
__device__ __forceinline__ ActOut2 act_eval2(f32x2 a, float alpha) {
  ActOut2 o;
  const ActCore2 c = act_core2(a * kLog2e);
  const f32x2 s = kLn2 * c.mxt + c.dl;
  o.ediff = (2.f * c.r + s) - 2.f;
  o.h = (-2.f * (1.f - alpha)) * c.r + (alpha * s + (1.f - 2.f * alpha));
  o.dact = (4.f * (1.f - alpha)) * (c.r - c.r * c.r) + alpha * c.dl;
  return o;
}

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------
// Sum over the 64 lanes without LDS traffic: four DPP adds inside each row of 16 lanes (quad swaps, half-row and row
// mirrors: every lane of a row ends up with the row total), then the four row totals through v_readlane -- ~12 issue
// slots and no memory latency, against six dependent ds_bpermute round trips (~100+ cycles each) of the shuffle form.
// Used where a reduction sits on a phase's critical path (the panel kernel's row phase and epilogue tails).
// The result is wave-uniform.  Summation order differs from wave_sum's (pairwise either way).
template <int CTRL>
__device__ __forceinline__ float dpp_src(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_src<0xB1>(v);     // quad_perm [1, 0, 3, 2]
  v += dpp_src<0x4E>(v);     // quad_perm [2, 3, 0, 1]
  v += dpp_src<0x141>(v);    // row_half_mirror
  v += dpp_src<0x140>(v);    // row_mirror
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Sum over each 32-lane half of the wave with DPP adds (no LDS traffic): the
// result is valid in lanes 16-31 (sum of lanes 0-31) and 48-63 (lanes 32-63).
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1, 3
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: no state, any element
// of any stream is computable independently -> results do not depend on how
// members are sharded over GPUs.
// ---------------------------------------------------------------------------
#ifndef BNF_PHILOX_ROUNDS
#define BNF_PHILOX_ROUNDS 10   // the standard count; other values only to price the generator (profiles/r04_panel_ab.md r04p)
#endif
struct Philox {
  uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox philox4x32(uint32_t c0, uint32_t c1, uint32_t c2,
                                                      uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < BNF_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// random streams (counter word 3, low byte)
enum : uint32_t {
  STREAM_INIT = 1,     // TruncatedNormal initial kernels
  STREAM_SHUFFLE = 2,  // per-member per-epoch row shuffle (MAP)
  STREAM_VI_EPS = 3,   // reparameterisation noise
  STREAM_VI_BATCH = 4, // shared random batch of a VI step
  STREAM_VI_DRAW = 5   // posterior draws after fitting
};

__device__ __forceinline__ float u01_open(uint32_t x) {  // (0,1), 24 bits
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// TruncatedNormal(0, 1, low=-2, high=2) by inverse CDF.
__device__ __forceinline__ float trunc_normal_m2p2(uint32_t bits) {
  const float lo = 0.02275013194817921f;            // Phi(-2)
  const float span = 0.9544997361036416f;           // Phi(2) - Phi(-2)
  float x = normcdfinvf(lo + span * u01_open(bits));
  return fminf(2.f, fmaxf(-2.f, x));
}

// standard normal from two 32-bit words (Box-Muller, cosine branch)
__device__ __forceinline__ float std_normal(uint32_t a, uint32_t b) {
  const float u1 = u01_open(a), u2 = u01_open(b);
  return sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}

// reparameterisation noise eps[member_global][sample][p] of VI step `step`.  One Philox call yields the four normals
// of ONE sample at four consecutive parameters (two Box-Muller pairs, both branches): S calls per four parameters,
// whatever S is (calls keyed by groups of four SAMPLES spent 8 per four parameters at S = 5, and Philox's 40 quarter-rate
// integer multiplies were what k_vi_sample was bound by).  Quads are cut at p = 3 (mod 4) -- quad = (p + 1) / 4 --
// because that is where every hidden Dense kernel starts in the parameter layout (three leading scalars, then leaves
// whose sizes are multiples of the width: spec.py), so a 64 x 64 tile of a kernel is made of whole quads
// (k_vi_sample_pack).  The hardware log2 / sqrt / sin / cos are plenty for noise (the tests read eps back from the
// device: bnf_debug_vi_eps).
constexpr int kEpsQuadPhase = 1;
struct Normal4 {
  float v[4];
};
__device__ __forceinline__ void box_muller_pair(uint32_t a, uint32_t b, float* n0, float* n1) {
  const float u1 = u01_open(a), u2 = u01_open(b);
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // -2 ln u1
  *n0 = rad * __builtin_amdgcn_cosf(u2);    // v_cos / v_sin take revolutions
  *n1 = rad * __builtin_amdgcn_sinf(u2);
}
// the normals of parameters 4 quad - 1 .. 4 quad + 2.  Counter word 3 = stream (bits 0..7) | sample (bits 8..23: up to
// 65,535 -- bnf_vi_posterior_draws' limit) | bits 32..39 of the step (bits 24..31): the fields do not overlap (round 4
// put the step's high bits at bit 20, which met the sample field from sample 4096 on; unchanged values for every
// sample < 4096 at step < 2^32, i.e. for every stream drawn so far).  ABI 5 changelog: round 4 re-keyed this stream per
// (sample, quad) -- 'philox' VI results for a given seed differ from rounds 1-3.
__device__ __forceinline__ Normal4 vi_eps_quad(uint64_t seed, uint32_t member_global, uint32_t sample,
                                               uint32_t quad, uint64_t step, uint32_t stream) {
  const Philox r = philox4x32(quad, member_global, (uint32_t)step,
                              (stream & 0xffu) | ((sample & 0xffffu) << 8) | ((uint32_t)(step >> 32) << 24), (uint32_t)seed,
                              (uint32_t)(seed >> 32));
  Normal4 n;
  box_muller_pair(r.v[0], r.v[1], &n.v[0], &n.v[1]);
  box_muller_pair(r.v[2], r.v[3], &n.v[2], &n.v[3]);
  return n;
}
__device__ __forceinline__ float vi_eps(uint64_t seed, uint32_t member_global, uint32_t sample,
                                        uint32_t p, uint64_t step, uint32_t stream) {
  const uint32_t u = p + (uint32_t)kEpsQuadPhase;
  const Normal4 n = vi_eps_quad(seed, member_global, sample, u >> 2, step, stream);
  return n.v[u & 3u];
}

// ---------------------------------------------------------------------------
// The reference's OWN VI noise (optional; full-batch fits started by BayesianNeuralFieldVI.fit):
// jax.random.normal(key, (E, *leaf shape)) per leaf, with the per-(step, sample, leaf) keys of
// tfp.vi.fit_surrogate_posterior_stateless's seed chain computed on the host once per fit
// (bayesnf_amd/jaxseed.py: vi_noise_keys).  Restated here: threefry2x32 over iota(n) split in
// halves (element i < ceil(n/2) is word 0 of the pair (i, i + h), element i >= h word 1 of (i - h, i);
// odd n is padded with one zero count), uniform on [nextafter(-1, 0), 1), sqrt(2) erfinv (the
// single-precision polynomial of Giles 2010 that XLA uses).
// ---------------------------------------------------------------------------
struct JaxNoise {
  const uint32_t* keys;      // (rows, S, n_leaves, 2) ; null = not in use
  const int32_t* leaf_off;   // (n_leaves + 1) offsets of the leaves in the packed parameter vector
  const uint8_t* leaf_id;    // (P) leaf index of every packed parameter
  int32_t n_leaves, S, members;
  int64_t row;               // key-table row of this launch (optimisation step or posterior draw)
};
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                                             uint32_t* y0, uint32_t* y1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += k0; x1 += k1;
#define BNF_TF(r) x0 += x1; x1 = ((x1 << (r)) | (x1 >> (32 - (r)))) ^ x0;
  BNF_TF(13) BNF_TF(15) BNF_TF(26) BNF_TF(6)  x0 += k1; x1 += k2 + 1u;
  BNF_TF(17) BNF_TF(29) BNF_TF(16) BNF_TF(24) x0 += k2; x1 += k0 + 2u;
  BNF_TF(13) BNF_TF(15) BNF_TF(26) BNF_TF(6)  x0 += k0; x1 += k1 + 3u;
  BNF_TF(17) BNF_TF(29) BNF_TF(16) BNF_TF(24) x0 += k1; x1 += k2 + 4u;
  BNF_TF(13) BNF_TF(15) BNF_TF(26) BNF_TF(6)  x0 += k2; x1 += k0 + 5u;
#undef BNF_TF
  *y0 = x0; *y1 = x1;
}
__device__ __forceinline__ float erfinv_f32(float x) {
  float w = -log1pf(-x * x), p;
  if (w < 5.0f) {
    w -= 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f); p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f); p = fmaf(p, w, -0.00125372503f); p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f); p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f); p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f); p = fmaf(p, w, -0.0076224613f); p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f); p = fmaf(p, w, 2.83297682f);
  }
  return p * x;
}
// standard normal of (member e, sample s, packed parameter p) in the reference's stream
__device__ __forceinline__ float jax_normal(const JaxNoise& jn, int e, int s, int p) {
  const int lf = jn.leaf_id[p];
  const int off = jn.leaf_off[lf], size = jn.leaf_off[lf + 1] - off;
  const uint32_t n = (uint32_t)jn.members * (uint32_t)size, idx = (uint32_t)e * (uint32_t)size + (uint32_t)(p - off);
  const uint32_t h = (n + 1u) >> 1;
  const uint32_t* k = jn.keys + (((jn.row * jn.S + s) * jn.n_leaves + lf) << 1);
  uint32_t y0, y1;
  if (idx < h) threefry2x32(k[0], k[1], idx, idx + h < n ? idx + h : 0u, &y0, &y1);
  else { threefry2x32(k[0], k[1], idx - h, idx, &y0, &y1); y0 = y1; }
  const float f = __builtin_bit_cast(float, (y0 >> 9) | 0x3F800000u) - 1.0f;
  const float lo = -0.99999994f;                       // nextafter(-1, 0)
  const float u = fmaxf(lo, f * (1.0f - lo) + lo);
  return 1.41421356237309504880f * erfinv_f32(u);
}

// ---------------------------------------------------------------------------
// Keyed Feistel bijection on [0, n): a pseudo-random permutation evaluated
// per element (no index array in HBM).  Replaces jax.random.permutation in
// permute_dataset (inference.py:35-39) and in ensemble_vi (inference.py:706).
// ---------------------------------------------------------------------------
struct FeistelKey {
  uint32_t k[6];
  uint32_t half_bits;  // domain = 2^(2*half_bits) >= n
  uint64_t n;
};

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {  // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ FeistelKey feistel_key(uint64_t seed, uint32_t a, uint64_t b,
                                                           uint32_t stream, uint64_t n) {
  FeistelKey fk;
  Philox r0 = philox4x32(0u, a, (uint32_t)b, (stream & 0xffu) | ((uint32_t)(b >> 32) << 8),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
  Philox r1 = philox4x32(1u, a, (uint32_t)b, (stream & 0xffu) | ((uint32_t)(b >> 32) << 8),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
  fk.k[0] = r0.v[0]; fk.k[1] = r0.v[1]; fk.k[2] = r0.v[2]; fk.k[3] = r0.v[3];
  fk.k[4] = r1.v[0]; fk.k[5] = r1.v[1];
  uint32_t bits = 2;
  while ((1ull << bits) < n) ++bits;
  fk.half_bits = (bits + 1) >> 1;
  fk.n = n;
  return fk;
}

__host__ __device__ __forceinline__ uint64_t feistel_perm(const FeistelKey& fk, uint64_t i) {
  const uint32_t hb = fk.half_bits;
  const uint32_t mask = (hb >= 32) ? 0xffffffffu : ((1u << hb) - 1u);
  uint64_t x = i;
  do {
    uint32_t L = (uint32_t)(x >> hb) & mask, R = (uint32_t)x & mask;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t f = mix32(R ^ fk.k[r]) & mask;
      const uint32_t nl = R;
      R = L ^ f;
      L = nl;
    }
    x = ((uint64_t)L << hb) | R;
  } while (x >= fk.n);
  return x;
}

}  // namespace bnf
