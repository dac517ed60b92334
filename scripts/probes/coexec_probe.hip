// Do the matrix pipe and the VALU of ONE gfx950 SIMD run at the same time when the two instruction streams come from
// DIFFERENT waves?  (Round 6, VERDICT r05 item 1: the two-panel row-panel pipeline rests on the answer.)
//
// A 512-thread workgroup per CU: waves w and w + 4 share a SIMD.  Waves 0-3 run role A, waves 4-7 role B; every wave
// stamps s_memtime after each chunk of its stream, the host reads the stamps and reports, for the window in which BOTH
// halves were running, the cycles each role spent per instruction -- next to the same role running beside an idle
// partner.  Roles: back-to-back v_mfma_f32_32x32x16_bf16 (4 independent accumulators), v_pk_fma_f32, v_fma_f32,
// v_exp_f32, an epilogue-like mix in packed and in single-issue form, and an MFMA stream with k single-issue fillers per
// MFMA inside the SAME wave.  Optional s_setprio for either half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { R_IDLE = 0, R_MFMA, R_PKFMA, R_FMA, R_EXP, R_EPI_PK, R_EPI_1, R_MIX_FMA3, R_MIX_FMA5, R_MIX_PK2, R_MIX_EPI1, R_COUNT };
static const char* kRoleName[R_COUNT] = {"idle", "mfma", "pk_fma", "fma", "exp", "epi(packed)", "epi(single)",
                                         "mfma+3fma", "mfma+5fma", "mfma+2pk", "mfma+epi1x6"};
// instructions per chunk (what one s_memtime stamp covers), by class: {mfma, valu}
__host__ __device__ constexpr int chunk_mfma(int r) { return (r == R_MFMA || r >= R_MIX_FMA3) ? 16 : 0; }
__host__ __device__ constexpr int chunk_valu(int r) {
  return r == R_PKFMA || r == R_FMA || r == R_EXP ? 64 : r == R_EPI_PK ? 4 * 11 : r == R_EPI_1 ? 4 * 18
       : r == R_MIX_FMA3 ? 48 : r == R_MIX_FMA5 ? 80 : r == R_MIX_PK2 ? 32 : r == R_MIX_EPI1 ? 96 : 0;
}

constexpr int kChunks = 256;

#define V8(op)  op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

template <int R>
__device__ __forceinline__ void run_role(unsigned long long* ts, int chunks, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed * (i + 1)); fb[i] = (__bf16)(seed * (8 - i)); }
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1); p[i] = f32x2{seed * i, -seed * (i + 2)}; }
  const float c0 = 1.0001f, c1 = 0.25f;
  const f32x2 q0 = {1.0001f, 0.9999f}, q1 = {0.25f, 0.125f};
  for (int c = 0; c < chunks; ++c) {
    if constexpr (R == R_MFMA) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
    } else if constexpr (R == R_PKFMA) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q0), "v"(q1));
        V8(OP)
#undef OP
      }
    } else if constexpr (R == R_FMA) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
        V8(OP)
#undef OP
      }
    } else if constexpr (R == R_EXP) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        V8(OP)
#undef OP
      }
    } else if constexpr (R == R_EPI_PK) {
      // per element pair: 2 exp + 2 rcp + 7 packed = 11 instructions; 4 pairs per chunk
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 e, r, t = p[i];
        asm volatile("v_exp_f32 %0, %1" : "=v"(e.x) : "v"(t.x));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e.y) : "v"(t.y));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(e), "v"(q0));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(r.x));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(r.y));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e) : "v"(q1));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(e) : "v"(t), "v"(q0));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(t) : "v"(r));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(q0), "v"(e));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(p[i + 4]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i + 4]) : "v"(t), "v"(q1));
        p[i] = t;
      }
    } else if constexpr (R == R_EPI_1) {
      // the same arithmetic element by element: 2 exp + 2 rcp + 14 single = 18 instructions per pair
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float e, r, t = h ? p[i].y : p[i].x, o = h ? p[i + 4].y : p[i + 4].x;
          asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(t));
          asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(e), "v"(c0));
          asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e) : "v"(c1));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e) : "v"(t), "v"(c0));
          asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(t) : "v"(r));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(c0), "v"(e));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(t) : "v"(o));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(o) : "v"(t), "v"(c1));
          if (h) { p[i].y = t; p[i + 4].y = o; } else { p[i].x = t; p[i + 4].x = o; }
        }
      }
    } else if constexpr (R >= R_MIX_FMA3) {
      constexpr int K = R == R_MIX_FMA3 ? 3 : R == R_MIX_FMA5 ? 5 : R == R_MIX_PK2 ? 2 : 6;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            if constexpr (R == R_MIX_PK2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i * 2 + k) & 7]) : "v"(q0), "v"(q1));
            else if constexpr (R == R_MIX_EPI1) {
              if (k == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i + u) & 7]));
              else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * 3 + k + u) & 7]) : "v"(c0), "v"(c1));
            } else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * 3 + k) & 7]) : "v"(c0), "v"(c1));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (R != R_IDLE) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) ts[c] = t;
    }
  }
  // keep everything alive
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  if (s == 12345.678f) ts[0] = 0;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* ts, int chunksA, int chunksB, int prioA, int prioB, float seed) {
  const int wave = threadIdx.x >> 6;
  unsigned long long* t = ts + ((size_t)blockIdx.x * 8 + wave) * (kChunks + 1);
  if (wave < 4) {
    if (prioA == 1) __builtin_amdgcn_s_setprio(1);
    if (prioA == 3) __builtin_amdgcn_s_setprio(3);
  } else {
    if (prioB == 1) __builtin_amdgcn_s_setprio(1);
    if (prioB == 3) __builtin_amdgcn_s_setprio(3);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) t[kChunks] = t0;
  if (wave < 4) run_role<RA>(t, chunksA, seed);
  else run_role<RB>(t, chunksB, seed);
}

struct Result { double cyc_mfma, cyc_valu; };

template <int RA, int RB>
void run(int prioA = 0, int prioB = 0) {
  const int blocks = 256;
  unsigned long long* d;
  const size_t n = (size_t)blocks * 8 * (kChunks + 1);
  hipMalloc(&d, n * 8);
  hipMemset(d, 0, n * 8);
  hipLaunchKernelGGL((k<RA, RB>), dim3(blocks), dim3(512), 0, 0, d, kChunks, kChunks, prioA, prioB, 1e-3f);
  hipLaunchKernelGGL((k<RA, RB>), dim3(blocks), dim3(512), 0, 0, d, kChunks, kChunks, prioA, prioB, 1e-3f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  hipFree(d);
  // per SIMD pair (wave w, w + 4) of every block: the window in which both ran; chunks completed inside it
  double sumA = 0, sumB = 0; int cnt = 0;
  double aloneA = 0, aloneB = 0; int cntaA = 0, cntaB = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* ta = h.data() + ((size_t)b * 8 + w) * (kChunks + 1);
      const unsigned long long* tb = h.data() + ((size_t)b * 8 + w + 4) * (kChunks + 1);
      const unsigned long long s = std::max(ta[kChunks], tb[kChunks]);
      const unsigned long long endA = RA == R_IDLE ? ~0ull : ta[kChunks - 1], endB = RB == R_IDLE ? ~0ull : tb[kChunks - 1];
      const unsigned long long e = std::min(endA, endB);
      auto rate = [&](const unsigned long long* t, unsigned long long lo, unsigned long long hi, double* cyc) {
        // first and last stamp inside [lo, hi]
        int i0 = -1, i1 = -1;
        for (int c = 0; c < kChunks; ++c)
          if (t[c] >= lo && t[c] <= hi) { if (i0 < 0) i0 = c; i1 = c; }
        if (i0 < 0 || i1 - i0 < 8) return false;
        *cyc = (double)(t[i1] - t[i0]) / (i1 - i0);
        return true;
      };
      double ca, cb;
      if (RA != R_IDLE && rate(ta, s, e, &ca)) { sumA += ca; }
      if (RB != R_IDLE && rate(tb, s, e, &cb)) { sumB += cb; }
      ++cnt;
      // the tail of the longer role: running beside a FINISHED partner
      if (RA != R_IDLE && RB != R_IDLE) {
        if (endA > endB && rate(ta, endB, endA, &ca)) { aloneA += ca; ++cntaA; }
        if (endB > endA && rate(tb, endA, endB, &cb)) { aloneB += cb; ++cntaB; }
      }
    }
  auto line = [&](const char* nm, int r, double cyc_chunk, double alone, int cnta) {
    if (r == R_IDLE) return;
    const int nm_ = chunk_mfma(r), nv = chunk_valu(r);
    printf("    %-13s %8.1f cycles per chunk", nm, cyc_chunk);
    if (nm_) printf("  = %6.1f per MFMA", cyc_chunk / nm_);
    if (nv) printf("  = %6.2f per VALU%s", cyc_chunk / nv, nm_ ? " (if the MFMAs were free)" : "");
    if (cnta) printf("   | after the partner finished: %.1f per chunk", alone / cnta);
    printf("\n");
  };
  printf("%-12s (prio %d) || %-12s (prio %d)\n", kRoleName[RA], prioA, kRoleName[RB], prioB);
  line(kRoleName[RA], RA, sumA / cnt, aloneA, cntaA);
  line(kRoleName[RB], RB, sumB / cnt, aloneB, cntaB);
}

int main() {
  printf("chunks: mfma roles 16 MFMAs; pk_fma / fma / exp 64 instructions; epi(packed) 44; epi(single) 72\n");
  run<R_MFMA, R_IDLE>();
  run<R_MFMA, R_MFMA>();
  run<R_PKFMA, R_IDLE>();
  run<R_PKFMA, R_PKFMA>();
  run<R_FMA, R_IDLE>();
  run<R_FMA, R_FMA>();
  run<R_EXP, R_IDLE>();
  run<R_EXP, R_EXP>();
  run<R_EPI_PK, R_IDLE>();
  run<R_EPI_PK, R_EPI_PK>();
  run<R_EPI_1, R_IDLE>();
  run<R_EPI_1, R_EPI_1>();
  printf("---- matrix wave beside a vector wave (same SIMD) ----\n");
  run<R_MFMA, R_PKFMA>();
  run<R_MFMA, R_PKFMA>(3, 0);
  run<R_MFMA, R_PKFMA>(0, 3);
  run<R_MFMA, R_FMA>();
  run<R_MFMA, R_FMA>(3, 0);
  run<R_MFMA, R_FMA>(0, 3);
  run<R_MFMA, R_EXP>();
  run<R_MFMA, R_EXP>(3, 0);
  run<R_MFMA, R_EPI_PK>();
  run<R_MFMA, R_EPI_PK>(3, 0);
  run<R_MFMA, R_EPI_PK>(0, 3);
  run<R_MFMA, R_EPI_1>();
  run<R_MFMA, R_EPI_1>(3, 0);
  run<R_MFMA, R_EPI_1>(0, 3);
  run<R_EPI_PK, R_MFMA>();     // (the vector wave is the older one)
  run<R_EPI_1, R_MFMA>();
  printf("---- fillers inside the matrix wave's own stream ----\n");
  run<R_MIX_FMA3, R_IDLE>();
  run<R_MIX_FMA5, R_IDLE>();
  run<R_MIX_PK2, R_IDLE>();
  run<R_MIX_EPI1, R_IDLE>();
  run<R_MIX_FMA3, R_MIX_FMA3>();
  run<R_MIX_FMA5, R_MIX_FMA5>();
  run<R_MIX_PK2, R_MIX_PK2>();
  run<R_MIX_EPI1, R_MIX_EPI1>();
  return 0;
}
