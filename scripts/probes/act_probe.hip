// What does one activation evaluation cost on a gfx950 SIMD?  Register-only loops over the epilogue math of
// bnf_panel.h (act_eval2: exp2 + rcp + ~13 arithmetic per element), with variants that drop the
// transcendentals or the arithmetic.  Prints cycles per element per wave (2 and 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bayesnf_amd/csrc/bnf_device.h"
using namespace bnf;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float alpha, int iters) {
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{0.01f * (threadIdx.x + i), -0.02f * (threadIdx.x + 3 * i)};
  f32x2 s = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {            // full backward evaluation + the epilogue's products / sums
        const ActOut2 o = act_eval2(v[i], alpha);
        s += o.h * o.dact + o.ediff;
        v[i] = v[i] * 0.999f + o.dact * 0.001f;
      } else if (MODE == 1) {     // transcendentals only
        const f32x2 e = {BNF_EXP2(-fabsf(v[i].x)), BNF_EXP2(-fabsf(v[i].y))};
        const f32x2 r = {BNF_RCP(e.x + 1.f), BNF_RCP(e.y + 1.f)};
        s += r;
        v[i] = v[i] * 0.999f + r * 0.001f;
      } else {                    // ~15 dependent-free packed f32 FMAs per pair (arithmetic only)
        f32x2 t = v[i];
#pragma unroll
        for (int k2 = 0; k2 < 7; ++k2) t = t * 1.0001f + 0.25f;
        s += t;
        v[i] = v[i] * 0.999f + t * 0.001f;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 4 * 512 * 4);
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 0.3f, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 0.3f, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: blocks_per_cu * 2 waves, each iters * 8 pairs = 16 elements per iteration
  const double elems_per_simd = (double)blocks_per_cu * 2 * iters * 16;
  printf("%-28s %d waves/SIMD: %.3f ms  -> %.2f ns per element per SIMD (= %.1f cycles at 2.1 GHz)\n", name,
         blocks_per_cu * 2, ms, ms * 1e6 / elems_per_simd, ms * 1e6 / elems_per_simd * 2.1);
  hipFree(out);
}
int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0>("act_eval2 + products", b);
    run<1>("exp2 + rcp only", b);
    run<2>("7 packed FMAs per pair", b);
  }
  return 0;
}
