// tr-read with the gemm_tn address pattern: LDS tile [64 rows][128 cols] bf16, row pitch 256 B,
// 64-byte segments XOR (row & 3).  Pass 0 stores value=row, pass 1 stores value=col.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_v4;
__global__ void probe(float* out, int pass, int use_builtin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int idx = threadIdx.x; idx < 64 * 128; idx += 64) {
    const int row = idx / 128, col = idx % 128;
    const float f = pass == 0 ? (float)row : (float)col;
    const int b = col * 2;
    const int phys = row * 256 + ((b & ~63) ^ ((row & 3) << 6)) + (b & 63);
    *(uint16_t*)(smem + phys) = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
  }
  __syncthreads();
  const int lane = threadIdx.x, kg = lane >> 5, p = lane & 15, half = (lane >> 4) & 1;
  const int prow = p >> 2, pcol = half * 16 + (p & 3) * 4;
  for (int t = 0; t < 2; ++t) {
    const int row = 16 + kg * 8 + t * 4 + prow;           // ks = 1
    const int rsw = (row & 3) << 6;
    const int ba = (64 + 32 + pcol) * 2;                   // wr = 1, i = 1
    char* addr = smem + row * 256 + ((ba & ~63) ^ rsw) + (ba & 63);
    s16x4 v;
    if (use_builtin) {
      v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)addr);
    } else {
      uint2 r; uint32_t a32 = (uint32_t)(uintptr_t)addr;
      asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a32) : "memory");
      v[0] = (short)(r.x & 0xffff); v[1] = (short)(r.x >> 16); v[2] = (short)(r.y & 0xffff); v[3] = (short)(r.y >> 16);
    }
    for (int q = 0; q < 4; ++q)
      out[(lane * 2 + t) * 4 + q] = __builtin_bit_cast(float, ((uint32_t)(uint16_t)v[q]) << 16);
  }
}
int main() {
  float* d; hipMalloc(&d, 64 * 8 * sizeof(float));
  for (int ub = 0; ub < 2; ++ub) for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 64 * 256, 0, d, pass, ub);
    std::vector<float> h(512);
    hipMemcpy(h.data(), d, 512 * sizeof(float), hipMemcpyDeviceToHost);
    printf("builtin=%d pass %s (expect lane l: %s)\n", ub, pass ? "col" : "row",
           pass ? "col = 96 + (l&31) for all 8" : "rows 16+kg*8+0..7");
    for (int l = 0; l < 64; l += (pass ? 1 : 5)) {
      printf("lane %2d:", l);
      for (int k = 0; k < 8; ++k) printf(" %4.0f", h[l * 8 + k]);
      printf("\n");
    }
  }
  return 0;
}
