// Empirical semantics of ds_read_b64_tr_b16 on gfx950: LDS holds bf16(value = element index),
// every lane reads with address = lane * 8 bytes (+ base); prints the 4 values each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(float* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) {
    float f = (float)i;  // exact in bf16 for i < 256; use i % 256 pattern + keep index separately
    uint32_t u = __builtin_bit_cast(uint32_t, (float)(i % 256));
    lds[i] = (uint16_t)(u >> 16);
    (void)f;
  }
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t addr = (uint32_t)(uintptr_t)lds + lane * stride_bytes;
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  uint32_t w[2] = {r.x, r.y};
  for (int j = 0; j < 4; ++j) {
    uint16_t b = (j & 1) ? (uint16_t)(w[j >> 1] >> 16) : (uint16_t)(w[j >> 1] & 0xffff);
    out[lane * 4 + j] = __builtin_bit_cast(float, ((uint32_t)b) << 16);
  }
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
  for (int stride : {8, 32}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 256 * sizeof(float), hipMemcpyDeviceToHost);
    printf("stride %d bytes per lane:\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
