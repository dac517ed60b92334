// Does global_load_lds_dwordx4 place lane l's 16 bytes at M0base + l*16 for a 256-byte-row image?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
__global__ void probe(const uint16_t* A, uint16_t* out, int a_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* Ab = (const char*)A;
  for (int i = 0; i < 4; ++i) {
    const int q = wave * 4 + i;
    const int row = q * 4 + lane / 16;
    const int cp = lane % 16;
    const int c = cp ^ ((row & 3) << 2);
    const int src = row * a_ld * 2 + c * 16;
    __builtin_amdgcn_global_load_lds((glb_void_t*)(Ab + src), (lds_void_t*)(smem + q * 1024), 16, 0, 0);
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * 128; idx += 256) out[idx] = ((uint16_t*)smem)[idx];
}
int main() {
  const int R = 64, M = 128;
  std::vector<uint16_t> h(R * M);
  for (int r = 0; r < R; ++r) for (int c = 0; c < M; ++c) h[r * M + c] = r * 128 + c;
  uint16_t *dA, *dO; hipMalloc(&dA, R * M * 2); hipMalloc(&dO, R * M * 2);
  hipMemcpy(dA, h.data(), R * M * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 64 * 256, 0, dA, dO, M);
  std::vector<uint16_t> o(R * M);
  hipMemcpy(o.data(), dO, R * M * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < R; ++r) for (int c = 0; c < M; ++c) {
    const int b = c * 2;
    const int phys = r * 256 + ((b & ~63) ^ ((r & 3) << 6)) + (b & 63);
    const uint16_t got = o[phys / 2];
    if (got != r * 128 + c) { if (bad < 20) printf("row %d col %d: got (row %d col %d)\n", r, c, got / 128, got % 128); ++bad; }
  }
  printf("mismatches %d of %d\n", bad, R * M);
  return 0;
}
