// Empirical semantics on gfx950 of what the fp8 operand-storage path relies on (round 5):
//  (1) ds_read_b64_tr_b8: LDS byte i holds its own index (two passes: low / high byte), lane L reads at L * stride;
//      prints the source byte index of each of the 8 bytes every lane receives;
//  (2) v_mfma_f32_32x32x16_fp8_bf8 operand layout: lane (m = l % 32, kg = l / 32) holds k = 8 kg + byte j -- checked
//      against a host contraction;
//  (3) v_cvt_scalef32_pk_{fp8,bf8}_bf16 / v_cvt_pk_fp8_f32: direction of the scale, rounding, saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));

__global__ void tr8(uint32_t* out, int stride_bytes, int hi) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = hi ? (uint8_t)(i >> 8) : (uint8_t)(i & 0xff);
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t addr = (uint32_t)(uintptr_t)lds + lane * stride_bytes;
  uint2 r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[lane * 2] = r.x; out[lane * 2 + 1] = r.y;
}

__global__ void mfma8(const uint8_t* A, const uint8_t* B, float* D) {   // A (32 x 16) e4m3 row-major, B (16 x 32) e5m2 [k][n]
  const int l = threadIdx.x, m = l & 31, kg = l >> 5;
  uint64_t a = 0, b = 0;
  for (int j = 0; j < 8; ++j) {
    a |= (uint64_t)A[m * 16 + kg * 8 + j] << (8 * j);
    b |= (uint64_t)B[(kg * 8 + j) * 32 + m] << (8 * j);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_bf8((long)a, (long)b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * kg + r % 4) * 32 + m] = c[r];   // D[row][col = lane % 32]
}

__global__ void cvt(const float* in, const float* scales, uint32_t* out, int n, int sat) {
  const int i = threadIdx.x;
  if (i >= n) return;
  if (sat) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);     // MODE.FP16_OVFL: conversions clamp to the largest finite value
  const float x = in[2 * i], y = in[2 * i + 1], s = scales[i];
  bf16x2 p = {(__bf16)x, (__bf16)y};
  v2s old = {0, 0};
  v2s r8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, p, s, false);
  v2s r5 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(old, p, s, false);
  int q = __builtin_amdgcn_cvt_pk_fp8_f32(x, y, 0, false);
  v2s rh = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, p, s, true);
  out[4 * i] = __builtin_bit_cast(uint32_t, r8); out[4 * i + 1] = __builtin_bit_cast(uint32_t, r5);
  out[4 * i + 2] = (uint32_t)q; out[4 * i + 3] = __builtin_bit_cast(uint32_t, rh);
}

static float dec(uint8_t v, int eb, int mb, int bias) {
  const int s = v >> 7, e = (v >> mb) & ((1 << eb) - 1), m = v & ((1 << mb) - 1);
  float f = e == 0 ? ldexpf((float)m, 1 - bias - mb) : ldexpf((float)((1 << mb) + m), e - bias - mb);
  return s ? -f : f;
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 2 * 4);
  for (int stride : {8, 16, 64}) {
    std::vector<uint32_t> lo(128), hi(128);
    hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, d, stride, 0); hipMemcpy(lo.data(), d, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, d, stride, 1); hipMemcpy(hi.data(), d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b8, lane address = lane * %d: source byte index of result bytes 0..7\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) {
        const uint32_t bl = (lo[l * 2 + j / 4] >> (8 * (j % 4))) & 0xff, bh = (hi[l * 2 + j / 4] >> (8 * (j % 4))) & 0xff;
        printf(" %5u", bh * 256 + bl);
      }
      printf("\n");
    }
  }
  // (2)
  std::vector<uint8_t> A(32 * 16), B(16 * 32);
  for (int i = 0; i < 512; ++i) { A[i] = (uint8_t)(0x30 + (i * 7) % 24); B[i] = (uint8_t)(0x38 + (i * 5) % 12); }
  uint8_t *dA, *dB; float* dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024 * 4);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma8, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(1024); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
    double r = 0;
    for (int k = 0; k < 16; ++k) r += (double)dec(A[m * 16 + k], 4, 3, 7) * dec(B[k * 32 + n], 5, 2, 15);
    worst = fmax(worst, fabs(r - D[m * 32 + n]));
  }
  printf("mfma_f32_32x32x16_fp8_bf8 with lane (m = l %% 32, k = 8 (l / 32) + byte): max |D - host| = %g (D[0][0] = %g)\n", worst, D[0]);
  // (3)
  const float in[] = {1.f, 3.f, 448.f, 1000.f, 0.001f, 0.01f, 1.f, 3.f, 1.f, 3.f, 6.5f, 7.5f, -0.3f, 100000.f, 1e-6f, 60000.f};
  const float sc[] = {1.f, 1.f, 1.f, 2.f, 0.5f, 1.f, 1.f, 4.f};
  float *din, *dsc; uint32_t* dout; hipMalloc(&din, sizeof(in)); hipMalloc(&dsc, sizeof(sc)); hipMalloc(&dout, 8 * 16);
  hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice); hipMemcpy(dsc, sc, sizeof(sc), hipMemcpyHostToDevice);
  for (int sat = 0; sat < 2; ++sat) {
  hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, dsc, dout, 8, sat);
  uint32_t o[32]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("MODE.FP16_OVFL = %d\n", sat);
  for (int i = 0; i < 8; ++i) {
    printf("(%g, %g) scale %g: scalef32_pk_fp8_bf16 -> %08x = (%g, %g) | pk_bf8 -> %08x = (%g, %g) | cvt_pk_fp8_f32 -> %08x = (%g, %g) | hi-sel %08x\n",
           in[2 * i], in[2 * i + 1], sc[i], o[4 * i], dec(o[4 * i] & 0xff, 4, 3, 7), dec((o[4 * i] >> 8) & 0xff, 4, 3, 7),
           o[4 * i + 1], dec(o[4 * i + 1] & 0xff, 5, 2, 15), dec((o[4 * i + 1] >> 8) & 0xff, 5, 2, 15),
           o[4 * i + 2], dec(o[4 * i + 2] & 0xff, 4, 3, 7), dec((o[4 * i + 2] >> 8) & 0xff, 4, 3, 7), o[4 * i + 3]);
  }
  }
  return 0;
}
