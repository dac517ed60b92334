// panel_probe.hip -- ceilings for the row-panel design (r02): what one CU can pull from L2 / HBM
// into VGPRs or LDS, and what a 128-row x 512-column panel contraction reaches when the A operand
// sits in LDS and the B operand (fragment-major weights) streams L2 -> VGPR with a PD-deep prefetch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/panel_probe.hip -o scripts/probes/panel_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------
// stream probes: every wave reads 1 KiB per instruction, sequentially, from its slice of a region
//   MODE 0: global_load_dwordx4 -> VGPR      MODE 1: global_load_lds_dwordx4 -> LDS
//   shared = 1: all workgroups of an XCD read the SAME 512 KiB region (L2-resident weights)
//   shared = 0: every workgroup streams its own region (HBM)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const char* buf, size_t region_bytes, int shared, int iters,
                                                uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const size_t region = shared ? (size_t)(blockIdx.x & 7) : (size_t)blockIdx.x;
  const char* base = buf + region * region_bytes;
  const size_t slice = region_bytes / nw;          // bytes per wave, multiple of 1 KiB
  const char* wbase = base + (size_t)wave * slice + lane * 16;
  const size_t n_kib = slice >> 10;
  u32x4 acc = {0, 0, 0, 0};
  size_t pos = (blockIdx.x >> 3) % n_kib;          // de-phase the workgroups of an XCD
  for (int it = 0; it < iters; it += DEPTH) {
    if constexpr (MODE == 0) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        v[d] = *reinterpret_cast<const u32x4*>(wbase + (pos << 10));
        pos = pos + 1 == n_kib ? 0 : pos + 1;
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)(wbase + (pos << 10)),
                                         (lds_void_t*)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
        pos = pos + 1 == n_kib ? 0 : pos + 1;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
  }
  if constexpr (MODE == 1) acc[0] = reinterpret_cast<uint32_t*>(smem)[tid];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// ---------------------------------------------------------------------------------------------
// panel contraction probe: 8 waves, wave w owns columns [64 w, 64 w + 64) of a 128 x 512 output;
// A = the 128 x 512 bf16 panel in LDS (row pitch 1040 B), B = fragment-major weights of member e
// (Wp[nt][ks][lane][8], nt = 32-column tile, ks = 16-deep k step), straight from global memory.
constexpr int kPitch = 512 * 2 + 16;
constexpr int kKS = 32;

template <int PD, int APRE>
__device__ __forceinline__ void panel_gemm(f32x16 (&acc)[4][2], const char* P, const __bf16* wp, int wave, int lane) {
  const int frow = lane & 31, kg = lane >> 5;
  // uniform (SGPR) stream bases + one 32-bit lane offset: every load is saddr + voffset + immediate
  const char* w0 = reinterpret_cast<const char*>(wp) + (size_t)(2 * wave) * kKS * 1024;   // nt = 2 wave
  const char* w1 = w0 + (size_t)kKS * 1024;                                               // nt = 2 wave + 1
  const uint32_t loff = (uint32_t)lane * 16u;
  const char* a01 = P + frow * kPitch + kg * 16;
  const char* a23 = a01 + 64 * kPitch;
  bf16x8 fb[PD][2];
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    fb[p][0] = *reinterpret_cast<const bf16x8*>(w0 + p * 1024 + loff);
    fb[p][1] = *reinterpret_cast<const bf16x8*>(w1 + p * 1024 + loff);
  }
  auto load_a = [&](bf16x8 (&fa)[4], int ks) {
    const int ko = ks * 32;
    fa[0] = *reinterpret_cast<const bf16x8*>(a01 + ko);
    fa[1] = *reinterpret_cast<const bf16x8*>(a01 + 32 * kPitch + ko);
    fa[2] = *reinterpret_cast<const bf16x8*>(a23 + ko);
    fa[3] = *reinterpret_cast<const bf16x8*>(a23 + 32 * kPitch + ko);
  };
  bf16x8 fa[2][4];
  if (APRE) load_a(fa[0], 0);
#pragma unroll 1
  for (int ks0 = 0; ks0 < kKS; ks0 += PD) {
    const char* n0 = w0 + (size_t)(ks0 + PD) * 1024;   // refill source (runs PD steps past the end: padded)
    const char* n1 = w1 + (size_t)(ks0 + PD) * 1024;
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int cur = APRE ? (p & 1) : 0;
      if (APRE) load_a(fa[cur ^ 1], (ks0 + p + 1) & (kKS - 1));   // next step's A fragments first
      else load_a(fa[0], ks0 + p);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][0], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[p][1], acc[i][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      fb[p][0] = *reinterpret_cast<const bf16x8*>(n0 + p * 1024 + loff);
      fb[p][1] = *reinterpret_cast<const bf16x8*>(n1 + p * 1024 + loff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int PD, int APRE>
__global__ __launch_bounds__(512, 2) void k_panel(const __bf16* Wp, const __bf16* Afill, float* out, int rounds,
                                                  int n_gemm, int members_per_xcd, int panels_per_member) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  // panel <- random bf16 (once; the real kernel rewrites it from its epilogues)
  for (int q = tid; q < 128 * 64; q += 512) {
    const int row = q >> 6, cc = q & 63;
    *reinterpret_cast<u32x4*>(smem + row * kPitch + cc * 16) =
        *reinterpret_cast<const u32x4*>(Afill + ((size_t)(blockIdx.x & 63) * 128 + row) * 512 + cc * 8);
  }
  __syncthreads();
  f32x16 acc[4][2];
  float keep = 0.f;
  for (int r = 0; r < rounds; ++r) {
    const int item = r * per_xcd + slot;
    const int e = xcd * members_per_xcd + (item / panels_per_member) % members_per_xcd;
    const __bf16* wp = Wp + (size_t)e * 512 * 512;
    for (int gsel = 0; gsel < n_gemm; ++gsel) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
      panel_gemm<PD, APRE>(acc, smem, wp, wave, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) keep += acc[i][j][(r + gsel) & 15];
      __syncthreads();   // the real kernel has a barrier between a contraction and the next panel write
    }
  }
  out[(size_t)blockIdx.x * 512 + tid] = keep;
}

static double time_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

template <int MODE, int DEPTH>
static void run_stream(const char* label, const char* buf, size_t region_bytes, int shared, int wgs, int threads) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  uint32_t* sink; CHECK(hipMalloc(&sink, 4));
  const int iters = 4096;
  const size_t lds = MODE == 1 ? (size_t)(threads / 64) * DEPTH * 1024 : 0;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stream<MODE, DEPTH>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_stream<MODE, DEPTH>), dim3(wgs), dim3(threads), lds, 0, buf, region_bytes, shared, iters, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
  }
  const double ms = time_ms(a, b);
  const double bytes = (double)wgs * (threads / 64) * iters * 1024.0;
  printf("stream %-34s wgs %4d x %4d thr depth %2d : %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU@2.4GHz\n", label, wgs, threads,
         DEPTH, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
  CHECK(hipFree(sink));
}

template <int PD, int APRE>
static void run_panel(const __bf16* Wp, const __bf16* Afill, float* out, int n_gemm) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int rounds = 20, wgs = 256;
  const size_t lds = 128 * kPitch;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<PD, APRE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_panel<PD, APRE>), dim3(wgs), dim3(512), lds, 0, Wp, Afill, out, rounds, n_gemm, 8, 80);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
  }
  CHECK(hipGetLastError());
  const double ms = time_ms(a, b);
  const double flops = (double)wgs * rounds * n_gemm * 2.0 * 128 * 512 * 512;
  printf("panel  PD %d  APRE %d  n_gemm %d : %8.3f ms  %7.1f TFLOP/s  (%4.1f %% of 2.5 PF)   B stream %6.2f TB/s\n", PD, APRE, n_gemm, ms,
         flops / ms / 1e9, 100.0 * flops / (ms * 1e-3) / 2.5e15,
         (double)wgs * rounds * n_gemm * 512.0 * 1024 / ms / 1e9);
}

int main() {
  // weights: 64 members x 512 KiB ; A fill: 64 x 128 x 512 bf16 ; HBM stream region: 2 GiB
  const size_t wbytes = (size_t)64 * 512 * 512 * 2;
  std::vector<uint16_t> h(wbytes / 2);
  uint32_t s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    // bf16 in (-1, 1): sign + exponent 0x3e/0x3f range
    v = (uint16_t)(((s >> 16) & 0x8000u) | 0x3e00u | ((s >> 8) & 0x1ffu));
  }
  __bf16 *Wp, *Afill;
  float* out;
  char* big;
  CHECK(hipMalloc(&Wp, wbytes + (1 << 20)));   // the B stream runs PD KiB past its end
  CHECK(hipMalloc(&Afill, (size_t)64 * 128 * 512 * 2));
  CHECK(hipMalloc(&out, (size_t)256 * 512 * 4));
  const size_t big_bytes = (size_t)2 << 30;
  CHECK(hipMalloc(&big, big_bytes));
  CHECK(hipMemcpy(Wp, h.data(), wbytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Afill, h.data(), (size_t)64 * 128 * 512 * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(big, 1, big_bytes));

  printf("== L2-resident (8 x 512 KiB regions, one per XCD)\n");
  run_stream<0, 8>("global->VGPR  L2", (const char*)Wp, 512 * 1024, 1, 256, 512);
  run_stream<0, 16>("global->VGPR  L2", (const char*)Wp, 512 * 1024, 1, 256, 512);
  run_stream<0, 8>("global->VGPR  L2", (const char*)Wp, 512 * 1024, 1, 512, 512);
  run_stream<1, 4>("LDS-DMA       L2", (const char*)Wp, 512 * 1024, 1, 256, 512);
  run_stream<1, 8>("LDS-DMA       L2", (const char*)Wp, 512 * 1024, 1, 256, 512);
  run_stream<1, 8>("LDS-DMA       L2", (const char*)Wp, 512 * 1024, 1, 512, 512);
  printf("== HBM (every workgroup streams its own 4 MiB region of a 2 GiB buffer)\n");
  run_stream<0, 8>("global->VGPR  HBM", big, (size_t)4 << 20, 0, 512, 512);
  run_stream<0, 16>("global->VGPR  HBM", big, (size_t)4 << 20, 0, 512, 512);
  run_stream<1, 8>("LDS-DMA       HBM", big, (size_t)4 << 20, 0, 512, 512);
  printf("== panel contraction (128 x 512 x 512 per workgroup pass, 256 workgroups x 20 panels)\n");
  run_panel<2, 0>(Wp, Afill, out, 3);
  run_panel<4, 0>(Wp, Afill, out, 3);
  run_panel<8, 0>(Wp, Afill, out, 3);
  run_panel<2, 1>(Wp, Afill, out, 3);
  run_panel<4, 1>(Wp, Afill, out, 3);
  run_panel<8, 1>(Wp, Afill, out, 3);
  return 0;
}
