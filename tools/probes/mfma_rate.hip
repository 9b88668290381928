// MFMA issue-rate probe for gfx950: ns per v_mfma_f32_32x32x16_bf16 and wave with NACC independent accumulators, one wave per
// SIMD (256 threads, 16 accumulators = 256 registers -> AGPRs) and two waves per SIMD (512 threads, 8 accumulators), every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/_build/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;

template <int NACC, int THREADS, bool ZERO>
__global__ __launch_bounds__(THREADS) void mfma_kernel(float* out, int seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  v4i a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = ZERO ? 0 : 0x3f803f80 + ((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x007f007f);
    a[i] = v4i{x, x ^ 0x00110011, x ^ 0x00230023, x ^ 0x00050005};
    b[i] = v4i{x ^ 0x00070007, x, x ^ 0x00310031, x ^ 0x00130013};
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[i & 3]), __builtin_bit_cast(bf16x8, a[(i >> 2) & 3]), acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}

// the same stream with the register classes pinned by asm constraints: ACC_V = accumulators in ArchVGPRs and operands in AccVGPRs
// (what a one-wave-per-SIMD 128 x 128 tile would need to escape the AGPR-accumulator ceiling), or the other way round
template <int NACC, bool ACC_V, bool ZERO>
__global__ __launch_bounds__(256) void mfma_pinned_kernel(float* out, int seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  v4i a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = ZERO ? 0 : 0x3f803f80 + ((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x007f007f);
    a[i] = v4i{x, x ^ 0x00110011, x ^ 0x00230023, x ^ 0x00050005};
    b[i] = v4i{x ^ 0x00070007, x, x ^ 0x00310031, x ^ 0x00130013};
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (ACC_V)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(b[i & 3]), "a"(a[(i >> 2) & 3]));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(b[i & 3]), "v"(a[(i >> 2) & 3]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool ACC_V, bool ZERO>
void run_pinned(const char* name, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_pinned_kernel<NACC, ACC_V, ZERO>), dim3(grid), dim3(256), 0, 0, out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((mfma_pinned_kernel<NACC, ACC_V, ZERO>), dim3(grid), dim3(256), 0, 0, out, 1);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double n = (double)ITERS * NACC;
  const double flops = (double)grid * 4 * n * 32768.0;
  printf("%-60s %.3f ms  %.1f ns per MFMA and SIMD  %.0f TFLOP/s\n", name, ms, ms * 1e6 / n, flops / ms / 1e9);
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// the same FLOPs with v_mfma_f32_16x16x32_bf16: NACC accumulators of 4 registers
template <int NACC, int THREADS, bool ZERO>
__global__ __launch_bounds__(THREADS) void mfma16_kernel(float* out, int seed) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  v4i a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int x = ZERO ? 0 : 0x3f803f80 + ((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x007f007f);
    a[i] = v4i{x, x ^ 0x00110011, x ^ 0x00230023, x ^ 0x00050005};
    b[i] = v4i{x ^ 0x00070007, x, x ^ 0x00310031, x ^ 0x00130013};
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[i & 7]), __builtin_bit_cast(bf16x8, a[(i >> 3) & 7]), acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int NACC, int THREADS, bool ZERO>
void run16(const char* name, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma16_kernel<NACC, THREADS, ZERO>), dim3(grid), dim3(THREADS), 0, 0, out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((mfma16_kernel<NACC, THREADS, ZERO>), dim3(grid), dim3(THREADS), 0, 0, out, 1);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double n = (double)ITERS * NACC;
  const double flops = (double)grid * (THREADS / 64) * n * 16384.0;
  printf("%-44s %.3f ms  %.1f ns per 16x16x32 MFMA and SIMD  %.0f TFLOP/s\n", name, ms, ms * 1e6 / (n * (THREADS / 256)), flops / ms / 1e9);
}

template <int NACC, int THREADS, bool ZERO>
void run(const char* name, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_kernel<NACC, THREADS, ZERO>), dim3(grid), dim3(THREADS), 0, 0, out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((mfma_kernel<NACC, THREADS, ZERO>), dim3(grid), dim3(THREADS), 0, 0, out, 1);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double n = (double)ITERS * NACC;                       // MFMAs per wave
  const double per_simd = n * (THREADS / 256);                 // MFMAs per SIMD (grid = one workgroup per CU)
  const double flops = (double)grid * (THREADS / 64) * n * 32768.0;
  printf("%-44s %.3f ms  %.1f ns per MFMA and SIMD  %.0f TFLOP/s\n", name, ms, ms * 1e6 / per_simd * (grid > 256 ? 256.0 / grid : 1.0), flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 512 * 4);
  run<16, 256, false>("1 wave/SIMD, 16 acc (AGPR), random operands", out, 256);
  run<16, 256, true>("1 wave/SIMD, 16 acc (AGPR), zero operands", out, 256);
  run<8, 256, false>("1 wave/SIMD, 8 acc, random operands", out, 256);
  run<4, 256, false>("1 wave/SIMD, 4 acc, random operands", out, 256);
  run<8, 512, false>("2 waves/SIMD, 8 acc each, random operands", out, 256);
  run<8, 512, true>("2 waves/SIMD, 8 acc each, zero operands", out, 256);
  run<16, 256, false>("1 wave/SIMD, 16 acc, 64 CUs only", out, 64);
  run<12, 256, false>("1 wave/SIMD, 12 acc, random operands", out, 256);
  run<4, 512, false>("2 waves/SIMD, 4 acc each, random operands", out, 256);
  run<16, 256, false>("1 wave/SIMD, 16 acc (again, after warm chip)", out, 256);
  run16<64, 256, false>("16x16x32: 1 wave/SIMD, 64 acc (256 regs)", out, 256);
  run16<32, 256, false>("16x16x32: 1 wave/SIMD, 32 acc (128 regs)", out, 256);
  run16<32, 512, false>("16x16x32: 2 waves/SIMD, 32 acc each", out, 256);
  run16<64, 256, true>("16x16x32: 1 wave/SIMD, 64 acc, zero operands", out, 256);
  // (operands: 4 A x 4 B registers as in the compiler-allocated kernels above, so the data toggling per MFMA is the same)
  run_pinned<16, true, false>("pinned: 16 acc in VGPRs, operands in AGPRs", out, 256);
  run_pinned<16, false, false>("pinned: 16 acc in AGPRs, operands in VGPRs", out, 256);
  run_pinned<15, true, false>("pinned: 15 acc in VGPRs, operands in AGPRs", out, 256);
  run_pinned<15, false, false>("pinned: 15 acc in AGPRs, operands in VGPRs", out, 256);
  run_pinned<14, false, false>("pinned: 14 acc in AGPRs, operands in VGPRs", out, 256);
  run_pinned<12, true, false>("pinned: 12 acc in VGPRs, operands in AGPRs", out, 256);
  run_pinned<12, false, false>("pinned: 12 acc in AGPRs, operands in VGPRs", out, 256);
  run_pinned<8, false, false>("pinned: 8 acc in AGPRs, operands in VGPRs", out, 256);
  run_pinned<16, true, false>("pinned: 16 acc in VGPRs again (warm chip)", out, 256);
  run_pinned<16, false, false>("pinned: 16 acc in AGPRs again (warm chip)", out, 256);
  run_pinned<16, false, true>("pinned: 16 acc in AGPRs, zero operands", out, 256);
  return 0;
}
