// VALU issue-rate probe for gfx950: cycles per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32 and
// v_cvt_pk_bf16_f32, with one and with two waves per SIMD (s_memtime deltas of one wave; all waves run the same loop).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o tools/probes/_build/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, CH = 16;

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, unsigned long long* cyc, float seed) {
  float a[CH];
  f2 p[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = f2{a[i], a[i] + 0.5f}; }
  const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
  const f2 m2 = f2{m, m}, c2 = f2{c, c};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if constexpr (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
      if constexpr (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
      if constexpr (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if constexpr (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if constexpr (MODE == 4) { uint32_t w; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(a[i]), "v"(a[(i + 1) % CH])); a[i] = __uint_as_float(w | 0x3f800000u); }
      if constexpr (MODE == 5) p[i] = p[i] * m2;
      if constexpr (MODE == 6) a[i] = a[i] * m;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, float* out, unsigned long long* cyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  const double n = (double)ITERS * CH;
  printf("%-22s waves/SIMD %d: %.2f counter ticks / instr / wave, kernel %.3f ms -> %.2f ns / instr / wave\n", name, threads / 256, avg / n, ms,
         ms * 1e6 / n);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  for (int threads : {256, 512}) {
    run<0>("v_fma_f32", threads, out, cyc);
    run<1>("v_pk_fma_f32 (2 elem)", threads, out, cyc);
    run<6>("v_mul_f32", threads, out, cyc);
    run<5>("v_pk_mul_f32 (2 elem)", threads, out, cyc);
    run<2>("v_exp_f32", threads, out, cyc);
    run<3>("v_rcp_f32", threads, out, cyc);
    run<4>("v_cvt_pk_bf16_f32", threads, out, cyc);
  }
  return 0;
}
