// Per-CU load-rate probe for gfx950: one workgroup per CU streams an L2-resident region, (A) with the LDS DMA
// (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR), (B) with global_load_dwordx4 into VGPRs, (C) = B + ds_write_b128.
// Prints GB/s per CU with all 256 CUs active.  Question behind it: is the ~67 GB/s per CU the GEMM loaders reach a limit
// of the DMA path or of the CU's vector-memory path as such?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/load_rate.hip -o tools/probes/_build/load_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define LDS_PTR(T) __attribute__((address_space(3))) T*

constexpr int REGION = 128 * 1024;  // bytes per workgroup (32 CUs of an XCD: 4 MiB = its L2)

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void rate_kernel(const uint4* __restrict__ src, uint4* __restrict__ sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* base = (const char*)src + (size_t)blockIdx.x * REGION;
  constexpr int PER_IT = NT * 16 * 8;  // bytes one iteration of the workgroup moves (8 loads of 16 B per lane)
  uint4 acc = make_uint4(0, 0, 0, 0);
  if constexpr (MODE == 0) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, REGION, 0x00020000);
    for (int it = 0; it < iters; ++it) {
      const uint32_t off0 = (uint32_t)((it * PER_IT) % REGION);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t o = (off0 + (uint32_t)((j * NT + wid * 64) * 16)) % REGION;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(smem + ((j * NT / 64 + wid) % 64) * 1024), 16, o + lane * 16, 0, 0, 0);
      }
      if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<uint4*>(smem + tid * 16);
  } else {
    for (int it = 0; it < iters; ++it) {
      const uint32_t off0 = (uint32_t)((it * PER_IT) % REGION);
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t o = (off0 + (uint32_t)((j * NT + tid) * 16)) % REGION;
        v[j] = *reinterpret_cast<const uint4*>(base + o);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 2) *reinterpret_cast<uint4*>(smem + ((j * NT + tid) * 16) % (64 * 1024)) = v[j];
        acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w;
      }
    }
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x * NT + tid] = acc;
}

template <int MODE, int NT>
void run(const char* name, const uint4* src, uint4* sink) {
  const int iters = 2048;
  hipFuncSetAttribute((const void*)rate_kernel<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 32}) {
    hipLaunchKernelGGL((rate_kernel<MODE, NT>), dim3(grid), dim3(NT), 96 * 1024, 0, src, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<MODE, NT>), dim3(grid), dim3(NT), 96 * 1024, 0, src, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * NT * 16 * 8;
    printf("%-44s %3d threads, %3d workgroups: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, NT, grid, bytes / ms / 1e6, bytes * grid / ms / 1e9);
  }
}

int main() {
  uint4 *src, *sink;
  hipMalloc(&src, (size_t)256 * REGION);
  hipMemset(src, 1, (size_t)256 * REGION);
  hipMalloc(&sink, 256 * 512 * 16);
  run<0, 256>("A: buffer_load_dwordx4 ... lds (DMA)", src, sink);
  run<0, 512>("A: buffer_load_dwordx4 ... lds (DMA)", src, sink);
  run<1, 256>("B: global_load_dwordx4 -> VGPR", src, sink);
  run<1, 512>("B: global_load_dwordx4 -> VGPR", src, sink);
  run<2, 256>("C: global_load_dwordx4 -> VGPR -> ds_write", src, sink);
  run<2, 512>("C: global_load_dwordx4 -> VGPR -> ds_write", src, sink);
  return 0;
}
