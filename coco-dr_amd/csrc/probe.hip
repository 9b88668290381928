// Hardware probes: pin the two gfx950 lane layouts every MFMA kernel in this library relies on.
// The GPU tests compare them with a plain matmul / gather, so a wrong assumption shows up as a
// failed probe rather than as a subtly wrong encoder.
#include "common.h"

namespace {

// One wave: out[i][j] = sum_k a[i][k] * b[j][k] with the operand / accumulator lane maps used by
// gemm.hip and attention.hip (lane l: row l & 31, k-slots 8*(l>>5)..+7; D: col = l & 31,
// row = (r & 3) + 8*(r >> 2) + 4*(l >> 5)).
__global__ void probe_mfma32_kernel(const uint16_t* a, const uint16_t* b, float* out) {
  const int l = threadIdx.x;
  const bf16x8 fa = as_bf16x8(*reinterpret_cast<const uint4*>(a + (l & 31) * 16 + (l >> 5) * 8));
  const bf16x8 fb = as_bf16x8(*reinterpret_cast<const uint4*>(b + (l & 31) * 16 + (l >> 5) * 8));
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    out[i * 32 + (l & 31)] = acc[r];
  }
}

// One wave: tile [16][64] bf16 row-major in LDS; group g = l >> 4 reads the [4][16] block at
// rows 4g.., cols 16*((g+1)&3)..; lane c passes &tile[4g + c/4][c0 + 4*(c%4)] and must get back
// tile[4g + j][c0 + c], j = 0..3.
__global__ void probe_tr16_kernel(const uint16_t* tile, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) char lds[16 * 128];
  const int l = threadIdx.x;
  for (int q = l; q < 16 * 8; q += 64) *reinterpret_cast<uint4*>(lds + q * 16) = *reinterpret_cast<const uint4*>(tile + q * 8);
  __syncthreads();
  const int g = l >> 4, c = l & 15;
  const int r0 = 4 * g, c0 = 16 * ((g + 1) & 3);
  const s16x4 v = lds_read_tr16(lds, (r0 + (c >> 2)) * 128 + (c0 + ((c & 3) << 2)) * 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

}  // namespace

extern "C" int cocodr_probe_mfma32(const uint16_t* a, const uint16_t* b, float* out, cocodr_stream_t stream) {
  CK_ARG(a && b && out, "probe_mfma32: null pointer");
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
  CK_LAUNCH("probe_mfma32");
  return COCODR_OK;
}

extern "C" int cocodr_probe_tr16(const uint16_t* tile, uint16_t* out, cocodr_stream_t stream) {
  CK_ARG(tile && out, "probe_tr16: null pointer");
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tile, out);
  CK_LAUNCH("probe_tr16");
  return COCODR_OK;
}
