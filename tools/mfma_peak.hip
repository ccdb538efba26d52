// developer micro-benchmark: sustained fp32 MFMA rate (v_mfma_f32_32x32x2f32 / 16x16x4f32) with register operands only
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  if (KIND == 0) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2];
  } else {
    f32x4 c[6];
    for (int j = 0; j < 6; ++j) c[j] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 6; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + c[4][0] + c[5][1];
  }
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int wpb = 1; wpb <= 2; ++wpb) {       // blocks per CU multiplier
      const int blocks = 256 * 2 * wpb, iters = 4000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 24 * (kind == 0 ? 4096.0 : 2048.0);
        if (rep == 2) printf("%s waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "32x32x2" : "16x16x4", 2 * wpb, ms, flops / ms / 1e9);
      }
    }
  return 0;
}
