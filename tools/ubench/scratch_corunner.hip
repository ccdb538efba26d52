// developer tool (tools/c4_probe.py, MHHIP_C4_FAKEKIND=scratch|noscratch): a co-runner for the selection kernel that does
// nothing but arithmetic on a per-thread array -- indexed at run time, so the array lives in scratch (private segment) in
// one build of the kernel and, fully unrolled, in registers in the other.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/scratch_corunner.hip -o tools/ubench/scratch_corunner.so
#include <hip/hip_runtime.h>
template <bool SCRATCH>
__global__ __launch_bounds__(256) void k_corunner(const int* idx, float* out, int iters) {
  float a[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) a[i] = (float)(threadIdx.x + i);
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (SCRATCH) {
      const int j = idx[(it + threadIdx.x) & 1023] & 63;      // run-time index: the array cannot stay in registers
      a[j] = a[j] * 1.0001f + s;
      s += a[(j + 7) & 63];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = a[i] * 1.0001f + s; s += a[(i + 7) & 7]; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mode 2: nothing but LDS -- a workgroup of 512 threads fills `lds_bytes` of dynamic LDS with zeros and sums it back, `iters` times
__global__ __launch_bounds__(512) void k_lds_corunner(float* out, int words, int iters) {
  extern __shared__ float sm[];
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < words; i += 512) sm[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += 512) s += sm[i];
    __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
extern "C" int corunner_launch_lds(float* out, int blocks, int lds_bytes, int iters, void* stream) {
  static int done = 0;
  if (!done) { hipFuncSetAttribute((const void*)k_lds_corunner, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = 1; }
  hipLaunchKernelGGL(k_lds_corunner, dim3(blocks), dim3(512), (size_t)lds_bytes, (hipStream_t)stream, out, lds_bytes / 4, iters);
  return (int)hipGetLastError();
}
extern "C" int corunner_launch(int scratch, const int* idx, float* out, int blocks, int iters, void* stream) {
  if (scratch) hipLaunchKernelGGL(k_corunner<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, out, iters);
  else hipLaunchKernelGGL(k_corunner<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, out, iters);
  return (int)hipGetLastError();
}
