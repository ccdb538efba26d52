// developer tool: stand-alone probe of the fault of DESIGN.md 7 -- packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32) in one kernel while ANOTHER kernel's waves run v_mfma_f32_32x32x16_f16 on the same SIMDs.
//   k_packed: every thread iterates x <- x * a + b on a float2 (packed instructions) and on two floats (plain instructions)
//             from the same inputs and counts the iterations after which the two disagree in any bit;
//   k_matrix: nothing but a chain of matrix instructions.
// Run 1: k_packed alone.  Run 2: k_packed on one stream, k_matrix on another, launched so that they overlap.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_mfma_hazard.hip -o tools/ubench/pk_mfma_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool DIVERGENT>
__global__ __launch_bounds__(512) void k_packed(const float* in, int iters, unsigned long long* bad, float* sink) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const float a0 = in[(tid * 4) & 4095], a1 = in[(tid * 4 + 1) & 4095], b0 = in[(tid * 4 + 2) & 4095], b1 = in[(tid * 4 + 3) & 4095];
  f32x2 x = {a0, b1}, a = {0.5f + 0.25f * a0, 0.5f + 0.25f * a1}, b = {b0, b1};
  float s0 = a0, s1 = b1;
  const float sa0 = 0.5f + 0.25f * a0, sa1 = 0.5f + 0.25f * a1;
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    if (!DIVERGENT || ((tid * 7 + it) % 3) != 0) {
      // packed: v_pk_mul_f32 + v_pk_add_f32 (two roundings, as the plain pair below)
      f32x2 t;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(a));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(t), "v"(b));
      float u0, u1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(s0), "v"(sa0));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(s1), "v"(sa1));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(u0), "v"(b0));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(u1), "v"(b1));
      if (__float_as_uint(x[0]) != __float_as_uint(s0) || __float_as_uint(x[1]) != __float_as_uint(s1)) {
        ++nbad;
        x[0] = s0; x[1] = s1;          // resynchronise: count events, not their consequences
      }
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[tid] = x[0] + x[1] + s0 + s1;
}

// the staging arithmetic of the selection kernel as hipcc's SLP vectoriser packs it (v_pk_add_f32 with op_sel / neg modifiers,
// v_pk_mul_f32, v_pk_fma_f32) against the same expressions in plain instructions (inline asm keeps them unpacked)
__device__ __forceinline__ float pl_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pl_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pl_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__global__ __launch_bounds__(512) void k_packed_staging(const float* in, int iters, unsigned long long* bad, float* sink) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nbad = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int o = (tid * 6 + it * 7) & 4089;
    const float x0 = in[o], y0 = in[o + 1], x1 = in[o + 2], y1 = in[o + 3], x2 = in[o + 4], y2 = in[o + 5];
    if (((tid + it) % 3) == 0) continue;                 // divergent, as the staging is (only lanes with candidates)
    // (the compiler is free to pack these)
    const float l01 = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
    const float l02 = (x2 - x0) * (x2 - x0) + (y2 - y0) * (y2 - y0);
    const float l12 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
    const float ar = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0);
    // plain twins, same operations in the same order (contraction: the compiler forms fma(a, a, b * b) and fma(a, b, -(c * d)))
    const float dx01 = pl_sub(x1, x0), dy01 = pl_sub(y1, y0), dx02 = pl_sub(x2, x0), dy02 = pl_sub(y2, y0), dx12 = pl_sub(x2, x1), dy12 = pl_sub(y2, y1);
    const float m01 = pl_fma(dx01, dx01, pl_mul(dy01, dy01)), m02 = pl_fma(dx02, dx02, pl_mul(dy02, dy02)), m12 = pl_fma(dx12, dx12, pl_mul(dy12, dy12));
    const float n01 = pl_fma(dy01, dy01, pl_mul(dx01, dx01)), n02 = pl_fma(dy02, dy02, pl_mul(dx02, dx02)), n12 = pl_fma(dy12, dy12, pl_mul(dx12, dx12));
    // either contraction order is accepted (a difference of one rounding is not the fault looked for: it is constant
    // from run to run -- the host compares the COUNT with and without the matrix kernel beside it)
    const bool ok = (l01 == m01 || l01 == n01) && (l02 == m02 || l02 == n02) && (l12 == m12 || l12 == n12);
    if (!ok) ++nbad;
    acc += l01 + l02 + l12 + ar;
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[tid] = acc;
}

__global__ __launch_bounds__(512) void k_matrix(const _Float16* in, int iters, float* sink) {
  extern __shared__ float pad[];          // LDS only to steer how many of these workgroups a CU holds
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 64) & 4095]; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i];
  if (threadIdx.x == 0 && pad) sink[blockIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000, reps = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<float> h(4096);
  std::vector<_Float16> hh(4096);
  unsigned sd = 12345u;
  for (int i = 0; i < 4096; ++i) { sd = sd * 1664525u + 1013904223u; h[i] = ((sd >> 8) & 0xffff) / 65536.f - 0.5f; hh[i] = (_Float16)(h[i] * 0.01f); }
  float *din, *sink, *sink2; _Float16* dh; unsigned long long* bad;
  CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&dh, 4096 * 2)); CK(hipMalloc(&sink, 4096 * 512 * 4)); CK(hipMalloc(&sink2, 4096 * 4)); CK(hipMalloc(&bad, 8));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hh.data(), 4096 * 2, hipMemcpyHostToDevice));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  CK(hipFuncSetAttribute((const void*)k_matrix, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int mode = 0; mode < 2; ++mode) {
    unsigned long long total = 0, z = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice));
      if (mode) hipLaunchKernelGGL(k_matrix, dim3(512), dim3(512), 40 * 1024, s2, dh, iters, sink2);
      hipLaunchKernelGGL(k_packed_staging, dim3(512), dim3(512), 0, s1, din, iters / 4, bad, sink);
      if (mode) hipLaunchKernelGGL(k_matrix, dim3(512), dim3(512), 40 * 1024, s2, dh, iters, sink2);
      CK(hipDeviceSynchronize());
      unsigned long long n; CK(hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost));
      total += n;
    }
    printf("compiler-packed staging arithmetic %s: %llu results differ from the plain twins (a constant count = contraction order; a count that changes with the matrix kernel = the fault)\n",
           mode ? "BESIDE the matrix kernel" : "alone", total);
  }
  for (int mode = 0; mode < 4; ++mode) {
    const bool with_matrix = mode & 1, divergent = mode & 2;
    unsigned long long total = 0, z = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice));
      // 512 workgroups of 512 threads each side: two of each fit a CU (the matrix kernel asks for 40 KB of LDS)
      if (with_matrix) hipLaunchKernelGGL(k_matrix, dim3(512), dim3(512), 40 * 1024, s2, dh, iters * 2, sink2);
      if (divergent) hipLaunchKernelGGL(k_packed<true>, dim3(512), dim3(512), 0, s1, din, iters, bad, sink);
      else hipLaunchKernelGGL(k_packed<false>, dim3(512), dim3(512), 0, s1, din, iters, bad, sink);
      if (with_matrix) hipLaunchKernelGGL(k_matrix, dim3(512), dim3(512), 40 * 1024, s2, dh, iters * 2, sink2);
      CK(hipDeviceSynchronize());
      unsigned long long n; CK(hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost));
      total += n;
    }
    printf("packed kernel %s, %s: %llu iterations (of %.3g) where packed and plain arithmetic disagree\n", with_matrix ? "BESIDE the matrix kernel" : "alone",
           divergent ? "divergent lanes" : "all lanes active", total, (double)reps * 512 * 512 * iters);
  }
  return 0;
}
