// developer micro-benchmark: LDS instruction cost on gfx950 by instruction, active lanes and address pattern
// (cycles of the CU's LDS pipe per wave-instruction, 16 waves per CU all issuing)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define UNR 16
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int nact, int stride_b, int spread) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((unsigned*)lds)[i] = 0xffffffffu;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // address: wave-private region (8 KB each), lane * stride (spread = 0: every lane its own; 1: pairs of lanes share; ...)
  unsigned addr = wave * 8192 + ((spread ? lane / spread : lane) * stride_b) % 8000;
  addr &= ~7u;
  unsigned long long key = ((unsigned long long)(0x40000000u - lane) << 32) | lane;
  float acc = 0.f;
  unsigned long long r64 = 0;
  unsigned r32 = 0, r32b = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (lane < nact) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if constexpr (OP == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(r32) : "v"(addr));
        if constexpr (OP == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(r64) : "v"(addr));
        if constexpr (OP == 2) asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:9" : "=v"(r64) : "v"(addr));
        if constexpr (OP == 3) asm volatile("ds_min_u64 %0, %1" : : "v"(addr), "v"(key) : "memory");
        if constexpr (OP == 4) asm volatile("ds_min_rtn_u64 %0, %1, %2" : "=v"(r64) : "v"(addr), "v"(key) : "memory");
        if constexpr (OP == 5) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(acc) : "memory");
        if constexpr (OP == 6) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(r32b) : "memory");
        if constexpr (OP == 7) asm volatile("ds_write_b16 %0, %1" : : "v"(addr), "v"(r32b) : "memory");
        if constexpr (OP == 8) asm volatile("ds_min_u32 %0, %1" : : "v"(addr), "v"(r32b) : "memory");
        if constexpr (OP == 9) asm volatile("ds_read_u16 %0, %1" : "=v"(r32) : "v"(addr));
        if constexpr (OP == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(*(__uint128_t*)&r64) : "v"(addr & ~15u));
        if constexpr (OP == 11) asm volatile("ds_max_rtn_u32 %0, %1, %2" : "=v"(r32) : "v"(addr), "v"(r32b) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      key -= 1ull << 32;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)r64 + (float)r32 + lds[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, int nact, int stride_b, int spread) {
  float* out; unsigned long long* cyc;
  const int iters = 400, blocks = 512;            // 2 workgroups of 8 waves per CU = 16 waves per CU
  CHK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  CHK(hipMalloc(&cyc, (size_t)blocks * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<OP><<<blocks, 512>>>(out, cyc, 4, nact, stride_b, spread);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  k<OP><<<blocks, 512>>>(out, cyc, iters, nact, stride_b, spread);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
  CHK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
  const double ninst = (double)iters * UNR;        // per wave
  // per CU: 16 waves x ninst instructions during avg ticks
  printf("%-16s lanes %2d stride %3d share %d: %.1f ticks per wave-instr per CU (%.1f per wave), wall %.3f ms\n", name, nact, stride_b, spread,
         avg / (ninst * 16), avg / ninst, ms);
  free(h); CHK(hipFree(out)); CHK(hipFree(cyc));
}
int main() {
  const int lanes[] = {64, 32, 16, 8, 1};
  for (int n : lanes) run<0>("ds_read_b32", n, 40, 0);
  run<0>("ds_read_b32", 64, 4, 0);
  for (int n : lanes) run<2>("ds_read2_b32", n, 40, 0);
  run<1>("ds_read_b64", 64, 40, 0); run<1>("ds_read_b64", 64, 8, 0); run<1>("ds_read_b64", 16, 40, 0);
  run<10>("ds_read_b128", 64, 80, 0); run<10>("ds_read_b128", 64, 64, 0); run<10>("ds_read_b128", 64, 16, 0); run<10>("ds_read_b128", 64, 64, 4);
  run<9>("ds_read_u16", 64, 2, 0);
  for (int n : lanes) run<3>("ds_min_u64", n, 40, 0);
  run<3>("ds_min_u64", 64, 40, 4); run<3>("ds_min_u64", 64, 40, 64);
  for (int n : lanes) run<4>("ds_min_rtn_u64", n, 40, 0);
  run<4>("ds_min_rtn_u64", 64, 40, 4);
  for (int n : lanes) run<5>("ds_add_f32", n, 12, 0);
  run<5>("ds_add_f32", 64, 4, 0); run<5>("ds_add_f32", 64, 12, 4);
  for (int n : lanes) run<6>("ds_write_b32", n, 40, 0);
  run<6>("ds_write_b32", 64, 4, 0);
  run<7>("ds_write_b16", 64, 2, 0); run<7>("ds_write_b16", 16, 2, 0);
  for (int n : lanes) run<8>("ds_min_u32", n, 40, 0);
  for (int n : lanes) run<11>("ds_max_rtn_u32", n, 40, 0);
  return 0;
}
