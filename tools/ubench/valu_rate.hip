// developer micro-benchmark: issue cost of VALU instruction classes on gfx950 (cycles per wave64 instruction per SIMD)
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NREG 16
#define UNR 8
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float a[NREG];
  f2 p[NREG];
  int ia[NREG];
  const float x = 1.0f + 1e-7f * threadIdx.x, y = 1e-9f * threadIdx.x;
  const f2 x2 = {x, x}, y2 = {y, y};
#pragma unroll
  for (int i = 0; i < NREG; ++i) { a[i] = (float)i + threadIdx.x; p[i] = (f2){a[i], a[i] + 1.f}; ia[i] = i + threadIdx.x; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(x2), "v"(y2));
      if constexpr (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 3) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
      if constexpr (OP == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if constexpr (OP == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(x2));
      if constexpr (OP == 6) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(y2));
      if constexpr (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 8) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
      if constexpr (OP == 9) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
      if constexpr (OP == 10) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 11) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if constexpr (OP == 12) asm volatile("v_fma_f32 %0, %2, %0, %3\n v_pk_fma_f32 %1, %4, %1, %5" : "+v"(a[i]), "+v"(p[i]) : "v"(x), "v"(y), "v"(x2), "v"(y2));
      if constexpr (OP == 13) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
      if constexpr (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));

      if constexpr (OP == 16) asm volatile("v_and_b32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 17) asm volatile("v_or_b32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 19) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 20) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(ia[i]));
      if constexpr (OP == 21) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 22) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 23) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 24) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % NREG]));
      if constexpr (OP == 25) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 26) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 27) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if constexpr (OP == 28) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 29) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia[i]));
      if constexpr (OP == 30) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ia[i]));
      if constexpr (OP == 31) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : "+v"(a[i]) : "v"(y) : "vcc");
      if constexpr (OP == 32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
      if constexpr (OP == 33) asm volatile("v_mul_f32 %0, %1, %0 clamp" : "+v"(a[i]) : "v"(x));
      if constexpr (OP == 34) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 35) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 36) asm volatile("v_min_u32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 37) asm volatile("v_max_i32 %0, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 38) asm volatile("v_cmp_lt_u32 vcc, %1, %0" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]) : "vcc");
      if constexpr (OP == 39) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(ia[i]));
      if constexpr (OP == 40) asm volatile("v_fma_f32 %0, %1, %0, %2 clamp" : "+v"(a[i]) : "v"(x), "v"(y));
      if constexpr (OP == 41) asm volatile("v_mul_f32 %0, |%1|, -%0" : "+v"(a[i]) : "v"(x));
      if constexpr (OP == 42) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 43) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == 44) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(ia[i]));
      if constexpr (OP == 45) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) % NREG]));
      if constexpr (OP == 46) asm volatile("v_ffbl_b32 %0, %0" : "+v"(ia[i]));
      if constexpr (OP == 47) asm volatile("v_readlane_b32 s20, %0, 3" : "+v"(ia[i]) : : "s20");
      if constexpr (OP == 49) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
      if constexpr (OP == 50) asm volatile("v_fma_f32 %0, %1, %0, 1.0" : "+v"(a[i]) : "v"(x));
      if constexpr (OP == 51) asm volatile("v_add_u32 %0, 12, %0" : "+v"(ia[i]));
      if constexpr (OP == 52) asm volatile("v_mul_f32 %0, s4, %0" : "+v"(a[i]));
      if constexpr (OP == 53) asm volatile("v_cmp_lt_f32 s[20:21], %1, %0\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(y) : "s20", "s21");
      if constexpr (OP == 15) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ia[i]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += a[i] + p[i][0] + p[i][1] + (float)ia[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, int per_op) {
  float* out; unsigned long long* cyc;
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 4) {         // waves per SIMD
    const int blocks = 256 * wps;                 // 256 threads = 4 waves = one per SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHK(hipMalloc(&cyc, (size_t)blocks * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(out, cyc, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(out, cyc, iters);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
    CHK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
    const double ninst = (double)iters * UNR * NREG * per_op;
    // per SIMD: wps waves each issuing ninst instructions
    printf("%-22s waves/SIMD %d: %.2f counter ticks per instr per wave, %.3f ns per instr per SIMD (wall %.3f ms)\n", name, wps,
           avg / ninst, ms * 1e6 / (ninst * wps), ms);
    free(h); CHK(hipFree(out)); CHK(hipFree(cyc));
  }
}
int main() {
  run<0>("v_fma_f32", 1); run<14>("v_fmac_f32", 1); run<1>("v_pk_fma_f32", 1); run<12>("fma + pk_fma", 2);
  run<4>("v_mul_f32", 1); run<5>("v_pk_mul_f32", 1); run<9>("v_sub_f32", 1); run<6>("v_pk_add_f32", 1);
  run<3>("v_max_f32", 1); run<13>("v_min_f32", 1); run<11>("v_med3_f32", 1); run<8>("v_cmp+v_cndmask", 2);
  run<2>("v_add_u32", 1); run<10>("v_mul_u32_u24", 1); run<15>("v_lshlrev_b32", 1); run<7>("v_rcp_f32", 1);
  run<49>("v_add_f32", 1); run<50>("v_fma_f32 const", 1); run<51>("v_add_u32 imm", 1); run<52>("v_mul_f32 sgpr", 1);
  run<16>("v_and_b32", 1); run<17>("v_or_b32", 1); run<35>("v_xor_b32", 1); run<18>("v_lshl_add_u32", 1); run<19>("v_mad_u32_u24", 1); run<20>("v_bfe_u32", 1);
  run<21>("v_cvt_f32_i32", 1); run<22>("v_cvt_i32_f32", 1); run<23>("v_floor_f32", 1); run<24>("v_mov_b32", 1); run<25>("v_add3_u32", 1);
  run<26>("v_lshl_or_b32", 1); run<27>("v_max3_f32", 1); run<28>("v_mul_lo_u32", 1); run<29>("v_mov_b32_dpp", 1); run<30>("v_add_u32_dpp", 1);
  run<31>("v_cmp_lt_f32", 1); run<32>("v_cndmask_b32", 1); run<53>("v_cmp sgpr+cndmask", 2); run<33>("v_mul_f32 clamp", 1); run<40>("v_fma_f32 clamp", 1); run<41>("v_mul_f32 abs/neg", 1);
  run<34>("v_sub_u32", 1); run<36>("v_min_u32", 1); run<37>("v_max_i32", 1); run<38>("v_cmp_lt_u32", 1); run<39>("v_ashrrev_i32", 1);
  run<42>("v_sqrt_f32", 1); run<43>("v_exp_f32", 1); run<44>("v_mbcnt_lo", 1); run<45>("v_bcnt", 1); run<46>("v_ffbl_b32", 1); run<47>("v_readlane", 1);
  return 0;
}
