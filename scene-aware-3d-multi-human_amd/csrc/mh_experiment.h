// Timing and ablation builds of the kernels (developer experiments: tools/mkvariant.sh <name> <file>.hip -DMH_EXPERIMENT -DR_TIMING=n,
// tools/pair_stats.py, tools/time_lbs_phases.py).  The PRODUCT build never defines MH_EXPERIMENT: every macro below is then
// empty and the kernels carry one line per probe instead of the probe.  (the #if blocks used to sit in
// the kernels.)
//
// k_raster_strip, -DR_TIMING=1|2: wave-elapsed shader cycles per phase, summed over the waves into the pair-counter slots (units
//   of 1024 cycles; four phases per build, two 32-bit halves per counter): 0 tile prologue, 1 round head (gathers issued, box,
//   staging, scan), 2 cull walk + pair list, 3 pair evaluation, 4 even-split path, 5 wait for the tile's other waves, 6 tile
//   epilogue, 7 everything.  -DR_TIMING=3: life span of every workgroup on the 100 MHz clock (MHHIP_SPANS=1 prints them).
// k_raster_grads, -DR_TIMING=4|5: 0 unit header + body sums, 1 classification loads + table clear, 2 compaction, 3 pixels,
//   4 reductions + flush, 7 everything.
// k_raster_prepare, -DR_TIMING=6|7 (workgroups that SORT): 0 window + motion test, 1 histogram clear + row coordinates, 2 first
//   pass (gathers, row ranges, histogram), 3 reductions + scan, 4 second pass (placement), 5 tiles, 6 everything, 7 their number.
#pragma once

#if defined(MH_EXPERIMENT) && defined(R_TIMING)
#define MH_EXPERIMENT_TIMING 1
#define R_TMARK(c) do { const unsigned long long t1_ = __builtin_readcyclecounter(); tacc[c] += (unsigned)(t1_ - tlast); tlast = t1_; } while (0)
#define R_TIMING_DECL() \
  unsigned tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  unsigned long long tlast = __builtin_readcyclecounter(); \
  const unsigned long long tbegin = tlast; \
  const unsigned long long twall = wall_clock64(); (void)tbegin; (void)twall
#else
#define R_TMARK(c) do { } while (0)
#define R_TIMING_DECL() do { } while (0)
#endif

// k_raster_strip, -DMH_EXPERIMENT -DR_NEAR_PROBE: how much work would an exact re-evaluation of the
// decisions the fast pair evaluation can get wrong be?  Counter A = candidate pairs one of whose insertion decisions lies
// inside the fast path's error bound (an edge function within R_NEAR_W of zero; a distance within R_NEAR_REL of a blur
// radius; a depth within R_NEAR_REL of one of the pixel's five keys of another face), counter B = 64-pair batches that hold
// at least one such pair -- a re-evaluation is a divergent branch: it costs a batch, not a pair.  tools/pair_stats.py prints both.
#if defined(MH_EXPERIMENT) && defined(R_NEAR_PROBE)
#define R_NEAR_W 1e-5f
#define R_NEAR_REL 1e-6f
#define R_NEAR_ARG , &wmin_
#define R_NEAR_VARS() unsigned n_near = 0u, n_nbatch = 0u
#define R_NEAR_FLUSH() \
  do { \
    __syncthreads(); \
    if (lane == 0) { s_cnt[wave][0] = 0ull; s_cnt[wave][1] = 0ull; } \
    __syncthreads(); \
    atomicAdd(&s_cnt[wave][0], (unsigned long long)n_near); atomicAdd(&s_cnt[wave][1], (unsigned long long)n_nbatch); \
  } while (0)
#define R_NEAR_DECL() float wmin_ = 1.f
#define R_NEAR_COUNT(q_, pz_, in_, dd_, f_) \
  do { \
    const unsigned long long* nq_ = (q_); \
    bool nr_ = wmin_ < R_NEAR_W; \
    if (!(in_)) nr_ = nr_ || fabsf((dd_) - BLUR_D) < R_NEAR_REL * BLUR_D || fabsf((dd_) - BLUR_S) < R_NEAR_REL * BLUR_S; \
    if ((in_) || (dd_) < BLUR_D) \
      for (int k_ = 0; k_ < 5; ++k_) { \
        const unsigned long long e_ = nq_[k_]; \
        if (e_ != RS_EMPTY && (int)(e_ & 0xffffffffu) != (f_)) nr_ = nr_ || fabsf((pz_) - __uint_as_float((unsigned)(e_ >> 32))) <= R_NEAR_REL * fabsf(pz_); \
      } \
    const unsigned long long nm_ = __ballot(nr_); \
    n_near += nr_ ? 1u : 0u; \
    if (nm_ && (int)(threadIdx.x & 63) == __ffsll((long long)nm_) - 1) n_nbatch += 1u; \
  } while (0)
#else
#define R_NEAR_ARG
#define R_NEAR_VARS() do { } while (0)
#define R_NEAR_FLUSH() do { } while (0)
#define R_NEAR_DECL() do { } while (0)
#define R_NEAR_COUNT(q_, pz_, in_, dd_, f_) do { } while (0)
#endif

#if defined(MH_EXPERIMENT) && defined(R_TIMING) && R_TIMING == 3
#define R_TIMING_FLUSH() \
do { \
  if (p.pairs && tid == 0 && (int)blockIdx.x < total) { \
    unsigned long long* slot = p.pairs + 2 + 2 * (size_t)blockIdx.x; \
    slot[0] = twall; slot[1] = wall_clock64(); \
    if (blockIdx.x == 0) p.pairs[0] += 1ull; \
  } \
} while (0)
#define R_TIMING_HOST_SPANS() \
  { \
    unsigned long long t0 = ~0ull, t1 = 0ull; \
    for (int i = 0; i < R_STRIP_GRID; ++i) { \
      const unsigned long long a = host[2 + 2 * i], b = host[3 + 2 * i]; \
      if (!a || b < a) continue; \
      t0 = a < t0 ? a : t0; t1 = b > t1 ? b : t1; \
      out_host[2] += b - a; \
    } \
    out_host[1] = t1 > t0 ? t1 - t0 : 0ull; \
    if (getenv("MHHIP_SPANS")) { \
      fprintf(stderr, "spans of the last launch (us): kernel %.1f\n", (double)(t1 - t0) / 100.); \
      for (int d = 0; d < 12; ++d) { \
        const int i0 = d * (R_STRIP_GRID / 12), i1 = (d + 1) * (R_STRIP_GRID / 12); \
        double sd = 0, me = 0, ms = 0; int n = 0; \
        for (int i = i0; i < i1; ++i) { \
          const unsigned long long a = host[2 + 2 * i], b = host[3 + 2 * i]; \
          if (!a || b < a) continue; \
          sd += (double)(b - a) / 100.; ++n; \
          if ((double)(b - t0) / 100. > me) me = (double)(b - t0) / 100.; \
          ms += (double)(a - t0) / 100.; \
        } \
        fprintf(stderr, "  workgroups %4d..%4d: %4d ran, mean start %6.1f, mean life %5.1f, latest end %6.1f\n", i0, i1 - 1, n, n ? ms / n : 0., n ? sd / n : 0., me); \
      } \
      for (int k = 0; k < 8; ++k) { \
        int best = -1; \
        for (int i = 0; i < R_STRIP_GRID; ++i) \
          if (host[2 + 2 * i] && host[3 + 2 * i] > host[2 + 2 * i] && (best < 0 || host[3 + 2 * i] > host[3 + 2 * best])) best = i; \
        if (best < 0) break; \
        fprintf(stderr, "  ends at %6.1f: workgroup %4d, started %6.1f, life %5.1f\n", (double)(host[3 + 2 * best] - t0) / 100., best, \
                (double)(host[2 + 2 * best] - t0) / 100., (double)(host[3 + 2 * best] - host[2 + 2 * best]) / 100.); \
        host[3 + 2 * best] = host[2 + 2 * best]; \
      } \
    } \
    return MH_OK; \
  }
#elif defined(MH_EXPERIMENT) && defined(R_TIMING) && R_TIMING >= 4
#define R_TIMING_FLUSH() do { } while (0)
#define R_TIMING_HOST_SPANS() do { } while (0)
#elif defined(MH_EXPERIMENT) && defined(R_TIMING)
#define R_TIMING_FLUSH() \
do { \
  if (p.pairs) { \
    tacc[7] = (unsigned)(__builtin_readcyclecounter() - tbegin); \
    if (lane == 0) { \
      unsigned long long* slot = p.pairs + 2 + 2 * (size_t)blockIdx.x; \
      const int c = (R_TIMING - 1) * 4; \
      atomicAdd(slot, (unsigned long long)(tacc[c] >> 10) | ((unsigned long long)(tacc[c + 1] >> 10) << 32)); \
      atomicAdd(slot + 1, (unsigned long long)(tacc[c + 2] >> 10) | ((unsigned long long)(tacc[c + 3] >> 10) << 32)); \
      if (blockIdx.x == 0 && wave == 0) p.pairs[0] += 1ull; \
    } \
  } \
} while (0)
#define R_TIMING_HOST_SPANS() do { } while (0)
#else
#define R_TIMING_HOST_SPANS() do { } while (0)
#endif

#if defined(MH_EXPERIMENT) && defined(R_TIMING) && R_TIMING >= 4
#define RG_TMARK(c) R_TMARK(c)
#define RG_TIMING_DECL() \
  unsigned tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  unsigned long long tlast = __builtin_readcyclecounter(); \
  const unsigned long long tbegin = tlast;
#else
#define RG_TMARK(c) do { } while (0)
#define RG_TIMING_DECL() do { } while (0)
#endif
#if defined(MH_EXPERIMENT) && defined(R_TIMING) && R_TIMING >= 4 && R_TIMING <= 5
#define RG_TIMING_FLUSH() \
do { \
  if (p.pairs) { \
    tacc[7] = (unsigned)(__builtin_readcyclecounter() - tbegin); \
    if ((tid & 63) == 0) { \
      unsigned long long* slot = p.pairs + 2 + 2 * (size_t)(blockIdx.x % R_STRIP_GRID); \
      const int c = (R_TIMING - 4) * 4; \
      atomicAdd(slot, (unsigned long long)(tacc[c] >> 10) | ((unsigned long long)(tacc[c + 1] >> 10) << 32)); \
      atomicAdd(slot + 1, (unsigned long long)(tacc[c + 2] >> 10) | ((unsigned long long)(tacc[c + 3] >> 10) << 32)); \
      if (blockIdx.x == 0 && tid == 0) p.pairs[0] += 1ull; \
    } \
  } \
} while (0)
#else
#define RG_TIMING_FLUSH() do { } while (0)
#endif

#if defined(MH_EXPERIMENT) && defined(R_TIMING) && R_TIMING >= 6
#define P_MARK(c) do { const unsigned long long t1_ = __builtin_readcyclecounter(); pacc[c] += (unsigned)(t1_ - plast); plast = t1_; } while (0)
#define P_TIMING_FLUSH() \
do { \
  if (p.pairs && tid == 0 && (p.margin == 0 || any_moved)) { \
    P_MARK(5); \
    pacc[6] = (unsigned)(__builtin_readcyclecounter() - pbegin); \
    pacc[7] = 1; \
    unsigned long long* slot = p.pairs + 2 + 2 * (size_t)(blockIdx.x % R_STRIP_GRID); \
    const int c = (R_TIMING - 6) * 4; \
    atomicAdd(slot, (unsigned long long)pacc[c] | ((unsigned long long)pacc[c + 1] << 32)); \
    atomicAdd(slot + 1, (unsigned long long)pacc[c + 2] | ((unsigned long long)pacc[c + 3] << 32)); \
    if (blockIdx.x == 0) p.pairs[0] += 1ull; \
  } \
} while (0)
#else
#define P_MARK(c) do { } while (0)
#define P_TIMING_FLUSH() do { } while (0)
#endif

// k_skin_fwd16 / k_skinbwd16, -DLBS_TIMING (tools/time_lbs_phases.py)
// timing builds (tools/mkvariant.sh x mh_lbs.hip -DLBS_TIMING; tools/time_lbs_phases.py): wave-elapsed shader cycles of the two
// skinning kernels by phase, summed over the waves.  forward: 0 staging, 1 constants + matrix phase, 2 epilogue, 3 everything,
// 4 waves; backward: 8 staging, 9 stage, 10 wait A, 11 blend, 12 wait B, 13 matrix phase, 14 everything, 15 waves
#if defined(MH_EXPERIMENT) && defined(LBS_TIMING)
__device__ unsigned long long g_lbs_t[16];
#define L_T0() unsigned long long lt_ = __builtin_readcyclecounter(); const unsigned long long lt_begin = lt_; unsigned lacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define L_MARK(c) do { const unsigned long long t1_ = __builtin_readcyclecounter(); lacc[c] += (unsigned)(t1_ - lt_); lt_ = t1_; } while (0)
#define L_OUT(base, n) do { if ((threadIdx.x & 63) == 0) { for (int c_ = 0; c_ < (n); ++c_) atomicAdd(&g_lbs_t[(base) + c_], (unsigned long long)lacc[c_]); \
    atomicAdd(&g_lbs_t[(base) + (n)], __builtin_readcyclecounter() - lt_begin); atomicAdd(&g_lbs_t[(base) + (n) + 1], 1ull); } } while (0)
extern "C" int mh_lbs_debug_timing(unsigned long long* out16) {
  MH_HIP(hipDeviceSynchronize());
  MH_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lbs_t), sizeof(g_lbs_t)));
  unsigned long long z[16] = {0};
  MH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lbs_t), z, sizeof(z)));
  return MH_OK;
}
#else
#define L_T0() do { } while (0)
#define L_MARK(c) do { } while (0)
#define L_OUT(base, n) do { } while (0)
#endif
