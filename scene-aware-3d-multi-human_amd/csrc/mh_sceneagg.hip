// Scene aggregation on the device: what the reference does on the host with numpy / OpenCV once
// per cycle >= 30 (optimizer.py:578-584; fhsog.py:180-202; utils.py:91-135, 174-209):
//   per-pixel masked median over time of the metric depth 1/target_disp  ->  bilateral filter of the disparity  ->
//   Sobel edge mask of disparity and depth (global std / mean thresholds), eroded twice, times the median's validity
//   ->  iterative 7x7 median fill of the masked pixels  ->  un-projection of the valid pixels (compacted on the device,
//   the count stays in device memory so nothing synchronises with the host).
// OpenCV is not available offline; bilateral / Sobel / erode follow its documented semantics (BORDER_REFLECT_101,
// circular bilateral support, constant +inf border for erode); the checker is oracle/scene_oracle.py + analytic pins.
#include "mh_common.h"

// ---- per-frame depth range (optimizer.py:683-688): inv_min = 1/min_z, inv_max = 1/max_z ------------------------------
__global__ void k_scene_ranges(int T, const float* zmin_lin, const float* zmax_lin, float* invz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float min_z = logf(1.f + expf(zmin_lin[t]));
  const float max_z = min_z + 1.f + logf(1.f + expf(zmax_lin[t]));
  invz[2 * t] = 1.f / min_z;
  invz[2 * t + 1] = 1.f / max_z;
}

// ---- masked median over time ---------------------------------------------------------------------------------------------
// One wave per 64 consecutive pixels (coalesced frame reads); every lane keeps its pixel's T depth values as raw bits in
// an LDS column [t][lane] (conflict-free) and radix-selects the upper median bit by bit (positive floats order like
// unsigned integers), then one more pass finds the lower median for an even count.  np.ma.median semantics: mean of
// the two middle values, 0 for an all-masked pixel.
#define SM_INVALID 0xffffffffu
template <bool IN_LDS>
__global__ __launch_bounds__(64) void k_scene_median(int T, int P, const float* depths, const unsigned char* backmask,
                                                     const float* invz, float* ma_depth, float* ma_mask) {
  extern __shared__ unsigned col[];                       // [T][64] when IN_LDS
  const int lane = threadIdx.x, p = blockIdx.x * 64 + lane;
  const bool live = p < P;
  auto value = [&](int t) -> unsigned {
    if (!live || backmask[(size_t)t * P + p] == 0) return SM_INVALID;
    if (!invz) return __float_as_uint(depths[(size_t)t * P + p]);                      // raw non-negative values (colour planes)
    const float inv_min = invz[2 * t], inv_max = invz[2 * t + 1];
    const float disp = depths[(size_t)t * P + p] * (inv_min - inv_max) + inv_max;      // optimizer.py:425
    return __float_as_uint(1.0f / disp);                                               // :426
  };
  int n = 0;
  for (int t = 0; t < T; ++t) {
    const unsigned v = value(t);
    if (IN_LDS) col[t * 64 + lane] = v;
    n += v != SM_INVALID;
  }
  if (!live) return;
  if (n == 0) { ma_depth[p] = 0.f; ma_mask[p] = 0.f; return; }
  auto get = [&](int t) -> unsigned { return IN_LDS ? col[t * 64 + lane] : value(t); };
  // k-th smallest (0-based) by bitwise descent
  int k = n >> 1;
  unsigned prefix = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned hi = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
    int c0 = 0;
    for (int t = 0; t < T; ++t) {
      const unsigned v = get(t);
      c0 += ((v & (hi | (1u << bit))) == prefix) ? 1 : 0;
    }
    if (k >= c0) { k -= c0; prefix |= 1u << bit; }
  }
  float med = __uint_as_float(prefix);
  if ((n & 1) == 0) {
    // lower middle: equal to the upper one unless exactly n/2 values are strictly smaller
    int less = 0;
    unsigned best = 0u;
    for (int t = 0; t < T; ++t) {
      const unsigned v = get(t);
      if (v < prefix) { ++less; best = v > best ? v : best; }
    }
    const float lo = (less == (n >> 1)) ? __uint_as_float(best) : med;
    med = (lo + med) / 2.f;
  }
  ma_depth[p] = med;
  ma_mask[p] = 1.f;
}

// The same median with the frames of a pixel contiguous in memory (depths_t / backmask_t are (P, T)): one wave per
// pixel, the values of a pixel spread over the lanes in registers, the bitwise descent done with ballots.  No LDS at
// all (the column form above pins 51 KB of LDS per wave for T = 200 and crowds the concurrently running loss kernels
// off the CUs) and ~6x less time.  T <= 64 * SMT_NV (8: 512 frames; 32: 2048, the whole sequence of a pixel-sharded
// multi-GPU run).
typedef float f32x2s __attribute__((ext_vector_type(2)));
// The per-frame depth range (k_scene_ranges' arithmetic, the same bits) is evaluated here by the lane that holds the frame:
// one launch less on the update's stream -- every kernel boundary there costs the optimiser's chain ~1.3 us (round 6).
__device__ __forceinline__ f32x2s sm_range(const float* zmin_lin, const float* zmax_lin, int t) {
  const float min_z = logf(1.f + expf(zmin_lin[t]));
  const float max_z = min_z + 1.f + logf(1.f + expf(zmax_lin[t]));
  return (f32x2s){1.f / min_z, 1.f / max_z};
}
template <int SMT_NV>
__global__ __launch_bounds__(256) void k_scene_median_t(int T, int P, const float* depths_t, const unsigned char* backmask_t,
                                                        const float* zmin_lin, const float* zmax_lin, float* ma_depth, float* ma_mask) {
  const bool invz = zmin_lin != nullptr;
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float* dp = depths_t + (size_t)p * T;
  const unsigned char* bp = backmask_t + (size_t)p * T;
  unsigned v[SMT_NV];
  int n = 0;
  if (SMT_NV <= 8) {
    // every load of the pixel asked for at once, unconditionally (clamped frame index): behind the mask test they were
    // two dependent round trips per 64 frames, and beside the optimiser's own kernels a wave waits long for each
    unsigned char bm[SMT_NV];
    float dv[SMT_NV];
    f32x2s iz[SMT_NV];
#pragma unroll
    for (int j = 0; j < SMT_NV; ++j) {
      const int t = min(j * 64 + lane, T - 1);
      bm[j] = bp[t];
      dv[j] = dp[t];
      iz[j] = invz ? sm_range(zmin_lin, zmax_lin, t) : (f32x2s){0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < SMT_NV; ++j) {
      const int t = j * 64 + lane;
      v[j] = SM_INVALID;
      if (t < T && bm[j] != 0) {
        if (invz) {
          const float inv_min = iz[j][0], inv_max = iz[j][1];
          const float disp = dv[j] * (inv_min - inv_max) + inv_max;      // optimizer.py:425
          v[j] = __float_as_uint(1.0f / disp);                           // :426
        } else {
          v[j] = __float_as_uint(dv[j]);                                 // raw non-negative values (colour planes)
        }
      }
      n += __popcll(__ballot(v[j] != SM_INVALID));
    }
  } else {
#pragma unroll
    for (int j = 0; j < SMT_NV; ++j) {
      const int t = j * 64 + lane;
      v[j] = SM_INVALID;
      if (t < T && bp[t] != 0) {
        if (invz) {
          const f32x2s iz = sm_range(zmin_lin, zmax_lin, t);
          const float inv_min = iz[0], inv_max = iz[1];
          const float disp = dp[t] * (inv_min - inv_max) + inv_max;      // optimizer.py:425
          v[j] = __float_as_uint(1.0f / disp);                           // :426
        } else {
          v[j] = __float_as_uint(dp[t]);                                 // raw non-negative values (colour planes)
        }
      }
      n += __popcll(__ballot(v[j] != SM_INVALID));
    }
  }
  if (n == 0) {
    if (lane == 0) { ma_depth[p] = 0.f; ma_mask[p] = 0.f; }
    return;
  }
  // bits on which all valid values agree need no counting pass (the depths of a pixel share sign and most of the exponent)
  unsigned all_or = 0u, all_and = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < SMT_NV; ++j)
    if (v[j] != SM_INVALID) { all_or |= v[j]; all_and &= v[j]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    all_or |= (unsigned)__shfl_xor((int)all_or, o, 64);
    all_and &= (unsigned)__shfl_xor((int)all_and, o, 64);
  }
  const unsigned vary = all_or ^ all_and;
  int k = n >> 1, left = n;                            // left: values that match the prefix so far
  unsigned prefix = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    if (((vary >> bit) & 1u) == 0u) {                  // wave-uniform
      prefix |= all_and & (1u << bit);
      continue;
    }
    const unsigned m = (bit == 31 ? 0u : (0xffffffffu << (bit + 1))) | (1u << bit);
    int c0 = 0;
#pragma unroll
    for (int j = 0; j < SMT_NV; ++j) c0 += __popcll(__ballot((v[j] & m) == prefix));
    if (k >= c0) { k -= c0; prefix |= 1u << bit; left -= c0; } else { left = c0; }
    if (left == 1) {
      // one value is left under this prefix: it is the answer (distinct depths part ways after ~log2 n of their ~25
      // varying bits; the counting passes over the rest were two thirds of the kernel)
#pragma unroll
      for (int j = 0; j < SMT_NV; ++j) {
        const unsigned long long bal = __ballot((v[j] & m) == prefix);
        if (bal) prefix = (unsigned)__builtin_amdgcn_readlane((int)v[j], __builtin_ctzll(bal));
      }
      break;
    }
  }
  float med = __uint_as_float(prefix);
  if ((n & 1) == 0) {
    int less = 0;
    unsigned best = 0u;
#pragma unroll
    for (int j = 0; j < SMT_NV; ++j) {
      const bool lt = v[j] < prefix;
      less += __popcll(__ballot(lt));
      if (lt) best = v[j] > best ? v[j] : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)best, o, 64);
      best = other > best ? other : best;
    }
    const float lo = (less == (n >> 1)) ? __uint_as_float(best) : med;
    med = (lo + med) / 2.f;
  }
  if (lane == 0) { ma_depth[p] = med; ma_mask[p] = 1.f; }
}

__device__ __forceinline__ int r101(int i, int n) {       // BORDER_REFLECT_101, |offset| < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---- bilateral filter of 1/clip(depth): d = 9 (radius 4, circular support), sigmaColor 0.05, sigmaSpace 25 -------------------
__global__ void k_scene_bilateral(int H, int W, const float* depth, float* depth_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  auto src = [&](int yy, int xx) { return 1.0f / fminf(fmaxf(depth[r101(yy, H) * W + r101(xx, W)], 0.01f), 100.f); };
  const float c = src(y, x);
  double num = 0.0, den = 0.0;
  for (int dy = -4; dy <= 4; ++dy)
    for (int dx = -4; dx <= 4; ++dx) {
      if (dy * dy + dx * dx > 16) continue;
      const float q = src(y + dy, x + dx);
      const float dq = q - c;
      const float w = expf((float)(-(double)(dy * dy + dx * dx) / (2.0 * 25.0 * 25.0)) - (dq * dq) / (float)(2.0 * 0.05 * 0.05));
      num += (double)(w * q);
      den += (double)w;
    }
  const float disp = (float)(num / den);
  depth_out[p] = 1.0f / fminf(fmaxf(disp, 0.01f), 100.f);
}

// ---- Sobel magnitudes of disparity and depth + their global sums --------------------------------------------------------------
__device__ __forceinline__ double blk_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

// Sobel magnitudes of disparity and depth (utils.py:174-190) and, per workgroup, their sums and sums of squares -- plain
// stores into part[workgroup][4] (rounds 3-5 added them to four global doubles with atomics: a memset before, an order of
// additions that changed from run to run).
__global__ __launch_bounds__(256) void k_scene_sobel(int H, int W, const float* depth, float* g_disp, float* g_depth, double* part) {
  __shared__ double sh[4];
  const int p = blockIdx.x * 256 + threadIdx.x;
  double a = 0.0, a2 = 0.0, b = 0.0, b2 = 0.0;
  if (p < H * W) {
    const int y = p / W, x = p - y * W;
    const float ks[3] = {1.f, 2.f, 1.f}, kd[3] = {-1.f, 0.f, 1.f};
    float gxd = 0.f, gyd = 0.f, gxz = 0.f, gyz = 0.f;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const float z = depth[r101(y + i - 1, H) * W + r101(x + j - 1, W)];
        const float d = 1.0f / fminf(fmaxf(z, 0.1f), 100.f);
        gxd += ks[i] * kd[j] * d; gyd += kd[i] * ks[j] * d;
        gxz += ks[i] * kd[j] * z; gyz += kd[i] * ks[j] * z;
      }
    const float gd = fabsf(gxd) + fabsf(gyd), gz = fabsf(gxz) + fabsf(gyz);
    g_disp[p] = gd; g_depth[p] = gz;
    a = gd; a2 = (double)gd * gd; b = gz; b2 = (double)gz * gz;
  }
  a = blk_sum(a, sh); a2 = blk_sum(a2, sh); b = blk_sum(b, sh); b2 = blk_sum(b2, sh);
  if (threadIdx.x == 0) { double* o = part + 4 * (size_t)blockIdx.x; o[0] = a; o[1] = a2; o[2] = b; o[3] = b2; }
}

// grad = g_disp / std(g_disp) + g_depth / std(g_depth); dmask = erode^2(1 - [grad > 3 mean(grad)]) * mask (two 3x3 erosions
// with an ignoring border = one 5x5 erosion) -- as ONE launch behind the Sobel kernel (rounds 3-5: two, with a third global
// sum between them; every kernel boundary on the update's stream costs the optimiser's chain ~1.3 us, round 6).  Every
// workgroup adds the Sobel workgroups' partial sums itself, in their order; grad is linear in the two magnitudes, so its
// mean is mean(g_disp) / std(g_disp) + mean(g_depth) / std(g_depth) -- no second global sum; and a neighbour's grad is
// evaluated where it is compared (the same two IEEE divisions and addition that used to be stored: the same bits).
__global__ __launch_bounds__(256) void k_scene_edges2(int H, int W, const float* g_disp, const float* g_depth, const float* ma_mask,
                                                      const double* part, int nparts, float* dmask) {
  __shared__ double s_tot[4];
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int w = 0; w < nparts; ++w) t += part[4 * (size_t)w + threadIdx.x];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
  const int P = H * W;
  const double n = (double)P;
  const double ma = s_tot[0] / n, mb = s_tot[2] / n;
  const float sa = (float)sqrt(fmax(s_tot[1] / n - ma * ma, 0.0)), sb = (float)sqrt(fmax(s_tot[3] / n - mb * mb, 0.0));
  const float thr = 3.f * (float)(ma / (double)sa + mb / (double)sb);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int y = p / W, x = p - y * W;
  float keep = 1.f;
  for (int dy = -2; dy <= 2; ++dy)
    for (int dx = -2; dx <= 2; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const int q = yy * W + xx;
      if (g_disp[q] / sa + g_depth[q] / sb > thr) keep = 0.f;
    }
  dmask[p] = ma_mask ? keep * ma_mask[p] : keep;
}

// ---- iterative median fill (utils.py:91-135 looped by :206-207) ---------------------------------------------------------------
// One workgroup: the masked pixels form a shrinking list; a sweep fills every listed pixel that sees a valid pixel in its
// window from the values of the valid pixels only (so a sweep is a pure function of the previous state), then the
// updates are applied together.  Ends when the list is empty (or nothing can be filled any more).
#define FILL_TH 1024
// median of the valid pixels of the KS x KS window around (y, x): all window loads are issued together (the loop form
// waited for one mask load and one depth load after the other: 49 x 2 dependent round trips per pixel, 270 us per
// sweep), invalid entries become +inf, and a bitonic network sorts the window in registers (static indexing only: a
// per-thread array indexed at run time lives in scratch memory).  Returns the number of valid pixels.
// windows of more than 64 pixels (9 x 9, 11 x 11: the scene image, once per fit): the same with rolled loops on a
// per-thread array (the network of 128 does not fit the register file next to 1024 threads anyway)
template <int KS>
__device__ int fill_window_median_big(int H, int W, int y, int x, const float* depth, const float* mask, float* med) {
  constexpr int k = KS / 2;
  float v[KS * KS];
  int c = 0;
  for (int yy = max(0, y - k); yy < min(H, y + k + 1); ++yy)
    for (int xx = max(0, x - k); xx < min(W, x + k + 1); ++xx)
      if (mask[yy * W + xx] > 0.f) {
        const float val = depth[yy * W + xx];       // insertion sort
        int j = c++;
        while (j > 0 && v[j - 1] > val) { v[j] = v[j - 1]; --j; }
        v[j] = val;
      }
  if (c == 0) return 0;
  *med = (c & 1) ? v[c >> 1] : (v[(c >> 1) - 1] + v[c >> 1]) / 2.f;
  return c;
}

template <int KS, int NP>
__device__ __forceinline__ int fill_window_median(int H, int W, int y, int x, const float* depth, const float* mask, float* med) {
  if constexpr (NP > 64) return fill_window_median_big<KS>(H, W, y, x, depth, mask, med);
  constexpr int k = KS / 2;
  float v[NP];
  int c = 0;
#pragma unroll
  for (int dy = 0; dy < KS; ++dy)
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      const int yy = y + dy - k, xx = x + dx - k;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int q = in ? yy * W + xx : 0;
      const float m = mask[q], d = depth[q];
      const bool ok = in && m > 0.f;
      v[dy * KS + dx] = ok ? d : INFINITY;
      c += ok ? 1 : 0;
    }
#pragma unroll
  for (int i = KS * KS; i < NP; ++i) v[i] = INFINITY;
  if (c == 0) return 0;
#pragma unroll
  for (int kk = 2; kk <= NP; kk <<= 1)
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const float lo = fminf(v[i], v[l]), hi = fmaxf(v[i], v[l]);
          if ((i & kk) == 0) { v[i] = lo; v[l] = hi; } else { v[i] = hi; v[l] = lo; }
        }
      }
  const int k1 = (c - 1) >> 1, k2 = c >> 1;
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) {
    m1 = i == k1 ? v[i] : m1;
    m2 = i == k2 ? v[i] : m2;
  }
  *med = (c & 1) ? m2 : (m1 + m2) / 2.f;
  return c;
}

// One workgroup: the masked pixels form a shrinking list; a sweep fills every listed pixel that sees a valid pixel in its
// window from the values of the valid pixels only (so a sweep is a pure function of the previous state), then the
// updates are applied together.  Ends when the list is empty (or nothing can be filled any more).
template <int KS, int NP, int NT>
__global__ __launch_bounds__(NT) void k_scene_fill(int H, int W, int truncate, float* depth, float* mask, int* list_a, int* list_b,
                                                   float* upd) {
  __shared__ int s_cnt, s_next, s_filled;
  const int tid = threadIdx.x, P = H * W;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int p = tid; p < P; p += NT)
    if (!(mask[p] > 0.f)) list_a[atomicAdd(&s_cnt, 1)] = p;
  __syncthreads();
  int* cur = list_a;
  int* nxt = list_b;
  for (;;) {
    const int n = s_cnt;
    if (n == 0) break;
    __syncthreads();
    if (tid == 0) { s_next = 0; s_filled = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      const int p = cur[i];
      const int y = p / W, x = p - y * W;
      float med = 0.f;
      const int c = fill_window_median<KS, NP>(H, W, y, x, depth, mask, &med);
      if (c > 0) {
        upd[i] = truncate ? floorf(med) : med;      // integer-valued planes (colour): the reference stores into uint8
        atomicAdd(&s_filled, 1);
      } else {
        upd[i] = -1.f;
        nxt[atomicAdd(&s_next, 1)] = p;
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT)
      if (upd[i] >= 0.f) { depth[cur[i]] = upd[i]; mask[cur[i]] = 1.f; }
    __threadfence_block();
    __syncthreads();
    const int filled = s_filled, rest = s_next;
    __syncthreads();
    if (tid == 0) s_cnt = filled > 0 ? rest : 0;       // nothing reachable: stop (the reference would loop forever)
    __syncthreads();
    int* t = cur; cur = nxt; nxt = t;
  }
}

static int scene_fill_launch(int H, int W, int ksize, int truncate, float* depth, float* mask, int* list_a, int* list_b, float* upd,
                             hipStream_t st) {
  // window sizes of the reference: 7 (scene depth, every cycle), 11 (scene image, once per fit); the others for completeness
#define FILL_CASE(KS, NP, NT)                                                                                              \
  case KS:                                                                                                                 \
    hipLaunchKernelGGL((k_scene_fill<KS, NP, NT>), dim3(1), dim3(NT), 0, st, H, W, truncate, depth, mask, list_a, list_b, upd); \
    break;
  switch (ksize | 1) {      // an even size covers the same pixels as the next odd one minus a row/column: not used by the reference
    FILL_CASE(3, 16, 1024)
    FILL_CASE(5, 32, 1024)
    FILL_CASE(7, 64, 1024)
    FILL_CASE(9, 128, 256)
    FILL_CASE(11, 128, 256)
    default: return -1;
  }
#undef FILL_CASE
  return 0;
}

// ---- un-projection of the valid pixels, compacted in row-major order (optimizer.py:605-613) ------------------------------------
__global__ __launch_bounds__(1024) void k_scene_points(int H, int W, const float* depth, const float* mask, float i00, float i01, float i10,
                                                       float i11, float cx, float cy, float* pts, int* count) {
  __shared__ int swave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = H * W;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int base = 0; base < P; base += 1024) {
    const int p = base + tid;
    const bool keep = p < P && mask[p] > 0.5f;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) swave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += swave[w];
    if (keep) {
      const int o = off + __popcll(m & ((1ull << lane) - 1ull));
      const float u = (float)(p % W) + 0.5f - cx, v = (float)(p / W) + 0.5f - cy;
      const float d = depth[p];
      pts[(size_t)o * 3] = d * (u * i00 + v * i10);
      pts[(size_t)o * 3 + 1] = d * (u * i01 + v * i11);
      pts[(size_t)o * 3 + 2] = d;
    }
    __syncthreads();
    if (tid == 0) {
      int s = 0;
      for (int w = 0; w < 16; ++w) s += swave[w];
      s_base += s;
    }
    __syncthreads();
  }
  if (tid == 0) *count = s_base;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static size_t sa_align(size_t x) { return (x + 255) & ~(size_t)255; }
struct SceneWs {
  float* invz;      // [T][2]
  float* depth1;    // [P] bilateral-filtered depth, then filled in place
  float* g_disp;    // [P]
  float* g_depth;   // [P]
  float* grad;      // [P] (round 6: the Sobel workgroups' partial sums, as doubles)
  float* dmask;     // [P]
  float* upd;       // [P]
  int* list_a;      // [P]
  int* list_b;      // [P]
  double* stats;    // [8]
};
static SceneWs scene_carve(void* ws, int P) {
  char* c = (char*)ws;
  SceneWs s;
  s.depth1 = (float*)c; c += sa_align((size_t)P * 4);
  s.g_disp = (float*)c; c += sa_align((size_t)P * 4);
  s.g_depth = (float*)c; c += sa_align((size_t)P * 4);
  s.grad = (float*)c; c += sa_align((size_t)P * 4);
  s.dmask = (float*)c; c += sa_align((size_t)P * 4);
  s.upd = (float*)c; c += sa_align((size_t)P * 4);
  s.list_a = (int*)c; c += sa_align((size_t)P * 4);
  s.list_b = (int*)c; c += sa_align((size_t)P * 4);
  s.stats = (double*)c; c += sa_align(64);
  s.invz = (float*)c;                                   // [T][2], last: the only T-dependent part
  return s;
}

extern "C" size_t mh_scene_workspace_bytes(int T, int H, int W) {
  const size_t P = (size_t)H * W;
  return 8 * sa_align(P * 4) + sa_align(64) + sa_align((size_t)(T > 0 ? T : 1) * 8);
}

extern "C" int mh_scene_median(int T, int H, int W, const float* depths, const uint8_t* backmask, const float* zmin_lin,
                               const float* zmax_lin, float* ma_depth, float* ma_mask, void* ws, void* stream) {
  MH_CHECK(depths && backmask && ma_depth && ma_mask && ws, "null argument");
  MH_CHECK((zmin_lin == nullptr) == (zmax_lin == nullptr), "depth-range leaves come in pairs (both NULL: median of the raw values)");
  MH_CHECK(T > 0 && H > 0 && W > 0, "empty input");
  const int P = H * W;
  SceneWs s = scene_carve(ws, P);
  hipStream_t st = (hipStream_t)stream;
  if (zmin_lin) {
    hipLaunchKernelGGL(k_scene_ranges, dim3((T + 127) / 128), dim3(128), 0, st, T, zmin_lin, zmax_lin, s.invz);
    MH_LAUNCH_CHECK();
  }
  const float* invz = zmin_lin ? (const float*)s.invz : (const float*)nullptr;
  const size_t lds = (size_t)T * 64 * sizeof(unsigned);
  if (lds <= 150 * 1024) {
    static unsigned char attr_set[MH_MAX_DEVICES];
    if (mh_first_on_device(attr_set))
      MH_HIP(hipFuncSetAttribute((const void*)k_scene_median<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(k_scene_median<true>, dim3((P + 63) / 64), dim3(64), lds, st, T, P, depths, backmask, invz,
                       ma_depth, ma_mask);
  } else {     // very long sequences: the values are recomputed from HBM/L2 in every pass
    hipLaunchKernelGGL(k_scene_median<false>, dim3((P + 63) / 64), dim3(64), 0, st, T, P, depths, backmask, invz,
                       ma_depth, ma_mask);
  }
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_scene_median_t(int T, int H, int W, const float* depths_t, const uint8_t* backmask_t, const float* zmin_lin,
                                 const float* zmax_lin, float* ma_depth, float* ma_mask, void* ws, void* stream) {
  MH_CHECK(depths_t && backmask_t && ma_depth && ma_mask && ws, "null argument");
  MH_CHECK((zmin_lin == nullptr) == (zmax_lin == nullptr), "depth-range leaves come in pairs (both NULL: median of the raw values)");
  MH_CHECK(T > 0 && H > 0 && W > 0, "empty input");
  MH_CHECK(T <= 2048, "sequence too long for the register form: use mh_scene_median");
  const int P = H * W;
  SceneWs s = scene_carve(ws, P);
  hipStream_t st = (hipStream_t)stream;
  (void)s;
  if (T <= 256) hipLaunchKernelGGL(k_scene_median_t<4>, dim3((P + 3) / 4), dim3(256), 0, st, T, P, depths_t, backmask_t, zmin_lin, zmax_lin, ma_depth, ma_mask);
  else if (T <= 512) hipLaunchKernelGGL(k_scene_median_t<8>, dim3((P + 3) / 4), dim3(256), 0, st, T, P, depths_t, backmask_t, zmin_lin, zmax_lin, ma_depth, ma_mask);
  else hipLaunchKernelGGL(k_scene_median_t<32>, dim3((P + 3) / 4), dim3(256), 0, st, T, P, depths_t, backmask_t, zmin_lin, zmax_lin, ma_depth, ma_mask);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_scene_postprocess(int H, int W, const float* ma_depth, const float* ma_mask, int use_bilateral, int fillin_ksize,
                                    float* scene_depth, void* ws, void* stream) {
  MH_CHECK(ma_depth && scene_depth && ws, "null argument");
  MH_CHECK(H > 0 && W > 0, "empty image");
  MH_CHECK(H > 4 && W > 4, "image smaller than the bilateral support");
  MH_CHECK(fillin_ksize > 1 && fillin_ksize <= 11, "fill-in window must be 2..11");
  const int P = H * W;
  SceneWs s = scene_carve(ws, P);
  hipStream_t st = (hipStream_t)stream;
  const dim3 g256((P + 255) / 256), b256(256);
  // the filtered depth is built in the caller's output buffer and filled there in place (no copy at the end)
  MH_CHECK(scene_depth != ma_depth, "scene_depth must not alias ma_depth");
  if (use_bilateral) {
    hipLaunchKernelGGL(k_scene_bilateral, g256, b256, 0, st, H, W, ma_depth, scene_depth);
    MH_LAUNCH_CHECK();
  } else {
    MH_HIP(hipMemcpyAsync(scene_depth, ma_depth, (size_t)P * 4, hipMemcpyDeviceToDevice, st));
  }
  double* part = (double*)s.grad;          // [workgroups of the Sobel kernel][4] (32 bytes per 256 pixels of a 4 P-byte array)
  hipLaunchKernelGGL(k_scene_sobel, g256, b256, 0, st, H, W, (const float*)scene_depth, s.g_disp, s.g_depth, part);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scene_edges2, g256, b256, 0, st, H, W, (const float*)s.g_disp, (const float*)s.g_depth, ma_mask, (const double*)part,
                     (int)g256.x, s.dmask);
  MH_LAUNCH_CHECK();
  MH_CHECK(scene_fill_launch(H, W, fillin_ksize, 0, scene_depth, s.dmask, s.list_a, s.list_b, s.upd, st) == 0, "fill-in window must be 2..11");
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_scene_fill(int H, int W, int ksize, int truncate, float* values, float* mask, void* ws, void* stream) {
  MH_CHECK(values && mask && ws, "null argument");
  MH_CHECK(H > 0 && W > 0, "empty image");
  MH_CHECK(ksize > 1 && ksize <= 11, "fill-in window must be 2..11");
  SceneWs s = scene_carve(ws, H * W);
  MH_CHECK(scene_fill_launch(H, W, ksize, truncate, values, mask, s.list_a, s.list_b, s.upd, (hipStream_t)stream) == 0,
           "fill-in window must be 2..11");
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_scene_points(int H, int W, const float* K_host, const float* scene_depth, const float* mask, float* points,
                               int* count_dev, void* stream) {
  MH_CHECK(K_host && scene_depth && mask && points && count_dev, "null argument");
  MH_CHECK(H > 0 && W > 0, "empty image");
  const double a = K_host[0], b = K_host[3], c = K_host[1], d = K_host[4];   // M = K[:2,:2]^T = [[a,b],[c,d]]
  const double det = a * d - b * c;
  MH_CHECK(det != 0.0, "singular intrinsics");
  const float i00 = (float)(d / det), i01 = (float)(-b / det), i10 = (float)(-c / det), i11 = (float)(a / det);
  hipLaunchKernelGGL(k_scene_points, dim3(1), dim3(1024), 0, (hipStream_t)stream, H, W, scene_depth, mask, i00, i01, i10, i11,
                     K_host[2], K_host[5], points, count_dev);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
