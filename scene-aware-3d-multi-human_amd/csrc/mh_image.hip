// Image-space staging and mask statistics: instance masks as one bit-plane word per pixel,
// binary erosion, validity gates, occlusion ordering of the silhouette term, priors.
#include "mh_common.h"

// ---------------------------------------------------------------------------------------------
// staging: (T,N,H,W) float {0,1} masks -> bits[t][p] (bit n = person n), area[t][n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_masks(const float* seg, int N, size_t P, uint32_t* bits, float* area) {
  const int t = blockIdx.y;
  __shared__ float s[32][4];
  float cnt[32];
  for (int n = 0; n < 32; ++n) cnt[n] = 0.f;
  for (size_t p = blockIdx.x * (size_t)256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
    uint32_t w = 0;
    for (int n = 0; n < N; ++n) {
      const float v = seg[((size_t)t * N + n) * P + p];
      if (v >= 0.5f) {
        w |= 1u << n;
        cnt[n] += 1.f;
      }
    }
    bits[(size_t)t * P + p] = w;
  }
  const int wave = threadIdx.x >> 6;
  for (int n = 0; n < N; ++n) {
    const float c = mh_wave_sum(cnt[n]);
    if ((threadIdx.x & 63) == 0) s[n][wave] = c;
  }
  __syncthreads();
  if (threadIdx.x < N) atomicAdd(&area[(size_t)t * N + threadIdx.x], s[threadIdx.x][0] + s[threadIdx.x][1] + s[threadIdx.x][2] + s[threadIdx.x][3]);
}

extern "C" int mh_pack_masks(const float* seg, int T, int N, int H, int W, uint32_t* bits, float* area, void* stream) {
  MH_CHECK(seg && bits && area, "null argument");
  MH_CHECK(T > 0 && N > 0 && N <= 32 && H > 0 && W > 0, "need 1..32 people per frame");
  hipStream_t st = (hipStream_t)stream;
  MH_HIP(hipMemsetAsync(area, 0, (size_t)T * N * sizeof(float), st));   // pixel counts are integers: order-free
  const size_t P = (size_t)H * W;
  int bx = (int)((P + 255) / 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_pack_masks, dim3(bx, T), dim3(256), 0, st, seg, N, P, bits, area);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// one Erode2D(kernel_size=3) on every bit plane (morphology.py:29-31): a pixel survives when none
// of its in-image 3x3 neighbours is background
__global__ __launch_bounds__(256) void k_erode_bits(const uint32_t* in, uint32_t* out, int H, int W) {
  const int t = blockIdx.y;
  const uint32_t* src = in + (size_t)t * H * W;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
    const int y = p / W, x = p % W;
    uint32_t w = 0xffffffffu;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) w &= src[yy * W + xx];
      }
    out[(size_t)t * H * W + p] = w;
  }
}

extern "C" int mh_erode_bits(const uint32_t* in, uint32_t* out, int T, int H, int W, void* stream) {
  MH_CHECK(in && out && in != out, "null or aliased argument");
  MH_CHECK(T > 0 && H > 0 && W > 0, "empty input");
  int bx = (H * W + 255) / 256;
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(k_erode_bits, dim3(bx, T), dim3(256), 0, (hipStream_t)stream, in, out, H, W);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// validity gates (optimizer.py:404-409)
__global__ void k_gates(const float* pose2d, const float* area, int B, float thr, float min_area, float* p2d_valid,
                        float* mask_valid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int c = 0;
  for (int j = 0; j < MH_NKP; ++j) c += pose2d[((size_t)b * MH_NKP + j) * 3 + 2] >= thr;
  p2d_valid[b] = c >= 2 ? 1.f : 0.f;
  mask_valid[b] = area[b] >= min_area ? 1.f : 0.f;
}

extern "C" int mh_stage_gates(const float* pose2d, const float* area, int B, float thr, float min_area,
                              float* pose2d_valid, float* mask_valid, void* stream) {
  MH_CHECK(pose2d && area && pose2d_valid && mask_valid, "null argument");
  MH_CHECK(B > 0, "B must be positive");
  hipLaunchKernelGGL(k_gates, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose2d, area, B, thr, min_area,
                     pose2d_valid, mask_valid);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------
// silhouette term, mask-only part (optimizer.py:450-477): near->far ordering by poses_T.z,
// front[t][n] = bit set of the people in front of n, apply[t][n] (gate indexed by RANK, as the
// reference does), D = sum(1-acc), S = sum((1-acc)*seg_n) over the whole image.
// ---------------------------------------------------------------------------------------------
template <int NMAX>
__global__ __launch_bounds__(256) void k_sil_stats(const uint32_t* bits, int N, int P, const float* pT,
                                                   const float* p2d_valid, const float* mask_valid, uint32_t* front,
                                                   float* apply, float* D, float* S) {
  const int t = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  __shared__ uint32_t sfront[32];
  __shared__ float sred[2][32][4];
  if (threadIdx.x < N) {
    const int n = threadIdx.x;
    const float z = pT[((size_t)t * N + n) * 3 + 2];
    uint32_t f = 0;
    int rank = 0;
    for (int m = 0; m < N; ++m) {
      const float zm = pT[((size_t)t * N + m) * 3 + 2];
      if (zm < z || (zm == z && m < n)) {
        f |= 1u << m;
        ++rank;
      }
    }
    sfront[n] = f;
    if (part == 0) {
      front[(size_t)t * N + n] = f;
      apply[(size_t)t * N + n] = mask_valid[(size_t)t * N + rank] * p2d_valid[(size_t)t * N + rank];   // optimizer.py:472 (sic)
    }
  }
  __syncthreads();
  float d[NMAX], s[NMAX];
  uint32_t fr[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    d[n] = s[n] = 0.f;
    fr[n] = n < N ? sfront[n] : 0xffffffffu;
  }
  const int p_lo = (int)((long long)P * part / nparts), p_hi = (int)((long long)P * (part + 1) / nparts);
  for (int p = p_lo + threadIdx.x; p < p_hi; p += 256) {
    const uint32_t w = bits[(size_t)t * P + p];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      const bool free_px = (w & fr[n]) == 0;
      d[n] += free_px ? 1.f : 0.f;
      s[n] += (free_px && ((w >> n) & 1u)) ? 1.f : 0.f;
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    const float a = mh_wave_sum(d[n]), b = mh_wave_sum(s[n]);
    if ((threadIdx.x & 63) == 0 && n < N) {
      sred[0][n][wave] = a;
      sred[1][n][wave] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int n = threadIdx.x;
    // pixel counts: integers below 2^24, so the float atomics are exact and order-free
    atomicAdd(&D[(size_t)t * N + n], sred[0][n][0] + sred[0][n][1] + sred[0][n][2] + sred[0][n][3]);
    atomicAdd(&S[(size_t)t * N + n], sred[1][n][0] + sred[1][n][1] + sred[1][n][2] + sred[1][n][3]);
  }
}

extern "C" int mh_sil_mask_stats(const uint32_t* bits, int T, int N, int H, int W, const float* pT,
                                 const float* pose2d_valid, const float* mask_valid, uint32_t* front, float* apply,
                                 float* D, float* S, void* stream) {
  MH_CHECK(bits && pT && pose2d_valid && mask_valid && front && apply && D && S, "null argument");
  MH_CHECK(T > 0 && N > 0 && N <= 32, "need 1..32 people per frame");
  const int parts = T >= 2048 ? 1 : (T >= 512 ? 4 : 8);
  MH_HIP(hipMemsetAsync(D, 0, (size_t)T * N * sizeof(float), (hipStream_t)stream));
  MH_HIP(hipMemsetAsync(S, 0, (size_t)T * N * sizeof(float), (hipStream_t)stream));
#define SIL_LAUNCH(NM) hipLaunchKernelGGL(k_sil_stats<NM>, dim3(T, parts), dim3(256), 0, (hipStream_t)stream, bits, N, \
                                          H * W, pT, pose2d_valid, mask_valid, front, apply, D, S)
  if (N <= 4) SIL_LAUNCH(4);
  else if (N <= 8) SIL_LAUNCH(8);
  else if (N <= 16) SIL_LAUNCH(16);
  else SIL_LAUNCH(32);
#undef SIL_LAUNCH
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// The same with the pixel pass taken only when it can change anything (round 4).  D and S are functions of the (constant)
// instance masks and of WHO IS IN FRONT OF WHOM in the frame; the optimiser changes that ordering a handful of times in a
// fit.  One workgroup per frame: the front sets are computed from the current translations and compared with the ones the
// frame's counts were taken for (tag[t] != 0: they exist) -- equal: only the gates are refreshed; different: the frame's
// pixels are counted again and D / S overwritten (no memsets, no atomics: per-cycle cost 2 launches of 5 us and a 17-us
// kernel less, all of which ran beside -- and were paid for by -- the selection kernel).
template <int NMAX>
__global__ __launch_bounds__(256) void k_sil_stats_cached(const uint32_t* bits, int N, int P, const float* pT,
                                                          const float* p2d_valid, const float* mask_valid, uint32_t* front,
                                                          float* apply, float* D, float* S, int* tag) {
  const int t = blockIdx.x;
  __shared__ uint32_t sfront[32];
  __shared__ float sred[2][32][4];
  __shared__ int s_changed;
  if (threadIdx.x == 0) s_changed = tag[t] == 0 ? 1 : 0;
  __syncthreads();
  if (threadIdx.x < N) {
    const int n = threadIdx.x;
    const float z = pT[((size_t)t * N + n) * 3 + 2];
    uint32_t f = 0;
    int rank = 0;
    for (int m = 0; m < N; ++m) {
      const float zm = pT[((size_t)t * N + m) * 3 + 2];
      if (zm < z || (zm == z && m < n)) {
        f |= 1u << m;
        ++rank;
      }
    }
    sfront[n] = f;
    if (front[(size_t)t * N + n] != f) s_changed = 1;
    front[(size_t)t * N + n] = f;
    apply[(size_t)t * N + n] = mask_valid[(size_t)t * N + rank] * p2d_valid[(size_t)t * N + rank];   // optimizer.py:472 (sic)
  }
  __syncthreads();
  if (!s_changed) return;
  float d[NMAX], s[NMAX];
  uint32_t fr[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    d[n] = s[n] = 0.f;
    fr[n] = n < N ? sfront[n] : 0xffffffffu;
  }
  for (int p = threadIdx.x; p < P; p += 256) {
    const uint32_t w = bits[(size_t)t * P + p];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      const bool free_px = (w & fr[n]) == 0;
      d[n] += free_px ? 1.f : 0.f;
      s[n] += (free_px && ((w >> n) & 1u)) ? 1.f : 0.f;
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    const float a = mh_wave_sum(d[n]), b = mh_wave_sum(s[n]);     // pixel counts: integers below 2^24, exact in any order
    if ((threadIdx.x & 63) == 0 && n < N) {
      sred[0][n][wave] = a;
      sred[1][n][wave] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int n = threadIdx.x;
    D[(size_t)t * N + n] = sred[0][n][0] + sred[0][n][1] + sred[0][n][2] + sred[0][n][3];
    S[(size_t)t * N + n] = sred[1][n][0] + sred[1][n][1] + sred[1][n][2] + sred[1][n][3];
  }
  if (threadIdx.x == 0) tag[t] = 1;
}

extern "C" int mh_sil_mask_stats_cached(const uint32_t* bits, int T, int N, int H, int W, const float* pT,
                                        const float* pose2d_valid, const float* mask_valid, uint32_t* front, float* apply,
                                        float* D, float* S, int32_t* tag, void* stream) {
  MH_CHECK(bits && pT && pose2d_valid && mask_valid && front && apply && D && S && tag, "null argument");
  MH_CHECK(T > 0 && N > 0 && N <= 32, "need 1..32 people per frame");
#define SIL_LAUNCH(NM) hipLaunchKernelGGL(k_sil_stats_cached<NM>, dim3(T), dim3(256), 0, (hipStream_t)stream, bits, N, \
                                          H * W, pT, pose2d_valid, mask_valid, front, apply, D, S, tag)
  if (N <= 4) SIL_LAUNCH(4);
  else if (N <= 8) SIL_LAUNCH(8);
  else if (N <= 16) SIL_LAUNCH(16);
  else SIL_LAUNCH(32);
#undef SIL_LAUNCH
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------
// priors (optimizer.py:523-532, 535-542)
// ---------------------------------------------------------------------------------------------
struct PriorP {
  int T, N, nbatches;
  const float* poses;
  const float* ref;
  const float* valid;
  const float* betas;
  const float* betas_ref;
  const float* xscale;
  float cp, cs;
  float* gposes;
  float* gbetas;
  float* gxscale;
  float* body_loss;   // [B] per-body pose-prior partial
  float* loss3;       // [3]: T*L1(beta), scale_avg, scale_person
};

__global__ __launch_bounds__(128) void k_priors(PriorP p) {
  const int B = p.T * p.N;
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    const float v = p.valid[b];
    float l = 0.f;
    if (threadIdx.x < 72) {
      const size_t o = (size_t)b * 72 + threadIdx.x;
      const float d = v * p.ref[o] - v * p.poses[o];             // L1(valid*ref, valid*pose), :523-525
      l = fabsf(d);
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      p.gposes[o] += p.cp * (-v * sg);
    }
    l = mh_wave_sum(l);
    __shared__ float s2[2];
    if ((threadIdx.x & 63) == 0) s2[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) p.body_loss[b] = s2[0] + s2[1];
    return;
  }
  // last block: shape and scale priors (shared leaves)
  __shared__ float sb[128];
  float l = 0.f;
  for (int i = threadIdx.x; i < p.N * MH_NUM_BETAS; i += 128) {
    const float d = p.betas[i] - p.betas_ref[i];
    l += fabsf(d);
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (p.gbetas) p.gbetas[i] += p.cp * (float)p.T * sg;           // sum over batches of batch_size = T (:526)
  }
  sb[threadIdx.x] = l;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (threadIdx.x < o) sb[threadIdx.x] += sb[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float sum = 0.f, sq = 0.f;
    for (int n = 0; n < p.N; ++n) {
      const float s = powf(1.1f, p.xscale ? p.xscale[n] : 0.f) - 1.f;
      sum += s;
      sq += s * s;
    }
    p.loss3[0] = (float)p.T * sb[0];
    p.loss3[1] = sum * sum;            // reg_scale_avg   (:531)
    p.loss3[2] = sq / (float)p.N;      // reg_scale_person (:532)
    if (p.gxscale && p.xscale)
      for (int n = 0; n < p.N; ++n) {
        const float s = powf(1.1f, p.xscale[n]);
        const float ds = s * 0.0953101798043249f;
        const float g = p.cs * 2.f * (s - 1.f) / (float)p.N + (p.cs > 0.f ? 1.f : 0.f) * 2.f * sum;   // :539
        p.gxscale[n] += (float)p.nbatches * g * ds;
      }
  }
}

extern "C" int mh_prior_terms(int T, int N, int nbatches, const float* poses, const float* poses_ref,
                              const float* valid, const float* betas, const float* betas_ref, const float* xscale,
                              float coef_poses, float coef_scales, float* gposes, float* gbetas, float* gxscale,
                              float* body_loss, float* loss3, void* stream) {
  MH_CHECK(poses && poses_ref && valid && betas && betas_ref && gposes && body_loss && loss3, "null argument");
  MH_CHECK(T > 0 && N > 0 && nbatches > 0, "empty input");
  PriorP p;
  p.T = T; p.N = N; p.nbatches = nbatches;
  p.poses = poses; p.ref = poses_ref; p.valid = valid; p.betas = betas; p.betas_ref = betas_ref; p.xscale = xscale;
  p.cp = coef_poses; p.cs = coef_scales;
  p.gposes = gposes; p.gbetas = gbetas; p.gxscale = gxscale; p.body_loss = body_loss; p.loss3 = loss3;
  hipLaunchKernelGGL(k_priors, dim3(T * N + 1), dim3(128), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// deterministic sum of n floats, scaled: out[0] = scale * sum(x)
__global__ __launch_bounds__(256) void k_reduce_sum(const float* x, size_t n, float scale, float* out, const float* x1, float* out1) {
  __shared__ float s[256];
  if (blockIdx.x == 1) { x = x1; out = out1; }        // second array of mh_reduce_sum2: same order, same launch
  float a = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) a += x[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * s[0];
}

// several small sums in ONE launch (one workgroup each, fixed summation order): the log entries of a cycle were five
// single-workgroup launches of their own
struct MultiSumP {
  int n;
  const float* x[8];
  unsigned long long len[8];
  float* out[8];
};
__global__ __launch_bounds__(256) void k_reduce_sum_multi(MultiSumP p) {
  __shared__ float s[256];
  const int k = blockIdx.x;
  float a = 0.f;
  for (size_t i = threadIdx.x; i < (size_t)p.len[k]; i += 256) a += p.x[k][i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[k][0] = s[0];
}

extern "C" int mh_reduce_sum_multi(int n, const float* const* xs, const size_t* lens, float* const* outs, void* stream) {
  MH_CHECK(xs && lens && outs, "null argument");
  MH_CHECK(n >= 1 && n <= 8, "1..8 sums per launch");
  MultiSumP p;
  p.n = n;
  for (int k = 0; k < n; ++k) {
    MH_CHECK(xs[k] && outs[k], "null argument");
    p.x[k] = xs[k]; p.len[k] = lens[k]; p.out[k] = outs[k];
  }
  hipLaunchKernelGGL(k_reduce_sum_multi, dim3(n), dim3(256), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_reduce_sum(const float* x, size_t n, float scale, float* out, void* stream) {
  MH_CHECK(x && out, "null argument");
  hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, scale, out, (const float*)nullptr, (float*)nullptr);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_reduce_sum2(const float* x0, const float* x1, size_t n, float scale, float* out0, float* out1, void* stream) {
  MH_CHECK(x0 && x1 && out0 && out1, "null argument");
  hipLaunchKernelGGL(k_reduce_sum, dim3(2), dim3(256), 0, (hipStream_t)stream, x0, n, scale, out0, x1, out1);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// =============================================================================================
// stand-alone forms of the two loss builders of losses.py and of morphology.py (call compatibility
// for users of mhmocap.losses / mhmocap.morphology; the optimiser itself uses the fused kernels)
// =============================================================================================
// losses.py:19-30: rows r = (frame, person); true rows are shared by `group` consecutive rows
__global__ __launch_bounds__(256) void k_depth_loss_rows(const float* pred, const float* tru, const float* mask, size_t P,
                                                         int group, float eps, float* sums /*[R][3]*/) {
  __shared__ float s[3][4];
  const int r = blockIdx.x;
  const float* pr = pred + (size_t)r * P;
  const float* tr = tru + (size_t)(r / group) * P;
  const float* mk = mask + (size_t)r * P;
  float a = 0.f, b = 0.f, c = 0.f;
  for (size_t i = threadIdx.x; i < P; i += 256) {
    const float m = mk[i];
    a += m * logf(fmaxf(pr[i], eps));
    b += m * logf(fmaxf(tr[i], eps));
    c += m;
  }
  a = mh_wave_sum(a); b = mh_wave_sum(b); c = mh_wave_sum(c);
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = a; s[1][threadIdx.x >> 6] = b; s[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x < 3) sums[(size_t)r * 3 + threadIdx.x] = s[threadIdx.x][0] + s[threadIdx.x][1] + s[threadIdx.x][2] + s[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void k_depth_loss_grads(const float* pred, const float* tru, const float* mask, size_t P,
                                                          int group, float eps, const float* sums, float gout, float* gpred,
                                                          float* gtrue_rows) {
  const int r = blockIdx.x;
  const float cnt = sums[(size_t)r * 3 + 2] + 1.f;
  const float diff = sums[(size_t)r * 3] / cnt - sums[(size_t)r * 3 + 1] / cnt;
  const float g = gout * 2.f * diff / cnt;
  const float* pr = pred + (size_t)r * P;
  const float* tr = tru + (size_t)(r / group) * P;
  const float* mk = mask + (size_t)r * P;
  for (size_t i = threadIdx.x; i < P; i += 256) {
    const float m = mk[i];
    if (gpred) gpred[(size_t)r * P + i] = pr[i] >= eps ? g * m / pr[i] : 0.f;
    if (gtrue_rows) gtrue_rows[(size_t)r * P + i] = tr[i] >= eps ? -g * m / tr[i] : 0.f;
  }
}

extern "C" int mh_avg_depth_loss(const float* pred, const float* tru, const float* mask, int rows, int group, size_t P,
                                 float eps, float* row_sums, float* row_loss, void* stream) {
  MH_CHECK(pred && tru && mask && row_sums && row_loss, "null argument");
  MH_CHECK(rows > 0 && group > 0 && P > 0, "empty input");
  hipLaunchKernelGGL(k_depth_loss_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, tru, mask, P, group, eps, row_sums);
  MH_LAUNCH_CHECK();
  (void)row_loss;
  return MH_OK;
}

extern "C" int mh_avg_depth_loss_backward(const float* pred, const float* tru, const float* mask, int rows, int group,
                                          size_t P, float eps, const float* row_sums, float grad_out, float* gpred,
                                          float* gtrue_rows, void* stream) {
  MH_CHECK(pred && tru && mask && row_sums, "null argument");
  MH_CHECK(rows > 0 && group > 0 && P > 0, "empty input");
  hipLaunchKernelGGL(k_depth_loss_grads, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, tru, mask, P, group, eps,
                     row_sums, grad_out, gpred, gtrue_rows);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// losses.py:33-40: sums[0] = sum((mask*(a-b))^2), sums[1] = sum(mask) (single block, fixed order)
__global__ __launch_bounds__(1024) void k_mse_sums(const float* a, const float* b, const float* mask, size_t n, float* sums) {
  __shared__ float s[2][16];
  float x = 0.f, c = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const float m = mask[i], d = m * (a[i] - b[i]);
    x += d * d;
    c += m;
  }
  x = mh_wave_sum(x); c = mh_wave_sum(c);
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = x; s[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += s[threadIdx.x][w];
    sums[threadIdx.x] = t;
  }
}
__global__ void k_mse_grad(const float* a, const float* b, const float* mask, size_t n, const float* sums, float gout, float* ga) {
  const float inv = gout * 2.f / (sums[1] + 1.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float m = mask[i];
    ga[i] = inv * m * m * (a[i] - b[i]);
  }
}

extern "C" int mh_masked_mse(const float* a, const float* b, const float* mask, size_t n, float* sums2, void* stream) {
  MH_CHECK(a && b && mask && sums2, "null argument");
  MH_CHECK(n > 0, "empty input");
  hipLaunchKernelGGL(k_mse_sums, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, mask, n, sums2);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
extern "C" int mh_masked_mse_backward(const float* a, const float* b, const float* mask, size_t n, const float* sums2,
                                      float grad_out, float* ga, void* stream) {
  MH_CHECK(a && b && mask && sums2 && ga, "null argument");
  const size_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_mse_grad, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, (hipStream_t)stream, a, b, mask, n, sums2,
                     grad_out, ga);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// morphology.py:21-34 on float maps: erode = 1 - clamp(sum over the k x k window of (x < .5)), dilate =
// clamp(sum of (x >= .5)); zero padding outside the image
__global__ void k_morph_f32(const float* in, float* out, int n_img, int H, int W, int k, int dilate) {
  const size_t n = (size_t)n_img * H * W;
  const int r = k / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float* img = in + (i / ((size_t)H * W)) * (size_t)H * W;
    bool hit = false;
    for (int dy = -r; dy <= r; ++dy)
      for (int dx = -r; dx <= r; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float v = img[(size_t)yy * W + xx];
        hit |= dilate ? (v >= 0.5f) : (v < 0.5f);
      }
    out[i] = dilate ? (hit ? 1.f : 0.f) : (hit ? 0.f : 1.f);
  }
}
extern "C" int mh_morph_f32(const float* in, float* out, int n_images, int H, int W, int kernel_size, int dilate, void* stream) {
  MH_CHECK(in && out && in != out, "null or aliased argument");
  MH_CHECK(n_images > 0 && H > 0 && W > 0 && kernel_size > 0 && (kernel_size & 1), "bad shape / even kernel");
  const size_t nb = ((size_t)n_images * H * W + 255) / 256;
  hipLaunchKernelGGL(k_morph_f32, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, in, out, n_images,
                     H, W, kernel_size, dilate);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
