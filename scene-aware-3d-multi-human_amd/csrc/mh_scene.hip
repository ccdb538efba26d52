// Scene terms: lowest vertex per body, k-nearest scene points (k=32 partial selection instead
// of the reference's full argsort), contact and foot-sliding residuals, scene unprojection.
// Reference: optimizer.py:485-518 (contact / foot sliding), :605-616 + transforms.py:114-130.
#include "mh_common.h"

#include <math.h>

// ---------------------------------------------------------------------------------------------
// lowest vertex = argmax of y (y points down), first index on ties (optimizer.py:487-489)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lowest_vertex(const float* verts, int V, int* low_idx, float* low_xyz) {
  const int b = blockIdx.x;
  const float* vb = verts + (size_t)b * V * 3;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float y = vb[(size_t)v * 3 + 1];
    if (y > best) {   // strictly greater: keeps the first index of this thread's stride
      best = y;
      bi = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float oy = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oy > best || (oy == best && oi < bi)) {
      best = oy;
      bi = oi;
    }
  }
  __shared__ float sy[4];
  __shared__ int si[4];
  if ((threadIdx.x & 63) == 0) {
    sy[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sy[w] > best || (sy[w] == best && si[w] < bi)) {
        best = sy[w];
        bi = si[w];
      }
    low_idx[b] = bi;
    low_xyz[(size_t)b * 3] = vb[(size_t)bi * 3];
    low_xyz[(size_t)b * 3 + 1] = vb[(size_t)bi * 3 + 1];
    low_xyz[(size_t)b * 3 + 2] = vb[(size_t)bi * 3 + 2];
  }
}

extern "C" int mh_lowest_vertex(const float* verts, int B, int V, int32_t* low_idx, float* low_xyz, void* stream) {
  MH_CHECK(verts && low_idx && low_xyz, "null argument");
  MH_CHECK(B > 0 && V > 0, "empty input");
  hipLaunchKernelGGL(k_lowest_vertex, dim3(B), dim3(256), 0, (hipStream_t)stream, verts, V, low_idx, low_xyz);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------
// k nearest scene points of one query per wave.  A wave keeps the best K squared distances seen
// so far (K <= 32) in LDS together with the y of those points; a chunk of 64 points only costs
// a compare unless one of them beats the current K-th distance.
// ---------------------------------------------------------------------------------------------
#define KNN_CAP 128   // 32 kept + up to 64 new, padded to a power of two for the bitonic network
#define KNN_WAVES 4   // waves (point ranges) per query

// ascending bitonic sort of 128 (key d, payload y) pairs by ONE wave: 2 elements per lane.  The list is
// private to the wave; LDS executes a wave's instructions in order, so no workgroup barrier is needed
// (volatile keeps the compiler from caching list elements in registers).
__device__ __forceinline__ void knn_sort128(volatile float* d, volatile float* y, int lane) {
  for (int k = 2; k <= KNN_CAP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = (i & k) == 0;
          const float a = d[i], b = d[ixj];
          if ((a > b) == up) {
            d[i] = b;
            d[ixj] = a;
            const float t = y[i];
            y[i] = y[ixj];
            y[ixj] = t;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ __launch_bounds__(64 * KNN_WAVES) void k_contact_knn(const float* pts, int M, const float* low_xyz, int K, float* dy) {
  __shared__ float sd_all[KNN_WAVES][KNN_CAP];
  __shared__ float sy_all[KNN_WAVES][KNN_CAP];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  volatile float* sd = sd_all[wave];
  volatile float* sy = sy_all[wave];
  const float qx = low_xyz[(size_t)b * 3], qy = low_xyz[(size_t)b * 3 + 1], qz = low_xyz[(size_t)b * 3 + 2];
  sd[lane] = INFINITY;
  sd[lane + 64] = INFINITY;
  sy[lane] = 0.f;
  sy[lane + 64] = 0.f;
  __builtin_amdgcn_wave_barrier();
  float tau = INFINITY;   // current K-th best distance of this wave's range (wave-uniform)
  int fill = K;           // slots [0,K) hold the kept set (INF until filled), new candidates go to [K, ...)
  const int per = (M + KNN_WAVES - 1) / KNN_WAVES;
  const int m0 = wave * per, m1 = min(M, m0 + per);
  // Scene points arrive in raster order (distance to the query varies monotonically over long runs),
  // which would tighten tau only slowly and trigger a sort per chunk.  Visiting the 64-point chunks in
  // a strided pseudo-random order makes the first few chunks a sample of the whole range.
  const int nchunks = (max(m1 - m0, 0) + 63) / 64;
  int stride = (int)(0.618f * (float)nchunks) | 1;
  while (stride > 1) {
    int a = stride, c = nchunks;
    while (c) { const int t = a % c; a = c; c = t; }
    if (a == 1) break;
    stride -= 2;
  }
  if (stride < 1) stride = 1;
  for (int cidx = 0, cc = 0; cidx < nchunks; ++cidx, cc = (cc + stride) % nchunks) {
    const int base = m0 + cc * 64;
    const int i = base + lane;
    float d2 = INFINITY, py = 0.f;
    if (i < m1) {
      const float dx = pts[(size_t)i * 3] - qx;
      py = pts[(size_t)i * 3 + 1];
      const float dyy = py - qy, dz = pts[(size_t)i * 3 + 2] - qz;
      d2 = dx * dx + dyy * dyy + dz * dz;       // optimizer.py:492
    }
    const bool take = d2 < tau;
    const unsigned long long m = __ballot(take);
    if (m == 0ull) continue;
    if (take) {
      const int pos = fill + __popcll(m & ((1ull << lane) - 1ull));
      sd[pos] = d2;
      sy[pos] = py;
    }
    fill += __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (fill > KNN_CAP - 64) {   // not enough room for another full chunk: keep the best K
      knn_sort128(sd, sy, lane);
      tau = sd[K - 1];
      if (lane + K < KNN_CAP) sd[lane + K] = INFINITY;
      if (lane + 64 + K < KNN_CAP) sd[lane + 64 + K] = INFINITY;
      fill = K;
      __builtin_amdgcn_wave_barrier();
    }
  }
  knn_sort128(sd, sy, lane);
  __syncthreads();
  if (wave == 0) {
    // merge the K best of every range: 4 x 32 = 128 entries, one more sort
    float md[2], my[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = lane + 64 * h, w = e / 32, r = e % 32;
      const bool ok = w < KNN_WAVES && r < K;
      md[h] = ok ? sd_all[w][r] : INFINITY;
      my[h] = ok ? sy_all[w][r] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    sd[lane] = md[0]; sd[lane + 64] = md[1];
    sy[lane] = my[0]; sy[lane + 64] = my[1];
    __builtin_amdgcn_wave_barrier();
    knn_sort128(sd, sy, lane);
    const int kk = M < K ? M : K;
    float s = (lane < kk) ? sy[lane] : 0.f;
    s = mh_wave_sum(s);
    if (lane == 0) dy[b] = s / (float)kk - qy;     // (mean of the nearest points - lowest vertex).y, :500-502
  }
}

extern "C" int mh_contact_knn(const float* points, int M, const float* low_xyz, int B, int k, float* dy, void* stream) {
  MH_CHECK(points && low_xyz && dy, "null argument");
  MH_CHECK(M > 0 && B > 0, "empty input");
  MH_CHECK(k >= 1 && k <= 32, "k must be in 1..32");
  hipLaunchKernelGGL(k_contact_knn, dim3(B), dim3(64 * KNN_WAVES), 0, (hipStream_t)stream, points, M, low_xyz, k, dy);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------
// contact residual + foot sliding (optimizer.py:502-518).  One block per batch of `batch` frames.
// ---------------------------------------------------------------------------------------------
struct ContactP {
  int T, N, V, batch;
  const float* verts;
  const int* low_idx;
  const float* low_xyz;
  const float* dy;
  float cc, cf;
  float* gpT;
  float* gverts;
  float* batch_contact;   // [nbatches]
  float* batch_foot;      // [nbatches]
};

__global__ __launch_bounds__(256) void k_contact_foot(ContactP p) {
  const int bt = blockIdx.x;
  const int t0 = bt * p.batch, t1 = min(t0 + p.batch, p.T);
  __shared__ float s[256];
  __shared__ float scnt;
  // contact: L1(pT, detach(pT) + [0, dy + 0.02, 0])  ->  |dy + 0.02|, gradient -sign on pT.y only
  float lc = 0.f, cnt = 0.f;
  for (int i = t0 * p.N + threadIdx.x; i < t1 * p.N; i += 256) {
    const float r = -(p.dy[i] + 0.02f);
    lc += fabsf(r);
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (p.gpT) p.gpT[(size_t)i * 3 + 1] += p.cc * sg;
    if (i >= (t0 + 1) * p.N) cnt += (p.dy[i] > -0.20f) ? 1.f : 0.f;   // gate of the in-batch pairs, :510-513
  }
  s[threadIdx.x] = lc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.batch_contact[bt] = s[0];
  __syncthreads();
  s[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scnt = fmaxf(s[0], 1.f);     // clamp(sum(gate), 1), :518
  __syncthreads();
  const float inv = 1.f / scnt;
  float lf = 0.f;
  for (int i = (t0 + 1) * p.N + threadIdx.x; i < t1 * p.N; i += 256) {
    if (!(p.dy[i] > -0.20f)) continue;
    const int vi = p.low_idx[i];
    const size_t cur = ((size_t)i * p.V + vi) * 3, prv = ((size_t)(i - p.N) * p.V + vi) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = p.low_xyz[(size_t)i * 3 + c] - p.verts[prv + c];   // :514-517
      lf += fabsf(d);
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      if (p.gverts) {
        atomicAdd(&p.gverts[cur + c], p.cf * sg * inv);
        atomicAdd(&p.gverts[prv + c], -p.cf * sg * inv);
      }
    }
  }
  s[threadIdx.x] = lf;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.batch_foot[bt] = s[0] * inv;
}

extern "C" int mh_contact_foot_terms(int T, int N, int V, int batch, const float* verts, const int32_t* low_idx,
                                     const float* low_xyz, const float* dy, float coef_contact, float coef_foot,
                                     float* gpT, float* gverts, float* batch_contact, float* batch_foot,
                                     void* stream) {
  MH_CHECK(verts && low_idx && low_xyz && dy && batch_contact && batch_foot, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && batch > 0, "empty input");
  ContactP p;
  p.T = T; p.N = N; p.V = V; p.batch = batch;
  p.verts = verts; p.low_idx = low_idx; p.low_xyz = low_xyz; p.dy = dy;
  p.cc = coef_contact; p.cf = coef_foot; p.gpT = gpT; p.gverts = gverts;
  p.batch_contact = batch_contact; p.batch_foot = batch_foot;
  hipLaunchKernelGGL(k_contact_foot, dim3((T + batch - 1) / batch), dim3(256), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------
// scene unprojection (optimizer.py:605-612, transforms.py:114-130): pixel centres + depth -> xyz
// ---------------------------------------------------------------------------------------------
__global__ void k_unproject(const float* depth, int H, int W, float i00, float i01, float i10, float i11, float cx,
                            float cy, float* pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const float u = (float)(p % W) + 0.5f - cx, v = (float)(p / W) + 0.5f - cy;   // grid_xy, :312-315
  const float d = depth[p];
  // row vector [u v] times inv(K[:2,:2]^T)
  pts[(size_t)p * 3] = d * (u * i00 + v * i10);
  pts[(size_t)p * 3 + 1] = d * (u * i01 + v * i11);
  pts[(size_t)p * 3 + 2] = d;
}

extern "C" int mh_scene_unproject(const float* depth, int H, int W, const float* K_host, float* points, void* stream) {
  MH_CHECK(depth && K_host && points, "null argument");
  MH_CHECK(H > 0 && W > 0, "empty image");
  // M = K[:2,:2]^T ; inverse of the 2x2
  const double a = K_host[0], b = K_host[3], c = K_host[1], d = K_host[4];   // M = [[a,b],[c,d]]
  const double det = a * d - b * c;
  MH_CHECK(det != 0.0, "singular intrinsics");
  const float i00 = (float)(d / det), i01 = (float)(-b / det), i10 = (float)(-c / det), i11 = (float)(a / det);
  hipLaunchKernelGGL(k_unproject, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth, H, W, i00, i01,
                     i10, i11, K_host[2], K_host[5], points);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
