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

// low_idx / low_xyz from the keys the LBS forward's projection epilogue left (mh_lbs_forward_proj: order-preserving bits
// of y << 32 | ~vertex, reported by the vertices above the previous launch's lowest minus a slack; 0 = nobody reported).
// One wave per body; a body without a key is scanned like k_lowest_vertex does and its key written back, so that the next
// forward has an extreme to filter with.
// (one wave) -> the body's lowest vertex, the same in every lane
__device__ __forceinline__ int mh_lowest_from_key(const float* vb, int V, unsigned long long* lowkey, int b, int lane) {
  const unsigned long long key = lowkey[b];
  int bi = (int)~(unsigned)(key & 0xffffffffull);
  if (key == 0ull || bi < 0 || bi >= V) {       // nobody reported (or not a key at all): scan
    float best = -INFINITY;
    bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      const float y = vb[(size_t)v * 3 + 1];
      if (y > best) { best = y; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float oy = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oy > best || (oy == best && oi < bi)) { best = oy; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;                  // nothing compares (NaN everywhere): any vertex
    if (lane == 0 && best > -INFINITY) {
      const unsigned u = __float_as_uint(best + 0.0f);
      lowkey[b] = ((unsigned long long)(u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u)) << 32) | (unsigned)~(unsigned)bi;
    }
  }
  return bi;
}

__global__ __launch_bounds__(64) void k_lowest_resolve(const float* verts, int V, unsigned long long* lowkey, int* low_idx,
                                                       float* low_xyz) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* vb = verts + (size_t)b * V * 3;
  const int bi = mh_lowest_from_key(vb, V, lowkey, b, lane);
  if (lane == 0) low_idx[b] = bi;
  if (lane < 3) low_xyz[(size_t)b * 3 + lane] = vb[(size_t)bi * 3 + lane];
}

extern "C" int mh_lowest_resolve(const float* verts, int B, int V, unsigned long long* lowkey, int32_t* low_idx,
                                 float* low_xyz, void* stream) {
  MH_CHECK(verts && lowkey && low_idx && low_xyz, "null argument");
  MH_CHECK(B > 0 && V > 0, "empty input");
  hipLaunchKernelGGL(k_lowest_resolve, dim3(B), dim3(64), 0, (hipStream_t)stream, verts, V, lowkey, low_idx,
                     low_xyz);
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
// The reference pairs IN-BATCH neighbours: position k of a batch with position k-1 (:512-517).  With a sequential
// dataloader those are consecutive frames; with the shipped `shuffle: True` (configs/predict_mupots.yml:14) a batch
// is a random set of frames in random order.  `frames` (optional) is the batch table of this cycle:
// frames[bt * batch + k] = frame at position k of batch bt, -1 = empty position (a shorter last batch);
// NULL = contiguous batches.  Every frame appears once, so the per-frame writes need no atomics.
// ---------------------------------------------------------------------------------------------
struct ContactP {
  int T, N, V, batch;
  const int* frames;      // [nbatches][batch] or NULL
  const float* verts;
  const int* low_idx;
  const float* low_xyz;
  const float* dy;
  float cc, cf;
  float* gpT;
  float* gverts;
  float* batch_contact;   // [nbatches]
  float* batch_foot;      // [nbatches]
  const int* live;        // device-resident switch (mh_contact_foot_terms_gated) or NULL: 0 = no scene yet, the terms are 0
};

__global__ __launch_bounds__(256) void k_contact_foot(ContactP p) {
  const int bt = blockIdx.x;
  const int pos0 = bt * p.batch;
  __shared__ float s[256];
  __shared__ float scnt;
  if (p.live && *p.live == 0) {            // before the first scene exists (optimizer.py:485: `if scene_pcd is not None`)
    if (threadIdx.x == 0) { p.batch_contact[bt] = 0.f; p.batch_foot[bt] = 0.f; }
    return;
  }
  auto frame_at = [&](int k) -> int {      // frame at position k of this batch, -1 = none
    const int t = p.frames ? p.frames[pos0 + k] : pos0 + k;
    return t < p.T ? t : -1;
  };
  // contact: L1(pT, detach(pT) + [0, dy + 0.02, 0])  ->  |dy + 0.02|, gradient -sign on pT.y only
  float lc = 0.f, cnt = 0.f;
  for (int j = threadIdx.x; j < p.batch * p.N; j += 256) {
    const int k = j / p.N, t = frame_at(k);
    if (t < 0) continue;
    const int i = t * p.N + (j - k * p.N);
    const float r = -(p.dy[i] + 0.02f);
    lc += fabsf(r);
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (p.gpT) p.gpT[(size_t)i * 3 + 1] += p.cc * sg;
    if (k >= 1 && frame_at(k - 1) >= 0) cnt += (p.dy[i] > -0.20f) ? 1.f : 0.f;   // gate of the in-batch pairs, :510-513
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
  // Every frame sits at exactly one position of exactly one batch, so element (frame, person, vertex) of dL/dverts is
  // touched by at most two pairs of this block: as the CURRENT frame of pair (k, k-1) and as the PREVIOUS frame of pair
  // (k+1, k).  Two sweeps separated by a barrier -- all current-frame adds, then all previous-frame adds -- make every
  // add the only writer of its element in its sweep: plain read-modify-writes in a fixed order instead of float
  // atomics (whose order, hence the last bit of a + x + y, would vary from run to run).
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int j = p.N + threadIdx.x; j < p.batch * p.N; j += 256) {
      const int k = j / p.N, t = frame_at(k), tp = frame_at(k - 1);
      if (t < 0 || tp < 0) continue;
      const int n = j - k * p.N;
      const int i = t * p.N + n, ip = tp * p.N + n;
      if (!(p.dy[i] > -0.20f)) continue;
      const int vi = p.low_idx[i];
      const size_t cur = ((size_t)i * p.V + vi) * 3, prv = ((size_t)ip * p.V + vi) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = p.low_xyz[(size_t)i * 3 + c] - p.verts[prv + c];   // :514-517
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (sweep == 0) {
          lf += fabsf(d);
          if (p.gverts) p.gverts[cur + c] += p.cf * sg * inv;
        } else if (p.gverts) {
          p.gverts[prv + c] -= p.cf * sg * inv;
        }
      }
    }
    __syncthreads();
  }
  s[threadIdx.x] = lf;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.batch_foot[bt] = s[0] * inv;
}

static int contact_foot_launch(int T, int N, int V, int batch, int nbatches, const int32_t* frames, const float* verts,
                               const int32_t* low_idx, const float* low_xyz, const float* dy, float coef_contact,
                               float coef_foot, float* gpT, float* gverts, float* batch_contact, float* batch_foot,
                               void* stream, const int32_t* live = nullptr) {
  MH_CHECK(verts && low_idx && low_xyz && dy && batch_contact && batch_foot, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && batch > 0, "empty input");
  ContactP p;
  p.T = T; p.N = N; p.V = V; p.batch = batch; p.frames = frames;
  p.verts = verts; p.low_idx = low_idx; p.low_xyz = low_xyz; p.dy = dy;
  p.cc = coef_contact; p.cf = coef_foot; p.gpT = gpT; p.gverts = gverts;
  p.batch_contact = batch_contact; p.batch_foot = batch_foot; p.live = (const int*)live;
  hipLaunchKernelGGL(k_contact_foot, dim3(nbatches), dim3(256), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_contact_foot_terms(int T, int N, int V, int batch, const float* verts, const int32_t* low_idx,
                                     const float* low_xyz, const float* dy, float coef_contact, float coef_foot,
                                     float* gpT, float* gverts, float* batch_contact, float* batch_foot,
                                     void* stream) {
  MH_CHECK(T > 0 && batch > 0, "empty input");
  return contact_foot_launch(T, N, V, batch, (T + batch - 1) / batch, nullptr, verts, low_idx, low_xyz, dy, coef_contact,
                             coef_foot, gpT, gverts, batch_contact, batch_foot, stream);
}

// batch_frames may be NULL (contiguous batches: nbatches must then be ceil(T / batch)); live_dev: see ContactP::live
extern "C" int mh_contact_foot_terms_gated(int T, int N, int V, int batch, int nbatches, const int32_t* batch_frames,
                                           const float* verts, const int32_t* low_idx, const float* low_xyz, const float* dy,
                                           float coef_contact, float coef_foot, float* gpT, float* gverts,
                                           float* batch_contact, float* batch_foot, const int32_t* live_dev, void* stream) {
  MH_CHECK(live_dev, "null argument");
  MH_CHECK(T > 0 && batch > 0 && nbatches > 0, "empty input");
  MH_CHECK(batch_frames || nbatches == (T + batch - 1) / batch, "contiguous batches: nbatches must be ceil(T / batch)");
  return contact_foot_launch(T, N, V, batch, nbatches, batch_frames, verts, low_idx, low_xyz, dy, coef_contact, coef_foot,
                             gpT, gverts, batch_contact, batch_foot, stream, live_dev);
}

extern "C" int mh_contact_foot_terms_idx(int T, int N, int V, int batch, int nbatches, const int32_t* batch_frames,
                                         const float* verts, const int32_t* low_idx, const float* low_xyz, const float* dy,
                                         float coef_contact, float coef_foot, float* gpT, float* gverts,
                                         float* batch_contact, float* batch_foot, void* stream) {
  MH_CHECK(batch_frames, "null argument");
  MH_CHECK(nbatches > 0, "empty input");
  return contact_foot_launch(T, N, V, batch, nbatches, batch_frames, verts, low_idx, low_xyz, dy, coef_contact, coef_foot,
                             gpT, gverts, batch_contact, batch_foot, stream);
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

// =============================================================================================
// uniform-grid k-nearest neighbours.  All queries of a cycle share one scene cloud, so the cloud is
// bucketed once per scene update (counting sort into cells) and every query visits cells in growing
// Chebyshev shells until the k-th best distance is provably final (points of shell rho+1 are at least
// rho * cell away).  Exactly the same neighbour set as the brute-force scan, ~50x fewer distance
// evaluations for M = 2e4..2e5.
// =============================================================================================
#define GRID_MAX_CELLS (1 << 20)

struct GridHdr {        // device-resident header (written by k_grid_setup)
  float mn[3];
  float cell;
  int dim[3];
  int ncells;
  int npts;          // points in the cloud (<= the capacity the workspace was sized for)
};

__global__ __launch_bounds__(1024) void k_grid_setup(const float* pts, int M_host, const int* M_dev, GridHdr* hdr, int* counts, int max_cells) {
  __shared__ float smn[3][16], smx[3][16];
  const int M = M_dev ? *M_dev : M_host;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < M; i += 1024)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = pts[(size_t)i * 3 + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smn[c][threadIdx.x >> 6] = mn[c]; smx[c][threadIdx.x >> 6] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ext[3];
    for (int c = 0; c < 3; ++c) {
      for (int w = 1; w < 16; ++w) { mn[c] = fminf(mn[c], smn[c][w]); mx[c] = fmaxf(mx[c], smx[c][w]); }
      ext[c] = fmaxf(mx[c] - mn[c], 1e-3f);
    }
    // the clouds are surfaces: aim at ~16 points per occupied cell of the largest face of the bbox
    const float area = fmaxf(ext[0] * ext[1], fmaxf(ext[1] * ext[2], ext[0] * ext[2]));
    float cell = fminf(fmaxf(sqrtf(area * 16.f / (float)max(M, 1)), 0.02f), 4.f);
    if (M == 0) { for (int c = 0; c < 3; ++c) { mn[c] = 0.f; ext[c] = 1e-3f; } cell = 1.f; }
    int d[3];
    for (;;) {
      long long n = 1;
      for (int c = 0; c < 3; ++c) { d[c] = (int)(ext[c] / cell) + 1; n *= d[c]; }
      if (n <= max_cells) break;
      cell *= 1.26f;
    }
    hdr->cell = cell;
    for (int c = 0; c < 3; ++c) { hdr->mn[c] = mn[c]; hdr->dim[c] = d[c]; }
    hdr->ncells = d[0] * d[1] * d[2];
    hdr->npts = M;
  }
  __syncthreads();
  const int nc = hdr->ncells;
  for (int i = threadIdx.x; i <= nc; i += 1024) counts[i] = 0;
}

__device__ __forceinline__ int grid_cell(const GridHdr* h, float x, float y, float z) {
  const int cx = min(max((int)((x - h->mn[0]) / h->cell), 0), h->dim[0] - 1);
  const int cy = min(max((int)((y - h->mn[1]) / h->cell), 0), h->dim[1] - 1);
  const int cz = min(max((int)((z - h->mn[2]) / h->cell), 0), h->dim[2] - 1);
  return (cz * h->dim[1] + cy) * h->dim[0] + cx;
}

__global__ void k_grid_count(const float* pts, const GridHdr* hdr, int* counts, int* pcell) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hdr->npts) return;
  const int c = grid_cell(hdr, pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]);
  pcell[i] = c;
  atomicAdd(&counts[c], 1);
}

// exclusive scan of counts[0..ncells] in place (single block), also resets the fill cursors: DPP scan inside the
// waves, one LDS hop for the 16 wave totals
__global__ __launch_bounds__(1024) void k_grid_scan(const GridHdr* hdr, int* counts, int* cursor) {
  __shared__ int swave[16];
  __shared__ int carry;
  const int nc = hdr->ncells, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nc; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nc ? counts[i] : 0;
    const int incl = mh_wave_scan_add(v);
    if (lane == 63) swave[wave] = incl;
    __syncthreads();
    int woff = carry;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wave) ? swave[w] : 0;
    if (i < nc) {
      const int start = woff + incl - v;
      counts[i] = start;
      cursor[i] = start;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[nc] = carry;
}

__global__ void k_grid_scatter(const float* pts, const GridHdr* hdr, const int* pcell, int* cursor, float* sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hdr->npts) return;
  const int pos = atomicAdd(&cursor[pcell[i]], 1);
  sorted[(size_t)pos * 3] = pts[(size_t)i * 3];
  sorted[(size_t)pos * 3 + 1] = pts[(size_t)i * 3 + 1];
  sorted[(size_t)pos * 3 + 2] = pts[(size_t)i * 3 + 2];
}

static size_t g_align(size_t x) { return (x + 255) & ~(size_t)255; }
struct GridWs {
  GridHdr* hdr;
  int* start;     // [GRID_MAX_CELLS + 1]
  int* cursor;    // [GRID_MAX_CELLS]
  int* pcell;     // [M]
  float* sorted;  // [M][3]
};
static GridWs grid_carve(void* ws, int M) {
  char* c = (char*)ws;
  GridWs g;
  g.hdr = (GridHdr*)c; c += g_align(sizeof(GridHdr));
  g.start = (int*)c; c += g_align((size_t)(GRID_MAX_CELLS + 1) * 4);
  g.cursor = (int*)c; c += g_align((size_t)GRID_MAX_CELLS * 4);
  g.pcell = (int*)c; c += g_align((size_t)M * 4);
  g.sorted = (float*)c;
  return g;
}

extern "C" size_t mh_scene_grid_bytes(int M) {
  return g_align(sizeof(GridHdr)) + g_align((size_t)(GRID_MAX_CELLS + 1) * 4) + g_align((size_t)GRID_MAX_CELLS * 4) +
         g_align((size_t)(M > 0 ? M : 1) * 4) + g_align((size_t)(M > 0 ? M : 1) * 12);
}

static int grid_build(const float* points, int M, const int* M_dev, void* grid_ws, hipStream_t st) {
  GridWs g = grid_carve(grid_ws, M);
  hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(1024), 0, st, points, M, M_dev, g.hdr, g.start, GRID_MAX_CELLS);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_grid_count, dim3((M + 255) / 256), dim3(256), 0, st, points, (const GridHdr*)g.hdr, g.start, g.pcell);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, st, (const GridHdr*)g.hdr, g.start, g.cursor);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_grid_scatter, dim3((M + 255) / 256), dim3(256), 0, st, points, (const GridHdr*)g.hdr, (const int*)g.pcell, g.cursor,
                     g.sorted);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_scene_grid_build(const float* points, int M, void* grid_ws, void* stream) {
  MH_CHECK(points && grid_ws, "null argument");
  MH_CHECK(M > 0, "empty scene");
  return grid_build(points, M, nullptr, grid_ws, (hipStream_t)stream);
}

extern "C" int mh_scene_grid_build_dev(const float* points, const int* M_dev, int M_cap, void* grid_ws, void* stream) {
  MH_CHECK(points && M_dev && grid_ws, "null argument");
  MH_CHECK(M_cap > 0, "empty capacity");
  return grid_build(points, M_cap, M_dev, grid_ws, (hipStream_t)stream);
}

// (Round 6 tried the end of the per-cycle scene update -- un-projection / compaction, bbox + header, cell counts, scan, scatter:
// five dependent launches, ~190 us + gaps -- as the phases of ONE 1024-thread workgroup.  Same results; the cycle with the
// device-built scene went from 0.74 to 0.90 ms: 32 000 counting and 32 000 returning atomics from a single CU, with
// device-scope fences between the phases, take far longer than the 127-workgroup launches they replace.  Removed.)
// Keep the K smallest of the 128 buffered (distance^2, y) candidates of a wave in slots [0, K), unsorted, and return
// the K-th smallest distance: the two entries of a lane live in registers, the K-th value comes from a bitwise
// descent with ballots (non-negative floats order like unsigned integers), ties at the bound are kept in lane order.
// (A bitonic sort of the LDS buffer costs 28 dependent LDS exchange steps per call and made the whole search
// latency-bound.)  Slots [K, 128) are reset to +inf.
__device__ __forceinline__ float knn_select(float* sd, float* sy, int lane, int K) {
  const float d0 = sd[lane], d1 = sd[lane + 64], y0 = sy[lane], y1 = sy[lane + 64];
  const unsigned u0 = __float_as_uint(d0), u1 = __float_as_uint(d1);
  int k = K - 1;
  unsigned prefix = 0u;
#pragma unroll 4
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned m = (bit == 31 ? 0u : (0xffffffffu << (bit + 1))) | (1u << bit);
    const int c0 = __popcll(__ballot((u0 & m) == prefix)) + __popcll(__ballot((u1 & m) == prefix));
    if (k >= c0) { k -= c0; prefix |= 1u << bit; }
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long l0 = __ballot(u0 < prefix), l1 = __ballot(u1 < prefix);
  const unsigned long long e0 = __ballot(u0 == prefix), e1 = __ballot(u1 == prefix);
  const int nl0 = __popcll(l0), nless = nl0 + __popcll(l1), need = K - nless;
  int p0 = -1, p1 = -1;
  if (u0 < prefix) p0 = __popcll(l0 & below);
  else if (u0 == prefix) { const int r = __popcll(e0 & below); if (r < need) p0 = nless + r; }
  if (u1 < prefix) p1 = nl0 + __popcll(l1 & below);
  else if (u1 == prefix) { const int r = __popcll(e0) + __popcll(e1 & below); if (r < need) p1 = nless + r; }
  __builtin_amdgcn_wave_barrier();
  sd[lane] = INFINITY; sd[lane + 64] = INFINITY;
  __builtin_amdgcn_wave_barrier();
  if (p0 >= 0) { sd[p0] = d0; sy[p0] = y0; }
  if (p1 >= 0) { sd[p1] = d1; sy[p1] = y1; }
  __builtin_amdgcn_wave_barrier();
  return __uint_as_float(prefix);
}

// one wave per query
// (verts != null: the query is the body's lowest vertex, taken from the key the LBS forward's projection epilogue reported --
// mh_contact_knn_grid_key -- and written to low_idx / low_xyz_out for the contact / foot-sliding kernel)
// sel != null (mh_contact_knn_grid_sel): the scene is one of TWO grids, chosen by device-resident words -- sel[0] = 0: no scene
// yet, nothing is written; sel[1] = which grid -- so that one captured launch serves every phase of a fit (the scene
// update of cycle c builds the grid the cycle does not read, optimizer.py:578-584; the words are set between cycles)
__global__ __launch_bounds__(64) void k_contact_knn_grid(const GridHdr* hdr, const int* start, const float* sorted, int M,
                                                         const float* low_xyz, int K, float* dy, const float* verts, int V,
                                                         unsigned long long* lowkey, int* low_idx, float* low_xyz_out,
                                                         const int* sel, const GridHdr* hdr1, const int* start1, const float* sorted1) {
  if (sel) {
    if (sel[0] == 0) return;
    if (sel[1] != 0) { hdr = hdr1; start = start1; sorted = sorted1; }
  }
  __shared__ float sd_s[KNN_CAP];
  __shared__ float sy_s[KNN_CAP];
  __shared__ int s_off[65], s_a0[64], s_la[64], s_b0[64];
  float* sd = sd_s;
  float* sy = sy_s;
  const int b = blockIdx.x, lane = threadIdx.x;
  float qx, qy, qz;
  if (verts) {
    const float* vb = verts + (size_t)b * V * 3;
    const int bi = mh_lowest_from_key(vb, V, lowkey, b, lane);
    qx = vb[(size_t)bi * 3]; qy = vb[(size_t)bi * 3 + 1]; qz = vb[(size_t)bi * 3 + 2];
    if (lane == 0) {
      low_idx[b] = bi;
      low_xyz_out[(size_t)b * 3] = qx; low_xyz_out[(size_t)b * 3 + 1] = qy; low_xyz_out[(size_t)b * 3 + 2] = qz;
    }
  } else {
    qx = low_xyz[(size_t)b * 3]; qy = low_xyz[(size_t)b * 3 + 1]; qz = low_xyz[(size_t)b * 3 + 2];
  }
  sd[lane] = INFINITY; sd[lane + 64] = INFINITY;
  sy[lane] = 0.f; sy[lane + 64] = 0.f;
  __builtin_amdgcn_wave_barrier();
  const float cell = hdr->cell;
  const int dx = hdr->dim[0], dyy = hdr->dim[1], dz = hdr->dim[2];
  // cell of the query clamped into the grid: for q outside the (convex) bbox with projection q', every cloud point p
  // has |p-q| >= |p-q'|, so the shell bound below stays valid
  const int cqx = min(max((int)floorf((qx - hdr->mn[0]) / cell), 0), dx - 1),
            cqy = min(max((int)floorf((qy - hdr->mn[1]) / cell), 0), dyy - 1),
            cqz = min(max((int)floorf((qz - hdr->mn[2]) / cell), 0), dz - 1);
  float tau = INFINITY;
  int fill = K, found = 0;
  const int npts = hdr->npts;
  if (npts == 0) { if (lane == 0) dy[b] = 0.f; return; }
  const int kk = npts < K ? npts : K;
  // enough shells to cover the whole grid from wherever the query is
  const int rmax = max(max(max(cqx, dx - 1 - cqx), max(cqy, dyy - 1 - cqy)), max(cqz, dz - 1 - cqz));
  // append the candidates closer than the current bound (wave-aggregated); keep the best K when the buffer fills up
  auto insert = [&](float d2, float py) {
    const bool take = d2 < tau;
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return;
    if (take) {
      const int pos = fill + __popcll(m & ((1ull << lane) - 1ull));
      sd[pos] = d2;
      sy[pos] = py;
    }
    fill += __popcll(m);
    found += __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (fill > KNN_CAP - 64) {
      tau = knn_select(sd, sy, lane, K);
      fill = K;
    }
  };
  auto tighten = [&]() {
    tau = knn_select(sd, sy, lane, K);
    fill = K;
  };
  auto axis_gap = [&](float q, float lo, float hi) { return fmaxf(fmaxf(lo - q, q - hi), 0.f); };
  const float mnx = hdr->mn[0], mny = hdr->mn[1], mnz = hdr->mn[2];
  for (int rho = 0; rho <= rmax; ++rho) {
    if (rho >= 1 && found >= kk && tau <= (float)(rho - 1) * cell * (float)(rho - 1) * cell) break;
    const int side = 2 * rho + 1, nrows = side * side;
    for (int rbase = 0; rbase < nrows; rbase += 64) {
      // lane -> one (y,z) row of the shell: rows on the border of the square contribute the whole x run,
      // interior rows only the two end cells.  key = squared distance from the query to the run's box.
      const int r = rbase + lane;
      int a0 = 0, a1 = 0, b0 = 0, b1 = 0, rowbase = 0, x0 = 0, x1 = -1;
      float keya = INFINITY, keyb = INFINITY, dyz2 = 0.f;
      if (r < nrows) {
        const int oy = r / side - rho, oz = r % side - rho;
        const int cy = cqy + oy, cz = cqz + oz;
        if (cy >= 0 && cy < dyy && cz >= 0 && cz < dz) {
          rowbase = (cz * dyy + cy) * dx;
          const float gy = axis_gap(qy, mny + cy * cell, mny + (cy + 1) * cell);
          const float gz = axis_gap(qz, mnz + cz * cell, mnz + (cz + 1) * cell);
          dyz2 = gy * gy + gz * gz;
          const bool border = (oy == -rho || oy == rho || oz == -rho || oz == rho);
          if (border) {
            x0 = max(cqx - rho, 0);
            x1 = min(cqx + rho, dx - 1);
            if (dyz2 < tau) {
              a0 = start[rowbase + x0];
              a1 = start[rowbase + x1 + 1];
              const float gx = axis_gap(qx, mnx + x0 * cell, mnx + (x1 + 1) * cell);
              if (a1 > a0) keya = dyz2 + gx * gx;
            }
          } else {
            if (cqx - rho >= 0) {
              x0 = x1 = cqx - rho;
              a0 = start[rowbase + x0];
              a1 = start[rowbase + x0 + 1];
              const float gx = axis_gap(qx, mnx + x0 * cell, mnx + (x0 + 1) * cell);
              if (a1 > a0) keya = dyz2 + gx * gx;
            }
            if (cqx + rho < dx) {
              const int xb = cqx + rho;
              b0 = start[rowbase + xb];
              b1 = start[rowbase + xb + 1];
              const float gx = axis_gap(qx, mnx + xb * cell, mnx + (xb + 1) * cell);
              if (b1 > b0) keyb = dyz2 + gx * gx;
            }
          }
        }
      }
      // every run of this chunk of rows whose box can still hold a neighbour, flattened into one index space: the
      // points are then gathered 64 at a time with independent loads (a run-by-run walk, even nearest-first with a
      // tightening bound, costs a dependent L2/HBM round trip per run and is latency-bound: 54 us per query)
      const int lenA = keya < tau ? a1 - a0 : 0, lenB = keyb < tau ? b1 - b0 : 0;
      const int incl = mh_wave_scan_add(lenA + lenB);
      const int P = __builtin_amdgcn_readlane(incl, 63);
      if (P > 0) {
        s_off[lane] = incl - (lenA + lenB);
        if (lane == 63) s_off[64] = P;
        s_a0[lane] = a0; s_la[lane] = lenA; s_b0[lane] = b0;
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < P; base += 64) {
          const int i = base + lane;
          float d2 = INFINITY, py = 0.f;
          if (i < P) {
            int lo = 0, hi = 64;                       // largest o with s_off[o] <= i (it has a non-empty run)
#pragma unroll
            for (int it = 0; it < 6; ++it) {
              const int mid = (lo + hi) >> 1;
              if (s_off[mid] <= i) lo = mid; else hi = mid;
            }
            const int k = i - s_off[lo], la = s_la[lo];
            const int idx = k < la ? s_a0[lo] + k : s_b0[lo] + (k - la);
            const float ex = sorted[(size_t)idx * 3] - qx;
            py = sorted[(size_t)idx * 3 + 1];
            const float ey = py - qy, ez = sorted[(size_t)idx * 3 + 2] - qz;
            d2 = ex * ex + ey * ey + ez * ez;
          }
          insert(d2, py);
        }
        __builtin_amdgcn_wave_barrier();
        if (found >= kk && fill > K) tighten();
      }
    }
    if (found >= kk && fill > K) tighten();
  }
  knn_select(sd, sy, lane, K);          // the K nearest in slots [0, K) (fewer points than K: the rest is +inf)
  float s = (lane < kk) ? sy[lane] : 0.f;
  s = mh_wave_sum(s);
  if (lane == 0) dy[b] = s / (float)kk - qy;
}

extern "C" int mh_contact_knn_grid(const void* grid_ws, int M, const float* low_xyz, int B, int k, float* dy, void* stream) {
  MH_CHECK(grid_ws && low_xyz && dy, "null argument");
  MH_CHECK(M > 0 && B > 0, "empty input");
  MH_CHECK(k >= 1 && k <= 32, "k must be in 1..32");
  GridWs g = grid_carve((void*)grid_ws, M);
  mh_prof_mark(MH_PROF_CONTACT_KNN, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(k_contact_knn_grid, dim3(B), dim3(64), 0, (hipStream_t)stream, (const GridHdr*)g.hdr, (const int*)g.start,
                     (const float*)g.sorted, M, low_xyz, k, dy, (const float*)nullptr, 0, (unsigned long long*)nullptr, (int*)nullptr,
                     (float*)nullptr, (const int*)nullptr, (const GridHdr*)nullptr, (const int*)nullptr, (const float*)nullptr);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_CONTACT_KNN, 1, (hipStream_t)stream);
  return MH_OK;
}

extern "C" int mh_contact_knn_grid_key(const void* grid_ws, int M, const float* verts, int V, unsigned long long* lowkey, int B,
                                       int k, int32_t* low_idx, float* low_xyz, float* dy, void* stream) {
  MH_CHECK(grid_ws && verts && lowkey && low_idx && low_xyz && dy, "null argument");
  MH_CHECK(B > 0 && M > 0 && V > 0, "empty input");
  MH_CHECK(k >= 1 && k <= 32, "k must be in 1..32");
  GridWs g = grid_carve((void*)grid_ws, M);
  mh_prof_mark(MH_PROF_CONTACT_KNN, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(k_contact_knn_grid, dim3(B), dim3(64), 0, (hipStream_t)stream, (const GridHdr*)g.hdr, (const int*)g.start,
                     (const float*)g.sorted, M, (const float*)nullptr, k, dy, verts, V, lowkey, low_idx, low_xyz, (const int*)nullptr,
                     (const GridHdr*)nullptr, (const int*)nullptr, (const float*)nullptr);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_CONTACT_KNN, 1, (hipStream_t)stream);
  return MH_OK;
}

// mh_contact_knn_grid_key over one of two grids of the same capacity M, chosen on the device: sel_dev[0] = 0 -> no scene yet
// (nothing is read or written), else grid_ws0 (sel_dev[1] = 0) or grid_ws1.  lowkey NULL: the queries are low_xyz (an input).
extern "C" int mh_contact_knn_grid_sel(const void* grid_ws0, const void* grid_ws1, int M, const int32_t* sel_dev, const float* verts,
                                       int V, unsigned long long* lowkey, int B, int k, int32_t* low_idx, float* low_xyz, float* dy,
                                       void* stream) {
  MH_CHECK(grid_ws0 && grid_ws1 && sel_dev && low_xyz && dy, "null argument");
  MH_CHECK(!lowkey || (verts && low_idx), "null argument");
  MH_CHECK(B > 0 && M > 0 && (!lowkey || V > 0), "empty input");
  MH_CHECK(k >= 1 && k <= 32, "k must be in 1..32");
  GridWs g = grid_carve((void*)grid_ws0, M), h = grid_carve((void*)grid_ws1, M);
  mh_prof_mark(MH_PROF_CONTACT_KNN, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(k_contact_knn_grid, dim3(B), dim3(64), 0, (hipStream_t)stream, (const GridHdr*)g.hdr, (const int*)g.start,
                     (const float*)g.sorted, M, lowkey ? (const float*)nullptr : (const float*)low_xyz, k, dy, lowkey ? verts : (const float*)nullptr,
                     V, lowkey, low_idx, low_xyz, (const int*)sel_dev, (const GridHdr*)h.hdr, (const int*)h.start, (const float*)h.sorted);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_CONTACT_KNN, 1, (hipStream_t)stream);
  return MH_OK;
}
