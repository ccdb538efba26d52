// Internal helpers shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mhmocap_hip.h"

#define MH_KD 224       // rows of the k-major basis planes: 10 shape + 207 pose + 7 zero
#define MH_FS 224       // MH_FEAT_STRIDE
#define MH_NJ 24
#define MH_NKP 17

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// Split-precision contraction: an fp32 operand x is carried as two 16-bit terms x = hi + lo and the
// product a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on the 16-bit matrix pipe (16x the fp32 MFMA
// rate) with fp32 accumulation.  Forward (values): fp16 terms, operands pre-scaled by powers of two so that the
// low terms stay normal -> x is carried to 2^-22, result error ~3 x 2^-22 relative to sum |a.b|.
// Backward (gradients, any magnitude): bf16 terms -> 2^-16..2^-18.
#define MH_F16_FEAT_SHIFT 8   // features (beta | R - I) are multiplied by 2^8 before the split (|beta| < 255)

void mh_set_error(const char* fmt, ...);
// event brackets around the named kernels (no-ops unless mh_profile_enable(1)); edge 0 = before, 1 = after
void mh_prof_mark(int which, int edge, hipStream_t st);
bool mh_prof_on();
int mh_prof_level();     // 1: event brackets; 2: + work counters inside the kernels (their stores perturb the timings)

#define MH_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      mh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return MH_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define MH_CHECK(cond, msg)                          \
  do {                                               \
    if (!(cond)) {                                   \
      mh_set_error("invalid argument: %s", msg);     \
      return MH_ERR_INVALID;                         \
    }                                                \
  } while (0)

#define MH_LAUNCH_CHECK()                                        \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) {                                     \
      mh_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return MH_ERR_HIP;                                         \
    }                                                            \
  } while (0)

// hipFuncSetAttribute is per DEVICE: latch "done" per (call site, device), not per process
#define MH_MAX_DEVICES 64
static inline bool mh_first_on_device(unsigned char* flags /*[MH_MAX_DEVICES], zero-initialised static*/) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MH_MAX_DEVICES) return true;
  if (flags[d]) return false;
  flags[d] = 1;
  return true;
}

struct mh_tree {
  int parent[MH_NJ];
  int level[MH_NJ];
  int maxlevel;
};

// CSR of a sparse joint regressor, rows = joints
struct mh_regressor {
  int J;           // 0 = absent
  int nnz;
  int* ptr;        // [J+1]
  int* vidx;       // [nnz]
  float* w;        // [nnz]
  float* rowsum;   // [J]
};

// the 2D key-point term from the pose features and joint transforms (mh_keypoints.hip): constants per (key-point, bone) pair
struct mh_kp_tables {
  int np;          // pairs (0: no key-point regressor)
  int rows_pad;    // 3 np rounded up to 32
  float* Qa;       // [rows_pad/32][112][64] rows of Q in MFMA A-operand order
  float* Qr;       // [rows_pad][224]        the same, row-major
  float* M0m;      // [np][4]  (M0 xyz, m)
  int* pair_j;     // [np]
  int* pair_k;     // [np]
  int* jptr;       // [18] pairs by key-point
  int* kptr;       // [25] pairs by bone ...
  int* kpairs;     // [np] ... their indices
};

struct mh_model {
  int V, VP, F, nw;
  float* vt;       // [VP][3]   template, zero padded
  float* D;        // [VP/32][MH_KD/16][3][64][8] basis tiled for the forward MFMA B operand (see mh_model.hip)
  float* Dt;       // [3][VP][16][16] the same, tiled for the backward MFMA B operand (see mh_model.hip)
  uint16_t* D16;   // [VP/32][14][3][2][64][8] fp16 (hi, lo) terms of 2^d16_shift x basis, forward B operand of
                   // v_mfma_f32_32x32x16_f16: lane l = (vertex l&31, k half l>>5) holds k = 16 s + 8 (l>>5) + 0..7
  int d16_shift;
  uint16_t* Dt16;  // [VP/16][3][7][2][64][8] bf16 (hi, lo) terms of the basis, backward B operand of
                   // v_mfma_f32_32x32x16_bf16 (contraction over the 16 vertices of a block): lane l = (basis column
                   // 32 ct + (l&31), vertex half l>>5) holds vertices 16 blk + 8 (l>>5) + 0..7 of component c
  uint16_t* W16;   // [VP/16][2][64][8] bf16 (hi, lo) terms of the dense skinning weights, B operand of
                   // v_mfma_f32_32x32x16_bf16: lane l = (joint l&31 (24 used), vertex half l>>5)
  int* skidx;      // [VP][nw] bones of the <= nw non-zero skinning weights per vertex
  float* skw;      // [VP][nw]
  float* Jt;       // [24][3]     J_regressor . v_template
  float* JS;       // [24][3][10] J_regressor . shapedirs
  int* faces;      // [F][3]
  mh_tree tree;
  mh_regressor reg[4];
  // alphapose regressor by vertex (CSR over vertices) for the backward scatter
  int* kpv_ptr;    // [VP+1]
  int* kpv_j;      // [nnz]
  float* kpv_w;    // [nnz]
  int* kpv_head;   // [VP][4] (first entry, entry count, first joint, first weight bits)
  mh_kp_tables kp;
};

int mh_kp_build(mh_model* m, const mh_model_host* h);
void mh_kp_free(mh_model* m);
// the extra chunk slot of the LBS backward's partial sums that mh_keypoint_terms fills (mh_lbs.hip)
int mh_lbs_backward_extra_slot(const mh_model* m, int B, void* ws2, float** pF, float** pA, float** pS);
// the pose features, bone transforms and scales the last mh_lbs_forward* on `ws` left there (mh_lbs.hip owns the layout)
int mh_lbs_forward_views(int B, const void* ws, const float** featT, const float** A, const float** scale);

static inline int mh_groups(int B) { return (B + 31) / 32; }

// ---- fp32 helpers used by several kernels --------------------------------------------------
__device__ __forceinline__ void mh_rodrigues(const float r[3], float R[9]) {
  // smpl.py:647-678 -- angle from the eps-shifted vector, axis from the plain one
  const float e = 1e-8f;
  float a0 = r[0] + e, a1 = r[1] + e, a2 = r[2] + e;
  float ang = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  float x = r[0] / ang, y = r[1] / ang, z = r[2] / ang;
  float s = sinf(ang), c = cosf(ang);
  float omc = 1.0f - c;
  // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  K.K written out as the explicit 3x3 product
  float k2_00 = -z * z - y * y, k2_01 = x * y, k2_02 = x * z;
  float k2_10 = x * y, k2_11 = -z * z - x * x, k2_12 = y * z;
  float k2_20 = x * z, k2_21 = y * z, k2_22 = -y * y - x * x;
  R[0] = 1.0f + omc * k2_00;
  R[1] = s * (-z) + omc * k2_01;
  R[2] = s * y + omc * k2_02;
  R[3] = s * z + omc * k2_10;
  R[4] = 1.0f + omc * k2_11;
  R[5] = s * (-x) + omc * k2_12;
  R[6] = s * (-y) + omc * k2_20;
  R[7] = s * x + omc * k2_21;
  R[8] = 1.0f + omc * k2_22;
}

__device__ __forceinline__ float mh_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// wave64 inclusive add-scan on the DPP network (row shifts inside the rows of 16, then the two row broadcasts)
__device__ __forceinline__ int mh_wave_scan_add(int x) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}

// ---- the rasterised terms' closing job (mh_raster_fin, include/mhmocap_hip.h): carried out by k_raster_finish (mh_raster.hip,
// one workgroup of its own) or by one extra workgroup of k_pose_bwd (mh_lbs.hip), NT = threads of that workgroup --------------
//   from_partials != 0 (gradients were requested: k_raster_grads has run): per body, the sums of its tiles in fixed order
//     -> depth loss, depth-range partials, silhouette loss with the alpha-dependent part k_raster_grads accumulated
//     (cleared here for the next launch);
//   from_partials == 0 (values only): k_raster_body_out has done that from k_raster_sums' totals;
//   then the chain of the depth-range leaves (optimizer.py:683-688: min_z = softplus(zmin), max_z = min_z.detach() + 1 +
//   softplus(zmax)) and, when asked for, the two loss sums of the log row.
#define MH_FIN_U 4            // bodies per thread and pass: their dependent loads (tile range -> tile sums) are in flight together
template <int NT>
__device__ __forceinline__ void mh_raster_finish_job(const mh_raster_fin& f, float* s_g0 /*[MH_FIN_U * NT]*/, float* s_g1 /*[MH_FIN_U * NT]*/) {
  const int N = f.N, tid = threadIdx.x, T = f.T;
  const int fpc = max(1, MH_FIN_U * NT / N);     // whole frames per pass: thread = body (MH_FIN_U of them), then thread = frame
  for (int t0 = 0; t0 < T; t0 += fpc) {
    const int nb = min(fpc, T - t0) * N;         // bodies of this pass
    int first[MH_FIN_U], ns[MH_FIN_U];
#pragma unroll
    for (int u = 0; u < MH_FIN_U; ++u) {
      const int i = tid + u * NT, b = t0 * N + i;
      first[u] = 0; ns[u] = 0;
      if (i < nb && b < f.B && f.from_partials) { first[u] = f.body_first[b]; ns[u] = f.body_ns[b]; }
    }
    float S[MH_FIN_U][6];
#pragma unroll
    for (int u = 0; u < MH_FIN_U; ++u) {
#pragma unroll
      for (int k = 0; k < 6; ++k) S[u][k] = 0.f;
      for (int s = first[u]; s < first[u] + ns[u]; ++s)      // fixed order over the body's tiles
#pragma unroll
        for (int k = 0; k < 6; ++k) S[u][k] += f.partial[(size_t)s * 6 + k];
    }
#pragma unroll
    for (int u = 0; u < MH_FIN_U; ++u) {
      const int i = tid + u * NT, b = t0 * N + i;
      float g0 = 0.f, g1 = 0.f;
      if (i < nb && b < f.B) {
        if (f.from_partials) {
          const float cnt = S[u][2] + 1.f;
          const float diff = S[u][0] / cnt - S[u][1] / cnt;                                            // losses.py:24-27
          f.depth_body[b] = diff * diff;
          const float gB = f.coef_depth * (-2.f) * diff / cnt;
          g0 = gB * S[u][3];                        // d/d(1/min_z) through the target disparity
          g1 = gB * S[u][4];                        // d/d(1/max_z)
          f.sil_body[b] = f.sil_apply[b] * (f.sil_S[b] + f.sil_corr[b]) / (f.sil_D[b] + 1.f);        // losses.py:35-38
          f.sil_corr[b] = 0.f;
        } else {
          g0 = f.dinv[(size_t)b * 2];
          g1 = f.dinv[(size_t)b * 2 + 1];
          f.sil_corr[b] = 0.f;                      // (gradients AND images in one call: k_raster_grads ran after k_raster_body_out)
        }
      }
      s_g0[i] = g0; s_g1[i] = g1;
    }
    __syncthreads();
    if (f.gzmin)
      for (int fr = tid; fr < fpc && t0 + fr < T; fr += NT) {
        const int t = t0 + fr;
        float a0 = 0.f, a1 = 0.f;
        for (int n = 0; n < N; ++n) { a0 += s_g0[fr * N + n]; a1 += s_g1[fr * N + n]; }
        const float e0 = expf(f.zmin_lin[t]), e1 = expf(f.zmax_lin[t]);
        const float min_z = logf(1.f + e0);
        const float max_z = min_z + 1.f + logf(1.f + e1);
        f.gzmin[t] += a0 * (-1.f / (min_z * min_z)) * (e0 / (1.f + e0));
        f.gzmax[t] += a1 * (-1.f / (max_z * max_z)) * (e1 / (1.f + e1));
      }
    __syncthreads();
  }
  if (f.log_depth || f.log_sil) {
    // sums of the per-body values in an order that does not depend on NT: aligned groups of 64 bodies by a wave butterfly,
    // then the groups in turn -- the log row is bit-identical whichever launch carries the job.  (The values were written
    // by this workgroup, behind the barriers above; every wave's loads are in flight together.)
    const int ngr = (f.B + 63) / 64, lane = tid & 63;
    float ta = 0.f, tc = 0.f;                      // thread 0: the running totals
    for (int g0 = 0; g0 < ngr; g0 += MH_FIN_U * NT) {
      const int nh = min(MH_FIN_U * NT, ngr - g0);
      for (int gr = tid >> 6; gr < nh; gr += NT / 64) {
        const int b = (g0 + gr) * 64 + lane;
        float a = b < f.B ? f.depth_body[b] : 0.f, c = b < f.B ? f.sil_body[b] : 0.f;
        a = mh_wave_sum(a); c = mh_wave_sum(c);
        if (lane == 0) { s_g0[gr] = a; s_g1[gr] = c; }
      }
      __syncthreads();
      if (tid == 0)
        for (int gr = 0; gr < nh; ++gr) { ta += s_g0[gr]; tc += s_g1[gr]; }
      __syncthreads();
    }
    if (tid == 0) {
      if (f.log_depth) *f.log_depth = ta;
      if (f.log_sil) *f.log_sil = tc;
    }
  }
}
