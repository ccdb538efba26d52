// The 2D key-point term without a pass over the vertices (round 4).
//
// Reference: joints_alphapose = J_regressor_alphapose . verts (smpl.py:374-376, 603-620), scaled and translated
// (optimizer.py:701-703), projected (transforms.py:57-95) and compared with the detections (optimizer.py:364-368, 404,
// 419-420).  Rounds 1-3 regressed the 17 key-points from the (B,V,3) vertex buffer (k_joints_regress: 673 scattered
// 12-byte gathers per body) and scattered their adjoint back into the vertices inside the LBS backward.
//
// The vertices are linear in what the pose kernel already has.  With v_posed = vt + D^T f (f = [beta; R - I], 217
// features) and verts_local = sum_k W[v][k] A_k [v_posed; 1]:
//
//     kp_local[j] = sum_v R[j][v] verts_local[v] = sum_k A_k . [ M0[j][k] + Q[j][k] f ;  m[j][k] ]
//
//     m [j][k]      = sum_v R[j][v] W[v][k]                    (scalar)
//     M0[j][k][c]   = sum_v R[j][v] W[v][k] vt[v][c]           (3)
//     Q [j][k][c][] = sum_v R[j][v] W[v][k] D[.][v][c]         (3 x 217)
//
// -- constants of the model, built once on the host in double precision (the same hoisting the reference's own joints get:
// J = J_regressor.(v_template + shapedirs.beta), smpl.py:532-535).  Only the (joint, bone) pairs that share a vertex
// exist: ~5 bones per key-point for SMPL.  Per group of 32 bodies the kernel below evaluates
//     P = Q f                  (3 np x 224) . (224 x 32)   exact-fp32 MFMA 32x32x2
//     key-points, projection, residual, d loss / d key-point
//     dL/dA_k, dL/dM           element-wise
//     dL/df = Q^T dL/dM        (224 x 3 np) . (3 np x 32)  exact-fp32 MFMA 32x32x2
// (three small launches, see below) and leaves dL/dA, dL/df, dL/dt as ONE MORE chunk of the LBS backward's partial sums (k_pose_bwd adds the chunks in fixed
// order), so neither the forward nor the backward of the skinning kernels knows about key-points any more.
#include <vector>

#include "mh_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------------
// host: tables
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int kp_upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
  MH_HIP(hipMalloc((void**)dst, bytes));
  if (!src.empty()) MH_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return MH_OK;
}

int mh_kp_build(mh_model* m, const mh_model_host* h) {
  mh_kp_tables& t = m->kp;
  memset(&t, 0, sizeof(t));
  if (!h->reg_alphapose) return MH_OK;
  const int V = h->num_verts;
  // pairs (joint, bone) that share a vertex, joint-major
  std::vector<int> pj, pk, jptr(MH_NKP + 1, 0);
  std::vector<std::vector<std::pair<int, double>>> rows(MH_NKP);           // non-zeros of the regressor by joint
  for (int j = 0; j < MH_NKP; ++j)
    for (int v = 0; v < V; ++v) {
      const float r = h->reg_alphapose[(size_t)j * V + v];
      if (r != 0.f) rows[j].push_back({v, (double)r});
    }
  for (int j = 0; j < MH_NKP; ++j) {
    bool has[MH_NJ] = {false};
    for (auto& e : rows[j])
      for (int k = 0; k < MH_NJ; ++k)
        if (h->lbs_weights[(size_t)e.first * MH_NJ + k] != 0.f) has[k] = true;
    for (int k = 0; k < MH_NJ; ++k)
      if (has[k]) { pj.push_back(j); pk.push_back(k); }
    jptr[j + 1] = (int)pj.size();
  }
  const int np = (int)pj.size(), nrows = 3 * np, rows_pad = ((nrows + 31) / 32) * 32;
  t.np = np; t.rows_pad = rows_pad;
  std::vector<float> Qr((size_t)std::max(rows_pad, 32) * MH_FS, 0.f), M0m((size_t)std::max(np, 1) * 4, 0.f);
  for (int p = 0; p < np; ++p) {
    const int j = pj[p], k = pk[p];
    double m0[3] = {0, 0, 0}, mm = 0;
    std::vector<double> q((size_t)3 * MH_FS, 0.0);
    for (auto& e : rows[j]) {
      const int v = e.first;
      const double rw = e.second * (double)h->lbs_weights[(size_t)v * MH_NJ + k];
      if (rw == 0.0) continue;
      mm += rw;
      for (int c = 0; c < 3; ++c) {
        m0[c] += rw * (double)h->v_template[(size_t)v * 3 + c];
        double* qc = q.data() + (size_t)c * MH_FS;
        const float* sd = h->shapedirs + ((size_t)v * 3 + c) * MH_NUM_BETAS;
        for (int l = 0; l < MH_NUM_BETAS; ++l) qc[l] += rw * (double)sd[l];
        const float* pd = h->posedirs + ((size_t)v * 3 + c) * MH_NUM_POSE_BASIS;
        for (int l = 0; l < MH_NUM_POSE_BASIS; ++l) qc[MH_NUM_BETAS + l] += rw * (double)pd[l];
      }
    }
    for (int c = 0; c < 3; ++c) {
      M0m[(size_t)p * 4 + c] = (float)m0[c];
      for (int l = 0; l < MH_FS; ++l) Qr[((size_t)3 * p + c) * MH_FS + l] = (float)q[(size_t)c * MH_FS + l];
    }
    M0m[(size_t)p * 4 + 3] = (float)mm;
  }
  // the same rows in the A-operand order of v_mfma_f32_32x32x2f32: [row tile][k step][lane = (k & 1) * 32 + row % 32]
  std::vector<float> Qa((size_t)std::max(rows_pad, 32) * MH_FS, 0.f);
  for (int rt = 0; rt < rows_pad / 32; ++rt)
    for (int s = 0; s < MH_FS / 2; ++s)
      for (int l = 0; l < 64; ++l)
        Qa[((size_t)rt * (MH_FS / 2) + s) * 64 + l] = Qr[((size_t)rt * 32 + (l & 31)) * MH_FS + 2 * s + (l >> 5)];
  // pairs by bone
  std::vector<int> kptr(MH_NJ + 1, 0), kpairs;
  for (int k = 0; k < MH_NJ; ++k) {
    for (int p = 0; p < np; ++p)
      if (pk[p] == k) kpairs.push_back(p);
    kptr[k + 1] = (int)kpairs.size();
  }
  int rc;
  if ((rc = kp_upload(&t.Qa, Qa))) return rc;
  if ((rc = kp_upload(&t.Qr, Qr))) return rc;
  if ((rc = kp_upload(&t.M0m, M0m))) return rc;
  if ((rc = kp_upload(&t.pair_j, pj))) return rc;
  if ((rc = kp_upload(&t.pair_k, pk))) return rc;
  if ((rc = kp_upload(&t.jptr, jptr))) return rc;
  if ((rc = kp_upload(&t.kptr, kptr))) return rc;
  if ((rc = kp_upload(&t.kpairs, kpairs))) return rc;
  return MH_OK;
}

void mh_kp_free(mh_model* m) {
  mh_kp_tables& t = m->kp;
  void* q[] = {t.Qa, t.Qr, t.M0m, t.pair_j, t.pair_k, t.jptr, t.kptr, t.kpairs};
  for (void* p : q)
    if (p) (void)hipFree(p);
  memset(&t, 0, sizeof(t));
}

// ---------------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------------
struct KpP {
  int B, G, np, rows_pad;
  size_t GB;
  const float* Qa;
  const float* Qr;
  const float* M0m;
  const int* pair_j;
  const int* pair_k;
  const int* jptr;
  const int* kptr;
  const int* kpairs;
  const float* featT;   // [G][224][32]
  const float* A;       // [G*32][24][12]
  const float* scale;   // [G*32]
  const float* transl;  // [B][3] or null
  // projection + residual (as k_project_loss, mode 0)
  int has_kd;
  float K[9];
  float Kd[5];
  float jw[MH_NKP];
  float thr, w, h, coef;
  const float* pose2d;  // [B][17][3]
  float* kp;            // [B][17][3]
  float* uv;            // [B][17][2] or null
  float* gkp;           // [B][17][3] or null
  float* loss;          // [B]
  // the extra chunk of the LBS backward's partial sums (null: values only)
  float* pF;            // [GB][224]
  float* pA;            // [GB][12][24]
  float* pS;            // [GB][4]
  // scratch
  float* P;             // [rows_pad][GB]
  float* GM;            // [rows_pad][GB]
};

// Three launches (the phases need all of a group's rows of the one before; 25 groups alone leave the device idle, and a wave
// that walks K = 224 alone is a chain of 112 dependent load -> MFMA steps -- the first, single-kernel form took 180 us):
//   k_kp_p    grid (row tiles, groups): P = Q f.  The four waves of a workgroup split K; every operand of a wave's 28 steps
//             is requested before its first MFMA; the four partial tiles are added through LDS in fixed order.
//   k_kp_mid  grid (groups): key-points, projection, residual, dL/dkp; dL/dA, dL/dM, dL/dt (element-wise)
//   k_kp_gf   grid (feature tiles, groups): dL/df = Q^T dL/dM, K = rows split over the four waves as above
#define KP_T 256
#define KP_KS (MH_FS / 2 / 4)          // 28 two-k steps per wave
__global__ __launch_bounds__(KP_T) void k_kp_p(KpP p) {
  __shared__ float sAcc[4][16][64];
  const int rt = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* fb = p.featT + (size_t)g * MH_FS * 32 + lane + (size_t)wave * KP_KS * 64;   // lane l: feature 2 s + (l >> 5), body l & 31
  const float* qa = p.Qa + ((size_t)rt * (MH_FS / 2) + (size_t)wave * KP_KS) * 64 + lane;
  float a[KP_KS], f[KP_KS];
#pragma unroll
  for (int s = 0; s < KP_KS; ++s) { a[s] = qa[(size_t)s * 64]; f[s] = fb[(size_t)s * 64]; }
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < KP_KS; ++s) acc = MFMA32(a[s], f[s], acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) sAcc[wave][r][lane] = acc[r];
  __syncthreads();
  for (int e = tid; e < 16 * 64; e += KP_T) {
    const int r = e >> 6, l = e & 63;
    const float v = ((sAcc[0][r][l] + sAcc[1][r][l]) + sAcc[2][r][l]) + sAcc[3][r][l];
    const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    p.P[(size_t)row * p.GB + (size_t)g * 32 + (l & 31)] = v;
  }
}

__global__ __launch_bounds__(KP_T) void k_kp_mid(KpP p) {
  __shared__ float sGK[MH_NKP * 3][32];      // d loss / d key-point
  __shared__ float sKL[MH_NKP * 3][32];      // key-point before scale and translation
  __shared__ float sL[MH_NKP][32];
  const int g = blockIdx.x, tid = threadIdx.x;
  const size_t GB = p.GB;
  // ---- 2. key-points, projection, residual -----------------------------------------------------------------------------------
  for (int idx = tid; idx < 32 * MH_NKP; idx += KP_T) {
    const int bi = idx & 31, j = idx >> 5, b = g * 32 + bi;
    float kl[3] = {0.f, 0.f, 0.f};
    if (b < p.B) {
      for (int q = p.jptr[j]; q < p.jptr[j + 1]; ++q) {
        const int k = p.pair_k[q];
        const f32x4 mm = *(const f32x4*)(p.M0m + (size_t)q * 4);
        const float M0 = mm[0] + p.P[((size_t)3 * q) * GB + b], M1 = mm[1] + p.P[((size_t)3 * q + 1) * GB + b],
                    M2 = mm[2] + p.P[((size_t)3 * q + 2) * GB + b];
        const f32x4* Ak = (const f32x4*)(p.A + ((size_t)b * MH_NJ + k) * 12);
        const f32x4 a0 = Ak[0], a1 = Ak[1], a2 = Ak[2];
        kl[0] += fmaf(a0[3], mm[3], fmaf(a0[2], M2, fmaf(a0[1], M1, a0[0] * M0)));
        kl[1] += fmaf(a1[3], mm[3], fmaf(a1[2], M2, fmaf(a1[1], M1, a1[0] * M0)));
        kl[2] += fmaf(a2[3], mm[3], fmaf(a2[2], M2, fmaf(a2[1], M1, a2[0] * M0)));
      }
    }
    float l = 0.f, g3[3] = {0.f, 0.f, 0.f};
    if (b < p.B) {
      const float sc = p.scale[b];
      const float X = fmaf(sc, kl[0], p.transl ? p.transl[(size_t)b * 3] : 0.f);           // optimizer.py:703
      const float Y = fmaf(sc, kl[1], p.transl ? p.transl[(size_t)b * 3 + 1] : 0.f);
      const float Z = fmaf(sc, kl[2], p.transl ? p.transl[(size_t)b * 3 + 2] : 0.f);
      const size_t o = (size_t)b * MH_NKP + j;
      p.kp[o * 3] = X; p.kp[o * 3 + 1] = Y; p.kp[o * 3 + 2] = Z;
      const float x = X / Z, y = Y / Z;                        // transforms.py:75-76
      float xx = x, yy = y, dxx_dx = 1, dxx_dy = 0, dyy_dx = 0, dyy_dy = 1;
      if (p.has_kd) {                                          // transforms.py:78-90
        const float k1 = p.Kd[0], k2 = p.Kd[1], p1 = p.Kd[2], p2 = p.Kd[3], k3 = p.Kd[4];
        const float r = x * x + y * y;
        const float rad = 1 + k1 * r + k2 * r * r + k3 * r * r * r;
        const float drad = k1 + 2 * k2 * r + 3 * k3 * r * r;
        xx = x * rad + 2 * p1 * x * y + p2 * (r + 2 * x * x);
        yy = y * rad + 2 * p2 * y * y + p1 * (r + 2 * y * y);  // (sic) the reference's second tangential term
        dxx_dx = rad + x * drad * 2 * x + 2 * p1 * y + p2 * (2 * x + 4 * x);
        dxx_dy = x * drad * 2 * y + 2 * p1 * x + p2 * (2 * y);
        dyy_dx = y * drad * 2 * x + p1 * (2 * x);
        dyy_dy = rad + y * drad * 2 * y + 4 * p2 * y + p1 * (2 * y + 4 * y);
      }
      const float u = xx * p.K[0] + yy * p.K[1] + p.K[2];      // transforms.py:92
      const float v = xx * p.K[3] + yy * p.K[4] + p.K[5];
      if (p.uv) { p.uv[o * 2] = u; p.uv[o * 2 + 1] = v; }
      const float conf = p.pose2d[o * 3 + 2];
      const float c = conf >= p.thr ? p.jw[j] : 0.f;           // mask = pose_weights * (conf >= thr)  (optimizer.py:404, 419-420)
      const float du = c * u / p.w - c * p.pose2d[o * 3] / p.w;                             // :364-368
      const float dv = c * v / p.h - c * p.pose2d[o * 3 + 1] / p.h;
      l = du * du + dv * dv;
      const float gu = 2 * du * c / p.w * p.coef, gvv = 2 * dv * c / p.h * p.coef;
      const float gxx = gu * p.K[0] + gvv * p.K[3], gyy = gu * p.K[1] + gvv * p.K[4];
      const float gx = gxx * dxx_dx + gyy * dyy_dx, gy = gxx * dxx_dy + gyy * dyy_dy;
      g3[0] = gx / Z;
      g3[1] = gy / Z;
      g3[2] = -(gx * x + gy * y) / Z;
      if (p.gkp) { p.gkp[o * 3] = g3[0]; p.gkp[o * 3 + 1] = g3[1]; p.gkp[o * 3 + 2] = g3[2]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { sGK[j * 3 + c][bi] = g3[c]; sKL[j * 3 + c][bi] = kl[c]; }
    sL[j][bi] = l;
  }
  __syncthreads();
  if (tid < 32) {
    const int b = g * 32 + tid;
    float l = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, sx = 0.f;
#pragma unroll
    for (int j = 0; j < MH_NKP; ++j) {                          // fixed order
      l += sL[j][tid];
      t0 += sGK[j * 3][tid]; t1 += sGK[j * 3 + 1][tid]; t2 += sGK[j * 3 + 2][tid];
      sx += sGK[j * 3][tid] * sKL[j * 3][tid] + sGK[j * 3 + 1][tid] * sKL[j * 3 + 1][tid] + sGK[j * 3 + 2][tid] * sKL[j * 3 + 2][tid];
    }
    if (b < p.B) p.loss[b] = l;
    if (p.pS) {
      float* o = p.pS + (size_t)b * 4;                           // dL/dt; [3] = sum g.x (the exact-fp32 backward's scale term)
      o[0] = t0; o[1] = t1; o[2] = t2; o[3] = sx;
    }
  }
  if (!p.pF) return;
  // ---- dL/dM per (body, pair) and dL/dA per (body, bone) ---------------------------------------------------------------------
  for (int idx = tid; idx < 32 * p.np; idx += KP_T) {
    const int bi = idx & 31, q = idx >> 5, b = g * 32 + bi;
    const int j = p.pair_j[q], k = p.pair_k[q];
    float gm[3] = {0.f, 0.f, 0.f};
    if (b < p.B) {
      const float sc = p.scale[b];
      const f32x4* Ak = (const f32x4*)(p.A + ((size_t)b * MH_NJ + k) * 12);
      const f32x4 a0 = Ak[0], a1 = Ak[1], a2 = Ak[2];
      const float g0 = sc * sGK[j * 3][bi], g1 = sc * sGK[j * 3 + 1][bi], g2 = sc * sGK[j * 3 + 2][bi];
#pragma unroll
      for (int c = 0; c < 3; ++c) gm[c] = fmaf(a2[c], g2, fmaf(a1[c], g1, a0[c] * g0));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p.GM[((size_t)3 * q + c) * GB + (size_t)g * 32 + bi] = gm[c];
  }
  for (int idx = tid; idx < 32 * (p.rows_pad - 3 * p.np); idx += KP_T)         // padding rows (their Q rows are zero; no NaNs)
    p.GM[((size_t)3 * p.np + (idx >> 5)) * GB + (size_t)g * 32 + (idx & 31)] = 0.f;
  for (int idx = tid; idx < 32 * MH_NJ; idx += KP_T) {
    const int bi = idx & 31, k = idx >> 5, b = g * 32 + bi;
    float ga[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) ga[e] = 0.f;
    if (b < p.B) {
      const float sc = p.scale[b];
      for (int i = p.kptr[k]; i < p.kptr[k + 1]; ++i) {
        const int q = p.kpairs[i], j = p.pair_j[q];
        const f32x4 mm = *(const f32x4*)(p.M0m + (size_t)q * 4);
        const float M[4] = {mm[0] + p.P[((size_t)3 * q) * GB + b], mm[1] + p.P[((size_t)3 * q + 1) * GB + b],
                            mm[2] + p.P[((size_t)3 * q + 2) * GB + b], mm[3]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float gr = sc * sGK[j * 3 + r][bi];
#pragma unroll
          for (int c = 0; c < 4; ++c) ga[r * 4 + c] = fmaf(gr, M[c], ga[r * 4 + c]);
        }
      }
    }
    float* o = p.pA + (size_t)b * 288 + k;
#pragma unroll
    for (int e = 0; e < 12; ++e) o[e * MH_NJ] = ga[e];
  }
}

#define KP_GB 16                       // steps whose operands are in flight together
__global__ __launch_bounds__(KP_T) void k_kp_gf(KpP p) {
  __shared__ float sAcc[4][16][64];
  const int kt = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int nstep = p.rows_pad / 2, per = (nstep + 3) / 4, s0 = wave * per, s1 = min(nstep, s0 + per);
  const float* qr = p.Qr + (size_t)lh * MH_FS + kt * 32 + li;            // lane l: row 2 s + (l >> 5), feature 32 kt + (l & 31)
  const float* gm = p.GM + (size_t)lh * p.GB + (size_t)g * 32 + li;      //         row 2 s + (l >> 5), body l & 31
  f32x16 acc = {0};
  for (int sb = s0; sb < s1; sb += KP_GB) {
    float a[KP_GB], f[KP_GB];
#pragma unroll
    for (int u = 0; u < KP_GB; ++u) {
      const int s = min(sb + u, s1 - 1);
      a[u] = qr[(size_t)2 * s * MH_FS];
      f[u] = (sb + u < s1) ? gm[(size_t)2 * s * p.GB] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < KP_GB; ++u) acc = MFMA32(a[u], f[u], acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sAcc[wave][r][lane] = acc[r];
  __syncthreads();
  for (int e = tid; e < 16 * 64; e += KP_T) {
    const int r = e >> 6, l = e & 63;
    const float v = ((sAcc[0][r][l] + sAcc[1][r][l]) + sAcc[2][r][l]) + sAcc[3][r][l];
    const int f = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    p.pF[((size_t)g * 32 + (l & 31)) * MH_FS + f] = v;
  }
}

// scratch for P and dL/dM
extern "C" size_t mh_keypoint_workspace_bytes(const mh_model* m, int B) {
  if (!m || B <= 0) return 0;
  const size_t GB = (size_t)mh_groups(B) * 32;
  return 2 * (size_t)(m->kp.rows_pad > 0 ? m->kp.rows_pad : 32) * GB * sizeof(float) + 512;
}

extern "C" int mh_keypoint_terms(const mh_model* m, int B, const float* transl, const float* K_host, const float* Kd_host,
                                 const float* joint_w_host, const float* pose2d, float thr, float img_w, float img_h,
                                 float coef, float* kp, float* uv, float* gkp, float* loss, const void* ws, void* ws2,
                                 void* kp_ws, void* stream) {
  MH_CHECK(m && K_host && pose2d && kp && loss && ws && kp_ws, "null argument");
  MH_CHECK(B > 0, "B must be positive");
  MH_CHECK(m->kp.np > 0, "the key-point regressor was not given to mh_model_create");
  KpP p;
  const int G = mh_groups(B);
  p.B = B; p.G = G; p.np = m->kp.np; p.rows_pad = m->kp.rows_pad; p.GB = (size_t)G * 32;
  p.Qa = m->kp.Qa; p.Qr = m->kp.Qr; p.M0m = m->kp.M0m; p.pair_j = m->kp.pair_j; p.pair_k = m->kp.pair_k;
  p.jptr = m->kp.jptr; p.kptr = m->kp.kptr; p.kpairs = m->kp.kpairs;
  // the forward's workspace as the immediately preceding mh_lbs_forward* on it left it (the layout is mh_lbs.hip's: one accessor)
  if (int rc = mh_lbs_forward_views(B, ws, &p.featT, &p.A, &p.scale)) return rc;
  p.transl = transl;
  p.has_kd = Kd_host != nullptr;
  for (int i = 0; i < 9; ++i) p.K[i] = K_host[i];
  for (int i = 0; i < 5; ++i) p.Kd[i] = Kd_host ? Kd_host[i] : 0.f;
  for (int i = 0; i < MH_NKP; ++i) p.jw[i] = joint_w_host ? joint_w_host[i] : 1.f;
  p.thr = thr; p.w = img_w; p.h = img_h; p.coef = coef;
  p.pose2d = pose2d; p.kp = kp; p.uv = uv; p.gkp = gkp; p.loss = loss;
  p.pF = p.pA = p.pS = nullptr;
  if (ws2) {
    int rc = mh_lbs_backward_extra_slot(m, B, ws2, &p.pF, &p.pA, &p.pS);
    if (rc) return rc;
  }
  p.P = (float*)(((uintptr_t)kp_ws + 255) & ~(uintptr_t)255);
  p.GM = p.P + (size_t)p.rows_pad * p.GB;
  mh_prof_mark(MH_PROF_KEYPOINTS, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(k_kp_p, dim3(p.rows_pad / 32, G), dim3(KP_T), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_kp_mid, dim3(G), dim3(KP_T), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  if (p.pF) {
    hipLaunchKernelGGL(k_kp_gf, dim3(MH_FS / 32, G), dim3(KP_T), 0, (hipStream_t)stream, p);
    MH_LAUNCH_CHECK();
  }
  mh_prof_mark(MH_PROF_KEYPOINTS, 1, (hipStream_t)stream);
  return MH_OK;
}
