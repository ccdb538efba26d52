// Batched SMPL linear-blend skinning for gfx950: forward and hand-written backward.
// Replaces smpl.py:490-576 (lbs), :647-678 (rodrigues), :692-746 (kinematic chain) and the
// scale/translate of optimizer.py:702-703, plus everything autograd did for them.
//
// Data flow (B bodies in groups of 32; V vertices in tiles of 32):
//   k_pose_fwd   per body, 32 lanes = joints: rodrigues, shape-dependent joints, level-parallel
//                kinematic chain  ->  featT[g][k][32] (beta | R-I, transposed for the MFMA A
//                operand), A[b][24][3x4], scale[b]
//   k_skin_fwd   per (vertex tile, 4 body groups): v_posed = v_template + [S;P]^T.feat as a
//                K=217 f32 MFMA (32x32x2) contraction, then sparse skinning in the epilogue
//                (lane = vertex, 16 accumulator rows = bodies) -> coalesced 384-B row stores
//   k_skin_bwd   per (body group, vertex chunk): lane = (body, vertex) so the element-wise
//                adjoints ARE the MFMA A operands: gfeat += g_vposed x Dt, gA += gT x W
//   k_pose_bwd   per body: adjoint of the chain and of rodrigues, priors-free
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "mh_common.h"
#include "mh_raster_p.h"
#include "mh_experiment.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// =============================================================================================
// forward
// =============================================================================================
// (phase probes L_T0 / L_MARK / L_OUT: csrc/mh_experiment.h -- empty in the product build)

struct PoseFwdP {
  int B, NB, G;
  const float* betas;
  const float* poses;     // [B][72] axis-angle, or
  const float* rotmats;   // [B][24][3][3] rotation matrices (lbs(pose2rot=False), smpl.py:553-558); one of the two
  const float* xscale;
  const float* Jt;
  const float* JS;
  float* featT;   // [G][224][32]
  uint16_t* F16;  // [G][14][2][64][8] fp16 (hi, lo) terms of 2^8 x features, MFMA A operand of k_skin_fwd16
  float* A;       // [G*32][24][12]
  float* scale;   // [G*32]
  float* posed;   // [B][24][3] or null
  mh_tree tree;
  // mh_lbs_forward_proj: the per-body report slots of the skinning epilogue (null otherwise)
  int* bbox;
  int* bbox_prev;
  unsigned long long* lowkey;
  unsigned long long* lowkey_prev;
  int* moved;
  float* clear;            // zeroed here (the cycle's gradient buffer), or null
  unsigned long long clear_n;
};

__device__ __forceinline__ void mh_joint_rest(const float* Jt, const float* JS, const float* beta, int j, float J[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = Jt[j * 3 + c];
#pragma unroll
    for (int l = 0; l < MH_NUM_BETAS; ++l) a = fmaf(JS[(j * 3 + c) * MH_NUM_BETAS + l], beta[l], a);
    J[c] = a;
  }
}

// G = Gp o [R | rel]   (row-major 3x4)
__device__ __forceinline__ void mh_compose(const float* Gp, const float R[9], const float rel[3], float G[12]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      G[r * 4 + c] = fmaf(Gp[r * 4 + 2], R[6 + c], fmaf(Gp[r * 4 + 1], R[3 + c], Gp[r * 4 + 0] * R[c]));
    G[r * 4 + 3] = fmaf(Gp[r * 4 + 2], rel[2], fmaf(Gp[r * 4 + 1], rel[1], Gp[r * 4 + 0] * rel[0])) + Gp[r * 4 + 3];
  }
}

__global__ __launch_bounds__(256) void k_pose_fwd(PoseFwdP p) {
  __shared__ float sG[8][MH_NJ][12];
  __shared__ float sJ[8][MH_NJ][3];
  if (p.clear)           // the cycle's gradient buffer: everything that adds into it is launched behind this kernel
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < p.clear_n; i += (size_t)gridDim.x * 256) p.clear[i] = 0.f;
  const int bl = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int b = blockIdx.x * 8 + bl;
  const bool valid = b < p.B, act = j < MH_NJ;
  float beta[MH_NUM_BETAS];
#pragma unroll
  for (int l = 0; l < MH_NUM_BETAS; ++l) beta[l] = valid ? p.betas[(size_t)(b % p.NB) * MH_NUM_BETAS + l] : 0.f;
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, J[3] = {0, 0, 0};
  if (valid && act) {
    mh_joint_rest(p.Jt, p.JS, beta, j, J);
    if (p.rotmats) {   // given rotations are used for all 24 joints, hands included (smpl.py:554-555)
#pragma unroll
      for (int e = 0; e < 9; ++e) R[e] = p.rotmats[((size_t)b * MH_NJ + j) * 9 + e];
    } else if (j < 22) {   // hands (22,23) stay identity: smpl.py:542-546
      float r[3] = {p.poses[(size_t)b * 72 + 3 * j], p.poses[(size_t)b * 72 + 3 * j + 1], p.poses[(size_t)b * 72 + 3 * j + 2]};
      mh_rodrigues(r, R);
    }
    sJ[bl][j][0] = J[0];
    sJ[bl][j][1] = J[1];
    sJ[bl][j][2] = J[2];
  }
  __syncthreads();
  float Gm[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const int par = act ? p.tree.parent[j] : -1;
  const int lev = act ? p.tree.level[j] : -1;
  float rel[3] = {J[0], J[1], J[2]};
  if (valid && act && j > 0) {
    rel[0] -= sJ[bl][par][0];
    rel[1] -= sJ[bl][par][1];
    rel[2] -= sJ[bl][par][2];
  }
  for (int l = 0; l <= p.tree.maxlevel; ++l) {
    if (valid && act && lev == l) {
      if (l == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          Gm[r * 4 + 0] = R[r * 3 + 0];
          Gm[r * 4 + 1] = R[r * 3 + 1];
          Gm[r * 4 + 2] = R[r * 3 + 2];
          Gm[r * 4 + 3] = J[r];
        }
      } else {
        float Gp[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) Gp[e] = sG[bl][par][e];
        mh_compose(Gp, R, rel, Gm);
      }
#pragma unroll
      for (int e = 0; e < 12; ++e) sG[bl][j][e] = Gm[e];
    }
    __syncthreads();
  }
  const int g = b >> 5, bi = b & 31;
  if (b < p.G * 32 && act) {
    float* Ao = p.A + ((size_t)b * MH_NJ + j) * 12;
    if (valid) {
      // A = [G.R | G.t - G.R.J]  (rest pose removed, smpl.py:743-744)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Ao[r * 4 + 0] = Gm[r * 4 + 0];
        Ao[r * 4 + 1] = Gm[r * 4 + 1];
        Ao[r * 4 + 2] = Gm[r * 4 + 2];
        float gj = fmaf(Gm[r * 4 + 2], J[2], fmaf(Gm[r * 4 + 1], J[1], Gm[r * 4 + 0] * J[0]));
        Ao[r * 4 + 3] = Gm[r * 4 + 3] - gj;
      }
      if (p.posed) {
        p.posed[((size_t)b * MH_NJ + j) * 3 + 0] = Gm[3];
        p.posed[((size_t)b * MH_NJ + j) * 3 + 1] = Gm[7];
        p.posed[((size_t)b * MH_NJ + j) * 3 + 2] = Gm[11];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 12; ++e) Ao[e] = 0.f;
    }
  }
  if (b < p.G * 32) {
    float* fT = p.featT + (size_t)g * MH_FS * 32 + bi;
    if (j < MH_NUM_BETAS) fT[j * 32] = beta[j];
    if (act && j >= 1) {
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        float id = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
        fT[(10 + (j - 1) * 9 + e) * 32] = valid ? (R[e] - id) : 0.f;   // smpl.py:547
      }
    }
    if (j >= 24 && j < 31) fT[(217 + (j - 24)) * 32] = 0.f;
    if (j == 0) p.scale[b] = (valid && p.xscale) ? powf(1.1f, p.xscale[b % p.NB]) : 1.f;   // optimizer.py:681
    if (p.F16) {
      // the same features as two fp16 terms in the A-operand layout of v_mfma_f32_32x32x16_f16:
      // lane = (k half) * 32 + body, eight consecutive k per lane
      auto put = [&](int k, float x) {
        const float xs = x * (float)(1 << MH_F16_FEAT_SHIFT);
        const _Float16 hi = (_Float16)xs;
        const _Float16 lo = (_Float16)(xs - (float)hi);
        _Float16* q = (_Float16*)p.F16 + ((((size_t)g * (MH_FS / 16) + (k >> 4)) * 2) * 64 + ((k >> 3) & 1) * 32 + bi) * 8 + (k & 7);
        q[0] = hi;
        q[64 * 8] = lo;
      };
      if (j < MH_NUM_BETAS) put(j, beta[j]);
      if (act && j >= 1) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          float id = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
          put(10 + (j - 1) * 9 + e, valid ? (R[e] - id) : 0.f);
        }
      }
      if (j >= 24 && j < 31) put(217 + (j - 24), 0.f);
    }
  }
  // report slots of the skinning epilogue (mh_fwd_proj): what the last launch left becomes the filter of this one -- only
  // COMPLETE entries, so that a launch nobody consumed (no fallback scan wrote the missing extreme back) does not wipe
  // the filter -- then the slots are reset
  if (p.bbox && valid && j == 31) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int unset = k < 2 ? 0x7fffffff : (int)0x80000000;
      const int c = p.bbox[(size_t)b * 4 + k];
      if (c != unset) p.bbox_prev[(size_t)b * 4 + k] = c;
      p.bbox[(size_t)b * 4 + k] = unset;
    }
    const unsigned long long lk = p.lowkey[b];
    if (lk) p.lowkey_prev[b] = lk;
    p.lowkey[b] = 0ull;
    p.moved[b] = 0;
    p.moved[p.B + b] = 0;
  }
}

struct SkinFwdP {
  int B, G, V, VP, nw;
  const float* featT;
  const float* A;
  const float* scale;
  const float* transl;   // [B][3] or null
  const float* D;
  const float* vt;
  const int* skidx;
  const float* skw;
  float* verts;          // [B][V][3]
  float* vposed;         // [B][V][3] or null
};

// One workgroup = one group of 32 bodies x four vertex tiles (one per wave).  The group's feature panel (224 x 32)
// and its 32 x 24 bone transforms are staged in LDS once and shared by the four waves: the MFMA A operand and the
// skinning blend of the epilogue then come from LDS, and the per-CU vector-memory path only carries the basis tiles
// (B operand, explicitly double buffered) and the output stores.  (Loading everything per wave from L1/L2 kept the
// texture-address path as busy as the matrix cores: 335 KB per wave against 21.5k MFMA cycles.)
// grid.x is padded to a multiple of 8 so that a vertex tile always lands on the same XCD and its basis tile stays in
// that XCD's L2 for all 25 groups.
__global__ __launch_bounds__(256, 2) void k_skin_fwd(SkinFwdP p) {
  __shared__ __attribute__((aligned(16))) float sF[MH_FS * 32];
  __shared__ __attribute__((aligned(16))) float sA[32 * MH_NJ * 12];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.y;
  const int ntiles = p.VP / 32;
  if ((int)blockIdx.x * 4 >= ntiles) return;
  // the feature panel is needed by the first MFMA; the bone transforms only by the epilogue: their loads are issued now
  // and parked in registers (9 x 16 bytes per lane) until the matrix phase is over
  constexpr int NA4 = 32 * MH_NJ * 12 / 4 / 256;      // = 9
  static_assert(32 * MH_NJ * 12 / 4 == NA4 * 256, "transform panel must split evenly over the workgroup");
  f32x4 ra[NA4];
  {
    const f32x4* srcF = (const f32x4*)(p.featT + (size_t)g * MH_FS * 32);
    for (int i = threadIdx.x; i < MH_FS * 32 / 4; i += 256) ((f32x4*)sF)[i] = srcF[i];
    const f32x4* srcA = (const f32x4*)(p.A + (size_t)g * 32 * MH_NJ * 12);
#pragma unroll
    for (int i = 0; i < NA4; ++i) ra[i] = srcA[threadIdx.x + i * 256];
  }
  __syncthreads();
  const int tile_ = blockIdx.x * 4 + wave;
  const bool has_tile = tile_ < ntiles;
  const int tile = has_tile ? tile_ : ntiles - 1;      // a wave without a tile still takes part in the barrier below
  const int v = tile * 32 + li;
  const float* fT = sF + li;
  // basis tile of this wave: [kg][c][lane][8]
  const f32x4* Dw = (const f32x4*)(p.D + (size_t)tile * (MH_KD / 16) * 3 * 64 * 8) + lane * 2;
  f32x16 ax = {0}, ay = {0}, az = {0};
  // K = 224 (217 used, the rest zero) in 14 groups of 8 two-k steps; the basis operands of group g+1 (six 16-byte
  // loads per lane) are fetched into the second register set while the 24 MFMAs of group g issue
  f32x4 rx[2][2], ry[2][2], rz[2][2];
  rx[0][0] = Dw[0]; rx[0][1] = Dw[1];
  ry[0][0] = Dw[128]; ry[0][1] = Dw[129];
  rz[0][0] = Dw[256]; rz[0][1] = Dw[257];
#pragma unroll
  for (int g8 = 0; g8 < 14; ++g8) {
    const int cur = g8 & 1, nxt = cur ^ 1;
    if (g8 + 1 < 14) {
      const f32x4* d = Dw + (size_t)(g8 + 1) * 3 * 128;
      rx[nxt][0] = d[0]; rx[nxt][1] = d[1];
      ry[nxt][0] = d[128]; ry[nxt][1] = d[129];
      rz[nxt][0] = d[256]; rz[nxt][1] = d[257];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a = fT[(2 * (8 * g8 + u) + lh) * 32];
      ax = MFMA32(a, rx[cur][u >> 2][u & 3], ax);
      ay = MFMA32(a, ry[cur][u >> 2][u & 3], ay);
      az = MFMA32(a, rz[cur][u >> 2][u & 3], az);
    }
  }
#pragma unroll
  for (int i = 0; i < NA4; ++i) ((f32x4*)sA)[threadIdx.x + i * 256] = ra[i];
  __syncthreads();
  if (!has_tile || v >= p.V) return;
  const float t0 = p.vt[(size_t)v * 3], t1 = p.vt[(size_t)v * 3 + 1], t2 = p.vt[(size_t)v * 3 + 2];
  int sj[4];
  float sw[4];
  const int nwl = p.nw < 4 ? p.nw : 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sj[k] = k < nwl ? p.skidx[(size_t)v * p.nw + k] : 0;
    sw[k] = k < nwl ? p.skw[(size_t)v * p.nw + k] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int b = g * 32 + row;
    if (b >= p.B) continue;
    const float vp0 = t0 + ax[r], vp1 = t1 + ay[r], vp2 = t2 + az[r];
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    const float* Ab = sA + row * MH_NJ * 12;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4* Aj = (const f32x4*)(Ab + sj[k] * 12);
      const f32x4 q0 = Aj[0], q1 = Aj[1], q2 = Aj[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T[e] = fmaf(sw[k], q0[e], T[e]);
        T[4 + e] = fmaf(sw[k], q1[e], T[4 + e]);
        T[8 + e] = fmaf(sw[k], q2[e], T[8 + e]);
      }
    }
    for (int k = 4; k < p.nw; ++k) {   // models with more than 4 bones per vertex
      const float w = p.skw[(size_t)v * p.nw + k];
      const float* Aj = Ab + p.skidx[(size_t)v * p.nw + k] * 12;
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Aj[e], T[e]);
    }
    const float x0 = fmaf(T[2], vp2, fmaf(T[1], vp1, T[0] * vp0)) + T[3];
    const float x1 = fmaf(T[6], vp2, fmaf(T[5], vp1, T[4] * vp0)) + T[7];
    const float x2 = fmaf(T[10], vp2, fmaf(T[9], vp1, T[8] * vp0)) + T[11];
    const float s = p.scale[b];
    float o0 = s * x0, o1 = s * x1, o2 = s * x2;
    if (p.transl) {
      o0 += p.transl[(size_t)b * 3];
      o1 += p.transl[(size_t)b * 3 + 1];
      o2 += p.transl[(size_t)b * 3 + 2];
    }
    float* o = p.verts + ((size_t)b * p.V + v) * 3;
    o[0] = o0;
    o[1] = o1;
    o[2] = o2;
    if (p.vposed) {
      float* q = p.vposed + ((size_t)b * p.V + v) * 3;
      q[0] = vp0;
      q[1] = vp1;
      q[2] = vp2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-fp16 forward (default): the K = 224 contraction v_posed = v_template + [S;P]^T.feat on the fp16 matrix pipe
// as three products of (hi, lo) terms per operand (mh_common.h), 16x the rate of the exact-fp32 MFMA form above, so
// that the kernel is bound by its stores instead of by the matrix phase.  Same tiling: one workgroup = one group of
// 32 bodies x four vertex tiles (one per wave); the group's feature panel (28 KB of fp16 terms) and its bone
// transforms (36 KB) are staged in LDS once.  Per 16-k step a wave issues 2 ds_read_b128 (A operand: both terms),
// six 16-byte global loads (B operand: three components x two terms, double buffered) and nine MFMAs.
// ---------------------------------------------------------------------------------------------------------------
// (round 5's timing-only ablation builds of this kernel -- no NDC store, no row loads, one k-step, ... : DESIGN App. A has the
// numbers -- lived here as FWD_ABL bits; they are gone from the product kernel)
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct SkinFwd16P {
  int B, G, V, VP, nw;
  int tpb;                 // vertex tiles per workgroup (<= FWD16_WAVES * FWD16_TPW): chosen by the host for an even load
  float unscale;           // 2^-(feature shift + basis shift)
  const uint16_t* F16;
  const float* A;
  const float* scale;
  const float* transl;
  const uint16_t* D16;
  const float* vt;
  const int* skidx;
  const float* skw;
  float* verts;
  float* vposed;
  mh_fwd_proj P;           // PROJ instantiations only
};

#ifndef FWD16_STAGES
#define FWD16_STAGES 2
#endif
#ifndef FWD16_WAVES
#define FWD16_WAVES 8
#endif
#ifndef FWD16_TPW
#define FWD16_TPW 2            // vertex tiles per wave: the staged panels serve two tiles (338 workgroups instead of 675; 1 / 2 / 3: 69.8 / 67.4 / 67.4 us)
#endif
#define FWD16_SF_BYTES ((MH_FS / 16) * 2 * 64 * 16)       // 28672: feature terms
#define FWD16_SA_BYTES (32 * MH_NJ * 12 * 4)              // 36864: bone transforms
#define FWD16_LDS_BYTES (FWD16_SF_BYTES + FWD16_SA_BYTES + 3 * 32 * 16)   // + (scale, translation), report thresholds per body

// order-preserving integer images of a float: signed (box extremes) and unsigned (high word of the lowest-vertex key)
__device__ __forceinline__ int mh_ord(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float mh_unord(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }
__device__ __forceinline__ unsigned mh_ordu(float x) { const unsigned u = __float_as_uint(x); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float mh_unordu(unsigned o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu)); }

typedef float f32x3 __attribute__((ext_vector_type(3)));

// FULL: every body of the group exists (no store guards); NW4: at most four bones per vertex (SMPL).
// The epilogue is written for instruction count: per accumulator row 13 LDS reads whose row part is an immediate
// offset (per-lane bases are set up once), 63 fp32 FMAs and two 12-byte stores addressed by a wave-uniform row base
// plus one 32-bit lane offset.
// PROJ (mh_lbs_forward_proj, the "LBS + projection" kernel): the epilogue also projects the vertex to NDC (third row store),
// tests it against the pixel row its body's face lists were sorted for, and reports it when it lies beyond the previous
// launch's box / lowest-vertex extremes minus a slack -- plain compares against per-body thresholds in LDS, atomics only
// from the few vertices near an extreme, no cross-lane reduction (include/mhmocap_hip.h, mh_fwd_proj).
template <bool FULL, bool NW4, bool PROJ>
#ifndef FWD16_MINB
#define FWD16_MINB (2 * FWD16_WAVES / 4)
#endif
__global__ __launch_bounds__(FWD16_WAVES * 64, FWD16_MINB) void k_skin_fwd16(SkinFwd16P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  f16x8* sF = (f16x8*)smem16;                                           // [14][2][64]
  float* sA = (float*)(smem16 + FWD16_SF_BYTES);                        // [32][24][12]
  f32x4* sS = (f32x4*)(smem16 + FWD16_SF_BYTES + FWD16_SA_BYTES);       // [32] (scale, tx, ty, tz)
  f32x4* sTH = sS + 32;                                                 // [32] report thresholds: min x, min y, max x, max y (NDC)
  f32x4* sLY = sTH + 32;                                                // [32] [0] = report threshold of the lowest vertex (camera y)
                                                                        // (three arrays of one stride: one per-lane base serves all)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.y;
  const int ntiles = p.VP / 32;
  if ((int)blockIdx.x * p.tpb >= ntiles) return;
  L_T0();
  {
    const f32x4* srcF = (const f32x4*)(p.F16 + (size_t)g * (MH_FS / 16) * 2 * 64 * 8);
    for (int i = threadIdx.x; i < FWD16_SF_BYTES / 16; i += FWD16_WAVES * 64) ((f32x4*)sF)[i] = srcF[i];
    const f32x4* srcA = (const f32x4*)(p.A + (size_t)g * 32 * MH_NJ * 12);
    for (int i = threadIdx.x; i < FWD16_SA_BYTES / 16; i += FWD16_WAVES * 64) ((f32x4*)sA)[i] = srcA[i];
    if (threadIdx.x < 32) {
      const int b = g * 32 + threadIdx.x;
      f32x4 q = {1.f, 0.f, 0.f, 0.f};
      if (b < p.B) {
        q[0] = p.scale[b];
        if (p.transl) { q[1] = p.transl[(size_t)b * 3]; q[2] = p.transl[(size_t)b * 3 + 1]; q[3] = p.transl[(size_t)b * 3 + 2]; }
      }
      sS[threadIdx.x] = q;
      if (PROJ) {
        // nothing reports for a padding body; an unset / garbage previous extreme gives a NaN or arbitrary threshold:
        // whatever then reports is still exact per slot (a slot is either the true extreme or stays unset)
        f32x4 th = {-3e38f, -3e38f, 3e38f, 3e38f};
        float ly = 3e38f;
        if (b < p.B) {
          const int4 pb = *(const int4*)(p.P.bbox_prev + (size_t)b * 4);
          th[0] = mh_unord(pb.x) + p.P.slack_ndc; th[1] = mh_unord(pb.y) + p.P.slack_ndc;
          th[2] = mh_unord(pb.z) - p.P.slack_ndc; th[3] = mh_unord(pb.w) - p.P.slack_ndc;
          ly = mh_unordu((unsigned)(p.P.lowkey_prev[b] >> 32)) - p.P.slack_y;
        }
        sTH[threadIdx.x] = th;
        sLY[threadIdx.x] = (f32x4){ly, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  L_MARK(0);
  // tile k of the workgroup goes to wave k % WAVES: with tpb = 11 three waves take two tiles, five take one
  for (int rep_ = 0; rep_ < FWD16_TPW; ++rep_) {
  if (wave + rep_ * FWD16_WAVES >= p.tpb) break;
  const int tile = blockIdx.x * p.tpb + wave + rep_ * FWD16_WAVES;
  if (tile >= ntiles) break;
  const int v = tile * 32 + li;
  // this lane's vertex constants: issued before the matrix phase, consumed after it
  const bool vok = v < p.V;
  const int vc = vok ? v : p.V - 1;
  const f32x3 tv = *(const f32x3*)(p.vt + (size_t)vc * 3);
  int sj[4] = {0, 0, 0, 0};
  float sw[4] = {0.f, 0.f, 0.f, 0.f};
  if (NW4 && p.nw == 4) {
    const int4 j4 = *(const int4*)(p.skidx + (size_t)vc * 4);
    const f32x4 w4 = *(const f32x4*)(p.skw + (size_t)vc * 4);
    sj[0] = j4.x; sj[1] = j4.y; sj[2] = j4.z; sj[3] = j4.w;
    sw[0] = w4[0]; sw[1] = w4[1]; sw[2] = w4[2]; sw[3] = w4[3];
  } else {
    const int nwl = p.nw < 4 ? p.nw : 4;
    for (int k = 0; k < nwl; ++k) {
      const int jj = p.skidx[(size_t)vc * p.nw + k];
      const float ww = p.skw[(size_t)vc * p.nw + k];
      if (k == 0) { sj[0] = jj; sw[0] = ww; }
      if (k == 1) { sj[1] = jj; sw[1] = ww; }
      if (k == 2) { sj[2] = jj; sw[2] = ww; }
      if (k == 3) { sj[3] = jj; sw[3] = ww; }
    }
  }
  // basis tile of this wave: [s][c][term][lane] in 16-byte units
  const f16x8* Dw = (const f16x8*)p.D16 + (size_t)tile * (MH_KD / 16) * 6 * 64 + lane;
  f32x16 ax = {0}, ay = {0}, az = {0};
  // B operand ring: FWD16_STAGES k-steps (6 KB per wave each) in flight -- one step ahead left the phase latency-bound
  // (8 waves x 6 KB per CU against ~1.5 us of L2 latency under load)
  f16x8 bq[FWD16_STAGES][6];
#pragma unroll
  for (int st = 0; st < FWD16_STAGES - 1; ++st)
#pragma unroll
    for (int i = 0; i < 6; ++i) bq[st][i] = Dw[(st * 6 + i) * 64];
#pragma unroll
  for (int s16 = 0; s16 < MH_KD / 16; ++s16) {
    const int cur = s16 % FWD16_STAGES, nxt = (s16 + FWD16_STAGES - 1) % FWD16_STAGES;
    if (s16 + FWD16_STAGES - 1 < MH_KD / 16) {
#pragma unroll
      for (int i = 0; i < 6; ++i) bq[nxt][i] = Dw[((s16 + FWD16_STAGES - 1) * 6 + i) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the ring: the scheduler otherwise sinks each load next to its first use
    const f16x8 ah = sF[(s16 * 2) * 64 + lane], al = sF[(s16 * 2 + 1) * 64 + lane];
    ax = MFMA_F16(ah, bq[cur][0], ax);
    ay = MFMA_F16(ah, bq[cur][2], ay);
    az = MFMA_F16(ah, bq[cur][4], az);
    ax = MFMA_F16(ah, bq[cur][1], ax);
    ay = MFMA_F16(ah, bq[cur][3], ay);
    az = MFMA_F16(ah, bq[cur][5], az);
    ax = MFMA_F16(al, bq[cur][0], ax);
    ay = MFMA_F16(al, bq[cur][2], ay);
    az = MFMA_F16(al, bq[cur][4], az);
    __builtin_amdgcn_sched_barrier(0);
  }
  L_MARK(1);
  if (!vok) continue;
  const float us = p.unscale;
  // per-lane LDS bases (bytes): the row part (r&3)+8(r>>2) is a compile-time offset, the half-wave part 4*lh is here
  const unsigned char* aB[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) aB[k] = (const unsigned char*)sA + (4 * lh) * (MH_NJ * 48) + sj[k] * 48;
  const unsigned char* sB = (const unsigned char*)sS + (4 * lh) * 16;
  const unsigned lane_off = (unsigned)(4 * lh * p.V + v) * 12u;          // bytes inside a group's row block (< 2^32)
  const size_t row_bytes = (size_t)p.V * 12;
  const unsigned row_b32 = (unsigned)p.V * 12u;   // every address below: wave-uniform group base + ONE 32-bit lane offset (saddr form)
  unsigned char* const vg = (unsigned char*)p.verts + (size_t)g * 32 * row_bytes;
  unsigned char* const qg = p.vposed ? (unsigned char*)p.vposed + (size_t)g * 32 * row_bytes : nullptr;
  unsigned char* const ng = PROJ ? (unsigned char*)p.P.ndc + (size_t)g * 32 * row_bytes : nullptr;
  const unsigned char* const rbg = PROJ ? (const unsigned char*)p.P.rowb + (size_t)g * 32 * p.V * 4 : nullptr;
  // the sixteen rows' pixel rows at the last face sort: requested HERE, ahead of the row stores -- a load issued between
  // the stores would be waited for with vmcnt(0), i.e. together with every store in flight (one memory round trip per row)
  float rbv[16];
  if (PROJ) {
    // (rows of padding bodies read the group's last real row: the array ends with the last body)
    const int last_row = FULL ? 31 : min(31, p.B - 1 - g * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const unsigned off4 = FULL ? (unsigned)(4 * lh * p.V + v) * 4u + (unsigned)((r & 3) + 8 * (r >> 2)) * ((unsigned)p.V * 4u)
                                 : (unsigned)(min(row, last_row) * p.V + v) * 4u;
      rbv[r] = *(const float*)(rbg + (size_t)off4);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    constexpr int dummy = 0; (void)dummy;
    const int row0 = (r & 3) + 8 * (r >> 2);                              // + 4*lh = accumulator row = body in group
    const float vp0 = fmaf(ax[r], us, tv[0]), vp1 = fmaf(ay[r], us, tv[1]), vp2 = fmaf(az[r], us, tv[2]);
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4* Aj = (const f32x4*)(aB[k] + row0 * (MH_NJ * 48));
      const f32x4 q0 = Aj[0], q1 = Aj[1], q2 = Aj[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T[e] = fmaf(sw[k], q0[e], T[e]);
        T[4 + e] = fmaf(sw[k], q1[e], T[4 + e]);
        T[8 + e] = fmaf(sw[k], q2[e], T[8 + e]);
      }
    }
    if (!NW4) {
      const float* Ab = sA + (row0 + 4 * lh) * MH_NJ * 12;
      for (int k = 4; k < p.nw; ++k) {   // models with more than 4 bones per vertex
        const float w = p.skw[(size_t)v * p.nw + k];
        const float* Aj = Ab + p.skidx[(size_t)v * p.nw + k] * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Aj[e], T[e]);
      }
    }
    const f32x4 st = *(const f32x4*)(sB + row0 * 16);                    // (scale, translation) of this body
    const float x0 = fmaf(T[2], vp2, fmaf(T[1], vp1, T[0] * vp0)) + T[3];
    const float x1 = fmaf(T[6], vp2, fmaf(T[5], vp1, T[4] * vp0)) + T[7];
    const float x2 = fmaf(T[10], vp2, fmaf(T[9], vp1, T[8] * vp0)) + T[11];
    const f32x3 o = {fmaf(st[0], x0, st[1]), fmaf(st[0], x1, st[2]), fmaf(st[0], x2, st[3])};
    if (FULL || g * 32 + row0 + 4 * lh < p.B) {
      const unsigned off_r = lane_off + (unsigned)row0 * row_b32;
      *(f32x3*)(vg + (size_t)off_r) = o;
      if (qg) {
        const f32x3 q = {vp0, vp1, vp2};
        *(f32x3*)(qg + (size_t)off_r) = q;
      }
      if (PROJ) {
        // the same three roundings per coordinate as k_raster_prepare's own projection (multiply, IEEE divide, add)
        const float Zc = o[2];
        const float xn = p.P.s * (-o[0]) / Zc + p.P.w1, yn = p.P.s * (-o[1]) / Zc + p.P.h1;
        const f32x3 nd = {xn, yn, Zc};
        *(f32x3*)(ng + (size_t)off_r) = nd;
        const float drow = fabsf(fmaf(-yn, p.P.rk, p.P.ra) - rbv[r]);
        const bool mv = !(drow < p.P.thr_soft);                                    // NaN-safe: anything odd rebuilds
        const f32x4 th = *(const f32x4*)(sB + 512 + row0 * 16);
        const float ly = *(const float*)(sB + 1024 + row0 * 16);
        const bool zok = Zc > 1e-8f;
        const bool c0 = zok && xn < th[0], c1 = zok && yn < th[1], c2 = zok && xn > th[2], c3 = zok && yn > th[3];
        const bool c4 = o[1] > ly;
        // ONE wave-uniform branch per row (a scalar test of the combined lane mask): nothing below it is on the path of a row
        // in which no vertex reports -- most rows
        if (__builtin_amdgcn_ballot_w64(c0 || c1 || c2 || c3 || c4 || mv) != 0ull) {
          // (the asm keeps the sixteen rows' slot addresses from being computed ahead of the matrix phase and spilled)
          int b = g * 32 + row0 + 4 * lh;
          asm volatile("" : "+v"(b));
          int* bb = p.P.bbox + (size_t)b * 4;
          if (c0) atomicMin(bb, mh_ord(xn));
          if (c1) atomicMin(bb + 1, mh_ord(yn));
          if (c2) atomicMax(bb + 2, mh_ord(xn));
          if (c3) atomicMax(bb + 3, mh_ord(yn));
          if (c4) atomicMax(p.P.lowkey + b, ((unsigned long long)mh_ordu(o[1] + 0.0f) << 32) | (unsigned)~(unsigned)v);
          // 2: out of the band the kept lists cover -- sorted on the chain, now; 1: on its way out -- sorted beside this cycle's
          // gradient kernel, for the next launch (k_raster_prepare / r_deferred_sorts)
          // (plain stores of a constant: every vertex of a moving body reports, same-address atomics would queue up in L2)
          if (mv) p.P.moved[(drow < p.P.thr ? p.B : 0) + b] = 1;
        }
      }
    }
  }
  L_MARK(2);
  }   // tiles of this wave
  L_OUT(0, 3);
}

// ---------------------------------------------------------------------------------------------------------------
// The same kernel, SOFTWARE-PIPELINED over a wave's tiles (round 6, VERDICT r05 item 2 / DESIGN 10).  The kernel above runs a
// tile's matrix phase (L2 -> matrix pipe) and then its epilogue (LDS gathers, fp32 FMAs, three row stores); all waves of a CU
// start behind the same barrier and walk through the same phases in step, so at any moment ONE pipe of the CU works: its time
// is the sum of the pipes' busy times (round 5: matrix 17 + blend 9 + stores 33 + arithmetic 21 us = 80 against 72 measured).
// Here a wave carries TWO accumulator sets: while the 9 MFMAs of k-step s of tile i + 1 run, the wave issues row s of tile i's
// epilogue (16 rows over the 14 k-steps + 2 slots), so the matrix pipe, the VALU, the LDS and the store path work at the
// same time within every wave.  256 registers (two waves per SIMD, one workgroup per CU): 22 tiles per workgroup, three per
// wave, i.e. per wave  M0 | M1+E0 | M2+E1 | E2  instead of  M0 E0 M1 E1 M2 E2.
// Order inside a slot (the vmcnt caution of DESIGN 10): the NEXT k-step's basis loads are issued FIRST, then the MFMAs of this
// step, then the row's stores -- a load queued behind stores would be waited for together with them.
// Same arithmetic, operand for operand, as k_skin_fwd16: the two must agree bit for bit (tests/test_fwd_proj_gpu.py).
// What it gives (measured, C3, with k_pose_fwd's 12 us): 80.0 us against 82.4 -- not the 47 us the sum-of-pipes picture promised.
// The k-step's wait for its operands is s_waitcnt vmcnt(6), "everything but the six loads just issued": with the previous
// slot's row stores in flight that is a wait for those stores to be acknowledged.  A hand-counted vmcnt(9) (operands issued
// through inline assembly: "the six loads and the three stores behind them may stay in flight") ran in 72 us -- and returned
// NaN vertices: on gfx950 stores retire OUT OF ORDER with respect to older loads, so a count that lets three younger
// operations stay in flight does not say the loads have landed.  One counter for loads and stores: a wave cannot wait for a
// load without waiting for the stores it has issued since.  (DESIGN 3.2 / App. A.)
template <bool FULL, bool NW4, bool PROJ>
__global__ __launch_bounds__(FWD16_WAVES * 64, 2) void k_skin_fwd16p(SkinFwd16P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  f16x8* sF = (f16x8*)smem16;                                           // [14][2][64]
  float* sA = (float*)(smem16 + FWD16_SF_BYTES);                        // [32][24][12]
  f32x4* sS = (f32x4*)(smem16 + FWD16_SF_BYTES + FWD16_SA_BYTES);       // [32] (scale, tx, ty, tz)
  f32x4* sTH = sS + 32;                                                 // [32] report thresholds: min x, min y, max x, max y (NDC)
  f32x4* sLY = sTH + 32;                                                // [32] [0] = report threshold of the lowest vertex (camera y)
                                                                        // (three arrays of one stride: one per-lane base serves all)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.y;
  const int ntiles = p.VP / 32;
  if ((int)blockIdx.x * p.tpb >= ntiles) return;
  L_T0();
  {
    const f32x4* srcF = (const f32x4*)(p.F16 + (size_t)g * (MH_FS / 16) * 2 * 64 * 8);
    for (int i = threadIdx.x; i < FWD16_SF_BYTES / 16; i += FWD16_WAVES * 64) ((f32x4*)sF)[i] = srcF[i];
    const f32x4* srcA = (const f32x4*)(p.A + (size_t)g * 32 * MH_NJ * 12);
    for (int i = threadIdx.x; i < FWD16_SA_BYTES / 16; i += FWD16_WAVES * 64) ((f32x4*)sA)[i] = srcA[i];
    if (threadIdx.x < 32) {
      const int b = g * 32 + threadIdx.x;
      f32x4 q = {1.f, 0.f, 0.f, 0.f};
      if (b < p.B) {
        q[0] = p.scale[b];
        if (p.transl) { q[1] = p.transl[(size_t)b * 3]; q[2] = p.transl[(size_t)b * 3 + 1]; q[3] = p.transl[(size_t)b * 3 + 2]; }
      }
      sS[threadIdx.x] = q;
      if (PROJ) {
        // nothing reports for a padding body; an unset / garbage previous extreme gives a NaN or arbitrary threshold:
        // whatever then reports is still exact per slot (a slot is either the true extreme or stays unset)
        f32x4 th = {-3e38f, -3e38f, 3e38f, 3e38f};
        float ly = 3e38f;
        if (b < p.B) {
          const int4 pb = *(const int4*)(p.P.bbox_prev + (size_t)b * 4);
          th[0] = mh_unord(pb.x) + p.P.slack_ndc; th[1] = mh_unord(pb.y) + p.P.slack_ndc;
          th[2] = mh_unord(pb.z) - p.P.slack_ndc; th[3] = mh_unord(pb.w) - p.P.slack_ndc;
          ly = mh_unordu((unsigned)(p.P.lowkey_prev[b] >> 32)) - p.P.slack_y;
        }
        sTH[threadIdx.x] = th;
        sLY[threadIdx.x] = (f32x4){ly, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  L_MARK(0);
  const float us = p.unscale;
  const unsigned char* const sB = (const unsigned char*)sS + (4 * lh) * 16;
  const size_t row_bytes = (size_t)p.V * 12;
  const unsigned row_b32 = (unsigned)p.V * 12u;
  unsigned char* const vg = (unsigned char*)p.verts + (size_t)g * 32 * row_bytes;
  unsigned char* const qg = p.vposed ? (unsigned char*)p.vposed + (size_t)g * 32 * row_bytes : nullptr;
  unsigned char* const ng = PROJ ? (unsigned char*)p.P.ndc + (size_t)g * 32 * row_bytes : nullptr;
  const unsigned char* const rbg = PROJ ? (const unsigned char*)p.P.rowb + (size_t)g * 32 * p.V * 4 : nullptr;
  // tiles of this wave: wave, wave + WAVES, ... below tpb (and below the model's tile count)
  int nt = 0;
  for (int k = wave; k < p.tpb && (int)blockIdx.x * p.tpb + k < ntiles; k += FWD16_WAVES) ++nt;
  // the tile whose EPILOGUE runs in this pass (accumulators of the previous pass) ...
  f32x16 ax = {0}, ay = {0}, az = {0};
  int v_c = 0;
  bool vok_c = false;
  f32x3 tv_c = {0.f, 0.f, 0.f};
  int sj_c[4] = {0, 0, 0, 0};
  float sw_c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it <= nt; ++it) {
    const bool dm = it < nt, de = it > 0;                   // wave-uniform: a matrix phase / an epilogue in this pass
    // ... and the tile whose MATRIX PHASE runs in this pass
    const int tile = blockIdx.x * p.tpb + wave + it * FWD16_WAVES;
    const int v_n = tile * 32 + li;
    const bool vok_n = dm && v_n < p.V;
    const int vc = vok_n ? v_n : p.V - 1;
    f32x3 tv_n = {0.f, 0.f, 0.f};
    int sj_n[4] = {0, 0, 0, 0};
    float sw_n[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 nx = {0}, ny = {0}, nz = {0};
    f16x8 bq[2][6];
    const f16x8* Dw = (const f16x8*)p.D16 + (size_t)(dm ? tile : 0) * (MH_KD / 16) * 6 * 64 + lane;
    if (dm) {
      tv_n = *(const f32x3*)(p.vt + (size_t)vc * 3);
      if (NW4 && p.nw == 4) {
        const int4 j4 = *(const int4*)(p.skidx + (size_t)vc * 4);
        const f32x4 w4 = *(const f32x4*)(p.skw + (size_t)vc * 4);
        sj_n[0] = j4.x; sj_n[1] = j4.y; sj_n[2] = j4.z; sj_n[3] = j4.w;
        sw_n[0] = w4[0]; sw_n[1] = w4[1]; sw_n[2] = w4[2]; sw_n[3] = w4[3];
      } else {
        const int nwl = p.nw < 4 ? p.nw : 4;
        for (int k = 0; k < nwl; ++k) {
          const int jj = p.skidx[(size_t)vc * p.nw + k];
          const float ww = p.skw[(size_t)vc * p.nw + k];
          if (k == 0) { sj_n[0] = jj; sw_n[0] = ww; }
          if (k == 1) { sj_n[1] = jj; sw_n[1] = ww; }
          if (k == 2) { sj_n[2] = jj; sw_n[2] = ww; }
          if (k == 3) { sj_n[3] = jj; sw_n[3] = ww; }
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) bq[0][i] = Dw[i * 64];
    }
    // epilogue tile: per-lane LDS bases, the sixteen rows' pixel rows at the last face sort (requested ahead of every store)
    const unsigned char* aB[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) aB[k] = (const unsigned char*)sA + (4 * lh) * (MH_NJ * 48) + sj_c[k] * 48;
    const unsigned lane_off = (unsigned)(4 * lh * p.V + v_c) * 12u;
    const bool epi = de && vok_c;                           // (lanes beyond V in the model's last tile: no epilogue)
    float rbv[16];
    if (PROJ && epi) {
      const int last_row = FULL ? 31 : min(31, p.B - 1 - g * 32);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const unsigned off4 = FULL ? (unsigned)(4 * lh * p.V + v_c) * 4u + (unsigned)((r & 3) + 8 * (r >> 2)) * ((unsigned)p.V * 4u)
                                   : (unsigned)(min(row, last_row) * p.V + v_c) * 4u;
        rbv[r] = *(const float*)(rbg + (size_t)off4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {                          // slot r: k-step r of the matrix tile (r < 14) + row r of the epilogue tile
      if (r < MH_KD / 16 && dm) {
        const int cur = r & 1, nxt = cur ^ 1;
        if (r + 1 < MH_KD / 16) {
#pragma unroll
          for (int i = 0; i < 6; ++i) bq[nxt][i] = Dw[((r + 1) * 6 + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);                  // (the next step's loads stay ahead of this slot's stores)
        const f16x8 ah = sF[(r * 2) * 64 + lane], al = sF[(r * 2 + 1) * 64 + lane];
        nx = MFMA_F16(ah, bq[cur][0], nx);
        ny = MFMA_F16(ah, bq[cur][2], ny);
        nz = MFMA_F16(ah, bq[cur][4], nz);
        nx = MFMA_F16(ah, bq[cur][1], nx);
        ny = MFMA_F16(ah, bq[cur][3], ny);
        nz = MFMA_F16(ah, bq[cur][5], nz);
        nx = MFMA_F16(al, bq[cur][0], nx);
        ny = MFMA_F16(al, bq[cur][2], ny);
        nz = MFMA_F16(al, bq[cur][4], nz);
      }
      if (epi) {
        const int row0 = (r & 3) + 8 * (r >> 2);                              // + 4*lh = accumulator row = body in group
        const float vp0 = fmaf(ax[r], us, tv_c[0]), vp1 = fmaf(ay[r], us, tv_c[1]), vp2 = fmaf(az[r], us, tv_c[2]);
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4* Aj = (const f32x4*)(aB[k] + row0 * (MH_NJ * 48));
          const f32x4 q0 = Aj[0], q1 = Aj[1], q2 = Aj[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T[e] = fmaf(sw_c[k], q0[e], T[e]);
            T[4 + e] = fmaf(sw_c[k], q1[e], T[4 + e]);
            T[8 + e] = fmaf(sw_c[k], q2[e], T[8 + e]);
          }
        }
        if (!NW4) {
          const float* Ab = sA + (row0 + 4 * lh) * MH_NJ * 12;
          for (int k = 4; k < p.nw; ++k) {   // models with more than 4 bones per vertex
            const float w = p.skw[(size_t)v_c * p.nw + k];
            const float* Aj = Ab + p.skidx[(size_t)v_c * p.nw + k] * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Aj[e], T[e]);
          }
        }
        const f32x4 st = *(const f32x4*)(sB + row0 * 16);                    // (scale, translation) of this body
        const float x0 = fmaf(T[2], vp2, fmaf(T[1], vp1, T[0] * vp0)) + T[3];
        const float x1 = fmaf(T[6], vp2, fmaf(T[5], vp1, T[4] * vp0)) + T[7];
        const float x2 = fmaf(T[10], vp2, fmaf(T[9], vp1, T[8] * vp0)) + T[11];
        const f32x3 o = {fmaf(st[0], x0, st[1]), fmaf(st[0], x1, st[2]), fmaf(st[0], x2, st[3])};
        if (FULL || g * 32 + row0 + 4 * lh < p.B) {
          const unsigned off_r = lane_off + (unsigned)row0 * row_b32;
          *(f32x3*)(vg + (size_t)off_r) = o;
          if (qg) {
            const f32x3 q = {vp0, vp1, vp2};
            *(f32x3*)(qg + (size_t)off_r) = q;
          }
          if (PROJ) {
            const float Zc = o[2];
            const float xn = p.P.s * (-o[0]) / Zc + p.P.w1, yn = p.P.s * (-o[1]) / Zc + p.P.h1;
            const f32x3 nd = {xn, yn, Zc};
            *(f32x3*)(ng + (size_t)off_r) = nd;
            const float drow = fabsf(fmaf(-yn, p.P.rk, p.P.ra) - rbv[r]);
            const bool mv = !(drow < p.P.thr_soft);
            const f32x4 th = *(const f32x4*)(sB + 512 + row0 * 16);
            const float ly = *(const float*)(sB + 1024 + row0 * 16);
            const bool zok = Zc > 1e-8f;
            const bool c0 = zok && xn < th[0], c1 = zok && yn < th[1], c2 = zok && xn > th[2], c3 = zok && yn > th[3];
            const bool c4 = o[1] > ly;
            if (__builtin_amdgcn_ballot_w64(c0 || c1 || c2 || c3 || c4 || mv) != 0ull) {
              int b = g * 32 + row0 + 4 * lh;
              asm volatile("" : "+v"(b));
              int* bb = p.P.bbox + (size_t)b * 4;
              if (c0) atomicMin(bb, mh_ord(xn));
              if (c1) atomicMin(bb + 1, mh_ord(yn));
              if (c2) atomicMax(bb + 2, mh_ord(xn));
              if (c3) atomicMax(bb + 3, mh_ord(yn));
              if (c4) atomicMax(p.P.lowkey + b, ((unsigned long long)mh_ordu(o[1] + 0.0f) << 32) | (unsigned)~(unsigned)v_c);
              if (mv) p.P.moved[(drow < p.P.thr ? p.B : 0) + b] = 1;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the matrix tile of this pass is the epilogue tile of the next
    ax = nx; ay = ny; az = nz;
    v_c = v_n; vok_c = vok_n; tv_c = tv_n;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sj_c[k] = sj_n[k]; sw_c[k] = sw_n[k]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same kernel as PRODUCER and CONSUMER waves (round 6, second half).  What the software pipeline above
// ran into: a wave has ONE vmcnt for loads and stores and stores retire out of order with loads, so a wave that stores cannot
// wait for a load without waiting for its stores to be acknowledged -- however its instructions are interleaved.  Here the two
// never meet in one wave: per SIMD one producer wave (basis loads, feature fragments from LDS, the 126 MFMAs of a tile; nothing
// stored to memory) and one consumer wave (bone blend from LDS, fp32 epilogue, the three row stores and the reports; nothing
// loaded from memory).  A tile goes from producer w to consumer w + 4 through one 19-KB LDS slot: the 48 accumulator words per
// lane, and what the epilogue needs from memory (template vertex, four bones and weights, the sixteen rows' pixel rows at the
// last face sort), loaded by the PRODUCER while its matrix phase waits for operands.  Hand-off: one flag word per pair in LDS
// (0 = slot free, k + 1 = tile k of the pair is in it), polled with s_sleep; the consumer copies the slot to registers and
// frees it at once, so the producer computes tile k + 1 while tile k's epilogue runs.  The waves of a workgroup are resident
// together, so a wave polling another one of its workgroup always sees it advance; every poll is bounded all the same.
// Same arithmetic, operand for operand, as k_skin_fwd16: the three forms agree bit for bit (tests/test_fwd_proj_gpu.py).
//
// What it gives (C3, stand-alone with k_pose_fwd, eager launches): 109.6 us against 108.5 for the pipelined form -- NOT the
// default (mh_lbs_set_forward_pipeline(2) / MHHIP_FWD_PIPE=2 selects it).  What its timing-only variants measured (same
// scale, ~23 us of which are the pose kernel and launch gaps; the variants lived behind -DMH_FWDPC_DBG and are gone):
//   consumers hand the slot back unread (producers alone)            70    loads + matrix phase, one wave per SIMD
//   producers skip loads and matrix phase (consumers alone)          97    = 54 without the row stores, 72 with ONLY the stores
//   ... stores only, as one dense 768-byte run per instruction       69    the addresses do not matter,
//   ... stores only, 16 bytes per lane (a third more bytes)          67    nor do the bytes: ~110 cycles per wave-store per CU
// (a plain fill writes the same 198 MB in 31 us with a quarter of the instructions: 16 bytes per lane and one stream).
// So: (i) the three row stores are bound by their NUMBER -- 1008 wave-stores per CU at ~110 cycles each = 46 us -- not by the
// 198 MB (4.2 TB/s) and not by their scatter over 96 rows; (ii) the stores of one wave do not overlap the LDS gathers and
// FMAs of the OTHER consumer waves of the CU: epilogue arithmetic (31) + stores (46) = consumers alone (74), with no load
// and no vmcnt wait anywhere in a consumer (checked in the ISA) -- the "sum of the pipes" of DESIGN 10 is not the wave's one
// counter (that was the pipelined form's problem, and this form removes it) but the CU's;
// (iii) a first cut polled the flags through a volatile pointer: every volatile access is fenced with s_waitcnt vmcnt(0),
// i.e. each consumer waited for all stores of its last tile to be acknowledged before it looked for the next one (123 us).
// What is left to take: 16-byte stores of rows repacked through LDS (36 instead of 48 wave-stores per tile: -12 us of
// stores for +40 % LDS instructions, net ~-6).  tools/fwd_probe.py times the three forms.
#ifndef FWDPC_STAGES
#define FWDPC_STAGES 3                 // k-steps of basis fragments in flight per producer (6 KB each)
#endif
#define FWDPC_Q 19                     // 16-byte words per lane in a slot: 12 accumulators, 4 pixel rows, 1 template vertex + index, bones, weights
#define FWDPC_SLOT_BYTES (FWDPC_Q * 64 * 16)
#define FWDPC_LDS_BYTES (FWD16_LDS_BYTES + 4 * FWDPC_SLOT_BYTES + 64)
#define FWDPC_SPIN (1 << 22)
#ifndef FWDPC_NC
#define FWDPC_NC 2                     // consumer waves per producer wave (tiles of a pair alternate between them)
#endif
#define FWDPC_THREADS (256 * (1 + FWDPC_NC))
template <bool FULL, bool NW4>
__global__ __launch_bounds__(FWDPC_THREADS) void k_skin_fwd16pc(SkinFwd16P p) {
  constexpr bool PROJ = true;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  f16x8* sF = (f16x8*)smem16;                                           // [14][2][64]
  float* sA = (float*)(smem16 + FWD16_SF_BYTES);                        // [32][24][12]
  f32x4* sS = (f32x4*)(smem16 + FWD16_SF_BYTES + FWD16_SA_BYTES);       // [32] (scale, tx, ty, tz)
  f32x4* sTH = sS + 32;
  f32x4* sLY = sTH + 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.y;
  const int ntiles = p.VP / 32;
  if ((int)blockIdx.x * p.tpb >= ntiles) return;
  const int pr = wave & 3;
  f32x4* const slot = (f32x4*)(smem16 + FWD16_LDS_BYTES + (size_t)pr * FWDPC_SLOT_BYTES) + lane;      // [q][lane]
  // (relaxed workgroup-scope atomics, NOT volatile: a volatile access is fenced with s_waitcnt vmcnt(0) -- a consumer would wait for
  // every store of its last tile to be acknowledged before it looks at the flag; the LDS orders its own operations per wave, the
  // explicit lgkmcnt waits and compiler barriers below are all the hand-off needs)
  int* const flag = (int*)(smem16 + FWD16_LDS_BYTES + 4 * FWDPC_SLOT_BYTES) + pr;
#define FWDPC_FLAG() __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FWDPC_SET(x) __hip_atomic_store(flag, (x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  {
    const f32x4* srcF = (const f32x4*)(p.F16 + (size_t)g * (MH_FS / 16) * 2 * 64 * 8);
    for (int i = threadIdx.x; i < FWD16_SF_BYTES / 16; i += FWDPC_THREADS) ((f32x4*)sF)[i] = srcF[i];
    const f32x4* srcA = (const f32x4*)(p.A + (size_t)g * 32 * MH_NJ * 12);
    for (int i = threadIdx.x; i < FWD16_SA_BYTES / 16; i += FWDPC_THREADS) ((f32x4*)sA)[i] = srcA[i];
    if (threadIdx.x < 32) {
      const int b = g * 32 + threadIdx.x;
      f32x4 q = {1.f, 0.f, 0.f, 0.f};
      if (b < p.B) {
        q[0] = p.scale[b];
        if (p.transl) { q[1] = p.transl[(size_t)b * 3]; q[2] = p.transl[(size_t)b * 3 + 1]; q[3] = p.transl[(size_t)b * 3 + 2]; }
      }
      sS[threadIdx.x] = q;
      f32x4 th = {-3e38f, -3e38f, 3e38f, 3e38f};
      float ly = 3e38f;
      if (b < p.B) {
        const int4 pb = *(const int4*)(p.P.bbox_prev + (size_t)b * 4);
        th[0] = mh_unord(pb.x) + p.P.slack_ndc; th[1] = mh_unord(pb.y) + p.P.slack_ndc;
        th[2] = mh_unord(pb.z) - p.P.slack_ndc; th[3] = mh_unord(pb.w) - p.P.slack_ndc;
        ly = mh_unordu((unsigned)(p.P.lowkey_prev[b] >> 32)) - p.P.slack_y;
      }
      sTH[threadIdx.x] = th;
      sLY[threadIdx.x] = (f32x4){ly, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x >= 64 && threadIdx.x < 68) ((int*)(smem16 + FWD16_LDS_BYTES + 4 * FWDPC_SLOT_BYTES))[threadIdx.x - 64] = 0;
  }
  __syncthreads();
  // tiles of this pair: first + pr, first + pr + 4, ... below the workgroup's end
  const int first = blockIdx.x * p.tpb, tend = min(first + p.tpb, ntiles);
  int nt = 0;
  for (int t = first + pr; t < tend; t += 4) ++nt;
  const size_t row_bytes = (size_t)p.V * 12;
  if (wave < 4) {
    // ------------------------------------------------ producer ------------------------------------------------
    const unsigned char* const rbg = (const unsigned char*)p.P.rowb + (size_t)g * 32 * p.V * 4;
    const int last_row = FULL ? 31 : min(31, p.B - 1 - g * 32);
    for (int k = 0; k < nt; ++k) {
      const int tile = first + pr + 4 * k;
      const int v = tile * 32 + li;
      const int vc = v < p.V ? v : p.V - 1;
      const f16x8* Dw = (const f16x8*)p.D16 + (size_t)tile * (MH_KD / 16) * 6 * 64 + lane;
      f16x8 bq[FWDPC_STAGES][6];
#pragma unroll
      for (int st = 0; st < FWDPC_STAGES - 1; ++st)
#pragma unroll
        for (int i = 0; i < 6; ++i) bq[st][i] = Dw[(st * 6 + i) * 64];
      // what the consumer's epilogue needs from memory (behind the first operands in the queue: they are not needed before the slot is written)
      const f32x3 tv = *(const f32x3*)(p.vt + (size_t)vc * 3);
      int4 j4 = {0, 0, 0, 0};
      f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
      if (NW4 && p.nw == 4) {
        j4 = *(const int4*)(p.skidx + (size_t)vc * 4);
        w4 = *(const f32x4*)(p.skw + (size_t)vc * 4);
      } else {
        const int nwl = p.nw < 4 ? p.nw : 4;
        for (int q = 0; q < nwl; ++q) {
          const int jj = p.skidx[(size_t)vc * p.nw + q];
          const float ww = p.skw[(size_t)vc * p.nw + q];
          if (q == 0) { j4.x = jj; w4[0] = ww; }
          if (q == 1) { j4.y = jj; w4[1] = ww; }
          if (q == 2) { j4.z = jj; w4[2] = ww; }
          if (q == 3) { j4.w = jj; w4[3] = ww; }
        }
      }
      float rbv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const unsigned off4 = FULL ? (unsigned)(4 * lh * p.V + vc) * 4u + (unsigned)((r & 3) + 8 * (r >> 2)) * ((unsigned)p.V * 4u)
                                   : (unsigned)(min(row, last_row) * p.V + vc) * 4u;
        rbv[r] = *(const float*)(rbg + (size_t)off4);
      }
      f32x16 ax = {0}, ay = {0}, az = {0};
#pragma unroll
      for (int s16 = 0; s16 < MH_KD / 16; ++s16) {
        const int cur = s16 % FWDPC_STAGES, nxt = (s16 + FWDPC_STAGES - 1) % FWDPC_STAGES;
        if (s16 + FWDPC_STAGES - 1 < MH_KD / 16) {
#pragma unroll
          for (int i = 0; i < 6; ++i) bq[nxt][i] = Dw[((s16 + FWDPC_STAGES - 1) * 6 + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 ah = sF[(s16 * 2) * 64 + lane], al = sF[(s16 * 2 + 1) * 64 + lane];
        ax = MFMA_F16(ah, bq[cur][0], ax);
        ay = MFMA_F16(ah, bq[cur][2], ay);
        az = MFMA_F16(ah, bq[cur][4], az);
        ax = MFMA_F16(ah, bq[cur][1], ax);
        ay = MFMA_F16(ah, bq[cur][3], ay);
        az = MFMA_F16(ah, bq[cur][5], az);
        ax = MFMA_F16(al, bq[cur][0], ax);
        ay = MFMA_F16(al, bq[cur][2], ay);
        az = MFMA_F16(al, bq[cur][4], az);
        __builtin_amdgcn_sched_barrier(0);
      }
      // hand the tile over
      for (int spin = 0; FWDPC_FLAG() != 0 && spin < FWDPC_SPIN; ++spin) __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        slot[(q) * 64] = (f32x4){ax[4 * q], ax[4 * q + 1], ax[4 * q + 2], ax[4 * q + 3]};
        slot[(4 + q) * 64] = (f32x4){ay[4 * q], ay[4 * q + 1], ay[4 * q + 2], ay[4 * q + 3]};
        slot[(8 + q) * 64] = (f32x4){az[4 * q], az[4 * q + 1], az[4 * q + 2], az[4 * q + 3]};
        slot[(12 + q) * 64] = (f32x4){rbv[4 * q], rbv[4 * q + 1], rbv[4 * q + 2], rbv[4 * q + 3]};
      }
      slot[16 * 64] = (f32x4){tv[0], tv[1], tv[2], __int_as_float(v)};
      slot[17 * 64] = (f32x4){__int_as_float(j4.x), __int_as_float(j4.y), __int_as_float(j4.z), __int_as_float(j4.w)};
      slot[18 * 64] = w4;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FWDPC_SET(k + 1);
    }
    return;
  }
  // -------------------------------------------------- consumer --------------------------------------------------
  const float us = p.unscale;
  const unsigned char* const sB = (const unsigned char*)sS + (4 * lh) * 16;
  const unsigned row_b32 = (unsigned)p.V * 12u;
  unsigned char* const vg = (unsigned char*)p.verts + (size_t)g * 32 * row_bytes;
  unsigned char* const qg = p.vposed ? (unsigned char*)p.vposed + (size_t)g * 32 * row_bytes : nullptr;
  unsigned char* const ng = (unsigned char*)p.P.ndc + (size_t)g * 32 * row_bytes;
  for (int k = (wave - 4) >> 2; k < nt; k += FWDPC_NC) {
    for (int spin = 0; FWDPC_FLAG() != k + 1 && spin < FWDPC_SPIN; ++spin) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
    f32x4 qq[FWDPC_Q];
#pragma unroll
    for (int q = 0; q < FWDPC_Q; ++q) qq[q] = slot[q * 64];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FWDPC_SET(0);
    const int v = __float_as_int(qq[16][3]);
    if (v >= p.V) continue;                                 // (lanes beyond V in the model's last tile: no epilogue)
    const float tv0 = qq[16][0], tv1 = qq[16][1], tv2 = qq[16][2];
    const int sj[4] = {__float_as_int(qq[17][0]), __float_as_int(qq[17][1]), __float_as_int(qq[17][2]), __float_as_int(qq[17][3])};
    const float sw[4] = {qq[18][0], qq[18][1], qq[18][2], qq[18][3]};
    const unsigned char* aB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aB[q] = (const unsigned char*)sA + (4 * lh) * (MH_NJ * 48) + sj[q] * 48;
    const unsigned lane_off = (unsigned)(4 * lh * p.V + v) * 12u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row0 = (r & 3) + 8 * (r >> 2);                              // + 4*lh = accumulator row = body in group
      const float vp0 = fmaf(qq[r >> 2][r & 3], us, tv0), vp1 = fmaf(qq[4 + (r >> 2)][r & 3], us, tv1), vp2 = fmaf(qq[8 + (r >> 2)][r & 3], us, tv2);
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4* Aj = (const f32x4*)(aB[q] + row0 * (MH_NJ * 48));
        const f32x4 q0 = Aj[0], q1 = Aj[1], q2 = Aj[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          T[e] = fmaf(sw[q], q0[e], T[e]);
          T[4 + e] = fmaf(sw[q], q1[e], T[4 + e]);
          T[8 + e] = fmaf(sw[q], q2[e], T[8 + e]);
        }
      }
      if (!NW4) {
        const float* Ab = sA + (row0 + 4 * lh) * MH_NJ * 12;
        for (int q = 4; q < p.nw; ++q) {   // models with more than 4 bones per vertex (the one place a consumer loads)
          const float w = p.skw[(size_t)v * p.nw + q];
          const float* Aj = Ab + p.skidx[(size_t)v * p.nw + q] * 12;
#pragma unroll
          for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Aj[e], T[e]);
        }
      }
      const f32x4 st = *(const f32x4*)(sB + row0 * 16);                    // (scale, translation) of this body
      const float x0 = fmaf(T[2], vp2, fmaf(T[1], vp1, T[0] * vp0)) + T[3];
      const float x1 = fmaf(T[6], vp2, fmaf(T[5], vp1, T[4] * vp0)) + T[7];
      const float x2 = fmaf(T[10], vp2, fmaf(T[9], vp1, T[8] * vp0)) + T[11];
      const f32x3 o = {fmaf(st[0], x0, st[1]), fmaf(st[0], x1, st[2]), fmaf(st[0], x2, st[3])};
      if (FULL || g * 32 + row0 + 4 * lh < p.B) {
        const unsigned off_r = lane_off + (unsigned)row0 * row_b32;
        *(f32x3*)(vg + (size_t)off_r) = o;
        if (qg) {
          const f32x3 q = {vp0, vp1, vp2};
          *(f32x3*)(qg + (size_t)off_r) = q;
        }
        if (PROJ) {
          const float Zc = o[2];
          const float xn = p.P.s * (-o[0]) / Zc + p.P.w1, yn = p.P.s * (-o[1]) / Zc + p.P.h1;
          const f32x3 nd = {xn, yn, Zc};
          *(f32x3*)(ng + (size_t)off_r) = nd;
          const float drow = fabsf(fmaf(-yn, p.P.rk, p.P.ra) - qq[12 + (r >> 2)][r & 3]);
          const bool mv = !(drow < p.P.thr_soft);
          const f32x4 th = *(const f32x4*)(sB + 512 + row0 * 16);
          const float ly = *(const float*)(sB + 1024 + row0 * 16);
          const bool zok = Zc > 1e-8f;
          const bool c0 = zok && xn < th[0], c1 = zok && yn < th[1], c2 = zok && xn > th[2], c3 = zok && yn > th[3];
          const bool c4 = o[1] > ly;
          if (__builtin_amdgcn_ballot_w64(c0 || c1 || c2 || c3 || c4 || mv) != 0ull) {
            int b = g * 32 + row0 + 4 * lh;
            asm volatile("" : "+v"(b));
            int* bb = p.P.bbox + (size_t)b * 4;
            if (c0) atomicMin(bb, mh_ord(xn));
            if (c1) atomicMin(bb + 1, mh_ord(yn));
            if (c2) atomicMax(bb + 2, mh_ord(xn));
            if (c3) atomicMax(bb + 3, mh_ord(yn));
            if (c4) atomicMax(p.P.lowkey + b, ((unsigned long long)mh_ordu(o[1] + 0.0f) << 32) | (unsigned)~(unsigned)v);
            if (mv) p.P.moved[(drow < p.P.thr ? p.B : 0) + b] = 1;
          }
        }
      }
    }
  }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct FwdWs {
  float* featT;
  float* A;
  float* scale;
  uint16_t* F16;
};
static FwdWs carve_fwd(void* ws, int G) {
  char* p = (char*)ws;
  FwdWs w;
  w.featT = (float*)p;
  p += align256((size_t)G * MH_FS * 32 * 4);
  w.A = (float*)p;
  p += align256((size_t)G * 32 * MH_NJ * 12 * 4);
  w.scale = (float*)p;
  p += align256((size_t)G * 32 * 4);
  w.F16 = (uint16_t*)p;
  return w;
}

// What other translation units may read of a forward workspace (mh_keypoints.hip: the key-point term is computed from the pose
// features and bone transforms the LAST forward on that workspace left): ONE definition of the layout, and a record of which
// workspaces hold a forward of how many bodies (the layout was re-derived by hand over there).
// (A map that only grows -- ADVICE r05: the 64-slot table of round 5 evicted by address bits, and every 2-MiB-aligned allocator
// block hashed to slot 0: a forward on a second workspace between a forward and its key-point term made the check fail on a
// correct call.  An address the allocator hands out again is noted again by the forward that uses it.)
static std::mutex g_fwd_seen_mu;
static std::unordered_map<const void*, int>& fwd_seen() {
  static std::unordered_map<const void*, int> m;
  return m;
}
static void fwd_note(const void* ws, int B) {
  std::lock_guard<std::mutex> lk(g_fwd_seen_mu);
  fwd_seen()[ws] = B;
}
int mh_lbs_forward_views(int B, const void* ws, const float** featT, const float** A, const float** scale) {
  MH_CHECK(ws && B > 0 && featT && A && scale, "null argument");
  {
    std::lock_guard<std::mutex> lk(g_fwd_seen_mu);
    const auto it = fwd_seen().find(ws);
    MH_CHECK(it != fwd_seen().end() && it->second == B,
             "this workspace does not hold an LBS forward of this many bodies: the key-point term reads the pose features "
             "and bone transforms of the forward that immediately preceded it on the same workspace");
  }
  FwdWs w = carve_fwd(const_cast<void*>(ws), mh_groups(B));
  *featT = w.featT; *A = w.A; *scale = w.scale;
  return MH_OK;
}

// 1 (default): split-fp16 / split-bf16 contractions on the 16-bit matrix pipe; 0: exact-fp32 MFMA (A/B reference)
static int g_lbs_mode = -1;
static int lbs_mode() {
  if (g_lbs_mode < 0) {
    const char* e = getenv("MHHIP_LBS_FP32");
    g_lbs_mode = (e && e[0] == '1') ? 0 : 1;
  }
  return g_lbs_mode;
}
extern "C" int mh_lbs_set_mode(int split16) {
  g_lbs_mode = split16 ? 1 : 0;
  return MH_OK;
}
extern "C" int mh_lbs_get_mode(void) { return lbs_mode(); }
// the "LBS + projection" forward software-pipelined over a wave's tiles (k_skin_fwd16p) or tile after tile (k_skin_fwd16):
// the same bits either way.  MHHIP_FWD_PIPE=0|1 at first use; process-wide, part of the cycle graphs' key.
static int g_fwd_pipe = -1;
static int fwd_pipe() {
  if (g_fwd_pipe < 0) {
    const char* e = getenv("MHHIP_FWD_PIPE");
    g_fwd_pipe = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;
  }
  return g_fwd_pipe;
}
extern "C" int mh_lbs_set_forward_pipeline(int on) {
  g_fwd_pipe = on == 2 ? 2 : on ? 1 : 0;         // 2: producer and consumer waves (k_skin_fwd16pc)
  return MH_OK;
}
extern "C" int mh_lbs_get_forward_pipeline(void) { return fwd_pipe(); }

extern "C" size_t mh_lbs_workspace_bytes(int B) {
  int G = mh_groups(B < 1 ? 1 : B);
  return align256((size_t)G * MH_FS * 32 * 4) + align256((size_t)G * 32 * MH_NJ * 12 * 4) + align256((size_t)G * 32 * 4) +
         align256((size_t)G * (MH_FS / 16) * 2 * 64 * 16);
}

static int lbs_forward_impl(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                            const float* xscale, const float* transl, float* verts, float* vposed, float* posed_joints,
                            void* ws, void* stream, const mh_fwd_proj* proj = nullptr);

extern "C" int mh_lbs_forward_proj(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                                   const float* xscale, const float* transl, float* verts, float* vposed,
                                   const mh_fwd_proj* proj, void* ws, void* stream) {
  MH_CHECK(poses && proj, "null argument");
  MH_CHECK(proj->ndc && proj->rowb && proj->bbox && proj->bbox_prev && proj->lowkey && proj->lowkey_prev && proj->moved,
           "mh_fwd_proj: null target (fill it with mh_raster_forward_targets)");
  MH_CHECK(lbs_mode() != 0, "mh_lbs_forward_proj: the fused epilogue exists for the split-16-bit kernels only (mh_lbs_set_mode(1))");
  return lbs_forward_impl(m, B, NB, betas, poses, nullptr, xscale, transl, verts, vposed, nullptr, ws, stream, proj);
}

extern "C" int mh_lbs_forward(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                              const float* xscale, const float* transl, float* verts, float* vposed,
                              float* posed_joints, void* ws, void* stream) {
  MH_CHECK(poses, "null argument");
  return lbs_forward_impl(m, B, NB, betas, poses, nullptr, xscale, transl, verts, vposed, posed_joints, ws, stream);
}

extern "C" int mh_lbs_forward_rotmats(const mh_model* m, int B, int NB, const float* betas, const float* rotmats,
                                      const float* xscale, const float* transl, float* verts, float* posed_joints,
                                      void* ws, void* stream) {
  MH_CHECK(rotmats, "null argument");
  return lbs_forward_impl(m, B, NB, betas, nullptr, rotmats, xscale, transl, verts, nullptr, posed_joints, ws, stream);
}

extern "C" int mh_lbs_forward_ex(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                                 const float* xscale, const float* transl, float* verts, float* vposed, float* posed_joints,
                                 void* ws, void* stream) {
  MH_CHECK((poses != nullptr) != (rotmats != nullptr), "exactly one of poses (axis-angle) and rotmats");
  return lbs_forward_impl(m, B, NB, betas, poses, rotmats, xscale, transl, verts, vposed, posed_joints, ws, stream);
}

static int lbs_forward_impl(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                            const float* xscale, const float* transl, float* verts, float* vposed, float* posed_joints,
                            void* ws, void* stream, const mh_fwd_proj* proj) {
  MH_CHECK(m && betas && verts && ws, "null argument");
  MH_CHECK(B > 0 && NB > 0, "B and NB must be positive");
  hipStream_t st = (hipStream_t)stream;
  const int G = mh_groups(B);
  FwdWs w = carve_fwd(ws, G);
  fwd_note(ws, B);
  PoseFwdP pp;
  pp.B = B; pp.NB = NB; pp.G = G;
  pp.betas = betas; pp.poses = poses; pp.rotmats = rotmats; pp.xscale = xscale;
  pp.Jt = m->Jt; pp.JS = m->JS;
  pp.featT = w.featT; pp.A = w.A; pp.scale = w.scale; pp.posed = posed_joints;
  pp.tree = m->tree;
  const bool split16 = lbs_mode() != 0;
  pp.F16 = split16 ? w.F16 : nullptr;
  pp.bbox = proj ? proj->bbox : nullptr; pp.bbox_prev = proj ? proj->bbox_prev : nullptr;
  pp.lowkey = proj ? proj->lowkey : nullptr; pp.lowkey_prev = proj ? proj->lowkey_prev : nullptr;
  pp.moved = proj ? proj->moved : nullptr;
  pp.clear = proj ? proj->clear : nullptr; pp.clear_n = proj ? proj->clear_n : 0ull;
  mh_prof_mark(MH_PROF_POSE_FWD, 0, st);
  hipLaunchKernelGGL(k_pose_fwd, dim3(G * 4), dim3(256), 0, st, pp);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_POSE_FWD, 1, st);
  if (split16) {
    SkinFwd16P sp;
    sp.B = B; sp.G = G; sp.V = m->V; sp.VP = m->VP; sp.nw = m->nw;
    sp.unscale = ldexpf(1.f, -(MH_F16_FEAT_SHIFT + m->d16_shift));
    sp.F16 = w.F16; sp.A = w.A; sp.scale = w.scale; sp.transl = transl;
    sp.D16 = m->D16; sp.vt = m->vt; sp.skidx = m->skidx; sp.skw = m->skw;
    sp.verts = verts; sp.vposed = vposed;
    MH_CHECK((size_t)32 * m->V * 12 < ((size_t)1 << 32), "model too large for 32-bit lane offsets");
#ifdef FWD16_LDS_PAD      // experiment builds (tools/c4_probe.py): more LDS than the form without the epilogue uses -- with 30000 no
    const size_t lds = FWD16_LDS_BYTES + (proj ? 0 : FWD16_LDS_PAD);      // selection workgroup fits on a CU beside it
#else
    const size_t lds = FWD16_LDS_BYTES;
#endif
    const bool full = (B % 32) == 0, nw4 = m->nw <= 4;
    auto kern = proj ? (full ? (nw4 ? k_skin_fwd16<true, true, true> : k_skin_fwd16<true, false, true>)
                             : (nw4 ? k_skin_fwd16<false, true, true> : k_skin_fwd16<false, false, true>))
                     : (full ? (nw4 ? k_skin_fwd16<true, true, false> : k_skin_fwd16<true, false, false>)
                             : (nw4 ? k_skin_fwd16<false, true, false> : k_skin_fwd16<false, false, false>));
    static unsigned char attr16[8][MH_MAX_DEVICES];
    const int ki = (proj ? 4 : 0) + (full ? 2 : 0) + (nw4 ? 1 : 0);
    sp.P = proj ? *proj : mh_fwd_proj{};
    if (mh_first_on_device(attr16[ki]))
      MH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mh_prof_mark(MH_PROF_SKIN_FWD, 0, st);
    // Two workgroups fit on a CU (66 KB of LDS each): aim at 2 x 256 of them, all resident at once and every CU with the same
    // number of vertex tiles.  (Rounds 1-2 gave every workgroup 16 tiles: 350 workgroups at C3, so 94 CUs carried two of
    // them and 162 one -- the kernel took as long as the CUs with 32 tiles; with 11 tiles per workgroup every CU has ~22.)
    const int ntiles = m->VP / 32;
    if (proj && fwd_pipe() == 2) {
      // producer and consumer waves (k_skin_fwd16pc): one workgroup per CU, its tiles dealt to four producer / consumer pairs
      auto kc = full ? (nw4 ? k_skin_fwd16pc<true, true> : k_skin_fwd16pc<true, false>)
                     : (nw4 ? k_skin_fwd16pc<false, true> : k_skin_fwd16pc<false, false>);
      static unsigned char attrc[4][MH_MAX_DEVICES];
      if (mh_first_on_device(attrc[(full ? 2 : 0) + (nw4 ? 1 : 0)]))
        MH_HIP(hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWDPC_LDS_BYTES));
      int bpg = std::max(1, 256 / G);
      int tpb = std::max(1, (ntiles + bpg - 1) / bpg);
      if (const char* e = getenv("MHHIP_FWD_TPB")) tpb = std::max(1, atoi(e));
      sp.tpb = tpb;
      bpg = (ntiles + tpb - 1) / tpb;
      hipLaunchKernelGGL(kc, dim3(bpg, G), dim3(FWDPC_THREADS), (size_t)FWDPC_LDS_BYTES, st, sp);
      MH_LAUNCH_CHECK();
      mh_prof_mark(MH_PROF_SKIN_FWD, 1, st);
      return MH_OK;
    }
    if (proj && fwd_pipe()) {
      // the software-pipelined form (k_skin_fwd16p): one workgroup per CU at 256 registers, three tiles per wave
      auto kp = full ? (nw4 ? k_skin_fwd16p<true, true, true> : k_skin_fwd16p<true, false, true>)
                     : (nw4 ? k_skin_fwd16p<false, true, true> : k_skin_fwd16p<false, false, true>);
      static unsigned char attrp[4][MH_MAX_DEVICES];
      if (mh_first_on_device(attrp[(full ? 2 : 0) + (nw4 ? 1 : 0)]))
        MH_HIP(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int bpg = std::max(1, 256 / G);
      int tpb = std::min(FWD16_WAVES * 3, (ntiles + bpg - 1) / bpg);
      if (const char* e = getenv("MHHIP_FWD_TPB")) tpb = std::max(1, std::min(FWD16_WAVES * 4, atoi(e)));
      sp.tpb = tpb;
      bpg = (ntiles + tpb - 1) / tpb;
      hipLaunchKernelGGL(kp, dim3(bpg, G), dim3(FWD16_WAVES * 64), lds, st, sp);
      MH_LAUNCH_CHECK();
      mh_prof_mark(MH_PROF_SKIN_FWD, 1, st);
      return MH_OK;
    }
    int bpg = std::max(1, (2 * 256) / G);
    int tpb = std::min(FWD16_WAVES * FWD16_TPW, (ntiles + bpg - 1) / bpg);
    if (const char* e = getenv("MHHIP_FWD_TPB")) tpb = std::max(1, std::min(FWD16_WAVES * FWD16_TPW, atoi(e)));
    sp.tpb = tpb;
    bpg = (ntiles + tpb - 1) / tpb;
    hipLaunchKernelGGL(kern, dim3(bpg, G), dim3(FWD16_WAVES * 64), lds, st, sp);
    MH_LAUNCH_CHECK();
    mh_prof_mark(MH_PROF_SKIN_FWD, 1, st);
    return MH_OK;
  }
  SkinFwdP sp;
  sp.B = B; sp.G = G; sp.V = m->V; sp.VP = m->VP; sp.nw = m->nw;
  sp.featT = w.featT; sp.A = w.A; sp.scale = w.scale; sp.transl = transl;
  sp.D = m->D; sp.vt = m->vt; sp.skidx = m->skidx; sp.skw = m->skw;
  sp.verts = verts; sp.vposed = vposed;
  mh_prof_mark(MH_PROF_SKIN_FWD, 0, st);
  hipLaunchKernelGGL(k_skin_fwd, dim3(((m->VP / 32 + 3) / 4 + 7) / 8 * 8, G), dim3(256), 0, st, sp);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_SKIN_FWD, 1, st);
  return MH_OK;
}

// =============================================================================================
// sparse joint regression
// =============================================================================================
__global__ __launch_bounds__(64) void k_joints_regress(int B, int V, int J, const int* ptr, const int* vidx,
                                                       const float* w, const float* rowsum, const float* verts,
                                                       const float* corr, int root, float* joints) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* vb = verts + (size_t)b * V * 3;
  float rx = 0, ry = 0, rz = 0;
  for (int pass = (root >= 0 ? 0 : 1); pass < 2; ++pass) {
    for (int jj = 0; jj < J; ++jj) {
      const int j = (pass == 0) ? root : jj;
      float a0 = 0, a1 = 0, a2 = 0;
      for (int e = ptr[j] + lane; e < ptr[j + 1]; e += 64) {
        const float ww = w[e];
        const float* q = vb + (size_t)vidx[e] * 3;
        a0 = fmaf(ww, q[0], a0);
        a1 = fmaf(ww, q[1], a1);
        a2 = fmaf(ww, q[2], a2);
      }
      a0 = mh_wave_sum(a0);
      a1 = mh_wave_sum(a1);
      a2 = mh_wave_sum(a2);
      if (corr) {
        // the vertices carry the translation `corr` with weight rowsum (regressor rows need not sum to one): plain joints
        // get the missing (1 - rowsum) t; root-relative joints are cleared of it (and get one t back below), so that
        // the result is s (J_j - J_root) + t -- what the reference forms as scale * joints + poses_T (optimizer.py:701-703)
        const float c = root >= 0 ? -rowsum[j] : 1.f - rowsum[j];
        a0 = fmaf(c, corr[(size_t)b * 3], a0);
        a1 = fmaf(c, corr[(size_t)b * 3 + 1], a1);
        a2 = fmaf(c, corr[(size_t)b * 3 + 2], a2);
      }
      if (pass == 0) {
        rx = a0; ry = a1; rz = a2;
        if (corr) { rx -= corr[(size_t)b * 3]; ry -= corr[(size_t)b * 3 + 1]; rz -= corr[(size_t)b * 3 + 2]; }
        break;
      }
      if (lane == 0) {
        float* o = joints + ((size_t)b * J + j) * 3;
        o[0] = a0 - rx;
        o[1] = a1 - ry;
        o[2] = a2 - rz;
      }
    }
  }
}

extern "C" int mh_joints_regress(const mh_model* m, int which, int B, const float* verts, const float* corr,
                                 int root_relative_to, float* joints, void* stream) {
  MH_CHECK(m && verts && joints, "null argument");
  MH_CHECK(which >= 0 && which < 4, "unknown joint set");
  const mh_regressor& r = m->reg[which];
  MH_CHECK(r.J > 0, "this joint regressor was not given to mh_model_create");
  MH_CHECK(root_relative_to < r.J, "root joint out of range");
  MH_CHECK(B > 0, "B must be positive");
  hipLaunchKernelGGL(k_joints_regress, dim3(B), dim3(64), 0, (hipStream_t)stream, B, m->V, r.J, r.ptr, r.vidx, r.w,
                     r.rowsum, verts, corr, root_relative_to, joints);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// adjoint of mh_joints_regress for ANY of the four regressors: gverts[b, v] += sum_j reg[j, v] g[b, j] (and the share of
// the translation correction).  One wave per body walks the joints one after the other -- inside a joint's row the
// vertices are distinct, so plain read-modify-writes in a fixed order, no atomics: deterministic.  (The fused 2D term
// of the optimiser has this built into k_skinbwd16 for the AlphaPose regressor; this is the path of the other
// smpl_sparse_joints_key values and of the overlay's autograd for joints_h36m17 / joints_mupots / j3d.)
__global__ __launch_bounds__(64) void k_joints_regress_bwd(int B, int V, int J, const int* ptr, const int* vidx, const float* w,
                                                           const float* rowsum, const float* gjoints, int root, float* gverts,
                                                           float* gcorr) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* gv = gverts + (size_t)b * V * 3;
  const float* gj = gjoints + (size_t)b * J * 3;
  // joints relative to a root joint: J_j - J_root, so the root's row receives minus the sum of all adjoints
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  if (root >= 0)
    for (int j = 0; j < J; ++j) { r0 += gj[j * 3]; r1 += gj[j * 3 + 1]; r2 += gj[j * 3 + 2]; }
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int j = 0; j < J; ++j) {
    float g0 = gj[j * 3], g1 = gj[j * 3 + 1], g2 = gj[j * 3 + 2];
    if (j == root) { g0 -= r0; g1 -= r1; g2 -= r2; }
    for (int e = ptr[j] + lane; e < ptr[j + 1]; e += 64) {
      const float ww = w[e];
      float* q = gv + (size_t)vidx[e] * 3;
      q[0] = fmaf(ww, g0, q[0]);
      q[1] = fmaf(ww, g1, q[1]);
      q[2] = fmaf(ww, g2, q[2]);
    }
    // EXPLICIT d joint / d corr (the share the vertices carry, rowsum . t, reaches the translation through gverts):
    // plain joints get (1 - rowsum_j) t added; the root-relative form subtracts rowsum_j t and the root's
    // (rowsum_root - 1) t, i.e. 1 - rowsum_j + rowsum_root (this was 1, 3e-4 off on the translation gradient
    // for regressors whose rows do not sum to exactly one)
    const float c = root >= 0 ? 1.f - rowsum[j] + rowsum[root] : 1.f - rowsum[j];
    c0 = fmaf(c, gj[j * 3], c0); c1 = fmaf(c, gj[j * 3 + 1], c1); c2 = fmaf(c, gj[j * 3 + 2], c2);
    __threadfence_block();                         // the next joint's row may touch the same vertices
    __builtin_amdgcn_wave_barrier();
  }
  if (gcorr && lane == 0) {
    gcorr[(size_t)b * 3] += c0; gcorr[(size_t)b * 3 + 1] += c1; gcorr[(size_t)b * 3 + 2] += c2;
  }
}

extern "C" int mh_joints_regress_backward(const mh_model* m, int which, int B, const float* gjoints, int root_relative_to,
                                          float* gverts, float* gcorr, void* stream) {
  MH_CHECK(m && gjoints && gverts, "null argument");
  MH_CHECK(which >= 0 && which < 4, "unknown joint set");
  const mh_regressor& r = m->reg[which];
  MH_CHECK(r.J > 0, "this joint regressor was not given to mh_model_create");
  MH_CHECK(root_relative_to < r.J, "root joint out of range");
  MH_CHECK(B > 0, "B must be positive");
  hipLaunchKernelGGL(k_joints_regress_bwd, dim3(B), dim3(64), 0, (hipStream_t)stream, B, m->V, r.J, r.ptr, r.vidx, r.w, r.rowsum,
                     gjoints, root_relative_to, gverts, gcorr);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// =============================================================================================
// backward
// =============================================================================================
#define BWD_AS 292   // LDS row stride (floats) of one body's 24x12 transforms: conflict-free b128 reads
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define BWD_NACC (14 + 24)   // f32x4 accumulators per lane: 14 feature tiles + 12 x 2 joint tiles

struct SkinBwdP {
  int B, G16, GB, V, VP, nw, CH, PQ;   // G16 = groups of 16 bodies, GB = padded body count, PQ = vertex quads per chunk
  const float* A;
  const float* scale;
  const float* vposed;
  const float* gverts;
  const float* gjoints;   // [B][17][3] or null
  const float* Dt;
  const int* skidx;
  const float* skw;
  const int* kpv_ptr;
  const int* kpv_j;
  const float* kpv_w;
  const int* kpv_head;
  float* pF;   // [CH][GB][224]
  float* pA;   // [CH][GB][12][24]
  float* pS;   // [CH][GB][4]  (gT.xyz, gscale)
};

// lane = (body li = lane&15, vertex lq = lane>>4): the element-wise adjoints of one (body, vertex)
// pair are exactly the A operand of v_mfma_f32_16x16x4_f32 (A[i=body][k=vertex]); the B operands
// are rows of the constant tables (B[k=vertex][j=column]).
__global__ __launch_bounds__(256, 2) void k_skin_bwd(SkinBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                       // [16][BWD_AS]
  float* sGj = sA + 16 * BWD_AS;          // [16][52]
  float* sRed = sGj + 16 * 52;            // [BWD_NACC*4][64] accumulators + [16][4] scalars
  const int g = blockIdx.x, ch = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 16 * MH_NJ * 12; i += 256) {
    const int bb = i / (MH_NJ * 12), e = i % (MH_NJ * 12);
    sA[bb * BWD_AS + e] = p.A[(size_t)(g * 16 + bb) * MH_NJ * 12 + e];
  }
  for (int i = tid; i < 16 * 51; i += 256) {
    const int bb = i / 51, e = i % 51;
    const int b = g * 16 + bb;
    sGj[bb * 52 + e] = (p.gjoints && b < p.B) ? p.gjoints[(size_t)b * 51 + e] : 0.f;
  }
  __syncthreads();
  const int b = g * 16 + li;
  const bool bvalid = b < p.B;
  const float sb = bvalid ? p.scale[b] : 0.f;
  f32x4 accF[14], accA[24];
#pragma unroll
  for (int i = 0; i < 14; ++i) accF[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 24; ++i) accA[i] = (f32x4){0, 0, 0, 0};
  float sT0 = 0, sT1 = 0, sT2 = 0, sS = 0;
  const int qend = min((ch + 1) * p.PQ, p.VP / 4);
  const float* sAb = sA + li * BWD_AS;
  // Dt tile of (vertex, body column li): [c][v][li][16] -> four 16-byte loads per component
  const f32x4* dtb = (const f32x4*)p.Dt + ((size_t)lq * 16 + li) * 4;
  const size_t cplane = (size_t)p.VP * 64;               // f32x4 units per component
  int qd = ch * p.PQ + wave;
  f32x4 dbuf[3][2];      // three half-components (8 column tiles each) in flight: 24 registers instead of 48
  float ng[3] = {0, 0, 0}, nq[3] = {0, 0, 0};
  int nsj[4] = {0, 0, 0, 0}, nke0 = 0, nke1 = 0;
  float nsw[4] = {0, 0, 0, 0};
  const bool has_g = p.gverts != nullptr;
  const int nw4 = p.nw < 4 ? p.nw : 4;
  auto load_vertex_tables = [&](int vv) {   // skinning row + regressor row bounds of vertex vv (padded rows exist up to VP)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      nsj[k] = k < nw4 ? p.skidx[(size_t)vv * p.nw + k] : 0;
      nsw[k] = k < nw4 ? p.skw[(size_t)vv * p.nw + k] : 0.f;
    }
    nke0 = p.kpv_ptr[vv];
    nke1 = p.kpv_ptr[vv + 1];
  };
  if (qd < qend) {
    load_vertex_tables(4 * qd + lq);
    const f32x4* d = dtb + (size_t)(4 * qd) * 64;
    dbuf[0][0] = d[0]; dbuf[0][1] = d[1];                       // component 0, tiles 0..7
    dbuf[1][0] = d[2]; dbuf[1][1] = d[3];                       // component 0, tiles 8..13
    dbuf[2][0] = d[cplane]; dbuf[2][1] = d[cplane + 1];         // component 1, tiles 0..7
    const int v = 4 * qd + lq;
    if (bvalid && v < p.V) {
      const size_t o = ((size_t)b * p.V + v) * 3;
      if (has_g) { ng[0] = p.gverts[o]; ng[1] = p.gverts[o + 1]; ng[2] = p.gverts[o + 2]; }
      nq[0] = p.vposed[o]; nq[1] = p.vposed[o + 1]; nq[2] = p.vposed[o + 2];
    }
  }
  for (; qd < qend; qd += 4) {
    const int v = 4 * qd + lq;
    float g0 = ng[0], g1 = ng[1], g2 = ng[2];
    const float q0 = nq[0], q1 = nq[1], q2 = nq[2];
    // the next quad's adjoints and posed vertices stream from HBM: issue them a whole iteration ahead
    int sj4[4];
    float sw4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sj4[k] = nsj[k]; sw4[k] = nsw[k]; }
    const int ke0 = nke0, ke1 = nke1;
    const int qn = qd + 4 < qend ? qd + 4 : qd;
    load_vertex_tables(4 * qn + lq);
    {
      const int vn = 4 * qn + lq;
      ng[0] = ng[1] = ng[2] = nq[0] = nq[1] = nq[2] = 0.f;
      if (bvalid && vn < p.V) {
        const size_t o = ((size_t)b * p.V + vn) * 3;
        if (has_g) { ng[0] = p.gverts[o]; ng[1] = p.gverts[o + 1]; ng[2] = p.gverts[o + 2]; }
        nq[0] = p.vposed[o]; nq[1] = p.vposed[o + 1]; nq[2] = p.vposed[o + 2];
      }
    }
    // key-point regressor adjoint: dL/dverts += R^T dL/djoints
    for (int e = ke0; e < ke1; ++e) {
      const float w = p.kpv_w[e];
      const float* gj = sGj + li * 52 + p.kpv_j[e] * 3;
      g0 = fmaf(w, gj[0], g0);
      g1 = fmaf(w, gj[1], g1);
      g2 = fmaf(w, gj[2], g2);
    }
    // blended transform of this (body, vertex)
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    float wd0 = 0.f, wd1 = 0.f;   // dense skinning weights W[v][li], W[v][16+li] (MFMA B operands)
    for (int k = 0; k < p.nw; ++k) {       // (kept as a loop: unrolling it costs more in spills than it saves)
      // selects instead of sj4[k]: a dynamically indexed register array would live in scratch memory
      int j = k == 0 ? sj4[0] : k == 1 ? sj4[1] : k == 2 ? sj4[2] : sj4[3];
      float w = k == 0 ? sw4[0] : k == 1 ? sw4[1] : k == 2 ? sw4[2] : sw4[3];
      if (k >= 4) { j = p.skidx[(size_t)v * p.nw + k]; w = p.skw[(size_t)v * p.nw + k]; }
      const f32x4* Aj = (const f32x4*)(sAb + j * 12);
      const f32x4 a0 = Aj[0], a1 = Aj[1], a2 = Aj[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T[e] = fmaf(w, a0[e], T[e]);
        T[4 + e] = fmaf(w, a1[e], T[4 + e]);
        T[8 + e] = fmaf(w, a2[e], T[8 + e]);
      }
      wd0 += (j == li) ? w : 0.f;
      wd1 += (j == li + 16) ? w : 0.f;
    }
    const float x0 = fmaf(T[2], q2, fmaf(T[1], q1, T[0] * q0)) + T[3];
    const float x1 = fmaf(T[6], q2, fmaf(T[5], q1, T[4] * q0)) + T[7];
    const float x2 = fmaf(T[10], q2, fmaf(T[9], q1, T[8] * q0)) + T[11];
    sT0 += g0; sT1 += g1; sT2 += g2;
    sS += g0 * x0 + g1 * x1 + g2 * x2;
    const float gx0 = sb * g0, gx1 = sb * g1, gx2 = sb * g2;
    // d/d v_posed = T.R^T gx  -> rows of the [shape | pose] basis
    const float gv[3] = {fmaf(T[8], gx2, fmaf(T[4], gx1, T[0] * gx0)), fmaf(T[9], gx2, fmaf(T[5], gx1, T[1] * gx0)),
                         fmaf(T[10], gx2, fmaf(T[6], gx1, T[2] * gx0))};
    // six half-components per quad; the buffer a half has just used is refilled with the half three steps ahead
    // (same quad, or the next quad's first halves)
    const f32x4* dc = dtb + (size_t)(4 * qd) * 64;
    const f32x4* dn = dtb + (size_t)(4 * qn) * 64;
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      const int c = st >> 1, h = st & 1, bi = st % 3;
#pragma unroll
      for (int t = 8 * h; t < (h ? 14 : 8); ++t) accF[t] = MFMA16(gv[c], dbuf[bi][(t >> 2) & 1][t & 3], accF[t]);
      const int s3 = st + 3;
      const f32x4* src = (s3 < 6 ? dc : dn) + (size_t)((s3 % 6) >> 1) * cplane + 2 * (s3 & 1);
      dbuf[bi][0] = src[0];
      dbuf[bi][1] = src[1];
    }
    // d/dA_j = sum_v w_vj gx (x) [v_posed;1]
    const float gxs[3] = {gx0, gx1, gx2};
    const float qh[4] = {q0, q1, q2, 1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float a = gxs[r] * qh[c];
        accA[(r * 4 + c) * 2] = MFMA16(a, wd0, accA[(r * 4 + c) * 2]);
        accA[(r * 4 + c) * 2 + 1] = MFMA16(a, wd1, accA[(r * 4 + c) * 2 + 1]);
      }
    }
  }
  // ---- reduce the 4 waves through LDS (fixed order -> deterministic) ----
  sT0 += __shfl_xor(sT0, 16, 64); sT0 += __shfl_xor(sT0, 32, 64);
  sT1 += __shfl_xor(sT1, 16, 64); sT1 += __shfl_xor(sT1, 32, 64);
  sT2 += __shfl_xor(sT2, 16, 64); sT2 += __shfl_xor(sT2, 32, 64);
  sS += __shfl_xor(sS, 16, 64); sS += __shfl_xor(sS, 32, 64);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 14; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* q = sRed + (t * 4 + r) * 64 + lane;
          *q = (w == 0) ? accF[t][r] : (*q + accF[t][r]);
        }
#pragma unroll
      for (int t = 0; t < 24; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* q = sRed + ((14 + t) * 4 + r) * 64 + lane;
          *q = (w == 0) ? accA[t][r] : (*q + accA[t][r]);
        }
      if (lq == 0) {
        float* q = sRed + BWD_NACC * 4 * 64 + li * 4;
        if (w == 0) { q[0] = sT0; q[1] = sT1; q[2] = sT2; q[3] = sS; }
        else { q[0] += sT0; q[1] += sT1; q[2] += sT2; q[3] += sS; }
      }
    }
    __syncthreads();
  }
  // ---- write this chunk's partials; accumulator (t, r) of lane l is C[row = (l>>4)*4 + r][col = l&15] ----
  const size_t base = (size_t)ch * p.GB + (size_t)g * 16;
  float* oF = p.pF + base * MH_FS;
  for (int i = tid; i < 14 * 4 * 64; i += 256) {
    const int t = i >> 8, r = (i >> 6) & 3, l = i & 63;
    const int row = (l >> 4) * 4 + r;
    oF[(size_t)row * MH_FS + t * 16 + (l & 15)] = sRed[i];
  }
  float* oA = p.pA + base * 288;
  for (int i = tid; i < 24 * 4 * 64; i += 256) {
    const int t = i >> 8, r = (i >> 6) & 3, l = i & 63;
    const int row = (l >> 4) * 4 + r;
    const int joint = (t & 1) * 16 + (l & 15);
    if (joint < MH_NJ) oA[(size_t)row * 288 + (t >> 1) * MH_NJ + joint] = sRed[14 * 256 + i];
  }
  float* oS = p.pS + base * 4;
  for (int i = tid; i < 64; i += 256) oS[i] = sRed[BWD_NACC * 4 * 64 + i];
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 backward (default).  The two dense adjoints of the skinning pass are contractions over the vertices
// with a per-(body, vertex) element-wise operand:
//     gfeat[b][k]      = sum_{v,c} gv[b][v][c] . D[k][v][c]              gv = T^T (s g)
//     gA[b][j][r][c]   = sum_v     W[v][j] . (s g_r)[b][v] . [q;1]_c[b][v]
// The element-wise values of 32 bodies x 16 vertices, as two bf16 terms, ARE the A operand of
// v_mfma_f32_32x32x16_bf16 (lane = (body, vertex half), eight vertices per lane); the B operands are the constant
// tables Dt16 / W16 (two bf16 terms each); three products per operand pair, fp32 accumulation.
// ---------------------------------------------------------------------------------------------------------------
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// eight fp32 values -> (hi, lo) bf16 operand fragments: hi = bf16(x), lo = bf16(x - hi), round to nearest even
__device__ __forceinline__ void mh_split_bf16x8(const float x[8], bfx8& hi, bfx8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bfx2));
    const f32x2 r = {v[0] - __builtin_bit_cast(float, u << 16), v[1] - __builtin_bit_cast(float, u & 0xffff0000u)};
    h[i] = u;
    l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bfx2));
  }
  hi = __builtin_bit_cast(bfx8, h);
  lo = __builtin_bit_cast(bfx8, l);
}

struct Bwd16P {
  int B, G, V, VP, nw, CH, PB;   // G groups of 32 bodies, PB = 16-vertex blocks per chunk
  const float* A;
  const float* scale;
  const float* vposed;
  const float* gverts;
  const float* gjoints;   // [B][17][3] or null
  const uint16_t* Dt16;
  const uint16_t* W16;
  const int* skidx;
  const float* skw;
  const int* kpv_ptr;
  const int* kpv_j;
  const float* kpv_w;
  const int* kpv_head;
  float* pF;   // [CH][GB][224]
  float* pA;   // [CH][GB][12][24]
  float* pS;   // [CH][GB][4]  (gT.xyz, gscale)
};

#define BW_TS 20                                   // padded body-row stride (floats): conflict-free ds_read_b128
#ifndef SB_OCC
#define SB_OCC 2
#endif
#ifndef SB_FENCE2
#define SB_FENCE2
#endif
#ifndef SB_FENCE
#define SB_FENCE
#endif
#define SB_GS 33                                   // vertex-row stride (floats) of the vertex-major adjoint tile
#define SB_LDS_FLOATS (32 * BWD_AS + 32 * 52 + 3 * 16 * SB_GS + 7 * 32 * BW_TS + 3 * 2 * 64 * 4 + 8 * 32 * 3)
// One kernel for both adjoints.  A workgroup = one body group x one vertex chunk, walked in blocks of 16 vertices; its
// four waves share each block through LDS instead of each owning blocks:
//   stage   256 threads fetch the block's adjoints and posed vertices row-coalesced (16 lanes = the 192 contiguous
//           bytes of one body) -- the next block's rows are requested before this block's arithmetic;
//   blend   thread = (body tid&31, vertex pair tid>>5): bone-blended transform, key-point adjoint, gv = T^T (s g); the
//           pair's (hi, lo) bf16 terms go to LDS already in A-fragment order (one dword per term and component), s g
//           goes back body-major.  A half wave shares its vertex, so the skinning rows are broadcast loads;
//   mfma    every wave reads the SAME A fragments and owns output columns: wave w the basis columns 64 w .. 64 w + 63
//           (two 32-wide tiles, the fourth wave one) and the joint-gradient entries 3 w .. 3 w + 2 of the twelve; it
//           alone reads the B fragments of those columns -- one pass over Dt16 per workgroup, not one per wave.
// 80 accumulator registers per lane instead of 112 / 192: two workgroups per CU (75 KB of LDS each), eight waves to
// cover the LDS and barrier waits that the one-wave-per-SIMD two-kernel version (r02a) exposed.  No cross-wave
// reduction: a wave's accumulators cover the chunk's every vertex.  Chunks are summed in fixed order by k_pose_bwd.
template <bool NW4, bool KP>
__global__ __launch_bounds__(256, SB_OCC) void k_skinbwd16(Bwd16P p) {
  extern __shared__ __attribute__((aligned(16))) float smb[];
  float* sA = smb;                            // [32][BWD_AS]
  float* sGj = sA + 32 * BWD_AS;              // [32][52]
  float* sGv = sGj + 32 * 52;                 // [3][16][SB_GS]   adjoints, vertex-major (blend: lanes = bodies)
  float* sGx = sGv + 3 * 16 * SB_GS;          // [3][32][BW_TS]   s (g + R^T gj), body-major (mfma phase)
  float* sQ = sGx + 3 * 32 * BW_TS;           // [4][32][BW_TS]   posed vertices, body-major; [3] = ones
  unsigned* sF = (unsigned*)(sQ + 4 * 32 * BW_TS);   // [3][2][64][4]  bf16 (hi, lo) A fragments of gv
  float* sTr = (float*)(sF + 3 * 2 * 64 * 4);        // [8][32][3]
  // Workgroups go to the eight XCDs round-robin; XCD x takes the x-th eighth of the (chunk-major) work list, so a
  // chunk's slice of Dt16 / W16 is fetched into ~one L2 instead of all eight (measured: 150 MB less HBM-side traffic).
  const int per = gridDim.x >> 3, item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (item >= p.G * p.CH) return;
  const int g = item % p.G, ch = item / p.G;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  L_T0();
  for (int i = tid; i < 32 * MH_NJ * 12; i += 256) {
    const int bb = i / (MH_NJ * 12), e = i % (MH_NJ * 12);
    sA[bb * BWD_AS + e] = p.A[(size_t)(g * 32 + bb) * MH_NJ * 12 + e];
  }
  for (int i = tid; i < 32 * 51; i += 256) {
    const int bb = i / 51, e = i % 51;
    const int b = g * 32 + bb;
    sGj[bb * 52 + e] = (p.gjoints && b < p.B) ? p.gjoints[(size_t)b * 51 + e] : 0.f;
  }
  const int bi = tid & 31, vs = tid >> 5;      // blend phase: body, vertex pair
  const float sb = (g * 32 + bi < p.B) ? p.scale[g * 32 + bi] : 0.f;
  const float* sAb = sA + bi * BWD_AS;
  f32x16 accF[2], accA[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) accF[i] = (f32x16){0};
#pragma unroll
  for (int i = 0; i < 3; ++i) accA[i] = (f32x16){0};
  float sT0 = 0, sT1 = 0, sT2 = 0;
  const int nblk = p.VP / 16;
  const int b0 = ch * p.PB, bend = min((ch + 1) * p.PB, nblk);
  const bool two = wave < 3;                   // the fourth wave owns one basis tile (and repeats it into a second
  const int ct0 = 2 * wave, ct1 = two ? 2 * wave + 1 : 6;   // accumulator it never writes: no branches in the loop)
  for (int i = tid; i < 32 * BW_TS; i += 256) sQ[3 * 32 * BW_TS + i] = 1.f;
  const bfx8* Db = (const bfx8*)p.Dt16 + lane;   // [blk][c][ct][term][lane] in 16-byte units
  const bfx8* Wb = (const bfx8*)p.W16 + lane;    // [blk][term][lane]
  // stage: thread = (row tid>>4 (+16), vertex tid&15)
  const int fvx = tid & 15, fr = tid >> 4;
  f32x3 rg[2], rq[2];
  auto fetch = [&](int blk) {
    const int v = blk * 16 + fvx;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // Clamped addresses, no masking: a padding body has scale 0 (its s g, gv and products vanish), a padding
      // vertex has zero rows in Dt16 / W16; only the plain sum of g (the translation gradient) masks them, below.
      const int b = g * 32 + fr + 16 * k;
      const size_t o = ((size_t)(b < p.B ? b : p.B - 1) * p.V + (v < p.V ? v : p.V - 1)) * 3;
      rq[k] = *(const f32x3*)(p.vposed + o);
      rg[k] = (f32x3){0.f, 0.f, 0.f};
      if (p.gverts) rg[k] = *(const f32x3*)(p.gverts + o);
    }
  };
  // skinning rows and key-point head rows of the thread's vertex pair: requested AHEAD of the block's B operands
  // (memory returns in order: behind them they would wait for all 14 KB)
  int4 j4[2], kh[2];
  f32x4 w4[2];
  auto fetch_rows = [&](int blk) {
    const int vp = blk * 16 + 2 * vs;            // padded rows exist up to VP
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (NW4) {
        j4[u] = *(const int4*)(p.skidx + (size_t)(vp + u) * 4);
        w4[u] = *(const f32x4*)(p.skw + (size_t)(vp + u) * 4);
      }
      if (KP) kh[u] = *(const int4*)(p.kpv_head + (size_t)(vp + u) * 4);
    }
  };
  if (b0 < bend) fetch(b0);
  L_MARK(0);
  for (int blk = b0; blk < bend; ++blk) {
    // ---- stage ----
    fetch_rows(blk);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) sGv[(c * 16 + fvx) * SB_GS + fr + 16 * k] = rg[k][c];
    // B operands of this block (constant tables, L2 resident): requested here, used after the blend
    const bfx8* Dk = Db + (size_t)blk * 42 * 64;
    bfx8 bD[3][2][2], bW[2];
    auto fetch_b = [&](int c) {
      bD[c][0][0] = Dk[((c * 7 + ct0) * 2) * 64];
      bD[c][0][1] = Dk[((c * 7 + ct0) * 2 + 1) * 64];
      bD[c][1][0] = Dk[((c * 7 + ct1) * 2) * 64];
      bD[c][1][1] = Dk[((c * 7 + ct1) * 2 + 1) * 64];
    };
    fetch_b(0);
    fetch_b(1);
    fetch_b(2);
    const int vp = blk * 16 + 2 * vs;            // first vertex of the thread's pair
    SB_FENCE;
    L_MARK(1);
    __syncthreads();                             // (A) adjoint tile complete; every wave is past the last mfma phase
    L_MARK(2);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) sQ[(c * 32 + fr + 16 * k) * BW_TS + fvx] = rq[k][c];
    if (blk + 1 < bend) fetch(blk + 1);
    // ---- blend ----
    float gvp[3][2], gxp[3][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int vl = 2 * vs + u, v = vp + u;
      float g0 = sGv[(0 * 16 + vl) * SB_GS + bi], g1 = sGv[(1 * 16 + vl) * SB_GS + bi], g2 = sGv[(2 * 16 + vl) * SB_GS + bi];
      if (KP && kh[u].y > 0) {   // key-point regressor adjoint: dL/dverts += R^T dL/djoints (~670 entries over the mesh)
        {
          const float w = __builtin_bit_cast(float, kh[u].w);
          const float* gj = sGj + bi * 52 + kh[u].z * 3;
          g0 = fmaf(w, gj[0], g0);
          g1 = fmaf(w, gj[1], g1);
          g2 = fmaf(w, gj[2], g2);
        }
        for (int e = kh[u].x + 1; e < kh[u].x + kh[u].y; ++e) {
          const float w = p.kpv_w[e];
          const float* gj = sGj + bi * 52 + p.kpv_j[e] * 3;
          g0 = fmaf(w, gj[0], g0);
          g1 = fmaf(w, gj[1], g1);
          g2 = fmaf(w, gj[2], g2);
        }
      }
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = 0.f;
      if (NW4) {
        const int jj[4] = {j4[u].x, j4[u].y, j4[u].z, j4[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4* Aj = (const f32x4*)(sAb + jj[k] * 12);
          const f32x4 a0 = Aj[0], a1 = Aj[1], a2 = Aj[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T[e] = fmaf(w4[u][k], a0[e], T[e]);
            T[4 + e] = fmaf(w4[u][k], a1[e], T[4 + e]);
            T[8 + e] = fmaf(w4[u][k], a2[e], T[8 + e]);
          }
          if (k == 1) __builtin_amdgcn_sched_barrier(0);      // two bones (24 registers of LDS data) in flight at a time
        }
      } else {
        for (int k = 0; k < p.nw; ++k) {
          const int j = p.skidx[(size_t)v * p.nw + k];
          const float w = p.skw[(size_t)v * p.nw + k];
          const f32x4* Aj = (const f32x4*)(sAb + j * 12);
          const f32x4 a0 = Aj[0], a1 = Aj[1], a2 = Aj[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T[e] = fmaf(w, a0[e], T[e]);
            T[4 + e] = fmaf(w, a1[e], T[4 + e]);
            T[8 + e] = fmaf(w, a2[e], T[8 + e]);
          }
        }
      }
      if (v < p.V) { sT0 += g0; sT1 += g1; sT2 += g2; }
      // d/d v_posed = R_T^T (s g).  (The scale gradient sum_v g.x needs neither x nor v_posed here: it equals
      // (1/s) sum_j <A_j, dL/dA_j> and is taken from the joint-gradient sums in k_pose_bwd.)
      const float gx0 = sb * g0, gx1 = sb * g1, gx2 = sb * g2;
      gxp[0][u] = gx0; gxp[1][u] = gx1; gxp[2][u] = gx2;
      gvp[0][u] = fmaf(T[8], gx2, fmaf(T[4], gx1, T[0] * gx0));
      gvp[1][u] = fmaf(T[9], gx2, fmaf(T[5], gx1, T[1] * gx0));
      gvp[2][u] = fmaf(T[10], gx2, fmaf(T[6], gx1, T[2] * gx0));
      __builtin_amdgcn_sched_barrier(0);
    }
    bW[0] = Wb[(size_t)blk * 128];               // (their registers were the blend's; used last in the mfma phase)
    bW[1] = Wb[(size_t)blk * 128 + 64];
    {
      // the pair (vertices 2 vs, 2 vs + 1) is one dword of the fragment of lane (body, half vs>>2), dword vs&3
      const int fl = (vs >> 2) * 32 + bi, fd = vs & 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x2 v2 = {gvp[c][0], gvp[c][1]};
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bfx2));
        const f32x2 r2 = {v2[0] - __builtin_bit_cast(float, hi << 16), v2[1] - __builtin_bit_cast(float, hi & 0xffff0000u)};
        sF[((c * 2) * 64 + fl) * 4 + fd] = hi;
        sF[((c * 2 + 1) * 64 + fl) * 4 + fd] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bfx2));
        *(f32x2a*)(sGx + (c * 32 + bi) * BW_TS + 2 * vs) = (f32x2a){gxp[c][0], gxp[c][1]};
      }
    }
    L_MARK(3);
    __syncthreads();                             // (B) fragments, s g and posed vertices of the block are in LDS
    L_MARK(4);
    // ---- mfma ----
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const bfx8 ah = *(const bfx8*)(sF + ((c * 2) * 64 + lane) * 4);
      const bfx8 al = *(const bfx8*)(sF + ((c * 2 + 1) * 64 + lane) * 4);
      accF[0] = MFMA_BF16(ah, bD[c][0][0], accF[0]);
      accF[0] = MFMA_BF16(ah, bD[c][0][1], accF[0]);
      accF[0] = MFMA_BF16(al, bD[c][0][0], accF[0]);
      accF[1] = MFMA_BF16(ah, bD[c][1][0], accF[1]);
      accF[1] = MFMA_BF16(ah, bD[c][1][1], accF[1]);
      accF[1] = MFMA_BF16(al, bD[c][1][0], accF[1]);
      SB_FENCE2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int rc = 3 * wave + k, r = rc >> 2, c = rc & 3;      // wave-uniform
      const f32x4* gp = (const f32x4*)(sGx + (r * 32 + li) * BW_TS + 8 * lh);
      const f32x4 ga = gp[0], gb = gp[1];
      const f32x4* qp = (const f32x4*)(sQ + (c * 32 + li) * BW_TS + 8 * lh);      // c = 3: the row of ones
      const f32x4 qa = qp[0], qb = qp[1];
      float a[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = ga[t] * qa[t]; a[4 + t] = gb[t] * qb[t]; }
      bfx8 ah, al;
      mh_split_bf16x8(a, ah, al);
      accA[k] = MFMA_BF16(ah, bW[0], accA[k]);
      accA[k] = MFMA_BF16(ah, bW[1], accA[k]);
      accA[k] = MFMA_BF16(al, bW[0], accA[k]);
      SB_FENCE2;
    }
    L_MARK(5);
  }
  // ---- write the chunk's partial sums; element r of lane l is C[body (r&3) + 8 (r>>2) + 4 (l>>5)][column l&31] ----
  const size_t GB = (size_t)p.G * 32;
  const size_t base = (size_t)ch * GB + (size_t)g * 32;
  float* oF = p.pF + base * MH_FS;
  float* oA = p.pA + base * 288;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    oF[(size_t)row * MH_FS + (2 * wave) * 32 + li] = accF[0][r];
    if (two) oF[(size_t)row * MH_FS + (2 * wave + 1) * 32 + li] = accF[1][r];
    if (li < MH_NJ) {
#pragma unroll
      for (int k = 0; k < 3; ++k) oA[(size_t)row * 288 + (3 * wave + k) * MH_NJ + li] = accA[k][r];
    }
  }
  sTr[(vs * 32 + bi) * 3 + 0] = sT0;
  sTr[(vs * 32 + bi) * 3 + 1] = sT1;
  sTr[(vs * 32 + bi) * 3 + 2] = sT2;
  __syncthreads();
  if (tid < 128) {
    const int bb = tid >> 2, c = tid & 3;
    float s = 0.f;
    if (c < 3)
      for (int k = 0; k < 8; ++k) s += sTr[(k * 32 + bb) * 3 + c];
    p.pS[base * 4 + tid] = s;                    // [3] (the scale term) comes from the joint sums, see k_pose_bwd
  }
  L_OUT(8, 6);
}

struct PoseBwdP {
  int B, NB, G, CH;
  const float* betas;
  const float* poses;
  const float* gjoints;     // null ok
  const float* kp_rowsum;   // [17] (null when no key-point regressor)
  const float* rotmats;     // [B][24][9]: the forward ran on given rotation matrices (lbs(pose2rot=False)); null: axis-angle
  const float* gposed;      // [B][24][3] dL/d(posed joints = translation of the global joint transforms), null ok
  float* grotmats;          // [B][24][9] += (with rotmats)
  const float* scale;
  const float* A;        // [G*32][24][12] joint transforms of the forward (k_pose_fwd), rest pose removed
  const float* Jt;
  const float* JS;
  const float* pF;
  const float* pA;
  const float* pS;
  float* gposes;    // [B][72] +=
  float* gtransl;   // [B][3] += (null ok)
  float* gbeta_b;   // [G*32][10]
  float* gxs_b;     // [G*32]
  int scale_from_joint_sums;   // 1: d/dlog-scale = sum_j <A_j, dL/dA_j> (pS[3] is not filled by the split kernels)
  mh_tree tree;
  int has_fin;                 // 1: the FIRST workgroup of the launch carries the rasteriser's closing job instead of a body;
                               // 2: and the second one rebuilds its work lists
                               // (first: the body workgroups outnumber the slots of the device, the last one starts late)
  mh_raster_fin fin;
};

__global__ __launch_bounds__(256) void k_pose_bwd(PoseBwdP p) {
  if (p.has_fin && blockIdx.x == 0) {
    // the rasterised terms' closing job (mh_raster_fin): nothing in this launch reads what it writes, it only has to be
    // over before the update -- beside 800 body workgroups its 12 us of dependent loads cost nothing, as a launch of its
    // own between the gradient kernel and the backward they cost the chain 16
    __shared__ float f_g0[MH_FIN_U * 256], f_g1[MH_FIN_U * 256];
    mh_raster_finish_job<256>(p.fin, f_g0, f_g1);
    return;
  }
  if (p.has_fin == 2 && blockIdx.x == 1) {
    // ... and the rasteriser's work lists for the next launch on that workspace (a schedule: tile and unit order), which
    // used to be a launch of their own between the preparation and the selection kernel
    r_finalize_lists<256>(*(const RasterP*)p.fin.lists);
    return;
  }
  __shared__ float sGA[1][12 * MH_NJ];   // summed dL/dA  [e][j]
  __shared__ float sGF[1][MH_FS];        // summed dL/dfeat
  __shared__ float sG[1][MH_NJ][12];
  __shared__ float sgG[1][MH_NJ][12];
  __shared__ float sJ[1][MH_NJ][3];
  __shared__ float sgJ[1][MH_NJ][3];
  __shared__ float sS[1][4];
  const size_t GB = (size_t)p.G * 32;
  // 1. sum the chunk partials.  The sums used to be the kernel's longest phase (five dependent trips of 64 loads per thread
  // in 100 workgroups): half-wave g of the body's workgroup now takes the chunks g, g + 8, g + 16 ... (one trip), and the
  // eight partial sums are added in fixed order -- ((chunks = 0 mod 8) + (= 1 mod 8)) + ... -- from LDS.  Only half-wave 0
  // goes on from there (a barrier only waits for the waves that are still alive).
  __shared__ float sP[8][12 * MH_NJ + MH_FS + 4];
  const int g = threadIdx.x >> 5, j = threadIdx.x & 31, bl = 0;
  const int b = (int)blockIdx.x - p.has_fin;
  const bool valid = b < p.B, act = j < MH_NJ;
  // the state of the forward, asked for before the chunk partials so that its round trip hides behind theirs: rest joints,
  // rotations, and the rotation part of the global transforms as k_pose_fwd left it in A = [G.R | G.t - G.R.J] (the chain
  // itself is not walked again; nothing below needs G.t other than in that difference)
  float beta[MH_NUM_BETAS];
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, J[3] = {0, 0, 0}, th[3] = {0, 0, 0};
  float Gm[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  if (g == 0) {
#pragma unroll
    for (int l = 0; l < MH_NUM_BETAS; ++l) beta[l] = valid ? p.betas[(size_t)(b % p.NB) * MH_NUM_BETAS + l] : 0.f;
    if (valid && act) {
#pragma unroll
      for (int e = 0; e < 12; ++e) Gm[e] = p.A[((size_t)b * MH_NJ + j) * 12 + e];
      mh_joint_rest(p.Jt, p.JS, beta, j, J);
      if (p.rotmats) {
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = p.rotmats[((size_t)b * MH_NJ + j) * 9 + e];
      } else if (j < 22) {
        th[0] = p.poses[(size_t)b * 72 + 3 * j];
        th[1] = p.poses[(size_t)b * 72 + 3 * j + 1];
        th[2] = p.poses[(size_t)b * 72 + 3 * j + 2];
        mh_rodrigues(th, R);
      }
    }
  }
  {
    float aA[9], aF[7], aS = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) aA[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) aF[i] = 0.f;
    // up to four chunks (64 loads) in flight per round trip; a missing chunk re-reads the last one and adds zero
    for (int c0 = g; c0 < p.CH; c0 += 32) {
      float tA[4][9], tF[4][7], tS[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = min(c0 + 8 * u, p.CH - 1);
        const float* qa = p.pA + ((size_t)c * GB + b) * 288 + j;
        const float* qf = p.pF + ((size_t)c * GB + b) * MH_FS + j;
#pragma unroll
        for (int i = 0; i < 9; ++i) tA[u][i] = qa[32 * i];
#pragma unroll
        for (int i = 0; i < 7; ++i) tF[u][i] = qf[32 * i];
        tS[u] = p.pS[((size_t)c * GB + b) * 4 + (j & 3)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float m = c0 + 8 * u < p.CH ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) aA[i] = fmaf(m, tA[u][i], aA[i]);
#pragma unroll
        for (int i = 0; i < 7; ++i) aF[i] = fmaf(m, tF[u][i], aF[i]);
        aS = fmaf(m, tS[u], aS);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) sP[g][j + 32 * i] = aA[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) sP[g][288 + j + 32 * i] = aF[i];
    if (j < 4) sP[g][288 + MH_FS + j] = aS;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 12 * MH_NJ + MH_FS + 4; e += 256) {
    float a = sP[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) a += sP[k][e];
    if (e < 288) sGA[0][e] = a;
    else if (e < 288 + MH_FS) sGF[0][e - 288] = a;
    else sS[0][e - 288 - MH_FS] = (e - 288 - MH_FS == 3 && p.scale_from_joint_sums) ? 0.f : a;
  }
  __syncthreads();
  if (g != 0) return;
  // 2. parents' transforms and rest joints where the chain's adjoint finds them
  if (act) {
    sJ[bl][j][0] = J[0]; sJ[bl][j][1] = J[1]; sJ[bl][j][2] = J[2];
#pragma unroll
    for (int e = 0; e < 12; ++e) sG[bl][j][e] = Gm[e];
  }
  __syncthreads();
  const int par = act ? p.tree.parent[j] : -1;
  const int lev = act ? p.tree.level[j] : -1;
  float rel[3] = {J[0], J[1], J[2]};
  if (act && j > 0) {
    rel[0] -= sJ[bl][par][0]; rel[1] -= sJ[bl][par][1]; rel[2] -= sJ[bl][par][2];
  }
  // 3. adjoint of A = [G.R | G.t - G.R.J]
  float gA[12];
  if (act) {
#pragma unroll
    for (int e = 0; e < 12; ++e) gA[e] = sGA[bl][e * MH_NJ + j];
    if (p.scale_from_joint_sums) {
      // verts = s.x + t with x = sum_j w_j A_j [q;1]  =>  sum_v g.x = (1/s) sum_j <A_j, dL/dA_j>  (dL/dA_j carries the s)
      float d = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        d += gA[r * 4] * Gm[r * 4] + gA[r * 4 + 1] * Gm[r * 4 + 1] + gA[r * 4 + 2] * Gm[r * 4 + 2] + gA[r * 4 + 3] * Gm[r * 4 + 3];
      atomicAdd(&sS[bl][3], d / fmaxf(p.scale[valid ? b : 0], 1e-30f));
    }
    float gJ[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gt = gA[r * 4 + 3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sgG[bl][j][r * 4 + c] = gA[r * 4 + c] - gt * J[c];
        gJ[c] -= Gm[r * 4 + c] * gt;
      }
      // the posed joint IS the translation of G (smpl.py:735: joints_smpl24), so its adjoint joins here
      sgG[bl][j][r * 4 + 3] = gt + ((p.gposed && valid) ? p.gposed[((size_t)b * MH_NJ + j) * 3 + r] : 0.f);
    }
    sgJ[bl][j][0] = gJ[0]; sgJ[bl][j][1] = gJ[1]; sgJ[bl][j][2] = gJ[2];
  }
  __syncthreads();
  // 4. adjoint of the chain, deepest level first
  float gR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int l = p.tree.maxlevel; l >= 1; --l) {
    if (act && lev == l) {
      float gG[12], Gp[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) { gG[e] = sgG[bl][j][e]; Gp[e] = sG[bl][par][e]; }
      float grel[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        grel[c] = Gp[0 * 4 + c] * gG[3] + Gp[1 * 4 + c] * gG[7] + Gp[2 * 4 + c] * gG[11];
#pragma unroll
        for (int d = 0; d < 3; ++d)   // gR = Gp.R^T gG.R
          gR[c * 3 + d] = Gp[0 * 4 + c] * gG[0 * 4 + d] + Gp[1 * 4 + c] * gG[1 * 4 + d] + Gp[2 * 4 + c] * gG[2 * 4 + d];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // gGp.R += gG.R R^T + gG.t (x) rel
          const float a = gG[r * 4 + 0] * R[c * 3 + 0] + gG[r * 4 + 1] * R[c * 3 + 1] + gG[r * 4 + 2] * R[c * 3 + 2] + gG[r * 4 + 3] * rel[c];
          atomicAdd(&sgG[bl][par][r * 4 + c], a);
        }
        atomicAdd(&sgG[bl][par][r * 4 + 3], gG[r * 4 + 3]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        atomicAdd(&sgJ[bl][j][c], grel[c]);
        atomicAdd(&sgJ[bl][par][c], -grel[c]);
      }
    }
    __syncthreads();
  }
  if (act && j == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      gR[r * 3] = sgG[bl][0][r * 4]; gR[r * 3 + 1] = sgG[bl][0][r * 4 + 1]; gR[r * 3 + 2] = sgG[bl][0][r * 4 + 2];
      sgJ[bl][0][r] += sgG[bl][0][r * 4 + 3];
    }
  }
  __syncthreads();
  // 5. pose-blend features are R_j - I for j >= 1
  if (act && j >= 1) {
#pragma unroll
    for (int e = 0; e < 9; ++e) gR[e] += sGF[bl][10 + (j - 1) * 9 + e];
  }
  // 6. given rotation matrices: their adjoint is the result (all 24 joints); else the adjoint of rodrigues (joints 0..21)
  if (p.rotmats) {
    if (valid && act && p.grotmats) {
#pragma unroll
      for (int e = 0; e < 9; ++e) p.grotmats[((size_t)b * MH_NJ + j) * 9 + e] += gR[e];
    }
  } else if (valid && act && j < 22) {
    const float e = 1e-8f;
    const float a0 = th[0] + e, a1 = th[1] + e, a2 = th[2] + e;
    const float ang = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float x = th[0] / ang, y = th[1] / ang, z = th[2] / ang;
    const float s = sinf(ang), c = cosf(ang), omc = 1.f - c;
    const float K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    const float K2[9] = {-z * z - y * y, x * y, x * z, x * y, -z * z - x * x, y * z, x * z, y * z, -y * y - x * x};
    float ga = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) ga += gR[i] * (c * K[i] + s * K2[i]);
    // gK = s gR + (1-c)(gR K^T + K^T gR)
    float gK[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        float t = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) t += gR[r * 3 + k] * K[cc * 3 + k] + K[k * 3 + r] * gR[k * 3 + cc];
        gK[r * 3 + cc] = s * gR[r * 3 + cc] + omc * t;
      }
    const float gd0 = gK[7] - gK[5], gd1 = gK[2] - gK[6], gd2 = gK[3] - gK[1];
    const float gat = ga - (gd0 * th[0] + gd1 * th[1] + gd2 * th[2]) / (ang * ang);
    float* o = p.gposes + (size_t)b * 72 + 3 * j;
    o[0] += gd0 / ang + gat * a0 / ang;
    o[1] += gd1 / ang + gat * a1 / ang;
    o[2] += gd2 / ang + gat * a2 / ang;
  }
  // 7. shape: direct rows of the basis + the joint-location path
  // (the 72 products of a coefficient in three runs of 24 on lanes j, j + 10, j + 20, added in that order)
  {
    float a = 0.f;
    if (valid && j < 3 * MH_NUM_BETAS) {
      const int l = j % MH_NUM_BETAS, q0 = (j / MH_NUM_BETAS) * (MH_NJ * 3 / 3);
#pragma unroll 8
      for (int q = q0; q < q0 + MH_NJ * 3 / 3; ++q) a = fmaf(p.JS[q * MH_NUM_BETAS + l], sgJ[bl][q / 3][q % 3], a);
    }
    sP[0][j] = a;
  }
  __syncthreads();
  if (b < p.G * 32 && j < MH_NUM_BETAS)
    p.gbeta_b[(size_t)b * MH_NUM_BETAS + j] = valid ? ((sGF[bl][j] + sP[0][j]) + sP[0][j + MH_NUM_BETAS]) + sP[0][j + 2 * MH_NUM_BETAS] : 0.f;
  // 8. translation and scale
  if (b < p.G * 32 && j == 31) p.gxs_b[b] = valid ? sS[bl][3] * p.scale[b] * 0.0953101798043249f /* ln 1.1 */ : 0.f;
  if (valid && j >= 24 && j < 27 && p.gtransl) {
    const int c = j - 24;
    float a = sS[bl][c];
    if (p.gjoints && p.kp_rowsum)
      for (int q = 0; q < MH_NKP; ++q) a = fmaf(1.f - p.kp_rowsum[q], p.gjoints[((size_t)b * MH_NKP + q) * 3 + c], a);
    p.gtransl[(size_t)b * 3 + c] += a;
  }
}

// per-person reduction over frames of the per-body shape/scale gradients (fixed order)
__global__ __launch_bounds__(256) void k_person_reduce(int B, int NB, const float* gbeta_b, const float* gxs_b,
                                                       float* gbetas, float* gxscale) {
  __shared__ float s[256];
  const int n = blockIdx.x, q = blockIdx.y;   // q < 10: beta component, q == 10: scale
  float a = 0;
  for (int b = n + threadIdx.x * NB; b < B; b += 256 * NB) a += (q < 10) ? gbeta_b[(size_t)b * MH_NUM_BETAS + q] : gxs_b[b];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (q < 10) {
      if (gbetas) gbetas[(size_t)n * MH_NUM_BETAS + q] += s[0];
    } else if (gxscale) {
      gxscale[n] += s[0];
    }
  }
}

static int bwd16_chunks(int G) {
  int ch = (500 + G / 2) / G;          // ~ two workgroups per CU
  if (ch < 1) ch = 1;
  if (ch > 54) ch = 54;
  return ch;
}

static int bwd_chunks(int G16) {
  int ch = (512 + G16 / 2) / G16;
  if (ch < 1) ch = 1;
  if (ch > 27) ch = 27;
  return ch;
}

struct BwdWs {
  float* pF;
  float* pA;
  float* pS;
  float* gbeta_b;
  float* gxs_b;
};
static BwdWs carve_bwd(void* ws, int G, int CH) {
  char* p = (char*)ws;
  BwdWs w;
  const size_t GB = (size_t)G * 32;
  w.pF = (float*)p; p += align256((size_t)CH * GB * MH_FS * 4);
  w.pA = (float*)p; p += align256((size_t)CH * GB * 288 * 4);
  w.pS = (float*)p; p += align256((size_t)CH * GB * 4 * 4);
  w.gbeta_b = (float*)p; p += align256(GB * MH_NUM_BETAS * 4);
  w.gxs_b = (float*)p;
  return w;
}

extern "C" size_t mh_lbs_backward_workspace_bytes(int B) {
  // (+ 1: the chunk slot mh_keypoint_terms fills)
  const int G = mh_groups(B < 1 ? 1 : B), CH = std::max(bwd_chunks(2 * G), bwd16_chunks(G)) + 1;
  const size_t GB = (size_t)G * 32;
  return align256((size_t)CH * GB * MH_FS * 4) + align256((size_t)CH * GB * 288 * 4) + align256((size_t)CH * GB * 16) +
         align256(GB * MH_NUM_BETAS * 4) + align256(GB * 4);
}

static int lbs_backward_impl(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                             const float* vposed, const float* gverts, const float* gjoints, const float* gposed,
                             float* gposes, float* grotmats, float* gtransl, float* gbetas, float* gxscale,
                             void* ws, void* ws2, void* stream, int kp_chunk = 0, const mh_raster_fin* fin = nullptr);

// The backward after mh_keypoint_terms: the key-point term's dL/dA, dL/dfeat and dL/dt are already in the extra chunk slot of
// ws2 (k_pose_bwd adds it behind the vertex chunks); the skinning kernel runs without key-point adjoints.
extern "C" int mh_lbs_backward_kp(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* vposed,
                                  const float* gverts, float* gposes, float* gtransl, float* gbetas, float* gxscale,
                                  void* ws, void* ws2, void* stream) {
  MH_CHECK(poses && gposes, "null argument");
  return lbs_backward_impl(m, B, NB, betas, poses, nullptr, vposed, gverts, nullptr, nullptr, gposes, nullptr, gtransl, gbetas,
                           gxscale, ws, ws2, stream, 1);
}

extern "C" int mh_lbs_backward_kp_fin(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* vposed,
                                      const float* gverts, float* gposes, float* gtransl, float* gbetas, float* gxscale,
                                      void* ws, void* ws2, const mh_raster_fin* fin, void* stream) {
  MH_CHECK(poses && gposes, "null argument");
  return lbs_backward_impl(m, B, NB, betas, poses, nullptr, vposed, gverts, nullptr, nullptr, gposes, nullptr, gtransl, gbetas,
                           gxscale, ws, ws2, stream, 1, fin);
}

int mh_lbs_backward_extra_slot(const mh_model* m, int B, void* ws2, float** pF, float** pA, float** pS) {
  MH_CHECK(m && ws2 && B > 0, "null argument");
  const bool split16 = lbs_mode() != 0;
  const int G = mh_groups(B), CH = split16 ? bwd16_chunks(G) : bwd_chunks(2 * G);
  BwdWs bw = carve_bwd(ws2, G, CH + 1);
  const size_t GB = (size_t)G * 32;
  *pF = bw.pF + (size_t)CH * GB * MH_FS;
  *pA = bw.pA + (size_t)CH * GB * 288;
  *pS = bw.pS + (size_t)CH * GB * 4;
  return MH_OK;
}

extern "C" int mh_lbs_backward_person_partials(const mh_model* m, int B, void* ws2, float** gbeta_b, float** gxs_b) {
  MH_CHECK(m && ws2 && gbeta_b && gxs_b, "null argument");
  MH_CHECK(B > 0, "B must be positive");
  const bool split16 = lbs_mode() != 0;
  const int G = mh_groups(B), CH = split16 ? bwd16_chunks(G) : bwd_chunks(2 * G);
  BwdWs bw = carve_bwd(ws2, G, CH + 1);
  *gbeta_b = bw.gbeta_b;
  *gxs_b = bw.gxs_b;
  return MH_OK;
}

extern "C" int mh_lbs_person_reduce(const mh_model* m, int B, int NB, void* ws2, float* gbetas, float* gxscale, void* stream) {
  float *gb = nullptr, *gx = nullptr;
  const int rc = mh_lbs_backward_person_partials(m, B, ws2, &gb, &gx);
  if (rc != MH_OK) return rc;
  MH_CHECK(NB > 0, "NB must be positive");
  if (!gbetas && !gxscale) return MH_OK;
  hipLaunchKernelGGL(k_person_reduce, dim3(NB, 11), dim3(256), 0, (hipStream_t)stream, B, NB, (const float*)gb, (const float*)gx, gbetas, gxscale);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_lbs_backward(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                               const float* xscale, const float* transl, const float* vposed, const float* gverts,
                               const float* gjoints, float* gposes, float* gtransl, float* gbetas, float* gxscale,
                               void* ws, void* ws2, void* stream) {
  (void)xscale; (void)transl;
  MH_CHECK(poses && gposes, "null argument");
  return lbs_backward_impl(m, B, NB, betas, poses, nullptr, vposed, gverts, gjoints, nullptr, gposes, nullptr, gtransl, gbetas,
                           gxscale, ws, ws2, stream);
}

extern "C" int mh_lbs_backward_ex(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                                  const float* vposed, const float* gverts, const float* gjoints, const float* gposed,
                                  float* gposes, float* grotmats, float* gtransl, float* gbetas, float* gxscale,
                                  void* ws, void* ws2, void* stream) {
  MH_CHECK((poses != nullptr) != (rotmats != nullptr), "exactly one of poses (axis-angle) and rotmats");
  MH_CHECK(poses ? gposes != nullptr : grotmats != nullptr, "null gradient output");
  return lbs_backward_impl(m, B, NB, betas, poses, rotmats, vposed, gverts, gjoints, gposed, gposes, grotmats, gtransl, gbetas,
                           gxscale, ws, ws2, stream);
}

static int lbs_backward_impl(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                             const float* vposed, const float* gverts, const float* gjoints, const float* gposed,
                             float* gposes, float* grotmats, float* gtransl, float* gbetas, float* gxscale,
                             void* ws, void* ws2, void* stream, int kp_chunk, const mh_raster_fin* fin) {
  MH_CHECK(m && betas && vposed && ws && ws2, "null argument");
  MH_CHECK(gverts || gjoints || kp_chunk, "need gverts and/or gjoints");
  MH_CHECK(B > 0 && NB > 0, "B and NB must be positive");
  MH_CHECK(!gjoints || m->reg[MH_REG_ALPHAPOSE].J == MH_NKP, "gjoints needs the key-point regressor");
  hipStream_t st = (hipStream_t)stream;
  const bool split16 = lbs_mode() != 0;
  const int G = mh_groups(B), G16 = 2 * G, CH = split16 ? bwd16_chunks(G) : bwd_chunks(G16);
  FwdWs fw = carve_fwd(ws, G);
  BwdWs bw = carve_bwd(ws2, G, CH + 1);      // (+ 1: the chunk slot of mh_keypoint_terms, always laid out)
  if (split16) {
    Bwd16P sp;
    sp.B = B; sp.G = G; sp.V = m->V; sp.VP = m->VP; sp.nw = m->nw; sp.CH = CH;
    sp.PB = (m->VP / 16 + CH - 1) / CH;
    sp.A = fw.A; sp.scale = fw.scale; sp.vposed = vposed; sp.gverts = gverts; sp.gjoints = gjoints;
    sp.Dt16 = m->Dt16; sp.W16 = m->W16; sp.skidx = m->skidx; sp.skw = m->skw;
    sp.kpv_ptr = m->kpv_ptr; sp.kpv_j = m->kpv_j; sp.kpv_w = m->kpv_w; sp.kpv_head = m->kpv_head;
    sp.pF = bw.pF; sp.pA = bw.pA; sp.pS = bw.pS;
    const size_t ldsB = (size_t)SB_LDS_FLOATS * 4;
    static unsigned char attr_b16[MH_MAX_DEVICES];
    if (mh_first_on_device(attr_b16)) {
      const void* fk[4] = {(const void*)k_skinbwd16<false, false>, (const void*)k_skinbwd16<false, true>,
                           (const void*)k_skinbwd16<true, false>, (const void*)k_skinbwd16<true, true>};
      for (int i = 0; i < 4; ++i) MH_HIP(hipFuncSetAttribute(fk[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
    }
    const bool nw4 = m->nw == 4, kp = gjoints != nullptr;
    auto kern = nw4 ? (kp ? k_skinbwd16<true, true> : k_skinbwd16<true, false>)
                    : (kp ? k_skinbwd16<false, true> : k_skinbwd16<false, false>);
    mh_prof_mark(MH_PROF_SKIN_BWD, 0, st);
    hipLaunchKernelGGL(kern, dim3(8 * ((G * CH + 7) / 8)), dim3(256), ldsB, st, sp);
    MH_LAUNCH_CHECK();
    mh_prof_mark(MH_PROF_SKIN_BWD, 1, st);
  } else {
  static unsigned char attr_set[MH_MAX_DEVICES];
  const size_t lds = (size_t)(16 * BWD_AS + 16 * 52 + BWD_NACC * 4 * 64 + 64) * 4;
  if (mh_first_on_device(attr_set))
    MH_HIP(hipFuncSetAttribute((const void*)k_skin_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SkinBwdP sp;
  sp.B = B; sp.G16 = G16; sp.GB = G * 32; sp.V = m->V; sp.VP = m->VP; sp.nw = m->nw; sp.CH = CH;
  sp.PQ = (m->VP / 4 + CH - 1) / CH;
  sp.A = fw.A; sp.scale = fw.scale; sp.vposed = vposed; sp.gverts = gverts; sp.gjoints = gjoints;
  sp.Dt = m->Dt; sp.skidx = m->skidx; sp.skw = m->skw;
  sp.kpv_ptr = m->kpv_ptr; sp.kpv_j = m->kpv_j; sp.kpv_w = m->kpv_w;
  sp.pF = bw.pF; sp.pA = bw.pA; sp.pS = bw.pS;
  mh_prof_mark(MH_PROF_SKIN_BWD, 0, st);
  hipLaunchKernelGGL(k_skin_bwd, dim3(G16, CH), dim3(256), lds, st, sp);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_SKIN_BWD, 1, st);
  }
  PoseBwdP pp;
  pp.B = B; pp.NB = NB; pp.G = G; pp.CH = CH + (kp_chunk ? 1 : 0);
  pp.betas = betas; pp.poses = poses; pp.gjoints = gjoints;
  pp.rotmats = rotmats; pp.gposed = gposed; pp.grotmats = grotmats;
  pp.kp_rowsum = m->reg[MH_REG_ALPHAPOSE].rowsum;
  pp.scale = fw.scale; pp.A = fw.A; pp.Jt = m->Jt; pp.JS = m->JS;
  pp.pF = bw.pF; pp.pA = bw.pA; pp.pS = bw.pS;
  pp.gposes = gposes; pp.gtransl = gtransl; pp.gbeta_b = bw.gbeta_b; pp.gxs_b = bw.gxs_b;
  pp.scale_from_joint_sums = split16 ? 1 : 0;
  pp.tree = m->tree;
  pp.has_fin = (fin && fin->B > 0) ? (fin->has_lists ? 2 : 1) : 0;
  if (pp.has_fin) pp.fin = *fin; else memset(&pp.fin, 0, sizeof(pp.fin));
  mh_prof_mark(MH_PROF_POSE_BWD, 0, st);
  hipLaunchKernelGGL(k_pose_bwd, dim3(G * 32 + pp.has_fin), dim3(256), 0, st, pp);
  MH_LAUNCH_CHECK();
  if (gbetas || gxscale) {
    hipLaunchKernelGGL(k_person_reduce, dim3(NB, 11), dim3(256), 0, st, B, NB, bw.gbeta_b, bw.gxs_b, gbetas, gxscale);
    MH_LAUNCH_CHECK();
  }
  mh_prof_mark(MH_PROF_POSE_BWD, 1, st);
  return MH_OK;
}
