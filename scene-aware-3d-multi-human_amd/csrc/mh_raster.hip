// Differentiable z-buffer + soft silhouette of every body, fused with the depth and silhouette
// residuals and their backward (reference optimizer.py:425-477 + losses.py:19-40; the rasteriser
// itself is PyTorch3D's MeshRasterizer / SoftSilhouetteShader, restated from its published
// semantics -- see oracle/raster_select.c for the provenance note).
//
// MI355X design.  SMPL triangles are sub-pixel at MuPoTs resolution (13776 faces on a few
// hundred pixels, blur radius larger than a face), so rasterisation is FACE-parallel with the per-pixel K-nearest
// lists kept in LDS:
//   k_raster_prepare      one workgroup per body: NDC projection of the vertices + screen window; pixel-row range of every
//                         face and counting sort by first row (a tile's candidate faces become contiguous ranges) -- only
//                         when a vertex has moved more than `margin` rows since the body's last sort (temporal coherence);
//                         the body's tiles of <= R_CAP pixels with their cost classes
//   k_raster_lists        one workgroup: tiles ordered by cost class (longest first), work units of the gradient kernel --
//                         device-side lists, no host sync.  A SCHEDULE only: every body's keys live in a region of their
//                         own, and the kernels below find every tile / unit with lists that are a launch old (or empty);
//                         in the optimisation cycle the lists are therefore rebuilt off the chain, by a workgroup of the
//                         LBS backward's pose kernel (mh_raster_p.h, mh_raster_fin), and this kernel is not launched
//   k_raster_strip        one workgroup per tile; every wave runs barrier-free rounds of 64 faces (3-deep gather
//                         pipeline, bbox, pair list / even split, depth cull) and inserts 64-bit (z, face) keys into
//                         the tile's LDS window with ds_min_u64: slot 0 = nearest face of the blur 1e-4 pass (all
//                         the reference reads of its K=8 rasterisation, optimizer.py:430), slots 1..4 = the K=4
//                         nearest of the blur 2e-5 silhouette pass (atomic-min cascade: the displaced key moves on).
//                         The finished window is written to HBM once (40 B per window pixel).
//   k_raster_sums         per tile: residual sums of the depth and silhouette terms
//   k_raster_finish       per frame: per-body loss values from the tile sums, chain of the depth-range leaves, log sums
//                         (the closing job of mh_common.h; in the cycle it rides in the pose kernel too)
//   k_raster_grads        per work unit (2048 window pixels of one body): live-pixel compaction, exact re-evaluation
//                         of the selected faces, gradients scattered to an LDS vertex table, flushed with atomics
// No (b,N,H,W,K) fragment tensor, z-buffer or alpha image is materialised (the reference builds
// two of them per batch).
#include "mh_common.h"
#include "mh_raster_p.h"
#include "mh_experiment.h"

#define RS_EMPTY 0xffffffffffffffffull
#define R_KEPS 1e-8f
#define BLUR_D 1e-4f         // optimizer.py:213
#define BLUR_S 2e-5f         // optimizer.py:223
#define SIGMA_S 1e-4f        // BlendParams.sigma default used by SoftSilhouetteShader
#ifndef RB
#define RB 512               // threads per strip workgroup
#endif
#ifndef RMINW
#define RMINW 4
#endif
// window pixels per tile (5 x u64 each = 28 KB of LDS; 2 workgroups per CU).  Larger tiles = fewer of them: less work repeated
// for the faces that reach across tiles, more rounds per wave and tile; but the selection kernel shares its CUs' LDS with the
// side branch's kernels.  Round 4 sweep with the pair list's size (same box, ms per cycle): 448 / 512 px 0.707 / 0.694, 640 px
// with 512 entries (rounds 2-4) 0.679 and 0.669 on a second box, 704 / 384 0.670 and 0.658, 704 / 256 0.671, 768 / 384 0.665,
// 800 / 512 0.667 (kernel alone 333 us instead of 357: the side branch then waits for LDS), 1408 px in 1024-thread
// workgroups 0.725.
#ifndef R_CAP
#define R_CAP 704
#endif
#define RT 13                // floats staged per face: 9 NDC coordinates, 1/area, 1/|edge|^2 x 3

// order-preserving int of a float (and back): a < b <=> r_ord(a) < r_ord(b)
__device__ __forceinline__ int r_ord(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float r_unord(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }
#define RS_TAG(b, m) (0x5bd1e995c0ffee00ull ^ ((unsigned long long)(b) * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)(m))

// NDC coordinate of a pixel centre (PixToNonSquareNdc of the CPU rasteriser): every operation rounded on its own.  The
// compiler's contraction of range * i + offset into one fma moves a centre by up to 1.3e-7 -- nothing for the selection
// (one more near-tie convention), but 2.6e-5 of a 0.3-pixel face, and the barycentric Jacobian of such a face turns that
// into 5e-4 of the largest entry of dL/dverts: the last systematic difference to the float64 oracle that
// tools/fuzz_raster_grads.py / tools/grad_debug.py found (round 3).
// (#pragma clang fp contract(off) is what keeps the operations apart: hip's __fmul_rn / __fadd_rn are plain operators.)
__device__ __forceinline__ float r_pix_to_ndc(int i, int S1, int S2) {
#pragma clang fp contract(off)
  float range = 2.0f;
  if (S1 > S2) range = ((float)S1 * range) / (float)S2;
  const float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}
// float pixel index (image order) of an NDC coordinate
__device__ __forceinline__ float r_ndc_to_pix(float ndc, int S1, int S2) {
  float range = 2.0f;
  if (S1 > S2) range = ((float)S1 * range) / (float)S2;
  return (float)S1 - 0.5f - (ndc + range / 2.0f) * (float)S1 / range;
}
__device__ __forceinline__ float r_edge(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
// the same without operation fusing: used where the VALUE decides something discrete in two different places (the
// degenerate-area exclusion of a face is taken by the face sort or by the strip kernel, depending on whether the lists
// are kept; the compiler contracts a * b - c * d into an fma differently from context to context, and a sliver near the
// 1e-8 threshold then existed for one and not for the other: 1 pixel of 850 000 differed, tests/test_full_size_gpu.py)
__device__ __forceinline__ float r_edge_exact(float px, float py, float ax, float ay, float bx, float by) {
#pragma clang fp contract(off)
  const float a = (px - ax) * (by - ay), b = (py - ay) * (bx - ax);
  return a - b;
}
// squared distance to segment ab; returns the clamped parameter in *t (deg: degenerate segment)
__device__ __forceinline__ float r_seg(float px, float py, float ax, float ay, float bx, float by, float* t, bool* deg) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= R_KEPS) {
    *t = 1.f;
    *deg = true;
    return (px - bx) * (px - bx) + (py - by) * (py - by);
  }
  float tt = (bax * (px - ax) + bay * (py - ay)) / l2;
  tt = fminf(fmaxf(tt, 0.f), 1.f);
  *t = tt;
  *deg = false;
  const float qx = ax + tt * bax - px, qy = ay + tt * bay - py;
  return qx * qx + qy * qy;
}

struct Tri {
  float x[3], y[3], z[3];     // NDC xy + view z
  int idx[3];
};

// a face from the projected-vertex buffer written by k_raster_windows (no divisions)
__device__ __forceinline__ void r_load_tri_ndc(const RasterP& p, const float* nb, int f, Tri& t) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int vi = p.faces[3 * f + k];
    t.idx[k] = vi;
    t.x[k] = nb[(size_t)vi * 3]; t.y[k] = nb[(size_t)vi * 3 + 1]; t.z[k] = nb[(size_t)vi * 3 + 2];
  }
}

// The gradient kernel divides by a handful of denominators many times over (1/area, 1/sum of clipped weights, 1/Z,
// 1/|edge|^2): one v_rcp_f32 (1 ulp) and multiplications instead of IEEE divisions (~10 instructions each, ~100 of
// them per live pixel before).  Only the gradient arithmetic is affected; which faces were selected, and the depth
// they were selected with, come from the exact forward pass.
__device__ __forceinline__ float r_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// squared distance to segment ab as r_seg, with the reciprocal
__device__ __forceinline__ float r_seg_rcp(float px, float py, float ax, float ay, float bx, float by, float* t, bool* deg) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= R_KEPS) {
    *t = 1.f;
    *deg = true;
    return (px - bx) * (px - bx) + (py - by) * (py - by);
  }
  float tt = (bax * (px - ax) + bay * (py - ay)) * r_rcp(l2);
  tt = fminf(fmaxf(tt, 0.f), 1.f);
  *t = tt;
  *deg = false;
  const float qx = ax + tt * bax - px, qy = ay + tt * bay - py;
  return qx * qx + qy * qy;
}

// Gradient scatter.  One workgroup owns one body: every contribution of the body's window pixels is
// summed into an LDS table indexed directly by the vertex (V x 3 floats = 82 KB for SMPL, LDS float
// atomics), then added to dL/dverts with plain coalesced read-modify-writes -- no global atomics.
// Bodies whose table does not fit in LDS (V > RG_MAXV) scatter with global atomics instead.
#define RG_MAXV 11500
#ifndef RG_WAVE_SUMS
#define RG_WAVE_SUMS 1
#endif
#ifndef RG_CAS
#define RG_CAS 1             // 0: ds_add_f32 (the form of rounds 2-5)
#endif
#define RG_LIST RG_UNIT
// dynamic LDS of the gradient kernel: [V][3] gradient table when it fits, then the live-pixel list.  The scatter
// goes through this symbol (not through a pointer chosen at run time) so that the compiler emits ds_add_f32 rather
// than flat atomics.
extern __shared__ __attribute__((aligned(16))) float rg_tab[];
// MODE 0: global float atomics (V > RG_MAXV); 1: LDS float table (production); 2: DETERMINISTIC -- the contributions of
// one body go into an LDS table of 64-bit fixed-point integers (integer addition is associative: any order of the
// atomics gives the same bits), in three passes over the body's pixels: pass 0 finds the largest contribution (the
// block exponent), passes 1 / 2 accumulate the lower / upper half of the vertices (the int64 table of all 6890 vertices
// is 1.5 KB larger than the LDS).  MHHIP_DETERMINISTIC=1 / mh_raster_set_deterministic(1).
struct RgDet {
  int pass, shift, v0, v1;
  float umax;
};
template <int MODE>
__device__ __forceinline__ void r_acc_add(float* gvb, int vid, float gx, float gy, float gz, RgDet& dc) {
  if (MODE == 2) {
    if (dc.pass == 0) {
      dc.umax = fmaxf(dc.umax, fmaxf(fabsf(gx), fmaxf(fabsf(gy), fabsf(gz))));
      return;
    }
    if (vid < dc.v0 || vid >= dc.v1) return;
    unsigned long long* o = (unsigned long long*)rg_tab + (size_t)(vid - dc.v0) * 3;
    atomicAdd(o, (unsigned long long)__double2ll_rn(ldexp((double)gx, dc.shift)));
    atomicAdd(o + 1, (unsigned long long)__double2ll_rn(ldexp((double)gy, dc.shift)));
    atomicAdd(o + 2, (unsigned long long)__double2ll_rn(ldexp((double)gz, dc.shift)));
  } else if (MODE == 1) {
#if RG_CAS
    // ds_add_f32 costs ~3 cycles per ACTIVE LANE on gfx950 (192 for a full wave instruction; integer LDS atomics and
    // compare-and-swaps ~4.5 per instruction whatever the lanes: profiles/r04_ubench_valu_lds.txt): the float add as a
    // read + compare-and-swap, the three components of a vertex in flight together, repeated for the lanes that lost
    // (the lanes of a wave are neighbouring pixels and share vertices).  Round 6: 99.3 -> 92.3 us in the cycle.
    unsigned* u = (unsigned*)&rg_tab[vid * 3];
    unsigned o0 = u[0], o1 = u[1], o2 = u[2];
    bool d0 = gx == 0.f, d1 = gy == 0.f, d2 = gz == 0.f;
    while (!(d0 && d1 && d2)) {
      unsigned r0 = o0, r1 = o1, r2 = o2;
      if (!d0) r0 = atomicCAS(u, o0, __float_as_uint(__uint_as_float(o0) + gx));
      if (!d1) r1 = atomicCAS(u + 1, o1, __float_as_uint(__uint_as_float(o1) + gy));
      if (!d2) r2 = atomicCAS(u + 2, o2, __float_as_uint(__uint_as_float(o2) + gz));
      d0 = d0 || r0 == o0; d1 = d1 || r1 == o1; d2 = d2 || r2 == o2;
      o0 = r0; o1 = r1; o2 = r2;
    }
#else
    atomicAdd(&rg_tab[vid * 3], gx);
    atomicAdd(&rg_tab[vid * 3 + 1], gy);
    atomicAdd(&rg_tab[vid * 3 + 2], gz);
#endif
  } else {
    float* o = gvb + (size_t)vid * 3;
    atomicAdd(o, gx);
    atomicAdd(o + 1, gy);
    atomicAdd(o + 2, gz);
  }
}
// scatter d/d(ndc x, ndc y, z) of one vertex to camera space: x_ndc = -s X / Z + w1, so d x_ndc / dX = -s / Z and
// d x_ndc / dZ = s X / Z^2 = -(x_ndc - w1) / Z
template <int MODE>
__device__ __forceinline__ void r_scatter(const RasterP& p, float* gvb, const Tri& t, int k, float gxn, float gyn, float gz, RgDet& dc) {
  const float rz = r_rcp(t.z[k]);
  const float gx = -p.s * rz * gxn, gy = -p.s * rz * gyn;
  const float gzz = gz - ((t.x[k] - p.w1) * gxn + (t.y[k] - p.h1) * gyn) * rz;
  r_acc_add<MODE>(gvb, t.idx[k], gx, gy, gzz, dc);
}

__device__ __forceinline__ float r_block_sum(float v, float* sh) {
  v = mh_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float a = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sh[w];
  return a;
}

// one (face, pixel-centre) candidate: interpolated depth with clipped barycentrics, inside test and
// squared distance to the triangle (CheckPixelInsideFace).  T9 = x0,y0,z0,x1,y1,z1,x2,y2,z2 (NDC).
__device__ __forceinline__ void r_eval(const float* T9, float xf, float yf, float* pz, bool* inside, float* dist) {
  const float x0 = T9[0], y0 = T9[1], z0 = T9[2], x1 = T9[3], y1 = T9[4], z1 = T9[5], x2 = T9[6], y2 = T9[7], z2 = T9[8];
  const float area = r_edge(x2, y2, x0, y0, x1, y1) + R_KEPS;
  const float w0 = r_edge(xf, yf, x1, y1, x2, y2) / area;
  const float w1 = r_edge(xf, yf, x2, y2, x0, y0) / area;
  const float w2 = r_edge(xf, yf, x0, y0, x1, y1) / area;
  *inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  const float c0 = fmaxf(w0, 0.f), c1 = fmaxf(w1, 0.f), c2 = fmaxf(w2, 0.f);
  const float cs = fmaxf(c0 + c1 + c2, 1e-5f);
  *pz = (c0 / cs) * z0 + (c1 / cs) * z1 + (c2 / cs) * z2;
  float tt;
  bool dg;
  *dist = fminf(fminf(r_seg(xf, yf, x0, y0, x1, y1, &tt, &dg), r_seg(xf, yf, x0, y0, x2, y2, &tt, &dg)),
                r_seg(xf, yf, x1, y1, x2, y2, &tt, &dg));
}

// insert one candidate into the LDS window: slot 0 = nearest face of the blur-1e-4 pass, slots 1..4 =
// the 4 nearest faces of the blur-2e-5 pass (atomicMin cascade, the displaced key moves on)
__device__ __forceinline__ void r_insert(unsigned long long* q, float pz, bool inside, float d, int f) {
  if (pz < 0.f || (!inside && d >= BLUR_D)) return;
  unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)f;
  if (key < q[0]) atomicMin(&q[0], key);
  // a candidate behind the pixel's current 4th key changes nothing (the keys only ever decrease, and a key still travelling
  // down another lane's cascade can only make the 4th slot smaller than what is read here): ONE read instead of three
  // atomics that leave their slots as they were
  if ((inside || d < BLUR_S) && key < q[4]) {
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      const unsigned long long old = atomicMin(&q[k], key);
      if (old == RS_EMPTY) break;
      key = old > key ? old : key;
    }
  }
}

// continuous row coordinate of an NDC y as one fused multiply-add: row = ra - y * rk (r_ndc_to_pix spelled out costs two
// IEEE divisions per call; the difference is ~1e-5 px, inside the 1e-3 px guard of the face row ranges)
__device__ __forceinline__ void r_row_affine(const RasterP& p, float* ra, float* rk) {
  float range = 2.0f;
  if (p.H > p.W) range = ((float)p.H * range) / (float)p.W;
  *rk = (float)p.H / range;
  *ra = (float)p.H - 0.5f - 0.5f * (float)p.H;
}

// =============================================================================================
// windows and strips
// =============================================================================================
#define R_TILE_W 64          // tile width for windows wider than one LDS strip

// tile geometry of a window: full-width row strips while a row fits into LDS, column tiles otherwise
__device__ __forceinline__ void r_tiling(int ww, int wh, int* tw, int* th, int* ncol, int* nrow) {
  if (ww <= R_CAP) {
    *tw = ww;
    *th = max(1, R_CAP / ww);
  } else {
    *tw = R_TILE_W;
    *th = R_CAP / R_TILE_W;
  }
  *ncol = (ww + *tw - 1) / *tw;
  *nrow = (wh + *th - 1) / *th;
}

// =============================================================================================
// rasterisation of one strip into LDS
// =============================================================================================
// squared distance to segment ab with the staged 1/|ab|^2, parametrised from b: il = 0 marks a degenerate segment and
// then yields |p - b|^2, the reference's answer for that case, without a branch
__device__ __forceinline__ float r_seg_fast(float px, float py, float ax, float ay, float bx, float by, float il) {
#pragma clang fp contract(off)
  const float abx = ax - bx, aby = ay - by, dx = px - bx, dy = py - by;
  const float tt = fminf(fmaxf(fmaf(abx, dx, aby * dy) * il, 0.f), 1.f);
  const float qx = fmaf(tt, abx, -dx), qy = fmaf(tt, aby, -dy);
  return fmaf(qx, qx, qy * qy);
}

// max(x, 0) as ONE v_max_f32: fmaxf(x, 0.f) compiles to two (the first only quiets a signalling NaN, which no
// arithmetic result is), and v_max_f32 is a full-rate-only instruction on gfx950 (tools/ubench/valu_rate.hip: 4.1 cycles
// against 2.3 for fma / mul / add)
__device__ __forceinline__ float r_max0(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// the selection-time twin of r_eval: per-face reciprocals are staged once per face, the only
// per-pair reciprocal is v_rcp_f32 of the clipped weight sum.  Values agree with r_eval to ~1 ulp;
// the residual kernels re-evaluate the SELECTED faces with r_eval, so only near-tie orderings and
// blur-band membership can differ in the last ulp.
// (An early exit for candidates provably outside the blur band -- distance to the line of a violated
// edge -- was measured and does not pay inside a 64-wide divergent loop: some lane always survives.)
__device__ __forceinline__ bool r_eval_fast(const float* T, float xf, float yf, float* pz, bool* inside, float* dist, float* wmin = nullptr) {
  // Every multiply-add of this function is spelled out and the compiler's own contraction is off: the function is inlined
  // into two loops of k_raster_strip (pair list / even split), and left to itself hipcc fused a * b - c * d one way in
  // one copy and the other way in the other -- the same (face, pixel) pair then got a depth one ulp apart depending on
  // which path its round took (found in round 4 by moving rounds from one path to the other: 47 753 of 1.2 M window
  // pixels changed a last bit).
#pragma clang fp contract(off)
  const float x0 = T[0], y0 = T[1], z0 = T[2], x1 = T[3], y1 = T[4], z1 = T[5], x2 = T[6], y2 = T[7], z2 = T[8];
  const float ia = T[9];
  const float d0x = xf - x0, d0y = yf - y0, d1x = xf - x1, d1y = yf - y1, d2x = xf - x2, d2y = yf - y2;
  const float e0 = fmaf(d1x, y2 - y1, -(d1y * (x2 - x1)));      // edge(p; v1, v2)
  const float e1 = fmaf(d2x, y0 - y2, -(d2y * (x0 - x2)));      // edge(p; v2, v0)
  const float e2 = fmaf(d0x, y1 - y0, -(d0y * (x1 - x0)));      // edge(p; v0, v1)
  const float w0 = e0 * ia, w1 = e1 * ia, w2 = e2 * ia;
  const bool in = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  *inside = in;
  if (wmin) *wmin = fminf(fminf(fabsf(w0), fabsf(w1)), fabsf(w2));      // (experiment builds only: mh_experiment.h, R_NEAR_PROBE)
  const float c0 = r_max0(w0), c1 = r_max0(w1), c2 = r_max0(w2);
  const float ics = __builtin_amdgcn_rcpf(fmaxf((c0 + c1) + c2, 1e-5f));
  *pz = fmaf(c2 * ics, z2, fmaf(c1 * ics, z1, (c0 * ics) * z0));
  *dist = in ? 0.f
             : fminf(fminf(r_seg_fast(xf, yf, x0, y0, x1, y1, T[10]), r_seg_fast(xf, yf, x0, y0, x2, y2, T[11])),
                     r_seg_fast(xf, yf, x1, y1, x2, y2, T[12]));
  return true;
}

// wave64 inclusive scans on the DPP network (row shifts inside the rows of 16, then the two row broadcasts):
// six VALU ops, no LDS round trips (a __shfl_up scan is six ds_bpermute each followed by a wait)
__device__ __forceinline__ int r_wave_scan_add(int x) {
  int v = x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int r_wave_scan_max(int x) {              // values >= -1
  int v = x;
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false));
  return v;
}

// conservative pixel-row range of every face + counting sort of the body's visible faces by their first row (one
// workgroup per body): a tile's candidate faces are then one contiguous range of fsort (first row in
// [tile_row0 - tallest_face, tile_last_row]); entries are hi << 20 | face.
#ifndef RFS_U
#define RFS_U 4              // faces whose gathers are in flight together (first pass)
#endif
#define RFS_V 7              // row words fetched together (second pass)
// lo | hi << 16 with bit 15 = sign of the screen-space area (which side of the face looks at the camera)
__device__ __forceinline__ unsigned r_face_rows_xyz(const RasterP& p, float ra, float rk, const float (&x)[3], const float (&y)[3],
                                                     const float (&z)[3], float* zmin_out) {
  const float farea = r_edge(x[0], y[0], x[1], y[1], x[2], y[2]);     // its sign only orders the list (near class first)
  unsigned out = 1u;                                     // lo = 1 > hi = 0: skipped
  // The two exclusions of the rasteriser -- a vertex behind the camera, a degenerate screen-space area -- are NOT applied
  // here: k_raster_strip decides them from the current coordinates, in ONE place whether the lists are kept or fresh (a
  // sliver seen edge-on crosses the 1e-8 area threshold under the smallest motion, and two compilations of the same
  // a * b - c * d differ in the last bit: decided here for fresh lists and there for kept ones, 1-3 of 850 000 window
  // pixels came out different, tests/test_full_size_gpu.py).  The list holds every face with a valid row range.
  const bool ok = true;
  if (ok) {
    const float blur_d = sqrtf(BLUR_D);
    const float bymin = fminf(y[0], fminf(y[1], y[2])) - blur_d, bymax = fmaxf(y[0], fmaxf(y[1], y[2])) + blur_d;
    // rows whose pixel centre lies inside the blurred bbox (centres at the integers of the row coordinate), 1e-3 px
    // absorbs the rounding.  A tight range matters: the tallest face of a body sets how far above a tile its
    // candidate range starts
    const float lo = ceilf(fmaf(-bymax, rk, ra) - 1e-3f);
    float hi = floorf(fmaf(-bymin, rk, ra) + 1e-3f);
    const float mg = (float)p.margin;
    // A blurred range that contains NO pixel-centre row (hi = lo - 1: images under 100 rows, where the blur band is
    // narrower than a pixel) is skipped by a fresh sort -- but a KEPT list must hold the face: after moving a fraction of
    // a row it may cover row lo - 1 or row lo.  It is listed at row lo with one row; the tiles' margins then find it
    // wherever it can have gone.  (Found at 96x54 by a test that fits twice: hundreds of window pixels of every kept
    // launch had lost such faces; at the bench's 240x135 the band alone spans 1.35 rows and the case cannot occur.)
    if (mg > 0.f && hi == lo - 1.f) hi = lo;
    if (hi >= lo && hi >= -mg && lo <= (float)(p.H - 1) + mg) {          // (false for NaN rows)
      const unsigned ulo = (unsigned)fminf(fmaxf(lo, 0.f), (float)(p.H - 1)), uhi = (unsigned)fminf(fmaxf(hi, 0.f), (float)(p.H - 1));
      out = ulo | (uhi << 16) | (farea > 0.f ? 0x8000u : 0u);
    }
  }
  *zmin_out = fminf(z[0], fminf(z[1], z[2]));
  return out;
}
// ---------------------------------------------------------------------------------------------------------------------
// Work lists.  Every body owns a fixed range of tile slots (s = b * cap + k), so the per-body sums over a body's tiles
// keep their fixed order and a body's workgroup can write its tiles without knowing about the others.  The ORDER in
// which tiles (and the work units of the gradient kernel) are processed is a schedule only -- any permutation yields the
// same keys: longest-processing-time first by cost class (without it the last tiles to start were often among the most
// expensive and the selection ended ~40 % later than its work divided by the CU count).  The dense, ordered lists are
// put together by whichever workgroup of k_raster_prepare finishes LAST (a ticket), with all its threads:
// prefix sum of the window sizes (where a body's keys live), counting sort of the tiles by the classes their owners
// computed, the gradient units.  (Rounds 1-2 did this in one extra workgroup that also evaluated the cost classes -- a
// chain of dependent loads per tile, 45 us of serial work that the face sort used to hide and the kept face lists
// exposed; a first version of this round appended to per-class lists with atomics from every workgroup: four
// same-address round trips on every workgroup's tail, 30 us slower than the serial pass it replaced.)

// cost class of a tile from the body's face lists: ~7 ns per candidate face and ~76 ns per window pixel (measured, C3)
__device__ __forceinline__ int r_tile_class(const RasterP& p, const int* rs, int mh, int sy0, int nrows, int ncols) {
  const int H = p.H;
  const long long cmax = (long long)p.F + 11ll * R_CAP + 1;
  const int sy1 = sy0 + nrows - 1;
  mh = min(max(mh, 0), H);
  const int ra = max(0, sy0 - R_SHORT - p.margin), rt = max(0, sy0 - mh - p.margin), rb = min(sy1 + 1 + p.margin, H);
  const long long n = (long long)(rs[rb] - rs[ra]) + (long long)(rs[H + 1 + rb] - rs[H + 1 + ra]) + (long long)(rs[2 * (H + 1) + rb] - rs[2 * (H + 1) + ra]) +
                      (long long)(rs[3 * (H + 1) + rb] - rs[3 * (H + 1) + rt]);
  const long long cost = min(max(n, 0ll), (long long)p.F) + 11ll * nrows * ncols;
  return R_NCLS - 1 - (int)min((long long)(R_NCLS - 1), max(0ll, cost * R_NCLS / cmax));
}

// conservative pixel-row range of every face + counting sort of the body's faces by their first row, by the NT threads
// of the body's workgroup: a tile's candidate faces are then contiguous ranges of fsort (first row in
// [tile_row0 - tallest_face - margin, tile_last_row + margin]); entries are hi << 20 | face.
// Two classes per row: the faces whose class (sign of the screen-space area) is nearer to the camera on average (for a
// closed mesh: the ones looking at it) come first in fsort, so that a tile rasterises them first and the depth cull of
// k_raster_strip then removes most of the far-side candidates.  row_start: [2][H+1] (+ total), class-major in that order.
// WINNERS FIRST (round 5).  The depth cull of k_raster_strip only bites once a pixel's own K=4 list has filled, and the
// faces that fill it are nearly the same from launch to launch: the faces that ended the PREVIOUS launch in some pixel's
// five keys ("winners", ~18 % of a body's faces) go into a list of their own that every tile rasterises first -- after a
// tile's first rounds its 4th keys are close to their final values and the other 82 % of the faces meet a cull that is
// nearly as tight as it will get.  An ORDER only: every face is still decided from the current coordinates, the keys are
// the same bits whatever the list holds (tests/test_raster_winners_gpu.py: garbage keys, stale winners).  The class is
// assigned when a body's lists are sorted (3 % of the bodies per cycle in a steady sequence) from the keys the body's
// window holds at that moment, and stays with the kept lists; a body sorted before it had any keys sorts once more.
// (Measured on the way: seeding the cull with last launch's 4th DEPTHS as sentinels -- exact through verification and
// per-pixel repair -- evaluates 28 % fewer pairs and is 33-44 us SLOWER: a pixel's list flips between "four front faces"
// and "three front faces + one 20 cm behind" under the smallest motion, ~1.5 failed pixels per body and cycle, each a
// re-run of its row's faces.  Real keys of real faces cannot fail.)
#define R_WIN_MAXF 65536     // models with more faces have no winners' list (the bitmap lives in LDS: F / 8 bytes)
template <int NT>
__device__ __forceinline__ void r_face_sort(const RasterP& p, int b, int* hist /*LDS [4][H+1] + winner bitmap*/, int pww, int pwh,
                                            unsigned* pacc, unsigned long long& plast) {
  __shared__ int s_maxh, s_flip;
  __shared__ float s_z[2];
  __shared__ int s_n[2];
  const int tid = threadIdx.x, H = p.H, HB = H + 1;
  const float* nb = p.ndc + (size_t)b * p.V * 3;
  unsigned* fr = p.frows + (size_t)b * p.F;
  unsigned* fs = p.fsort + (size_t)b * p.F;
  int* rs = p.row_start + (size_t)b * (4 * HB + 1);
  unsigned* wbits = (unsigned*)(hist + 4 * HB);
  const bool use_win = p.F <= R_WIN_MAXF && p.winners_on;
  const int nwords = use_win ? (p.F + 31) / 32 : 0;
  for (int i = tid; i < 4 * HB + nwords; i += NT) hist[i] = 0;
  if (tid == 0) { s_maxh = 0; s_z[0] = s_z[1] = 0.f; s_n[0] = s_n[1] = 0; }
  __syncthreads();
  (void)pacc; (void)plast;
  // winners of the previous launch on this workspace: every face id in the five keys of the body's (previous) window pixels
  const bool have_prev = use_win && p.kvalid[b] != 0 && pww > 0 && pwh > 0 && pww <= p.W && pwh <= p.H;
  if (have_prev) {
    const unsigned long long* gk = p.gkeys + (size_t)p.body_koff[b] * 5;
    const int nk = pww * pwh * 5;
    for (int i = tid; i < nk; i += NT) {
      const unsigned f = (unsigned)gk[i];                  // (an empty key has face id all ones)
      if (f < (unsigned)p.F) atomicOr(&wbits[f >> 5], 1u << (f & 31u));
    }
  }
  // "sorted with the winners ATTEMPTED", not "found": a body that had keys once and has left the image since (empty previous
  // window) must not be taken for one that was sorted before it had any -- it would sort again on every launch
  if (tid == 0) p.wstate[b] = (use_win && p.kvalid[b] != 0) ? 1 : 0;
  int mh = 0, n0 = 0, n1 = 0;
  float z0 = 0.f, z1 = 0.f, ra, rk;
  r_row_affine(p, &ra, &rk);
  if (p.margin > 0) {                      // where the vertices are now = what the next launches measure their motion from
    float* rowb = p.rowb + (size_t)b * p.V;
    for (int v = tid; v < p.V; v += NT) rowb[v] = fmaf(-nb[(size_t)v * 3 + 1], rk, ra);
    if (tid == 0) p.sort_tag[b] = RS_TAG(b, p.margin);
  } else if (tid == 0) {
    p.sort_tag[b] = 0ull;                  // lists without the margin's slack: never to be kept by a later launch
  }
  __syncthreads();                         // (the bitmap is complete)
  // fourth list: the few TALL faces (more than R_SHORT rows: 5 % of them at C3, slivers and close-ups).  A tile must start
  // reading a list `tallest face of the list` rows above its first row; one list for all faces made every tile wade through
  // the faces of the nine rows above it (the mean tallest face) to find the handful that reach down -- 1.44x the entries
  // that really overlap a tile, 1.13x with the tall ones in a list of their own
  P_MARK(1);
  // row word of a face: lo (15 bits) | sign of the screen-space area << 15 | hi << 16 (15 bits) | winner << 31
  // histogram bins: 0 / 1 = short faces by the sign of their area, 2 = tall, 3 = winners (short)
  auto bin_of = [&](unsigned r) {
    const int lo = (int)(r & 0x7fffu), hi = (int)((r >> 16) & 0x7fffu);
    return hi - lo > R_SHORT ? 2 : ((r >> 31) ? 3 : (int)((r >> 15) & 1u));
  };
  auto tally = [&](unsigned r, float zm, bool live) {
    const int lo = (int)(r & 0x7fffu), hi = (int)((r >> 16) & 0x7fffu), sg = (int)((r >> 15) & 1u);
    if (live && lo <= hi) {
      atomicAdd(&hist[bin_of(r) * HB + lo], 1);
      mh = max(mh, hi - lo);
      if (sg) { z1 += zm; ++n1; } else { z0 += zm; ++n0; }
    }
  };
  // The gathers of RFS_U faces are in flight together: the sort is bound by the chain index load -> vertex gather ->
  // histogram of one face after the other.  (Measured in round 2: it stays at the rate of its 12-byte gathers; combining
  // the histogram atomics of a wave by ballot was 4x slower -- a wave's 64 faces fall into too many distinct bins.)
  for (int f0 = tid; f0 < p.F; f0 += RFS_U * NT) {
    int vi[RFS_U][3];
#pragma unroll
    for (int u = 0; u < RFS_U; ++u) {
      const int f = min(f0 + u * NT, p.F - 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) vi[u][k] = p.faces[3 * f + k];
    }
    float x[RFS_U][3], y[RFS_U][3], z[RFS_U][3];
#pragma unroll
    for (int u = 0; u < RFS_U; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* q = nb + (size_t)vi[u][k] * 3;
        x[u][k] = q[0]; y[u][k] = q[1]; z[u][k] = q[2];
      }
#pragma unroll
    for (int u = 0; u < RFS_U; ++u) {
      const int f = f0 + u * NT;
      float zm;
      unsigned r = r_face_rows_xyz(p, ra, rk, x[u], y[u], z[u], &zm);
      if (f < p.F) {
        if (use_win && ((wbits[f >> 5] >> (f & 31)) & 1u)) r |= 0x80000000u;
        fr[f] = r;
      }
      tally(r, zm, f < p.F);
    }
  }
  P_MARK(2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mh = max(mh, __shfl_xor(mh, o, 64));
    z0 += __shfl_xor(z0, o, 64); z1 += __shfl_xor(z1, o, 64);
    n0 += __shfl_xor(n0, o, 64); n1 += __shfl_xor(n1, o, 64);
  }
  if ((tid & 63) == 0) {
    atomicMax(&s_maxh, mh);
    atomicAdd(&s_z[0], z0); atomicAdd(&s_z[1], z1);
    atomicAdd(&s_n[0], n0); atomicAdd(&s_n[1], n1);
  }
  __syncthreads();
  if (tid == 0) {
    // class 1 first when it is the nearer one (or the only one)
    const float m0 = s_n[0] ? s_z[0] / (float)s_n[0] : 3e38f, m1 = s_n[1] ? s_z[1] / (float)s_n[1] : 3e38f;
    s_flip = m1 < m0 ? 1 : 0;
  }
  __syncthreads();
  const int flip = s_flip;
  if (tid < 64) {                                   // exclusive scan of the (list order, row) histogram by one wave
    int carry = 0;
    for (int base = 0; base < 4 * HB; base += 64) {
      const int i = base + tid;                     // position in the output order: winners, near short, far short, tall
      const int o = min(i / HB, 3), rrow = i - o * HB;
      const int bin = (o == 0 ? 3 : (o == 3 ? 2 : ((o - 1) ^ flip) & 1)) * HB + rrow;
      const int v = i < 4 * HB ? hist[bin] : 0;
      const int incl = r_wave_scan_add(v);
      if (i < 4 * HB) {
        hist[bin] = carry + incl - v;
        rs[i] = carry + incl - v;
      }
      carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (tid == 0) { p.maxh[b] = s_maxh; rs[4 * HB] = carry; }
  }
  __syncthreads();
  P_MARK(3);
  auto place = [&](unsigned r, int f, bool live) {
    const int lo = (int)(r & 0x7fffu), hi = (int)((r >> 16) & 0x7fffu);
    if (live && lo <= hi) fs[atomicAdd(&hist[bin_of(r) * HB + lo], 1)] = ((unsigned)hi << 20) | (unsigned)f;
  };
  for (int f0 = tid; f0 < p.F; f0 += RFS_V * NT) {        // the row words of RFS_V faces are fetched together
    unsigned r[RFS_V];
#pragma unroll
    for (int u = 0; u < RFS_V; ++u) r[u] = fr[min(f0 + u * NT, p.F - 1)];
#pragma unroll
    for (int u = 0; u < RFS_V; ++u) place(r[u], f0 + u * NT, f0 + u * NT < p.F);
  }
  __syncthreads();             // rs / maxh of this body are read by the tile classes below (same workgroup: L2-coherent stores + barrier)
  P_MARK(4);
}

// The deferred sorts of a launch (RasterP::resort), carried by the FIRST workgroups of its gradient kernel before they take
// their units: the flagged bodies are ranked (every carrier scans the B flags itself: 3 KB, no counter to reset) and carrier w
// sorts the bodies of rank w, w + carriers, ...  At the head of the kernel a sort costs its 25 us of one CU among 256; at the
// tail (round 6, first version: workgroup w looked at body w AFTER its units) it cost the kernel those 25 us whenever one of
// the last workgroups had one to do -- what the sort had cost on the chain.  At most R_DEFER_MAX bodies per launch, the
// flagged ones of lowest index (~6 us of the kernel): a body that is not served stays flagged and is served by a later
// launch, or sorted by the preparation once it has used up its whole margin -- where a crowd (RMSprop's first, largest
// steps: a third of the bodies every cycle) sorts all at once, in parallel, for the price of one.
#define R_DEFER_WGS 256
#define R_DEFER_MAX 64
template <int NT>
__device__ __forceinline__ void r_deferred_sorts(const RasterP& p, int* lds /* >= r_prepare_lds(p) bytes */) {
  if (p.resort == nullptr) return;                     // (host: the launch's dynamic LDS does not hold the sort's tables)
  const int carriers = min((int)gridDim.x, R_DEFER_WGS);
  if ((int)blockIdx.x >= carriers) return;
  __shared__ int sd_cnt[NT / 64], sd_pick[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int total = 0;
  for (int base = 0; base < p.B; base += NT) {
    const int b = base + tid;
    const bool f = b < p.B && p.resort[b] != 0;
    const unsigned long long m = __ballot(f);
    if (lane == 0) sd_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = total, all = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { off += w < wave ? sd_cnt[w] : 0; all += sd_cnt[w]; }
    if (f) {
      const int rank = off + __popcll(m & ((1ull << lane) - 1ull));
      if (rank % carriers == (int)blockIdx.x && rank / carriers < 8) sd_pick[rank / carriers] = b;
    }
    total += all;
    __syncthreads();
  }
  total = min(total, R_DEFER_MAX);
  unsigned pacc[8];
  unsigned long long plast = 0ull;
  for (int k = 0; k < 8 && (int)blockIdx.x + k * carriers < total; ++k) {
    const int b = sd_pick[k];
    r_face_sort<NT>(p, b, lds, p.win[b * 4 + 2], p.win[b * 4 + 3], pacc, plast);      // (ends with a barrier)
    if (tid == 0) p.stale[b] = 2;
  }
  __syncthreads();
}

// One workgroup per body: NDC projection of the vertices (kept in HBM, 12 B per vertex) + screen window, how far the
// vertices have moved since the body's face lists were sorted, the sort itself when they moved too far (temporal
// coherence: the optimiser moves a body by a small fraction of a pixel per cycle; the lists stay a SUPERSET of every
// tile's candidates while no vertex has moved `margin` rows, tiles read `margin` rows further out and k_raster_strip
// decides every face from the current coordinates, so the keys are bit-identical to those of a fresh sort), and the
// body's tiles and gradient work units appended to the cost-class lists.
#ifndef RPREP
#define RPREP 512
#endif
#ifndef RPV
#define RPV 7                // vertices per thread whose loads are in flight together (6890 = 2 x 7 x 512 - 278)
#endif
__global__ __launch_bounds__(RPREP) void k_raster_prepare(RasterP p) {
  extern __shared__ int hist[];                     // [4][H + 1] of the sort + the winners' bitmap (r_prepare_lds)
  unsigned pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long plast = __builtin_readcyclecounter();
  const unsigned long long pbegin = plast;
  __shared__ float sbb[RPREP / 64][4];
  __shared__ int s_win[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int pww = p.win[b * 4 + 2], pwh = p.win[b * 4 + 3];      // the window of the previous launch (its keys: the winners)
  // extremes in NDC; the (monotonically decreasing) NDC -> pixel map is applied once to the four results
  float mnx = 1e30f, mny = 1e30f, mxx = -1e30f, mxy = -1e30f;
  float ra, rk;
  r_row_affine(p, &ra, &rk);
  const float* rowb = p.rowb + (size_t)b * p.V;
  const bool tagged = p.margin > 0 && p.sort_tag[b] == RS_TAG(b, p.margin);
  bool moved = !tagged;
  // lists that were sorted before the body had any keys carry no winners' list: one more sort now that it has
  if (tagged && p.winners_on && p.F <= R_WIN_MAXF && p.wstate[b] == 0 && p.kvalid[b] != 0) moved = true;
  const float thr = (float)p.margin - 0.02f;
  float* nbo = p.ndc + (size_t)b * p.V * 3;
  // projected: mh_lbs_forward_proj has written the NDC vertices, the motion flag and -- unless nobody reported one of the
  // four extremes -- the box: nothing to read per vertex.  An incomplete box is scanned from the projected vertices and
  // written back, so that the next forward has a complete previous box to filter with.
  bool scan = true, have_box = false, soft = false;
  int4 fb = {0, 0, 0, 0};
  if (p.projected) {
    fb = *(const int4*)(p.fbbox + (size_t)b * 4);
    have_box = fb.x != 0x7fffffff && fb.y != 0x7fffffff && fb.z != (int)0x80000000 && fb.w != (int)0x80000000;
    scan = !have_box;
    moved = moved || p.fmoved[b] != 0;
    soft = p.fmoved[p.B + b] != 0;
  }
  if (scan) {
  const float* vb = (p.projected ? p.ndc : p.verts) + (size_t)b * p.V * 3;
  // RPV vertices per thread and trip with all their loads in flight together: one load -> wait -> divide -> store per
  // iteration was a chain of 2 x 14 memory latencies per thread (the loop bound is a kernel argument, the compiler does
  // not batch the loads by itself)
  for (int v0 = tid; v0 < p.V; v0 += RPV * RPREP) {
    float X[RPV], Y[RPV], Z[RPV], rb[RPV];
#pragma unroll
    for (int u = 0; u < RPV; ++u) {
      const unsigned o = __umul24((unsigned)min(v0 + u * RPREP, p.V - 1), 3u);
      X[u] = vb[o]; Y[u] = vb[o + 1u]; Z[u] = vb[o + 2u];
    }
    if (tagged && !p.projected) {
#pragma unroll
      for (int u = 0; u < RPV; ++u) rb[u] = rowb[min(v0 + u * RPREP, p.V - 1)];
    }
#pragma unroll
    for (int u = 0; u < RPV; ++u) {
      const int v = v0 + u * RPREP;
      if (v < p.V) {
        float xn = X[u], yn = Y[u];
        if (!p.projected) {
          xn = p.s * (-X[u]) / Z[u] + p.w1; yn = p.s * (-Y[u]) / Z[u] + p.h1;
          float* o = nbo + __umul24((unsigned)v, 3u);
          o[0] = xn; o[1] = yn; o[2] = Z[u];
          if (tagged) moved = moved || !(fabsf(fmaf(-yn, rk, ra) - rb[u]) < thr);      // NaN-safe: anything odd rebuilds
        }
        if (Z[u] > R_KEPS) {
          mnx = fminf(mnx, xn); mxx = fmaxf(mxx, xn);
          mny = fminf(mny, yn); mxy = fmaxf(mxy, yn);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
    mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
  }
  }   // scan
  const int any_moved = __syncthreads_or(moved ? 1 : 0);
  if (scan && (tid & 63) == 0) {
    sbb[tid >> 6][0] = mnx; sbb[tid >> 6][1] = mny; sbb[tid >> 6][2] = mxx; sbb[tid >> 6][3] = mxy;
  }
  __syncthreads();
  if (tid == 0) {
    if (scan) {
      for (int w = 1; w < RPREP / 64; ++w) {
        mnx = fminf(mnx, sbb[w][0]); mny = fminf(mny, sbb[w][1]);
        mxx = fmaxf(mxx, sbb[w][2]); mxy = fmaxf(mxy, sbb[w][3]);
      }
      if (p.projected && mnx <= mxx) {
        int* o = p.fbbox + (size_t)b * 4;
        o[0] = r_ord(mnx); o[1] = r_ord(mny); o[2] = r_ord(mxx); o[3] = r_ord(mxy);
      }
    } else {
      mnx = r_unord(fb.x); mny = r_unord(fb.y); mxx = r_unord(fb.z); mxy = r_unord(fb.w);
    }
    if (mnx <= mxx) {       // at least one vertex in front of the camera
      const float px0 = r_ndc_to_pix(mxx, p.W, p.H), px1 = r_ndc_to_pix(mnx, p.W, p.H);
      const float py0 = r_ndc_to_pix(mxy, p.H, p.W), py1 = r_ndc_to_pix(mny, p.H, p.W);
      mnx = px0; mxx = px1; mny = py0; mxy = py1;
    }
    // clamp in float first: a body far outside the image must not overflow the int conversion
    const float big = 1e6f;
    mnx = fminf(fmaxf(mnx, -big), big); mxx = fminf(fmaxf(mxx, -big), big);
    mny = fminf(fmaxf(mny, -big), big); mxy = fminf(fmaxf(mxy, -big), big);
    const int x0 = max(0, (int)floorf(mnx) - 2), y0 = max(0, (int)floorf(mny) - 2);
    const int x1 = min(p.W - 1, (int)ceilf(mxx) + 2), y1 = min(p.H - 1, (int)ceilf(mxy) + 2);
    int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
    p.win[b * 4] = x0;
    p.win[b * 4 + 1] = y0;
    p.win[b * 4 + 2] = ww;
    p.win[b * 4 + 3] = wh;
    if (ww <= 0 || wh <= 0) ww = wh = 0;
    s_win[0] = x0; s_win[1] = y0; s_win[2] = ww; s_win[3] = wh;
  }
  P_MARK(0);
  if (p.margin == 0 || any_moved) r_face_sort<RPREP>(p, b, hist, pww, pwh, pacc, plast);       // (ends with a barrier)
  else __syncthreads();
  // ---- tiles of the window: geometry and cost class into the body's own slots -------------------------------------------
  const int x0 = s_win[0], y0 = s_win[1], ww = s_win[2], wh = s_win[3];
  int tw = 1, th = 1, ncol = 0, nrow = 0;
  if (ww > 0) r_tiling(ww, wh, &tw, &th, &ncol, &nrow);
  const int cap = r_cap(p), ns = min(ncol * nrow, cap), first = b * cap;
  if (tid == 0) {
    p.body_first[b] = first; p.body_ns[b] = ns; p.stale[b] = any_moved;
    p.resort[b] = (soft && !any_moved && p.margin > 0) ? 1 : 0;      // (r_deferred_sorts, beside this launch's gradient kernel)
    // a body's keys live in a region of its own (the key array has room for a full image per body): where they are does not
    // depend on the other bodies' windows, so the work lists -- which may be a launch old -- carry no addresses
    p.body_koff[b] = (long long)b * p.H * p.W;
  }
  const int* rs = p.row_start + (size_t)b * (4 * (p.H + 1) + 1);
  const int mh = p.maxh[b];
  for (int k = tid; k < ns; k += RPREP) {
    const int tr = k / ncol, tc = k - tr * ncol, s = first + k;
    const int r0 = y0 + tr * th, nr = min(th, wh - tr * th), c0 = x0 + tc * tw, nc = min(tw, ww - tc * tw);
    p.strip_body[s] = b;
    p.strip_row0[s] = r0; p.strip_rows[s] = nr;
    p.strip_col0[s] = c0; p.strip_cols[s] = nc;
    p.strip_cls[s] = r_tile_class(p, rs, mh, r0, nr, nc);
  }
  P_TIMING_FLUSH();
}

__global__ __launch_bounds__(RLISTS) void k_raster_lists(RasterP p) { r_finalize_lists<RLISTS>(p); }

// depth-term sums of one tile straight from its LDS key window (optimizer.py:432-442): only the nearest key of a pixel is
// needed -- no face gathers -- so this rides in the epilogue of k_raster_strip; the silhouette sum needs alpha and is taken
// where alpha is evaluated anyway (k_raster_grads; k_raster_sums when no gradients are requested)
__device__ __forceinline__ void r_tile_depth_sums(const RasterP& p, int s, int b, const unsigned long long* keys, int tw, int x0, int sy0,
                                                  int npx, float* sh) {
  const int tid = threadIdx.x;
  const int W = p.W, P = p.H * W;
  const int t = b / p.N, n = b % p.N;
  const float min_z = logf(1.f + expf(p.zmin_lin[t]));                    // optimizer.py:683-688
  const float max_z = min_z + 1.f + logf(1.f + expf(p.zmax_lin[t]));
  const float inv_min = 1.f / min_z, inv_max = 1.f / max_z, dspan = inv_min - inv_max;
  const float pvalid = p.p2d_valid[b];
  float lA = 0.f, lB = 0.f, lC = 0.f, lS1 = 0.f, lS2 = 0.f;
  for (int i = tid; i < npx; i += RB) {
    const unsigned long long k0 = keys[(size_t)i * 5];
    if (k0 == RS_EMPTY) continue;
    const int yi = sy0 + i / tw, xi = x0 + i % tw;
    const size_t gp = (size_t)t * P + (size_t)yi * W + xi;
    const float z = __uint_as_float((unsigned)(k0 >> 32));
    const float m = (z > 0.f ? 1.f : 0.f) * (float)((p.ebits[gp] >> n) & 1u) * pvalid;     // :432-438
    if (m != 0.f) {
      const float pred = 1.f / fmaxf(z + 0.2f, p.eps);                                    // :440
      const float dh = p.depths[gp];
      const float tg = dh * dspan + inv_max;                                              // :425
      lA += logf(fmaxf(pred, 1e-3f));
      lB += logf(fmaxf(tg, 1e-3f));
      lC += 1.f;
      if (tg >= 1e-3f) {
        lS1 += dh / tg;
        lS2 += (1.f - dh) / tg;
      }
    }
  }
  // the five sums behind ONE pair of barriers (same order of additions as five r_block_sum calls: lanes by xor
  // butterflies, then the waves in turn)
  lA = mh_wave_sum(lA); lB = mh_wave_sum(lB); lC = mh_wave_sum(lC); lS1 = mh_wave_sum(lS1); lS2 = mh_wave_sum(lS2);
  __syncthreads();
  if ((tid & 63) == 0) {
    float* w = sh + (tid >> 6) * 5;
    w[0] = lA; w[1] = lB; w[2] = lC; w[3] = lS1; w[4] = lS2;
  }
  __syncthreads();
  if (tid < 5) {
    float a = 0.f;
    for (int w = 0; w < RB / 64; ++w) a += sh[w * 5 + tid];
    p.partial[(size_t)s * 6 + tid] = a;
  }
  if (tid == 5) p.partial[(size_t)s * 6 + 5] = 0.f;
}

#define RW (RB / 64)         // waves per tile workgroup
#ifndef R_KEYS_NT
#define R_KEYS_NT 1
#endif
#ifdef R_SYNC_HARD
#define R_WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define R_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif
#ifndef RPL
#define RPL 384              // pair descriptors per wave and round (pair-list path; see R_CAP for the sweep)
#endif

// One workgroup per tile.  Every wave runs its own rounds of 64 candidate faces with no workgroup barrier in
// between: gather (software-pipelined three rounds deep: sort entry -> vertex ids -> projected vertices), blurred
// bbox against the tile, candidate counts, wave prefix sum, then all (face, pixel-centre) pairs of the round are
// split evenly over the 64 lanes (every lane walks a contiguous run of pairs, the staged face stays in registers
// while the run stays inside one face).  The run start -> face lookup is a scatter + prefix-max instead of a
// search.  Only the key window is shared by the waves (LDS atomics).
__global__ __launch_bounds__(RB, RMINW) void k_raster_strip(RasterP p) {       // 4 waves/SIMD: two workgroups per CU
  __shared__ unsigned long long keys[R_CAP * 5];
  __shared__ float wT[RW][64 * RT];         // staged faces of the wave's current round
  __shared__ int wDesc[RW][64];             // xa | ya << 10 | nx << 20 (tile-relative)
  __shared__ int wFid[RW][64];
  // pair list of the pair-list path (face slot | pair index << 6) and, in the same bytes, the even-split path's prefix
  // and run-start tables (a round that takes both paths takes them one after the other); the tile's pixel-centre
  // coordinates share one array (columns + rows <= pixels + 1): LDS the side branch's kernels can have
  static_assert(RPL * 2 >= (65 + 64) * 4, "the pair list must cover the even-split path's tables");
  __shared__ __attribute__((aligned(8))) unsigned short wPl[RW][RPL];
  __shared__ float s_sums[RB / 64 * 5];
  __shared__ float sXY[R_CAP + 2];          // NDC x of the tile's columns, then NDC y of its rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, W = p.W;
  const float blur_d = sqrtf(BLUR_D);
  const int total = p.total[0];
  const float rx = W > H ? 2.f * (float)W / (float)H : 2.f, ry = H > W ? 2.f * (float)H / (float)W : 2.f;
  const float kx = (float)W / rx, ky = (float)H / ry;      // pixels per NDC unit (approximate index only)
  // wave-private staging.  __builtin_amdgcn_wave_barrier() is the only ordering needed: it keeps the compiler from
  // moving LDS accesses across it, and LDS executes one wave's accesses in order (a fence would also wait for the
  // prefetched global loads and serialise the gather pipeline)
  float* T_ = wT[wave];
  int* desc = wDesc[wave];
  int* fid = wFid[wave];
  unsigned short* pl = wPl[wave];
  int* pre = (int*)wPl[wave];               // [65]
  int* mark = pre + 65;                     // [64]
  unsigned long long n_cand = 0ull, n_eval = 0ull;        // wave-uniform
  R_NEAR_VARS();
  R_TIMING_DECL();
  // Work items: the listed tiles first (most expensive class first), then, per body, the tiles its window has gained since
  // the lists were put together -- the lists may be a launch old (mh_raster_fin): a listed tile that no longer exists is
  // skipped, a new one is picked up by the workgroup that looks at its body (none in a steady sequence).
  const int cap_ = r_cap(p);
  int xb = -1, xk = 0, xend = 0;               // body whose new tiles this workgroup is working through
  for (int si = blockIdx.x;;) {
    int s;
    if (xk < xend) {
      s = xb * cap_ + xk++;
    } else if (si < total) {
      s = p.strip_order[si];
      si += gridDim.x;
      const int b_ = s / cap_;
      if (s - b_ * cap_ >= p.body_ns[b_]) continue;      // (the same for every thread of the workgroup)
    } else if (si < total + p.B) {
      xb = si - total;
      si += gridDim.x;
      xk = min(p.ns_listed[xb], p.body_ns[xb]); xend = p.body_ns[xb];
      continue;
    } else {
      break;
    }
    const int b = s / cap_;
    const int x0 = p.strip_col0[s], tw = p.strip_cols[s], x1 = x0 + tw - 1;
    const int sy0 = p.strip_row0[s], nrows = p.strip_rows[s], sy1 = sy0 + nrows - 1;
    const int npx = nrows * tw;
    float* const sXf = sXY;
    float* const sYf = sXY + tw;
    const float* nb = p.ndc + (size_t)b * p.V * 3;
    const unsigned* fs = p.fsort + (size_t)b * p.F;
    const int* rs = p.row_start + (size_t)b * (4 * (H + 1) + 1);
    __syncthreads();
    for (int i = tid; i < npx * 5; i += RB) keys[i] = RS_EMPTY;
    for (int i = tid; i < tw; i += RB) sXf[i] = r_pix_to_ndc(W - 1 - (x0 + i), W, H);
    for (int i = tid; i < nrows; i += RB) sYf[i] = r_pix_to_ndc(H - 1 - (sy0 + i), H, W);
    // candidate faces: the near class of the rows first, then the far class (two contiguous ranges of fsort)
    // three contiguous ranges of fsort: near short faces, far short faces, tall faces.  Kept lists: a face's first row may
    // have moved by up to `margin` rows either way since the sort
    const int ra_ = max(0, sy0 - R_SHORT - p.margin), rt_ = max(0, sy0 - min(max(p.maxh[b], 0), H) - p.margin), rb_ = min(sy1 + 1 + p.margin, H);
    // four contiguous ranges of fsort: last launch's winners (see r_face_sort), near short faces, far short faces, tall faces
    const int w0 = rs[ra_], nw = rs[rb_] - w0;
    const int a0 = rs[H + 1 + ra_], nwa = nw + rs[H + 1 + rb_] - a0, b0 = rs[2 * (H + 1) + ra_], nab = nwa + rs[2 * (H + 1) + rb_] - b0;
    const int c0 = rs[3 * (H + 1) + rt_], nc = rs[3 * (H + 1) + rb_] - c0;
    const int i1 = nab + nc;
    auto fs_at = [&](int j) { return fs[j < nwa ? (j < nw ? w0 + j : a0 + (j - nw)) : (j < nab ? b0 + (j - nwa) : c0 + (j - nab))]; };
    __syncthreads();
    R_TMARK(0);
    if (i1 > 0) {
      const int last = i1 - 1, stride = RW * 64;
      int idx = wave * 64 + lane;
      // pipeline registers: entry of round r+2, vertex ids of round r+1, coordinates of round r
      unsigned e_a = fs_at(min(idx, last)), e_b = fs_at(min(idx + stride, last)), e_c = fs_at(min(idx + 2 * stride, last));
      int va[3], vb_[3];
      float ca[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) va[k] = p.faces[3 * (int)(e_a & 0xfffffu) + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) vb_[k] = p.faces[3 * (int)(e_b & 0xfffffu) + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const unsigned o = __umul24((unsigned)va[k], 3u);        // 32-bit offsets: SGPR base + VGPR offset loads
        ca[3 * k] = nb[o]; ca[3 * k + 1] = nb[o + 1u]; ca[3 * k + 2] = nb[o + 2u];
      }
      for (; idx - lane < i1; idx += stride) {
        // ---- issue the next rounds' gathers before working on this one ------------------------------------
        const unsigned e_n = fs_at(min(idx + 3 * stride, last));
        int vn[3];
        float cn[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) vn[k] = p.faces[3 * (int)(e_c & 0xfffffu) + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const unsigned o = __umul24((unsigned)vb_[k], 3u);
          cn[3 * k] = nb[o]; cn[3 * k + 1] = nb[o + 1u]; cn[3 * k + 2] = nb[o + 2u];
        }
        // ---- this round: bbox against the tile, candidate count, staging -----------------------------------
        int cnt = 0;
        int f_pix = 0, f_nx = 1, f_ny = 1;       // first window pixel (tile-relative index), width and height of the face's pixel box
        unsigned f_zb = 0u;                      // bits of its nearest vertex depth (minus the margin below)
        if (idx < i1 && (int)(e_a >> 20) + p.margin >= sy0) {
          const float bxmin = fminf(ca[0], fminf(ca[3], ca[6])) - blur_d, bxmax = fmaxf(ca[0], fmaxf(ca[3], ca[6])) + blur_d;
          const float bymin = fminf(ca[1], fminf(ca[4], ca[7])) - blur_d, bymax = fmaxf(ca[1], fmaxf(ca[4], ca[7])) + blur_d;
          // pixel range of the blurred bbox: the continuous pixel coordinate of each bound, widened by 1e-3 px (its
          // rounding error is ~1e-5 px), can only be one pixel too generous; one comparison per bound against the
          // tabulated pixel-centre NDC values makes it exact (no refinement loops)
          int xa = max(x0, (int)ceilf((float)W - 0.5f - (bxmax + 0.5f * rx) * kx - 1e-3f));
          int xb = min(x1, (int)floorf((float)W - 0.5f - (bxmin + 0.5f * rx) * kx + 1e-3f));
          int ya = max(sy0, (int)ceilf((float)H - 0.5f - (bymax + 0.5f * ry) * ky - 1e-3f));
          int yb = min(sy1, (int)floorf((float)H - 0.5f - (bymin + 0.5f * ry) * ky + 1e-3f));
          if (xa <= xb && ya <= yb) {
            xa += sXf[xa - x0] > bxmax ? 1 : 0;
            xb -= sXf[xb - x0] < bxmin ? 1 : 0;
            ya += sYf[ya - sy0] > bymax ? 1 : 0;
            yb -= sYf[yb - sy0] < bymin ? 1 : 0;
          }
          cnt = (int)__umul24((unsigned)max(0, xb - xa + 1), (unsigned)max(0, yb - ya + 1));
          const float farea = r_edge(ca[6], ca[7], ca[0], ca[1], ca[3], ca[4]);
          {
            // the rasteriser's exclusions (RasterizeMeshesNaive: zmin < kEpsilon, |face area| <= kEpsilon with
            // area = edge(v0; v1, v2)), from the current coordinates, unfused like the CPU reference computes them
            const float fa = r_edge_exact(ca[0], ca[1], ca[3], ca[4], ca[6], ca[7]);
            if (!(fminf(ca[2], fminf(ca[5], ca[8])) >= R_KEPS) || (fa <= R_KEPS && fa >= -R_KEPS)) cnt = 0;
          }
          if (cnt > 0) {
            float* T = T_ + lane * RT;
            const float ia_ = __builtin_amdgcn_rcpf(farea + R_KEPS);
            const float l01 = (ca[3] - ca[0]) * (ca[3] - ca[0]) + (ca[4] - ca[1]) * (ca[4] - ca[1]);
            const float l02 = (ca[6] - ca[0]) * (ca[6] - ca[0]) + (ca[7] - ca[1]) * (ca[7] - ca[1]);
            const float l12 = (ca[6] - ca[3]) * (ca[6] - ca[3]) + (ca[7] - ca[4]) * (ca[7] - ca[4]);
            const float r01_ = l01 <= R_KEPS ? 0.f : __builtin_amdgcn_rcpf(l01);
            const float r02_ = l02 <= R_KEPS ? 0.f : __builtin_amdgcn_rcpf(l02);
            const float r12_ = l12 <= R_KEPS ? 0.f : __builtin_amdgcn_rcpf(l12);
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = ca[k];
            T[9] = ia_;
            T[10] = r01_;
            T[11] = r02_;
            T[12] = r12_;
            desc[lane] = (xa - x0) | ((ya - sy0) << 10) | ((xb - xa + 1) << 20);
            f_pix = (int)__umul24((unsigned)(ya - sy0), (unsigned)tw) + (xa - x0);
            f_nx = xb - xa + 1;
            f_ny = yb - ya + 1;
            fid[lane] = (int)(e_a & 0xfffffu);
            {
              // nearest vertex depth MINUS 16 ulps: the interpolated depth (normalised weights through v_rcp_f32) can fall a
              // few ulps short of the nearest vertex, and on such a tie the depth cull below dropped a candidate or not
              // depending on which wave got to the pixel first (round 1: ~20 of 1.2 M pixels differed from run to run in
              // the face id of their 4th silhouette key).  With the margin the selection is order-independent.
              const unsigned zq = __float_as_uint(fminf(ca[2], fminf(ca[5], ca[8])));
              f_zb = zq > 16u ? zq - 16u : 0u;
            }
          }
        }
        // ---- small boxes (at most 4 columns x 8 rows: nearly every face at this resolution) go through the depth cull and
        // the compacted pair list; the others -- and everything when the list could overflow: the 3 x 3 boxes of the bodies
        // nearest to the camera -- through the even split below, whose cost per pair is lower when a face has many
        // (staged face in registers, no pair list).  A round may take both paths.  (Filling the list in several passes
        // instead of falling back was measured in round 4: 3 % fewer pairs evaluated, kernel 3 % slower.)
        int cnt_s = (cnt > 0 && f_nx <= 4 && f_ny <= 8 && !p.all_even) ? cnt : 0;
        int npairs_s = __builtin_amdgcn_readlane(r_wave_scan_add(cnt_s), 63);
        // (Round 5, measured: deciding whether a round fits the list on the SURVIVORS of the cull instead of on its candidates --
        // the walk for every round, the even split only when more than RPL pairs survive -- evaluates 6 % fewer pairs and is
        // 4 us slower: the rounds that overflow are mostly the winners' own, which meet empty lists and lose the walk.)
        // (Round 5, measured with the winners first, same box, keys bit-identical: (i) deciding whether a round fits the list
        // on the SURVIVORS of the cull instead of on its candidates -- 6 % fewer pairs evaluated, +4 us: the rounds that
        // overflow are mostly the winners' own, which meet empty lists and lose the walk; (ii) pairs outside the NARROW band
        // -- two thirds of a box -- culled against the pixel's nearest key instead of its 4th: 31.4 -> 26.9 M pairs, +-0 us;
        // (iii) the per-pair depth cull on the even-split path again: +4 us.  Fewer evaluated pairs no longer buy time.)
        if (npairs_s > RPL) { cnt_s = 0; npairs_s = 0; }
        R_TMARK(1);
        unsigned keepm = 0u;
        int nk = 0, kincl = 0, nkeep = 0;
        if (npairs_s > 0) {
          if (cnt_s > 0) {
            const char* kb = (const char*)keys + __umul24((unsigned)f_pix, 40u);
            const unsigned rowstep = __umul24((unsigned)tw, 40u);
            const unsigned colm = (1u << f_nx) - 1u;
            for (int r4 = 0; r4 < 4 * f_ny; r4 += 4) {
              const unsigned* qh = (const unsigned*)kb;
              const unsigned q0 = qh[9], q1 = qh[19], q2 = qh[29], q3 = qh[39];
              const unsigned bits = (f_zb <= q0 ? 1u : 0u) | (f_zb <= q1 ? 2u : 0u) | (f_zb <= q2 ? 4u : 0u) | (f_zb <= q3 ? 8u : 0u);
              keepm |= (bits & colm) << r4;
              kb += rowstep;
            }
          }
          nk = __popc(keepm);
          kincl = r_wave_scan_add(nk);
          nkeep = __builtin_amdgcn_readlane(kincl, 63);
        }
        const int cnt_g = cnt - cnt_s;
        if (npairs_s > 0) {
          // Depth cull first, per face lane: the clipped-barycentric depth of a face is never below its nearest vertex, so
          // a pair whose face lies entirely behind the pixel's current 4th silhouette key cannot change the window (the keys
          // only ever decrease; the nearest key of the wide pass never lies behind it: every key of the K=4 list was
          // offered to slot 0 first -- r_insert -- and a face inside the narrow band is inside the wide one).  A lane walks
          // the rows of its own box, four pixels per trip from ONE address (columns past the box read other LDS words and
          // are masked out), and keeps a bit per survivor at position 4 * row + column; only the survivors are written to
          // the pair list, behind a wave prefix sum of the counts, and evaluated with full lanes.
          // (Round 1 listed every pair, then culled the list 64 pairs at a time with a decode per pair; rounds 2-3 walked the
          // box in pixel order, four pixels per trip with a wrap test per pixel and a division by the box width per pair.)
#ifdef R_COUNT_PATHS          // counter A = pairs of the even-split path, counter B = survivors of the pair-list path
          n_eval += (unsigned)nkeep;
#else
          n_cand += (unsigned)npairs_s; n_eval += (unsigned)nkeep;
#endif
          {
            int pos = kincl - nk;
            for (unsigned m = keepm; m; m &= m - 1u) pl[pos++] = (unsigned short)(lane | ((__ffs((int)m) - 1) << 6));
          }
          R_WAVE_SYNC();
          R_TMARK(2);
          for (int i = lane; i < nkeep; i += 64) {
            const unsigned e = pl[i], lo = e & 63u, k = e >> 6;
            const unsigned d = (unsigned)desc[lo], fw = (unsigned)fid[lo];
            const unsigned xi = (d & 1023u) + (k & 3u), yi = ((d >> 10) & 1023u) + (k >> 2);
            float T[RT];
#pragma unroll
            for (int q = 0; q < RT; ++q) T[q] = T_[lo * RT + q];
            float pz, dd;
            bool inside;
            R_NEAR_DECL();
            r_eval_fast(T, sXf[xi], sYf[yi], &pz, &inside, &dd R_NEAR_ARG);
            R_NEAR_COUNT((const unsigned long long*)((const char*)keys + __umul24(__umul24(yi, (unsigned)tw) + xi, 40u)), pz, inside, dd, (int)fw);
            r_insert((unsigned long long*)((char*)keys + __umul24(__umul24(yi, (unsigned)tw) + xi, 40u)), pz, inside, dd, (int)fw);
          }
          R_WAVE_SYNC();
          R_TMARK(3);
        }
        if (__ballot(cnt_g > 0) != 0ull) {
          const int incl = r_wave_scan_add(cnt_g);
          const int npairs = __builtin_amdgcn_readlane(incl, 63);
          const int excl = incl - cnt_g;
          // ---- larger faces: the pairs are split evenly over the lanes, every lane walks a contiguous run ----------
#ifdef R_COUNT_PATHS
          n_cand += (unsigned)npairs;
#else
          n_cand += (unsigned)npairs; n_eval += (unsigned)npairs;
#endif
          pre[lane] = excl;
          if (lane == 63) pre[64] = npairs;
          mark[lane] = -1;
          R_WAVE_SYNC();
          // run start -> face: face o opens at the first lane whose run starts at or after pre[o]
          const int per = (npairs + 63) >> 6;
          if (cnt_g > 0) {
            const int tf = (excl + per - 1) / per;
            if (tf < 64) atomicMax(&mark[tf], lane);
          }
          R_WAVE_SYNC();
          int lo = r_wave_scan_max(mark[lane]);
          const int j0 = lane * per, j1 = min(j0 + per, npairs);
          if (j0 < j1) {
            float T[RT];
#pragma unroll
            for (int q = 0; q < RT; ++q) T[q] = T_[lo * RT + q];
            int d = desc[lo], f = fid[lo];
            int nx = d >> 20, xa = d & 1023, ya = (d >> 10) & 1023;
            const int k = j0 - pre[lo];
            int ky_ = k / nx, kx_ = k - ky_ * nx;
            int nextp = pre[lo + 1];
            for (int j = j0;;) {
              const int xi = xa + kx_, yi = ya + ky_;
              float pz, dd;
              bool inside;
              R_NEAR_DECL();
              r_eval_fast(T, sXf[xi], sYf[yi], &pz, &inside, &dd R_NEAR_ARG);
              R_NEAR_COUNT((const unsigned long long*)((const char*)keys + __umul24(__umul24((unsigned)yi, (unsigned)tw) + (unsigned)xi, 40u)), pz, inside, dd, f);
              r_insert((unsigned long long*)((char*)keys + __umul24(__umul24((unsigned)yi, (unsigned)tw) + (unsigned)xi, 40u)), pz, inside, dd, f);
              if (++j >= j1) break;
              if (++kx_ == nx) { kx_ = 0; ++ky_; }
              if (j >= nextp) {                          // the run moves on to the next face with candidates
                ++lo;
                while (pre[lo + 1] <= j) ++lo;
#pragma unroll
                for (int q = 0; q < RT; ++q) T[q] = T_[lo * RT + q];
                d = desc[lo]; f = fid[lo];
                nx = d >> 20; xa = d & 1023; ya = (d >> 10) & 1023;
                kx_ = 0; ky_ = 0;
                nextp = pre[lo + 1];
              }
            }
          }
          R_WAVE_SYNC();
          R_TMARK(4);
        }
        // ---- rotate the pipeline ----------------------------------------------------------------------------------
        e_a = e_b; e_b = e_c; e_c = e_n;
#pragma unroll
        for (int k = 0; k < 3; ++k) vb_[k] = vn[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) ca[k] = cn[k];
      }
    }
    R_TMARK(1);
    __syncthreads();
    R_TMARK(5);
    // finished tile -> HBM (40 B per pixel), window row-major
    const int wx0 = p.win[b * 4], wy0 = p.win[b * 4 + 1], ww = p.win[b * 4 + 2];
    unsigned long long* gk = p.gkeys + (size_t)p.body_koff[b] * 5;
    if (tw == ww && x0 == wx0) {   // a full-width strip of rows (the usual tiling) is one contiguous range of the body's keys
      unsigned long long* dst = gk + (size_t)(sy0 - wy0) * ww * 5;
#if R_KEYS_NT       // written once, read a kernel later: kept out of the L2 sets the other tiles' gathers live in (-6 us same-box)
      for (int i = tid; i < npx * 5; i += RB) __builtin_nontemporal_store(keys[i], dst + i);
#else
      for (int i = tid; i < npx * 5; i += RB) dst[i] = keys[i];
#endif
    } else {
      for (int i = tid; i < npx * 5; i += RB) {
        const int px = i / 5, c = i - px * 5;
        const int r = px / tw, cc = px - r * tw;
        gk[((size_t)(sy0 - wy0 + r) * ww + (x0 - wx0 + cc)) * 5 + c] = keys[i];
      }
    }
    if (tid == 0) p.kvalid[b] = 1;               // the body has keys: its next face sort finds its winners
    r_tile_depth_sums(p, s, b, keys, tw, x0, sy0, npx, s_sums);
    R_TMARK(6);
  }
#ifdef MH_EXPERIMENT_TIMING
  R_TIMING_FLUSH();
#else
  if (p.pairs) {
    // per-workgroup slots, plain adds (49 000 same-address atomics at the end of the kernel doubled its duration)
    __shared__ unsigned long long s_cnt[RW][2];
    __syncthreads();
    if (lane == 0) { s_cnt[wave][0] = n_cand; s_cnt[wave][1] = n_eval; }
    R_NEAR_FLUSH();
    __syncthreads();
    if (tid == 0) {
      unsigned long long a = 0ull, b2 = 0ull;
      for (int w = 0; w < RW; ++w) { a += s_cnt[w][0]; b2 += s_cnt[w][1]; }
      unsigned long long* slot = p.pairs + 2 + 2 * (size_t)blockIdx.x;
      slot[0] += a; slot[1] += b2;
      if (blockIdx.x == 0) p.pairs[0] += 1ull;
    }
  }
#endif
}

// =============================================================================================
// residual sums per strip (optimizer.py:432-442, 447-477)
// =============================================================================================
__device__ __forceinline__ float r_alpha(const RasterP& p, const float* nb, const unsigned long long* q, float xf, float yf) {
  float qprod = 1.f;
  for (int k = 1; k < 5; ++k) {
    const unsigned long long kk = q[k];
    if (kk == RS_EMPTY) break;
    Tri tr;
    r_load_tri_ndc(p, nb, (int)(kk & 0xffffffffu), tr);
    const float T9[9] = {tr.x[0], tr.y[0], tr.z[0], tr.x[1], tr.y[1], tr.z[1], tr.x[2], tr.y[2], tr.z[2]};
    float pz, d;
    bool inside;
    r_eval(T9, xf, yf, &pz, &inside, &d);
    const float sd = inside ? -d : d;
    qprod *= 1.f - 1.f / (1.f + expf(sd / SIGMA_S));          // 1 - sigmoid(-sd/sigma)
  }
  return 1.f - qprod;
}

__global__ __launch_bounds__(RB) void k_raster_sums(RasterP p) {
  __shared__ float sh[RB / 64];
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W, P = H * W;
  const int total = p.total[0];
  for (int si = blockIdx.x; si < total; si += gridDim.x) {
    const int s = p.strip_order[si];
    const int b = p.strip_body[s], t = b / p.N, n = b % p.N;
    const int x0 = p.strip_col0[s], tw = p.strip_cols[s];
    const int sy0 = p.strip_row0[s], npx = p.strip_rows[s] * tw;
    const int wx0 = p.win[b * 4], wy0 = p.win[b * 4 + 1], ww = p.win[b * 4 + 2];
    const float* vb = p.ndc + (size_t)b * p.V * 3;
    const unsigned long long* gk = p.gkeys + (size_t)p.body_koff[b] * 5;
    const float min_z = logf(1.f + expf(p.zmin_lin[t]));                    // optimizer.py:683-688
    const float max_z = min_z + 1.f + logf(1.f + expf(p.zmax_lin[t]));
    const float inv_min = 1.f / min_z, inv_max = 1.f / max_z, dspan = inv_min - inv_max;
    const float pvalid = p.p2d_valid[b];
    const uint32_t fr = p.front[b];
    float lA = 0.f, lB = 0.f, lC = 0.f, lS1 = 0.f, lS2 = 0.f, lCorr = 0.f;
    for (int i = tid; i < npx; i += RB) {
      const int yi = sy0 + i / tw, xi = x0 + i % tw;
      const size_t gp = (size_t)t * P + (size_t)yi * W + xi;
      const unsigned long long* q = gk + ((size_t)(yi - wy0) * ww + (xi - wx0)) * 5;
      const unsigned long long k0 = q[0];
      if (k0 != RS_EMPTY) {
        const float z = __uint_as_float((unsigned)(k0 >> 32));
        const float m = (z > 0.f ? 1.f : 0.f) * (float)((p.ebits[gp] >> n) & 1u) * pvalid;     // :432-438
        if (m != 0.f) {
          const float pred = 1.f / fmaxf(z + 0.2f, p.eps);                                    // :440
          const float dh = p.depths[gp];
          const float tg = dh * dspan + inv_max;                                              // :425
          lA += logf(fmaxf(pred, 1e-3f));
          lB += logf(fmaxf(tg, 1e-3f));
          lC += 1.f;
          if (tg >= 1e-3f) {
            lS1 += dh / tg;
            lS2 += (1.f - dh) / tg;
          }
        }
        if (p.zbuf_out) p.zbuf_out[(size_t)b * P + (size_t)yi * W + xi] = z;
      }
      if (q[1] != RS_EMPTY) {
        const float alpha = r_alpha(p, vb, q, r_pix_to_ndc(W - 1 - xi, W, H), r_pix_to_ndc(H - 1 - yi, H, W));
        const uint32_t wb = p.bits[gp];
        if ((wb & fr) == 0u) {                                                                  // 1 - acc
          const float seg = (float)((wb >> n) & 1u);
          lCorr += alpha * alpha - 2.f * alpha * seg;
        }
        if (p.alpha_out) p.alpha_out[(size_t)b * P + (size_t)yi * W + xi] = alpha;
      }
    }
    lA = r_block_sum(lA, sh); lB = r_block_sum(lB, sh); lC = r_block_sum(lC, sh);
    lS1 = r_block_sum(lS1, sh); lS2 = r_block_sum(lS2, sh); lCorr = r_block_sum(lCorr, sh);
    if (tid == 0) {
      float* o = p.partial + (size_t)s * 6;
      o[0] = lA; o[1] = lB; o[2] = lC; o[3] = lS1; o[4] = lS2; o[5] = lCorr;
    }
  }
}

// per-body totals (fixed order over the body's strips)
__device__ __forceinline__ void r_body_sums(const RasterP& p, int b, float out[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = 0.f;
  const int first = p.body_first[b], ns = p.body_ns[b];
  for (int s = first; s < first + ns; ++s)
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] += p.partial[(size_t)s * 6 + k];
}

// The same totals, the same order of additions, for a whole wave at once (b wave-uniform, every lane active): the strips'
// partial sums come in with ONE load per ten strips (lane = strip x 6 + k) instead of one scalar round trip per strip.
// (Round 6: a gradient unit spends 2.2 us -- 12 % of the kernel by the timing build -- on its header's chain of dependent
// loads; this link of it is worth 0.4 us of the kernel's 92: the chain's other links are the unit list, the body's window
// and the per-body coefficients.)
__device__ __forceinline__ void r_body_sums_wave(const RasterP& p, int b, float out[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = 0.f;
  const int lane = threadIdx.x & 63;
  const int first = p.body_first[b], ns = p.body_ns[b];
  for (int s0 = 0; s0 < ns; s0 += 10) {
    const int m = min(10, ns - s0);
    const float v = lane < m * 6 ? p.partial[(size_t)(first + s0) * 6 + lane] : 0.f;
    for (int s = 0; s < m; ++s)
#pragma unroll
      for (int k = 0; k < 6; ++k) out[k] += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s * 6 + k));
  }
}

// per-body loss values + depth-range partials (also covers bodies that are entirely off screen)
__global__ void k_raster_body_out(RasterP p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  float S[6];
  r_body_sums(p, b, S);
  const float cnt = S[2] + 1.f;
  const float diff = S[0] / cnt - S[1] / cnt;                                                  // losses.py:24-27
  p.depth_body[b] = diff * diff;
  p.sil_body[b] = p.sil_apply[b] * (p.sil_S[b] + S[5]) / (p.sil_D[b] + 1.f);                  // losses.py:35-38
  const float gB = p.coef_depth * (-2.f) * diff / cnt;
  p.dinv[(size_t)b * 2] = gB * S[3];          // d/d(1/min_z) through the target disparity
  p.dinv[(size_t)b * 2 + 1] = gB * S[4];      // d/d(1/max_z)
  p.sil_corr[b] = 0.f;                        // accumulated by k_raster_grads, consumed and cleared by k_raster_finish
}

// Last kernel of the rasterised terms (a single workgroup: a few microseconds of work that used to be three launches --
// k_raster_body_out between selection and gradients, this kernel, and a reduction for the log): the closing job of
// mh_common.h / mh_raster_fin.  In the optimisation cycle the job rides in the LBS backward's pose kernel instead
// (mh_raster_terms_deferred + mh_lbs_backward_kp_fin) and this kernel is not launched.
#define RFIN 1024
__global__ __launch_bounds__(RFIN) void k_raster_finish(mh_raster_fin f) {
  __shared__ float s_g0[MH_FIN_U * RFIN], s_g1[MH_FIN_U * RFIN];
  mh_raster_finish_job<RFIN>(f, s_g0, s_g1);
}

// per-body constants of the gradient kernels
struct RgBody {
  int t, n, x0, sy0, ww;
  const float* vb;                   // projected vertices of the body
  float* gvb;                        // dL/dverts of the body
  const unsigned long long* gk;      // the body's window of selection keys
  float gA, gAlphaScale, pvalid;
  bool sil_on;
  uint32_t fr;
};

// gradient contributions of window pixel i of a body (PyTorch3D's rasteriser backward through the SELECTED faces:
// clip Jacobian, barycentric Jacobian, PointLineDistanceBackward with the clamped parameter held constant) into the
// accumulator MODE selects; lcorr accumulates the silhouette value term
template <int MODE>
__device__ __forceinline__ void rg_pixel(const RasterP& p, const RgBody& bd, const int i, float& lcorr, RgDet& dc) {
  const int H = p.H, W = p.W, P = H * W;
  const int t = bd.t, n = bd.n, x0 = bd.x0, sy0 = bd.sy0, ww = bd.ww;
  const float* vb = bd.vb;
  float* gvb = bd.gvb;
  const unsigned long long* gk = bd.gk;
  const float gA = bd.gA, gAlphaScale = bd.gAlphaScale, pvalid = bd.pvalid;
  const bool sil_on = bd.sil_on;
  const uint32_t fr = bd.fr;
  const int yi = sy0 + i / ww, xi = x0 + i % ww;
  const size_t gp = (size_t)t * P + (size_t)yi * W + xi;
  const float yf = r_pix_to_ndc(H - 1 - yi, H, W), xf = r_pix_to_ndc(W - 1 - xi, W, H);
  const unsigned long long* q = gk + (size_t)i * 5;
  const unsigned long long k0 = q[0];
  if (k0 != RS_EMPTY && gA != 0.f) {
    const float z = __uint_as_float((unsigned)(k0 >> 32));
    const float m = (z > 0.f ? 1.f : 0.f) * (float)((p.ebits[gp] >> n) & 1u) * pvalid;
    const float zc = z + 0.2f;
    if (m != 0.f && zc > p.eps && 1.f / zc > 1e-3f) {
      const float gpz = -gA * r_rcp(zc);
      Tri tr;
      r_load_tri_ndc(p, vb, (int)(k0 & 0xffffffffu), tr);
      const float area = r_edge(tr.x[2], tr.y[2], tr.x[0], tr.y[0], tr.x[1], tr.y[1]) + R_KEPS;
      const float ia = r_rcp(area);
      float w[3] = {r_edge(xf, yf, tr.x[1], tr.y[1], tr.x[2], tr.y[2]) * ia,
                    r_edge(xf, yf, tr.x[2], tr.y[2], tr.x[0], tr.y[0]) * ia,
                    r_edge(xf, yf, tr.x[0], tr.y[0], tr.x[1], tr.y[1]) * ia};
      const float c[3] = {fmaxf(w[0], 0.f), fmaxf(w[1], 0.f), fmaxf(w[2], 0.f)};
      const float craw = c[0] + c[1] + c[2];
      const float cs = fmaxf(craw, 1e-5f);
      const float ics = r_rcp(cs);
      // pz = sum (c_i/cs) z_i.  The normalised weights n_i = c_i / cs are IEEE divisions: outside the triangle one or
      // two of the clipped weights are zero, and with a single survivor n is EXACTLY 1, so that the Jacobian of the
      // normalisation, (g_k - sum_j g_j n_j) / cs, cancels exactly as it does in the reference's autograd.  With the
      // 1-ulp reciprocal the residue g (1 - c rcp(c)) ~ 1e-7 g survived and was then multiplied by 1/area ~ 1e6 of a
      // sub-pixel face: the largest error of the whole gradient (4e-4 of the largest entry, found with the
      // deterministic scatter, round 3).
      float gwc[3], gz[3], nw[3];
      float dotn = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        nw[k] = c[k] / cs;
        gz[k] = gpz * nw[k];
        gwc[k] = gpz * tr.z[k];          // d/d(normalised clipped weight)
        dotn += gwc[k] * nw[k];
      }
      // d pz / d c_k = (z_k - sum_j n_j z_j) / cs.  With sum_j n_j = 1 this is sum_j (z_k - z_j) n_j / cs: the depths of a
      // face's vertices agree to ~1e-3, so subtracting them FIRST (exact in fp32) keeps the digits that
      // gwc[k] - sum_j gwc[j] n_j loses to cancellation (2e-7 / 1e-3 = 2e-4 relative, then times 1/area of a sliver:
      // up to 5e-3 of the largest entry of the whole gradient on 16 entries of one random scene, where the float32
      // autograd of the oracle is itself 1e-3 off its float64 self; tools/fuzz_raster_grads.py, tools/grad_debug.py)
      float gw[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float gc;
        if (craw > 1e-5f) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (j != k) acc += (tr.z[k] - tr.z[j]) * nw[j];
          gc = gpz * acc * ics;
        } else {
          gc = gwc[k] * ics;               // the clamp of the weight sum is active: no normalisation term
        }
        gw[k] = w[k] > 0.f ? gc : 0.f;
      }
      (void)dotn;
      // w_i = e_i / area
      const float ge[3] = {gw[0] * ia, gw[1] * ia, gw[2] * ia};
      const float garea = -(gw[0] * w[0] + gw[1] * w[1] + gw[2] * w[2]) * ia;
      float gx[3] = {0, 0, 0}, gy[3] = {0, 0, 0};
      // e0 = edge(p; v1, v2), e1 = edge(p; v2, v0), e2 = edge(p; v0, v1); area = edge(v2; v0, v1)
#define EDGE_ADJ(gE, A, Bv)                                           \
  gx[A] += (gE) * (yf - tr.y[Bv]);  gy[A] += (gE) * (tr.x[Bv] - xf);  \
  gx[Bv] += (gE) * (-(yf - tr.y[A])); gy[Bv] += (gE) * (xf - tr.x[A]);
      EDGE_ADJ(ge[0], 1, 2)
      EDGE_ADJ(ge[1], 2, 0)
      EDGE_ADJ(ge[2], 0, 1)
#undef EDGE_ADJ
      // area = (x2-x0)(y1-y0) - (y2-y0)(x1-x0)
      gx[2] += garea * (tr.y[1] - tr.y[0]);  gy[2] += garea * (-(tr.x[1] - tr.x[0]));
      gx[0] += garea * (tr.y[2] - tr.y[1]);  gy[0] += garea * (tr.x[1] - tr.x[2]);
      gx[1] += garea * (-(tr.y[2] - tr.y[0])); gy[1] += garea * (tr.x[2] - tr.x[0]);
#pragma unroll
      for (int k = 0; k < 3; ++k) r_scatter<MODE>(p, gvb, tr, k, gx[k], gy[k], gz[k], dc);
    }
  }
  // silhouette: the (up to) four selected faces are fetched together (independent gathers in
  // flight), evaluated, and scattered from registers
  const uint32_t wb = p.bits[gp];
  if (sil_on && (wb & fr) == 0u && q[1] != RS_EMPTY) {
    Tri trs[4];
    bool have[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long kk = q[k + 1];
      have[k] = kk != RS_EMPTY;
      if (have[k]) r_load_tri_ndc(p, vb, (int)(kk & 0xffffffffu), trs[k]);
    }
    float pk[4], sgn[4], gqx[4], gqy[4], wa[4], wb_[4];
    int ea[4], eb[4];
    float qprod = 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pk[k] = 0.f; sgn[k] = 0.f; gqx[k] = gqy[k] = wa[k] = wb_[k] = 0.f; ea[k] = 0; eb[k] = 1;
      if (!have[k]) continue;
      const Tri& tr = trs[k];
      const float area = r_edge(tr.x[2], tr.y[2], tr.x[0], tr.y[0], tr.x[1], tr.y[1]) + R_KEPS;
      const float ia = r_rcp(area);
      const bool inside = r_edge(xf, yf, tr.x[1], tr.y[1], tr.x[2], tr.y[2]) * ia > 0.f &&
                          r_edge(xf, yf, tr.x[2], tr.y[2], tr.x[0], tr.y[0]) * ia > 0.f &&
                          r_edge(xf, yf, tr.x[0], tr.y[0], tr.x[1], tr.y[1]) * ia > 0.f;
      float t01, t02, t12;
      bool g01, g02, g12;
      const float d01 = r_seg_rcp(xf, yf, tr.x[0], tr.y[0], tr.x[1], tr.y[1], &t01, &g01);
      const float d02 = r_seg_rcp(xf, yf, tr.x[0], tr.y[0], tr.x[2], tr.y[2], &t02, &g02);
      const float d12 = r_seg_rcp(xf, yf, tr.x[1], tr.y[1], tr.x[2], tr.y[2], &t12, &g12);
      float d, tt;
      bool dg;
      int a, bb;
      if (d01 <= d02 && d01 <= d12) { d = d01; a = 0; bb = 1; tt = t01; dg = g01; }
      else if (d02 <= d01 && d02 <= d12) { d = d02; a = 0; bb = 2; tt = t02; dg = g02; }
      else { d = d12; a = 1; bb = 2; tt = t12; dg = g12; }
      const float xa_ = a == 0 ? tr.x[0] : tr.x[1], ya_ = a == 0 ? tr.y[0] : tr.y[1];
      const float xb_ = bb == 1 ? tr.x[1] : tr.x[2], yb_ = bb == 1 ? tr.y[1] : tr.y[2];
      if (dg) { gqx[k] = xb_ - xf; gqy[k] = yb_ - yf; wa[k] = 0.f; wb_[k] = 1.f; }
      else {
        gqx[k] = xa_ + tt * (xb_ - xa_) - xf;
        gqy[k] = ya_ + tt * (yb_ - ya_) - yf;
        wa[k] = 1.f - tt; wb_[k] = tt;
      }
      ea[k] = a; eb[k] = bb;
      const float sd = inside ? -d : d;
      pk[k] = r_rcp(1.f + expf(sd * (1.f / SIGMA_S)));
      sgn[k] = inside ? -1.f : 1.f;
      qprod *= 1.f - pk[k];
    }
    const float alpha = 1.f - qprod;
    const float seg = (float)((wb >> n) & 1u);
    lcorr += alpha * alpha - 2.f * alpha * seg;                        // losses.py:35-38 through 1 - acc (value only)
    const float galpha = gAlphaScale * (alpha - seg);
    if (galpha != 0.f) {
      // Every selected face contributes to the two end points of its nearest edge.  The (up to) eight end points
      // of a pixel are gathered in registers (picked with selects, mapped to camera space) and contributions to the
      // same vertex are merged before they go to the LDS table: neighbouring faces share contour vertices, and an
      // LDS float atomic costs ~4 cycles per active LANE on gfx950 whatever the addresses (SQ_LDS_IDX_ACTIVE:
      // 26 M cycles for the 7 M lane-atomics this scatter issued unmerged -- a third of the kernel; unique
      // addresses, fewer instructions under the same masks or de-correlated lanes changed nothing), while the vector
      // instructions of the merge are nearly free here (VALU 23 % busy).  113 -> 90 us.  (Round 6, with the adds as
      // compare-and-swaps -- r_acc_add -- the merge is still worth 1.2 us: -DRG_NOMERGE 93.5 against 92.3.)  (Merging the three
      // vertices of the depth term's face as well costs 45 spilled registers at 1024 threads: 139 us.)
      int eid[8];
      float egx[8], egy[8], egz[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        eid[2 * k] = eid[2 * k + 1] = -1;
        egx[2 * k] = egy[2 * k] = egz[2 * k] = egx[2 * k + 1] = egy[2 * k + 1] = egz[2 * k + 1] = 0.f;
        if (!have[k]) continue;
        // d alpha / d sd_k = -(1/sigma) p_k prod_j (1 - p_j)
        const float gd = galpha * (-(1.f / SIGMA_S)) * pk[k] * qprod * sgn[k];
        if (gd == 0.f) continue;
        const Tri& tr = trs[k];
        const bool a0 = ea[k] == 0, b1 = eb[k] == 1;       // a in {0,1}, b in {1,2}
        const int ia_ = a0 ? tr.idx[0] : tr.idx[1], ib_ = b1 ? tr.idx[1] : tr.idx[2];
        const float xa = a0 ? tr.x[0] : tr.x[1], ya = a0 ? tr.y[0] : tr.y[1], za = a0 ? tr.z[0] : tr.z[1];
        const float xb = b1 ? tr.x[1] : tr.x[2], yb = b1 ? tr.y[1] : tr.y[2], zb = b1 ? tr.z[1] : tr.z[2];
        const float gxn = gd * 2.f * gqx[k], gyn = gd * 2.f * gqy[k];
        if (wa[k] != 0.f) {
          const float rz = r_rcp(za), ux = wa[k] * gxn, uy = wa[k] * gyn;
          eid[2 * k] = ia_;
          egx[2 * k] = -p.s * rz * ux; egy[2 * k] = -p.s * rz * uy; egz[2 * k] = -((xa - p.w1) * ux + (ya - p.h1) * uy) * rz;
        }
        if (wb_[k] != 0.f) {
          const float rz = r_rcp(zb), ux = wb_[k] * gxn, uy = wb_[k] * gyn;
          eid[2 * k + 1] = ib_;
          egx[2 * k + 1] = -p.s * rz * ux; egy[2 * k + 1] = -p.s * rz * uy; egz[2 * k + 1] = -((xb - p.w1) * ux + (yb - p.h1) * uy) * rz;
        }
      }
#ifndef RG_NOMERGE
#pragma unroll
      for (int i = 1; i < 8; ++i) {
        bool merged = false;
#pragma unroll
        for (int j = 0; j < i; ++j) {
          const bool hit = !merged && eid[i] >= 0 && eid[j] == eid[i];
          egx[j] += hit ? egx[i] : 0.f;
          egy[j] += hit ? egy[i] : 0.f;
          egz[j] += hit ? egz[i] : 0.f;
          merged = merged || hit;
        }
        if (merged) eid[i] = -1;
      }
#endif
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (eid[i] >= 0) r_acc_add<MODE>(gvb, eid[i], egx[i], egy[i], egz[i], dc);
    }
  }
}

// =============================================================================================
// gradients per strip
// =============================================================================================
#ifndef RGB
#define RGB 1024             // threads per body workgroup of the gradient kernel (one workgroup per CU: LDS table)
#endif
template <bool TAB>
__global__ __launch_bounds__(RGB) void k_raster_grads(RasterP p) {
  __shared__ float s_red[RGB / 64];
  float* gtab = rg_tab;
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W, P = H * W;
  const bool use_tab = TAB;
  int* plist = (int*)(gtab + (use_tab ? p.V * 3 : 0));
  int* s_n = plist + RG_LIST;
  const int nunits = p.gunit_total[0];
  RG_TIMING_DECL();
  // ---- deferred face sorts (RasterP::resort): the flagged bodies' lists, from this launch's coordinates and the keys the
  // selection has just written, for the NEXT launch.  Nothing in this kernel reads what a sort writes.
  r_deferred_sorts<RGB>(p, (int*)gtab);
  // Work items as in k_raster_strip: the listed units (the lists may be a launch old: a listed piece beyond the body's
  // current window is skipped), then per body the pieces its window has gained since (none in a steady sequence).
  int xb = -1, xk = 0, xend = 0;
  for (int u = blockIdx.x;;) {
    int b, piece;
    if (xk < xend) {
      b = xb; piece = xk++;
    } else if (u < nunits) {
      const unsigned long long ue = p.gunit_list[u];
      u += gridDim.x;
      b = (int)(ue >> 32); piece = (int)(ue & 0xffffffffu);
    } else if (u < nunits + p.B) {
      xb = u - nunits;
      u += gridDim.x;
      const int wx = p.win[xb * 4 + 2], wy = p.win[xb * 4 + 3];
      const int nu = (wx > 0 && wy > 0) ? (wx * wy + RG_UNIT - 1) / RG_UNIT : 0;
      xk = min(p.nu_listed[xb], nu); xend = nu;
      continue;
    } else {
      break;
    }
    const int up0 = piece * RG_UNIT;
    const int t = b / p.N, n = b % p.N;
    const int x0 = p.win[b * 4], ww = p.win[b * 4 + 2], y0 = p.win[b * 4 + 1], wh = p.win[b * 4 + 3];
    if (ww <= 0 || wh <= 0 || up0 >= ww * wh) continue;          // a listed piece the window no longer has
    const int npx = min(ww * wh, up0 + RG_UNIT);
    const int sy0 = y0;
    const float* vb = p.ndc + (size_t)b * p.V * 3;
    float* gvb = p.gverts + (size_t)b * p.V * 3;
    // the strips of a body are consecutive in the work list, so its window is one contiguous key range
    const unsigned long long* gk = p.gkeys + (size_t)p.body_koff[b] * 5;
    float S[6];
#if RG_WAVE_SUMS
    r_body_sums_wave(p, b, S);
#else
    r_body_sums(p, b, S);
#endif
    const float cnt = S[2] + 1.f;
    const float diff = S[0] / cnt - S[1] / cnt;
    const float gA = p.coef_depth * 2.f * diff / cnt;
    const float gAlphaScale = p.coef_sil * p.sil_apply[b] * 2.f / (p.sil_D[b] + 1.f);
    const bool sil_on = p.sil_apply[b] != 0.f;     // alpha is also needed for the loss value (sil_corr), not only for gradients
    float lcorr = 0.f;
    const float pvalid = p.p2d_valid[b];
    const uint32_t fr = p.front[b];
    RgBody bd;
    bd.t = t; bd.n = n; bd.x0 = x0; bd.sy0 = sy0; bd.ww = ww; bd.vb = vb; bd.gvb = gvb; bd.gk = gk;
    bd.gA = gA; bd.gAlphaScale = gAlphaScale; bd.pvalid = pvalid; bd.sil_on = sil_on; bd.fr = fr;
    RgDet dc;
    // most window pixels carry no gradient (outside the blur band, masked out, occluded by a nearer person's mask):
    // classify RG_LIST pixels at a time, compact the live ones into an LDS list and evaluate those with full waves.
    // The classification loads of all of a thread's pixels are issued first, and the table is cleared while they
    // are in flight.
    bool table_clear = !use_tab;
    RG_TMARK(0);
    for (int cbase = up0; cbase < npx; cbase += RG_LIST) {
    constexpr int NPT = (RG_LIST + RGB - 1) / RGB;
    unsigned long long c0[NPT], c1[NPT];
    uint32_t ceb[NPT], cbt[NPT];
    const int cend = min(cbase + RG_LIST, npx);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = min(cbase + tid + j * RGB, cend - 1);
      const int yi = sy0 + i / ww, xi = x0 + i % ww;
      const size_t gp = (size_t)t * P + (size_t)yi * W + xi;
      const unsigned long long* q = gk + (size_t)i * 5;
      c0[j] = q[0]; c1[j] = q[1];
      ceb[j] = p.ebits[gp]; cbt[j] = p.bits[gp];
    }
    if (tid == 0) *s_n = 0;
    if (!table_clear) {
      __syncthreads();          // the previous unit's flush has read the table
      for (int i = tid; i < p.V * 3; i += RGB) gtab[i] = 0.f;
      table_clear = true;
    }
    __syncthreads();
    RG_TMARK(1);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = cbase + tid + j * RGB;
      const bool dep = gA != 0.f && pvalid != 0.f && c0[j] != RS_EMPTY && ((ceb[j] >> n) & 1u);
      const bool sil = sil_on && c1[j] != RS_EMPTY && (cbt[j] & fr) == 0u;
      const bool live = i < cend && (dep || sil);
      const unsigned long long m = __ballot(live);
      if (m) {
        int base = 0;
        const int lane = tid & 63;
        if (lane == 0) base = atomicAdd(s_n, __popcll(m));
        base = __shfl(base, 0, 64);
        if (live) plist[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
      }
    }
    __syncthreads();
    RG_TMARK(2);
    const int nlive = *s_n;
    for (int li_ = tid; li_ < nlive; li_ += RGB) {
      const int i = plist[li_];
      rg_pixel<TAB ? 1 : 0>(p, bd, i, lcorr, dc);
    }
    __syncthreads();
    RG_TMARK(3);
    }   // classification pass
    lcorr = r_block_sum(lcorr, s_red);
    if (tid == 0 && lcorr != 0.f) atomicAdd(&p.sil_corr[b], lcorr);
    if (use_tab)
      for (int i = tid; i < p.V * 3; i += RGB) {
        const float g = gtab[i];
        if (g != 0.f) atomicAdd(&gvb[i], g);      // several units of one body may flush concurrently
      }
    RG_TMARK(4);
  }
  RG_TIMING_FLUSH();
}

// Deterministic form of the gradient scatter (mh_raster_set_deterministic / MHHIP_DETERMINISTIC=1): one workgroup per
// BODY, every window pixel evaluated by a fixed thread, contributions summed as 64-bit fixed-point integers in LDS
// (integer atomics commute exactly), one plain read-modify-write of dL/dverts per element -- the result is the same
// bits from run to run whatever the wave schedule.  Three passes over the body's pixels (see RgDet): ~4-5x the time of
// the production kernel; a verification mode, not a fast path.  The fixed-point step is 2^-46 of the body's largest
// contribution, i.e. 2^22 times finer than the fp32 sums of the production kernel.
__global__ __launch_bounds__(RGB) void k_raster_grads_det(RasterP p) {
  __shared__ float s_red[RGB / 64];
  __shared__ int s_shift;
  unsigned long long* tab = (unsigned long long*)rg_tab;
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W, P = H * W;
  const int VH = (p.V + 1) / 2;
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    const int t = b / p.N, n = b % p.N;
    const int x0 = p.win[b * 4], ww = p.win[b * 4 + 2], y0 = p.win[b * 4 + 1], wh = p.win[b * 4 + 3];
    if (ww <= 0 || wh <= 0) continue;
    const int npx = ww * wh;
    float S[6];
    r_body_sums(p, b, S);
    const float cnt = S[2] + 1.f;
    const float diff = S[0] / cnt - S[1] / cnt;
    RgBody bd;
    bd.t = t; bd.n = n; bd.x0 = x0; bd.sy0 = y0; bd.ww = ww;
    bd.vb = p.ndc + (size_t)b * p.V * 3;
    bd.gvb = p.gverts + (size_t)b * p.V * 3;
    bd.gk = p.gkeys + (size_t)p.body_koff[b] * 5;
    bd.gA = p.coef_depth * 2.f * diff / cnt;
    bd.gAlphaScale = p.coef_sil * p.sil_apply[b] * 2.f / (p.sil_D[b] + 1.f);
    bd.sil_on = p.sil_apply[b] != 0.f;
    bd.pvalid = p.p2d_valid[b];
    bd.fr = p.front[b];
    RgDet dc;
    dc.umax = 0.f; dc.shift = 0;
    float lcorr = 0.f;
    for (int pass = 0; pass < 3; ++pass) {
      dc.pass = pass;
      dc.v0 = pass == 2 ? VH : 0;
      dc.v1 = pass == 1 ? VH : p.V;
      __syncthreads();
      if (pass > 0) {
        dc.shift = s_shift;
        for (int i = tid; i < VH * 3; i += RGB) tab[i] = 0ull;
        __syncthreads();
      }
      float lc = 0.f;
      for (int i = tid; i < npx; i += RGB) {
        const int yi = y0 + i / ww, xi = x0 + i % ww;
        const size_t gp = (size_t)t * P + (size_t)yi * W + xi;
        const unsigned long long* q = bd.gk + (size_t)i * 5;
        const bool dep = bd.gA != 0.f && bd.pvalid != 0.f && q[0] != RS_EMPTY && ((p.ebits[gp] >> n) & 1u);
        const bool sil = bd.sil_on && q[1] != RS_EMPTY && (p.bits[gp] & bd.fr) == 0u;
        if (dep || sil) rg_pixel<2>(p, bd, i, lc, dc);
      }
      if (pass == 0) {
        lcorr = r_block_sum(lc, s_red);                     // fixed thread -> pixel map, fixed tree: deterministic
        float m = dc.umax;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
          float mm = 0.f;
          for (int w = 0; w < RGB / 64; ++w) mm = fmaxf(mm, s_red[w]);
          int e = 0;
          if (mm > 0.f && mm < INFINITY) (void)frexpf(mm, &e);   // mm < 2^e
          s_shift = 46 - e;                                      // |contribution| 2^shift < 2^46; < 2^16 of them per vertex
        }
      } else {
        __syncthreads();
        const int nv = (dc.v1 - dc.v0) * 3;
        float* o = bd.gvb + (size_t)dc.v0 * 3;
        for (int i = tid; i < nv; i += RGB) {
          const long long a = (long long)tab[i];
          if (a != 0) o[i] += (float)ldexp((double)a, -dc.shift);   // the only writer of this element in this kernel
        }
      }
    }
    if (tid == 0 && lcorr != 0.f) p.sil_corr[b] += lcorr;
  }
  __syncthreads();
  r_deferred_sorts<RGB>(p, (int*)rg_tab);
}

__global__ void k_fill(float* x, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}

// 0 (default): float atomics in the gradient scatter (fast, summation order varies from run to run);
// 1: k_raster_grads_det (bit-reproducible)
static int g_raster_det = -1;
static bool raster_deterministic() {
  if (g_raster_det < 0) {
    const char* e = getenv("MHHIP_DETERMINISTIC");
    g_raster_det = (e && e[0] == '1') ? 1 : 0;
  }
  return g_raster_det != 0;
}
extern "C" int mh_raster_set_deterministic(int on) {
  g_raster_det = on ? 1 : 0;
  return MH_OK;
}
extern "C" int mh_raster_get_deterministic(void) { return raster_deterministic() ? 1 : 0; }

// rows a vertex may move before its body's face lists are sorted again (0 = sort every launch, the round-2 behaviour)
static int g_raster_margin = -1;
static int raster_sort_margin() {
  if (g_raster_margin < 0) {
    const char* e = getenv("MHHIP_RASTER_SORT_MARGIN");
    g_raster_margin = e ? atoi(e) : 1;
    if (g_raster_margin < 0 || g_raster_margin > 8) g_raster_margin = 1;
  }
  return g_raster_margin;
}
extern "C" int mh_raster_set_sort_margin(int rows) {
  MH_CHECK(rows >= 0 && rows <= 8, "margin must be 0..8 rows");
  g_raster_margin = rows;
  return MH_OK;
}
extern "C" int mh_raster_get_sort_margin(void) { return raster_sort_margin(); }
// deferred sorts: the fraction of the margin from which a body's lists are sorted again beside the gradient kernel (for the
// next launch) instead of on the chain once the whole margin is used up.  0 (or >= 1): no deferred sorts.
static float g_raster_defer = -1.f;
static float raster_sort_defer() {
  if (g_raster_defer < 0.f) {
    const char* e = getenv("MHHIP_RASTER_SORT_DEFER");
    g_raster_defer = e ? (float)atof(e) : 0.6f;
    if (!(g_raster_defer >= 0.f && g_raster_defer <= 1.f)) g_raster_defer = 0.6f;
  }
  return g_raster_defer;
}
static float raster_sort_soft(int margin) {
  const float hard = (float)margin - 0.02f, f = raster_sort_defer();
  return (f > 0.f && f < 1.f && margin > 0) ? fminf(f * (float)margin, hard) : hard;
}
extern "C" int mh_raster_set_sort_defer(float fraction) {
  MH_CHECK(fraction >= 0.f && fraction <= 1.f, "fraction of the margin: 0..1 (0 or 1: no deferred sorts)");
  g_raster_defer = fraction;
  return MH_OK;
}
extern "C" float mh_raster_get_sort_defer(void) { return raster_sort_defer(); }
// test aid: 1 sends every round of the selection kernel down the even-split path (no depth cull, no pair list).  The keys
// must not depend on the path a round takes (tests/test_raster_paths_gpu.py): the pair arithmetic is spelled out for that.
static int g_raster_all_even = 0;
extern "C" int mh_raster_set_path(int all_even) {
  g_raster_all_even = all_even ? 1 : 0;
  return MH_OK;
}
extern "C" int mh_raster_get_path(void) { return g_raster_all_even; }

// winners' list of the face sort (r_face_sort): 1 (default; MHHIP_RASTER_WINNERS=0 switches it off) = the faces that held a
// key in the previous launch are rasterised first by every tile.  An order only: the keys are the same bits either way.
static int g_raster_winners = -1;
static int raster_winners() {
  if (g_raster_winners < 0) {
    const char* e = getenv("MHHIP_RASTER_WINNERS");
    g_raster_winners = (e && e[0] == '0') ? 0 : 1;
  }
  return g_raster_winners;
}
extern "C" int mh_raster_set_winners(int on) {
  g_raster_winners = on ? 1 : 0;
  return MH_OK;
}
extern "C" int mh_raster_get_winners(void) { return raster_winners(); }
// dynamic LDS of k_raster_prepare: the sort's [4][H+1] histogram + the winners' bitmap
static size_t r_prepare_lds(const RasterP& p) {
  return ((size_t)4 * (p.H + 1) + (p.F <= R_WIN_MAXF ? (size_t)(p.F + 31) / 32 : 0)) * sizeof(int);
}

static size_t r_align(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t r_max_units(size_t B, int H, int W) { return B + B * (size_t)H * W / RG_UNIT + 1; }
static int r_max_strips(int B, int H, int W) {
  // full-width windows give the most tiles per body
  const int th = W <= R_CAP ? (R_CAP / W > 0 ? R_CAP / W : 1) : R_CAP / R_TILE_W;
  const int tw = W <= R_CAP ? W : R_TILE_W;
  const int per_body = ((W + tw - 1) / tw) * ((H + th - 1) / th);
  return B * (per_body > H ? per_body : H);
}

static size_t r_carve(RasterP& p, void* ws) {
  const size_t B = (size_t)p.B;
  const int V = p.V, F = p.F, H = p.H, W = p.W;
  p.max_strips = r_max_strips(p.B, H, W);
  const size_t ms = (size_t)p.max_strips;
  char* c = (char*)ws;
  p.win = (int*)c; c += r_align(B * 4 * 4);
  p.body_first = (int*)c; c += r_align(B * 4);
  p.body_ns = (int*)c; c += r_align(B * 4);
  p.strip_body = (int*)c; c += r_align(ms * 4);
  p.strip_row0 = (int*)c; c += r_align(ms * 4);
  p.strip_rows = (int*)c; c += r_align(ms * 4);
  p.strip_col0 = (int*)c; c += r_align(ms * 4);
  p.strip_cols = (int*)c; c += r_align(ms * 4);
  p.partial = (float*)c; c += r_align(ms * 6 * 4);
  p.dinv = (float*)c; c += r_align(B * 2 * 4);
  p.ndc = (float*)c; c += r_align(B * V * 3 * 4);
  p.frows = (unsigned*)c; c += r_align(B * F * 4);
  p.fsort = (unsigned*)c; c += r_align(B * F * 4);
  p.row_start = (int*)c; c += r_align(B * (size_t)(4 * (H + 1) + 1) * 4);
  p.maxh = (int*)c; c += r_align(B * 4);
  p.body_koff = (long long*)c; c += r_align(B * 8);
  p.rowb = (float*)c; c += r_align(B * V * 4);
  p.margin = raster_sort_margin();
  p.all_even = g_raster_all_even;
  p.max_units = (int)r_max_units(B, H, W);
  p.strip_cls = (int*)c; c += r_align(ms * 4);
  p.strip_order = (int*)c; c += r_align(ms * 4);
  p.gunit_list = (unsigned long long*)c; c += r_align((size_t)p.max_units * 8);
  p.stale = (int*)c; c += r_align(B * 4);
  // control words: everything mh_raster_workspace_init clears, contiguous (the work lists' totals and coverage among them: an
  // untouched workspace lists nothing, and the selection / gradient kernels then take every tile / unit from the bodies)
  p.ctl = (unsigned*)c; c += r_align(16);
  p.total = (int*)c; c += r_align(4);
  p.gunit_total = (int*)c; c += r_align(4);
  p.ns_listed = (int*)c; c += r_align(B * 4);
  p.nu_listed = (int*)c; c += r_align(B * 4);
  p.sort_count = (unsigned long long*)c; c += r_align(24);
  p.pairs = (unsigned long long*)c; c += r_align((2 + 2 * (size_t)R_STRIP_GRID) * 8);
  p.sort_tag = (unsigned long long*)c; c += r_align(B * 8);
  p.sil_corr = (float*)c; c += r_align(B * 4);
  p.kvalid = (int*)c; c += r_align(B * 4);
  p.wstate = (int*)c; c += r_align(B * 4);
  p.resort = (int*)c; c += r_align(B * 4);
  p.winners_on = raster_winners();
  p.soft = raster_sort_soft(p.margin);
  p.ctl_end = c;
  p.fbbox = (int*)c; c += r_align(B * 4 * 4);
  p.fbbox_prev = (int*)c; c += r_align(B * 4 * 4);
  p.flowkey = (unsigned long long*)c; c += r_align(B * 8);
  p.flowkey_prev = (unsigned long long*)c; c += r_align(B * 8);
  p.fmoved = (int*)c; c += r_align(2 * B * 4);
  p.projected = 0;
  p.gkeys = (unsigned long long*)c; c += r_align(B * (size_t)H * W * 5 * 8);
  return (size_t)(c - (char*)ws);
}

extern "C" size_t mh_raster_workspace_bytes(int T, int N, int V, int F, int H, int W) {
  if (T <= 0 || N <= 0 || V <= 0 || F <= 0 || H <= 0 || W <= 0) return 0;
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  return r_carve(p, nullptr);
}

// A workspace must be initialised ONCE before its first launch (and again if its bytes were overwritten): the control
// words -- the ticket of the preparation kernel, the re-sort counters, face-list tags, the silhouette accumulator -- are cleared;
// everything else is rebuilt by the launches themselves.  Stream-ordered.
extern "C" int mh_raster_workspace_init(int T, int N, int V, int F, int H, int W, void* ws, void* stream) {
  MH_CHECK(ws, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, ws);
  MH_HIP(hipMemsetAsync(p.ctl, 0, (size_t)(p.ctl_end - (char*)p.ctl), (hipStream_t)stream));
  return MH_OK;
}

// byte offsets inside the workspace of what an inspection tool reads after a launch: [0] win (B x 4 int32),
// [1] body_koff (B x int64: first window pixel of a body in the key array), [2] the key array (5 x uint64 per window pixel)
extern "C" int mh_raster_workspace_offsets(int T, int N, int V, int F, int H, int W, size_t* out /*[3]*/) {
  MH_CHECK(out, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, nullptr);
  out[0] = (size_t)((char*)p.win - (char*)nullptr);
  out[1] = (size_t)((char*)p.body_koff - (char*)nullptr);
  out[2] = (size_t)((char*)p.gkeys - (char*)nullptr);
  return MH_OK;
}

// (developer aid, not in the public header) more offsets: ndc, frows, fsort, row_start, maxh, rowb; mh_raster_debug_offsets2:
// kvalid, wstate (B int32 each), sort_tag (B uint64)
extern "C" int mh_raster_debug_offsets2(int T, int N, int V, int F, int H, int W, size_t* out /*[3]*/) {
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, nullptr);
  out[0] = (size_t)((char*)p.kvalid - (char*)nullptr);
  out[1] = (size_t)((char*)p.wstate - (char*)nullptr);
  out[2] = (size_t)((char*)p.sort_tag - (char*)nullptr);
  return MH_OK;
}
extern "C" int mh_raster_debug_offsets(int T, int N, int V, int F, int H, int W, size_t* out /*[6]*/) {
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, nullptr);
  out[0] = (size_t)((char*)p.ndc - (char*)nullptr);
  out[1] = (size_t)((char*)p.frows - (char*)nullptr);
  out[2] = (size_t)((char*)p.fsort - (char*)nullptr);
  out[3] = (size_t)((char*)p.row_start - (char*)nullptr);
  out[4] = (size_t)((char*)p.maxh - (char*)nullptr);
  out[5] = (size_t)((char*)p.rowb - (char*)nullptr);
  return MH_OK;
}

// {launches, candidate pairs, evaluated pairs} of k_raster_strip, counted while mh_profile_enable(1) (synchronises)
extern "C" int mh_raster_pair_counters(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host, void* stream) {
  MH_CHECK(ws && out_host, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, ws);
  static unsigned long long host[2 + 2 * R_STRIP_GRID];
  MH_HIP(hipMemcpyAsync(host, p.pairs, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MH_HIP(hipStreamSynchronize((hipStream_t)stream));
  out_host[0] = host[0]; out_host[1] = 0ull; out_host[2] = 0ull;
  R_TIMING_HOST_SPANS();
  for (int i = 0; i < R_STRIP_GRID; ++i) { out_host[1] += host[2 + 2 * i]; out_host[2] += host[3 + 2 * i]; }
  return MH_OK;
}

extern "C" int mh_raster_sort_counters(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host, void* stream) {
  MH_CHECK(ws, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, ws);
  if (out_host) {
    MH_HIP(hipMemcpyAsync(out_host, p.sort_count, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MH_HIP(hipStreamSynchronize((hipStream_t)stream));
  } else {            // out_host NULL: reset (a workspace holds anything when it is handed over)
    MH_HIP(hipMemsetAsync(p.sort_count, 0, 24, (hipStream_t)stream));
  }
  return MH_OK;
}

// {bodies seen, bodies re-sorted, of which beside the gradient kernel (deferred)}
extern "C" int mh_raster_sort_counters3(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host, void* stream) {
  MH_CHECK(ws && out_host, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_carve(p, ws);
  MH_HIP(hipMemcpyAsync(out_host, p.sort_count, 24, hipMemcpyDeviceToHost, (hipStream_t)stream));
  MH_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MH_OK;
}

// NDC calibration of transforms.py:222-255 with image_size = (W, H) and PyTorch3D's R = diag(-1,-1,1)
static void r_calibration(RasterP& p, const float* cam_K_host) {
  const int W = p.W, H = p.H;
  const float fx = cam_K_host[0], fy = cam_K_host[4], cx = cam_K_host[2], cy = cam_K_host[5];
  if (W > H) {
    p.s = 2.f * fy / H;
    const float u = (float)W / H;
    p.w1 = u * (W - 2.f * cx) / W;
    p.h1 = (H - 2.f * cy) / H;
  } else if (H > W) {
    p.s = 2.f * fx / W;
    const float u = (float)H / W;
    p.w1 = (W - 2.f * cx) / W;
    p.h1 = u * (H - 2.f * cy) / H;
  } else {
    p.s = 2.f * (fx + fy) / (W + H);
    p.w1 = (W - 2.f * cx) / W;
    p.h1 = (H - 2.f * cy) / H;
  }
}

static int raster_terms_impl(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                             const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                             const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                             const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                             float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                             float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                             int phases, float* log_depth, float* log_sil, void* stream, int projected = 0,
                             mh_raster_fin* fin_out = nullptr) {
  MH_CHECK(cam_K_host && (verts || projected) && faces && bits && ebits && depths && zmin_lin && zmax_lin && pose2d_valid && front &&
               sil_apply && sil_D && sil_S && depth_body && sil_body && ws,
           "null argument");
  MH_CHECK(T > 0 && N > 0 && N <= 32 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  MH_CHECK(H <= 4095 && W <= 65535 && F < (1 << 20), "sorted face entries hold 12-bit rows and 20-bit face ids");
  MH_CHECK(V < (1 << 22), "vertex offsets of the gathers are 24-bit products (3 * vertex index)");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_calibration(p, cam_K_host);
  p.verts = verts; p.faces = faces; p.bits = bits; p.ebits = ebits; p.depths = depths;
  p.zmin_lin = zmin_lin; p.zmax_lin = zmax_lin; p.p2d_valid = pose2d_valid; p.front = front;
  p.sil_apply = sil_apply; p.sil_D = sil_D; p.sil_S = sil_S;
  p.coef_depth = coef_depth; p.coef_sil = coef_sil; p.eps = eps;
  p.gverts = gverts; p.depth_body = depth_body; p.sil_body = sil_body;
  p.zbuf_out = zbuf_out; p.alpha_out = alpha_out;
  r_carve(p, ws);
  p.projected = projected ? 1 : 0;
  if (mh_prof_level() < 2) p.pairs = nullptr;
  hipStream_t st = (hipStream_t)stream;
  // phase bits: 1 = preparation + selection, 4 = preparation only (window, face lists, work lists), 8 = selection only
  const bool do_prep = (phases & (1 | 4)) != 0, do_sel = (phases & (1 | 8)) != 0;
  if (do_prep) {
  if (zbuf_out) {   // -1 = empty, like fragments.zbuf
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, st, zbuf_out, (size_t)p.B * H * W, -1.f);
    MH_LAUNCH_CHECK();
  }
  if (alpha_out) MH_HIP(hipMemsetAsync(alpha_out, 0, (size_t)p.B * H * W * sizeof(float), st));
  mh_prof_mark(MH_PROF_RASTER_PREP, 0, st);
  hipLaunchKernelGGL(k_raster_prepare, dim3(p.B), dim3(RPREP), r_prepare_lds(p), st, p);
  MH_LAUNCH_CHECK();
  // The work lists are a schedule (the selection and gradient kernels find every tile and unit with lists that are a launch
  // old, or empty).  phases & 128: they are NOT rebuilt here, between the preparation and the selection, but by whoever
  // carries the closing job of this launch's gradient half (mh_raster_fin.lists, beside the LBS backward): 16 us off the
  // chain.  Values-only launches read the lists in k_raster_sums: always rebuilt here.
  if (!(phases & 128) || !gverts || zbuf_out || alpha_out) {
    hipLaunchKernelGGL(k_raster_lists, dim3(1), dim3(RLISTS), 0, st, p);
    MH_LAUNCH_CHECK();
  }
  mh_prof_mark(MH_PROF_RASTER_PREP, 1, st);
  }
  if (do_sel) {
  // persistent grids over the device-side work list (the strip count is only known on the device)
  const int grid = R_STRIP_GRID;
  mh_prof_mark(MH_PROF_RASTER_STRIP, 0, st);
  hipLaunchKernelGGL(k_raster_strip, dim3(grid), dim3(RB), 0, st, p);
  MH_LAUNCH_CHECK();
  mh_prof_mark(MH_PROF_RASTER_STRIP, 1, st);
  if (!gverts || zbuf_out || alpha_out) {   // values only (render, loss evaluation) or images wanted: alpha is evaluated by k_raster_sums
    mh_prof_mark(MH_PROF_RASTER_SUMS, 0, st);
    hipLaunchKernelGGL(k_raster_sums, dim3(grid), dim3(RB), 0, st, p);
    MH_LAUNCH_CHECK();
    mh_prof_mark(MH_PROF_RASTER_SUMS, 1, st);
  }
  if (!gverts || zbuf_out || alpha_out) {   // with gradients the per-body values come out of k_raster_finish
    hipLaunchKernelGGL(k_raster_body_out, dim3((p.B + 255) / 256), dim3(256), 0, st, p);
    MH_LAUNCH_CHECK();
  }
  }   // selection phase
  if (!(phases & 2)) return MH_OK;
  if (gverts) {
    const bool use_tab = V <= RG_MAXV;
    const size_t tab = (use_tab ? (size_t)V * 3 * sizeof(float) : 0) + (RG_LIST + 1) * sizeof(int);
    static unsigned char attr_set[MH_MAX_DEVICES];
    if (mh_first_on_device(attr_set))
      MH_HIP(hipFuncSetAttribute((const void*)k_raster_grads<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_MAXV * 3 * 4 + (RG_LIST + 1) * 4));
    // sil_corr is zero here: cleared by k_raster_body_out (phase 1) and again by k_raster_finish after every use
    mh_prof_mark(MH_PROF_RASTER_GRADS, 0, st);
    // one workgroup per CU is resident (LDS table); the work-unit count is only known on the device
    const int ggrid = 256 * 6;
    RasterP pg = p;            // the deferred sorts ride in this launch when its dynamic LDS holds their tables
    if (raster_deterministic()) {
      MH_CHECK(V <= RG_MAXV, "deterministic gradient scatter: model too large for the LDS table");
      static unsigned char attr_det[MH_MAX_DEVICES];
      if (mh_first_on_device(attr_det))
        MH_HIP(hipFuncSetAttribute((const void*)k_raster_grads_det, hipFuncAttributeMaxDynamicSharedMemorySize, ((RG_MAXV + 1) / 2) * 3 * 8));
      if (r_prepare_lds(p) > (size_t)((V + 1) / 2) * 3 * 8) pg.resort = nullptr;
      hipLaunchKernelGGL(k_raster_grads_det, dim3(p.B < 4096 ? p.B : 4096), dim3(RGB), (size_t)((V + 1) / 2) * 3 * 8, st, pg);
    } else {
      if (r_prepare_lds(p) > tab) pg.resort = nullptr;
      if (use_tab) hipLaunchKernelGGL(k_raster_grads<true>, dim3(ggrid), dim3(RGB), tab, st, pg);
      else hipLaunchKernelGGL(k_raster_grads<false>, dim3(ggrid), dim3(RGB), tab, st, pg);
    }
    MH_LAUNCH_CHECK();
    mh_prof_mark(MH_PROF_RASTER_GRADS, 1, st);
  }
  if (fin_out) memset(fin_out, 0, sizeof(*fin_out));
  if (gverts || (gzmin && gzmax) || log_depth || log_sil) {
    // values-only launches with images requested went through k_raster_body_out (sums of k_raster_sums): then only the chain
    mh_raster_fin f;
    f.T = T; f.N = N; f.B = p.B; f.from_partials = (gverts && !zbuf_out && !alpha_out) ? 1 : 0;
    f.coef_depth = p.coef_depth;
    f.body_first = p.body_first; f.body_ns = p.body_ns; f.partial = p.partial; f.dinv = p.dinv;
    f.sil_apply = p.sil_apply; f.sil_D = p.sil_D; f.sil_S = p.sil_S; f.sil_corr = p.sil_corr;
    f.depth_body = p.depth_body; f.sil_body = p.sil_body;
    f.zmin_lin = zmin_lin; f.zmax_lin = zmax_lin;
    f.gzmin = (gzmin && gzmax) ? gzmin : (float*)nullptr; f.gzmax = gzmax;
    f.log_depth = log_depth; f.log_sil = log_sil;
    if (fin_out) {
      *fin_out = f;            // the caller's next launch carries the job (mh_lbs_backward_kp_fin)
      if (phases & 128) {      // ... and rebuilds the work lists for the next launch on this workspace
        static_assert(sizeof(RasterP) <= sizeof(fin_out->lists), "mh_raster_fin.lists too small for the parameter block");
        memcpy(fin_out->lists, &p, sizeof(RasterP));
        fin_out->has_lists = 1;
      }
    } else {
      hipLaunchKernelGGL(k_raster_finish, dim3(1), dim3(RFIN), 0, st, f);
      MH_LAUNCH_CHECK();
    }
  }
  return MH_OK;
}

extern "C" int mh_raster_terms(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                               const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                               const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                               const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                               float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                               float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                               void* stream) {
  return raster_terms_impl(T, N, V, F, H, W, cam_K_host, verts, faces, bits, ebits, depths, zmin_lin, zmax_lin, pose2d_valid, front,
                           sil_apply, sil_D, sil_S, coef_depth, coef_sil, eps, gverts, gzmin, gzmax, depth_body, sil_body, ws, zbuf_out,
                           alpha_out, 3, nullptr, nullptr, stream);
}

extern "C" int mh_raster_terms_phase(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                                     const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                                     const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                                     const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                                     float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                                     float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                                     int phases, void* stream) {
  MH_CHECK(phases >= 1 && phases <= 3, "phases: 1 = selection + values, 2 = gradients, 3 = both");
  return raster_terms_impl(T, N, V, F, H, W, cam_K_host, verts, faces, bits, ebits, depths, zmin_lin, zmax_lin, pose2d_valid, front,
                           sil_apply, sil_D, sil_S, coef_depth, coef_sil, eps, gverts, gzmin, gzmax, depth_body, sil_body, ws, zbuf_out,
                           alpha_out, phases, nullptr, nullptr, stream);
}

extern "C" int mh_raster_terms_phase_log(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                                         const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                                         const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                                         const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                                         float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                                         float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                                         int phases, float* log_depth, float* log_sil, void* stream) {
  MH_CHECK(phases >= 1 && phases <= 3, "phases: 1 = selection + values, 2 = gradients, 3 = both");
  return raster_terms_impl(T, N, V, F, H, W, cam_K_host, verts, faces, bits, ebits, depths, zmin_lin, zmax_lin, pose2d_valid, front,
                           sil_apply, sil_D, sil_S, coef_depth, coef_sil, eps, gverts, gzmin, gzmax, depth_body, sil_body, ws, zbuf_out,
                           alpha_out, phases, log_depth, log_sil, stream);
}

extern "C" int mh_raster_terms_projected(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                                         const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                                         const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                                         const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                                         float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                                         float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                                         int phases, float* log_depth, float* log_sil, int projected, void* stream) {
  MH_CHECK(phases >= 1 && (phases & ~(15 | 128)) == 0, "phases: 1 = preparation + selection + values, 2 = gradients, 4 = preparation only, 8 = selection only, 128 = work lists deferred");
  return raster_terms_impl(T, N, V, F, H, W, cam_K_host, verts, faces, bits, ebits, depths, zmin_lin, zmax_lin, pose2d_valid, front,
                           sil_apply, sil_D, sil_S, coef_depth, coef_sil, eps, gverts, gzmin, gzmax, depth_body, sil_body, ws, zbuf_out,
                           alpha_out, phases, log_depth, log_sil, stream, projected);
}

extern "C" int mh_raster_terms_deferred(int T, int N, int V, int F, int H, int W, const float* cam_K_host, const float* verts,
                                         const int32_t* faces, const uint32_t* bits, const uint32_t* ebits, const float* depths,
                                         const float* zmin_lin, const float* zmax_lin, const float* pose2d_valid,
                                         const uint32_t* front, const float* sil_apply, const float* sil_D, const float* sil_S,
                                         float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin, float* gzmax,
                                         float* depth_body, float* sil_body, void* ws, float* zbuf_out, float* alpha_out,
                                         int phases, float* log_depth, float* log_sil, int projected, mh_raster_fin* fin_out, void* stream) {
  MH_CHECK(phases >= 1 && (phases & ~(15 | 128)) == 0, "phases: 1 = preparation + selection + values, 2 = gradients, 4 = preparation only, 8 = selection only, 128 = work lists deferred");
  MH_CHECK(fin_out, "null argument");
  return raster_terms_impl(T, N, V, F, H, W, cam_K_host, verts, faces, bits, ebits, depths, zmin_lin, zmax_lin, pose2d_valid, front,
                           sil_apply, sil_D, sil_S, coef_depth, coef_sil, eps, gverts, gzmin, gzmax, depth_body, sil_body, ws, zbuf_out,
                           alpha_out, phases, log_depth, log_sil, stream, projected, fin_out);
}

// where mh_lbs_forward_proj writes for this workspace, and the constants of the projection / motion test (the host-side
// twins of r_row_affine and of the threshold in k_raster_prepare: one definition each)
extern "C" int mh_raster_forward_targets(int T, int N, int V, int F, int H, int W, const float* cam_K_host, void* ws,
                                         mh_fwd_proj* out) {
  MH_CHECK(cam_K_host && ws && out, "null argument");
  MH_CHECK(T > 0 && N > 0 && V > 0 && F > 0 && H > 0 && W > 0, "empty input");
  RasterP p;
  p.B = T * N; p.N = N; p.V = V; p.F = F; p.H = H; p.W = W;
  r_calibration(p, cam_K_host);
  r_carve(p, ws);
  out->s = p.s; out->w1 = p.w1; out->h1 = p.h1;
  float range = 2.0f;                                   // r_row_affine
  if (H > W) range = ((float)H * range) / (float)W;
  out->rk = (float)H / range;
  out->ra = (float)H - 0.5f - 0.5f * (float)H;
  out->thr = (float)p.margin - 0.02f;
  out->thr_soft = p.soft;
  // half a pixel: the optimiser moves a body by a small fraction of a pixel per cycle; a faster body is scanned
  out->slack_ndc = 0.5f * 2.0f / (float)(H < W ? H : W);
  out->slack_y = 0.005f;
  if (const char* e = getenv("MHHIP_PROJ_SLACK_PX")) out->slack_ndc = (float)atof(e) * 2.0f / (float)(H < W ? H : W);
  out->ndc = p.ndc; out->rowb = p.rowb;
  out->bbox = p.fbbox; out->bbox_prev = p.fbbox_prev;
  out->lowkey = p.flowkey; out->lowkey_prev = p.flowkey_prev;
  out->moved = p.fmoved;
  out->clear = nullptr; out->clear_n = 0ull;
  return MH_OK;
}
