/* ORACLE (test infrastructure only) -- brute-force face selection of the mesh rasteriser.
 *
 * The reference calls PyTorch3D's MeshRasterizer (optimizer.py:211-232, 428-431, 447-448).
 * PyTorch3D is a third-party dependency that is neither vendored in the reference nor available
 * offline (environment.yml:13, un-pinned), so this file restates its published algorithm
 * (pytorch3d/csrc/rasterize_meshes: CheckPixelInsideFace / RasterizeMeshesNaive, v0.6-0.7):
 * for every pixel keep the K nearest faces (by interpolated z) among the faces whose blurred
 * footprint covers the pixel centre.  PARITY UNPINNED: no PyTorch3D output exists to pin this
 * against; tests/test_raster_oracle.py pins it with analytic known-answer cases instead.
 *
 * WHICH BACKEND each rule restates (from the published sources as recalled -- there is no copy to read offline):
 *   - pixel centres (PixToNonSquareNdc), blurred bounding-box test, |face area| <= kEpsilon skip, clipped barycentrics
 *     (clip_barycentric_coords, blur_radius > 0), pz < 0 skip, inside-or-(dist < blur_radius), K nearest by pz with the
 *     earlier face winning ties: common to RasterizeMeshesNaiveCpu (rasterize_meshes_cpu.cpp) and the CUDA kernels
 *     (rasterize_meshes.cu: CheckPixelInsideFace);
 *   - faces with a vertex behind the camera, zmin < kEpsilon, skipped ENTIRELY: the CUDA kernels' rule (CheckPixelInsideFace:
 *     `zmax < 0` / the coarse pass's `z_invalid = zmin < kEpsilon`), which newer CPU sources apply through the face bounding
 *     boxes as well.  The plain naive CPU loop drops only the PIXELS whose interpolated pz < 0 of such a face.
 *     raster_select_set_behind_camera_rule(1) = whole-face rule (DEFAULT; what the HIP kernel implements: the reference
 *     fits on a GPU, predict.py), 0 = per-pixel rule only.  The two differ only for a face that straddles the camera plane
 *     z = 0 -- never the case at MuPoTs depths (bodies 2-10 m in front of the camera); tests/test_raster_oracle.py holds a
 *     known-answer case for each.
 *
 * Only the discrete selection happens here; the differentiable quantities (barycentrics, z,
 * distances) are recomputed from the selected faces in torch (oracle/raster_oracle.py) so that
 * autograd provides the reference gradients.
 *
 * Input vertices are already in PyTorch3D NDC (+x left, +y up, z = view depth).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define K_EPS 1e-8f

static int g_behind_rule = 1;   /* 1: skip faces with zmin < kEpsilon (CUDA kernels; default), 0: only pixels with pz < 0 */
void raster_select_set_behind_camera_rule(int whole_face) { g_behind_rule = whole_face ? 1 : 0; }
int raster_select_get_behind_camera_rule(void) { return g_behind_rule; }

static float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
  float bax = bx - ax, bay = by - ay;
  float l2 = bax * bax + bay * bay;
  if (l2 <= K_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
  float qx = ax + t * bax - px, qy = ay + t * bay - py;
  return qx * qx + qy * qy;
}

/* pixel index -> NDC of the pixel centre (non-square images: the short side spans [-1,1]) */
static float pix_to_ndc(int i, int S1, int S2) {
  float range = 2.0f;
  if (S1 > S2) range = ((float)S1 * range) / (float)S2;
  float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}

/* indices i (0..S1-1, already flipped back to image order) whose centre may fall in [lo, hi] */
static void ndc_range(float lo, float hi, int S1, int S2, int* i_lo, int* i_hi) {
  float range = 2.0f;
  if (S1 > S2) range = ((float)S1 * range) / (float)S2;
  /* ndc(i') = -r/2 + r (i' + 1/2)/S1 with i' = S1-1-i  =>  i = S1 - 1/2 - (ndc + r/2) S1 / r */
  float a = (float)S1 - 0.5f - (hi + range / 2.0f) * (float)S1 / range;
  float b = (float)S1 - 0.5f - (lo + range / 2.0f) * (float)S1 / range;
  int ia = (int)floorf(a) - 1, ib = (int)ceilf(b) + 1;
  if (ia < 0) ia = 0;
  if (ib > S1 - 1) ib = S1 - 1;
  *i_lo = ia;
  *i_hi = ib;
}

/* verts_ndc (V,3); faces (F,3); out_face (H,W,K) int32 (-1 = empty), out_z (H,W,K) */
void raster_select(const float* verts_ndc, const int32_t* faces, int F, int H, int W, float blur_radius, int K,
                   int32_t* out_face, float* out_z) {
  const float blur = sqrtf(blur_radius);
  for (int i = 0; i < H * W * K; ++i) {
    out_face[i] = -1;
    out_z[i] = -1.f;
  }
  for (int f = 0; f < F; ++f) {
    const float* v0 = verts_ndc + 3 * faces[3 * f];
    const float* v1 = verts_ndc + 3 * faces[3 * f + 1];
    const float* v2 = verts_ndc + 3 * faces[3 * f + 2];
    float zmin = fminf(v0[2], fminf(v1[2], v2[2]));
    if (g_behind_rule && zmin < K_EPS) continue;              /* face (partly) behind the camera: whole-face rule */
    float face_area = edge_fn(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]);
    if (face_area <= K_EPS && face_area >= -K_EPS) continue;  /* degenerate */
    float xmin = fminf(v0[0], fminf(v1[0], v2[0])) - blur, xmax = fmaxf(v0[0], fmaxf(v1[0], v2[0])) + blur;
    float ymin = fminf(v0[1], fminf(v1[1], v2[1])) - blur, ymax = fmaxf(v0[1], fmaxf(v1[1], v2[1])) + blur;
    /* candidate pixel ranges from the bbox (one pixel of slack; the exact test follows) */
    int y_lo, y_hi, x_lo, x_hi;
    ndc_range(ymin, ymax, H, W, &y_lo, &y_hi);
    ndc_range(xmin, xmax, W, H, &x_lo, &x_hi);
    for (int yi = y_lo; yi <= y_hi; ++yi) {
      float yf = pix_to_ndc(H - 1 - yi, H, W);
      if (yf > ymax || yf < ymin) continue;
      for (int xi = x_lo; xi <= x_hi; ++xi) {
        float xf = pix_to_ndc(W - 1 - xi, W, H);
        if (xf > xmax || xf < xmin) continue;
        float area = edge_fn(v2[0], v2[1], v0[0], v0[1], v1[0], v1[1]) + K_EPS;
        float w0 = edge_fn(xf, yf, v1[0], v1[1], v2[0], v2[1]) / area;
        float w1 = edge_fn(xf, yf, v2[0], v2[1], v0[0], v0[1]) / area;
        float w2 = edge_fn(xf, yf, v0[0], v0[1], v1[0], v1[1]) / area;
        int inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
        float c0 = fmaxf(w0, 0.f), c1 = fmaxf(w1, 0.f), c2 = fmaxf(w2, 0.f);   /* clip_barycentric_coords (blur > 0) */
        float cs = fmaxf(c0 + c1 + c2, 1e-5f);
        float pz = (c0 / cs) * v0[2] + (c1 / cs) * v1[2] + (c2 / cs) * v2[2];
        if (pz < 0.f) continue;
        float d = fminf(fminf(seg_dist2(xf, yf, v0[0], v0[1], v1[0], v1[1]), seg_dist2(xf, yf, v0[0], v0[1], v2[0], v2[1])),
                        seg_dist2(xf, yf, v1[0], v1[1], v2[0], v2[1]));
        if (!inside && d >= blur_radius) continue;
        /* sorted insertion (ascending z, earlier face wins ties), keep K */
        int32_t* qf = out_face + ((size_t)yi * W + xi) * K;
        float* qz = out_z + ((size_t)yi * W + xi) * K;
        int pos = K;
        for (int k = 0; k < K; ++k)
          if (qf[k] < 0 || pz < qz[k]) {
            pos = k;
            break;
          }
        if (pos == K) continue;
        for (int k = K - 1; k > pos; --k) {
          qf[k] = qf[k - 1];
          qz[k] = qz[k - 1];
        }
        qf[pos] = f;
        qz[pos] = pz;
      }
    }
  }
}
