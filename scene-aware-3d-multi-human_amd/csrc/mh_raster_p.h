// The rasteriser's parameter block and the part of it that another translation unit carries out: the work lists
// (r_finalize_lists).  Internal to the library (not part of the C ABI); included by mh_raster.hip and mh_lbs.hip.
#pragma once
#include "mh_common.h"

#ifndef RG_UNIT
#define RG_UNIT 3072           // window pixels per work unit of the gradient kernel (one classification pass; swept 1536..4096)
#endif

struct RasterP {
  int B, N, V, F, H, W;
  float s, w1, h1;           // x_ndc = -s*x/z + w1, y_ndc = -s*y/z + h1 (transforms.py:222-255, R=diag(-1,-1,1))
  const float* verts;
  const int* faces;
  const uint32_t* bits;
  const uint32_t* ebits;
  const float* depths;
  const float* zmin_lin;
  const float* zmax_lin;
  const float* p2d_valid;
  const uint32_t* front;
  const float* sil_apply;
  const float* sil_D;
  const float* sil_S;
  float coef_depth, coef_sil, eps;
  float* gverts;
  float* depth_body;
  float* sil_body;
  float* zbuf_out;           // (B,H,W) or null
  float* alpha_out;          // (B,H,W) or null
  // workspace
  int max_strips;
  int* win;                  // [B][4] x0,y0,ww,wh (ww <= 0: nothing on screen)
  int* body_first;           // [B] first tile slot of the body (b * max_strips / B)
  int* body_ns;              // [B] tiles of the body
  int* strip_body;           // [max_strips]
  int* strip_row0;           // [max_strips] first image row of the tile
  int* strip_rows;           // [max_strips]
  int* strip_col0;           // [max_strips] first image column of the tile
  int* strip_cols;           // [max_strips]
  long long* body_koff;      // [B] first window pixel of the body in gkeys (window row-major)
  float* partial;            // [max_strips][6]
  float* dinv;               // [B][2]
  unsigned long long* gkeys; // [sum of window pixels][5]
  float* ndc;                // [B][V][3] projected vertices (NDC x, y, view z)
  unsigned* frows;           // [B][F] conservative pixel-row range of every face: lo | hi << 16 (lo > hi: skip)
  unsigned* fsort;           // [B][F] faces ordered by their first row: hi << 20 | face
  int* row_start;            // [B][4][H+1] (+1): per list (winners, near short, far short, tall), first entry of fsort with lo >= row
  int* maxh;                 // [B] tallest face (rows) of the body
  // work lists (put together by the last workgroup of k_raster_prepare)
  int max_units;
  unsigned* ctl;             // (reserved control words)
  int* total;                // [1] number of tiles
  int* strip_cls;            // [max_strips] cost class of a tile (by slot)
  int* strip_order;          // [max_strips] tile slots, most expensive class first
  int* gunit_total;          // [1]
  unsigned long long* gunit_list;   // [max_units] body << 32 | piece: RG_UNIT window pixels of one body, full pieces first
  int* stale;                // [B] 1 = the body's face lists were rebuilt this launch
  // what the work lists COVER (r_finalize_lists): tiles / gradient units of every body when the lists were last put together.
  // The lists may be one launch old (they are rebuilt beside the LBS backward, off the chain: mh_raster_fin): k_raster_strip
  // and k_raster_grads skip listed entries that no longer exist and pick up what is new from these counts.
  int* ns_listed;            // [B]
  int* nu_listed;            // [B]
  float* sil_corr;           // [B] sum over the silhouette pixels of alpha^2 - 2 alpha seg (accumulated by k_raster_grads)
  // temporal coherence of the face sort (see k_raster_face_sort): the sorted lists of a body are kept until one of its
  // vertices has moved `margin` pixel rows away from where it was when the lists were built
  int margin;                // rows (0: rebuild every launch)
  int all_even;              // test aid (mh_raster_set_path): 1 = every round of k_raster_strip takes the even-split path
  float* rowb;               // [B][V] continuous pixel-row coordinate of every vertex at the body's last sort
  unsigned long long* sort_tag;   // [B] validity tag of the body's lists (a fresh workspace holds anything)
  char* ctl_end;             // (host) end of the control words
  unsigned long long* sort_count;  // [3] launches x bodies seen, bodies rebuilt, of which deferred (cumulative)
  unsigned long long* pairs;       // [2 + 2 x R_STRIP_GRID]: launches, -, then per workgroup: candidate (face, pixel-centre) pairs,
                                   // pairs evaluated after the depth cull (cumulative); NULL unless mh_profile_enable(1)
  // what mh_lbs_forward_proj leaves here (include/mhmocap_hip.h, mh_fwd_proj): with projected != 0 the preparation reads
  // these instead of passing over the vertices
  int projected;
  int* fbbox;                // [B][4] order-preserving ints of the NDC extremes (min x, min y, max x, max y); INT_MAX / INT_MIN = unset
  int* fbbox_prev;           // [B][4]
  unsigned long long* flowkey;       // [B]
  unsigned long long* flowkey_prev;  // [B]
  int* fmoved;               // [2][B]: left the band / on the way out (mh_fwd_proj.moved)
  // winners' list of the face sort (round 5, r_face_sort)
  int winners_on;            // mh_raster_set_winners / MHHIP_RASTER_WINNERS (default 1)
  int* kvalid;               // [B] 1 = the body's key region holds the keys of a launch on this workspace
  int* wstate;               // [B] 1 = the body's lists were sorted with a winners' list
  // deferred sorts (round 6): a body whose vertices are on their way out of the band its kept lists cover (mh_fwd_proj.thr_soft
  // <= motion < thr) is still served by those lists THIS launch; k_raster_prepare only flags it, and its lists are sorted
  // again beside this launch's gradient kernel (r_deferred_sorts), from the same coordinates and the keys the selection has
  // just written -- off the chain.  Only a jump of a whole margin inside one cycle sorts on the chain.
  int* resort;               // [B] 1 = sort this body's lists beside the gradient kernel
  float soft;                // rows (mh_raster_set_sort_defer; >= margin - 0.02: no deferred sorts)
};

#define R_SHORT 2            // faces of up to R_SHORT + 1 rows go to the two short lists, taller ones to the third
#define R_NCLS 64            // cost classes of the tiles (0 = most expensive)
#define R_STRIP_GRID (256 * 3 * 4)   // persistent grid of the selection kernel
#define R_NGCLS 33           // gradient work units: class 0 = full units, 1..32 = partial units by decreasing size
__device__ __forceinline__ int r_cap(const RasterP& p) { return p.max_strips / p.B; }

// k_raster_lists (ONE workgroup of NT threads, its own launch behind k_raster_prepare): dense work lists from the per-body
// tables.  (Measured on the way, MI355X: doing this in the workgroup of k_raster_prepare that finishes last -- a ticket
// behind __threadfence() -- cost 14 us for the fence + ticket of 800 workgroups and 26 us for this function reading the
// other workgroups' tables with device-scope loads: the eight XCDs have their own L2s, so device-scope ordering inside a
// kernel means write-backs and L2 bypasses; a kernel boundary is cheaper than that.)
#define R_FCLS 16            // tile classes of a body kept in registers between the histogram and the placement
#define RLISTS 1024
template <int NT>
__device__ __forceinline__ void r_finalize_lists(const RasterP& p) {
  __shared__ int f_hist[R_NCLS], f_cur[R_NCLS], f_ghist[64], f_gcur[64];
  __shared__ int f_stale, f_defer;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cap = r_cap(p);
  if (tid == 0) { f_stale = 0; f_defer = 0; }
  if (tid < R_NCLS) { f_hist[tid] = 0; f_ghist[tid] = 0; }
  __syncthreads();
  auto unit_class = [&](int rem) { return 1 + (31 - min(31, rem * 32 / RG_UNIT)); };
  const int nchunk = (p.B + NT - 1) / NT;
  // per-thread body of a chunk: window size, tile count, the first R_FCLS tile classes (all loads issued together)
  int ns = 0, cls[R_FCLS];
  long long v = 0;
  auto load_body = [&](int b) {
    int ww = 0, wh = 0, st = 0;
    ns = 0;
    if (b < p.B) { ww = p.win[b * 4 + 2]; wh = p.win[b * 4 + 3]; ns = p.body_ns[b]; st = p.stale[b]; }
#pragma unroll
    for (int k = 0; k < R_FCLS; ++k) cls[k] = (b < p.B && k < ns) ? p.strip_cls[(size_t)b * cap + k] : 0;
    v = (ww > 0 && wh > 0) ? (long long)ww * wh : 0ll;
    return st;
  };
  for (int ch = 0; ch < nchunk; ++ch) {
    const int b = ch * NT + tid;
    const int st = load_body(b);
    // ---- class histograms ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < R_FCLS; ++k)
      if (k < ns) atomicAdd(&f_hist[cls[k]], 1);
    for (int k = R_FCLS; k < ns; ++k) atomicAdd(&f_hist[p.strip_cls[(size_t)b * cap + k]], 1);
    const long long nfull = v / RG_UNIT;
    const int rem = (int)(v % RG_UNIT);
    if (nfull) atomicAdd(&f_ghist[0], (int)min(nfull, (long long)p.max_units));
    if (rem) atomicAdd(&f_ghist[unit_class(rem)], 1);
    if (st) atomicAdd(&f_stale, 1);
    if (st == 2) atomicAdd(&f_defer, 1);
    if (b < p.B) { p.ns_listed[b] = ns; p.nu_listed[b] = (int)min(nfull, (long long)p.max_units) + (rem ? 1 : 0); }
  }
  __syncthreads();
  // ---- class offsets: exclusive scans of the two histograms by the first two waves --------------------------------------
  if (wave < 2) {
    int* hist = wave == 0 ? f_hist : f_ghist;
    int* cur = wave == 0 ? f_cur : f_gcur;
    const int h = hist[lane];
    const int incl = mh_wave_scan_add(h);
    cur[lane] = incl - h;
    if (lane == 63) {
      if (wave == 0) {
        p.total[0] = min(incl, p.max_strips);
        p.sort_count[0] += (unsigned long long)p.B;
        p.sort_count[1] += (unsigned long long)f_stale;
        p.sort_count[2] += (unsigned long long)f_defer;     // ... of which beside the gradient kernel (r_deferred_sorts)
      } else {
        p.gunit_total[0] = min(incl, p.max_units);
      }
    }
  }
  __syncthreads();
  // ---- placement (up to NT bodies: still in registers) ---------------------------------------------------------------------
  for (int ch = 0; ch < nchunk; ++ch) {
    const int b = ch * NT + tid;
    if (nchunk > 1) load_body(b);
    if (b >= p.B) continue;
#pragma unroll
    for (int k = 0; k < R_FCLS; ++k)
      if (k < ns) {
        const int pos = atomicAdd(&f_cur[cls[k]], 1);
        if (pos < p.max_strips) p.strip_order[pos] = b * cap + k;
      }
    for (int k = R_FCLS; k < ns; ++k) {
      const int pos = atomicAdd(&f_cur[p.strip_cls[(size_t)b * cap + k]], 1);
      if (pos < p.max_strips) p.strip_order[pos] = b * cap + k;
    }
    const long long nfull = v / RG_UNIT;
    const int rem = (int)(v % RG_UNIT);
    if (nfull) {
      const int n = (int)min(nfull, (long long)p.max_units);
      const int pos = atomicAdd(&f_gcur[0], n);
      for (int k = 0; k < n; ++k)
        if (pos + k < p.max_units) p.gunit_list[pos + k] = ((unsigned long long)(unsigned)b << 32) | (unsigned)k;
    }
    if (rem) {
      const int pos = atomicAdd(&f_gcur[unit_class(rem)], 1);
      if (pos < p.max_units) p.gunit_list[pos] = ((unsigned long long)(unsigned)b << 32) | (unsigned)nfull;
    }
  }
}

