// mh_model: immutable SMPL constants, re-laid for the gfx950 kernels and uploaded once.
// Replaces SMPL.__init__ (reference smpl.py:124-275).
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "mh_common.h"

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 1; }

// ---- measurement aid --------------------------------------------------------------------------------------
static bool g_prof_on = false;
static int g_prof_level = 0;
static hipEvent_t g_prof_ev[MH_PROF_COUNT][2];
static bool g_prof_have[MH_PROF_COUNT];
static bool g_prof_init = false;

extern "C" int mh_profile_enable(int on) {
  if (on && !g_prof_init) {
    for (int i = 0; i < MH_PROF_COUNT; ++i)
      for (int e = 0; e < 2; ++e) MH_HIP(hipEventCreate(&g_prof_ev[i][e]));
    g_prof_init = true;
  }
  for (int i = 0; i < MH_PROF_COUNT; ++i) g_prof_have[i] = false;
  g_prof_on = on != 0;
  g_prof_level = on;
  return MH_OK;
}

bool mh_prof_on() { return g_prof_on; }
int mh_prof_level() { return g_prof_level; }
void mh_prof_mark(int which, int edge, hipStream_t st) {
  if (!g_prof_on || which < 0 || which >= MH_PROF_COUNT) return;
  (void)hipEventRecord(g_prof_ev[which][edge], st);
  if (edge == 1) g_prof_have[which] = true;
}

extern "C" int mh_profile_read(int which, float* ms) {
  MH_CHECK(ms && which >= 0 && which < MH_PROF_COUNT, "bad profile slot");
  MH_CHECK(g_prof_init && g_prof_have[which], "no launch recorded for this slot");
  MH_HIP(hipEventSynchronize(g_prof_ev[which][1]));
  MH_HIP(hipEventElapsedTime(ms, g_prof_ev[which][0], g_prof_ev[which][1]));
  return MH_OK;
}
extern "C" int mh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// host-side table building over independent index ranges: the layouts below touch 4.6 M entries each, 110 ms on one
// core, which was a sixth of a whole 250-cycle fit
template <typename F>
static void parallel_for(int n, F fn) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
  nt = std::max(1, std::min(nt, n));
  if (nt == 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=]() {
      for (int i = t; i < n; i += nt) fn(i);
    });
  for (auto& x : th) x.join();
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  MH_HIP(hipMalloc((void**)dst, bytes));
  if (!src.empty()) MH_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return MH_OK;
}

static int build_regressor(mh_regressor* r, const float* dense, int J, int V) {
  r->J = 0;
  r->nnz = 0;
  r->ptr = nullptr;
  r->vidx = nullptr;
  r->w = nullptr;
  r->rowsum = nullptr;
  if (!dense) return MH_OK;
  std::vector<int> ptr(J + 1, 0), vidx;
  std::vector<float> w, rs(J, 0.f);
  for (int j = 0; j < J; ++j) {
    double s = 0;
    for (int v = 0; v < V; ++v) {
      float x = dense[(size_t)j * V + v];
      if (x != 0.f) {
        vidx.push_back(v);
        w.push_back(x);
        s += x;
      }
    }
    ptr[j + 1] = (int)vidx.size();
    rs[j] = (float)s;
  }
  r->J = J;
  r->nnz = (int)vidx.size();
  int rc;
  if ((rc = upload(&r->ptr, ptr))) return rc;
  if ((rc = upload(&r->vidx, vidx))) return rc;
  if ((rc = upload(&r->w, w))) return rc;
  if ((rc = upload(&r->rowsum, rs))) return rc;
  return MH_OK;
}

extern "C" int mh_model_create(mh_model** out, const mh_model_host* h) {
  MH_CHECK(out && h, "null argument");
  MH_CHECK(h->num_verts > 0 && h->num_faces > 0, "empty model");
  MH_CHECK(h->v_template && h->shapedirs && h->posedirs && h->J_regressor && h->lbs_weights && h->parents && h->faces,
           "missing model array");
  if (mh_device_count() <= 0) {
    mh_set_error("no HIP device visible: the MI355X path cannot run here");
    return MH_ERR_NO_DEVICE;
  }
  const int V = h->num_verts, F = h->num_faces;
  const int VP = ((V + 31) / 32) * 32;
  mh_model* m = new mh_model();
  memset(m, 0, sizeof(*m));
  m->V = V;
  m->VP = VP;
  m->F = F;

  // kinematic tree (smpl.py:270-272): parents must precede children
  for (int j = 0; j < MH_NJ; ++j) {
    int p = h->parents[j];
    if (j == 0) p = -1;
    if (j > 0 && (p < 0 || p >= j)) {
      delete m;
      mh_set_error("parents[%d]=%d: parents must precede children", j, p);
      return MH_ERR_INVALID;
    }
    m->tree.parent[j] = p;
    m->tree.level[j] = (j == 0) ? 0 : m->tree.level[p] + 1;
    m->tree.maxlevel = std::max(m->tree.maxlevel, m->tree.level[j]);
  }

  int rc = MH_OK;
  {
    std::vector<float> vt((size_t)VP * 3, 0.f);
    memcpy(vt.data(), h->v_template, (size_t)V * 3 * sizeof(float));
    if ((rc = upload(&m->vt, vt))) return rc;
  }
  {
    // basis, k<10: shapedirs[v][c][k]; 10<=k<217: posedirs[v][c][k-10], in two layouts:
    //   D  (forward MFMA B operand)  [tile = v/32][kg = k/16][c][lane = (k%2)*32 + v%32][u = (k%16)/2]: the eight
    //      values one lane feeds into the eight two-k MFMA steps of a k-group are 32 contiguous bytes
    //   Dt (backward MFMA B operand) [c][v][li = k%16][t = k/16, padded to 16]: the 14 column-tile values one lane
    //      (body li, vertex) feeds into the feature-gradient MFMAs are 64 contiguous bytes
    std::vector<float> D((size_t)3 * MH_KD * VP, 0.f), Dt((size_t)3 * VP * 256, 0.f);
    auto tidx = [&](int c, int k, int v) { return (((size_t)c * VP + v) * 16 + (k & 15)) * 16 + (k >> 4); };
    auto didx = [&](int c, int k, int v) {
      const int tile = v >> 5, li = v & 31, kg = k >> 4, lh = k & 1, u = (k & 15) >> 1;
      return ((((size_t)tile * (MH_KD / 16) + kg) * 3 + c) * 64 + (lh * 32 + li)) * 8 + u;
    };
    parallel_for(V, [&](int v) {
      for (int c = 0; c < 3; ++c) {
        for (int k = 0; k < MH_NUM_BETAS; ++k) {
          float x = h->shapedirs[((size_t)v * 3 + c) * MH_NUM_BETAS + k];
          D[didx(c, k, v)] = x;
          Dt[tidx(c, k, v)] = x;
        }
        for (int k = 0; k < MH_NUM_POSE_BASIS; ++k) {
          float x = h->posedirs[((size_t)v * 3 + c) * MH_NUM_POSE_BASIS + k];
          D[didx(c, 10 + k, v)] = x;
          Dt[tidx(c, 10 + k, v)] = x;
        }
      }
    });
    if ((rc = upload(&m->D, D))) return rc;
    if ((rc = upload(&m->Dt, Dt))) return rc;
    // split-fp16 forward operand: the basis scaled by 2^shift so that its largest entry lies in [2^12, 2^13) (the
    // low terms of all but vanishing entries stay normal fp16 numbers), as hi = fp16(x), lo = fp16(x - hi)
    float dmax = 0.f;
    for (float x : D) dmax = std::max(dmax, std::fabs(x));
    int shift = 0;
    if (dmax > 0.f) {
      int e;
      (void)std::frexp(dmax, &e);        // dmax = f * 2^e, f in [0.5, 1)
      shift = 13 - e;
    }
    m->d16_shift = shift;
    std::vector<uint16_t> D16((size_t)(VP / 32) * (MH_KD / 16) * 3 * 2 * 64 * 8, 0);
    auto bits16 = [](_Float16 hv) { uint16_t u; memcpy(&u, &hv, 2); return u; };
    auto load = [&](int c, int k, int v) -> float {
      if (v >= V) return 0.f;
      if (k < MH_NUM_BETAS) return h->shapedirs[((size_t)v * 3 + c) * MH_NUM_BETAS + k];
      if (k < MH_NUM_BETAS + MH_NUM_POSE_BASIS) return h->posedirs[((size_t)v * 3 + c) * MH_NUM_POSE_BASIS + (k - MH_NUM_BETAS)];
      return 0.f;
    };
    parallel_for(VP / 32, [&](int tile) {
      for (int s16 = 0; s16 < MH_KD / 16; ++s16)
        for (int c = 0; c < 3; ++c)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 8; ++t) {
              const int v = tile * 32 + (lane & 31), k = 16 * s16 + 8 * (lane >> 5) + t;
              const float x = std::ldexp(load(c, k, v), shift);
              const _Float16 hi = (_Float16)x;
              const _Float16 lo = (_Float16)(x - (float)hi);
              const size_t base = ((((size_t)tile * (MH_KD / 16) + s16) * 3 + c) * 2) * 64 * 8;
              D16[base + (size_t)lane * 8 + t] = bits16(hi);
              D16[base + 64 * 8 + (size_t)lane * 8 + t] = bits16(lo);
            }
    });
    if ((rc = upload(&m->D16, D16))) return rc;
    // split-bf16 backward operands (round to nearest even; hi = bf16(x), lo = bf16(x - hi))
    auto f2bf = [](float x) -> uint16_t {
      uint32_t u;
      memcpy(&u, &x, 4);
      u += 0x7fffu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto bf2f = [](uint16_t hb) -> float {
      const uint32_t u = (uint32_t)hb << 16;
      float f;
      memcpy(&f, &u, 4);
      return f;
    };
    const int NBLK = VP / 16;
    std::vector<uint16_t> Dt16((size_t)NBLK * 3 * 7 * 2 * 64 * 8, 0), W16((size_t)NBLK * 2 * 64 * 8, 0);
    parallel_for(NBLK, [&](int blk) {
      for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < 8; ++t) {
          const int v = blk * 16 + 8 * (lane >> 5) + t, n = lane & 31;
          for (int c = 0; c < 3; ++c)
            for (int ct = 0; ct < 7; ++ct) {
              const float x = load(c, ct * 32 + n, v);
              const uint16_t hi = f2bf(x), lo = f2bf(x - bf2f(hi));
              const size_t base = ((((size_t)blk * 3 + c) * 7 + ct) * 2) * 64 * 8;
              Dt16[base + (size_t)lane * 8 + t] = hi;
              Dt16[base + 64 * 8 + (size_t)lane * 8 + t] = lo;
            }
        }
    });
    // W16 [VP/16][term 2][lane][8]: B operand of v_mfma_f32_32x32x16_bf16, lane l = (joint l&31, vertex half l>>5)
    // holds vertices 16 blk + 8 (l>>5) + 0..7
    for (int blk = 0; blk < NBLK; ++blk)
      for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < 8; ++t) {
          const int v = blk * 16 + 8 * (lane >> 5) + t, n = lane & 31;
          const float w = (v < V && n < MH_NJ) ? h->lbs_weights[(size_t)v * MH_NJ + n] : 0.f;
          const uint16_t hi = f2bf(w), lo = f2bf(w - bf2f(hi));
          W16[((size_t)blk * 2) * 64 * 8 + (size_t)lane * 8 + t] = hi;
          W16[((size_t)blk * 2 + 1) * 64 * 8 + (size_t)lane * 8 + t] = lo;
        }
    if ((rc = upload(&m->Dt16, Dt16))) return rc;
    if ((rc = upload(&m->W16, W16))) return rc;
  }
  {
    // skinning weights: keep the non-zeros per vertex (<= 4 in SMPL); nw = max count
    int nw = 1;
    for (int v = 0; v < V; ++v) {
      int c = 0;
      for (int j = 0; j < MH_NJ; ++j) c += h->lbs_weights[(size_t)v * MH_NJ + j] != 0.f;
      nw = std::max(nw, c);
    }
    if (nw > 4 && nw < 8) nw = 8;
    if (nw > 8) nw = MH_NJ;
    m->nw = nw;
    std::vector<int> idx((size_t)VP * nw, 0);
    std::vector<float> w((size_t)VP * nw, 0.f);
    for (int v = 0; v < V; ++v) {
      int c = 0;
      for (int j = 0; j < MH_NJ; ++j) {
        float x = h->lbs_weights[(size_t)v * MH_NJ + j];
        if (x != 0.f) {
          idx[(size_t)v * nw + c] = j;
          w[(size_t)v * nw + c] = x;
          ++c;
        }
      }
    }
    if ((rc = upload(&m->skidx, idx))) return rc;
    if ((rc = upload(&m->skw, w))) return rc;
  }
  {
    // J = Jt + JS.beta  ==  J_regressor.(v_template + shapedirs.beta)  (smpl.py:532-535), hoisted
    std::vector<float> Jt(MH_NJ * 3), JS(MH_NJ * 3 * MH_NUM_BETAS);
    for (int j = 0; j < MH_NJ; ++j)
      for (int c = 0; c < 3; ++c) {
        double a = 0;
        std::vector<double> s(MH_NUM_BETAS, 0.0);
        for (int v = 0; v < V; ++v) {
          double r = h->J_regressor[(size_t)j * V + v];
          if (r == 0.0) continue;
          a += r * h->v_template[(size_t)v * 3 + c];
          for (int l = 0; l < MH_NUM_BETAS; ++l) s[l] += r * h->shapedirs[((size_t)v * 3 + c) * MH_NUM_BETAS + l];
        }
        Jt[j * 3 + c] = (float)a;
        for (int l = 0; l < MH_NUM_BETAS; ++l) JS[(j * 3 + c) * MH_NUM_BETAS + l] = (float)s[l];
      }
    if ((rc = upload(&m->Jt, Jt))) return rc;
    if ((rc = upload(&m->JS, JS))) return rc;
  }
  {
    std::vector<int> f(h->faces, h->faces + (size_t)F * 3);
    for (int x : f)
      if (x < 0 || x >= V) {
        mh_set_error("face index out of range");
        return MH_ERR_INVALID;
      }
    if ((rc = upload(&m->faces, f))) return rc;
  }
  if ((rc = build_regressor(&m->reg[MH_REG_ALPHAPOSE], h->reg_alphapose, MH_NKP, V))) return rc;
  if ((rc = build_regressor(&m->reg[MH_REG_H36M17], h->reg_h36m17, 17, V))) return rc;
  if ((rc = build_regressor(&m->reg[MH_REG_MUPOTS], h->reg_mupots, 17, V))) return rc;
  if ((rc = build_regressor(&m->reg[MH_REG_EXTRA9], h->reg_extra9, 9, V))) return rc;
  {
    std::vector<int> ptr(VP + 1, 0), js, head((size_t)VP * 4, 0);
    std::vector<float> ws;
    for (int v = 0; v < VP; ++v) {
      if (h->reg_alphapose && v < V)
        for (int j = 0; j < MH_NKP; ++j) {
          float x = h->reg_alphapose[(size_t)j * V + v];
          if (x != 0.f) {
            js.push_back(j);
            ws.push_back(x);
          }
        }
      ptr[v + 1] = (int)js.size();
      // head row: (first entry index, entry count, first joint, first weight bits) -- one 16-byte load gives the
      // backward everything it needs for the common single-entry vertex
      head[(size_t)v * 4] = ptr[v];
      head[(size_t)v * 4 + 1] = ptr[v + 1] - ptr[v];
      if (ptr[v + 1] > ptr[v]) {
        head[(size_t)v * 4 + 2] = js[ptr[v]];
        memcpy(&head[(size_t)v * 4 + 3], &ws[ptr[v]], 4);
      }
    }
    if ((rc = upload(&m->kpv_head, head))) return rc;
    if ((rc = upload(&m->kpv_ptr, ptr))) return rc;
    if ((rc = upload(&m->kpv_j, js))) return rc;
    if ((rc = upload(&m->kpv_w, ws))) return rc;
  }
  if ((rc = mh_kp_build(m, h))) return rc;
  *out = m;
  return MH_OK;
}

extern "C" int mh_model_destroy(mh_model* m) {
  if (!m) return MH_OK;
  void* ptrs[] = {m->vt, m->D, m->Dt, m->D16, m->Dt16, m->W16, m->skidx, m->skw, m->Jt, m->JS, m->faces, m->kpv_ptr, m->kpv_j, m->kpv_w, m->kpv_head};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  mh_kp_free(m);
  for (int i = 0; i < 4; ++i) {
    void* q[] = {m->reg[i].ptr, m->reg[i].vidx, m->reg[i].w, m->reg[i].rowsum};
    for (void* p : q)
      if (p) (void)hipFree(p);
  }
  delete m;
  return MH_OK;
}

extern "C" const int32_t* mh_model_faces(const mh_model* m) { return m ? m->faces : nullptr; }
