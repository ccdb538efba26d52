// Streaming (HBM-bound) kernels of the optimisation loop: key-point projection + 2D residual,
// RMSprop / Adam updates, one-euro scan, temporal terms.
#include "mh_common.h"

// =============================================================================================
// a11/a12: projection of the 17 key-points + masked 2D residual and its gradient
// =============================================================================================
struct ProjP {
  int B, mode, has_kd;
  float K[9];
  float Kd[5];
  float jw[MH_NKP];   // per-key-point weights (optimizer.py:75-130, normalised to mean 1 by the caller)
  float thr, w, h, coef;
  const float* joints;
  const float* pose2d;
  float* uv;
  float* gj;
  float* loss;
  // warm-up form: joints = 1.1^xscale[b%NB] * local + transl, gtransl = sum_j gj
  int NB;
  const float* local;
  const float* xscale;
  const float* transl;
  float* gtransl;
};

__global__ __launch_bounds__(64) void k_project_loss(ProjP p) {
  const int b = blockIdx.x, j = threadIdx.x;
  float l = 0.f;
  float g3[3] = {0.f, 0.f, 0.f};
  if (j < MH_NKP) {
    const size_t o = (size_t)b * MH_NKP + j;
    float X, Y, Z;
    if (p.local) {                                           // optimizer.py:749-750
      const float sc = p.xscale ? powf(1.1f, p.xscale[b % p.NB]) : 1.f;
      X = sc * p.local[o * 3] + p.transl[(size_t)b * 3];
      Y = sc * p.local[o * 3 + 1] + p.transl[(size_t)b * 3 + 1];
      Z = sc * p.local[o * 3 + 2] + p.transl[(size_t)b * 3 + 2];
    } else {
      X = p.joints[o * 3]; Y = p.joints[o * 3 + 1]; Z = p.joints[o * 3 + 2];
    }
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
    if (p.uv) {
      p.uv[o * 2] = u;
      p.uv[o * 2 + 1] = v;
    }
    const float conf = p.pose2d[o * 3 + 2];
    float gu, gvv;
    if (p.mode == 0) {   // optimizer.py:364-368, 404, 419-420
      const float c = conf >= p.thr ? p.jw[j] : 0.f;   // mask = pose_weights * (conf >= thr)  (:404, 419-420)
      const float du = c * u / p.w - c * p.pose2d[o * 3] / p.w;
      const float dv = c * v / p.h - c * p.pose2d[o * 3 + 1] / p.h;
      l = du * du + dv * dv;
      gu = 2 * du * c / p.w;
      gvv = 2 * dv * c / p.h;
    } else {             // optimizer.py:735, 754-756 (mean over B*17*2 elements, pixels)
      const float c = conf > p.thr ? p.jw[j] : 0.f;    // w = pose_weights * vis  (:753)
      const float du = c * u - c * p.pose2d[o * 3];
      const float dv = c * v - c * p.pose2d[o * 3 + 1];
      const float inv = 1.f / ((float)p.B * MH_NKP * 2);
      l = (du * du + dv * dv) * inv;
      gu = 2 * du * c * inv;
      gvv = 2 * dv * c * inv;
    }
    gu *= p.coef;
    gvv *= p.coef;
    const float gxx = gu * p.K[0] + gvv * p.K[3], gyy = gu * p.K[1] + gvv * p.K[4];
    const float gx = gxx * dxx_dx + gyy * dyy_dx, gy = gxx * dxx_dy + gyy * dyy_dy;
    g3[0] = gx / Z;
    g3[1] = gy / Z;
    g3[2] = -(gx * x + gy * y) / Z;
    if (p.gj) {
      p.gj[o * 3] = g3[0];
      p.gj[o * 3 + 1] = g3[1];
      p.gj[o * 3 + 2] = g3[2];
    }
  }
  l = mh_wave_sum(l);
  if (j == 0) p.loss[b] = l;
  if (p.gtransl) {
    const float t0 = mh_wave_sum(g3[0]), t1 = mh_wave_sum(g3[1]), t2 = mh_wave_sum(g3[2]);
    if (j == 0) {
      p.gtransl[(size_t)b * 3] = t0;
      p.gtransl[(size_t)b * 3 + 1] = t1;
      p.gtransl[(size_t)b * 3 + 2] = t2;
    }
  }
}

extern "C" int mh_project_joints_loss_w(int B, const float* joints, const float* K_host, const float* Kd_host,
                                        const float* joint_w_host, const float* pose2d, float thr, int mode, float img_w,
                                        float img_h, float coef, float* uv, float* gjoints, float* loss, void* stream) {
  MH_CHECK(joints && K_host && pose2d && gjoints && loss, "null argument");
  MH_CHECK(B > 0, "B must be positive");
  ProjP p;
  for (int i = 0; i < MH_NKP; ++i) p.jw[i] = joint_w_host ? joint_w_host[i] : 1.f;
  p.B = B; p.mode = mode; p.has_kd = Kd_host != nullptr;
  for (int i = 0; i < 9; ++i) p.K[i] = K_host[i];
  for (int i = 0; i < 5; ++i) p.Kd[i] = Kd_host ? Kd_host[i] : 0.f;
  p.thr = thr; p.w = img_w; p.h = img_h; p.coef = coef;
  p.joints = joints; p.pose2d = pose2d; p.uv = uv; p.gj = gjoints; p.loss = loss;
  p.NB = 1; p.local = nullptr; p.xscale = nullptr; p.transl = nullptr; p.gtransl = nullptr;
  hipLaunchKernelGGL(k_project_loss, dim3(B), dim3(64), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_project_joints_loss(int B, const float* joints, const float* K_host, const float* Kd_host,
                                      const float* pose2d, float thr, int mode, float img_w, float img_h, float coef,
                                      float* uv, float* gjoints, float* loss, void* stream) {
  return mh_project_joints_loss_w(B, joints, K_host, Kd_host, nullptr, pose2d, thr, mode, img_w, img_h, coef, uv, gjoints,
                                  loss, stream);
}

extern "C" int mh_warmup_project_w(int B, int NB, const float* local_joints, const float* xscale, const float* transl,
                                   const float* K_host, const float* Kd_host, const float* joint_w_host,
                                   const float* pose2d, float thr, float coef, float* gtransl, float* loss, void* stream) {
  MH_CHECK(local_joints && transl && K_host && pose2d && gtransl && loss, "null argument");
  MH_CHECK(B > 0 && NB > 0, "B and NB must be positive");
  ProjP p;
  for (int i = 0; i < MH_NKP; ++i) p.jw[i] = joint_w_host ? joint_w_host[i] : 1.f;
  p.B = B; p.mode = 1; p.has_kd = Kd_host != nullptr;
  for (int i = 0; i < 9; ++i) p.K[i] = K_host[i];
  for (int i = 0; i < 5; ++i) p.Kd[i] = Kd_host ? Kd_host[i] : 0.f;
  p.thr = thr; p.w = 1.f; p.h = 1.f; p.coef = coef;
  p.joints = nullptr; p.pose2d = pose2d; p.uv = nullptr; p.gj = nullptr; p.loss = loss;
  p.NB = NB; p.local = local_joints; p.xscale = xscale; p.transl = transl; p.gtransl = gtransl;
  hipLaunchKernelGGL(k_project_loss, dim3(B), dim3(64), 0, (hipStream_t)stream, p);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_warmup_project(int B, int NB, const float* local_joints, const float* xscale, const float* transl,
                                 const float* K_host, const float* Kd_host, const float* pose2d, float thr, float coef,
                                 float* gtransl, float* loss, void* stream) {
  return mh_warmup_project_w(B, NB, local_joints, xscale, transl, K_host, Kd_host, nullptr, pose2d, thr, coef, gtransl,
                             loss, stream);
}

// =============================================================================================
// a20: optimiser updates
// =============================================================================================
__global__ void k_rmsprop(float* p, const float* g, float* sq, float* buf, size_t n, float lr, float alpha,
                          float mom, float eps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;   // square_avg.mul_(alpha).addcmul_(g, g, 1-alpha)
    sq[i] = s;
    const float b = mom * buf[i] + gi / (sqrtf(s) + eps);      // buf.mul_(momentum).addcdiv_(g, avg)
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}

// the same update and, riding along in its first workgroup, the cycle's log entries copied from the staging row the captured
// graph wrote them to into their row of the log (one small launch less behind every graph replay).
// Round 6 (mh_rmsprop_step_person): the launch also carries the per-person reduction of the LBS backward -- the sums over the
// frames of the per-body shape / scale gradients, k_person_reduce's job, 4 us + a 5-us gap at the end of the cycle's chain --
// in NB * (nbeta + 1) extra workgroups, one per shared element: the same strided sum and tree as k_person_reduce (same bits),
// added to the element's gradient, then the element's update.  The element-wise workgroups leave those elements alone.
struct PersonP {
  const float* gbeta_b;
  const float* gxs_b;
  int B, NB, nbeta, on;
  long long off_betas, off_xscale;
};
__global__ __launch_bounds__(256) void k_rmsprop_log(float* p, float* g, float* sq, float* buf, size_t n, float lr, float alpha,
                              float mom, float eps, const float* log_src, float* log_dst, int nlog, int* poke_dst, int npoke,
                              int poke0, int poke1, int nmain, PersonP ps) {
  if ((int)blockIdx.x >= nmain) {
    __shared__ float s[256];
    const int e = (int)blockIdx.x - nmain, pn = e % ps.NB, q = e / ps.NB;     // q < nbeta: shape component, == nbeta: scale
    float a = 0;
    for (int b = pn + threadIdx.x * ps.NB; b < ps.B; b += 256 * ps.NB) a += (q < ps.nbeta) ? ps.gbeta_b[(size_t)b * ps.nbeta + q] : ps.gxs_b[b];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const size_t i = (size_t)(q < ps.nbeta ? ps.off_betas + (long long)pn * ps.nbeta + q : ps.off_xscale + pn);
      const float gi = g[i] + s[0];
      g[i] = gi;
      const float sv = alpha * sq[i] + (1.f - alpha) * gi * gi;
      sq[i] = sv;
      const float bv = mom * buf[i] + gi / (sqrtf(sv) + eps);
      buf[i] = bv;
      p[i] = p[i] - lr * bv;
    }
    return;
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < nlog; i += blockDim.x) log_dst[i] = log_src[i];
    // (mh_rmsprop_step_log_poke) up to two device-resident switch words of the NEXT captured cycle, set in this launch
    if (threadIdx.x == 0 && npoke > 0) { poke_dst[0] = poke0; if (npoke > 1) poke_dst[1] = poke1; }
  }
  const long long b0 = ps.on ? ps.off_betas : -1, b1 = ps.on ? b0 + (long long)ps.NB * ps.nbeta : -1;
  const bool xs = ps.on && ps.off_xscale >= 0;
  const long long x0 = xs ? ps.off_xscale : -1, x1 = xs ? x0 + ps.NB : -1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)nmain * blockDim.x) {
    if (ps.on && (((long long)i >= b0 && (long long)i < b1) || ((long long)i >= x0 && (long long)i < x1))) continue;
    const float gi = g[i];
    const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;
    sq[i] = s;
    const float b = mom * buf[i] + gi / (sqrtf(s) + eps);
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}

static int rmsprop_log_launch(float* params, float* grads, float* square_avg, float* momentum_buf, size_t n,
                              float lr, float alpha, float momentum, float eps, const float* log_src, float* log_dst,
                              int nlog, int32_t* poke_dst, int npoke, int32_t poke0, int32_t poke1, const mh_person_sums* person,
                              void* stream) {
  MH_CHECK(params && grads && square_avg && momentum_buf, "null argument");
  MH_CHECK(nlog == 0 || (log_src && log_dst && nlog > 0), "log_src / log_dst / nlog");
  MH_CHECK(npoke >= 0 && npoke <= 2 && (npoke == 0 || poke_dst), "poke_dst / npoke");
  PersonP ps;
  memset(&ps, 0, sizeof(ps));
  int extra = 0;
  if (person) {
    MH_CHECK(person->gbeta_b && person->gxs_b, "null per-body sums");
    MH_CHECK(person->B > 0 && person->NB > 0 && person->nbeta > 0, "B, NB and nbeta must be positive");
    MH_CHECK(person->off_betas >= 0 && (size_t)person->off_betas + (size_t)person->NB * person->nbeta <= n, "betas outside the parameter vector");
    MH_CHECK(person->off_xscale < 0 || (size_t)person->off_xscale + (size_t)person->NB <= n, "xscale outside the parameter vector");
    ps.gbeta_b = person->gbeta_b; ps.gxs_b = person->gxs_b; ps.B = person->B; ps.NB = person->NB; ps.nbeta = person->nbeta;
    ps.on = 1; ps.off_betas = person->off_betas; ps.off_xscale = person->off_xscale;
    extra = person->NB * (person->nbeta + (person->off_xscale >= 0 ? 1 : 0));
  }
  if (n == 0 && nlog == 0 && npoke == 0) return MH_OK;
  const int blocks = n == 0 ? 1 : (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_rmsprop_log, dim3(blocks + extra), dim3(256), 0, (hipStream_t)stream, params, grads, square_avg,
                     momentum_buf, n, lr, alpha, momentum, eps, log_src, log_dst, nlog, (int*)poke_dst, npoke, (int)poke0, (int)poke1,
                     blocks, ps);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_rmsprop_step_person(float* params, float* grads, float* square_avg, float* momentum_buf, size_t n,
                                      float lr, float alpha, float momentum, float eps, const float* log_src, float* log_dst,
                                      int nlog, int32_t* poke_dst, int npoke, int32_t poke0, int32_t poke1,
                                      const mh_person_sums* person, void* stream) {
  MH_CHECK(person, "null argument");
  return rmsprop_log_launch(params, grads, square_avg, momentum_buf, n, lr, alpha, momentum, eps, log_src, log_dst, nlog, poke_dst,
                            npoke, poke0, poke1, person, stream);
}

extern "C" int mh_rmsprop_step_log_poke(float* params, const float* grads, float* square_avg, float* momentum_buf, size_t n,
                                        float lr, float alpha, float momentum, float eps, const float* log_src, float* log_dst,
                                        int nlog, int32_t* poke_dst, int npoke, int32_t poke0, int32_t poke1, void* stream) {
  return rmsprop_log_launch(params, (float*)grads, square_avg, momentum_buf, n, lr, alpha, momentum, eps, log_src, log_dst, nlog,
                            poke_dst, npoke, poke0, poke1, nullptr, stream);
}

extern "C" int mh_rmsprop_step_log(float* params, const float* grads, float* square_avg, float* momentum_buf, size_t n,
                                   float lr, float alpha, float momentum, float eps, const float* log_src, float* log_dst,
                                   int nlog, void* stream) {
  return mh_rmsprop_step_log_poke(params, grads, square_avg, momentum_buf, n, lr, alpha, momentum, eps, log_src, log_dst, nlog,
                                  nullptr, 0, 0, 0, stream);
}

extern "C" int mh_rmsprop_step(float* params, const float* grads, float* square_avg, float* momentum_buf, size_t n,
                               float lr, float alpha, float momentum, float eps, void* stream) {
  MH_CHECK(params && grads && square_avg && momentum_buf, "null argument");
  if (n == 0) return MH_OK;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_rmsprop, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, square_avg,
                     momentum_buf, n, lr, alpha, momentum, eps);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// device-resident learning rate (lets a captured hipGraph replay the step while the ExponentialLR
// schedule of optimizer.py:356 advances): p -= lr[0] * buf, then lr[0] *= gamma in a second launch
__global__ void k_rmsprop_dev(float* p, const float* g, float* sq, float* buf, size_t n, const float* lr_dev, float alpha,
                              float mom, float eps) {
  const float lr = lr_dev[0];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float s = alpha * sq[i] + (1.f - alpha) * gi * gi;
    sq[i] = s;
    const float b = mom * buf[i] + gi / (sqrtf(s) + eps);
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}
__global__ void k_scale_scalar(float* x, float gamma) { x[0] *= gamma; }

extern "C" int mh_rmsprop_step_dev(float* params, const float* grads, float* square_avg, float* momentum_buf, size_t n,
                                   float* lr_dev, float gamma, float alpha, float momentum, float eps, void* stream) {
  MH_CHECK(params && grads && square_avg && momentum_buf && lr_dev, "null argument");
  if (n == 0) return MH_OK;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_rmsprop_dev, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, square_avg, momentum_buf,
                     n, (const float*)lr_dev, alpha, momentum, eps);
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scale_scalar, dim3(1), dim3(1), 0, (hipStream_t)stream, lr_dev, gamma);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

__global__ void k_adam(float* p, const float* g, float* m, float* v, size_t n, float step_size, float b1, float b2,
                       float inv_sqrt_bc2, float eps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

extern "C" int mh_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, int step,
                            float lr, float beta1, float beta2, float eps, void* stream) {
  MH_CHECK(params && grads && exp_avg && exp_avg_sq, "null argument");
  MH_CHECK(step >= 1, "step counts from 1");
  if (n == 0) return MH_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// =============================================================================================
// a19: one-euro filter, sequential in t, parallel over channels
// =============================================================================================
__global__ void k_one_euro(const float* x, float* y, int T, size_t E, float min_cutoff, float beta, float frame_rate,
                           int i0, float t0, const float* xprev_in, const float* dxprev_in, float* dxprev_out) {
  const float two_pi = 6.283185307179586f;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < E; e += (size_t)gridDim.x * blockDim.x) {
    float xp, dxp, tp, ti;
    int first;
    if (xprev_in) {            // continuing a sequence whose frames [0, i0) live on another rank
      xp = xprev_in[e];
      dxp = dxprev_in[e];
      tp = ti = t0;
      first = 0;
    } else {
      xp = x[e];
      dxp = 0.f;
      tp = ti = 0.f;
      y[e] = xp;
      first = 1;
    }
    // the frames of a channel are a dependent chain of ~15 operations each; the loads are not: eight frames' values are
    // asked for before the chain walks them (the chain alone is ~12 us of the kernel at T = 200)
    int k = first;
    for (; k + 8 <= T; k += 8) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(size_t)(k + u) * E + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma clang fp contract(off)
        const int i = i0 + k + u;
        ti = ti + (float)((double)i / (double)frame_rate);
        const float xi = xv[u];
        const float te = ti - tp;
        float r = two_pi * te;
        const float ad = r / (r + 1.f);
        const float dx = (xi - xp) / te;
        const float dxh = ad * dx + (1.f - ad) * dxp;
        const float cutoff = min_cutoff + beta * fabsf(dxh);
        r = (two_pi * cutoff) * te;
        const float a = r / (r + 1.f);
        const float xh = a * xi + (1.f - a) * xp;
        y[(size_t)(k + u) * E + e] = xh;
        xp = xh;
        dxp = dxh;
        tp = ti;
      }
    }
    for (; k < T; ++k) {
#pragma clang fp contract(off)
      const int i = i0 + k;                                   // global frame index
      ti = __fadd_rn(ti, (float)((double)i / (double)frame_rate));     // optimizer.py:671 (float32 running sum)
      const float xi = x[(size_t)k * E + e];
      // numpy evaluates every product/sum separately in float32: no FMA contraction (hip's __fmul_rn / __fadd_rn are plain
      // operators and do not prevent it -- the pragma does)
      const float te = __fsub_rn(ti, tp);
      float r = __fmul_rn(two_pi, te);                        // d_cutoff = 1
      const float ad = __fdiv_rn(r, __fadd_rn(r, 1.f));
      const float dx = __fdiv_rn(__fsub_rn(xi, xp), te);
      const float dxh = __fadd_rn(__fmul_rn(ad, dx), __fmul_rn(__fsub_rn(1.f, ad), dxp));
      const float cutoff = __fadd_rn(min_cutoff, __fmul_rn(beta, fabsf(dxh)));
      r = __fmul_rn(__fmul_rn(two_pi, cutoff), te);
      const float a = __fdiv_rn(r, __fadd_rn(r, 1.f));
      const float xh = __fadd_rn(__fmul_rn(a, xi), __fmul_rn(__fsub_rn(1.f, a), xp));
      y[(size_t)k * E + e] = xh;
      xp = xh;
      dxp = dxh;
      tp = ti;
    }
    if (dxprev_out) dxprev_out[e] = dxp;
  }
}

static int one_euro_launch(const float* x, float* y, int T, size_t E, float min_cutoff, float beta, float frame_rate,
                           int i0, float t0, const float* xprev_in, const float* dxprev_in, float* dxprev_out,
                           void* stream) {
  MH_CHECK(x && y, "null argument");
  MH_CHECK(T >= 1 && E >= 1, "empty input");
  MH_CHECK((xprev_in == nullptr) == (dxprev_in == nullptr), "state arrays come in pairs");
  MH_CHECK(xprev_in || i0 == 0, "a shard that does not start the sequence needs the incoming state");
  const size_t nb = (E + 255) / 256;
  hipLaunchKernelGGL(k_one_euro, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, x, y, T, E,
                     min_cutoff, beta, frame_rate, i0, t0, xprev_in, dxprev_in, dxprev_out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_one_euro_scan(const float* x, float* y, int T, size_t E, float min_cutoff, float beta,
                                float frame_rate, void* stream) {
  return one_euro_launch(x, y, T, E, min_cutoff, beta, frame_rate, 0, 0.f, nullptr, nullptr, nullptr, stream);
}

extern "C" int mh_one_euro_scan_shard(const float* x, float* y, int T, size_t E, float min_cutoff, float beta,
                                      float frame_rate, int first_frame, float time_before, const float* xprev_in,
                                      const float* dxprev_in, float* dxprev_out, void* stream) {
  return one_euro_launch(x, y, T, E, min_cutoff, beta, frame_rate, first_frame, time_before, xprev_in, dxprev_in,
                         dxprev_out, stream);
}

// =============================================================================================
// a18: temporal terms
// =============================================================================================
__global__ __launch_bounds__(256) void k_velocity(int T, int N, const float* pT, const float* prev, const float* next,
                                                  float coef, float* gpT, float* loss_out) {
  // single block: the problem is T*N*3 floats
  __shared__ float s[256];
  const size_t E = (size_t)N * 3, n = (size_t)T * E;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const size_t t = i / E, e = i % E;
    const float c = pT[i];
    float g = 0.f;
    if (t > 0 || prev) {
      const float d = c - (t > 0 ? pT[i - E] : prev[e]);
      acc += d * d;          // the pair (t-1, t) belongs to the owner of t
      g += 2.f * d;
    }
    if (t + 1 < (size_t)T || next) {
      const float d = (t + 1 < (size_t)T ? pT[i + E] : next[e]) - c;
      g -= 2.f * d;
    }
    gpT[i] += coef * g;
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = s[0];
}

extern "C" int mh_velocity_term(int T, int N, const float* pT, const float* prev_halo, const float* next_halo,
                                float coef, float* gpT, float* loss_out, void* stream) {
  MH_CHECK(pT && gpT && loss_out, "null argument");
  MH_CHECK(T >= 1 && N >= 1, "empty input");
  hipLaunchKernelGGL(k_velocity, dim3(1), dim3(256), 0, (hipStream_t)stream, T, N, pT, prev_halo, next_halo, coef, gpT,
                     loss_out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// VEC = float4 (E = N*V*3 divisible by 4 and 16-byte aligned buffers: SMPL with an even N) or float (any E).  OVERWRITE: gverts = term (the caller
// initialises the vertex-gradient buffer with this term instead of clearing it first).
// Sliding window over time: a workgroup owns FV_TB consecutive frames of a slice of the elements and walks through
// them with (t-1, t, t+1) in registers, so every vertex and filtered vertex is read once
// (plus one halo frame per FV_TB) instead of three times.
#ifndef FV_TB
#define FV_TB 8
#endif
// The term's three streams (vertices and filtered vertices in, gradient buffer out: 200 MB at C3) go through the caches as
// non-temporal accesses: the kernel runs beside the rasteriser's selection kernel, whose gathers live on L2 hits -- 7 us of
// the cycle (0.665 -> 0.658 ms same-box, three interleaved runs; the kernel itself unchanged).
#ifndef FV_NT
#define FV_NT 1
#endif
template <typename VEC, bool OVERWRITE>
__global__ __launch_bounds__(256) void k_filtered_verts(int T, size_t E, const float* v, const float* vf, const float* pv,
                                                           const float* pvf, const float* nv, const float* nvf, float coef,
                                                           float* gv, float* partial, const int* live) {
  __shared__ float s[256];
  constexpr int L = sizeof(VEC) / sizeof(float);
  const size_t EV = E / L;
  const int t0 = blockIdx.y * FV_TB, t1 = min(t0 + FV_TB, T);
  float acc = 0.f;
  // gated form (mh_filtered_verts_term_init_gated): while the device-resident switch is 0 the term does not exist yet
  // (optimizer.py:383-392: the filters first run at cycle 50) -- the overwriting form then only clears its rows of the
  // gradient buffer, the adding form does nothing, the sum is 0; the SAME captured launch serves both phases of a fit
  const bool off = live != nullptr && *live == 0;
  if (off) {
    if (OVERWRITE) {
      VEC z = {};
      for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < EV; e += (size_t)gridDim.x * 256)
        for (int t = t0; t < t1; ++t) ((VEC*)(gv + (size_t)t * E))[e] = z;
    }
  } else
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < EV; e += (size_t)gridDim.x * 256) {
#if FV_NT
    typedef float fv_x4 __attribute__((ext_vector_type(4)));
    auto row = [&](const float* base, int t) {
      VEC r;
      if constexpr (sizeof(VEC) == 16) { const fv_x4 q = __builtin_nontemporal_load((const fv_x4*)(base + (size_t)t * E) + e); __builtin_memcpy(&r, &q, 16); }
      else { const float q = __builtin_nontemporal_load(base + (size_t)t * E + e); __builtin_memcpy(&r, &q, 4); }
      return r;
    };
#else
    auto row = [&](const float* base, int t) { return ((const VEC*)(base + (size_t)t * E))[e]; };
#endif
    bool has_p = t0 > 0 || pv != nullptr;
    VEC a = {}, b = {};                                   // frame t-1: vertices, filtered vertices
    if (t0 > 0) { a = row(v, t0 - 1); b = row(vf, t0 - 1); }
    else if (pv) { a = ((const VEC*)pv)[e]; b = ((const VEC*)pvf)[e]; }
    VEC c = row(v, t0), cf = row(vf, t0);
    for (int t = t0; t < t1; ++t) {
      const bool has_n = t + 1 < T || nv != nullptr;
      VEC n = {}, nf = {};
      if (t + 1 < T) { n = row(v, t + 1); nf = row(vf, t + 1); }
      else if (nv) { n = ((const VEC*)nv)[e]; nf = ((const VEC*)nvf)[e]; }
      float gr[L];
#pragma unroll
      for (int k = 0; k < L; ++k) gr[k] = 0.f;
      if (has_p) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
          const float d = (((const float*)&c)[k] - ((const float*)&a)[k]) - (((const float*)&cf)[k] - ((const float*)&b)[k]);
          acc += d * d;          // the pair (t-1, t) belongs to the owner of t
          gr[k] += 2.f * d;
        }
      }
      if (has_n) {
#pragma unroll
        for (int k = 0; k < L; ++k) {
          const float d = (((const float*)&n)[k] - ((const float*)&c)[k]) - (((const float*)&nf)[k] - ((const float*)&cf)[k]);
          gr[k] -= 2.f * d;
        }
      }
      VEC* g = (VEC*)(gv + (size_t)t * E);
      VEC o;
      if (OVERWRITE) {
#pragma unroll
        for (int k = 0; k < L; ++k) ((float*)&o)[k] = coef * gr[k];
      } else {
        o = g[e];
#pragma unroll
        for (int k = 0; k < L; ++k) ((float*)&o)[k] += coef * gr[k];
      }
#if FV_NT
      if constexpr (sizeof(VEC) == 16) { fv_x4 q; __builtin_memcpy(&q, &o, 16); __builtin_nontemporal_store(q, (fv_x4*)g + e); }
      else { float q; __builtin_memcpy(&q, &o, 4); __builtin_nontemporal_store(q, (float*)g + e); }
#else
      g[e] = o;
#endif
      a = c; b = cf; c = n; cf = nf;
      has_p = true;
    }
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void k_sum_partials(const float* partial, int n, float* out) {
  __shared__ float s[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

// per-block partial sums live in a caller-provided workspace (one per engine: captured graphs bake its address; a
// process-global buffer would be shared by every engine, stream and device)
static void fv_grid(int T, size_t E, bool vec, unsigned* gx, unsigned* gy) {
  const size_t EVh = vec ? E / 4 : E;
  *gx = (unsigned)std::min<size_t>((EVh + 255) / 256, 1024);
  *gy = (unsigned)((T + FV_TB - 1) / FV_TB);
}
extern "C" size_t mh_filtered_verts_workspace_bytes(int T, size_t E) {
  unsigned gx, gy;
  fv_grid(T < 1 ? 1 : T, E < 1 ? 1 : E, false, &gx, &gy);      // the scalar form has the larger grid
  return ((size_t)gx * gy * sizeof(float) + 255) & ~(size_t)255;
}

static int filtered_verts_term(int T, size_t E, const float* verts, const float* verts_filt, const float* prev_v,
                               const float* prev_vf, const float* next_v, const float* next_vf, float coef, float* gverts,
                               float* loss_out, bool overwrite, void* ws, void* stream, const int32_t* live = nullptr) {
  MH_CHECK(verts && verts_filt && gverts && loss_out && ws, "null argument");
  float* g_fv_partial = (float*)ws;
  MH_CHECK(T >= 1 && E >= 1, "empty input");
  MH_CHECK((prev_v == nullptr) == (prev_vf == nullptr) && (next_v == nullptr) == (next_vf == nullptr),
           "halo vertices and filtered halo vertices come in pairs");
  hipStream_t st = (hipStream_t)stream;
  auto aligned = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15u) == 0; };
  const bool vec = (E % 4 == 0) && aligned(verts) && aligned(verts_filt) && aligned(prev_v) && aligned(prev_vf) && aligned(next_v) &&
                   aligned(next_vf) && aligned(gverts);
  unsigned gx, gy;
  fv_grid(T, E, vec, &gx, &gy);
  const size_t nblk = (size_t)gx * gy;
  const dim3 grid(gx, gy), blk(256);
#define FV_KERNEL k_filtered_verts
#define FV_LAUNCH(VEC, OW)                                                                                       \
  hipLaunchKernelGGL((FV_KERNEL<VEC, OW>), grid, blk, 0, st, T, E, verts, verts_filt, prev_v, prev_vf, next_v, \
                     next_vf, coef, gverts, g_fv_partial, (const int*)live)
  if (vec) { if (overwrite) FV_LAUNCH(float4, true); else FV_LAUNCH(float4, false); }
  else { if (overwrite) FV_LAUNCH(float, true); else FV_LAUNCH(float, false); }
#undef FV_LAUNCH
#undef FV_KERNEL
  MH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, (const float*)g_fv_partial, (int)nblk, loss_out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_filtered_verts_term(int T, size_t E, const float* verts, const float* verts_filt,
                                      const float* prev_v, const float* prev_vf, const float* next_v,
                                      const float* next_vf, float coef, float* gverts, float* loss_out, void* ws, void* stream) {
  return filtered_verts_term(T, E, verts, verts_filt, prev_v, prev_vf, next_v, next_vf, coef, gverts, loss_out, false, ws, stream);
}

extern "C" int mh_filtered_verts_term_init(int T, size_t E, const float* verts, const float* verts_filt,
                                           const float* prev_v, const float* prev_vf, const float* next_v,
                                           const float* next_vf, float coef, float* gverts, float* loss_out, void* ws, void* stream) {
  return filtered_verts_term(T, E, verts, verts_filt, prev_v, prev_vf, next_v, next_vf, coef, gverts, loss_out, true, ws, stream);
}

extern "C" int mh_filtered_verts_term_init_gated(int T, size_t E, const float* verts, const float* verts_filt,
                                                 const float* prev_v, const float* prev_vf, const float* next_v,
                                                 const float* next_vf, float coef, float* gverts, float* loss_out,
                                                 const int32_t* live_dev, void* ws, void* stream) {
  MH_CHECK(live_dev, "null argument");
  return filtered_verts_term(T, E, verts, verts_filt, prev_v, prev_vf, next_v, next_vf, coef, gverts, loss_out, true, ws, stream,
                             live_dev);
}

// =============================================================================================
// generic forms of transforms.py:57-95 / 114-130 (call compatibility of mhmocap.transforms)
// =============================================================================================
__global__ void k_project_points(int B, int M, const float* pts, const float* K, int has_kd, float k1, float k2, float p1,
                                 float p2, float k3, int with_depth, float* out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)B * M) return;
  const float* Kb = K + (i / M) * 9;
  const float X = pts[i * 3], Y = pts[i * 3 + 1], Z = pts[i * 3 + 2];
  float x = X / Z, y = Y / Z;
  if (has_kd) {
    const float r = x * x + y * y;
    const float rad = 1 + k1 * r + k2 * r * r + k3 * r * r * r;
    const float xx = x * rad + 2 * p1 * x * y + p2 * (r + 2 * x * x);
    const float yy = y * rad + 2 * p2 * y * y + p1 * (r + 2 * y * y);
    x = xx;
    y = yy;
  }
  const int os = with_depth ? 3 : 2;
  out[i * os] = x * Kb[0] + y * Kb[1] + Kb[2];
  out[i * os + 1] = x * Kb[3] + y * Kb[4] + Kb[5];
  if (with_depth) out[i * 3 + 2] = Z;
}

extern "C" int mh_project_points(int B, int M, const float* pts, const float* K_dev, const float* Kd_host, int with_depth,
                                 float* out, void* stream) {
  MH_CHECK(pts && K_dev && out, "null argument");
  MH_CHECK(B > 0 && M > 0, "empty input");
  const float* d = Kd_host;
  hipLaunchKernelGGL(k_project_points, dim3((unsigned)(((size_t)B * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, M,
                     pts, K_dev, d != nullptr, d ? d[0] : 0.f, d ? d[1] : 0.f, d ? d[2] : 0.f, d ? d[3] : 0.f, d ? d[4] : 0.f,
                     with_depth, out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

__global__ void k_unproject_points(int B, int M, const float* uvd, const float* K, float* out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)B * M) return;
  const float* Kb = K + (i / M) * 9;
  // [u - cx, v - cy] . inv(K[:2,:2]^T)
  const float a = Kb[0], b = Kb[3], c = Kb[1], d = Kb[4];     // K[:2,:2]^T = [[a, b], [c, d]]
  const float det = a * d - b * c;
  const float u = uvd[i * 3] - Kb[2], v = uvd[i * 3 + 1] - Kb[5], z = uvd[i * 3 + 2];
  out[i * 3] = z * (u * (d / det) + v * (-c / det));
  out[i * 3 + 1] = z * (u * (-b / det) + v * (a / det));
  out[i * 3 + 2] = z;
}

extern "C" int mh_unproject_points(int B, int M, const float* uvd, const float* K_dev, float* out, void* stream) {
  MH_CHECK(uvd && K_dev && out, "null argument");
  MH_CHECK(B > 0 && M > 0, "empty input");
  hipLaunchKernelGGL(k_unproject_points, dim3((unsigned)(((size_t)B * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, M,
                     uvd, K_dev, out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
