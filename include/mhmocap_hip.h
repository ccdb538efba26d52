/*
 * mhmocap_hip.h -- C ABI of the MI355X-native (gfx950) scene-constrained SMPL optimisation path.
 *
 * The reference (dluvizon/scene-aware-3d-multi-human) has no FFI of its own: its boundary for
 * this path is the Python API of mhmocap/smpl.py and mhmocap/optimizer.py.  This header is the
 * thin C ABI underneath the build's Python mirror of that API
 * (scene-aware-3d-multi-human_amd/mhmocap/{smpl,optimizer,losses,...}.py binds it with ctypes;
 * INTEGRATION.md shows the binding).  Every entry point cites the reference code it replaces.
 *
 * Conventions
 *   - all functions return 0 on success, a negative mh_status otherwise; mh_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - every pointer that is not marked HOST is a device pointer owned by the caller (PyTorch-ROCm
 *     allocations, tensor.data_ptr()); the library never frees or keeps caller memory and
 *     allocates nothing except the immutable model constants inside mh_model;
 *   - all floating point data is fp32 like the reference (smpl.py:134); indices are int32;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work;
 *   - a body is one (frame, person) pair; bodies are stored frame-major: b = t*N + n;
 *     a "person table" of NB rows is addressed as row (b % NB), so NB == N shares shape/scale
 *     across frames (optimizer.py:297,691) and NB == B gives per-body values (smpl.py:357).
 */
#ifndef MHMOCAP_HIP_H
#define MHMOCAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_NUM_JOINTS 24
#define MH_NUM_BETAS 10
#define MH_NUM_POSE_BASIS 207
#define MH_NUM_KP 17          /* sparse key-points of the 2D term (optimizer.py:75) */
#define MH_FEAT_STRIDE 224    /* [beta(10) | R_1..R_23 - I (207) | 0 pad] */

typedef enum mh_status {
  MH_OK = 0,
  MH_ERR_INVALID = -1,   /* bad argument */
  MH_ERR_HIP = -2,       /* a HIP runtime call failed */
  MH_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
  MH_ERR_UNSUPPORTED = -4
} mh_status;

typedef struct mh_model mh_model; /* opaque: immutable SMPL constants resident in HBM */

/* HOST arrays describing an SMPL-shaped model: the fields smpl.py:201-275 reads from the pickle
 * plus the optional joint regressors of smpl.py:234-261 (NULL = absent). */
typedef struct mh_model_host {
  int32_t num_verts;            /* V (6890) */
  int32_t num_faces;            /* F (13776) */
  const float* v_template;      /* (V,3)            smpl.py:211 */
  const float* shapedirs;       /* (V,3,10)         smpl.py:227 */
  const float* posedirs;        /* (V,3,207) as in the pickle; smpl.py:264-267 reshapes it */
  const float* J_regressor;     /* (24,V) dense     smpl.py:231 */
  const float* lbs_weights;     /* (V,24)           smpl.py:274 */
  const int32_t* parents;       /* (24), parents[0] = -1   smpl.py:270-272 */
  const int32_t* faces;         /* (F,3)            smpl.py:205 */
  const float* reg_alphapose;   /* (17,V) or NULL   smpl.py:250 (already transposed)  */
  const float* reg_h36m17;      /* (17,V) or NULL   smpl.py:243 (already re-ordered)  */
  const float* reg_mupots;      /* (17,V) or NULL   smpl.py:257 */
  const float* reg_extra9;      /* (9,V)  or NULL   smpl.py:235 */
} mh_model_host;

/* joint sets of mh_joints_regress */
enum { MH_REG_ALPHAPOSE = 0, MH_REG_H36M17 = 1, MH_REG_MUPOTS = 2, MH_REG_EXTRA9 = 3 };

const char* mh_last_error(void);
int mh_version(void);

/* Measurement aid (bench.py, DESIGN.md "measurement"): when enabled, the library brackets the launches of its
 * dominant kernels with HIP events on the caller's stream; mh_profile_read synchronises on the pair of the most recent
 * launch and returns its duration.  Off by default; not meant to be on during stream capture.  on = 2 additionally
 * switches on the work counters inside the kernels (mh_raster_pair_counters); their stores cost the kernel a few percent. */
enum mh_profile_kernel {
  MH_PROF_RASTER_STRIP = 0, MH_PROF_RASTER_GRADS = 1, MH_PROF_SKIN_FWD = 2, MH_PROF_SKIN_BWD = 3,
  MH_PROF_CONTACT_KNN = 4, MH_PROF_RASTER_SUMS = 5,
  /* the rest of the "LBS + projection" unit of SURVEY 8(d) (round 4) */
  MH_PROF_POSE_FWD = 6, MH_PROF_KEYPOINTS = 7, MH_PROF_POSE_BWD = 8 /* k_pose_bwd + k_person_reduce */,
  MH_PROF_RASTER_PREP = 9 /* k_raster_prepare + k_raster_lists */, MH_PROF_COUNT = 10
};
int mh_profile_enable(int on);
int mh_profile_read(int which, float* ms);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int mh_device_count(void);

/* Streams and the hardware queues behind them (csrc/mh_streams.hip).  The captured cycle (optimizer.py:375-602 as one
 * hipGraph) runs its chain, its side branch and the scene update of optimizer.py:578-584 on three streams; the runtime
 * multiplexes streams onto a few hardware queues, and two of the three on one queue run one after the other.
 * mh_stream_create: a NEW runtime stream (non-blocking), already bound to its queue; mh_streams_share_queue: shared = 1
 * when a and b (NULL = the default stream) drain through the same hardware queue (a spin kernel of spin_us on a, an
 * empty one on b: in-order queues show it); both synchronise.                                                        */
int mh_stream_create(void** out_stream);
int mh_stream_destroy(void* stream);
int mh_streams_share_queue(void* stream_a, void* stream_b, float spin_us, int* shared);
/* one spin kernel (one thread) of ~spin_us on the stream, asynchronous: keeps the stream's hardware queue busy */
int mh_stream_spin(void* stream, float spin_us);
/* shared = 1 when x drains through the hardware queue of ANY of busy[0 .. n), n <= 8 (all spin at once: one probe) */
int mh_stream_shares_any(void* const* busy, int n, void* x, float spin_us, int* shared);
/* the same for m <= 32 candidates in one probe: shared[j] = 1 when cand[j] shares a hardware queue with any busy stream */
int mh_streams_classify(void* const* busy, int n, void* const* cand, int m, float spin_us, int* shared);

/* replaces SMPL.__init__ (smpl.py:124-275): uploads and re-lays the constants once. */
int mh_model_create(mh_model** out, const mh_model_host* host);
int mh_model_destroy(mh_model* m);
/* device pointer to the (F,3) int32 face table (optimizer.py:73-74) */
const int32_t* mh_model_faces(const mh_model* m);

/* ---- a2-a6: batched LBS forward (smpl.py:490-576 + optimizer.py:702-703) -------------------
 * verts[b] = s * LBS(beta[b%NB], pose[b]) + transl[b],  s = 1.1^xscale[b%NB]
 * xscale == NULL -> s = 1; transl == NULL -> 0.
 * vposed (B,V,3) (rest-pose vertices after the blend shapes, kept for the backward) may be NULL.
 * posed_joints (B,24,3) = joints_smpl24 (smpl.py:362) without scale/translation, may be NULL.
 * ws: workspace of mh_lbs_workspace_bytes(B) bytes (keeps per-body joint transforms).       */
size_t mh_lbs_workspace_bytes(int B);
int mh_lbs_forward(const mh_model* m, int B, int NB, const float* betas /*(NB,10)*/,
                   const float* poses /*(B,72)*/, const float* xscale /*(NB) or NULL*/,
                   const float* transl /*(B,3) or NULL*/, float* verts /*(B,V,3)*/,
                   float* vposed /*(B,V,3) or NULL*/, float* posed_joints /*(B,24,3) or NULL*/,
                   void* ws, void* stream);

/* lbs(..., pose2rot=False) (smpl.py:490-491, 553-558): the 24 joint rotations are given as matrices
 * rotmats (B,24,3,3) instead of axis-angle vectors.  mh_lbs_forward_ex: either input form, and v_posed kept for
 * mh_lbs_backward_ex.                                                                                         */
int mh_lbs_forward_ex(const mh_model* m, int B, int NB, const float* betas, const float* poses /*or NULL*/,
                      const float* rotmats /*or NULL*/, const float* xscale, const float* transl, float* verts,
                      float* vposed /*or NULL*/, float* posed_joints /*or NULL*/, void* ws, void* stream);
int mh_lbs_forward_rotmats(const mh_model* m, int B, int NB, const float* betas, const float* rotmats /*(B,24,3,3)*/,
                           const float* xscale, const float* transl, float* verts, float* posed_joints /*or NULL*/,
                           void* ws, void* stream);

/* ---- the "LBS + projection" form of the forward (round 4): what the rasteriser needs of the vertices is produced in the
 * skinning epilogue, where they are in registers -- NDC projection (transforms.py:222-255 with PyTorch3D's R = diag(-1,-1,1);
 * the same three roundings per coordinate as the stand-alone projection pass of mh_raster_terms), the body's screen
 * bounding box, whether a vertex has left the pixel-row band its face lists were sorted for, and the lowest vertex
 * (arg max of y, first index on ties: optimizer.py:487-489).  mh_raster_forward_targets fills the struct for a raster
 * workspace; mh_raster_terms_projected then skips its own pass over the vertices.
 * Box and lowest vertex are found WITHOUT cross-lane reductions: a vertex only reports (one atomic min / max) when it
 * lies beyond the previous launch's extreme minus `slack`; an extreme nobody reported is left at its reset value and the
 * consumer falls back to a scan of that body (mh_raster_terms_projected / mh_lowest_resolve), so any content of the
 * previous-launch arrays gives exact results -- it only decides how many vertices report.                     */
typedef struct mh_fwd_proj {
  float s, w1, h1;             /* x_ndc = s * (-x) / z + w1, y_ndc = s * (-y) / z + h1                      */
  float ra, rk, thr;           /* pixel row = fma(-y_ndc, rk, ra); a vertex has moved when |row - rowb| >= thr */
  float slack_ndc, slack_y;    /* report when beyond (previous extreme -/+ slack): NDC units / metres        */
  float thr_soft;              /* from |row - rowb| >= thr_soft on the body's face lists are sorted again OFF the chain, beside
                                  this cycle's gradient kernel, for the next launch; from thr on: now.  thr_soft = thr: no
                                  deferred sorts                                                                 */
  float* ndc;                  /* (B,V,3) out: x_ndc, y_ndc, z                                              */
  const float* rowb;           /* (B,V)   in : pixel row of every vertex at the body's last face sort        */
  int32_t* bbox;               /* (B,4)   out: order-preserving int of min x, min y, max x, max y (z > 1e-8 only);
                                                untouched entries stay at INT_MAX / INT_MIN                  */
  int32_t* bbox_prev;          /* (B,4)   the previous launch's bbox (copied, then bbox reset, by the launch itself) */
  unsigned long long* lowkey;  /* (B)     out: order-preserving bits of y << 32 | ~vertex, 0 = nobody reported */
  unsigned long long* lowkey_prev; /* (B) */
  int32_t* moved;              /* (2B)    out: [b] = 1: some vertex left its band (thr); [B + b] = 1: some vertex is on its way
                                                out (thr_soft).  Plain stores: thousands of vertices of a body report the same */
  float* clear;                /* or NULL: clear_n floats zeroed by the launch's first kernel -- the cycle's gradient buffer,
                                  so that a captured cycle starts with the pose kernel instead of a fill + a gap        */
  unsigned long long clear_n;
} mh_fwd_proj;
int mh_lbs_forward_proj(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                        const float* xscale, const float* transl, float* verts, float* vposed /*or NULL*/,
                        const mh_fwd_proj* proj, void* ws, void* stream);
/* low_idx / low_xyz of mh_lowest_vertex from the keys of mh_lbs_forward_proj (a body nobody reported for is scanned) */
int mh_lowest_resolve(const float* verts /*(B,V,3)*/, int B, int V, unsigned long long* lowkey /*(B): a missing key is written back*/,
                      int32_t* low_idx /*(B)*/, float* low_xyz /*(B,3)*/, void* stream);

/* Arithmetic of the two dense contractions of the LBS pair (pose/shape blend and its adjoint):
 * split16 = 1 (default): operands carried as two 16-bit terms, three products on the 16-bit matrix pipe with fp32
 * accumulation -- fp16 terms in the forward (vertices within ~1e-7 m of the fp32 result), bf16 terms in the backward
 * (gradients within ~2e-5 relative).  split16 = 0: exact fp32 MFMA (the round-1 kernels; A/B reference, 3-4x slower).
 * Process-wide; also selectable with MHHIP_LBS_FP32=1 in the environment before the first call.             */
int mh_lbs_set_mode(int split16);
int mh_lbs_get_mode(void);
/* mh_lbs_forward_proj as a kernel that is software-pipelined over a wave's vertex tiles (tile i's skinning / projection
 * epilogue issued between the matrix instructions of tile i + 1; on = 1, default), tile after tile (0), or as producer and
 * consumer waves (2: loads and matrix instructions in one wave, epilogue and stores in another, tiles handed over through
 * LDS): the same bits in every form.  MHHIP_FWD_PIPE=0|1|2 in the environment before the first call.  Process-wide.   */
int mh_lbs_set_forward_pipeline(int on);
int mh_lbs_get_forward_pipeline(void);

/* sparse joint regression from vertices (smpl.py:603-620, 367-386):
 * joints[b][j] = sum_v R[j][v] * verts[b][v]  (+ (1 - rowsum_j) * corr[b] when corr != NULL,
 * which turns regression of translated/scaled vertices into s*R*x + t exactly).
 * root_relative_to >= 0 subtracts that joint (h36m17: 14, smpl.py:371-372).                 */
int mh_joints_regress(const mh_model* m, int which, int B, const float* verts /*(B,V,3)*/,
                      const float* corr /*(B,3) or NULL*/, int root_relative_to,
                      float* joints /*(B,J,3)*/, void* stream);

/* ---- the 2D key-point term without a pass over the vertices (round 4; csrc/mh_keypoints.hip) -----------------------------
 * joints_alphapose (smpl.py:374-376), scaled and translated (optimizer.py:701-703), projected (transforms.py:57-95) and
 * compared with the detections (optimizer.py:364-368, 404, 419-420) -- evaluated from what mh_lbs_forward left in `ws`
 * (pose features, joint transforms): the regressed key-points are linear in those, the constants per (key-point, bone)
 * pair are built by mh_model_create.  kp (B,17,3), uv (B,17,2) or NULL, gkp (B,17,3) = dL/dkp or NULL, loss (B).
 * ws2 != NULL: the term's adjoint (dL/d joint transforms, dL/d pose features, dL/d translation) goes into the extra
 * chunk slot of the LBS backward's workspace; mh_lbs_backward_kp then runs the skinning adjoint without key-points and
 * adds that slot.  kp_ws: mh_keypoint_workspace_bytes(m, B) bytes of scratch.                                    */
size_t mh_keypoint_workspace_bytes(const mh_model* m, int B);
int mh_keypoint_terms(const mh_model* m, int B, const float* transl /*(B,3) or NULL*/, const float* K_host,
                      const float* Kd_host /*or NULL*/, const float* joint_w_host /*[17] or NULL*/,
                      const float* pose2d /*(B,17,3)*/, float thr, float img_w, float img_h, float coef,
                      float* kp, float* uv, float* gkp, float* loss, const void* ws, void* ws2 /*or NULL*/,
                      void* kp_ws, void* stream);
int mh_lbs_backward_kp(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                       const float* vposed, const float* gverts, float* gposes, float* gtransl,
                       float* gbetas, float* gxscale, void* ws, void* ws2, void* stream);
/* the same, with the rasteriser's closing job (mh_raster_fin, below; or NULL) carried out by one more workgroup of the pose
 * kernel */
typedef struct mh_raster_fin mh_raster_fin;
int mh_lbs_backward_kp_fin(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                       const float* vposed, const float* gverts, float* gposes, float* gtransl,
                       float* gbetas, float* gxscale, void* ws, void* ws2, const mh_raster_fin* fin, void* stream);

/* With gbetas == gxscale == NULL the backward entries leave the shape / log-scale gradients per BODY in ws2 and skip
 * the launch that sums them over a person's frames; these two entries are the other half: where those rows are (for
 * mh_rmsprop_step_person, which sums them inside the update's launch), and the sum as a launch of its own (+= into gbetas
 * (NB,10) and gxscale (NB), NULL = skip; same order and bits as the undeferred backward).  B and the LBS arithmetic mode
 * must be those of the backward call.                                                                                  */
int mh_lbs_backward_person_partials(const mh_model* m, int B, void* ws2, float** gbeta_b, float** gxs_b);
int mh_lbs_person_reduce(const mh_model* m, int B, int NB, void* ws2, float* gbetas, float* gxscale, void* stream);

/* ---- LBS backward (hand-written adjoint of the above) ---------------------------------------
 * In : gverts (B,V,3) = dL/dverts, gjoints (B,17,3) = dL/d(alphapose joints of the translated,
 *      scaled body) or NULL, vposed from the forward, the same parameters, ws from the forward.
 * Out: gposes (B,72) [+=], gtransl (B,3) [+=], gbetas (NB,10) [+=], gxscale (NB) [+=].
 * Accumulates (+=) so that several loss terms / the priors can share one gradient buffer, like
 * autograd accumulation across the reference's per-batch backward() calls (optimizer.py:544).
 * ws2: workspace of mh_lbs_backward_workspace_bytes(B) bytes.                               */
size_t mh_lbs_backward_workspace_bytes(int B);
int mh_lbs_backward(const mh_model* m, int B, int NB, const float* betas, const float* poses,
                    const float* xscale, const float* transl, const float* vposed,
                    const float* gverts, const float* gjoints, float* gposes, float* gtransl,
                    float* gbetas, float* gxscale, void* ws, void* ws2, void* stream);
/* The same with the inputs / outputs smpl.py's autograd also covers: exactly one of `poses` (axis-angle, gposes) and
 * `rotmats` ((B,24,3,3) as given to mh_lbs_forward_rotmats -- lbs(pose2rot=False), smpl.py:541-558 -- with grotmats
 * (B,24,9) +=); gposed (B,24,3) or NULL: dL/d(posed joints) = joints_smpl24 of SMPL.forward (smpl.py:362, 735). */
int mh_lbs_backward_ex(const mh_model* m, int B, int NB, const float* betas, const float* poses, const float* rotmats,
                       const float* vposed, const float* gverts, const float* gjoints, const float* gposed,
                       float* gposes, float* grotmats, float* gtransl, float* gbetas, float* gxscale,
                       void* ws, void* ws2, void* stream);
/* adjoint of mh_joints_regress for any of the four regressors (smpl.py:367-386 under autograd; optimizer.py:41, 75, 695-696
 * with another smpl_sparse_joints_key): gverts (B,V,3) += reg^T gjoints; gcorr (B,3) += the EXPLICIT share of the
 * translation correction: sum_j (1 - rowsum_j) gjoints_j for plain joints, sum_j (1 - rowsum_j + rowsum_root) gjoints_j
 * for joints relative to a root (the rest of d/dt reaches the translation through gverts).  NULL when the joints were
 * regressed without a correction.  Deterministic (no atomics). */
int mh_joints_regress_backward(const mh_model* m, int which, int B, const float* gjoints /*(B,J,3)*/, int root_relative_to,
                               float* gverts, float* gcorr, void* stream);

/* ---- a11/a12: pinhole projection of the 17 key-points + masked 2D residual ------------------
 * (transforms.py:74-95, optimizer.py:364-368, 404-405, 414-420).
 * K: HOST 3x3 row-major intrinsics; Kd: HOST 5 distortion coefficients or NULL.
 * pose2d (B,17,3) = (x, y, confidence).  mode 0 ("fit"): residual sum((c*(uv-gt)/[W,H])^2),
 * c = conf >= thr; mode 1 ("warm-up", optimizer.py:735,754-756): mean over all B*17*2 elements
 * of (c*(uv-gt))^2 in pixels, c = conf > thr.  Writes uv (B,17,2) (may be NULL), the gradient
 * gjoints (B,17,3) scaled by `coef` (overwritten) and per-body loss partials loss (B).      */
int mh_project_joints_loss(int B, const float* joints /*(B,17,3)*/, const float* K_host,
                           const float* Kd_host, const float* pose2d, float thr, int mode,
                           float img_w, float img_h, float coef, float* uv, float* gjoints,
                           float* loss, void* stream);

/* the same with per-key-point weights joint_w (HOST, 17 floats, or NULL = all ones): the reference multiplies the
 * confidence mask by `pose17j_weights` normalised to mean 1 (optimizer.py:75-130, 259, 419-420).           */
int mh_project_joints_loss_w(int B, const float* joints, const float* K_host, const float* Kd_host,
                             const float* joint_w_host, const float* pose2d, float thr, int mode, float img_w,
                             float img_h, float coef, float* uv, float* gjoints, float* loss, void* stream);

/* a9 warm-up (optimizer.py:710-770): only poses_T is a leaf there, so the 17 key-points of every
 * body are computed ONCE (mh_lbs_forward + mh_joints_regress) and each Adam iteration is
 * joints = 1.1^xscale[b%NB] * local + transl -> projection -> mean-reduced pixel residual
 * (optimizer.py:754-756) -> gtransl (B,3) = coef * d loss / d transl (overwritten), loss (B).  */
int mh_warmup_project(int B, int NB, const float* local_joints /*(B,17,3)*/, const float* xscale,
                      const float* transl /*(B,3)*/, const float* K_host, const float* Kd_host,
                      const float* pose2d, float thr, float coef, float* gtransl, float* loss,
                      void* stream);

int mh_warmup_project_w(int B, int NB, const float* local_joints, const float* xscale, const float* transl,
                        const float* K_host, const float* Kd_host, const float* joint_w_host, const float* pose2d,
                        float thr, float coef, float* gtransl, float* loss, void* stream);

/* ---- a20: optimiser updates (optimizer.py:355-356, 586-587, 738-739, 764-765) --------------- */
int mh_rmsprop_step(float* params, const float* grads, float* square_avg, float* momentum_buf,
                    size_t n, float lr, float alpha, float momentum, float eps, void* stream);
/* mh_rmsprop_step, and in the same launch nlog floats copied from log_src to log_dst: the captured
 * cycle writes its loss sums to a staging row at a fixed address, the row of the log they belong
 * to changes every cycle (optimizer.py:588-590 appends a dict per cycle); nlog = 0: no copy.    */
int mh_rmsprop_step_log(float* params, const float* grads, float* square_avg, float* momentum_buf,
                        size_t n, float lr, float alpha, float momentum, float eps,
                        const float* log_src, float* log_dst, int nlog, void* stream);
/* mh_rmsprop_step_log, and in the same launch up to two 32-bit words written to poke_dst[0 .. npoke): the
 * device-resident switches the NEXT captured cycle reads (which of the two scene grids is live: the scene
 * is rebuilt every cycle from cycle 30 on, optimizer.py:578-584, and one captured graph serves the whole
 * fit) travel with the update instead of in a launch of their own.  npoke = 0: mh_rmsprop_step_log.     */
int mh_rmsprop_step_log_poke(float* params, const float* grads, float* square_avg, float* momentum_buf,
                             size_t n, float lr, float alpha, float momentum, float eps,
                             const float* log_src, float* log_dst, int nlog, int32_t* poke_dst, int npoke,
                             int32_t poke0, int32_t poke1, void* stream);
/* The update that also closes the LBS backward (round 6).  mh_lbs_backward* called with gbetas == gxscale == NULL
 * leaves the shape / scale gradients as per-body rows in its workspace (mh_lbs_backward_person_partials) and skips
 * the launch that sums them over the frames of a person; this entry does those sums -- same order, same bits --
 * in NB * (nbeta + 1) extra workgroups of the update's launch, adds each to its element of grads (which is
 * therefore written: not const) and updates the element.  One launch and one launch gap less at the end of every
 * optimisation cycle (optimizer.py:343-356, 586-587).  off_xscale < 0: the scale sums are dropped (the reference
 * zeroes that gradient when optim_scale_factor is off).  Everything else: mh_rmsprop_step_log_poke.             */
typedef struct mh_person_sums {
  const float* gbeta_b;   /* [>= B][nbeta] per-body shape gradients (device) */
  const float* gxs_b;     /* [>= B]        per-body log-scale gradients      */
  int B, NB, nbeta;       /* bodies (frame-major: body b belongs to person b % NB), people, shape components */
  long long off_betas;    /* element offset of the (NB, nbeta) shape leaf in params / grads */
  long long off_xscale;   /* of the (NB) scale leaf; < 0: leave it alone */
} mh_person_sums;
int mh_rmsprop_step_person(float* params, float* grads, float* square_avg, float* momentum_buf, size_t n,
                           float lr, float alpha, float momentum, float eps, const float* log_src,
                           float* log_dst, int nlog, int32_t* poke_dst, int npoke, int32_t poke0,
                           int32_t poke1, const mh_person_sums* person, void* stream);
/* the same update with the learning rate resident on the device (lr_dev[0]); afterwards
 * lr_dev[0] *= gamma (ExponentialLR).  No host scalar changes between calls, so the whole cycle can
 * be captured in a hipGraph and replayed.                                                      */
int mh_rmsprop_step_dev(float* params, const float* grads, float* square_avg, float* momentum_buf,
                        size_t n, float* lr_dev, float gamma, float alpha, float momentum, float eps,
                        void* stream);
int mh_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                 int step, float lr, float beta1, float beta2, float eps, void* stream);

/* ---- a19: one-euro filter over the leading (time) axis (optimizer.py:664-675,
 * one_euro_filter.py:32-53 with d_cutoff = 1): x (T,E) -> y (T,E), E independent channels.  */
int mh_one_euro_scan(const float* x, float* y, int T, size_t E, float min_cutoff, float beta,
                     float frame_rate, void* stream);

/* frame-sharded form: this rank holds frames [first_frame, first_frame+T) of the sequence.
 * time_before = the float32 running time stamp after frame first_frame-1 (host-computable),
 * xprev_in / dxprev_in (E) = filtered value and filtered derivative of frame first_frame-1 from
 * the previous rank (NULL on the first rank); dxprev_out (E) = state handed to the next rank
 * (its xprev is y[T-1]).                                                                      */
int mh_one_euro_scan_shard(const float* x, float* y, int T, size_t E, float min_cutoff, float beta,
                           float frame_rate, int first_frame, float time_before,
                           const float* xprev_in, const float* dxprev_in, float* dxprev_out,
                           void* stream);

/* ---- a18: temporal terms (optimizer.py:560-575) ---------------------------------------------
 * velocity: loss = sum_t ||pT[t]-pT[t-1]||^2 over t=1..T-1; gpT += coef * d loss.
 * prev_halo / next_halo (N,3): poses_T of the frame before the first / after the last local
 * frame when the sequence is sharded over GPUs (NULL at the sequence ends).  The pair
 * (t-1,t) is owned by the rank that owns frame t.  loss_out: 1 float (overwritten).         */
int mh_velocity_term(int T, int N, const float* pT /*(T,N,3)*/, const float* prev_halo,
                     const float* next_halo, float coef, float* gpT, float* loss_out, void* stream);
/* filtered-vertex smoothness: loss = sum ||(v[t]-v[t-1]) - (vf[t]-vf[t-1])||^2,
 * gverts += coef * d loss / d v.  E = N*V*3 floats per frame.  prev_* / next_* (E): the
 * neighbouring ranks' boundary frames (NULL at the sequence ends).  ws: device workspace of
 * mh_filtered_verts_workspace_bytes(T, E) bytes (per-block partial sums; one per engine).    */
size_t mh_filtered_verts_workspace_bytes(int T, size_t E);
int mh_filtered_verts_term(int T, size_t E, const float* verts, const float* verts_filt,
                           const float* prev_v, const float* prev_vf, const float* next_v,
                           const float* next_vf, float coef, float* gverts, float* loss_out,
                           void* ws, void* stream);
/* the same, but gverts = coef * d loss / d v (overwrites: the caller starts its vertex-gradient buffer with this
 * term instead of clearing it first) */
int mh_filtered_verts_term_init(int T, size_t E, const float* verts, const float* verts_filt,
                                const float* prev_v, const float* prev_vf, const float* next_v,
                                const float* next_vf, float coef, float* gverts, float* loss_out,
                                void* ws, void* stream);
/* mh_filtered_verts_term_init behind a device-resident switch: while live_dev[0] == 0 the term does not exist yet
 * (optimizer.py:383-392: the one-euro filters first run at cycle 50; :571-573 `if self.verts_filtered is not None`)
 * -- gverts is cleared, loss_out = 0 -- so that ONE captured launch sequence serves every phase of a fit.    */
int mh_filtered_verts_term_init_gated(int T, size_t E, const float* verts, const float* verts_filt,
                                      const float* prev_v, const float* prev_vf, const float* next_v,
                                      const float* next_vf, float coef, float* gverts, float* loss_out,
                                      const int32_t* live_dev, void* ws, void* stream);

/* ---- staging of the constant per-frame inputs (once per sequence; optimizer.py:396-409, 434) --
 * The N float {0,1} instance masks of a frame become ONE 32-bit word per pixel (bit n = person n,
 * N <= 32): 4 B/pixel instead of 4N, and the erosion of morphology.py:6-41 runs once instead of
 * once per batch per cycle.                                                                  */
int mh_pack_masks(const float* seg /*(T,N,H,W)*/, int T, int N, int H, int W,
                  uint32_t* bits /*(T,H,W)*/, float* area /*(T,N) pixel counts*/, void* stream);
int mh_erode_bits(const uint32_t* in, uint32_t* out /*must not alias*/, int T, int H, int W, void* stream);
/* pose2d_valid = (#joints with conf >= thr) >= 2, mask_valid = area >= min_area (optimizer.py:404-409) */
int mh_stage_gates(const float* pose2d /*(B,17,3)*/, const float* area /*(B)*/, int B, float thr,
                   float min_area, float* pose2d_valid /*(B)*/, float* mask_valid /*(B)*/, void* stream);

/* ---- a14 (mask part): occlusion ordering of the silhouette term (optimizer.py:450-477) -------
 * front[t][n] = bit set of the people closer than n (poses_T z, ties by index); apply[t][n] =
 * gate of the term (indexed by RANK like the reference, :472); D = sum_px (1-acc);
 * S = sum_px (1-acc)*seg_n over the full image.                                              */
int mh_sil_mask_stats(const uint32_t* bits, int T, int N, int H, int W, const float* pT /*(T,N,3)*/,
                      const float* pose2d_valid, const float* mask_valid, uint32_t* front /*(T,N)*/,
                      float* apply /*(T,N)*/, float* D /*(T,N)*/, float* S /*(T,N)*/, void* stream);
/* the same with the frame's pixel counts kept while the near-to-far ordering of its people is unchanged: tag (T) int32,
 * zero before the first call on these arrays (and after the masks changed); front / D / S must then persist between calls */
int mh_sil_mask_stats_cached(const uint32_t* bits, int T, int N, int H, int W, const float* pT,
                             const float* pose2d_valid, const float* mask_valid, uint32_t* front, float* apply,
                             float* D, float* S, int32_t* tag, void* stream);

/* ---- a17: priors (optimizer.py:523-532, 535-542) ---------------------------------------------
 * pose prior L1(valid*ref, valid*pose) per body -> body_loss (B), gposes +=;
 * shape prior T*L1(beta, beta_ref) (the reference adds batch_size*L1 per batch), gbetas +=;
 * scale regularisers, added once per batch (nbatches), gxscale +=.
 * loss3 = { T*L1(beta), (sum(s-1))^2, mean((s-1)^2) }.                                       */
int mh_prior_terms(int T, int N, int nbatches, const float* poses, const float* poses_ref,
                   const float* valid /*(T*N)*/, const float* betas, const float* betas_ref,
                   const float* xscale /*(N) or NULL*/, float coef_poses, float coef_scales,
                   float* gposes, float* gbetas, float* gxscale, float* body_loss /*(T*N)*/,
                   float* loss3, void* stream);
/* out[0] = scale * sum(x[0..n)) in a fixed order (loss logging, optimizer.py:546-554, 588-593) */
int mh_reduce_sum(const float* x, size_t n, float scale, float* out, void* stream);
/* two arrays of the same length in one launch (the depth / silhouette log entries of a cycle); each sum in the order
 * of mh_reduce_sum */
/* n <= 8 independent sums in one launch: xs / lens / outs are HOST arrays of device pointers and lengths */
int mh_reduce_sum_multi(int n, const float* const* xs, const size_t* lens, float* const* outs, void* stream);
int mh_reduce_sum2(const float* x0, const float* x1, size_t n, float scale, float* out0, float* out1,
                   void* stream);

/* ---- a15/a16: scene contact and foot sliding (optimizer.py:485-518) --------------------------- */
/* argmax over vertices of y (first index on ties) and that vertex                           */
int mh_lowest_vertex(const float* verts /*(B,V,3)*/, int B, int V, int32_t* low_idx /*(B)*/,
                     float* low_xyz /*(B,3)*/, void* stream);
/* dy[b] = (mean of the k nearest scene points).y - low_xyz[b].y; k <= 32 (reference: 32, by a
 * full argsort over M, optimizer.py:494-502); if M < k all M points are used.              */
int mh_contact_knn(const float* points /*(M,3)*/, int M, const float* low_xyz, int B, int k,
                   float* dy /*(B)*/, void* stream);
/* the same neighbours through a uniform grid over the scene cloud: build once per scene update
 * (mh_scene_grid_build: bbox, counting sort into cells), then every query visits cells in growing
 * shells until its k-th distance is final.  grid_ws: mh_scene_grid_bytes(M) bytes owned by the caller. */
size_t mh_scene_grid_bytes(int M);
int mh_scene_grid_build(const float* points /*(M,3)*/, int M, void* grid_ws, void* stream);
/* the same with the point count in device memory (written by mh_scene_points): grid_ws sized for M_cap, which is also
 * the M to pass to mh_contact_knn_grid */
int mh_scene_grid_build_dev(const float* points, const int* M_dev, int M_cap, void* grid_ws, void* stream);
int mh_contact_knn_grid(const void* grid_ws, int M, const float* low_xyz, int B, int k,
                        float* dy /*(B)*/, void* stream);
/* the same with the query = the lowest vertex of each body, taken from the keys mh_lbs_forward_proj reported (as
 * mh_lowest_resolve; low_idx / low_xyz are outputs) -- one launch instead of two                                  */
int mh_contact_knn_grid_key(const void* grid_ws, int M, const float* verts /*(B,V,3)*/, int V,
                            unsigned long long* lowkey /*(B)*/, int B, int k, int32_t* low_idx /*(B)*/,
                            float* low_xyz /*(B,3)*/, float* dy /*(B)*/, void* stream);
/* mh_contact_knn_grid_key over one of TWO grids of the same capacity M, chosen by device-resident words: sel_dev[0] = 0 --
 * there is no scene yet (optimizer.py:485 `if self.scene_pcd is not None`), nothing is read or written; otherwise grid_ws0
 * (sel_dev[1] = 0) or grid_ws1.  The scene update of cycle c builds the grid cycle c does not read (optimizer.py:578-584);
 * the words change between cycles, the captured launch does not.  lowkey NULL: the queries are low_xyz (input).     */
int mh_contact_knn_grid_sel(const void* grid_ws0, const void* grid_ws1, int M, const int32_t* sel_dev,
                            const float* verts /*(B,V,3)*/, int V, unsigned long long* lowkey /*(B)*/, int B, int k,
                            int32_t* low_idx /*(B)*/, float* low_xyz /*(B,3)*/, float* dy /*(B)*/, void* stream);
/* contact: sum |dy+0.02| per batch, gpT.y += coef * (-sign(dy+0.02)); foot sliding between
 * IN-BATCH consecutive frames (batch = frames per batch, optimizer.py:512-518), gverts +=
 * (atomic).  batch_contact / batch_foot: one value per batch (nbatches = ceil(T/batch)).     */
int mh_contact_foot_terms(int T, int N, int V, int batch, const float* verts, const int32_t* low_idx,
                          const float* low_xyz, const float* dy, float coef_contact, float coef_foot,
                          float* gpT, float* gverts, float* batch_contact, float* batch_foot,
                          void* stream);
/* the same with an explicit batch table: batch_frames[bt * batch + k] = frame at position k of batch bt (-1 = empty
 * position of a shorter batch); position k is paired with position k-1 of the same batch exactly like the reference does
 * with whatever its dataloader delivered -- with the shipped `shuffle: True` (configs/predict_mupots.yml:14,
 * predict.py:273-277) random frames of a random batch (optimizer.py:394, 512-518).  Every frame must appear at most
 * once in the table. */
int mh_contact_foot_terms_idx(int T, int N, int V, int batch, int nbatches, const int32_t* batch_frames /*(nbatches,batch)*/,
                              const float* verts, const int32_t* low_idx, const float* low_xyz, const float* dy,
                              float coef_contact, float coef_foot, float* gpT, float* gverts, float* batch_contact,
                              float* batch_foot, void* stream);
/* mh_contact_foot_terms[_idx] (batch_frames NULL: contiguous batches) behind the same switch as mh_contact_knn_grid_sel:
 * live_dev[0] = 0 -> both sums are 0 and no gradient is touched.                                                    */
int mh_contact_foot_terms_gated(int T, int N, int V, int batch, int nbatches, const int32_t* batch_frames,
                                const float* verts, const int32_t* low_idx, const float* low_xyz, const float* dy,
                                float coef_contact, float coef_foot, float* gpT, float* gverts, float* batch_contact,
                                float* batch_foot, const int32_t* live_dev, void* stream);
/* a21: pixel centres + depth -> camera-space points (H*W,3) (optimizer.py:605-612,
 * transforms.py:114-130).  K_host: HOST 3x3 intrinsics.                                       */
int mh_scene_unproject(const float* depth /*(H,W)*/, int H, int W, const float* K_host,
                       float* points /*(H*W,3)*/, void* stream);

/* ---- a24: scene aggregation on the device (optimizer.py:578-584; fhsog.py:180-202 aggegrate_scene_geometry_median;
 * utils.py:174-209 postprocess_depthmap, :91-135 fillin_values).  ws: mh_scene_workspace_bytes(T,H,W) bytes.
 * mh_scene_median: per-pixel masked median over the T frames of 1/target_disp (target_disp from the normalised
 *   disparity `depths` and the depth-range leaves, optimizer.py:425-426), backmask (T,H,W) bytes, non-zero = background;
 *   np.ma.median semantics (mean of the two middle values; 0 and mask 0 where no frame is valid).
 * mh_scene_postprocess: bilateral(1/clip(depth)) d=9 sigmaColor .05 sigmaSpace 25 (optional), Sobel edge mask of
 *   disparity and depth (> 3 x mean of the std-normalised sum), eroded twice, times ma_mask (may be NULL), then the
 *   fillin_ksize x fillin_ksize median fill looped until no masked pixel is left.
 * mh_scene_points: un-projection of the pixels with mask > 0.5 (row-major order) into points (<= H*W rows); the number
 *   of points is written to count_dev (device memory: no host synchronisation). */
size_t mh_scene_workspace_bytes(int T, int H, int W);
int mh_scene_median(int T, int H, int W, const float* depths /*(T,H,W)*/, const uint8_t* backmask /*(T,H,W)*/,
                    const float* zmin_lin /*(T)*/, const float* zmax_lin /*(T)*/, float* ma_depth /*(H,W)*/,
                    float* ma_mask /*(H,W) 0/1*/, void* ws, void* stream);
/* the same with pixel-major inputs (P = H*W rows of T frames: the frames of a pixel contiguous), T <= 2048: one wave
 * per pixel, no LDS.  Any (H, W) with H*W = number of pixel rows works: a pixel-sharded caller passes its slice. */
int mh_scene_median_t(int T, int H, int W, const float* depths_t /*(H*W,T)*/, const uint8_t* backmask_t /*(H*W,T)*/,
                      const float* zmin_lin, const float* zmax_lin, float* ma_depth, float* ma_mask, void* ws, void* stream);
/* scene_depth must not alias ma_depth: the filtered map is built and filled in scene_depth itself (round 6: no copy at the end) */
int mh_scene_postprocess(int H, int W, const float* ma_depth, const float* ma_mask, int use_bilateral, int fillin_ksize,
                         float* scene_depth /*(H,W)*/, void* ws, void* stream);
/* utils.py:91-135 looped until no masked pixel is left (optimizer.py:595-600 uses it with ksize 11 on the colour median):
 * values / mask (H,W) are updated in place; truncate != 0 stores floor(median) (integer-valued planes).
 * mh_scene_median_t with zmin_lin = zmax_lin = NULL takes the median of the raw (non-negative) values. */
int mh_scene_fill(int H, int W, int ksize, int truncate, float* values, float* mask, void* ws, void* stream);
int mh_scene_points(int H, int W, const float* K_host, const float* scene_depth, const float* mask, float* points,
                    int* count_dev, void* stream);

/* ---- a13/a14: differentiable z-buffer + soft silhouette fused with their residuals ------------
 * Replaces, per cycle, for all T*N bodies at once: Meshes -> MeshRasterizer(K=8, blur 1e-4) ->
 * zbuf[...,0]; MeshRenderer(K=4, blur 2e-5) + SoftSilhouetteShader (optimizer.py:211-232,
 * 427-431, 447-448); the masked mean-log-disparity residual (optimizer.py:425-442,
 * losses.py:19-30) and the occlusion-ordered silhouette residual (optimizer.py:450-477,
 * losses.py:33-40); and the backward of all of it into the vertices and the depth-range leaves.
 * PyTorch3D conventions: camera R = diag(-1,-1,1), NDC from transforms.py:222-255, pixel centres,
 * perspective_correct = False, clipped barycentrics, no culling.  cam_K_host: HOST 3x3.
 * In : verts (T*N,V,3) camera space; faces (F,3); bits / ebits (T,H,W) raw / twice-eroded
 *      instance-mask words; depths (T,H,W) normalised disparity; zmin_lin, zmax_lin (T);
 *      pose2d_valid (T*N); front, sil_apply, sil_D, sil_S from mh_sil_mask_stats.
 * Out: gverts (T*N,V,3) += coef * dL/dverts (atomic; may be NULL = losses only);
 *      gzmin, gzmax (T) += ; depth_body, sil_body (T*N) per-body loss values (un-weighted);
 *      ws: mh_raster_workspace_bytes(T,N,V,F,H,W) bytes of scratch (work list + the per-pixel
 *      K-nearest windows; sized for the worst case of every body filling the image, only the
 *      windows actually covered are touched); zbuf_out / alpha_out (T*N,H,W) or NULL: the rendered
 *      nearest-face depth (-1 = empty) and soft-silhouette images (inspection / data synthesis;
 *      the optimisation loop never materialises them).                                          */
size_t mh_raster_workspace_bytes(int T, int N, int V, int F, int H, int W);
int mh_raster_terms(int T, int N, int V, int F, int H, int W, const float* cam_K_host,
                    const float* verts, const int32_t* faces, const uint32_t* bits,
                    const uint32_t* ebits, const float* depths, const float* zmin_lin,
                    const float* zmax_lin, const float* pose2d_valid, const uint32_t* front,
                    const float* sil_apply, const float* sil_D, const float* sil_S,
                    float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin,
                    float* gzmax, float* depth_body, float* sil_body, void* ws,
                    float* zbuf_out, float* alpha_out, void* stream);
/* the same call in two halves, so that a caller can order other writers of gverts between them: phases 1 = windows,
 * face sort, selection and the loss values (does not touch gverts; depth_body final, sil_body final only without
 * gradients), 2 = gradients into gverts / gzmin / gzmax and the final sil_body (same arguments, same ws), 3 = both */
int mh_raster_terms_phase(int T, int N, int V, int F, int H, int W, const float* cam_K_host,
                          const float* verts, const int32_t* faces, const uint32_t* bits,
                          const uint32_t* ebits, const float* depths, const float* zmin_lin,
                          const float* zmax_lin, const float* pose2d_valid, const uint32_t* front,
                          const float* sil_apply, const float* sil_D, const float* sil_S,
                          float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin,
                          float* gzmax, float* depth_body, float* sil_body, void* ws,
                          float* zbuf_out, float* alpha_out, int phases, void* stream);
/* the same; phase 2 also writes the sums over the bodies of the depth / silhouette loss values (one float each, device
 * memory, may be NULL) -- the two entries of a cycle's log row (optimizer.py:546-554) without a reduction launch */
int mh_raster_terms_phase_log(int T, int N, int V, int F, int H, int W, const float* cam_K_host,
                          const float* verts, const int32_t* faces, const uint32_t* bits,
                          const uint32_t* ebits, const float* depths, const float* zmin_lin,
                          const float* zmax_lin, const float* pose2d_valid, const uint32_t* front,
                          const float* sil_apply, const float* sil_D, const float* sil_S,
                          float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin,
                          float* gzmax, float* depth_body, float* sil_body, void* ws,
                          float* zbuf_out, float* alpha_out, int phases, float* log_depth, float* log_sil, void* stream);
/* the same after mh_lbs_forward_proj wrote into this workspace (mh_raster_forward_targets): projected != 0 -> the
 * preparation kernel reads the bounding boxes / motion flags the forward left instead of passing over `verts` (which
 * may then be NULL); a body whose box is incomplete is scanned from the projected vertices.  Same results, bit for bit.
 * phases here is a bit mask: 1 = preparation + selection, 2 = gradients, 4 = preparation only (windows, face lists, work
 * lists: a chain of small latency-bound launches), 8 = selection only -- so that a caller can start other work between the
 * two halves of phase 1. 128 (with 4 / 2, mh_raster_terms_deferred): the work lists -- a schedule: the kernels find every
 * tile and gradient unit with lists that are a launch old, or empty -- are not rebuilt between preparation and selection
 * but by the carrier of the closing job (mh_raster_fin.lists), for the next launch on the workspace. */
/* The rasterised terms' CLOSING job (per-body values from the tile sums, gradients of the depth-range leaves, the two log
 * sums: a few microseconds of latency-bound work in one workgroup) as a description that another launch can carry out:
 * nothing of the LBS backward reads what it writes, so mh_raster_terms_deferred leaves it here instead of launching
 * k_raster_finish between the gradient kernel and the backward, and mh_lbs_backward_kp_fin runs it as one more workgroup
 * of its pose kernel -- one launch and one dependent kernel less on the cycle's chain.  Plain pointers into the caller's
 * buffers and the rasteriser's workspace; valid until those are released. */
struct mh_raster_fin {
  int T, N, B, from_partials;
  float coef_depth;
  const int* body_first;
  const int* body_ns;
  const float* partial;
  const float* dinv;
  const float* sil_apply;
  const float* sil_D;
  const float* sil_S;
  float* sil_corr;
  float* depth_body;
  float* sil_body;
  const float* zmin_lin;
  const float* zmax_lin;
  float* gzmin;
  float* gzmax;
  float* log_depth;
  float* log_sil;
  /* phases & 128 of mh_raster_terms_deferred: the carrier also rebuilds the rasteriser's work lists (tile and gradient-unit
   * order: a schedule for the NEXT launch on that workspace) -- the parameter block it needs, opaque to the caller */
  int has_lists;
  unsigned long long lists[80];
};
int mh_raster_forward_targets(int T, int N, int V, int F, int H, int W, const float* cam_K_host, void* ws,
                              mh_fwd_proj* out);
int mh_raster_terms_projected(int T, int N, int V, int F, int H, int W, const float* cam_K_host,
                          const float* verts, const int32_t* faces, const uint32_t* bits,
                          const uint32_t* ebits, const float* depths, const float* zmin_lin,
                          const float* zmax_lin, const float* pose2d_valid, const uint32_t* front,
                          const float* sil_apply, const float* sil_D, const float* sil_S,
                          float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin,
                          float* gzmax, float* depth_body, float* sil_body, void* ws,
                          float* zbuf_out, float* alpha_out, int phases, float* log_depth, float* log_sil,
                          int projected, void* stream);
/* phases & 2 without the closing kernel: its job goes to *fin_out (mh_raster_fin above) */
int mh_raster_terms_deferred(int T, int N, int V, int F, int H, int W, const float* cam_K_host,
                          const float* verts, const int32_t* faces, const uint32_t* bits,
                          const uint32_t* ebits, const float* depths, const float* zmin_lin,
                          const float* zmax_lin, const float* pose2d_valid, const uint32_t* front,
                          const float* sil_apply, const float* sil_D, const float* sil_S,
                          float coef_depth, float coef_sil, float eps, float* gverts, float* gzmin,
                          float* gzmax, float* depth_body, float* sil_body, void* ws,
                          float* zbuf_out, float* alpha_out, int phases, float* log_depth, float* log_sil,
                          int projected, mh_raster_fin* fin_out, void* stream);
/* Deterministic gradient scatter (default off; MHHIP_DETERMINISTIC=1 in the environment switches it on at first use).
 * The production kernel sums the per-pixel vertex gradients of the rasterised terms with fp32 atomics (LDS table,
 * then global): the summation order, hence the last bits, vary from run to run.  on != 0: one workgroup per body,
 * 64-bit fixed-point integer accumulation (exact, order-free), one writer per element -- dL/dverts and all six
 * gradient leaves are bit-identical between runs, ~4-5x slower for this kernel. */
/* Temporal coherence of the rasteriser's preparation: the per-body face lists (sorted by first pixel
 * row) are kept across launches and rebuilt for a body only when one of its vertices has moved `rows` pixel rows since the
 * body's last sort; tiles read their candidates `rows` further out and every face is decided from the current
 * coordinates, so the selection is bit-identical to a fresh sort.  Default 1 (MHHIP_RASTER_SORT_MARGIN overrides at
 * first use); 0 = sort every launch.  mh_raster_sort_counters: {bodies seen, bodies re-sorted}, cumulative over the
 * launches on this workspace (synchronises the stream); out_host NULL resets them (stream-ordered). */
/* A workspace must be initialised ONCE before its first launch (stream-ordered; again if its bytes were overwritten):
 * clears the control words (re-sort and pair counters, face-list tags, silhouette accumulator).
 * mh_raster_workspace_offsets: byte offsets of {window table (B x 4 int32), first key of every body (B x int64, in
 * window pixels), key array (5 x uint64 per window pixel)} for inspection tools. */
int mh_raster_workspace_init(int T, int N, int V, int F, int H, int W, void* ws, void* stream);
int mh_raster_workspace_offsets(int T, int N, int V, int F, int H, int W, size_t* out /*[3]*/);
/* work of the selection kernel, counted while mh_profile_enable(2) (level 2 = event brackets + in-kernel work counters): {launches, candidate (face, pixel-centre) pairs,
 * pairs evaluated after the depth cull}, cumulative (achieved pair tests per second); synchronises */
int mh_raster_pair_counters(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host /*[3]*/, void* stream);
int mh_raster_set_sort_margin(int rows);
int mh_raster_get_sort_margin(void);
/* Deferred sorts (round 6): from `fraction` of the margin on, a body's lists are sorted again BESIDE the gradient kernel of the
 * launch that notices it, for the next launch -- off the chain of the cycle; only a jump of the whole margin inside one launch
 * still sorts in the preparation.  The keys do not depend on it (every face is decided from the current coordinates).
 * 0 or 1: no deferred sorts.  Default 0.6; MHHIP_RASTER_SORT_DEFER in the environment.  Process-wide.  At most 64
 * bodies per launch (the rest stay flagged for a later one, or reach the margin and sort in the preparation).       */
int mh_raster_set_sort_defer(float fraction);
float mh_raster_get_sort_defer(void);
/* test aid: all_even != 0 sends every round of the selection kernel down its even-split path (no depth cull, no pair list);
 * the selection keys must not depend on the path a round takes.  Process-wide; part of the cycle graphs' key. */
int mh_raster_set_path(int all_even);
int mh_raster_get_path(void);
int mh_raster_sort_counters(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host /*[2]*/, void* stream);
/* {bodies seen, bodies re-sorted, of which beside the gradient kernel (deferred sorts)}, cumulative */
int mh_raster_sort_counters3(int T, int N, int V, int F, int H, int W, void* ws, unsigned long long* out_host /*[3]*/, void* stream);
/* Winners first (round 5): on != 0 (default; MHHIP_RASTER_WINNERS=0 switches it off) = when a body's face lists are sorted, the
 * faces that held one of the five keys of some pixel in the previous launch on this workspace go into a list of their own
 * that every tile of the selection kernel rasterises FIRST, so that the depth cull meets nearly final 4th keys for all other
 * faces.  An order only: the selection keys are the same bits with and without it, whatever the workspace held.
 * Process-wide; part of the cycle graphs' key. */
int mh_raster_set_winners(int on);
int mh_raster_get_winners(void);
int mh_raster_set_deterministic(int on);
int mh_raster_get_deterministic(void);

/* ---- stand-alone forms of losses.py:19-40 and morphology.py:6-41 (call compatibility of
 * mhmocap.losses / mhmocap.morphology; the optimiser uses the fused kernels above) --------------
 * mh_avg_depth_loss: rows = b*N maps of P pixels; `tru` has rows/group maps (group = N when the
 * target is (b,1,H,W)).  row_sums (rows,3) = {sum m*log(clamp pred), sum m*log(clamp true), sum m};
 * the loss is sum_r ((s0 - s1)/(s2+1))^2 (host adds rows_loss from row_sums, or reads it back).
 * _backward: gpred (rows,P), gtrue_rows (rows,P) [caller sums the group], scaled by grad_out.      */
int mh_avg_depth_loss(const float* pred, const float* tru, const float* mask, int rows, int group,
                      size_t P, float eps, float* row_sums, float* row_loss, void* stream);
int mh_avg_depth_loss_backward(const float* pred, const float* tru, const float* mask, int rows,
                               int group, size_t P, float eps, const float* row_sums,
                               float grad_out, float* gpred, float* gtrue_rows, void* stream);
/* sums2 = { sum((mask*(a-b))^2), sum(mask) }; loss = sums2[0] / (sums2[1] + 1)                     */
int mh_masked_mse(const float* a, const float* b, const float* mask, size_t n, float* sums2, void* stream);
int mh_masked_mse_backward(const float* a, const float* b, const float* mask, size_t n,
                           const float* sums2, float grad_out, float* ga, void* stream);
/* binary erosion (dilate = 0) / dilation (dilate = 1) with a k x k window of ones on float maps    */
int mh_morph_f32(const float* in, float* out, int n_images, int H, int W, int kernel_size,
                 int dilate, void* stream);

/* generic forms of camera_projection_torch / camera_inverse_projection_torch (transforms.py:57-95,
 * 114-130): pts (B,M,3), K_dev (B,3,3) DEVICE, Kd_host 5 HOST coefficients or NULL;
 * out (B,M,2) or (B,M,3) with the depth appended (return_depth).                                 */
int mh_project_points(int B, int M, const float* pts, const float* K_dev, const float* Kd_host,
                      int with_depth, float* out, void* stream);
int mh_unproject_points(int B, int M, const float* uvd, const float* K_dev, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MHMOCAP_HIP_H */
