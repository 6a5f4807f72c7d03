/* nrs.h -- C ABI of the MI355X-native NR-SLAM hot path (libnrs_hip.so).
 *
 * The reference (endomapper/NR-SLAM) has no FFI; its seam is a handful of C++ free functions and
 * two classes.  Each entry point below replaces the *body* of one of them; the C++
 * shim in nr-slam_amd/host/ keeps the reference signatures, flattens Frame/Map/KeyFrame into the
 * plain arrays declared here and calls through.
 *
 * Conventions
 *   - all functions return 0 on success or a negative nrs_status; nothing throws across the ABI;
 *     nrs_last_error(ctx) gives a human-readable message for the last failure on that context.
 *   - host pointers are caller-owned and only read/written during the call; the context owns all
 *     device memory and one HIP stream; a context is NOT thread-safe (the reference calls these
 *     functions from its single main thread, modules/SLAM/system.cc:113-132).
 *   - poses are double[7] = {qx,qy,qz,qw,tx,ty,tz} of T_camera_world, i.e. what the reference
 *     hands to g2o::SE3Quat (modules/optimization/g2o_optimization.cc:68-71,165-168,907-910).
 *   - camera model: 0 = PinHole (fx fy cx cy), 1 = KannalaBrandt8 (fx fy cx cy k0 k1 k2 k3)
 *     (modules/calibration/pin_hole.cc:22-25, kannala_brandt_8.cc:25-32).
 *   - LandmarkStatus values are the reference's (modules/utilities/landmark_status.h:23-30).
 *   - there is no CPU fallback: every compute entry point fails with NRS_ERR_NO_DEVICE when no
 *     HIP device is usable.
 */
#ifndef NRS_H
#define NRS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRS_VERSION 100

typedef enum {
    NRS_OK = 0,
    NRS_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, index out of range) */
    NRS_ERR_NO_DEVICE = -2,   /* no usable HIP device / HIP runtime error at context creation    */
    NRS_ERR_HIP = -3,         /* HIP runtime error during the call (message in nrs_last_error)    */
    NRS_ERR_ALLOC = -4,       /* host or device allocation failed                                 */
    NRS_ERR_STATE = -5,       /* call sequence error (e.g. optimize before upload)                */
    NRS_ERR_NUMERIC = -6,     /* non-finite values reached the solver                             */
    NRS_ERR_COMM = -7         /* RCCL not loadable / a collective failed (message in nrs_last_error) */
} nrs_status;

typedef enum { NRS_CAM_PINHOLE = 0, NRS_CAM_KB8 = 1 } nrs_camera_model;

/* LandmarkStatus (modules/utilities/landmark_status.h:23-30) */
enum { NRS_TRACKED_WITH_3D = 0, NRS_TRACKED = 1, NRS_JUST_TRIANGULATED = 2, NRS_BAD = 3,
       NRS_OUT_IMAGE_BOUNDARIES = 4, NRS_BAD_FEATURE = 5 };

/* RegularizationGraph::Status (modules/map/regularization_graph.h:42-47) */
enum { NRS_GRAPH_VERIFIED = 0, NRS_GRAPH_NEIGHBOR = 1, NRS_GRAPH_NEUTRAL = 2, NRS_GRAPH_BAD = 3 };

typedef struct nrs_ctx nrs_ctx;

typedef struct {
    int32_t model;            /* nrs_camera_model */
    float params[8];
} nrs_camera;

 /* nrs_options grows at its END only and says how long the caller's version of it is: nrs_create reads struct_size bytes and takes
 * the defaults for everything behind them, so a binding compiled against an older header keeps working.  Fill it with
 * nrs_options_init() (device = -1, everything else 0 = default, struct_size = sizeof) or set struct_size = sizeof(nrs_options) by
 * hand; struct_size = 0 is read as the first published layout (fields up to exact_trials, 32 bytes) -- a caller that does not use
 * nrs_options_init MUST set struct_size or zero the whole struct: a value larger than 4096 is rejected as garbage (NRS_ERR_INVALID). */
typedef struct {
    int32_t device;           /* HIP device ordinal; -1 = current device                         */
    uint32_t struct_size;     /* sizeof(nrs_options) as the CALLER was compiled (sits where the layout had padding) */
    double pcg_rtol;          /* relative residual ||b-Ax||/||b|| at which the inner solve stops;
                                 0 = default 1e-10 (SURVEY.md 7.2 hard part 1)                    */
    int32_t pcg_max_iters;    /* 0 = default 2000                                                */
    int32_t pcg_batch;        /* PCG iterations enqueued between host checks; 0 = default 8       */
    int32_t profile;          /* 1 = time the linearise and SpMV kernels with HIP events on the
                                 context stream (serialises launches; for bench roofline only)    */
    int32_t exact_trials;     /* 0 (default): an LM trial whose gain ratio is already < -1 / -0.25 / -0.1 /
                                 -0.03 when the inner solve has reached 1e-1 / 1e-2 / 1e-3 / 1e-4 (and whose
                                 chi2 increase exceeds 1e-5 chi2) is rejected there.  This is a HEURISTIC with
                                 measured margins (the gain ratio of a partially converged step is
                                 within 0.15 / 0.017 / 0.004 / 0.002 of its final value at those milestones;
                                 0 decision changes in 200-problem sweeps): the step of a rejected trial is
                                 discarded and accepted steps are always solved to pcg_rtol, so the iterates
                                 are the reference's as long as no rejection is mispredicted.
                                 1: every trial is solved to pcg_rtol (g2o's behaviour; use it to verify). */
    int32_t direct_solve;     /* linear solver of the single-frame problems (nrs_track_deform_solve[_rg]):
                                 0 (default): a sparse direct solve (nested dissection, multifrontal Cholesky on the
                                 matrix cores: what LinearSolverEigen does, linear_solver_eigen.h:92-173) for frames of
                                 up to 8000 free rows (1.1 .. 4.4 x faster than the PCG at every measured size, 129 ..
                                 4525 points: DESIGN.md section 1), PCG beyond; 1: direct whenever the problem has the
                                 single-frame structure; 2: always
                                 PCG.  Same LM iterates either way (both are held to the oracle).  BA windows always
                                 use PCG.  Values outside 0..2 are read as 0. */
    int32_t embedded_solver;  /* linear solver of the EMBEDDED BA windows (nrs_dba_*_embedded):
                                 the system of an embedded window is block tridiagonal over the keyframes -- dense keyframe
                                 blocks (node copies + pose, coupled through the skinned observations), sparse damper
                                 couplings -- and can be factorised EXACTLY per LM trial (what LinearSolverEigen does for the
                                 reference, linear_solver_eigen.h:92-173): the keyframe blocks are inverted on the matrix
                                 cores along two elimination chains, and the factorisation preconditions the PCG, which then
                                 converges to pcg_rtol in one or two iterations (csrc/nrs_engine_kft.hpp).
                                 0 (default): that solver where a cost model of the two says it is the faster one (measured:
                                 5 x at 100 nodes x 20 keyframes, 2.3 x at 300, 1.24 x at 440, break-even at the literal C2 --
                                 458 node copies x 20 keyframes; the PCG wins beyond), block-Jacobi PCG else;
                                 1: always the factorisation (if its K x (3 nodes + 6)^2 x 8 bytes fit 6 GB);
                                 2: always block-Jacobi PCG (hundreds of iterations per trial).  Same LM iterates either
                                 way.  Plain (every point a node) and sharded windows use the block-Jacobi PCG.
                                 (Sits where the 40-byte layout had padding: values outside 0..2 are read as 0.) */
} nrs_options;
void nrs_options_init(nrs_options* opt);

/* One Levenberg-Marquardt trial as executed by g2o
 * (third_party/g2o/g2o/core/optimization_algorithm_levenberg.cpp:94-145). */
typedef struct {
    int32_t round;            /* outer round of the entry point (inlier rounds), 0 for BA         */
    int32_t iter;             /* LM iteration inside optimize()                                  */
    int32_t trial;            /* qmax at the time of the trial                                   */
    int32_t accepted;
    int32_t solver_ok;        /* 0 = linear solve reported "not positive definite"               */
    int32_t inner_iters;      /* PCG iterations of this trial (0 for dense 6x6 solves)           */
    int32_t early_rejected;   /* 1 = rejected at a peek (chi2_new / rho are the peek values) */
    int32_t reserved;
    double lambda;
    double chi2;              /* currentChi                                                      */
    double chi2_new;          /* tempChi                                                         */
    double rho;
} nrs_lm_trial;

typedef struct {
    nrs_lm_trial* trials;     /* caller-allocated, may be NULL                                   */
    int32_t capacity;
    int32_t count;            /* out: trials executed (may exceed capacity; extra are dropped)   */
    int32_t iterations;       /* out: LM iterations executed (cjIterations)                      */
} nrs_lm_trace;

/* Kernel timing accumulated while nrs_options.profile = 1 */
typedef struct {
    double linearize_ms;  int64_t linearize_launches;
    double spmv_ms;       int64_t spmv_launches;
    double vec_ms;        int64_t vec_launches;
    double update_ms;     int64_t update_launches;
} nrs_profile;

int nrs_create(nrs_ctx** out, const nrs_options* opt);

/* ---- Debug switches -----------------------------------------------------------------------------------------------------
 * None is needed in normal use.  A context reads the NRS_* variables of the environment ONCE, in nrs_create (also
 * NRS_DEBUG="NAME=VALUE,NAME2=VALUE2", names without the prefix); afterwards the library never looks at the environment again --
 * a host application may call setenv at any time -- and a switch is changed through nrs_debug_set only (value NULL: unset;
 * takes effect at the next problem upload / solve of that context).  The host edge builders, which take no context, read
 * NRS_HOST_THREADS from a process-wide snapshot taken at their first call.
 *   measurement   NRS_TIMING (stage marks of engine_create and the a2 driver on stderr), NRS_ND_DBG / NRS_ND_DBG2 (phase clocks of one
 *                 direct solve through nrs_debug_nd_solve), NRS_PEEK_DEBUG (gain ratios at the early-rejection looks)
 *   checks        NRS_POISON (fresh device allocations filled with NaN patterns), NRS_CHECK_EVAL / NRS_CHECK_FUSED / NRS_CHECK_CHI (an
 *                 evaluation / a single-launch PCG iteration / the carried chi2 computed twice and compared)
 *   direct solver NRS_ND=0|1, NRS_ND_MAX_ROWS, NRS_ND_NO_CACHE, NRS_ND_NO_COVER, NRS_ND_CHAIN, NRS_ND_LEVELS, NRS_ND_THREADS=256,
 *                 NRS_ND_STEP32=0, NRS_ND_BACK_FLAGS, NRS_ND_LEAF=<n>, NRS_ND_NO_SPLIT, NRS_ND_PLAN_PAR=<n>
 *   a2 LM trials  NRS_SPEC_TRIALS=<0..3> (shadow sets for the speculative trials of a run of rejections; 0: one at a time; read when an
 *                 engine is created), NRS_SPEC_FIXED=<n>, NRS_SPEC_FIRST=<n>, NRS_SPEC_NO_ABORT, NRS_SPEC_DBG
 *   PCG / packing NRS_NO_LDS, NRS_NO_FUSED, NRS_FUSED_MAX_ROWS=<n>, NRS_NO_COARSE, NRS_COARSE_MIN_TILES=<n>, NRS_NO_ONE_XCD, NRS_NO_ECD, NRS_HIER,
 *                 NRS_NO_PLAIN, NRS_NO_H4, NRS_RC=<0..3>, NRS_NT=0|1, NRS_DFORM, NRS_NO_EDGE_CHI, NRS_SELL_T=<lanes>, NRS_NO_MORTON,
 *                 NRS_NO_TILE_SORT, NRS_ONE_CLASS, NRS_TILE_CUT_PCT=<p>, NRS_HOST_PACK, NRS_HOST_THREADS=<n>, NRS_HOST_THREADS_SMALL=<n>,
 *                 NRS_SKIN_OP_OWN_LAUNCH, NRS_SKIN_ROWS_OWN_LAUNCH, NRS_PCG_RTOL=<r> (experiments)
 *   embedded BA   NRS_KFT_TWO_LAUNCHES (a sweep step of the keyframe-block factorisation as two launches), NRS_KFT_SCALAR_SWEEP (the pivot
 *                 block's sweep in the 4-pivot register form), NRS_KFT_FOUR_WAVES (panel workgroups without the four helper waves), NRS_KFT_NO_RESIDUAL_TEST (M^-1 applied again after the first PCG step
 *                 instead of the step's residual tested on its own)
 *   sharding      NRS_SHARD_PACK_ALL, NRS_SHARD_FULL_VECTORS
 *   a1 / graph    NRS_PO_MULTI_MIN=<n>, NRS_HOST_WALK, NRS_WALK_MAX_PASSES=<n>, NRS_RG_NO_MIRROR
 * (what each selects: README.md "Debug switches"; the A/B tests drive every launch form through nrs_debug_set). */
int nrs_debug_set(nrs_ctx* ctx, const char* name /* "NRS_..." */, const char* value /* NULL: unset */);
void nrs_destroy(nrs_ctx* ctx);
const char* nrs_last_error(const nrs_ctx* ctx);
int nrs_device_name(const nrs_ctx* ctx, char* buf, int32_t buf_len);
int nrs_get_profile(const nrs_ctx* ctx, nrs_profile* out);
int nrs_reset_profile(nrs_ctx* ctx);
void* nrs_stream(nrs_ctx* ctx);                 /* hipStream_t the context launches on */

/* ---- a1: CameraPoseOptimization (modules/optimization/g2o_optimization.cc:50-146) ------------
 * 3 rounds x optimize(10), restart from the seed each round, chi2 > 5.99 edges to level 1 between
 * rounds; all three rounds use the Huber kernel (the reference's setRobustKernel(0) at OPT:134-137 runs
 * after the last optimize() and has no effect on the solve).  uv: n x 2 keypoints, X: n x 3 landmark positions of
 * the TRACKED_WITH_3D observations in frame index order.  pose_qt in: frame pose, out: refined.
 * inlier (may be NULL): final classification of every edge. */
int nrs_pose_only_solve(nrs_ctx* ctx, const nrs_camera* cam, int32_t n, const float* uv,
                        const float* X, double pose_qt[7], uint8_t* inlier, nrs_lm_trace* trace);

/* ---- a3: LocalDeformableBundleAdjustment (g2o_optimization.cc:880-1161) ----------------------
 * Host-side edge construction of OPT:927-1137 from flattened keyframes and the ordered
 * neighbour lists that RegularizationGraph::GetEdges returns (a19).  kf_rowptr[n_kf+1] delimits,
 * oldest keyframe first, the map-point index of every TRACKED_WITH_3D observation in keyframe
 * index order (kf_pt).  Landmark l = position in that concatenation.  Two-call pattern: pass
 * sp_ij = NULL to obtain the counts, then call again with buffers of that size. */
int nrs_dba_build_edges(int32_t n_kf, const int32_t* kf_rowptr, const int32_t* kf_pt,
                        int32_t n_points, const int32_t* nbr_rowptr, const int32_t* nbr_col,
                        const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status,
                        int32_t* n_spring, int32_t* sp_ij, float* sp_d0,
                        int32_t* n_damper, int32_t* dm_idx, float* dm_w);

/* One-shot: upload, optimize(iters), download.  lm_kf[l] = keyframe (pose index) of landmark l,
 * must be non-decreasing.  sp_ij: n_spring x 2 landmark indices; dm_idx: n_damper x 4
 * (1c,2c,1n,2n).  scale = Map::GetMapScale().  The reference calls optimize(5). */
int nrs_dba_solve(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_kf, double* poses_qt,
                  int32_t n_lm, float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                  int32_t n_spring, const int32_t* sp_ij, const float* sp_d0,
                  int32_t n_damper, const int32_t* dm_idx, const float* dm_w,
                  float scale, int32_t iters, nrs_lm_trace* trace);

/* ---- N2b: the EMBEDDED form of the window -- BASELINE configs[1] as written, "5k map points x 500 deformation-graph nodes x 20
 * keyframes".  The reference has no such estimator (its graph has a vertex per map point); this is
 * LocalDeformableBundleAdjustment (g2o_optimization.cc:880-1161) with the vertices restricted to the keyframe copies of a NODE set:
 * springs (OPT:1031-1072) and dampers (OPT:1076-1132) by the reference's walks between node copies only, and every other observed
 * point SKINNED in its keyframe to the <= 11 node copies its walk accepts, x = X0 + sum_k omega_k (x_{n_k} - x_start_{n_k}) with the
 * connection weights normalised; its ReprojectionError edge (reprojection_error.cc:32-64) keeps residual, information and Huber
 * kernel and constrains those node copies and the keyframe pose (Jacobian omega_k x the reference's block).  With every point a
 * node all of it IS the plain window (same lists, same bits); beyond that it is held to oracle/embedded_oracle.py
 * (dba_build_embedded / dba_solve_embedded), "parity unpinned".  Observation o = position in the concatenation kf_pt.
 *   nrs_dba_build_edges_embedded  host: node copies (lm_obs[n_lm]: their observation, keyframe-major), springs / dampers over
 *                                 node-copy indices, skinned observations (sk_obs, sk_node [n x 11, -1 pads], sk_omega [n x 11]);
 *                                 two-call pattern: lm_obs = NULL returns the four counts
 *   nrs_dba_upload_embedded       the window resident: lm_* of the node copies, sk_kf / sk_uv / sk_xyz of the skinned observations
 *                                 (sk_xyz = X0, their positions at the start); then nrs_dba_reset / optimize / download as before
 *   nrs_dba_download_skinned      the skinned points at the current estimate (n_skin x 3, fp64)
 *   nrs_dba_solve_embedded        one shot: upload, optimize(iters), download (poses_qt, lm_xyz, sk_xyz in/out)
 * One GPU (no communicator); the linear solve is the PCG with the observations' blocks applied as hyper-edges
 * (csrc/nrs_engine_skin.hpp). */
int nrs_dba_build_edges_embedded(int32_t n_kf, const int32_t* kf_rowptr, const int32_t* kf_pt, int32_t n_points, const uint8_t* is_node,
                                 const int32_t* nbr_rowptr, const int32_t* nbr_col, const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status,
                                 int32_t* n_lm, int32_t* lm_obs, int32_t* n_spring, int32_t* sp_ij, float* sp_d0,
                                 int32_t* n_damper, int32_t* dm_idx, float* dm_w,
                                 int32_t* n_skin, int32_t* sk_obs, int32_t* sk_node, double* sk_omega);
int nrs_dba_upload_embedded(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                            int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                            int32_t n_spring, const int32_t* sp_ij, const float* sp_d0,
                            int32_t n_damper, const int32_t* dm_idx, const float* dm_w,
                            int32_t n_skin, const int32_t* sk_kf, const float* sk_uv, const float* sk_xyz,
                            const int32_t* sk_node, const double* sk_omega, float scale);
int nrs_dba_download_skinned(nrs_ctx* ctx, double* sk_xyz /* n_skin x 3, fp64 */);
int nrs_dba_solve_embedded(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_kf, double* poses_qt,
                           int32_t n_lm, float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                           int32_t n_spring, const int32_t* sp_ij, const float* sp_d0,
                           int32_t n_damper, const int32_t* dm_idx, const float* dm_w,
                           int32_t n_skin, const int32_t* sk_kf, const float* sk_uv, float* sk_xyz,
                           const int32_t* sk_node, const double* sk_omega, float scale, int32_t iters, nrs_lm_trace* trace);

/* The whole of LocalDeformableBundleAdjustment in one call, as mapping.cc:57 makes it: flattened keyframes (kf_rowptr / kf_pt as in
 * nrs_dba_build_edges; lm_xyz / lm_uv per landmark = position in that concatenation, lm_xyz in/out) + the ordered neighbour lists
 * RegularizationGraph::GetEdges returns (a19).  The edge construction of OPT:927-1137 runs on the device as well (index for index
 * the lists nrs_dba_build_edges returns), followed by the device-side problem construction and optimize(iters).  Windows that do
 * not qualify for the device path (a single keyframe, tiny windows, a communicator on the context) take nrs_dba_build_edges +
 * nrs_dba_solve internally: the result is the same.  nrs_dba_window_edges (parity tap): the edge lists of the resident window
 * when they were built on the device (two-call pattern: null arrays return the counts). */
int nrs_dba_solve_window(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_kf, double* poses_qt, const int32_t* kf_rowptr,
                         const int32_t* kf_pt, float* lm_xyz, const float* lm_uv, int32_t n_points, const int32_t* nbr_rowptr,
                         const int32_t* nbr_col, const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status, float scale,
                         int32_t iters, nrs_lm_trace* trace);
int nrs_dba_window_edges(nrs_ctx* ctx, int32_t* n_spring, int32_t* sp_ij, float* sp_d0, int32_t* n_damper, int32_t* dm_idx, float* dm_w);

/* Device-resident form of the same solve (used by bench.py so that the timed region starts with
 * the inputs already in HBM): upload once, then any number of {reset, optimize}. */
int nrs_dba_upload(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                   int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                   int32_t n_spring, const int32_t* sp_ij, const float* sp_d0,
                   int32_t n_damper, const int32_t* dm_idx, const float* dm_w, float scale);
int nrs_dba_reset(nrs_ctx* ctx);                                  /* estimates <- uploaded values */
int nrs_dba_optimize(nrs_ctx* ctx, int32_t iters, nrs_lm_trace* trace);
int nrs_dba_download(nrs_ctx* ctx, double* poses_qt, double* lm_xyz /* n_lm x 3, fp64 */);
/* Debug/parity taps on the resident problem (host buffers, fp64): per-edge residuals at the
 * current estimate and the assembled gradient b = -J^T W r in solver order (poses then landmarks). */
int nrs_dba_residuals(nrs_ctx* ctx, double* r_reproj /* n_lm x 2 */, double* r_spring /* n_spring */,
                      double* r_damper /* n_damper x 3 */);
int nrs_dba_gradient(nrs_ctx* ctx, double* b /* 6 n_kf + 3 n_lm */, double* diag /* same size */);
/* Parity tap of the embedded window's keyframe-block factorisation (nrs_options.embedded_solver = 0; csrc/nrs_engine_kft.hpp) on the resident
 * window, linearised at the current estimate with damping lam:
 *   what 0  out_i[0..6) = {in use, keyframes K, block dimension ld, ld / 64, middle keyframe, factor MiB}, then K free-node counts, K pose sizes (6 / 0)
 *   what 1  out_d = the assembled diagonal block of keyframe k (ld x ld: its free node copies in row order, then its pose)
 *   what 2  out_d = the coupling of keyframes k and k + 1 as a dense ld x ld matrix (rows: k)
 *   what 3  out_d = M^-1 in_d, both in solver order (6 per pose, then 3 per node copy): the factorisation applied as the PCG applies it
 *   what 4  out_i = per node copy its (keyframe, compact index) pair, -1: fixed */
int nrs_debug_kft(nrs_ctx* ctx, double lam, int32_t what, int32_t k, const double* in_d, double* out_d, int32_t* out_i);

/* Parity tap of the problem construction (edge lists -> row layout, sliced-ELL incidence streams, halo lists, chi2 edge lists;
 * g2o_optimization.cc:927-1137 ends where this starts): FNV-1a checksums of every packed array of the resident problem,
 * out[0..24).  A plain BA window (>= 2 keyframes, >= 2048 padded rows, nothing fixed, no masks / offsets / unary or incomplete
 * dampers, no communicator: both the two-kernel path, T = 2, and the fused one, T = 8) is packed on the device
 * (csrc/nrs_engine_devpack.hpp), everything else on the host -- as is everything under NRS_HOST_PACK=1 or any of the A/B switches
 * the host path honours (NRS_SELL_T, NRS_NO_FUSED, NRS_FUSED_MAX_ROWS, NRS_NO_PLAIN, NRS_NO_LDS, NRS_DFORM, NRS_NO_EDGE_CHI,
 * NRS_TILE_CUT_PCT, NRS_HIER, NRS_NO_ECD); the two constructions produce the same bits (out[21] says which one ran: 1 = device;
 * it is not part of the comparison). */
int nrs_dba_pack_hash(nrs_ctx* ctx, uint64_t* out /* 24 */);

/* Parity tap for a18 (the linear solve): solves (H + lambda I) x = b for an explicitly given block system with
 * the engine's own PCG kernels (block-Jacobi preconditioner, operator on stored blocks, single-reduction CG),
 * as g2o's linear solvers are handed an explicit SparseBlockMatrix in the reference's known-answer test
 * (third_party/g2o/unit_test/solver/linear_solver_test.cpp:72-85).  Shape: one 6x6 block H_pp (21 packed upper
 * entries, row-major), n_rows 3x3 diagonal blocks D6 (xx xy xz yy yz zz), and one 6x3 coupling block per row
 * (Hpl18: entry p*3 + c = H[pose p][row component c]); no row-row coupling.  x = [x_pose(6), x_rows(3 n_rows)].
 * Not used by any solve entry point. */
int nrs_debug_pcg_solve(nrs_ctx* ctx, int32_t n_rows, const double* Hpp21, const double* bp /*6*/, const double* D6,
                        const double* Hpl18, const double* bl /*3 n_rows*/, double lambda, double* x, int32_t* iters);

/* ---- N1 parity tap: the direct (nested-dissection, multifrontal Cholesky) solve of a2's system on an explicitly given
 * block system -- what replaces LinearSolverEigen::solve (third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-173:
 * ordering + symbolic step once, numeric sparse Cholesky per LM trial) when nrs_track_deform_solve[_rg] runs a frame on the
 * direct path.  n_nodes unknown blocks of 3 scalars at positions pos (n x 3, used for the
 * dissection only), `last` (n, may be null) marks blocks that are eliminated at the root whatever their position (the two
 * halves of a pose block), n_pairs unique couplings (a, b) with their 3x3 blocks Vp (row-major, rows = a's components),
 * diagonal blocks Dn (9 per node), right-hand side bn (3 per node): solves (A + lambda I) x = b.  Returns
 * NRS_ERR_NUMERIC when a pivot is not positive (linear_solver_eigen.h:124-136: the LM trial then counts as failed).
 * stats (8, may be null): fronts, levels, largest own / boundary size, L and U doubles, factorisation flops, workgroups.
 * Runs the device kernels (k_nd_level / k_nd_back) `repeats` times (timing: ms_per_solve, may be null).  Not used by any solve
 * entry point; the host reference it is held to lives in oracle/nd_host.cpp. */
int nrs_debug_nd_solve(nrs_ctx* ctx, int32_t n_nodes, const double* pos, const uint8_t* last, int32_t n_pairs, const int32_t* pairs,
                       const double* Dn, const double* Vp, const double* bn, double lambda, int32_t repeats, double* x, int64_t* stats,
                       double* ms_per_solve);
/* The context keeps the symbolic factorisation (ordering, fronts, device arrays) of the last few single-frame problems and reuses
 * it when a later problem has the same structure -- as a caller of linear_solver_eigen.h:144-169 does who does not request a new
 * ordering (the symbolic step runs once, then only numeric factorisations).  out[0] = problems that reused a plan, out[1] = plans
 * built, since the context was created. */
int nrs_debug_nd_cache_stats(nrs_ctx* ctx, int64_t out[2]);

/* ---- f3: ShiTomasi (modules/features/shi_tomasi.{h,cc}) + Tracking::ExtractFeatures (tracking.cc:118-134) --
 * nrs_shi_configure = ShiTomasi::ShiTomasi(Options) (shi_tomasi.cc:29-31): a fresh extractor (zeroed
 * buffers on first use, feature ids from 0).  nrs_shi_extract = ShiTomasi::Extract (shi_tomasi.cc:38-54)
 * followed by the caller's mask filter: prev_xy are the keypoints the frame already holds (their score
 * cells are set to -1, shi_tomasi.cc:93-96; they are not returned), out = the NEW keypoints in the
 * reference's row-major order with their class ids (ids are consumed before the mask drops points).
 * The extractor is stateful exactly like the reference object: gradient and score buffers persist
 * between calls (reallocated, zeroed, only when the image size changes) and the reference's single pass
 * leaves some cells to the next call.  width >= height >= 5 is required: the reference's
 * first / last row loops run their column index to rows-1 (shi_tomasi.cc:187,336).
 * *n_out = number of keypoints found; if it exceeds capacity only the first `capacity` were written. */
int nrs_shi_configure(nrs_ctx* ctx, int32_t nms_window /* Options::non_max_suprresion_window_size, default 5 */);
int nrs_shi_extract(nrs_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                    const uint8_t* mask /* nullable: keep points where mask != 0 */, int32_t mask_stride,
                    int32_t n_prev, const float* prev_xy /* n_prev x 2 */,
                    int32_t capacity, float* out_xy /* capacity x 2 */, int32_t* out_id /* capacity */, int32_t* n_out);
/* parity tap: the extractor's buffers after the last call (h x w each, any pointer may be null) */
int nrs_shi_buffers(nrs_ctx* ctx, float* scores, int16_t* xgrad, int16_t* ygrad);

/* ---- multi-GPU: one deformable-BA window sharded over the GPUs of a node ---------
 * The reference solves LocalDeformableBundleAdjustment (g2o_optimization.cc:880-1161) as one g2o graph
 * in one thread; there is no reference interface for this -- it is the build's own extension of a3.
 * One process (or thread) per GPU, one context each.  After nrs_comm_init_* every rank calls
 * nrs_dba_upload with the SAME complete problem; each rank then runs the row kernels for its own
 * contiguous range of keyframes (nrs_shard_plan) and the ranks exchange, on the context's stream,
 *   - per linearisation / trial evaluation: one all-reduce (sum) of the pose blocks of the normal
 *     equations (H_pp 21 + b_p 6 doubles per keyframe) together with chi2, the LM scale and the
 *     max-diagonal slots, and the landmark rows of the boundary keyframes with the two neighbour ranks;
 *   - per PCG iteration: the boundary rows of the search direction and one all-reduce of 3 + 6 n_kf
 *     doubles (dot products, pose rows of the operator).
 * nrs_dba_reset / optimize / download / residuals are collective: every rank calls them in the same
 * order; they return the same trace and, after download, the same complete result on every rank.
 * RCCL is bound at run time (dlopen): librccl must be loadable only if nrs_comm_init_rccl is used.   */
#define NRS_COMM_ID_BYTES 128
int nrs_comm_unique_id(uint8_t* id, int32_t capacity /* >= NRS_COMM_ID_BYTES */);      /* rank 0; broadcast by the caller */
int nrs_comm_init_rccl(nrs_ctx* ctx, int32_t world, int32_t rank, const uint8_t* id, int32_t id_bytes);
int nrs_comm_rank(const nrs_ctx* ctx, int32_t* rank, int32_t* world);
/* Sizes of the BA problem resident on THIS rank: stats[0] padded landmark rows of the window, [1] rows whose incidence
 * records this rank packed and holds (sharded: the rows of its keyframe range), [2] / [3] spring / damper incidence
 * slots held, [4] device bytes of the problem.  (A rank also holds the per-row arrays -- state, PCG vectors, diagonal blocks --
 * of its own keyframes and one ghost keyframe either side only; they are addressed by the window's row index all the same.
 * nrs_dba_download and the residual taps gather through transient full-length scratch that is released when they return.) */
int nrs_dba_stats(nrs_ctx* ctx, int64_t stats[5]);
/* keyframe ranges: rank r owns keyframes kf_begin[r] .. kf_begin[r+1]-1 (balanced by padded landmark
 * rows, every rank at least one keyframe).  Host only, needs no device. */
int nrs_shard_plan(int32_t n_kf, int32_t n_lm, const int32_t* lm_kf, int32_t world, int32_t* kf_begin /* world+1 */);
/* Test harness: ranks are threads of one process, their contexts on the same GPU; rendezvous on the
 * host.  Runs the sharded arithmetic on a 1-GPU box. */
int nrs_local_group_create(int32_t world /* <= 8 */, void** group);
void nrs_local_group_destroy(void* group);
int nrs_comm_init_local(nrs_ctx* ctx, void* group, int32_t rank);

/* ---- a19 / a20: RegularizationGraph (modules/map/regularization_graph.{h,cc}) ------------------
 * Flat form of the graph: undirected edges carry the fields of RegularizationGraph::Edge
 * (regularization_graph.h:49-59); the raw CSR lists, per map point, its neighbours in ascending
 * index order (= the btree_map ID order GetEdges starts from) together with the undirected edge id.
 * Weights are (float)exp((double)arg) of the float argument -(d*d)/(2*sigma*sigma)
 * (utilities/geometry_toolbox.cc:26-28; see DESIGN.md "weights"). */
typedef struct {
    int32_t n_points;
    const int32_t* rowptr;    /* n_points+1 */
    const int32_t* col;       /* nnz = 2 * n_edges, ascending inside a row */
    const int32_t* eid;       /* nnz -> undirected edge */
    int32_t n_edges;
    float* e_w;               /* in/out: weight                */
    const float* e_d0;        /*         first_distance        */
    float* e_max;             /* in/out: max_distance          */
    float* e_min;             /* in/out: min_distance          */
    int32_t* e_status;        /* in/out: NRS_GRAPH_*           */
    float sigma;              /* options_.weight_sigma         */
    float stretch_th;         /* options_.streching_th (1.1)   */
} nrs_graph;

/* RegularizationGraph::GetEdges for every point (regularization_graph.cc:61-87): neighbours sorted
 * by (status asc, weight desc, index asc) and cut at the first weight below min_weight =
 * InterpolationWeight(1.5 sigma, sigma).  o_col / o_eid need room for nnz entries. */
int nrs_graph_select_neighbours(nrs_ctx* ctx, const nrs_graph* g, int32_t* o_rowptr, int32_t* o_col,
                                int32_t* o_eid);

/* RegularizationGraph::UpdateVertex for the listed points (regularization_graph.cc:89-146): every
 * edge of such a point is refreshed from pos (n_points x 3 last world positions); good_count[i] =
 * connections of ids[i] that pass the stretch test. */
int nrs_graph_update(nrs_ctx* ctx, nrs_graph* g, const float* pos, int32_t n_ids, const int32_t* ids,
                     int32_t* good_count);

/* ---- a19 / a20 at the reference's density: RegularizationGraph as a device-resident object ----------------------
 * The reference graph is all-pairs (Map::InitializeRegularizationGraph, modules/map/map.cc:148-166; graph growth
 * modules/mapping/mapping.cc:240-256): every vertex has N - 1 connections, UpdateVertex counts ALL of them (its
 * return value feeds the caller's "fewer than 5 good connections -> BAD", g2o_optimization.cc:468-473), so a radius
 * cut-off is not equivalent.  nrs_rgraph keeps the dense edge state in HBM (capacity^2 x 13 bytes: 5k points =
 * 325 MB) and replaces the bodies of the class' methods (modules/map/regularization_graph.cc):
 *   nrs_rgraph_create     RegularizationGraph(Options&, Map*) :27-31       point indices are 0 .. capacity-1
 *   nrs_rgraph_set_sigma  SetSigma :33-36
 *   nrs_rgraph_add_edges  AddEdge :38-55 for every (new, other) pair, new != other, relative position pos[other] -
 *                         pos[new]; all-pairs initialisation: new = other = every initial point
 *   nrs_rgraph_update     UpdateVertex :130-146 for the listed points; good_count[i] = its return value
 *   nrs_rgraph_get_edges  GetEdges :71-87 for the listed points: (status asc, weight desc, index asc), cut at the first
 *                         weight below min_weight; fixed-stride outputs [n_ids][cap_per_point] + count[n_ids] = the
 *                         FULL length of the list, of which the first min(count, cap_per_point) entries are written
 *                         (the reference's callers walk a prefix: 11 accepted neighbours or the first BAD edge)
 *   nrs_rgraph_edge       GetEdge :57-59 (out = weight, first, max, min distance; status -1 = no such edge)
 * pos is capacity x 3 floats (positions by point index; rows of unlisted points are not read). */
typedef struct nrs_rgraph nrs_rgraph;
int nrs_rgraph_create(nrs_ctx* ctx, int32_t capacity, float sigma, float stretch_th, nrs_rgraph** out);
void nrs_rgraph_destroy(nrs_rgraph* g);
int nrs_rgraph_set_sigma(nrs_rgraph* g, float sigma);
float nrs_rgraph_min_weight(const nrs_rgraph* g);
int nrs_rgraph_add_edges(nrs_rgraph* g, const float* pos, int32_t n_new, const int32_t* new_ids, int32_t n_other,
                         const int32_t* other_ids);
int nrs_rgraph_update(nrs_rgraph* g, const float* pos, int32_t n_ids, const int32_t* ids, int32_t* good_count);
int nrs_rgraph_get_edges(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, int32_t cap_per_point, int32_t* count,
                         int32_t* col, float* w, float* d0, int32_t* status);
int nrs_rgraph_edge(nrs_rgraph* g, int32_t i, int32_t j, float out[4], int32_t* status);
/* parity tap: rows of the dense state, n_ids x capacity each (status 255 = no edge; any pointer may be null) */
int nrs_rgraph_rows(nrs_rgraph* g, int32_t n_ids, const int32_t* ids, float* maxd, float* mind, float* d0, uint8_t* status);

/* ---- f2: DeformableTriangulation, batched (modules/optimization/g2o_optimization.cc:559-814) --------------------
 * One call triangulates every candidate feature of a frame (the reference calls the function once per candidate from
 * Mapping::LandmarkTriangulation, modules/mapping/mapping.cc:65-116).  Input = the TemporalBuffer flattened
 * (modules/map/temporal_buffer.h:43-63): n_frames <= 21 snapshots, oldest first; per snapshot its
 * camera_transform_world (Sophus::SE3f as qx qy qz qw tx ty tz) and, per keypoint id 0 .. n_ids-1, whether the id has
 * a keypoint there (keypoint_tracks) and a landmark position (mapppoint_tracks_); last_status = keypoint_tracks_status
 * of the LAST snapshot (GetClosestMapPointsToFeature looks for TRACKED_WITH_3D neighbours there).
 * out_status: 0 ok, else the InternalError the reference returns: 1 "Feature too close to other ones." 2 / 3 "High
 * reprojection error at first / second camera." 4 "Low parallax." 5 "Found no neighbours in a temporal point."
 * 6 "Negative initial depth." 7 "Optimization is empty." 8 "Triangulation has to many bad neighbors." 9 "Triangulation
 * has to much error." 10 the caller's "Short track" (track shorter than min_track; mapping.cc:88 uses 5).
 * out_xyz: the triangulated world position (zeros unless status 0).  out_debug (nullable, n_cand x 4): final chi2,
 * LM iterations, LM trials, regulariser edges.  The reprojection edge has no analytic Jacobian in the reference: it is
 * g2o's numeric one (delta 1e-9, central) through the fp32 projection, reproduced as such. */
int nrs_triangulate_batch(nrs_ctx* ctx, const nrs_camera* cam, int32_t n_frames, const float* poses /* n_frames x 7 */,
                          int32_t n_ids, const uint8_t* has_kp /* n_frames x n_ids */, const float* kp_xy /* .. x 2 */,
                          const uint8_t* has_lm, const float* lm_xyz /* .. x 3 */, const int32_t* last_status /* n_ids */,
                          int32_t n_cand, const int32_t* cand_ids, int32_t min_track, int32_t* out_status,
                          float* out_xyz /* n_cand x 3 */, double* out_debug);

/* ---- a2: CameraPoseAndDeformationOptimization (g2o_optimization.cc:148-557) ------------------
 * Frame side: n_f landmarks in frame index order with their map-point index (f_map, -1 = none),
 * LandmarkStatus (in/out), keypoint (f_uv) and position (f_pos, in/out).  Map side: the graph
 * (updated in place, OPT:458-474) and MapPoint::GetLastWorldPosition of every map point (map_pos,
 * in/out).  pose_qt in: frame pose; out: refined pose.  Outputs: Frame::SetDeformationMaginitud
 * value, and the re-located lost map points (lost needs room for n_points entries). */
int nrs_track_deform_solve(nrs_ctx* ctx, const nrs_camera* cam, nrs_graph* g, float* map_pos,
                           int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                           float* f_pos, double pose_qt[7], float scale, float* deform_median,
                           int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace);

/* The same function with the RegularizationGraph held on the device at the reference's all-pairs density (nrs_rgraph,
 * above): GetEdges (OPT:252, 496) and UpdateVertex (OPT:468) are served from it, so a point's good-connection count is
 * taken over all of its N - 1 connections as in the reference (the "fewer than 5 -> BAD" rule, OPT:470-473).
 * n_points = the graph's capacity = rows of map_pos; cap_per_point = how much of each point's GetEdges list is fetched at
 * first (if a walk of OPT:255-279 reaches the end of a truncated list, four times as much is fetched and the walks start
 * over: the result does not depend on cap_per_point, the time does).  Connections a walk passes over without any effect
 * are left out of the fetched lists on the device: in the lost-point stage (OPT:483-520) those to points that were not
 * optimised, in the embedded mode below those to optimised points that carry no vertex (BAD ones stay: they end a walk). */
int nrs_track_deform_solve_rg(nrs_ctx* ctx, const nrs_camera* cam, nrs_rgraph* g, int32_t n_points, int32_t cap_per_point,
                              float* map_pos, int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                              float* f_pos, double pose_qt[7], float scale, float* deform_median, int32_t* n_lost,
                              int32_t* lost, nrs_lm_trace* trace);

/* ---- N2: embedded deformation -- "points x graph nodes" as SURVEY.md 8(d) words it ------------------------------------
 * nrs_track_deform_solve_rg with a node set: f_node[i] != 0 marks the frame landmarks (among the TRACKED_WITH_3D ones) that carry
 * a free deformation; the regularisers of g2o_optimization.cc:255-335 are built between nodes only.  Every other TRACKED_WITH_3D
 * landmark is SKINNED: its deformation is sum_k omega_k delta_{n_k} over the <= 11 nodes its own GetEdges walk accepts (the walk of
 * OPT:255-279: stop after more than 10 accepted or at the first BAD connection; omega = connection weight / their sum), and its
 * reprojection edge (reprojection_error_with_deformation.cc:37-68: same residual, information, Huber kernel) constrains those nodes
 * and the pose with the Jacobian omega_k x the reference's block.  Rounds, inlier levels, IQR rejection, write-back and graph
 * update treat all optimised landmarks alike; the lost-point stage (OPT:476-553) follows nodes and skinned landmarks (the latter as
 * constants).  The reference has no such estimator -- its only skinning is that second stage -- so this mode is this build's,
 * stated in oracle/embedded_oracle.py; with every landmark a node it IS nrs_track_deform_solve_rg, bit for bit (tests).  Runs on
 * the direct solver (nrs_options.direct_solve != 2). */
int nrs_track_deform_solve_embedded(nrs_ctx* ctx, const nrs_camera* cam, nrs_rgraph* graph, int32_t n_points, int32_t cap_per_point,
                                    float* map_pos, int32_t n_f, const int32_t* f_map, int32_t* f_status, const float* f_uv,
                                    float* f_pos, const uint8_t* f_node, double pose_qt[7], float scale, float* deform_median,
                                    int32_t* n_lost, int32_t* lost, nrs_lm_trace* trace);


/* ---- a21-a23: LucasKanadeTracker (modules/matching/lucas_kanade_tracker.{h,cc}) ----------------
 * The context holds what the reference's tracker object holds: the reference points and, per point
 * and pyramid level, the cached 21x21 template (intensity x32 as int16, Scharr derivative as
 * 2 x int16, the two window means, a "filled" flag) -- members Iref_/Idref_/vMeanI_/vMeanI2_/prevPts_
 * (lucas_kanade_tracker.h:76-91).  Images are 8-bit single channel, row stride in bytes.
 * The pyramid is built on the device the way cv::buildOpticalFlowPyramid(img, pyr, Size(21,21),
 * maxLevel) builds it (LK:50,184; DESIGN.md "pyramid"). */
typedef struct {
    int32_t win_size;            /* 21 (the only size the reference uses, SLAM/system.cc:78)        */
    int32_t max_level;           /* 4                                                              */
    int32_t max_iters;           /* 10                                                             */
    float epsilon;               /* 1e-4, on |delta|^2                                             */
    float min_eig_threshold;     /* 1e-4                                                           */
} nrs_klt_config;

int nrs_klt_configure(nrs_ctx* ctx, const nrs_klt_config* cfg);     /* defaults = the values above */
int nrs_klt_clear(nrs_ctx* ctx);                                    /* LucasKanadeTracker::clear   */
int nrs_klt_num_points(nrs_ctx* ctx);

/* LucasKanadeTracker::SetReferenceImage (LK:47-168).  mask: NULL or w x h bytes, 0 = masked. */
int nrs_klt_set_reference(nrs_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride,
                          const uint8_t* mask, int32_t n, const float* xy);

/* LucasKanadeTracker::Track (LK:170-596).  xy in: initial guess (used when use_initial_flow),
 * out: tracked positions; status in/out: LandmarkStatus, only usable points are tracked and they
 * may become OUT_IMAGE_BOUNDARIES / BAD_FEATURE / BAD; n_good = points that pass the SSIM gate;
 * ssim (may be NULL): SSIM of every point that reached the gate. */
int nrs_klt_track(nrs_ctx* ctx, const uint8_t* img, int32_t w, int32_t h, int32_t stride, int32_t n,
                  float* xy, int32_t* status, int32_t use_initial_flow, float min_ssim, int32_t* n_good,
                  float* ssim);

/* GetPhotometricInformationOfPoint / InsertPhotometricInformation (LK:598-620).  Buffers hold
 * (max_level+1) levels: gray (levels x 441), grad (levels x 441 x 2), mean (levels x 2: meanI,
 * meanI2), valid (levels). */
int nrs_klt_get_template(nrs_ctx* ctx, int32_t idx, float xy[2], int16_t* gray, int16_t* grad,
                         float* mean, uint8_t* valid);
int nrs_klt_insert_template(nrs_ctx* ctx, const float xy[2], const int16_t* gray, const int16_t* grad,
                            const float* mean, const uint8_t* valid);

/* The same two calls for `count` consecutive points at once (no reference counterpart: the caller
 * loops, tracking.cc:203-209,383-391,457-460): buffers are point-major, i.e. the single-point layout
 * repeated; get reads points [first, first+count), insert appends. */
int nrs_klt_get_templates(nrs_ctx* ctx, int32_t first, int32_t count, float* xy, int16_t* gray, int16_t* grad,
                          float* mean, uint8_t* valid);
int nrs_klt_insert_templates(nrs_ctx* ctx, int32_t count, const float* xy, const int16_t* gray,
                             const int16_t* grad, const float* mean, const uint8_t* valid);

/* The same hand-over without the host in between.  The reference keeps a map point's photometric information in its Map
 * (GetPhotometricInformation at keyframes, tracking.cc:383-391) and inserts it into a tracker when the point is reused
 * (InsertPhotometricInformation, tracking.cc:457-460; PointReuse's own tracker, :422-448).  Here the context keeps an ARCHIVE in
 * device memory: nrs_klt_archive_templates copies the templates of tracker slots `slots[i]` into the archive entries `keys[i]`
 * (the caller's ids, e.g. map point ids; an entry is overwritten when archived again), nrs_klt_insert_archived appends the entries
 * `keys[i]` of context `src`'s archive to THIS context's tracker with the positions xy (n x 2).  Both contexts on one device; a
 * tracker with fewer pyramid levels than the archive takes the levels it has (PointReuse's tracker: maxLevel 1).  Byte for
 * byte what nrs_klt_get_templates + nrs_klt_insert_templates hand over (tests/test_gpu_klt.py).  The archive is DENSE by key (the key
 * indexes its entry: no lookup on the device), window^2 x levels x 6 bytes per key up to the largest key seen: keys must be
 * below 2^20 (NRS_ERR_INVALID otherwise); an archive that was built with another pyramid level count is dropped and starts over. */
int nrs_klt_archive_templates(nrs_ctx* ctx, int32_t n, const int32_t* slots, const int32_t* keys);
int nrs_klt_insert_archived(nrs_ctx* ctx, nrs_ctx* src, int32_t n, const int32_t* keys, const float* xy);

/* ---- N2: the skinned mode ("5k points x 500 graph nodes") --------------------------------------------------------
 * The reference has no separate node set; its skinning is stage 2 of CameraPoseAndDeformationOptimization
 * (modules/optimization/g2o_optimization.cc:476-553, spatial_regularizer_fixed.cc:32-43): points of the frame that are not
 * optimised follow <= 11 optimised graph neighbours.  Skinned mode = that algorithm with a chosen node set: the nodes
 * keep NRS_TRACKED_WITH_3D, every other point of the frame is handed over as NRS_TRACKED (in the frame, no 3D), and
 * nrs_track_deform_solve / nrs_track_deform_solve_rg run unchanged.  Coverage is the reference's, not "every point": stage 2
 * carries the points of the lost set, i.e. the non-node points met during some node's GetEdges walk before that walk stops
 * (11 accepted node neighbours or the first BAD edge, OPT:255-279).  A non-node point outside every node's walked prefix is
 * not in `lost`, keeps its old position and its NRS_TRACKED status: callers find them as "status NRS_TRACKED and not listed
 * in lost" (farthest point sampling with n_nodes >= n_points / 10 leaves < 10 % of them on the synthetic frames).  This call chooses the nodes: farthest point
 * sampling over the eligible points (eligible == NULL: all), first pick = lowest eligible index, fp32 squared distances,
 * ties to the lowest index.  node_ids[n_nodes] in pick order.  NRS_ERR_INVALID if fewer points are eligible. */
int nrs_skin_select_nodes(nrs_ctx* ctx, int32_t n_points, const float* pos /* n_points x 3 */, const uint8_t* eligible /* nullable */,
                          int32_t n_nodes, int32_t* node_ids);

#ifdef __cplusplus
}
#endif
#endif /* NRS_H */
