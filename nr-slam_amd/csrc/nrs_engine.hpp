// Host-side interface of the graph Levenberg-Marquardt engine (nrs_engine.hip).
//
// The engine solves the reference's g2o problems on the GPU for one fixed vertex layout:
// K pose vertices (6 dof, VertexSE3Expmap) followed by M landmark / deformation vertices (3 dof,
// LandmarkVertex), with the reference's edge types:
//   * one optional reprojection edge per landmark vertex (ReprojectionError /
//     ReprojectionErrorWithDeformation), point = X0 + x
//   * springs   (PositionRegularizer / PositionRegularizerWithDeformation)  on (i, j)
//   * dampers   (SpatialRegularizer 4 vertices (1c,2c,1n,2n); the 2-vertex
//                SpatialRegularizerWithDeformation is the same edge with 1c = 2c = -1)
//   * unary dampers (SpatialRegularizerFixed): Jacobian on i only, j read as a value
// Levels and fixed flags are plain byte masks that the drivers rewrite between rounds.
#pragma once
#include "nrs_ctx.hpp"
#include "nrs_device.hpp"

namespace nrs {

enum : uint8_t { RF_OBS = 1, RF_REPROJ_ACTIVE = 2, RF_FIXED = 4 };

struct EngineSpec {
    int K = 0, M = 0;
    const Pose* poses = nullptr;          // K, already normalised
    const uint8_t* pose_fixed = nullptr;  // K or null (none fixed)
    const double* x = nullptr;            // M x 3 initial estimates
    const double* X0 = nullptr;           // M x 3 constant offset (tracking form) or null
    const int* lm_pose = nullptr;         // M, non-decreasing pose index of each vertex
    const float* uv = nullptr;            // M x 2
    const uint8_t* rflag = nullptr;       // M, RF_* bits
    int n_sp = 0;
    const int* sp_ij = nullptr;           // n_sp x 2
    const float* sp_d0 = nullptr;
    const uint8_t* sp_active = nullptr;   // level == 0, null = all
    int n_dm = 0;
    const int* dm_idx = nullptr;          // n_dm x 4 (1c,2c,1n,2n), -1 = absent vertex
    const float* dm_w = nullptr;
    const uint8_t* dm_active = nullptr;
    int n_un = 0;
    const int* un_ij = nullptr;           // n_un x 2: (vertex, value-only vertex)
    const float* un_w = nullptr;
    Cam cam;
    double info_reproj = 0, delta_reproj = 0, info_pos = 0, delta_pos = 0, info_spatial = 0, delta_spatial = 0;
    double k_spring = 0;
    int spring_form = 0;                  // 0: BA Jacobian as written, 1: tracking form
    // embedded-deformation mode (N2, nrs_engine_skin.hpp; single-frame engines on the direct solver only): observations of points
    // WITHOUT a vertex, placed at X0 + sum_k om[k] x[node[k]] over <= 11 vertices (nodes); they constrain those vertices and the pose
    int n_skin = 0;
    const float* sk_uv = nullptr;         // n_skin x 2
    const double* sk_X0 = nullptr;        // n_skin x 3
    const int* sk_node = nullptr;         // n_skin x 11 vertex indices, -1 pads
    const double* sk_om = nullptr;        // n_skin x 11 normalised weights
    const int* sk_pose = nullptr;         // n_skin pose index of every observation (null: pose 0); BA windows (K >= 1, PCG path): the point sits at
                                          // X0 + sum_k om[k] (x[node[k]] - x_start[node[k]]) -- N2b, oracle/embedded_oracle.py dba_solve_embedded
    bool shard = false;                   // split the poses over the ranks of the context's communicator (BA windows only)
    bool force_gather = false;            // stored-block operator (k_spmv gather path) instead of the LDS-staged factored one
    bool edges_on_device = false;         // sp_ij / sp_d0 / dm_idx / dm_w are DEVICE pointers (engine_build_edges_device): plain BA windows only
};

struct Engine;
struct DevEdges { int n_sp = 0, n_dm = 0; const int *sp_ij = nullptr, *dm_idx = nullptr; const float *sp_d0 = nullptr, *dm_w = nullptr; };

int engine_create(nrs_ctx* c, const EngineSpec& s, Arena* arena, Engine** out);
void engine_destroy(nrs_ctx* c, Engine* e);
int engine_update_flags(nrs_ctx* c, Engine* e, const uint8_t* rflag, const uint8_t* pose_fixed,
                        const uint8_t* sp_active, const uint8_t* dm_active);
int engine_reset(nrs_ctx* c, Engine* e);                                   // estimates <- initial values
int engine_optimize(nrs_ctx* c, Engine* e, int iters, int round, nrs_lm_trace* trace);
int engine_download(nrs_ctx* c, Engine* e, Pose* poses, double* x);      // caller vertex order
// fresh computeError() at the current estimate: chi2 = r^T Omega r of every edge
int engine_edge_chi2(nrs_ctx* c, Engine* e, double* reproj /*M*/, double* spring /*n_sp*/, double* damper /*n_dm*/);
int engine_residuals(nrs_ctx* c, Engine* e, double* r_reproj, double* r_spring, double* r_damper);
int engine_gradient(nrs_ctx* c, Engine* e, double* b, double* diag);
int engine_pack_hash(nrs_ctx* c, Engine* e, uint64_t* out /*24*/);
// embedded mode: levels of the skinned observations (1 = level 0) / their chi2 = r^T Omega r at the current estimate
int engine_skin_set_active(nrs_ctx* c, Engine* e, const uint8_t* active);
int engine_skin_chi2(nrs_ctx* c, Engine* e, double* chi /*n_skin*/);
int engine_skin_positions(nrs_ctx* c, Engine* e, double* xyz /*n_skin x 3*/);   // embedded BA windows: the skinned points at the current estimate
// OPT:927-1137's edge construction on the device (index for index what nrs_dba_build_edges returns); arrays live in ctx scratch
int engine_build_edges_device(nrs_ctx* c, int n_kf, const int* kf_rowptr, const int* kf_pt, const int* lm_kf, int n_points, const int* nbr_rowptr,
                              const int* nbr_col, const float* nbr_w, const float* nbr_d0, const int* nbr_status, DevEdges* out);
bool engine_device_pack_ok(nrs_ctx* c, const EngineSpec& s);      // would engine_create build this window on the device?
int engine_edges_to_host(nrs_ctx* c, Engine* e, int* sp_ij, float* sp_d0, int* dm_idx, float* dm_w);   // parity tap of the device edge builder    // checksums of the packed arrays (host- or device-built)       // solver order, caller vertex order
// parity tap: (H + lam I) x = b for explicitly given blocks (one pose, M landmark rows, no regularisers) through
// the engine's own PCG kernels; the engine must have been created with force_gather and K = 1
int engine_debug_solve(nrs_ctx* c, Engine* e, const double* Hpp21, const double* bp, const double* D6, const double* Hpl18,
                       const double* bl, double lam, double* xp, double* xl, int* iters, int* ok);
// parity tap of the direct (nested-dissection) solver on an explicit block system: include/nrs.h nrs_debug_nd_solve
int engine_nd_debug_solve(nrs_ctx* c, int n_nodes, const double* pos, const uint8_t* last, int n_pairs, const int* pairs, const double* Dn, const double* Vp,
                          const double* bn, double lam, int repeats, double* x, int64_t* stats, double* ms_per_solve);
int engine_kft_debug(nrs_ctx* c, Engine* e, double lam, int what, int k, const double* in_d, double* out_d, int32_t* out_i);   // include/nrs.h nrs_debug_kft
void nd_cache_stats(nrs_ctx* c, int64_t out[2]);                   // plans reused / built by the direct solver's cache
void arena_release(Arena* a);
void engine_stats(const Engine* e, int64_t stats[5]);              // rows, rows packed here, spring / damper incidence slots, device bytes
void shard_plan(int K, const int* grp_ptr, int world, int* kb);     // contiguous keyframe ranges, balanced by rows
int engine_num_poses(const Engine* e);
void engine_edge_counts(const Engine* e, int* n_sp, int* n_dm);
void ba_constants(EngineSpec& s, float scale);            // thresholds / informations of OPT:195-210,958-973

}  // namespace nrs
