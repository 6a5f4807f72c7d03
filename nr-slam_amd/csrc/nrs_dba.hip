// a3: LocalDeformableBundleAdjustment C-ABI entry points (reference
// modules/optimization/g2o_optimization.cc:880-1161) on top of the graph LM engine.
#include <algorithm>
#include <vector>
#include "nrs_engine.hpp"

namespace nrs {

void dba_free(nrs_ctx* c) {
    if (c->dba) engine_destroy(c, c->dba);
    c->dba = nullptr;
}

// constants of OPT:958-973 (float arithmetic, widened to double like g2o does)
void ba_constants(EngineSpec& s, float scale) {
    const float th2 = sqrtf(5.99f), th3 = sqrtf(0.584f);
    const float sigma_rep = 0.5f, sigma_pos = 0.1f;
    const float sigma_spatial = (float)(0.1 * (double)scale);
    s.info_reproj = (double)(1.0f / (sigma_rep * sigma_rep));
    s.delta_reproj = (double)th2;
    s.info_pos = (double)(1.0f / (sigma_pos * sigma_pos));
    s.info_spatial = (double)(1.0f / (sigma_spatial * sigma_spatial));
    s.delta_spatial = (double)th3;
    s.k_spring = (double)1.1f;
}

}  // namespace nrs

using namespace nrs;

struct SkinIn { int32_t n = 0; const int32_t* kf = nullptr; const float* uv = nullptr; const float* xyz = nullptr; const int32_t* node = nullptr; const double* omega = nullptr; };

static int dba_upload(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                      int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                      int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                      int32_t n_dm, const int32_t* dm_idx, const float* dm_w, float scale, const SkinIn& sk) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || n_kf <= 0 || n_lm <= 0 || !poses_qt || !lm_xyz || !lm_kf || !lm_uv || n_sp < 0 || n_dm < 0 ||
        (n_sp > 0 && (!sp_ij || !sp_d0)) || (n_dm > 0 && (!dm_idx || !dm_w)))
        return c->fail(NRS_ERR_INVALID, "nrs_dba_upload: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    for (int64_t i = 0; i < 4 * (int64_t)n_dm; ++i)
        if (dm_idx[i] < 0) return c->fail(NRS_ERR_INVALID, "damper index out of range");
    dba_free(c);
    EngineSpec s;
    s.K = n_kf;
    s.M = n_lm;
    std::vector<Pose> poses(n_kf);
    for (int k = 0; k < n_kf; ++k) {
        for (int i = 0; i < 4; ++i) poses[k].q[i] = poses_qt[7 * k + i];
        for (int i = 0; i < 3; ++i) poses[k].t[i] = poses_qt[7 * k + 4 + i];
        quat_normalize(poses[k].q);                                  // SE3Quat ctor (se3quat.h:56-58)
    }
    std::vector<double> x(3 * (size_t)n_lm);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (double)lm_xyz[i];  // OPT:943 cast<double>
    std::vector<uint8_t> rflag(n_lm, RF_OBS | RF_REPROJ_ACTIVE);
    s.poses = poses.data();
    s.x = x.data();
    s.lm_pose = lm_kf;
    s.uv = lm_uv;
    s.rflag = rflag.data();
    s.n_sp = n_sp; s.sp_ij = sp_ij; s.sp_d0 = sp_d0;
    s.n_dm = n_dm; s.dm_idx = dm_idx; s.dm_w = dm_w;
    s.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) s.cam.p[i] = cam->params[i];
    ba_constants(s, scale);
    s.delta_pos = 0.0;                  // no robust kernel on the BA springs (OPT:1057-1071)
    s.spring_form = 0;                  // PositionRegularizer Jacobian as written (position_regularizer.cc:51-60)
    s.shard = true;                     // with a communicator on the context: one window over its ranks (include/nrs.h)
    std::vector<double> sk_X0;
    if (sk.n > 0) {                     // embedded window (N2b): observations of points without a vertex
        if (c->comm) return c->fail(NRS_ERR_STATE, "embedded BA windows are not sharded over a communicator");
        sk_X0.resize(3 * (size_t)sk.n);
        for (size_t i = 0; i < sk_X0.size(); ++i) sk_X0[i] = (double)sk.xyz[i];
        s.n_skin = sk.n; s.sk_uv = sk.uv; s.sk_X0 = sk_X0.data(); s.sk_node = sk.node; s.sk_om = sk.omega; s.sk_pose = sk.kf;
    }
    // rank-local checks and allocations can fail on one rank only: the ranks agree before the first collective
    int rc = engine_create(c, s, &c->arena_dba, &c->dba);
    // Every failure every rank sees alike (argument validation, more ranks than keyframes: all of them checked BEFORE any
    // rank-local work) returns without a collective; from there on a failure may be one rank's alone (err_local) and the ranks
    // agree on the outcome before the first collective of the solve.
    if (rc == NRS_OK || c->err_local) rc = comm_agree(c, rc);
    if (rc != NRS_OK) dba_free(c);
    return rc;
}

extern "C" int nrs_dba_upload(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                              int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                              int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                              int32_t n_dm, const int32_t* dm_idx, const float* dm_w, float scale) {
    return dba_upload(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, scale, SkinIn());
}

// ---- N2b: the embedded form of the window (include/nrs.h) -- vertices = the node copies, every other observation skinned
extern "C" int nrs_dba_upload_embedded(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt,
                                       int32_t n_lm, const float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                                       int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                                       int32_t n_dm, const int32_t* dm_idx, const float* dm_w,
                                       int32_t n_skin, const int32_t* sk_kf, const float* sk_uv, const float* sk_xyz,
                                       const int32_t* sk_node, const double* sk_omega, float scale) {
    if (!c) return NRS_ERR_INVALID;
    if (n_skin < 0 || (n_skin > 0 && (!sk_kf || !sk_uv || !sk_xyz || !sk_node || !sk_omega))) return c->fail(NRS_ERR_INVALID, "nrs_dba_upload_embedded: bad argument");
    for (int64_t i = 0; i < (int64_t)11 * n_skin; ++i)
        if (sk_node[i] < -1 || sk_node[i] >= n_lm) return c->fail(NRS_ERR_INVALID, "skinned observation: node copy out of range");
    SkinIn sk;
    sk.n = n_skin; sk.kf = sk_kf; sk.uv = sk_uv; sk.xyz = sk_xyz; sk.node = sk_node; sk.omega = sk_omega;
    return dba_upload(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, scale, sk);
}

extern "C" int nrs_dba_download_skinned(nrs_ctx* c, double* sk_xyz) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!sk_xyz) return c->fail(NRS_ERR_INVALID, "null output");
    return engine_skin_positions(c, c->dba, sk_xyz);
}

extern "C" int nrs_dba_stats(nrs_ctx* c, int64_t stats[5]) {
    if (!c || !stats) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    engine_stats(c->dba, stats);
    return NRS_OK;
}

extern "C" int nrs_shard_plan(int32_t n_kf, int32_t n_lm, const int32_t* lm_kf, int32_t world, int32_t* kf_begin) {
    if (n_kf <= 0 || n_lm < 0 || (n_lm > 0 && !lm_kf) || world < 1 || world > n_kf || !kf_begin) return NRS_ERR_INVALID;
    std::vector<int> cnt(n_kf, 0), grp(n_kf + 1, 0);
    for (int i = 0; i < n_lm; ++i) {
        if (lm_kf[i] < 0 || lm_kf[i] >= n_kf) return NRS_ERR_INVALID;
        cnt[lm_kf[i]]++;
    }
    for (int k = 0; k < n_kf; ++k) grp[k + 1] = grp[k] + std::max(1, (cnt[k] + 255) / 256);   // rows are padded to 256 per keyframe
    shard_plan(n_kf, grp.data(), world, kf_begin);
    return NRS_OK;
}

extern "C" int nrs_dba_reset(nrs_ctx* c) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    return engine_reset(c, c->dba);
}

extern "C" int nrs_dba_optimize(nrs_ctx* c, int32_t iters, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (trace) { trace->count = 0; trace->iterations = 0; }
    return engine_optimize(c, c->dba, iters, 0, trace);
}

static int download(nrs_ctx* c, int n_kf, double* poses_qt, double* xyz) {
    std::vector<Pose> poses(n_kf);
    NRS_TRY(engine_download(c, c->dba, poses.data(), xyz));
    if (poses_qt)
        for (int k = 0; k < n_kf; ++k) {
            for (int i = 0; i < 4; ++i) poses_qt[7 * k + i] = poses[k].q[i];
            for (int i = 0; i < 3; ++i) poses_qt[7 * k + 4 + i] = poses[k].t[i];
        }
    return NRS_OK;
}

extern "C" int nrs_dba_download(nrs_ctx* c, double* poses_qt, double* lm_xyz) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    return download(c, engine_num_poses(c->dba), poses_qt, lm_xyz);
}

extern "C" int nrs_dba_solve(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, double* poses_qt,
                             int32_t n_lm, float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                             int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                             int32_t n_dm, const int32_t* dm_idx, const float* dm_w,
                             float scale, int32_t iters, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    NRS_TRY(nrs_dba_upload(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, scale));
    NRS_TRY(nrs_dba_optimize(c, iters, trace));
    std::vector<double> xyz((size_t)n_lm * 3);
    NRS_TRY(download(c, n_kf, poses_qt, xyz.data()));
    for (size_t i = 0; i < xyz.size(); ++i) lm_xyz[i] = (float)xyz[i];      // OPT:1158 cast<float>
    return NRS_OK;
}

extern "C" int nrs_dba_solve_embedded(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, double* poses_qt,
                                      int32_t n_lm, float* lm_xyz, const int32_t* lm_kf, const float* lm_uv,
                                      int32_t n_sp, const int32_t* sp_ij, const float* sp_d0,
                                      int32_t n_dm, const int32_t* dm_idx, const float* dm_w,
                                      int32_t n_skin, const int32_t* sk_kf, const float* sk_uv, float* sk_xyz,
                                      const int32_t* sk_node, const double* sk_omega, float scale, int32_t iters, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    NRS_TRY(nrs_dba_upload_embedded(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf, lm_uv, n_sp, sp_ij, sp_d0, n_dm, dm_idx, dm_w, n_skin, sk_kf, sk_uv, sk_xyz,
                                    sk_node, sk_omega, scale));
    NRS_TRY(nrs_dba_optimize(c, iters, trace));
    std::vector<double> xyz((size_t)n_lm * 3), sk((size_t)n_skin * 3);
    NRS_TRY(download(c, n_kf, poses_qt, xyz.data()));
    if (n_skin > 0) NRS_TRY(engine_skin_positions(c, c->dba, sk.data()));
    for (size_t i = 0; i < xyz.size(); ++i) lm_xyz[i] = (float)xyz[i];      // OPT:1158 cast<float>
    for (size_t i = 0; i < sk.size(); ++i) sk_xyz[i] = (float)sk[i];
    return NRS_OK;
}

// LocalDeformableBundleAdjustment in ONE call, as mapping.cc:57 makes it: the edge construction of OPT:927-1137 on the device,
// then the device-side problem construction and the solve.  Windows that do not qualify take nrs_dba_build_edges + nrs_dba_solve.
static int window_upload(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, const double* poses_qt, const int32_t* kf_rowptr, const int32_t* kf_pt,
                         const float* lm_xyz, const float* lm_uv, int32_t n_points, const int32_t* nbr_rowptr, const int32_t* nbr_col, const float* nbr_w,
                         const float* nbr_d0, const int32_t* nbr_status, float scale, std::vector<int32_t>& lm_kf) {
    if (!cam || n_kf <= 0 || !poses_qt || !kf_rowptr || !kf_pt || !lm_xyz || !lm_uv || n_points <= 0 || !nbr_rowptr || !nbr_col || !nbr_w || !nbr_d0 || !nbr_status)
        return c->fail(NRS_ERR_INVALID, "nrs_dba_solve_window: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    const int32_t n_lm = kf_rowptr[n_kf];
    if (kf_rowptr[0] != 0 || n_lm <= 0) return c->fail(NRS_ERR_INVALID, "nrs_dba_solve_window: empty window");
    for (int k = 0; k < n_kf; ++k) if (kf_rowptr[k + 1] < kf_rowptr[k]) return c->fail(NRS_ERR_INVALID, "kf_rowptr must be non-decreasing");
    for (int32_t i = 0; i < n_lm; ++i) if (kf_pt[i] < 0 || kf_pt[i] >= n_points) return c->fail(NRS_ERR_INVALID, "map point index out of range");
    if (nbr_rowptr[0] != 0) return c->fail(NRS_ERR_INVALID, "nbr_rowptr must start at 0");
    for (int32_t p = 0; p < n_points; ++p) if (nbr_rowptr[p + 1] < nbr_rowptr[p]) return c->fail(NRS_ERR_INVALID, "nbr_rowptr must be non-decreasing");
    for (int32_t i = 0; i < nbr_rowptr[n_points]; ++i) if (nbr_col[i] < 0 || nbr_col[i] >= n_points) return c->fail(NRS_ERR_INVALID, "neighbour index out of range");
    lm_kf.resize(n_lm);
    for (int k = 0; k < n_kf; ++k) for (int32_t i = kf_rowptr[k]; i < kf_rowptr[k + 1]; ++i) lm_kf[i] = k;
    dba_free(c);
    EngineSpec s;
    s.K = n_kf; s.M = n_lm;
    std::vector<Pose> poses(n_kf);
    for (int k = 0; k < n_kf; ++k) {
        for (int i = 0; i < 4; ++i) poses[k].q[i] = poses_qt[7 * k + i];
        for (int i = 0; i < 3; ++i) poses[k].t[i] = poses_qt[7 * k + 4 + i];
        quat_normalize(poses[k].q);
    }
    std::vector<double> x(3 * (size_t)n_lm);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (double)lm_xyz[i];
    std::vector<uint8_t> rflag(n_lm, RF_OBS | RF_REPROJ_ACTIVE);
    s.poses = poses.data(); s.x = x.data(); s.lm_pose = lm_kf.data(); s.uv = lm_uv; s.rflag = rflag.data();
    s.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) s.cam.p[i] = cam->params[i];
    ba_constants(s, scale);
    s.delta_pos = 0.0; s.spring_form = 0; s.shard = true;
    s.n_sp = 1; s.n_dm = 1;                                          // (placeholders for the eligibility test: counts follow)
    if (!c->comm && engine_device_pack_ok(c, s)) {
        DevEdges de;
        NRS_HIP(c, hipSetDevice(c->device));
        NRS_TRY(engine_build_edges_device(c, n_kf, kf_rowptr, kf_pt, lm_kf.data(), n_points, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status, &de));
        if (de.n_sp > 0 && de.n_dm > 0) {
            s.n_sp = de.n_sp; s.sp_ij = de.sp_ij; s.sp_d0 = de.sp_d0;
            s.n_dm = de.n_dm; s.dm_idx = de.dm_idx; s.dm_w = de.dm_w;
            s.edges_on_device = true;
            const int rc = engine_create(c, s, &c->arena_dba, &c->dba);
            if (rc == NRS_OK) return NRS_OK;
            dba_free(c);
            if (rc != NRS_ERR_STATE) return rc;                      // (NRS_ERR_STATE: the window's halos exceed the device path's limits)
        }
    }
    // host path: nrs_dba_build_edges, then the upload as nrs_dba_upload makes it
    int32_t ns = 0, nd = 0;
    int rc = nrs_dba_build_edges(n_kf, kf_rowptr, kf_pt, n_points, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status, &ns, nullptr, nullptr, &nd, nullptr, nullptr);
    if (rc != NRS_OK) return c->fail(rc, "nrs_dba_build_edges failed");
    std::vector<int32_t> sp((size_t)2 * ns + 2), dm((size_t)4 * nd + 4);          // (+ slack: empty lists still need non-null arrays)
    std::vector<float> d0((size_t)ns + 1), dw((size_t)nd + 1);
    rc = nrs_dba_build_edges(n_kf, kf_rowptr, kf_pt, n_points, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status, &ns, sp.data(), d0.data(), &nd, dm.data(), dw.data());
    if (rc != NRS_OK) return c->fail(rc, "nrs_dba_build_edges failed");
    return nrs_dba_upload(c, cam, n_kf, poses_qt, n_lm, lm_xyz, lm_kf.data(), lm_uv, ns, sp.data(), d0.data(), nd, dm.data(), dw.data(), scale);
}

extern "C" int nrs_dba_solve_window(nrs_ctx* c, const nrs_camera* cam, int32_t n_kf, double* poses_qt, const int32_t* kf_rowptr, const int32_t* kf_pt,
                                    float* lm_xyz, const float* lm_uv, int32_t n_points, const int32_t* nbr_rowptr, const int32_t* nbr_col,
                                    const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status, float scale, int32_t iters, nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    std::vector<int32_t> lm_kf;
    NRS_TRY(window_upload(c, cam, n_kf, poses_qt, kf_rowptr, kf_pt, lm_xyz, lm_uv, n_points, nbr_rowptr, nbr_col, nbr_w, nbr_d0, nbr_status, scale, lm_kf));
    NRS_TRY(nrs_dba_optimize(c, iters, trace));
    const size_t n_lm = lm_kf.size();
    std::vector<double> xyz(n_lm * 3);
    NRS_TRY(download(c, n_kf, poses_qt, xyz.data()));
    for (size_t i = 0; i < xyz.size(); ++i) lm_xyz[i] = (float)xyz[i];      // OPT:1158 cast<float>
    return NRS_OK;
}

// parity tap of the device edge builder: the edge lists of the resident window (built by nrs_dba_solve_window on the device)
extern "C" int nrs_dba_window_edges(nrs_ctx* c, int32_t* n_spring, int32_t* sp_ij, float* sp_d0, int32_t* n_damper, int32_t* dm_idx, float* dm_w) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    int ns = 0, nd = 0;
    engine_edge_counts(c->dba, &ns, &nd);
    if (n_spring) *n_spring = ns;
    if (n_damper) *n_damper = nd;
    if (!sp_ij && !sp_d0 && !dm_idx && !dm_w) return NRS_OK;
    return engine_edges_to_host(c, c->dba, sp_ij, sp_d0, dm_idx, dm_w);
}

extern "C" int nrs_dba_residuals(nrs_ctx* c, double* r_reproj, double* r_spring, double* r_damper) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!r_reproj || !r_spring || !r_damper) return c->fail(NRS_ERR_INVALID, "null output");
    return engine_residuals(c, c->dba, r_reproj, r_spring, r_damper);
}

extern "C" int nrs_dba_gradient(nrs_ctx* c, double* b, double* diag) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!b || !diag) return c->fail(NRS_ERR_INVALID, "null output");
    return engine_gradient(c, c->dba, b, diag);
}

extern "C" int nrs_debug_kft(nrs_ctx* c, double lam, int32_t what, int32_t k, const double* in_d, double* out_d, int32_t* out_i) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if ((what == 0 || what == 4) ? !out_i : !out_d) return c->fail(NRS_ERR_INVALID, "null output");
    if (what == 3 && !in_d) return c->fail(NRS_ERR_INVALID, "null input");
    return engine_kft_debug(c, c->dba, lam, what, k, in_d, out_d, out_i);
}

extern "C" int nrs_dba_pack_hash(nrs_ctx* c, uint64_t* out) {
    if (!c) return NRS_ERR_INVALID;
    if (!c->dba) return c->fail(NRS_ERR_STATE, "no BA problem uploaded");
    if (!out) return c->fail(NRS_ERR_INVALID, "null output");
    return engine_pack_hash(c, c->dba, out);
}

// parity tap (include/nrs.h): an explicit one-pose / n-row block system through the engine's PCG kernels
extern "C" int nrs_debug_pcg_solve(nrs_ctx* c, int32_t n_rows, const double* Hpp21, const double* bp, const double* D6,
                                   const double* Hpl18, const double* bl, double lambda, double* x, int32_t* iters) {
    if (!c) return NRS_ERR_INVALID;
    if (n_rows <= 0 || !Hpp21 || !bp || !D6 || !Hpl18 || !bl || !x || !(lambda >= 0)) return c->fail(NRS_ERR_INVALID, "nrs_debug_pcg_solve: bad argument");
    EngineSpec s;
    s.K = 1; s.M = n_rows;
    Pose id;
    id.q[0] = id.q[1] = id.q[2] = 0; id.q[3] = 1; id.t[0] = id.t[1] = id.t[2] = 0;
    std::vector<double> xs(3 * (size_t)n_rows);
    for (int i = 0; i < n_rows; ++i) { xs[3 * i] = i; xs[3 * i + 1] = 0; xs[3 * i + 2] = 1; }
    std::vector<int> lm_pose(n_rows, 0);
    std::vector<float> uv(2 * (size_t)n_rows, 0.f);
    std::vector<uint8_t> rflag(n_rows, 0);
    s.poses = &id; s.x = xs.data(); s.lm_pose = lm_pose.data(); s.uv = uv.data(); s.rflag = rflag.data();
    s.cam.model = NRS_CAM_PINHOLE;
    for (int i = 0; i < 8; ++i) s.cam.p[i] = 1.f;
    ba_constants(s, 1.f);
    s.force_gather = true;
    Engine* e = nullptr;
    Arena arena;
    nrs::Comm* keep = c->comm;
    c->comm = nullptr;                                               // never sharded
    int rc = engine_create(c, s, &arena, &e);
    int it = 0, ok = 0;
    if (rc == NRS_OK) rc = engine_debug_solve(c, e, Hpp21, bp, D6, Hpl18, bl, lambda, x, x + 6, &it, &ok);
    if (e) engine_destroy(c, e);
    arena_release(&arena);
    c->comm = keep;
    if (iters) *iters = it;
    if (rc == NRS_OK && !ok) return c->fail(NRS_ERR_NUMERIC, "debug solve: not positive definite or not converged after %d iterations", it);
    return rc;
}

extern "C" int nrs_debug_nd_solve(nrs_ctx* c, int32_t n_nodes, const double* pos, const uint8_t* last, int32_t n_pairs, const int32_t* pairs,
                                  const double* Dn, const double* Vp, const double* bn, double lambda, int32_t repeats, double* x, int64_t* stats,
                                  double* ms_per_solve) {
    if (!c) return NRS_ERR_INVALID;
    if (n_nodes <= 0 || n_pairs < 0 || !pos || (n_pairs > 0 && (!pairs || !Vp)) || !Dn || !bn || !x || repeats < 0)
        return c->fail(NRS_ERR_INVALID, "nrs_debug_nd_solve: bad argument");
    return engine_nd_debug_solve(c, n_nodes, pos, last, n_pairs, pairs, Dn, Vp, bn, lambda, repeats, x, stats, ms_per_solve);
}

extern "C" int nrs_debug_nd_cache_stats(nrs_ctx* c, int64_t out[2]) {
    if (!c || !out) return NRS_ERR_INVALID;
    nd_cache_stats(c, out);
    return NRS_OK;
}
