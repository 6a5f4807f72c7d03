// a1: CameraPoseOptimization on MI355X (reference modules/optimization/g2o_optimization.cc:50-146).
//
// The problem is one 6-dof vertex with N unary reprojection edges: every LM trial is a full pass
// over N points followed by a 28-value reduction and a 6x6 solve, and the trials are strictly
// serial (g2o LM, third_party/g2o/g2o/core/optimization_algorithm_levenberg.cpp:57-174).  The
// whole 3-round x 10-iteration x <=10-trial schedule therefore runs inside ONE launch of ONE
// 1024-thread workgroup: 16 wave64s stride over the points (coalesced SoA-ish float2/float3
// loads that stay in L1/L2: 20 B/point), reduce with wave butterflies + LDS, and lane 0 runs
// the LM control flow, the Cholesky of H+lambda*I and the SE(3) retraction.  No host round trips,
// no inter-workgroup hand-offs; latency per trial is a few microseconds.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include "nrs_ctx.hpp"
#include "nrs_device.hpp"

namespace nrs {

constexpr int PO_THREADS = 512;
constexpr int PO_WAVES = PO_THREADS / 64;
constexpr int PO_NACC = 28;          // 21 upper H + 6 b + chi2

struct PoseOnlyArgs {
    const float* uv;
    const float* X;
    int n;
    Cam cam;
    Pose seed;
    double info;        // Omega = info * I2   (identity in the reference, OPT:90)
    double delta;       // Huber delta = (float)sqrt(5.99f)
    float th_sq;        // 5.99f
    double* err;        // n x 2, "stored _error" of every edge
    uint8_t* level;     // n, g2o edge level (0 active, 1 outlier)
    uint8_t* inlier;    // n, out
    Pose* pose_out;
    nrs_lm_trial* trace;
    int trace_cap;
    int* counters;      // [0] trials, [1] iterations
    int cache_pts;      // 1: uv/X of all points are cached in LDS (n * 20 B of dynamic LDS)
};

__device__ inline bool chol6_solve(const double* Hu /*21 upper, row-major packed*/, double lam,
                                   const double* b, double* x) {
    // fully unrolled so that A, L, y live in registers (dynamic indexing would put them in scratch
    // and every access of the serial LM lane would be a memory round trip)
    double A[6][6], L[6][6], y[6], xx[6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; ++k; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) A[i][i] += lam;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) L[i][j] = 0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k2 = 0; k2 < j; ++k2) d -= L[j][k2] * L[j][k2];
        if (!(d > 0.0)) ok = false;              // also catches NaN
        const double l = sqrt(d);
        L[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i][j];
#pragma unroll
            for (int k2 = 0; k2 < j; ++k2) s -= L[i][k2] * L[j][k2];
            L[i][j] = s / l;
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) s -= L[i][k2] * y[k2];
        y[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k2 = i + 1; k2 < 6; ++k2) s -= L[k2][i] * xx[k2];
        xx[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = xx[i];
    return true;
}

enum { PH_EVAL = 0, PH_CLASSIFY = 1, PH_DONE = 2 };

__global__ __launch_bounds__(PO_THREADS) void pose_only_kernel(PoseOnlyArgs a) {
    __shared__ double s_red[PO_WAVES][PO_NACC];
    __shared__ double s_tot[PO_NACC];
    __shared__ double s_R[9], s_t[3];
    __shared__ int s_phase;
    __shared__ int s_nactive;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // lane-0 LM state lives in LDS so that it does not cost VGPRs in the 1023 other lanes
    __shared__ Pose s_cur, s_bak;
    __shared__ double s_H[21], s_b[6], s_x[6];
    __shared__ double s_lm[4];           // chi, lam, ni, rho
    __shared__ int s_ctl[8];             // round, it, qmax, ok, ntrials, niters, have_base
    Pose& cur = s_cur; Pose& bak = s_bak;
    double* H = s_H; double* b = s_b; double* x = s_x;
    double& chi = s_lm[0]; double& lam = s_lm[1]; double& ni = s_lm[2]; double& rho = s_lm[3];
    int& round = s_ctl[0]; int& it = s_ctl[1]; int& qmax = s_ctl[2]; int& ok = s_ctl[3];
    int& ntrials = s_ctl[4]; int& niters = s_ctl[5]; int& have_base = s_ctl[6];
    if (tid == 0) {
        cur = a.seed; bak = a.seed;
        for (int k = 0; k < 6; ++k) x[k] = 0;
        chi = 0; lam = -1; ni = 2; rho = 0;
        round = 0; it = 0; qmax = 0; ok = 1; ntrials = 0; niters = 0;
        have_base = 0;                   // 0: the coming eval is the iteration-0 linearisation
    }

    if (tid == 0) {
        quat_to_R(cur.q, s_R);
        s_t[0] = cur.t[0]; s_t[1] = cur.t[1]; s_t[2] = cur.t[2];
        s_phase = PH_EVAL;
    }
    // the points are constant over all ~40 passes: keep them in LDS when they fit (the passes are
    // otherwise a chain of dependent L2 round trips per point)
    extern __shared__ float s_pts[];
    const float* pX = a.X;
    const float* pU = a.uv;
    if (a.cache_pts) {
        for (int i = tid; i < 3 * a.n; i += PO_THREADS) s_pts[i] = a.X[i];
        for (int i = tid; i < 2 * a.n; i += PO_THREADS) s_pts[3 * a.n + i] = a.uv[i];
        pX = s_pts;
        pU = s_pts + 3 * a.n;
    }
    for (int i = tid; i < a.n; i += PO_THREADS) a.level[i] = 0;
    __syncthreads();

    while (true) {
        const int phase = s_phase;
        if (phase == PH_DONE) break;

        if (phase == PH_EVAL) {
            // ---- computeActiveErrors + linearise + quadratic form over the active edges ----------
            double acc[PO_NACC];
#pragma unroll
            for (int k = 0; k < PO_NACC; ++k) acc[k] = 0;
            int nact = 0;
            const double R0 = s_R[0], R1 = s_R[1], R2 = s_R[2], R3 = s_R[3], R4 = s_R[4], R5 = s_R[5],
                         R6 = s_R[6], R7 = s_R[7], R8 = s_R[8], t0 = s_t[0], t1 = s_t[1], t2 = s_t[2];
            for (int i = tid; i < a.n; i += PO_THREADS) {
                if (a.level[i] != 0) continue;
                ++nact;
                const double X0 = pX[3 * i], X1 = pX[3 * i + 1], X2 = pX[3 * i + 2];
                const double px = R0 * X0 + R1 * X1 + R2 * X2 + t0;
                const double py = R3 * X0 + R4 * X1 + R5 * X2 + t1;
                const double pz = R6 * X0 + R7 * X1 + R8 * X2 + t2;
                float u, v, Jf[6];
                project_f32(a.cam, (float)px, (float)py, (float)pz, u, v);
                projection_jacobian_f32(a.cam, (float)px, (float)py, (float)pz, Jf);
                const double r0 = (double)pU[2 * i] - (double)u, r1 = (double)pU[2 * i + 1] - (double)v;
                a.err[2 * i] = r0;
                a.err[2 * i + 1] = r1;
                double rho0, rho1;
                huber(a.info * (r0 * r0 + r1 * r1), a.delta, rho0, rho1);
                acc[27] += rho0;
                const double w = rho1 * a.info;
                // J = -Jpi * [ -[p]x | I ]   (reprojection_error_only_pose.cc:60-74)
                double J[2][6];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                    J[rr][0] = -j1 * pz + j2 * py;
                    J[rr][1] = j0 * pz - j2 * px;
                    J[rr][2] = -j0 * py + j1 * px;
                    J[rr][3] = j0; J[rr][4] = j1; J[rr][5] = j2;
                }
                int k = 0;
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int q = p; q < 6; ++q) { acc[k] += w * (J[0][p] * J[0][q] + J[1][p] * J[1][q]); ++k; }
#pragma unroll
                for (int p = 0; p < 6; ++p) acc[21 + p] -= w * (J[0][p] * r0 + J[1][p] * r1);
            }
#pragma unroll
            for (int k = 0; k < PO_NACC; ++k) {
                const double s = wave_sum(acc[k]);
                if (lane == 0) s_red[wave][k] = s;
            }
            const int nact_w = (int)wave_sum((double)nact);
            if (tid == 0) s_nactive = 0;
            __syncthreads();
            if (lane == 0) atomicAdd(&s_nactive, nact_w);
            if (tid < PO_NACC) {
                double s = 0;
                for (int w2 = 0; w2 < PO_WAVES; ++w2) s += s_red[w2][tid];
                s_tot[tid] = s;
            }
            __syncthreads();

            if (tid == 0) {
                // ---- g2o Levenberg-Marquardt control flow -------------------------------------
                int next = PH_EVAL;
                if (!have_base) {
                    if (s_nactive == 0) {
                        next = PH_CLASSIFY;          // optimize() returns -1: nothing to optimise
                    } else {
                        chi = s_tot[27];
                        for (int k = 0; k < 21; ++k) H[k] = s_tot[k];
                        for (int k = 0; k < 6; ++k) b[k] = s_tot[21 + k];
                        have_base = 1;
                        it = 0;
                        // computeLambdaInit: tau * max diag
                        double md = 0;
                        int k = 0;
                        for (int p = 0; p < 6; ++p) { md = fmax(md, fabs(H[k])); k += 6 - p; }
                        lam = 1e-5 * md;
                        ni = 2;
                        qmax = 0;
                    }
                } else {
                    const double tempChi = ok ? s_tot[27] : 1.7976931348623157e308;
                    double scale = 0;
                    for (int k = 0; k < 6; ++k) scale += x[k] * (lam * x[k] + b[k]);
                    scale += 1e-3;
                    rho = (chi - tempChi) / scale;
                    const bool accepted = (rho > 0) && isfinite(tempChi);
                    if (ntrials < a.trace_cap) {
                        nrs_lm_trial& T = a.trace[ntrials];
                        T.round = round; T.iter = it; T.trial = qmax; T.accepted = accepted; T.solver_ok = ok;
                        T.inner_iters = 0; T.early_rejected = 0; T.reserved = 0; T.lambda = lam; T.chi2 = chi; T.chi2_new = tempChi; T.rho = rho;
                    }
                    ++ntrials;
                    bool lam_bad = false;
                    if (accepted) {
                        double alpha = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                        alpha = fmin(alpha, 2.0 / 3.0);
                        lam *= fmax(1.0 / 3.0, alpha);
                        ni = 2;
                        chi = tempChi;
                        for (int k = 0; k < 21; ++k) H[k] = s_tot[k];
                        for (int k = 0; k < 6; ++k) b[k] = s_tot[21 + k];
                    } else {
                        lam *= ni;
                        ni *= 2;
                        cur = bak;
                        if (!isfinite(lam)) lam_bad = true;
                    }
                    if (!lam_bad) ++qmax;
                    const bool again = !lam_bad && (rho < 0) && (qmax < 10);
                    if (!again) {
                        ++niters;
                        const bool terminate = (qmax == 10) || (rho == 0) || !isfinite(lam);
                        ++it;
                        if (terminate || it == 10) next = PH_CLASSIFY;
                        else qmax = 0;
                    }
                }
                if (next == PH_EVAL) {
                    // one trial: push, solve (H + lam I) x = b, update
                    bak = cur;
                    ok = chol6_solve(H, lam, b, x) ? 1 : 0;
                    pose_oplus(cur, x);
                }
                quat_to_R(cur.q, s_R);
                s_t[0] = cur.t[0]; s_t[1] = cur.t[1]; s_t[2] = cur.t[2];
                s_phase = next;
            }
            __syncthreads();
        } else {
            // ---- PH_CLASSIFY: OPT:115-140.  Inliers keep the error stored by the last
            //      computeActiveErrors; outliers are re-evaluated at the final pose of the round.
            const double R0 = s_R[0], R1 = s_R[1], R2 = s_R[2], R3 = s_R[3], R4 = s_R[4], R5 = s_R[5],
                         R6 = s_R[6], R7 = s_R[7], R8 = s_R[8], t0 = s_t[0], t1 = s_t[1], t2 = s_t[2];
            for (int i = tid; i < a.n; i += PO_THREADS) {
                double r0, r1;
                if (a.level[i] != 0) {
                    const double X0 = pX[3 * i], X1 = pX[3 * i + 1], X2 = pX[3 * i + 2];
                    const double px = R0 * X0 + R1 * X1 + R2 * X2 + t0;
                    const double py = R3 * X0 + R4 * X1 + R5 * X2 + t1;
                    const double pz = R6 * X0 + R7 * X1 + R8 * X2 + t2;
                    float u, v;
                    project_f32(a.cam, (float)px, (float)py, (float)pz, u, v);
                    r0 = (double)pU[2 * i] - (double)u;
                    r1 = (double)pU[2 * i + 1] - (double)v;
                    a.err[2 * i] = r0;
                    a.err[2 * i + 1] = r1;
                } else {
                    r0 = a.err[2 * i];
                    r1 = a.err[2 * i + 1];
                }
                const float chi2 = (float)(a.info * (r0 * r0 + r1 * r1));
                const bool out = chi2 > a.th_sq;
                a.level[i] = out ? 1 : 0;
                a.inlier[i] = out ? 0 : 1;
            }
            __syncthreads();
            if (tid == 0) {
                ++round;
                if (round == 3) {
                    *a.pose_out = cur;
                    a.counters[0] = ntrials;
                    a.counters[1] = niters;
                    s_phase = PH_DONE;
                } else {
                    cur = a.seed;                      // OPT:108-110 restart from the seed
                    have_base = 0;
                    quat_to_R(cur.q, s_R);
                    s_t[0] = cur.t[0]; s_t[1] = cur.t[1]; s_t[2] = cur.t[2];
                    s_phase = PH_EVAL;
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace nrs

using namespace nrs;

extern "C" void nrs_options_init(nrs_options* opt) {
    if (!opt) return;
    memset(opt, 0, sizeof(*opt));
    opt->device = -1;
    opt->struct_size = (uint32_t)sizeof(*opt);
}

extern "C" int nrs_create(nrs_ctx** out, const nrs_options* opt) {
    if (!out) return NRS_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NRS_ERR_NO_DEVICE;
    nrs_ctx* c = new (std::nothrow) nrs_ctx();
    if (!c) return NRS_ERR_ALLOC;
    memset(&c->opt, 0, sizeof(c->opt));
    c->opt.device = -1;
    if (opt) {
        // the caller's struct may be shorter than this build's (an older header): its bytes only, defaults behind them
        size_t n = opt->struct_size ? opt->struct_size : offsetof(nrs_options, direct_solve);
        if (n < offsetof(nrs_options, pcg_rtol)) { delete c; return NRS_ERR_INVALID; }
        memcpy(&c->opt, opt, std::min(n, sizeof(c->opt)));
        c->opt.struct_size = (uint32_t)sizeof(c->opt);
    }
    if (c->opt.direct_solve < 0 || c->opt.direct_solve > 2) c->opt.direct_solve = 0;
    if (c->opt.pcg_rtol <= 0) c->opt.pcg_rtol = 1e-10;
    if (const char* e = getenv("NRS_PCG_RTOL")) { const double v = atof(e); if (v > 0) c->opt.pcg_rtol = v; }   // experiments only
    if (c->opt.pcg_max_iters <= 0) c->opt.pcg_max_iters = 2000;
    if (c->opt.pcg_batch <= 0) c->opt.pcg_batch = 8;
    c->err[0] = 0;
    memset(&c->prof, 0, sizeof(c->prof));
    int dev = c->opt.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) { delete c; return NRS_ERR_NO_DEVICE; }
    }
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess) { delete c; return NRS_ERR_NO_DEVICE; }
    c->device = dev;
    if (hipGetDeviceProperties(&c->prop, dev) != hipSuccess) { delete c; return NRS_ERR_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return NRS_ERR_NO_DEVICE; }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return NRS_ERR_NO_DEVICE;
    }
    *out = c;
    return NRS_OK;
}

extern "C" void nrs_destroy(nrs_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    nrs::dba_free(c);
    nrs::klt_free(c);
    nrs::shi_free(c);
    nrs::comm_free(c);
    if (c->pin_scal) (void)hipHostFree(c->pin_scal);
    if (c->pin_flags) (void)hipHostFree(c->pin_flags);
    if (c->arena_dba.base) (void)hipFree(c->arena_dba.base);
    if (c->arena_trk.base) (void)hipFree(c->arena_trk.base);
    c->release(c->po_uv); c->release(c->po_X); c->release(c->po_err);
    c->release(c->po_level); c->release(c->po_out); c->release(c->po_trace); c->release(c->comm_flag); c->release(c->tap); c->release(c->pack_ws); c->release(c->pack_ws2); c->release(c->pack_ws3); c->release(c->pack_ws4); c->release(c->nd_skin); c->release(c->dba_skin);
    nrs::nd_cache_free(c);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* nrs_last_error(const nrs_ctx* c) { return c ? c->err : "null context"; }

extern "C" int nrs_device_name(const nrs_ctx* c, char* buf, int32_t len) {
    if (!c || !buf || len <= 0) return NRS_ERR_INVALID;
    snprintf(buf, (size_t)len, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return NRS_OK;
}

extern "C" int nrs_get_profile(const nrs_ctx* c, nrs_profile* out) {
    if (!c || !out) return NRS_ERR_INVALID;
    *out = c->prof;
    return NRS_OK;
}

extern "C" int nrs_reset_profile(nrs_ctx* c) {
    if (!c) return NRS_ERR_INVALID;
    memset(&c->prof, 0, sizeof(c->prof));
    return NRS_OK;
}

extern "C" void* nrs_stream(nrs_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int nrs_pose_only_solve(nrs_ctx* c, const nrs_camera* cam, int32_t n, const float* uv,
                                   const float* X, double pose_qt[7], uint8_t* inlier,
                                   nrs_lm_trace* trace) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || n < 0 || !pose_qt || (n > 0 && (!uv || !X))) return c->fail(NRS_ERR_INVALID, "nrs_pose_only_solve: bad argument");
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    NRS_HIP(c, hipSetDevice(c->device));
    const int cap = trace && trace->trials ? trace->capacity : 0;
    NRS_TRY(c->ensure(c->po_uv, sizeof(float) * 2 * (size_t)n + 16));
    NRS_TRY(c->ensure(c->po_X, sizeof(float) * 3 * (size_t)n + 16));
    NRS_TRY(c->ensure(c->po_err, sizeof(double) * 2 * (size_t)n + 16));
    NRS_TRY(c->ensure(c->po_level, 2 * (size_t)n + 16));
    NRS_TRY(c->ensure(c->po_out, sizeof(Pose) + 4 * sizeof(int)));
    NRS_TRY(c->ensure(c->po_trace, sizeof(nrs_lm_trial) * (size_t)(cap > 0 ? cap : 1)));
    if (n > 0) {
        NRS_HIP(c, hipMemcpyAsync(c->po_uv.p, uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(c->po_X.p, X, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    }
    PoseOnlyArgs a;
    a.uv = c->po_uv.as<float>();
    a.X = c->po_X.as<float>();
    a.n = n;
    a.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) a.cam.p[i] = cam->params[i];
    for (int i = 0; i < 4; ++i) a.seed.q[i] = pose_qt[i];
    for (int i = 0; i < 3; ++i) a.seed.t[i] = pose_qt[4 + i];
    quat_normalize(a.seed.q);                       // SE3Quat ctor normalizeRotation (se3quat.h:56-58)
    a.info = 1.0;                                   // OPT:90
    const float th_sq = 5.99f;                      // OPT:63-64
    a.delta = (double)sqrtf(th_sq);
    a.th_sq = th_sq;
    a.err = c->po_err.as<double>();
    a.level = c->po_level.as<uint8_t>();
    a.inlier = c->po_level.as<uint8_t>() + n;
    a.pose_out = c->po_out.as<Pose>();
    a.counters = reinterpret_cast<int*>(c->po_out.as<char>() + sizeof(Pose));
    a.trace = c->po_trace.as<nrs_lm_trial>();
    a.trace_cap = cap;
    const size_t shm = (size_t)n * 20;
    a.cache_pts = shm <= 140 * 1024 ? 1 : 0;                 // 160 KiB of LDS per CU on gfx950
    if (a.cache_pts && shm > 48 * 1024)
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pose_only_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    hipLaunchKernelGGL(pose_only_kernel, dim3(1), dim3(PO_THREADS), a.cache_pts ? shm : 0, c->stream, a);
    NRS_HIP(c, hipGetLastError());
    Pose out;
    int counters[2] = {0, 0};
    NRS_HIP(c, hipMemcpyAsync(&out, a.pose_out, sizeof(Pose), hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(counters, a.counters, sizeof(counters), hipMemcpyDeviceToHost, c->stream));
    if (inlier && n > 0) NRS_HIP(c, hipMemcpyAsync(inlier, a.inlier, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (trace) {
        trace->count = counters[0];
        trace->iterations = counters[1];
        const int ncopy = counters[0] < cap ? counters[0] : cap;
        if (ncopy > 0) NRS_HIP(c, hipMemcpy(trace->trials, a.trace, sizeof(nrs_lm_trial) * (size_t)ncopy, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < 4; ++i) pose_qt[i] = out.q[i];
    for (int i = 0; i < 3; ++i) pose_qt[4 + i] = out.t[i];
    for (int i = 0; i < 7; ++i)
        if (!std::isfinite(pose_qt[i])) return c->fail(NRS_ERR_NUMERIC, "pose-only solve produced a non-finite pose");
    return NRS_OK;
}
