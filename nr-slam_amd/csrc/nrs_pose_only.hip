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
#include <algorithm>
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

// sums of 29 values over a 256-thread workgroup, totals to out[0..29): wave butterflies, then the four waves in order
__device__ inline void block_sum_29(const double* v, double* lds /* 4 x 29 */, int tid, double* out) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 29; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * 29 + k] = s;
    }
    __syncthreads();
    if (tid < 29) out[tid] = lds[tid] + lds[29 + tid] + lds[58 + tid] + lds[87 + tid];
}

enum { PH_EVAL = 0, PH_CLASSIFY = 1, PH_DONE = 2 };

// State of the serial part: g2o's Levenberg-Marquardt control flow (optimization_algorithm_levenberg.cpp:57-174) and the three rounds of
// OPT:103-141.  One thread runs it: in LDS for the single-workgroup kernel, in global memory for the multi-workgroup form.
struct PoState {
    Pose cur, bak;
    double H[21], b[6], x[6];
    double chi, lam, ni, rho;
    double R[9], t[3];                   // the pose the next pass evaluates at
    int round, it, qmax, ok, ntrials, niters, have_base, phase;
};

__device__ inline void po_publish_pose(PoState& S) {
    quat_to_R(S.cur.q, S.R);
    S.t[0] = S.cur.t[0]; S.t[1] = S.cur.t[1]; S.t[2] = S.cur.t[2];
}
__device__ inline void po_init(PoState& S, const PoseOnlyArgs& a) {
    S.cur = a.seed; S.bak = a.seed;
    for (int k = 0; k < 6; ++k) S.x[k] = 0;
    S.chi = 0; S.lam = -1; S.ni = 2; S.rho = 0;
    S.round = 0; S.it = 0; S.qmax = 0; S.ok = 1; S.ntrials = 0; S.niters = 0;
    S.have_base = 0;                     // 0: the coming eval is the iteration-0 linearisation
    po_publish_pose(S);
    S.phase = PH_EVAL;
}
// behind an evaluation pass (tot: 21 upper H, 6 b, chi2 of the active edges): the LM decision and, if the loop goes on, the next trial
__device__ inline void po_after_eval(PoState& S, const double* tot, int nactive, const PoseOnlyArgs& a) {
    int next = PH_EVAL;
    if (!S.have_base) {
        if (nactive == 0) {
            next = PH_CLASSIFY;          // optimize() returns -1: nothing to optimise
        } else {
            S.chi = tot[27];
            for (int k = 0; k < 21; ++k) S.H[k] = tot[k];
            for (int k = 0; k < 6; ++k) S.b[k] = tot[21 + k];
            S.have_base = 1;
            S.it = 0;
            // computeLambdaInit: tau * max diag
            double md = 0;
            int k = 0;
            for (int p = 0; p < 6; ++p) { md = fmax(md, fabs(S.H[k])); k += 6 - p; }
            S.lam = 1e-5 * md;
            S.ni = 2;
            S.qmax = 0;
        }
    } else {
        const double tempChi = S.ok ? tot[27] : 1.7976931348623157e308;
        double scale = 0;
        for (int k = 0; k < 6; ++k) scale += S.x[k] * (S.lam * S.x[k] + S.b[k]);
        scale += 1e-3;
        S.rho = (S.chi - tempChi) / scale;
        const bool accepted = (S.rho > 0) && isfinite(tempChi);
        if (S.ntrials < a.trace_cap) {
            nrs_lm_trial& T = a.trace[S.ntrials];
            T.round = S.round; T.iter = S.it; T.trial = S.qmax; T.accepted = accepted; T.solver_ok = S.ok;
            T.inner_iters = 0; T.early_rejected = 0; T.reserved = 0; T.lambda = S.lam; T.chi2 = S.chi; T.chi2_new = tempChi; T.rho = S.rho;
        }
        ++S.ntrials;
        bool lam_bad = false;
        if (accepted) {
            double alpha = 1.0 - (2 * S.rho - 1) * (2 * S.rho - 1) * (2 * S.rho - 1);
            alpha = fmin(alpha, 2.0 / 3.0);
            S.lam *= fmax(1.0 / 3.0, alpha);
            S.ni = 2;
            S.chi = tempChi;
            for (int k = 0; k < 21; ++k) S.H[k] = tot[k];
            for (int k = 0; k < 6; ++k) S.b[k] = tot[21 + k];
        } else {
            S.lam *= S.ni;
            S.ni *= 2;
            S.cur = S.bak;
            if (!isfinite(S.lam)) lam_bad = true;
        }
        if (!lam_bad) ++S.qmax;
        const bool again = !lam_bad && (S.rho < 0) && (S.qmax < 10);
        if (!again) {
            ++S.niters;
            const bool terminate = (S.qmax == 10) || (S.rho == 0) || !isfinite(S.lam);
            ++S.it;
            if (terminate || S.it == 10) next = PH_CLASSIFY;
            else S.qmax = 0;
        }
    }
    if (next == PH_EVAL) {
        // one trial: push, solve (H + lam I) x = b, update
        S.bak = S.cur;
        S.ok = chol6_solve(S.H, S.lam, S.b, S.x) ? 1 : 0;
        pose_oplus(S.cur, S.x);
    }
    po_publish_pose(S);
    S.phase = next;
}
// behind a classification pass (OPT:115-140): next round from the seed, or the end
__device__ inline void po_after_classify(PoState& S, const PoseOnlyArgs& a) {
    ++S.round;
    if (S.round == 3) {
        *a.pose_out = S.cur;
        a.counters[0] = S.ntrials;
        a.counters[1] = S.niters;
        S.phase = PH_DONE;
    } else {
        S.cur = a.seed;                  // OPT:108-110 restart from the seed
        S.have_base = 0;
        po_publish_pose(S);
        S.phase = PH_EVAL;
    }
}

// one point of an evaluation pass: residual, Huber, its share of the 6 x 6 normal equations (reprojection_error_only_pose.cc:50-75)
__device__ inline void po_eval_point(const PoseOnlyArgs& a, const double* R, const double* t, const float* pX, const float* pU, int i, double* acc) {
    const double X0 = pX[3 * i], X1 = pX[3 * i + 1], X2 = pX[3 * i + 2];
    const double px = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0];
    const double py = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
    const double pz = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
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
// one point of a classification pass: inliers keep the error stored by the last computeActiveErrors, outliers are re-evaluated at the
// final pose of the round (OPT:115-140)
__device__ inline void po_classify_point(const PoseOnlyArgs& a, const double* R, const double* t, const float* pX, const float* pU, int i) {
    double r0, r1;
    if (a.level[i] != 0) {
        const double X0 = pX[3 * i], X1 = pX[3 * i + 1], X2 = pX[3 * i + 2];
        const double px = R[0] * X0 + R[1] * X1 + R[2] * X2 + t[0];
        const double py = R[3] * X0 + R[4] * X1 + R[5] * X2 + t[1];
        const double pz = R[6] * X0 + R[7] * X1 + R[8] * X2 + t[2];
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

__global__ __launch_bounds__(PO_THREADS) void pose_only_kernel(PoseOnlyArgs a) {
    __shared__ double s_red[PO_WAVES][PO_NACC];
    __shared__ double s_tot[PO_NACC];
    __shared__ int s_nactive;
    __shared__ PoState S;                // (the serial lane's state lives in LDS so that it does not cost VGPRs in the other lanes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) po_init(S, a);
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
        const int phase = S.phase;
        if (phase == PH_DONE) break;
        double R[9], t[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = S.R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = S.t[k];
        if (phase == PH_EVAL) {
            // ---- computeActiveErrors + linearise + quadratic form over the active edges ----------
            double acc[PO_NACC];
#pragma unroll
            for (int k = 0; k < PO_NACC; ++k) acc[k] = 0;
            int nact = 0;
            for (int i = tid; i < a.n; i += PO_THREADS) {
                if (a.level[i] != 0) continue;
                ++nact;
                po_eval_point(a, R, t, pX, pU, i, acc);
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
            if (tid == 0) po_after_eval(S, s_tot, s_nactive, a);       // g2o's Levenberg-Marquardt control flow
            __syncthreads();
        } else {
            for (int i = tid; i < a.n; i += PO_THREADS) po_classify_point(a, R, t, pX, pU, i);
            __syncthreads();
            if (tid == 0) po_after_classify(S, a);
            __syncthreads();
        }
    }
}

// ---- the same solve over MANY workgroups, for frames whose passes one workgroup would serialise (a 100k-point map: ~200 points per
// thread and pass).  A pass (k_po_pass: every workgroup its share of the points, partial sums in a slot of its own) and a step
// (k_po_step: one workgroup adds the slots up in slot order -- bit-reproducible -- and runs the serial part) alternate on the stream; the
// state lives in global memory, launches behind PH_DONE are no-ops, the host looks at the phase once per batch of launches.
constexpr int POM_THREADS = 256;
__global__ __launch_bounds__(POM_THREADS) void k_po_pass(PoseOnlyArgs a, PoState* S, double* part /* gridDim x 32 */, int first) {
    __shared__ double lds[4 * 29];
    const int tid = threadIdx.x, gid = blockIdx.x * POM_THREADS + tid, stride = gridDim.x * POM_THREADS;
    if (first) { for (int i = gid; i < a.n; i += stride) a.level[i] = 0; }
    const int phase = first ? (int)PH_EVAL : S->phase;
    if (phase == PH_DONE) return;
    double R[9], t[3];
    if (first) { Pose p = a.seed; quat_to_R(p.q, R); t[0] = p.t[0]; t[1] = p.t[1]; t[2] = p.t[2]; }
    else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = S->R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = S->t[k];
    }
    if (phase == PH_EVAL) {
        double acc[29];
#pragma unroll
        for (int k = 0; k < 29; ++k) acc[k] = 0;
        for (int i = gid; i < a.n; i += stride) {
            if (!first && a.level[i] != 0) continue;
            acc[28] += 1.0;                                           // active edges (exact in fp64)
            po_eval_point(a, R, t, a.X, a.uv, i, acc);
        }
        block_sum_29(acc, lds, tid, part + (size_t)blockIdx.x * 32);
    } else {
        for (int i = gid; i < a.n; i += stride) po_classify_point(a, R, t, a.X, a.uv, i);
    }
}
__global__ __launch_bounds__(64) void k_po_step(PoseOnlyArgs a, PoState* S, const double* part, int n_part, int first, int* h_phase) {
    __shared__ double tot[32];
    const int tid = threadIdx.x;
    if (first && tid == 0) po_init(*S, a);
    __syncthreads();
    const int phase = S->phase;
    if (phase == PH_EVAL) {
        if (tid < 29) {
            double s = 0;
            for (int g = 0; g < n_part; ++g) s += part[(size_t)g * 32 + tid];
            tot[tid] = s;
        }
        __syncthreads();
        if (tid == 0) po_after_eval(*S, tot, (int)tot[28], a);
    } else if (phase == PH_CLASSIFY) {
        if (tid == 0) po_after_classify(*S, a);
    }
    if (tid == 0 && h_phase) { __threadfence_system(); *reinterpret_cast<volatile int*>(h_phase) = S->phase; }
}

}  // namespace nrs

using namespace nrs;

extern char** environ;
namespace nrs {
void DebugOpts::load_environment() {
    kv.clear();
    if (!environ) return;
    const char* list = nullptr;
    for (char** e = environ; *e; ++e) {
        if (strncmp(*e, "NRS_", 4) != 0) continue;
        const char* eq = strchr(*e, '=');
        if (!eq) continue;
        const std::string name(*e, eq - *e);
        if (name == "NRS_DEBUG") { list = eq + 1; continue; }
        kv.emplace_back(name, eq + 1);
    }
    for (const char* p = list; p && *p;) {                          // NRS_DEBUG="ND=0,NO_LDS=1": NAME=VALUE pairs, names without the NRS_ prefix
        const char* end = strchr(p, ',');
        const std::string item = end ? std::string(p, end - p) : std::string(p);
        const size_t eq = item.find('=');
        if (!item.empty()) set(("NRS_" + (eq == std::string::npos ? item : item.substr(0, eq))).c_str(), eq == std::string::npos ? "1" : item.c_str() + eq + 1);
        p = end ? end + 1 : nullptr;
    }
}
const char* process_debug_option(const char* name) { return DebugOpts::process().get(name); }
const DebugOpts& DebugOpts::process() {
    static const DebugOpts snap = [] { DebugOpts d; d.load_environment(); return d; }();
    return snap;
}
}  // namespace nrs

extern "C" int nrs_debug_set(nrs_ctx* c, const char* name, const char* value) {
    if (!c || !name || strncmp(name, "NRS_", 4) != 0) return NRS_ERR_INVALID;
    c->dbg.set(name, value);
    return NRS_OK;
}

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
        if (n < offsetof(nrs_options, pcg_rtol) || n > 4096) { delete c; return NRS_ERR_INVALID; }   // (shorter than the first field / an uninitialised word)
        memcpy(&c->opt, opt, std::min(n, sizeof(c->opt)));
        c->opt.struct_size = (uint32_t)sizeof(c->opt);
    }
    if (c->opt.direct_solve < 0 || c->opt.direct_solve > 2) c->opt.direct_solve = 0;
    if (c->opt.embedded_solver < 0 || c->opt.embedded_solver > 2) c->opt.embedded_solver = 0;
    if (c->opt.pcg_rtol <= 0) c->opt.pcg_rtol = 1e-10;
    c->dbg.load_environment();
    if (const char* e = c->env("NRS_PCG_RTOL")) { const double v = atof(e); if (v > 0) c->opt.pcg_rtol = v; }   // experiments only
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
    if (c->pin_spec_scal) (void)hipHostFree(c->pin_spec_scal);
    if (c->pin_spec_flags) (void)hipHostFree(c->pin_spec_flags);
    for (int j = 0; j < 3; ++j) {
        if (c->spec_stream[j]) { (void)hipStreamSynchronize(c->spec_stream[j]); (void)hipStreamDestroy(c->spec_stream[j]); }
        if (c->spec_join[j]) (void)hipEventDestroy(c->spec_join[j]);
    }
    if (c->spec_fork) (void)hipEventDestroy(c->spec_fork);
    for (int j = 0; j < 4; ++j) if (c->spec_back[j]) (void)hipEventDestroy(c->spec_back[j]);
    if (c->arena_dba.base) (void)hipFree(c->arena_dba.base);
    if (c->arena_trk.base) (void)hipFree(c->arena_trk.base);
    c->release(c->po_uv); c->release(c->po_X); c->release(c->po_err);
    c->release(c->po_level); c->release(c->po_out); c->release(c->po_trace); c->release(c->comm_flag); c->release(c->gather_ws); c->release(c->tap); c->release(c->pack_ws); c->release(c->pack_ws2); c->release(c->pack_ws3); c->release(c->pack_ws4); c->release(c->nd_skin); c->release(c->dba_skin); c->release(c->dba_kft); c->release(c->po_multi);
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
    // One workgroup runs the whole function in ONE launch while its passes are short (the points then sit in its LDS: <= 7168 points);
    // beyond that a pass is spread over the chip and the serial part runs in a kernel of its own between passes (k_po_pass / k_po_step:
    // ~10 us per LM trial whatever the size, where the single workgroup needs n / 512 point evaluations per thread and trial).
    // NRS_PO_MULTI_MIN moves the hand-over (tests run both forms on the same frames).
    int multi_min = 32768;                                    // (measured: 13.0 ms on one workgroup, 6.0 ms on many at 90k points; the two meet near 35k)
    if (const char* ev = c->env("NRS_PO_MULTI_MIN")) multi_min = atoi(ev);
    if (n >= multi_min && n > 0) {
        const int G = std::max(1, std::min(2 * c->prop.multiProcessorCount, (n + POM_THREADS - 1) / POM_THREADS));
        NRS_TRY(c->ensure(c->po_multi, sizeof(PoState) + 256 + sizeof(double) * 32 * (size_t)G));
        PoState* S = c->po_multi.as<PoState>();
        double* part = reinterpret_cast<double*>(c->po_multi.as<char>() + ((sizeof(PoState) + 255) & ~(size_t)255));
        a.cache_pts = 0;
        int phase = PH_EVAL, launched = 0;
        // (at most 3 rounds x (1 + 10 iterations x 10 trials) evaluation passes + 3 classification passes)
        while (phase != PH_DONE && launched < 320) {
            for (int q = 0; q < 16; ++q, ++launched) {
                hipLaunchKernelGGL(k_po_pass, dim3(G), dim3(POM_THREADS), 0, c->stream, a, S, part, launched == 0 ? 1 : 0);
                hipLaunchKernelGGL(k_po_step, dim3(1), dim3(64), 0, c->stream, a, S, part, G, launched == 0 ? 1 : 0, (int*)nullptr);
            }
            NRS_HIP(c, hipGetLastError());
            NRS_HIP(c, hipMemcpyAsync(&phase, &S->phase, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            NRS_HIP(c, hipStreamSynchronize(c->stream));
        }
        if (phase != PH_DONE) return c->fail(NRS_ERR_HIP, "pose-only solve: the pass / step sequence did not finish");
    } else {
        if (a.cache_pts && shm > 48 * 1024)
            NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(pose_only_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        hipLaunchKernelGGL(pose_only_kernel, dim3(1), dim3(PO_THREADS), a.cache_pts ? shm : 0, c->stream, a);
        NRS_HIP(c, hipGetLastError());
    }
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
