// f2: batched DeformableTriangulation (reference modules/optimization/g2o_optimization.cc:559-814, called once per
// candidate feature from Mapping::LandmarkTriangulation, modules/mapping/mapping.cc:65-116).
//
// The reference builds, per candidate, a g2o graph of <= 21 LandmarkVertex (one per buffered frame of the feature's
// track, in that frame's CAMERA coordinates), one ReprojectionErrorOnlyDeformation per vertex (no analytic Jacobian:
// g2o's central difference with delta = 1e-9 THROUGH the fp32 projection, base_fixed_sized_edge.hpp:159-200) and one
// SpatialRegularizerWithObservation per (frame pair, neighbour) -- up to 210 x 11 edges -- and runs optimize(10).
// Here: one wave per candidate, the whole function in one launch.
//   * the <= 63 x 63 normal equations live in LDS; the regulariser part is a graph Laplacian (every edge of a frame
//     pair has the Jacobian (+I, -I) as written, spatial_regularizer_with_observation.cc:49-52), so the edges of a
//     pair are aggregated once into {count, sum of flows, sum of |flow|^2} and never visited again by the LM loop;
//   * Levenberg-Marquardt exactly as optimization_algorithm_levenberg.cpp:57-174, the dense solve by an in-LDS
//     Cholesky (the reference: BlockSolverX + LinearSolverEigen);
//   * the gates before (mid-point triangulation, reprojection, parallax, depth seeds) and after the solve (bad
//     neighbours, reprojection error) and GetClosestMapPointsToFeature (temporal_buffer.cc:97-141) are part of the launch.
// The fp32 steps use the operation order of oracle/triang_oracle.py (contraction off).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include "nrs_ctx.hpp"
#include "nrs_device.hpp"

namespace nrs {

constexpr int TR_MAXF = 21;                 // TemporalBuffer size (SLAM/system.cc:42: 20, + the current frame)
constexpr int TR_MAXN = 11;                 // `size() > num_neighbors` with num_neighbors = 10
constexpr int TR_N = 3 * TR_MAXF;           // unknowns
constexpr int TR_LD = TR_N + 1;             // leading dimension of the LDS matrices
constexpr int TR_PAIRS = TR_MAXF * (TR_MAXF - 1) / 2;
enum { TR_OK = 0, TR_CLOSE, TR_REPROJ1, TR_REPROJ2, TR_PARALLAX, TR_NO_NEIGHBOUR, TR_NEG_DEPTH, TR_EMPTY, TR_BAD_NEIGHBOURS,
       TR_BAD_ERROR, TR_SHORT };

struct TriArgs {
    Cam cam;
    int F, n, n_cand, min_track;
    const float* poses;          // F x 7: camera_transform_world (qx qy qz qw tx ty tz), Sophus::SE3f
    const uint8_t* has_kp;       // F x n
    const float* kp_xy;          // F x n x 2
    const uint8_t* has_lm;       // F x n
    const float* lm_xyz;         // F x n x 3
    const int* status;           // n: LandmarkStatus in the last snapshot
    const int* cand;             // n_cand
    int* o_status;               // n_cand
    float* o_xyz;                // n_cand x 3
    double* o_dbg;               // n_cand x 4 or null: final chi2, LM iterations, trials, regulariser edges
};

// ---- Sophus SE3f in float (so3.hpp:388-395, se3.hpp:222-225), contraction off
struct Se3f { float q[4], t[3]; };
__device__ inline void crossf(const float* a, const float* b, float* o) {
#pragma clang fp contract(off)
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline void so3_point(const float* q, const float* p, float* o) {
#pragma clang fp contract(off)
    float uv[3], c[3];
    crossf(q, p, uv);
    uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
    crossf(q, uv, c);
    o[0] = p[0] + q[3] * uv[0] + c[0]; o[1] = p[1] + q[3] * uv[1] + c[1]; o[2] = p[2] + q[3] * uv[2] + c[2];
}
__device__ inline void se3_point(const Se3f& T, const float* p, float* o) {
#pragma clang fp contract(off)
    so3_point(T.q, p, o);
    o[0] = o[0] + T.t[0]; o[1] = o[1] + T.t[1]; o[2] = o[2] + T.t[2];
}
__device__ inline Se3f se3_inv(const Se3f& T) {
#pragma clang fp contract(off)
    Se3f r;
    r.q[0] = -T.q[0]; r.q[1] = -T.q[1]; r.q[2] = -T.q[2]; r.q[3] = T.q[3];
    const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
    so3_point(r.q, nt, r.t);
    return r;
}
__device__ inline Se3f se3_mul(const Se3f& A, const Se3f& B) {
#pragma clang fp contract(off)
    Se3f r;
    const float* a = A.q; const float* b = B.q;
    r.q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r.q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r.q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    r.q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    float rt[3];
    so3_point(A.q, B.t, rt);
    r.t[0] = rt[0] + A.t[0]; r.t[1] = rt[1] + A.t[1]; r.t[2] = rt[2] + A.t[2];
    return r;
}
__device__ inline float normf3(const float* v) {
#pragma clang fp contract(off)
    return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

// CameraModel::Unproject (pin_hole.cc:33-38, kannala_brandt_8.cc:53-85)
__device__ inline void unproject_f32(const Cam& c, float u, float v, float* ray) {
#pragma clang fp contract(off)
    const float x = (u - c.p[2]) / c.p[0], y = (v - c.p[3]) / c.p[1];
    if (c.model == 0) { ray[0] = x; ray[1] = y; ray[2] = 1.f; return; }
    const float theta_d = sqrtf(x * x + y * y);
    float th = 0.f;
    if (theta_d > 1e-8f) {
        float theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const float t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
            const float a = c.p[4] * t2, b = c.p[5] * t4, cc = c.p[6] * t6, d = c.p[7] * t8;
            const float fix = (theta * (1.f + a + b + cc + d) - theta_d) / (1.f + 3.f * a + 5.f * b + 7.f * cc + 9.f * d);
            theta = theta - fix;
            if (fabsf(fix) < 1e-6f) break;
        }
        th = theta;
    }
    const float s = (float)sin((double)th), co = (float)cos((double)th);
    ray[0] = s * x / theta_d; ray[1] = s * y / theta_d; ray[2] = co;
}

__device__ inline double wave_sum_all(double v) { return wave_sum(v); }
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(64) void k_triangulate(TriArgs A) {
    extern __shared__ double sm[];
    double* H = sm;                               // TR_N x TR_LD
    double* L = H + TR_N * TR_LD;                 // TR_N x TR_LD (Cholesky work copy)
    double* bvec = L + TR_N * TR_LD;              // TR_N
    double* dx = bvec + TR_N;                     // TR_N
    double* xv = dx + TR_N;                       // TR_N   estimates (camera coordinates)
    double* xb = xv + TR_N;                       // TR_N   backup
    double* pw = xb + TR_N;                       // TR_N   world positions T_wc x
    double* Tq = pw + TR_N;                       // TR_MAXF x 4  world_transform_camera (double)
    double* Tt = Tq + 4 * TR_MAXF;                // TR_MAXF x 3
    double* pairF = Tt + 3 * TR_MAXF;             // TR_PAIRS x 5: count, sum flow (3), sum |flow|^2
    double* uvd = pairF + 5 * TR_PAIRS;           // TR_MAXF x 2 observations
    int* nb = reinterpret_cast<int*>(uvd + 2 * TR_MAXF);      // TR_MAXN neighbour ids
    int* fr = nb + TR_MAXN + 1;                   // TR_MAXF frames of the track
    const int lane = threadIdx.x, ci = blockIdx.x;
    if (ci >= A.n_cand) return;
    const int cand = A.cand[ci];
    const int last = A.F - 1, n = A.n;
    auto finish = [&](int code, float x, float y, float z) {
        if (lane == 0) {
            A.o_status[ci] = code;
            A.o_xyz[3 * ci] = x; A.o_xyz[3 * ci + 1] = y; A.o_xyz[3 * ci + 2] = z;
        }
    };
    if (A.o_dbg && lane < 4) A.o_dbg[4 * ci + lane] = 0;
    // ================= GetClosestMapPointsToFeature(candidate, 10, 20, 500) on the last snapshot
    const float cx = A.kp_xy[2 * ((size_t)last * n + cand)], cy = A.kp_xy[2 * ((size_t)last * n + cand) + 1];
    auto dist_to = [&](int j) -> float {
        const double ddx = (double)(cx - A.kp_xy[2 * ((size_t)last * n + j)]), ddy = (double)(cy - A.kp_xy[2 * ((size_t)last * n + j) + 1]);
        return (float)sqrt(ddx * ddx + ddy * ddy);                // cv::norm(Point2f) -> double, stored as float
    };
    auto eligible = [&](int j) { return j != cand && A.has_kp[(size_t)last * n + j] && A.status[j] == 0; };
    int close = 0;
    for (int j = lane; j < n; j += 64)
        if (eligible(j)) { const float d = dist_to(j); if (!(d > 500.f) && d < 20.f) close = 1; }
    close = __any(close);
    int n_nb = 0;
    {
        float last_d = -1.f;
        int last_j = -1;
        for (int k = 0; k < TR_MAXN; ++k) {                       // k-th smallest (distance, id) pair
            float bd = FLT_MAX;
            int bj = 0x7fffffff;
            for (int j = lane; j < n; j += 64) {
                if (!eligible(j)) continue;
                const float d = dist_to(j);
                if (d > 500.f) continue;
                const bool after = d > last_d || (d == last_d && j > last_j);
                if (after && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(bd, off, 64);
                const int oj = __shfl_xor(bj, off, 64);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            if (bj == 0x7fffffff) break;
            if (lane == 0) nb[k] = bj;
            last_d = bd; last_j = bj;
            ++n_nb;
        }
    }
    if (close || n_nb == 0) { finish(TR_CLOSE, 0, 0, 0); return; }
    // ================= GetFeatureTrack
    int V = 0;
    for (int f = 0; f < A.F; ++f)
        if (A.has_kp[(size_t)f * n + cand]) { if (lane == 0) fr[V] = f; ++V; }
    __syncthreads();
    if (V < A.min_track) { finish(TR_SHORT, 0, 0, 0); return; }          // mapping.cc:88: TrackLenght(candidate) >= 5
    const int first = fr[0], lastf = fr[V - 1];
    auto pose_of = [&](int f) { Se3f T; for (int k = 0; k < 4; ++k) T.q[k] = A.poses[7 * f + k]; for (int k = 0; k < 3; ++k) T.t[k] = A.poses[7 * f + 4 + k]; return T; };
    // ================= rigid gates (every lane computes them: uniform control flow, no broadcast)
    {
#pragma clang fp contract(off)
        const float* kc = A.kp_xy + 2 * ((size_t)first * n + cand);
        const float* kp = A.kp_xy + 2 * ((size_t)lastf * n + cand);
        float cr[3], pr[3];
        unproject_f32(A.cam, kc[0], kc[1], cr);
        unproject_f32(A.cam, kp[0], kp[1], pr);
        float nn = normf3(cr); cr[0] /= nn; cr[1] /= nn; cr[2] /= nn;
        nn = normf3(pr); pr[0] /= nn; pr[1] /= nn; pr[2] /= nn;
        const Se3f Tc = pose_of(first), Tp = pose_of(lastf);
        // TriangulateMidPoint(previous_ray, current_ray, previous_T, current_T)  (geometry_toolbox.cc:45-79)
        float f0[3] = {pr[0], pr[1], pr[2]}, f1[3] = {cr[0], cr[1], cr[2]};
        nn = normf3(f0); f0[0] /= nn; f0[1] /= nn; f0[2] /= nn;
        nn = normf3(f1); f1[0] /= nn; f1[1] /= nn; f1[2] /= nn;
        const Se3f T10 = se3_mul(Tc, se3_inv(Tp));
        const float x = T10.q[0], y = T10.q[1], z = T10.q[2], w = T10.q[3];
        const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
                    tyy = ty * y, tyz = tz * y, tzz = tz * z;
        const float R[9] = {1.f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.f - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.f - (txx + tyy)};
        float Rf0[3];
        for (int i = 0; i < 3; ++i) Rf0[i] = (R[3 * i] * f0[0] + R[3 * i + 1] * f0[1]) + R[3 * i + 2] * f0[2];
        float p[3], q[3], r[3];
        crossf(Rf0, f1, p); crossf(Rf0, T10.t, q); crossf(f1, T10.t, r);
        const float nq = normf3(q), nr = normf3(r), np_ = normf3(p);
        const float s1 = nq / (nq + nr), s2 = nr / np_;
        float x1[3];
        for (int i = 0; i < 3; ++i) x1[i] = s1 * (T10.t[i] + s2 * (Rf0[i] + f1[i]));
        float X[3];
        se3_point(se3_inv(Tc), x1, X);
        float pc[3], u, v;
        se3_point(Tc, X, pc);
        project_f32(A.cam, pc[0], pc[1], pc[2], u, v);
        float ex = kc[0] - u, ey = kc[1] - v;
        if ((double)(ex * ex + ey * ey) > 5.991) { finish(TR_REPROJ1, 0, 0, 0); return; }
        se3_point(Tp, X, pc);
        project_f32(A.cam, pc[0], pc[1], pc[2], u, v);
        ex = kp[0] - u; ey = kp[1] - v;
        if ((double)(ex * ex + ey * ey) > 5.991) { finish(TR_REPROJ2, 0, 0, 0); return; }
        const Se3f Tci = se3_inv(Tc), Tpi = se3_inv(Tp);
        const float n1[3] = {X[0] - Tci.t[0], X[1] - Tci.t[1], X[2] - Tci.t[2]}, n2[3] = {X[0] - Tpi.t[0], X[1] - Tpi.t[1], X[2] - Tpi.t[2]};
        const float dot = (n1[0] * n2[0] + n1[1] * n2[1]) + n1[2] * n2[2];
        const float cs = dot / (normf3(n1) * normf3(n2));
        const float par = (float)acos((double)((1.f < cs) ? 1.f : cs));          // std::min(cs, 1.f): a NaN cosine stays NaN and passes the gate, as in the reference
        if ((double)par < 0.0025 * 5.0) { finish(TR_PARALLAX, 0, 0, 0); return; }
    }
    // ================= depth seeds, vertices, world_transform_camera per vertex (lane v)
    int code = 0;
    if (lane < V) {
#pragma clang fp contract(off)
        const int f = fr[lane];
        const Se3f T = pose_of(f);
        float depth = 0.f;
        int cnt = 0;
        for (int k = 0; k < n_nb; ++k) {
            const int j = nb[k];
            if (A.has_lm[(size_t)f * n + j]) {
                float pc[3];
                se3_point(T, A.lm_xyz + 3 * ((size_t)f * n + j), pc);
                depth = depth + pc[2];
                ++cnt;
            }
        }
        if (cnt == 0) code = TR_NO_NEIGHBOUR;
        else {
            depth = depth / (float)cnt;
            if (depth < 0.f) code = TR_NEG_DEPTH;
        }
        const float* kp = A.kp_xy + 2 * ((size_t)f * n + cand);
        float ray[3];
        unproject_f32(A.cam, kp[0], kp[1], ray);
        for (int k = 0; k < 3; ++k) xv[3 * lane + k] = (double)(ray[k] * depth);
        uvd[2 * lane] = (double)kp[0]; uvd[2 * lane + 1] = (double)kp[1];
        const Se3f Ti = se3_inv(T);
        double q[4] = {(double)Ti.q[0], (double)Ti.q[1], (double)Ti.q[2], (double)Ti.q[3]};
        quat_normalize(q);                                               // SE3Quat constructor (se3quat.h:56-58)
        for (int k = 0; k < 4; ++k) Tq[4 * lane + k] = q[k];
        for (int k = 0; k < 3; ++k) Tt[3 * lane + k] = (double)Ti.t[k];
    }
    {   // the first failing frame (in track order) decides the message
        int first_bad = code ? lane : 64;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, off, 64));
        if (first_bad < 64) { finish(__shfl(code, first_bad, 64), 0, 0, 0); return; }
    }
    // ================= regulariser edges, aggregated per frame pair (a < b)
    const int n_pairs = V * (V - 1) / 2;
    double n_edges = 0;
    for (int pi = lane; pi < n_pairs; pi += 64) {
        int a = 0, rem = pi;
        while (rem >= V - 1 - a) { rem -= V - 1 - a; ++a; }
        const int b = a + 1 + rem;
        double cnt = 0, f0 = 0, f1 = 0, f2 = 0, s2 = 0;
        for (int k = 0; k < n_nb; ++k) {
            const int j = nb[k];
            if (A.has_lm[(size_t)fr[a] * n + j] && A.has_lm[(size_t)fr[b] * n + j] && A.has_lm[(size_t)first * n + j]) {
                const float* xa = A.lm_xyz + 3 * ((size_t)fr[a] * n + j);
                const float* xb_ = A.lm_xyz + 3 * ((size_t)fr[b] * n + j);
                const double g0 = (double)(xb_[0] - xa[0]), g1 = (double)(xb_[1] - xa[1]), g2 = (double)(xb_[2] - xa[2]);   // flow (float), cast<double>
                cnt += 1; f0 += g0; f1 += g1; f2 += g2; s2 += g0 * g0 + g1 * g1 + g2 * g2;
            }
        }
        double* P = pairF + 5 * pi;
        P[0] = cnt; P[1] = f0; P[2] = f1; P[3] = f2; P[4] = s2;
        n_edges += cnt;
    }
    n_edges = wave_sum_all(n_edges);
    __syncthreads();
    const int N = 3 * V;
    const double info_r = (double)(1.0f / (0.5f * 0.5f)), info_s = (double)(1.0f / (0.1f * 0.1f));
    auto pair_index = [&](int a, int b) { return a * (V - 1) - a * (a - 1) / 2 + (b - a - 1); };   // a < b
    // world positions + chi2 of the current estimates
    auto eval_chi = [&]() -> double {
        if (lane < V) {
            const double* q = Tq + 4 * lane;
            const double* x = xv + 3 * lane;
            double o[3];
            quat_rotate(q, x, o);
            pw[3 * lane] = o[0] + Tt[3 * lane]; pw[3 * lane + 1] = o[1] + Tt[3 * lane + 1]; pw[3 * lane + 2] = o[2] + Tt[3 * lane + 2];
        }
        __syncthreads();
        double chi = 0;
        if (lane < V) {
            float u, v;
            project_f32(A.cam, (float)xv[3 * lane], (float)xv[3 * lane + 1], (float)xv[3 * lane + 2], u, v);
            const double r0 = uvd[2 * lane] - (double)u, r1 = uvd[2 * lane + 1] - (double)v;
            chi = info_r * (r0 * r0 + r1 * r1);
        }
        for (int pi = lane; pi < n_pairs; pi += 64) {
            const double* P = pairF + 5 * pi;
            if (P[0] == 0) continue;
            int a = 0, rem = pi;
            while (rem >= V - 1 - a) { rem -= V - 1 - a; ++a; }
            const int b = a + 1 + rem;
            const double d0 = pw[3 * b] - pw[3 * a], d1 = pw[3 * b + 1] - pw[3 * a + 1], d2 = pw[3 * b + 2] - pw[3 * a + 2];
            chi += info_s * (P[4] - 2.0 * (d0 * P[1] + d1 * P[2] + d2 * P[3]) + P[0] * (d0 * d0 + d1 * d1 + d2 * d2));
        }
        chi = wave_sum_all(chi);
        __syncthreads();
        return chi;
    };
    // ================= optimize(10): OptimizationAlgorithmLevenberg::solve
    double lam = -1, ni = 2, chi = 0;
    int iters = 0, trials = 0;
    for (int it = 0; it < 10; ++it) {
        chi = eval_chi();
        // ---- buildSystem: H (dense, LDS), b
        for (int i = lane; i < N * TR_LD; i += 64) H[i] = 0;
        __syncthreads();
        if (lane < V) {
            const int v = lane;
            // regularisers: row block v of the Laplacian, and its part of b
            double diag = 0, g0 = 0, g1 = 0, g2 = 0;
            for (int o = 0; o < V; ++o) {
                if (o == v) continue;
                const int a = v < o ? v : o, b = v < o ? o : v;
                const double* P = pairF + 5 * pair_index(a, b);
                if (P[0] == 0) continue;
                diag += info_s * P[0];
                for (int k = 0; k < 3; ++k) H[(3 * v + k) * TR_LD + 3 * o + k] = -info_s * P[0];
                // sum of residuals of the pair's edges: sum flow - n (p_b - p_a); b_a -= Omega sum r, b_b += Omega sum r
                const double s0 = P[1] - P[0] * (pw[3 * b] - pw[3 * a]), s1 = P[2] - P[0] * (pw[3 * b + 1] - pw[3 * a + 1]),
                             s2 = P[3] - P[0] * (pw[3 * b + 2] - pw[3 * a + 2]);
                const double sg = v == a ? -1.0 : 1.0;
                g0 += sg * info_s * s0; g1 += sg * info_s * s1; g2 += sg * info_s * s2;
            }
            // reprojection: numeric Jacobian, delta = 1e-9 central through the fp32 projection (base_fixed_sized_edge.hpp:159-200)
            const double delta = 1e-9, scalar = 1 / (2 * delta);
            double J[2][3];
            float u, w_;
            for (int d = 0; d < 3; ++d) {
                double xp[3] = {xv[3 * v], xv[3 * v + 1], xv[3 * v + 2]}, xm[3] = {xv[3 * v], xv[3 * v + 1], xv[3 * v + 2]};
                xp[d] += delta;
                xm[d] += -delta;
                project_f32(A.cam, (float)xp[0], (float)xp[1], (float)xp[2], u, w_);
                const double ep0 = uvd[2 * v] - (double)u, ep1 = uvd[2 * v + 1] - (double)w_;
                project_f32(A.cam, (float)xm[0], (float)xm[1], (float)xm[2], u, w_);
                const double em0 = uvd[2 * v] - (double)u, em1 = uvd[2 * v + 1] - (double)w_;
                J[0][d] = scalar * (ep0 - em0);
                J[1][d] = scalar * (ep1 - em1);
            }
            project_f32(A.cam, (float)xv[3 * v], (float)xv[3 * v + 1], (float)xv[3 * v + 2], u, w_);
            const double r0 = uvd[2 * v] - (double)u, r1 = uvd[2 * v + 1] - (double)w_;
            for (int p = 0; p < 3; ++p) {
                for (int q = 0; q < 3; ++q)
                    H[(3 * v + p) * TR_LD + 3 * v + q] = (p == q ? diag : 0.0) + info_r * (J[0][p] * J[0][q] + J[1][p] * J[1][q]);
                bvec[3 * v + p] = (p == 0 ? g0 : p == 1 ? g1 : g2) - info_r * (J[0][p] * r0 + J[1][p] * r1);
            }
        }
        __syncthreads();
        if (it == 0) {
            double md = 0;
            for (int i = lane; i < N; i += 64) md = fmax(md, fabs(H[i * TR_LD + i]));
            lam = 1e-5 * wave_max(md);
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            for (int i = lane; i < N; i += 64) xb[i] = xv[i];                          // push
            // ---- (H + lam I) dx = b by Cholesky in LDS (lower triangle of L)
            for (int i = lane; i < N * TR_LD; i += 64) L[i] = H[i];
            __syncthreads();
            for (int i = lane; i < N; i += 64) L[i * TR_LD + i] += lam;
            __syncthreads();
            bool ok = true;
            for (int k = 0; k < N; ++k) {
                const double dkk = L[k * TR_LD + k];
                if (!(dkk > 0)) { ok = false; break; }                                    // uniform: every lane reads the same value
                const double lkk = sqrt(dkk);
                __syncthreads();
                if (lane == 0) L[k * TR_LD + k] = lkk;
                for (int i = k + 1 + lane; i < N; i += 64) L[i * TR_LD + k] /= lkk;
                __syncthreads();
                for (int i = k + 1 + lane; i < N; i += 64) {
                    const double lik = L[i * TR_LD + k];
                    for (int j = k + 1; j <= i; ++j) L[i * TR_LD + j] -= lik * L[j * TR_LD + k];
                }
                __syncthreads();
            }
            if (ok) {
                // two triangular solves, column oriented: lane j finishes unknown j, every lane below (above) it takes its
                // share of column j -- 63 short steps instead of 63^2 dependent LDS round trips on one lane
                for (int i = lane; i < N; i += 64) dx[i] = bvec[i];
                __syncthreads();
                for (int j = 0; j < N; ++j) {
                    if (lane == 0) dx[j] = dx[j] / L[j * TR_LD + j];
                    __syncthreads();
                    const double xj = dx[j];
                    for (int i = j + 1 + lane; i < N; i += 64) dx[i] -= L[i * TR_LD + j] * xj;
                    __syncthreads();
                }
                for (int j = N - 1; j >= 0; --j) {
                    if (lane == 0) dx[j] = dx[j] / L[j * TR_LD + j];
                    __syncthreads();
                    const double xj = dx[j];
                    for (int i = lane; i < j; i += 64) dx[i] -= L[j * TR_LD + i] * xj;
                    __syncthreads();
                }
            }
            __syncthreads();
            for (int i = lane; i < N; i += 64) xv[i] += dx[i];                             // update (stale dx after a failed solve, as g2o)
            __syncthreads();
            const double temp = ok ? eval_chi() : DBL_MAX;
            double sc = 0;
            for (int i = lane; i < N; i += 64) sc += dx[i] * (lam * dx[i] + bvec[i]);
            sc = wave_sum_all(sc) + 1e-3;
            rho = (chi - temp) / sc;
            ++trials;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1.0 - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2.0 / 3.0);
                lam *= fmax(1.0 / 3.0, alpha);
                ni = 2;
                chi = temp;
            } else {
                lam *= ni;
                ni *= 2;
                __syncthreads();
                for (int i = lane; i < N; i += 64) xv[i] = xb[i];                          // pop
                __syncthreads();
                if (!isfinite(lam)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10);
        ++iters;
        if (qmax == 10 || rho == 0 || !isfinite(lam)) break;
    }
    // ================= outlier checks after the solve
    const double chi_final = eval_chi();                                                   // (also refreshes pw)
    double bad = 0;
    for (int pi = lane; pi < n_pairs; pi += 64) {
        if (pairF[5 * pi] == 0) continue;
        int a = 0, rem = pi;
        while (rem >= V - 1 - a) { rem -= V - 1 - a; ++a; }
        const int b = a + 1 + rem;
        const double d0 = pw[3 * b] - pw[3 * a], d1 = pw[3 * b + 1] - pw[3 * a + 1], d2 = pw[3 * b + 2] - pw[3 * a + 2];
        for (int k = 0; k < n_nb; ++k) {
            const int j = nb[k];
            if (A.has_lm[(size_t)fr[a] * n + j] && A.has_lm[(size_t)fr[b] * n + j] && A.has_lm[(size_t)first * n + j]) {
                const float* xa = A.lm_xyz + 3 * ((size_t)fr[a] * n + j);
                const float* xb_ = A.lm_xyz + 3 * ((size_t)fr[b] * n + j);
                const double e0 = (double)(xb_[0] - xa[0]) - d0, e1 = (double)(xb_[1] - xa[1]) - d1, e2 = (double)(xb_[2] - xa[2]) - d2;
                if (info_s * (e0 * e0 + e1 * e1 + e2 * e2) > (double)7.815f) bad += 1;
            }
        }
    }
    bad = wave_sum_all(bad);
    if (A.o_dbg && lane == 0) { A.o_dbg[4 * ci] = chi_final; A.o_dbg[4 * ci + 1] = iters; A.o_dbg[4 * ci + 2] = trials; A.o_dbg[4 * ci + 3] = n_edges; }
    if (n_edges > 0 && (float)bad / (float)n_edges > 0.5f) { finish(TR_BAD_NEIGHBOURS, 0, 0, 0); return; }
    double nbad = 0;
    if (lane < V) {
        float u, v;
        project_f32(A.cam, (float)xv[3 * lane], (float)xv[3 * lane + 1], (float)xv[3 * lane + 2], u, v);
        const double r0 = uvd[2 * lane] - (double)u, r1 = uvd[2 * lane + 1] - (double)v;
        if (info_r * (r0 * r0 + r1 * r1) > 5.99 * 10) nbad = 1;
    }
    nbad = wave_sum_all(nbad);
    if ((float)nbad / (float)V > 0.5f) { finish(TR_BAD_ERROR, 0, 0, 0); return; }
    {
#pragma clang fp contract(off)
        const float depth = (float)xv[3 * (V - 1) + 2];
        const float* kp = A.kp_xy + 2 * ((size_t)lastf * n + cand);
        float un[3], o[3];
        unproject_f32(A.cam, kp[0], kp[1], un);
        const float z = un[2];
        un[0] = un[0] / z; un[1] = un[1] / z; un[2] = un[2] / z;
        un[0] = un[0] * depth; un[1] = un[1] * depth; un[2] = un[2] * depth;
        se3_point(se3_inv(pose_of(lastf)), un, o);
        finish(TR_OK, o[0], o[1], o[2]);
    }
}

}  // namespace nrs

using namespace nrs;

extern "C" int nrs_triangulate_batch(nrs_ctx* c, const nrs_camera* cam, int32_t n_frames, const float* poses, int32_t n_ids,
                                     const uint8_t* has_kp, const float* kp_xy, const uint8_t* has_lm, const float* lm_xyz,
                                     const int32_t* last_status, int32_t n_cand, const int32_t* cand_ids, int32_t min_track,
                                     int32_t* out_status, float* out_xyz, double* out_debug) {
    if (!c) return NRS_ERR_INVALID;
    if (!cam || n_frames < 1 || n_frames > TR_MAXF || n_ids <= 0 || n_cand < 0 || !poses || !has_kp || !kp_xy || !has_lm || !lm_xyz ||
        !last_status || (n_cand > 0 && (!cand_ids || !out_status || !out_xyz)))
        return c->fail(NRS_ERR_INVALID, "nrs_triangulate_batch: bad argument (at most %d buffered frames)", TR_MAXF);
    if (cam->model != NRS_CAM_PINHOLE && cam->model != NRS_CAM_KB8) return c->fail(NRS_ERR_INVALID, "unknown camera model %d", cam->model);
    for (int i = 0; i < n_cand; ++i) {
        if (cand_ids[i] < 0 || cand_ids[i] >= n_ids) return c->fail(NRS_ERR_INVALID, "candidate id out of range");
        if (!has_kp[(size_t)(n_frames - 1) * n_ids + cand_ids[i]]) return c->fail(NRS_ERR_INVALID, "candidate %d has no keypoint in the last snapshot", cand_ids[i]);
    }
    if (n_cand == 0) return NRS_OK;
    NRS_HIP(c, hipSetDevice(c->device));
    const size_t fn = (size_t)n_frames * n_ids;
    const size_t bytes = sizeof(float) * 7 * n_frames + 2 * fn + sizeof(float) * 5 * fn + sizeof(int) * ((size_t)n_ids + 2 * (size_t)n_cand) +
                         sizeof(float) * 3 * (size_t)n_cand + sizeof(double) * 4 * (size_t)n_cand + 1024;
    DevBuf big;
    NRS_TRY(c->ensure(big, bytes + 256 * 12));
    struct Free2 { nrs_ctx* c; DevBuf* b; ~Free2() { c->release(*b); } } fr2{c, &big};
    char* p = big.as<char>();                                          // carved with 256-byte alignment
    TriArgs A;
    A.cam.model = cam->model;
    for (int i = 0; i < 8; ++i) A.cam.p[i] = cam->params[i];
    A.F = n_frames; A.n = n_ids; A.n_cand = n_cand; A.min_track = min_track;
    auto up = [&](const void* src, size_t nbytes) -> char* {
        char* d = p;
        p += (nbytes + 255) / 256 * 256;
        if (hipMemcpyAsync(d, src, nbytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
        return d;
    };
    A.poses = reinterpret_cast<float*>(up(poses, sizeof(float) * 7 * n_frames));
    A.has_kp = reinterpret_cast<uint8_t*>(up(has_kp, fn));
    A.kp_xy = reinterpret_cast<float*>(up(kp_xy, sizeof(float) * 2 * fn));
    A.has_lm = reinterpret_cast<uint8_t*>(up(has_lm, fn));
    A.lm_xyz = reinterpret_cast<float*>(up(lm_xyz, sizeof(float) * 3 * fn));
    A.status = reinterpret_cast<int*>(up(last_status, sizeof(int) * (size_t)n_ids));
    A.cand = reinterpret_cast<int*>(up(cand_ids, sizeof(int) * (size_t)n_cand));
    if (!A.poses || !A.has_kp || !A.kp_xy || !A.has_lm || !A.lm_xyz || !A.status || !A.cand) return c->fail(NRS_ERR_HIP, "nrs_triangulate_batch: upload failed");
    A.o_status = reinterpret_cast<int*>(p); p += (sizeof(int) * (size_t)n_cand + 255) / 256 * 256;
    A.o_xyz = reinterpret_cast<float*>(p); p += (sizeof(float) * 3 * (size_t)n_cand + 255) / 256 * 256;
    A.o_dbg = out_debug ? reinterpret_cast<double*>(p) : nullptr;
    const size_t shm = sizeof(double) * (2 * TR_N * TR_LD + 7 * TR_N + 7 * TR_MAXF + 5 * TR_PAIRS + 2 * TR_MAXF) + sizeof(int) * (TR_MAXN + 1 + TR_MAXF + 8);
    NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_triangulate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_triangulate, dim3(n_cand), dim3(64), shm, c->stream, A);
    NRS_HIP(c, hipGetLastError());
    NRS_HIP(c, hipMemcpyAsync(out_status, A.o_status, sizeof(int) * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(out_xyz, A.o_xyz, sizeof(float) * 3 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    if (out_debug) NRS_HIP(c, hipMemcpyAsync(out_debug, A.o_dbg, sizeof(double) * 4 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}
