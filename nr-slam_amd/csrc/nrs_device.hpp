// Device-side math shared by all NR-SLAM kernels (gfx950).
//
// Mixed precision is part of the reference's arithmetic, not a tuning choice: solver state and
// residuals are fp64 (g2o number_t), the projection and its Jacobian are evaluated in fp32
// (reference modules/calibration/camera_model.h:89-95,131-137).  The fp32 part is compiled with FP
// contraction off so that the pinhole path is the same sequence of IEEE mul/div/add the CPU
// executes; the fp64 part is free to use FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrs {

struct Cam {
    int model;
    float p[8];
};

struct Pose {          // T_camera_world
    double q[4];       // x y z w
    double t[3];
};

// ---- quaternion / SE(3), restating g2o SE3Quat on Eigen quaternions
// (reference third_party/g2o/g2o/types/slam3d/se3quat.h:96-102,201-229,250-255)
__host__ __device__ inline void quat_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_normalize(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

__host__ __device__ inline void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// Eigen QuaternionBase::_transformVector: v + w*uv + qv x uv, uv = 2 (qv x v)
__host__ __device__ inline void quat_rotate(const double* q, const double* v, double* o) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

__host__ __device__ inline void R_to_quat(const double* R, double* q) {
    // Eigen's matrix -> quaternion conversion; the three "largest diagonal" cases are spelled out
    // so that no array is indexed dynamically (dynamic indices put q/R in scratch memory on the GPU)
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {                 // i = 0, j = 1, k = 2
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t;
        q[1] = (R[3] + R[1]) * t;
        q[2] = (R[6] + R[2]) * t;
    } else if ((R[4] > R[0]) && !(R[8] > R[4])) {                  // i = 1, j = 2, k = 0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t;
        q[2] = (R[7] + R[5]) * t;
        q[0] = (R[1] + R[3]) * t;
    } else {                                                        // i = 2, j = 0, k = 1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t;
        q[0] = (R[2] + R[6]) * t;
        q[1] = (R[5] + R[7]) * t;
    }
}

// pose <- exp([omega, upsilon]) * pose   (VertexSE3Expmap::oplusImpl, vertex_se3_expmap.cpp:48-51)
__host__ __device__ inline void pose_oplus(Pose& P, const double* upd) {
    const double wx = upd[0], wy = upd[1], wz = upd[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double Om[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double Om2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Om[i * 3 + k] * Om[k * 3 + j];
            Om2[i * 3 + j] = s;
        }
    double a, b, c, d;                 // R = I + a Om + b Om2 ; V = I + c Om + d Om2
    if (theta < 0.00001) {
        a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0;
    } else {
        a = sin(theta) / theta;
        b = (1 - cos(theta)) / (theta * theta);
        c = b;
        d = (theta - sin(theta)) / (theta * theta * theta);
    }
    double R[9], V[9];
    for (int i = 0; i < 9; ++i) {
        const double id = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = id + a * Om[i] + b * Om2[i];
        V[i] = id + c * Om[i] + d * Om2[i];
    }
    double qe[4], te[3];
    R_to_quat(R, qe);
    quat_normalize(qe);
    for (int i = 0; i < 3; ++i) te[i] = V[i * 3] * upd[3] + V[i * 3 + 1] * upd[4] + V[i * 3 + 2] * upd[5];
    // SE3Quat::operator*: t = te + qe * t ; q = qe * q ; normalize
    double rt[3], qn[4];
    quat_rotate(qe, P.t, rt);
    quat_mul(qe, P.q, qn);
    quat_normalize(qn);
    for (int i = 0; i < 3; ++i) P.t[i] = te[i] + rt[i];
    for (int i = 0; i < 4; ++i) P.q[i] = qn[i];
}

// ---- fp32 camera models (reference modules/calibration/pin_hole.cc:27-49,
//      kannala_brandt_8.cc:34-51,87-116).  Contraction off: same IEEE op sequence as the CPU.
__device__ inline void project_f32(const Cam& c, float x, float y, float z, float& u, float& v) {
#pragma clang fp contract(off)
    if (c.model == 0) {
        u = c.p[0] * x / z + c.p[2];
        v = c.p[1] * y / z + c.p[3];
    } else {
        const float r2 = x * x + y * y;
        // atan2f / cosf / sinf are DEFINED here (and in the oracle) as the double-precision function
        // rounded to float: the value a correctly rounded float routine returns.  libm, ocml and
        // NumPy float routines differ in the last ulp, which flips inlier decisions once in a while.
        const float th = (float)atan2((double)sqrtf(r2), (double)z);
        const float psi = (float)atan2((double)y, (double)x);
        const float th2 = th * th, th3 = th * th2, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
        const float r = th + c.p[4] * th3 + c.p[5] * th5 + c.p[6] * th7 + c.p[7] * th9;
        u = c.p[0] * r * (float)cos((double)psi) + c.p[2];
        v = c.p[1] * r * (float)sin((double)psi) + c.p[3];
    }
}

// J row-major 2x3
__device__ inline void projection_jacobian_f32(const Cam& c, float x, float y, float z, float* J) {
#pragma clang fp contract(off)
    if (c.model == 0) {
        J[0] = c.p[0] / z; J[1] = 0.f; J[2] = -c.p[0] * x / (z * z);
        J[3] = 0.f; J[4] = c.p[1] / z; J[5] = -c.p[1] * y / (z * z);
    } else {
        const float fx = c.p[0], fy = c.p[1], k0 = c.p[4], k1 = c.p[5], k2 = c.p[6], k3 = c.p[7];
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        const float r2 = x2 + y2;
        const float r = sqrtf(r2);
        const float r3 = r2 * r;
        const float th = (float)atan2((double)r, (double)z);
        const float th2 = th * th, th3 = th2 * th, th4 = th2 * th2, th5 = th4 * th;
        const float th6 = th2 * th4, th7 = th6 * th, th8 = th4 * th4, th9 = th8 * th;
        const float f = th + th3 * k0 + th5 * k1 + th7 * k2 + th9 * k3;
        const float fd = 1 + 3 * k0 * th2 + 5 * k1 * th4 + 7 * k2 * th6 + 9 * k3 * th8;
        J[0] = fx * (fd * z * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
        J[1] = fx * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        J[2] = -fx * fd * x / (r2 + z2);
        J[3] = fy * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        J[4] = fy * (fd * z * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
        J[5] = -fy * fd * y / (r2 + z2);
    }
}

// ---- Huber (reference third_party/g2o/g2o/core/robust_kernel_impl.cpp:60-74); delta<=0: no kernel
__device__ inline void huber(double e, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (delta <= 0 || e <= dsqr) { rho0 = e; rho1 = 1.0; }
    else {
        const double sq = sqrt(e);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
}

// ---- wave64 reductions (fp64) on DPP cross-lane moves.  hipcc lowers __shfl_xor to ds_bpermute
// (an LDS-pipe round trip per 32-bit half and step: 12 per double); the DPP forms below are plain
// VALU moves.  Fixed combination order => bit-reproducible.
template <int CTRL, int ROW_MASK = 0xf>
__device__ inline double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// sum over the 64 lanes; every lane receives the total
__device__ inline double wave_sum(double v) {
    v += dpp_mov<DPP_QUAD_1032>(v);
    v += dpp_mov<DPP_QUAD_2301>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);                   // every lane of a 16-lane row holds the row sum
    v += dpp_mov<DPP_ROW_BCAST15, 0xa>(v);             // rows 1 and 3 += rows 0 and 2
    v += dpp_mov<DPP_ROW_BCAST31, 0xc>(v);             // rows 2 and 3 += lane 31 (rows 0+1): lane 63 = total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// sum over aligned groups of T lanes (T = 1, 2, 4, 8, 16, 32, 64); every lane of a group receives its sum
template <int T>
__device__ inline double group_sum(double v) {
    if (T >= 2) v += dpp_mov<DPP_QUAD_1032>(v);
    if (T >= 4) v += dpp_mov<DPP_QUAD_2301>(v);
    if (T >= 8) v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    if (T >= 16) v += dpp_mov<DPP_ROW_MIRROR>(v);
    if (T >= 32) v += __shfl_xor(v, 16, 64);                       // (across DPP rows)
    if (T >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

}  // namespace nrs
