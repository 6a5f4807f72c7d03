// Embedded-deformation mode of the single-frame engine (N2; SURVEY.md 8d: "other points interpolate from <= 11 nodes with
// normalised weights"): observations of points that carry no vertex.  Part of nrs_engine.hip (one translation unit).
//
// The reference has no such edge; it is ReprojectionErrorWithDeformation (reprojection_error_with_deformation.cc:37-68: residual
// z - pi(T (X0 + delta)), fp32 projection and projection Jacobian, information 1 / 0.5^2, Huber sqrt(5.99), OPT:203-204,236) with
// delta = sum_k omega_k delta_{n_k}, the interpolation rule of the reference's own skinning (stage 2 of
// CameraPoseAndDeformationOptimization, OPT:476-553, spatial_regularizer_fixed.cc:32-43) turned into a differentiable one: the
// Jacobian with respect to node k is omega_k times the reference's 2 x 3 block.  oracle/embedded_oracle.py states the mode; with
// every point a node nothing of this file runs and the solve is the parity-mode one, bit for bit.
//
// One thread per observation: residual, Huber, and -- at a linearisation -- its three dense pieces (J_l^T w J_l, -J_l^T w r,
// J_p^T w J_l), which k_nd_values folds into the direct solver's blocks through fixed-order lists (node diagonal blocks, node-node
// pairs of every observation's node set, pose-node blocks, gradients): no atomics on values, bit-reproducible.
#pragma once

namespace nrs {

constexpr int SK_MAX = 11;           // nodes per skinned observation (the walk of OPT:255-279 accepts 11)

template <bool LIN>
__global__ __launch_bounds__(BLK) void k_skin(Dev P, const Pose* __restrict__ poses, const double* __restrict__ xl) {
    __shared__ double lds[4 * 28];
    const int tid = threadIdx.x, i = blockIdx.x * BLK + tid;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    if (LIN && i == 0) *P.sk_maxdiag = 0.0;
    if (i < P.sk_n) {
        const Pose Tcw = poses[0];
        double R[9];
        quat_to_R(Tcw.q, R);
        double x0 = P.sk_X0[3 * (size_t)i], x1 = P.sk_X0[3 * (size_t)i + 1], x2 = P.sk_X0[3 * (size_t)i + 2];
#pragma unroll
        for (int k = 0; k < SK_MAX; ++k) {
            const int row = P.sk_row[SK_MAX * (size_t)i + k];
            const double om = P.sk_om[SK_MAX * (size_t)i + k];
            if (row >= 0) { x0 += om * xl[3 * (size_t)row]; x1 += om * xl[3 * (size_t)row + 1]; x2 += om * xl[3 * (size_t)row + 2]; }
        }
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
        float u, v;
        project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
        const double r0 = (double)P.sk_uv[2 * (size_t)i] - (double)u, r1 = (double)P.sk_uv[2 * (size_t)i + 1] - (double)v;
        const double e = P.info_reproj * (r0 * r0 + r1 * r1);
        P.sk_chi[i] = e;
        const bool active = P.sk_active[i] != 0;
        double rho0, rho1;
        huber(e, P.delta_reproj, rho0, rho1);
        if (active) acc[27] = rho0;
        if (LIN) {
            double* rec = P.sk_rec + 27 * (size_t)i;
            if (active) {
                float Jf[6];
                projection_jacobian_f32(P.cam, (float)px, (float)py, (float)pz, Jf);
                const double w = rho1 * P.info_reproj, pm = P.pose_fixed[0] ? 0.0 : 1.0;
                double Jp[2][6], Jl[2][3];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                    Jp[rr][0] = pm * (-j1 * pz + j2 * py);
                    Jp[rr][1] = pm * (j0 * pz - j2 * px);
                    Jp[rr][2] = pm * (-j0 * py + j1 * px);
                    Jp[rr][3] = pm * j0; Jp[rr][4] = pm * j1; Jp[rr][5] = pm * j2;
                    Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
                    Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
                    Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
                }
                rec[0] = w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]);
                rec[1] = w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]);
                rec[2] = w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
                rec[3] = w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]);
                rec[4] = w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]);
                rec[5] = w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
#pragma unroll
                for (int c = 0; c < 3; ++c) rec[6 + c] = -w * (Jl[0][c] * r0 + Jl[1][c] * r1);
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int c = 0; c < 3; ++c) rec[9 + 3 * p + c] = w * (Jp[0][p] * Jl[0][c] + Jp[1][p] * Jl[1][c]);
                int k = 0;
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int q = p; q < 6; ++q) { acc[k] = w * (Jp[0][p] * Jp[0][q] + Jp[1][p] * Jp[1][q]); ++k; }
#pragma unroll
                for (int p = 0; p < 6; ++p) acc[21 + p] = -w * (Jp[0][p] * r0 + Jp[1][p] * r1);
            } else {
#pragma unroll
                for (int k = 0; k < 27; ++k) rec[k] = 0.0;
            }
        }
    }
    block_sum_store<28>(acc, lds, tid, P.sk_part + (size_t)blockIdx.x * 32);
}

}  // namespace nrs
