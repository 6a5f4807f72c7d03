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


template <bool LIN>
__global__ __launch_bounds__(BLK) void k_skin(Dev P, const Pose* __restrict__ poses, const double* __restrict__ xl) {
    __shared__ double lds[4 * 28];
    const int tid = threadIdx.x, i = blockIdx.x * BLK + tid;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    if (LIN && i == 0) *P.sk_maxdiag = 0.0;
    const int kp = P.sk_blk_pose ? P.sk_blk_pose[blockIdx.x] : 0;  // (a workgroup's observations share a pose)
    if (i < P.sk_n) {
        const Pose Tcw = poses[kp];
        double R[9];
        quat_to_R(Tcw.q, R);
        double x0 = P.sk_X0[3 * (size_t)i], x1 = P.sk_X0[3 * (size_t)i + 1], x2 = P.sk_X0[3 * (size_t)i + 2];
#pragma unroll
        for (int k = 0; k < SK_MAX; ++k) {
            const int row = P.sk_row[(size_t)k * P.sk_n + i];       // (node lists slot-major inside a k: coalesced)
            const double om = P.sk_om[(size_t)k * P.sk_n + i];
            if (row >= 0 && P.sk_base) {                           // BA form: the node's displacement from where the window started
                x0 += om * (xl[3 * (size_t)row] - P.sk_base[3 * (size_t)row]); x1 += om * (xl[3 * (size_t)row + 1] - P.sk_base[3 * (size_t)row + 1]);
                x2 += om * (xl[3 * (size_t)row + 2] - P.sk_base[3 * (size_t)row + 2]);
            } else if (row >= 0) { x0 += om * xl[3 * (size_t)row]; x1 += om * xl[3 * (size_t)row + 1]; x2 += om * xl[3 * (size_t)row + 2]; }
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
                const double w = rho1 * P.info_reproj, pm = P.pose_fixed[kp] ? 0.0 : 1.0;
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
            if (P.sk_pcg) {                                        // PCG form: the operator's 24 values once more, value-major (k_skin_op reads them coalesced)
#pragma unroll
                for (int k = 0; k < 6; ++k) P.sk_recT[(size_t)k * P.sk_n + i] = rec[k];
#pragma unroll
                for (int k = 0; k < 18; ++k) P.sk_recT[(size_t)(6 + k) * P.sk_n + i] = rec[9 + k];
            }
        }
    }
    block_sum_store<28>(acc, lds, tid, P.sk_part + (size_t)blockIdx.x * 32);
}

// ---- embedded BA windows (N2b): K poses, the skinned observations act through the PCG path -------------------------------------------
// Per linearisation, behind the lineariser and k_pose_sums (which WRITE D, b_l, H_pp, b_p): what the observations add to them.
//   k_skin_rows : a node row's diagonal block += sum om^2 J_l^T w J_l, its gradient += sum om (-J_l^T w r) over its list
//   k_skin_pose : H_pp / b_p of a pose += the block sums k_skin left (its blocks in order)
// Per PCG iteration, behind k_spmv_f:  H u of the observations' blocks, with s_o = sum_k om_k u_{n_k}:
//   k_skin_op      (per observation): g_o = A_o s_o + B_o^T u_p; block sums of B_o s_o (pose rows), of (B_o^T u_p) . s_o (the
//                                     observations' share of the cross term u_l . H_lp u_p) and of g_o . s_o (their share of w.u)
//   per node row                    : w_row += sum om g_o -- by k_pcg_update<true> for the rows it updates (k_skin_op_rows where the
//                                     update is the generic kernel)
// k_pcg_update / k_reduce_partials add the block sums in (fixed order): every scalar of the iteration is known after the FIRST launch
// (k_spmv_f_skin), so the row pass needs no reduction of its own and the iteration is two launches.  SK_RL lanes per list, combined by the fixed butterfly of
// sub_sum_t.  Both kernels are bound by the number of cache lines a wave's loads touch (one wave per SIMD at this size): the
// per-observation operands are value-major (sk_recT, sk_row, sk_om: a load instruction of 64 observations touches 4 lines, not 64),
// g_o is 3 values in a 32-byte slot (two 16-byte loads per list entry).

__device__ inline void sk_atomic_max(double* addr, double v) {     // v >= 0: the bit patterns of non-negative doubles order like integers
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(BLK) void k_skin_rows(Dev P) {
    const int tid = threadIdx.x, j = blockIdx.x * SK_RPB + tid / SK_RL, t = tid % SK_RL;
    double D[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    const bool live = j < P.sk_nrl && !(P.rflag[P.sk_rl_row[min(j, P.sk_nrl - 1)]] & RF_FIXED);   // (a fixed row is an identity row: nothing is added to it)
    if (live)
        for (int q = P.sk_rl_ptr[j] + t; q < P.sk_rl_ptr[j + 1]; q += SK_RL) {
            const double om = P.sk_rl_om[q];
            const double* rec = P.sk_rec + 27 * (size_t)P.sk_rl_obs[q];
            const double o2 = om * om;
#pragma unroll
            for (int k = 0; k < 6; ++k) D[k] += o2 * rec[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) b[k] += om * rec[6 + k];
        }
#pragma unroll
    for (int k = 0; k < 6; ++k) D[k] = sub_sum_t<SK_RL>(D[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = sub_sum_t<SK_RL>(b[k]);
    if (live && t == 0) {
        const size_t row = (size_t)P.sk_rl_row[j];
        double* Dr = P.D + 6 * row;
        double dd[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { dd[k] = Dr[k] + D[k]; Dr[k] = dd[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) P.bl[3 * row + k] += b[k];
        sk_atomic_max(P.sk_maxdiag, fmax(fabs(dd[0]), fmax(fabs(dd[3]), fabs(dd[5]))));
    }
}

__global__ __launch_bounds__(BLK) void k_skin_pose(Dev P) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= 27 * P.K) return;
    const int k = i / 27, cc = i % 27;
    double s = 0;
    for (int b = P.sk_pose_blk[k]; b < P.sk_pose_blk[k + 1]; ++b) s += P.sk_part[(size_t)b * 32 + cc];
    if (cc < 21) {
        const double v = P.Hpp[21 * k + cc] + s;
        P.Hpp[21 * k + cc] = v;
        if (cc == 0 || cc == 6 || cc == 11 || cc == 15 || cc == 18 || cc == 20) sk_atomic_max(P.sk_maxdiag, fabs(v));
    } else P.bp[6 * k + (cc - 21)] += s;
}

__device__ __forceinline__ void skin_op_body(const Dev& P, const int it, const int blk) {
    __shared__ double lds[4 * 8];
    const int tid = threadIdx.x, i = blk * BLK + tid;
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int done = P.flags[0];                                   // (read beside the first operands, not before them)
    if (i < P.sk_n) {
        const int kp = P.sk_blk_pose[blk];
        const double* up = ((it & 1) ? P.up2 : P.up) + 6 * kp;
        const size_t n = (size_t)P.sk_n;
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < SK_MAX; ++k) {
            const int row = P.sk_row[k * n + i];
            const double om = P.sk_om[k * n + i];
            if (row >= 0) { s0 += om * P.uv3[3 * (size_t)row]; s1 += om * P.uv3[3 * (size_t)row + 1]; s2 += om * P.uv3[3 * (size_t)row + 2]; }
        }
        const double* rec = P.sk_recT + i;                         // A_o (6), B_o (18), value-major
        const double A0 = rec[0], A1 = rec[n], A2 = rec[2 * n], A3 = rec[3 * n], A4 = rec[4 * n], A5 = rec[5 * n];
        const double a0 = A0 * s0 + A1 * s1 + A2 * s2, a1 = A1 * s0 + A3 * s1 + A4 * s2, a2 = A2 * s0 + A4 * s1 + A5 * s2;
        double g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const double b0 = rec[(6 + 3 * p) * n], b1 = rec[(7 + 3 * p) * n], b2 = rec[(8 + 3 * p) * n], u = up[p];
            g0 += b0 * u; g1 += b1 * u; g2 += b2 * u;
            q[p] = b0 * s0 + b1 * s1 + b2 * s2;
        }
        q[6] = g0 * s0 + g1 * s1 + g2 * s2;                        // (B_o^T u_p) . s_o
        q[7] = (a0 + g0) * s0 + (a1 + g1) * s1 + (a2 + g2) * s2;   // g_o . s_o = sum over the observation's rows of u_row . (om g_o): its share of w.u
        if (!done) {
            double* g = P.sk_g + 4 * (size_t)i;
            g[0] = a0 + g0; g[1] = a1 + g1; g[2] = a2 + g2;
        } else {                                                   // (the solve has finished: nothing is added any more)
#pragma unroll
            for (int p = 0; p < 8; ++p) q[p] = 0;
        }
    }
    block_sum<8>(q, lds, tid & 63, tid >> 6);
    if (tid == 0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) P.sk_opart[(size_t)blk * 8 + p] = q[p];
    }
}
__global__ __launch_bounds__(BLK) void k_skin_op(Dev P, int it) { skin_op_body(P, it, (int)blockIdx.x); }
// The operator of the regularisers and the observations' pass in ONE launch: the two read u and write different things (w's rows /
// sk_g and the block sums), so the first n_op workgroups are k_spmv_f's and the rest k_skin_op's -- one launch and one kernel's
// chain of dependent loads less per PCG iteration (both are latency-bound at a BA window's size).  Same bodies, same bits.
template <int T>
__global__ __launch_bounds__(BLK, 4) void k_spmv_f_skin(Dev P, double lam, int cls, int it, double tol2, int n_op) {
    if ((int)blockIdx.x < n_op) spmv_f_body<T, false>(P, lam, cls, it, tol2, (int)blockIdx.x);
    else skin_op_body(P, it, (int)blockIdx.x - n_op);
}

// The row pass in a launch of its own (large windows on the hierarchical reduction, NRS_SKIN_ROWS_OWN_LAUNCH=1): w_row += sum om g_o.
// Otherwise k_pcg_update<true> does it for the rows it is about to update.  The rows' shares of the dot products are k_skin_op's.
__global__ __launch_bounds__(BLK) void k_skin_op_rows(Dev P) {
    const int tid = threadIdx.x, r = blockIdx.x * SK_RPB + tid / SK_RL, t = tid % SK_RL;     // (n_rows is a multiple of BLK)
    const int done = P.flags[0], q0 = P.sk_row_q[2 * (size_t)r], q1 = P.sk_row_q[2 * (size_t)r + 1];
    const uint8_t rf = P.rflag[r];
    const double w0 = P.wv[3 * (size_t)r], w1 = P.wv[3 * (size_t)r + 1], w2 = P.wv[3 * (size_t)r + 2];
    const bool live = !done && q1 > q0 && !(rf & RF_FIXED);       // (a fixed row is an identity row: nothing is added to it)
    double a[3] = {0, 0, 0};
    if (live) skin_row_gather(P, q0, q1, t, a);
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = sub_sum_t<SK_RL>(a[k]);
    if (live && t == 0) { P.wv[3 * (size_t)r] = w0 + a[0]; P.wv[3 * (size_t)r + 1] = w1 + a[1]; P.wv[3 * (size_t)r + 2] = w2 + a[2]; }
}

}  // namespace nrs
