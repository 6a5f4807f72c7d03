// Linearisation kernels: residuals, Jacobians, Huber weights, per-incidence factors, fixed-order sums.
// Part of nrs_engine.hip (one translation unit); see that file's header for the design.
#pragma once

namespace nrs {

// =====================================================================================
// linearisation, part 1: reprojection edges.  One thread per row, one pose per workgroup.
// LIN = true is the stored-block (gather) path only: on the LDS path the reprojection edges are linearised inside
// k_reg, in the same pass as the regularisers (D, b_l written once).  LIN = false: chi2 of a trial state, both paths.
//   ReprojectionError / ReprojectionErrorWithDeformation computeError + linearizeOplus
//   (reference reprojection_error.cc:32-64, reprojection_error_with_deformation.cc:37-68),
//   quadratic form with Huber weight (base_fixed_sized_edge.hpp:49-63, base_edge.h:158-164).
// =====================================================================================
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_reproj(Dev P, const Pose* __restrict__ poses,
                                                const double* __restrict__ xl) {
    __shared__ double lds[4 * 28];
    const int gi = xcd_tile(blockIdx.x, P.sh_ng);
    if (gi >= P.sh_ng) return;
    const int g = P.sh_g0 + gi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = g * ROW_ALIGN + tid;
    const int kf = P.grp_pose[g];
    const Pose Tcw = poses[kf];
    const bool pfix = P.pose_fixed[kf] != 0;
    double R[9];
    quat_to_R(Tcw.q, R);
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    const int rf = P.rflag[row];
    const bool rfix = (rf & RF_FIXED) != 0;
    // an edge whose vertices are all fixed is not part of the optimisation (sparse_optimizer.cpp:236)
    const bool active = (rf & RF_OBS) && (rf & RF_REPROJ_ACTIVE) && !(pfix && rfix);
    bool wrote = false;
    if (active) {
        double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
        if (P.X0) { x0 += P.X0[3 * row]; x1 += P.X0[3 * row + 1]; x2 += P.X0[3 * row + 2]; }
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
        float u, v;
        project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
        const double r0 = (double)P.uv[2 * row] - (double)u, r1 = (double)P.uv[2 * row + 1] - (double)v;
        double rho0, rho1;
        huber(P.info_reproj * (r0 * r0 + r1 * r1), P.delta_reproj, rho0, rho1);
        acc[27] = rho0;
        if (LIN) {
            float Jf[6];
            projection_jacobian_f32(P.cam, (float)px, (float)py, (float)pz, Jf);
            const double w = rho1 * P.info_reproj;
            const double pm = pfix ? 0.0 : 1.0, lm = rfix ? 0.0 : 1.0;
            double Jp[2][6], Jl[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                Jp[rr][0] = pm * (-j1 * pz + j2 * py);
                Jp[rr][1] = pm * (j0 * pz - j2 * px);
                Jp[rr][2] = pm * (-j0 * py + j1 * px);
                Jp[rr][3] = pm * j0; Jp[rr][4] = pm * j1; Jp[rr][5] = pm * j2;
                Jl[rr][0] = lm * (j0 * R[0] + j1 * R[3] + j2 * R[6]);
                Jl[rr][1] = lm * (j0 * R[1] + j1 * R[4] + j2 * R[7]);
                Jl[rr][2] = lm * (j0 * R[2] + j1 * R[5] + j2 * R[8]);
            }
            int k = 0;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q) { acc[k] = w * (Jp[0][p] * Jp[0][q] + Jp[1][p] * Jp[1][q]); ++k; }
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[21 + p] = -w * (Jp[0][p] * r0 + Jp[1][p] * r1);
            if (P.use_lds) {
                // factored form: the PCG kernels rebuild J_l, J_p from these 32 bytes
                RowRec rc;
#pragma unroll
                for (int q = 0; q < 6; ++q) rc.J[q] = Jf[q];
                rc.w = lm * w;
                P.rowrec[row] = rc;
            } else {
                // H_pl (6x3), component-major so that a wave writes 18 contiguous runs
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        P.Hpl[(size_t)(p * 3 + c) * P.n_rows + row] = w * (Jp[0][p] * Jl[0][c] + Jp[1][p] * Jl[1][c]);
            }
            double* D = P.D + 6 * (size_t)row;
            D[0] = w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]);
            D[1] = w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]);
            D[2] = w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
            D[3] = w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]);
            D[4] = w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]);
            D[5] = w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
            P.bl[3 * row] = -w * (Jl[0][0] * r0 + Jl[1][0] * r1);
            P.bl[3 * row + 1] = -w * (Jl[0][1] * r0 + Jl[1][1] * r1);
            P.bl[3 * row + 2] = -w * (Jl[0][2] * r0 + Jl[1][2] * r1);
            wrote = true;
        }
    }
    if (LIN && !wrote) {
        if (P.use_lds) {
            RowRec rc;
#pragma unroll
            for (int q = 0; q < 6; ++q) rc.J[q] = 0.f;
            rc.w = 0;
            P.rowrec[row] = rc;
        } else {
#pragma unroll
            for (int c = 0; c < 18; ++c) P.Hpl[(size_t)c * P.n_rows + row] = 0;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) P.D[6 * (size_t)row + c] = 0;
        P.bl[3 * row] = P.bl[3 * row + 1] = P.bl[3 * row + 2] = 0;
    }
    if (LIN) {
        block_sum_store<28>(acc, lds, tid, P.part_lin + (size_t)g * 32);
    } else {
        double c1[1] = {acc[27]};
        block_sum<1>(c1, lds, lane, wave);
        if (tid == 0) P.part_rchi[g] = c1[0];
    }
}

// =====================================================================================
// linearisation, part 2: springs and dampers from the incidence lists (T lanes per row).
//   PositionRegularizer (position_regularizer.cc:32-61, Jacobian as written),
//   PositionRegularizerWithDeformation (position_regularizer_with_deformation.cc:31-57),
//   SpatialRegularizer (spatial_regularizer.cc:32-59), SpatialRegularizerWithDeformation
//   (spatial_regularizer_with_deformation.cc:36-49), SpatialRegularizerFixed
//   (spatial_regularizer_fixed.cc:32-43).
// =====================================================================================
template <int T, bool LIN, bool LDS, bool DF = false>
__global__ __launch_bounds__(BLK) void k_reg(Dev P, const double* __restrict__ xl_g, int cls) {
    static_assert(!DF || (LIN && LDS), "temporal-difference dampers: linearisation on the LDS path only");
    __shared__ double lds[4 * 2];
    __shared__ double lds28[(LIN && LDS) ? 4 * 28 : 1];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    int b = xcd_tile(blockIdx.x, LDS ? P.sh_nt[cls] : P.n_regblk);
    if (b >= (LDS ? P.sh_nt[cls] : P.n_regblk)) return;
    if (LDS) b = P.tile_list[(cls ? P.n_tiles_cls[0] : 0) + P.sh_t0[cls] + b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const int rf = P.rflag[row];
    const bool rfix = (rf & RF_FIXED) != 0;
    // incidence headers are double-buffered in two-record chunks; the first chunk of both streams is
    // requested before the tile is staged
    constexpr int U = 2;
    const int s_beg = P.ss_ptr[slice], s_end = P.ss_ptr[slice + 1];
    const int d_beg = P.sd_ptr[slice], d_end = P.sd_ptr[slice + 1];
    uint2 shA[U], shB[U], dhA[U], dhB[U];
    float dwA[U], dwB[U];
    auto load_sh = [&](uint2* h, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            h[q] = make_uint2(0xFFFFu, 0u);
            if (j < s_end) h[q] = make_uint2(P.s_om[j], __float_as_uint(P.s_d0[j]));
        }
    };
    auto load_dh = [&](uint2* h, float* w, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            h[q] = make_uint2(0u, 0xFFFF0000u);
            w[q] = 0.f;
            if (j < d_end) {
                if (DF) { const uint32_t om = P.d_om[j]; h[q] = make_uint2(om & 0xFFFFu, om & 0xFFFF0000u); }   // partner in .x, meta in the high half of .y
                else h[q] = P.d_hdr[j];
                w[q] = P.d_w[j];
            }
        }
    };
    if (LDS) { load_sh(shA, s_beg + lane); load_dh(dhA, dwA, d_beg + lane); }
    // xl: estimates (dampers act on them), xp: X0 + estimates (springs); with LDS staging both are
    // tile-local arrays indexed by the local neighbour ids
    const double* xl = xl_g;
    const double* xp = nullptr;
    int self = row;
    const double* lgf = nullptr;
    const double* lgb = nullptr;
    if (LDS && DF) {
        const size_t nst = (size_t)(P.tile_rows + P.cap_h[cls] + 1);
        double* lx = dyn;
        stage_rows_d(P, b, tid, xl_g, lx, dyn + 3 * nst, dyn + 6 * nst);
        xl = lx; lgf = dyn + 3 * nst; lgb = dyn + 6 * nst;
        __syncthreads();
        self = row - b * P.tile_rows;
    } else if (LDS) {
        double* lx = dyn;
        stage_rows(P, b, tid, xl_g, nullptr, lx);
        xl = lx;
        if (P.X0) {
            double* lp = dyn + 3 * (size_t)(P.tile_rows + P.cap_h[cls]);
            stage_rows(P, b, tid, xl_g, P.X0, lp);
            xp = lp;
        }
        __syncthreads();
        self = row - b * P.tile_rows;
    }
    const double xo0 = xl[3 * self], xo1 = xl[3 * self + 1], xo2 = xl[3 * self + 2];
    double xs0 = xo0, xs1 = xo1, xs2 = xo2;                 // spring position = X0 + x
    if (P.X0) {
        if (LDS) { xs0 = xp[3 * self]; xs1 = xp[3 * self + 1]; xs2 = xp[3 * self + 2]; }
        else { xs0 += P.X0[3 * row]; xs1 += P.X0[3 * row + 1]; xs2 += P.X0[3 * row + 2]; }
    }
    double D[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0}, chi = 0;
    // ---- springs
    const size_t nz = (size_t)P.ss_nnz;
    auto spring = [&](int idx, int o, int meta, double d0) {
        if (!(meta & SM_ACTIVE)) {
            if (LIN) {
                if (LDS) P.s_qc[idx] = 0;
                else { P.s_g[idx] = 0; P.s_g[nz + idx] = 0; P.s_g[2 * nz + idx] = 0; }
            }
            return;
        }
        if (!LIN && !(meta & SM_COUNT)) return;           // chi2 only: every edge is counted from one of its rows
        double y0, y1, y2;
        if (LDS && P.X0) { y0 = xp[3 * o]; y1 = xp[3 * o + 1]; y2 = xp[3 * o + 2]; }
        else {
            y0 = xl[3 * o]; y1 = xl[3 * o + 1]; y2 = xl[3 * o + 2];
            if (P.X0) { y0 += P.X0[3 * o]; y1 += P.X0[3 * o + 1]; y2 += P.X0[3 * o + 2]; }
        }
        const double v0 = xs0 - y0, v1 = xs1 - y1, v2 = xs2 - y2;
        const double d2 = v0 * v0 + v1 * v1 + v2 * v2;
        // The linearisation pass is bound by fp64 VALU work as much as by traffic, and this is the bulk of it (two
        // sqrt and three divisions as the reference writes it, position_regularizer.cc:40-60).  Same quantities from one
        // rsqrt, one reciprocal (and one sqrt for the BA form's 1/sqrt(d)): results within 2 ulp of the literal form.
        double d, id0 = 0, rs = 0;
        if (LIN) { rs = rsqrt(d2); d = d2 * rs; id0 = 1.0 / d0; }
        else d = sqrt(d2);
        const double r = LIN ? P.k_spring * (d - d0) * id0 : P.k_spring * (d - d0) / d0;
        double rho0, rho1;
        huber(P.info_pos * r * r, P.delta_pos, rho0, rho1);
        if (meta & SM_COUNT) chi += rho0;
        if (LIN) {
            const double cg = P.spring_form == 0 ? 2.0 * P.k_spring * id0 * sqrt(rs)      // (k/d0) (1/sqrt(d)) 2
                                                 : P.k_spring * id0 * rs;               // (k/(2 d0 d)) 2
            const double q = rfix ? 0.0 : rho1 * P.info_pos;
            const double g0 = cg * v0, g1 = cg * v1, g2 = cg * v2;
            if (LDS) P.s_qc[idx] = q * cg * cg;
            else { const double sq = sqrt(q); P.s_g[idx] = sq * g0; P.s_g[nz + idx] = sq * g1; P.s_g[2 * nz + idx] = sq * g2; }
            D[0] += q * g0 * g0; D[1] += q * g0 * g1; D[2] += q * g0 * g2;
            D[3] += q * g1 * g1; D[4] += q * g1 * g2; D[5] += q * g2 * g2;
            const double qr = q * r;
            bb[0] -= qr * g0; bb[1] -= qr * g1; bb[2] -= qr * g2;
        }
    };
    {
        const int beg = P.ss_ptr[slice], end = P.ss_ptr[slice + 1];
        if (LDS) {
            auto do_sh = [&](const uint2* hdr, int idx) {
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int o = (int)(hdr[q].x & 0xFFFFu), m16 = (int)(hdr[q].x >> 16);
                    if (o == REC_NONE) {                           // padding slot: the operator reads its factor
                        if (LIN && idx + 64 * q < end) P.s_qc[idx + 64 * q] = 0;
                        continue;
                    }
                    const int meta = ((m16 & SR_ACTIVE) ? SM_ACTIVE : 0) | ((m16 & SR_COUNT) ? SM_COUNT : 0);
                    spring(idx + 64 * q, o, meta, (double)__uint_as_float(hdr[q].y));
                }
            };
            for (int base = beg; base < end; base += 128 * U) {    // wave-uniform trip count
                load_sh(shB, base + 64 * U + lane);
                do_sh(shA, base + lane);
                load_sh(shA, base + 128 * U + lane);
                do_sh(shB, base + 64 * U + lane);
            }
        } else {
            for (int idx = beg + lane; idx < end; idx += 64) {
                const int o = P.s_other[idx];
                if (o < 0) continue;
                spring(idx, o, P.s_meta[idx], (double)P.s_d0[idx]);
            }
        }
    }
    // ---- dampers: r = w((x1n - x1c) - (x2n - x2c)), roles (1c,2c,1n,2n), signs (-,+,+,-)
    double gfo0 = 0, gfo1 = 0, gfo2 = 0, gbo0 = 0, gbo1 = 0, gbo2 = 0;
    if (DF) {
        gfo0 = lgf[3 * self]; gfo1 = lgf[3 * self + 1]; gfo2 = lgf[3 * self + 2];
        gbo0 = lgb[3 * self]; gbo1 = lgb[3 * self + 1]; gbo2 = lgb[3 * self + 2];
    }
    auto damper = [&](int idx, int meta, const int* o, float wf) {
        if (!(meta & DM_ACTIVE)) {
            if (LIN) P.d_s[idx] = 0;
            return;
        }
        if (DF) {
            // temporal-difference form: r = +-w (G^d_i - G^d_o), G^d of the positions (forward: roles 1c / 2c)
            const double w = (double)wf;
            const bool bw = (meta & 2) != 0;
            const double* lg = bw ? lgb : lgf;
            const double g0 = (bw ? gbo0 : gfo0) - lg[3 * o[0]], g1 = (bw ? gbo1 : gfo1) - lg[3 * o[0] + 1], g2 = (bw ? gbo2 : gfo2) - lg[3 * o[0] + 2];
            double rho0, rho1;
            huber(P.info_spatial * ((w * g0) * (w * g0) + (w * g1) * (w * g1) + (w * g2) * (w * g2)), P.delta_spatial, rho0, rho1);
            if (meta & DM_COUNT) chi += rho0;
            const double sfac = (rfix ? 0.0 : 1.0) * rho1 * P.info_spatial * w * w;
            P.d_s[idx] = sfac;
            D[0] += sfac; D[3] += sfac; D[5] += sfac;
            bb[0] -= sfac * g0; bb[1] -= sfac * g1; bb[2] -= sfac * g2;
            return;
        }
        if (!LIN && !(meta & DM_COUNT)) return;           // chi2 only: counted from one of the edge's rows
        const double w = (double)wf;
        if (LDS) {
            // LDS records list the others in canonical order (engine_create): sg_i * sum_k sg_k x_k =
            // (x_i - x[o1]) - (x[o0] - x[o2]) for every role, absent vertices read as zero
            double v[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool has = o[k] >= 0;
                const int ok = has ? o[k] : 0;
                v[k][0] = has ? xl[3 * ok] : 0.0; v[k][1] = has ? xl[3 * ok + 1] : 0.0; v[k][2] = has ? xl[3 * ok + 2] : 0.0;
            }
            const double g0 = (xo0 - v[1][0]) - (v[0][0] - v[2][0]), g1 = (xo1 - v[1][1]) - (v[0][1] - v[2][1]), g2 = (xo2 - v[1][2]) - (v[0][2] - v[2][2]);
            const double r0 = w * g0, r1 = w * g1, r2 = w * g2;
            double rho0, rho1;
            huber(P.info_spatial * (r0 * r0 + r1 * r1 + r2 * r2), P.delta_spatial, rho0, rho1);
            if (meta & DM_COUNT) chi += rho0;
            if (LIN) {
                const double sfac = (rfix ? 0.0 : 1.0) * rho1 * P.info_spatial * w * w;
                P.d_s[idx] = sfac;
                D[0] += sfac; D[3] += sfac; D[5] += sfac;
                bb[0] -= sfac * g0; bb[1] -= sfac * g1; bb[2] -= sfac * g2;
            }
            return;
        }
        const int role = meta & 3;
        const double sgn_own = damper_sign(role);
        double s0 = sgn_own * xo0, s1 = sgn_own * xo1, s2 = sgn_own * xo2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double sg = damper_sign(k + (k >= role ? 1 : 0));   // role of the k-th other vertex
            if (o[k] >= 0) {
                s0 += sg * xl[3 * o[k]]; s1 += sg * xl[3 * o[k] + 1]; s2 += sg * xl[3 * o[k] + 2];
            }
        }
        const double r0 = w * s0, r1 = w * s1, r2 = w * s2;
        double rho0, rho1;
        huber(P.info_spatial * (r0 * r0 + r1 * r1 + r2 * r2), P.delta_spatial, rho0, rho1);
        if (meta & DM_COUNT) chi += rho0;
        if (LIN) {
            const double fx = rfix ? 0.0 : 1.0;
            const double sfac = fx * rho1 * P.info_spatial * w * w;
            P.d_s[idx] = sfac;
            D[0] += sfac; D[3] += sfac; D[5] += sfac;
            const double c = fx * sgn_own * rho1 * P.info_spatial * w;
            bb[0] -= c * r0; bb[1] -= c * r1; bb[2] -= c * r2;
        }
    };
    {
        const int beg = P.sd_ptr[slice], end = P.sd_ptr[slice + 1];
        if (LDS) {
            auto do_dh = [&](const uint2* hdr, const float* ww, int idx) {
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int m16 = (int)(hdr[q].y >> 16);
                    if (m16 == REC_NONE) continue;
                    const int r0 = (int)(hdr[q].x & 0xFFFFu), r1 = (int)(hdr[q].x >> 16), r2 = (int)(hdr[q].y & 0xFFFFu);
                    const int o[3] = {r0 == REC_NONE ? -1 : r0, r1 == REC_NONE ? -1 : r1, r2 == REC_NONE ? -1 : r2};
                    damper(idx + 64 * q, m16, o, ww[q]);
                }
            };
            for (int base = beg; base < end; base += 128 * U) {
                load_dh(dhB, dwB, base + 64 * U + lane);
                do_dh(dhA, dwA, base + lane);
                load_dh(dhA, dwA, base + 128 * U + lane);
                do_dh(dhB, dwB, base + 64 * U + lane);
            }
        } else {
            for (int idx = beg + lane; idx < end; idx += 64) {
                const int meta = P.d_meta[idx];
                if (meta < 0) continue;
                const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
                damper(idx, meta, o, P.d_w[idx]);
            }
        }
    }
    // ---- reprojection edge of the row (LDS path, linearisation; after the incidence loops: the staged positions are dead by
    // then and their LDS holds the operands of the pose-block product): ReprojectionError / ...WithDeformation
    // computeError + linearizeOplus (reference reprojection_error.cc:32-64, reprojection_error_with_deformation.cc:37-68)
    // and its quadratic form (base_fixed_sized_edge.hpp:49-63).  The first two lanes of a row take one residual
    // component each (T = 1: one lane takes both): the H_pp / b_p partials and the row's diagonal block are sums
    // over the two components anyway, and they leave through the reductions that follow.
    if (LIN && LDS) {
        constexpr int NRR = T == 1 ? 2 : 1;                 // residual components per participating lane
        const int kf = P.grp_pose[row / ROW_ALIGN];
        const bool pfix = P.pose_fixed[kf] != 0;
        // an edge whose vertices are all fixed is not part of the optimisation (sparse_optimizer.cpp:236)
        const bool active = (rf & RF_OBS) && (rf & RF_REPROJ_ACTIVE) && !(pfix && rfix);
        RowRec rc;
#pragma unroll
        for (int q = 0; q < 6; ++q) rc.J[q] = 0.f;
        rc.w = 0;
        double Jp[NRR][6], rres[NRR], w = 0, chi_r = 0;
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            rres[a] = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) Jp[a][q] = 0;
        }
        if (active && t < (T == 1 ? 1 : 2)) {
            const Pose Tcw = P.lin_pose[kf];
            double Rm[9];
            quat_to_R(Tcw.q, Rm);
            const double px = Rm[0] * xs0 + Rm[1] * xs1 + Rm[2] * xs2 + Tcw.t[0];
            const double py = Rm[3] * xs0 + Rm[4] * xs1 + Rm[5] * xs2 + Tcw.t[1];
            const double pz = Rm[6] * xs0 + Rm[7] * xs1 + Rm[8] * xs2 + Tcw.t[2];
            float u, v, Jf[6];
            project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
            projection_jacobian_f32(P.cam, (float)px, (float)py, (float)pz, Jf);
            const double r2[2] = {(double)P.uv[2 * row] - (double)u, (double)P.uv[2 * row + 1] - (double)v};
            double rho0, rho1;
            huber(P.info_reproj * (r2[0] * r2[0] + r2[1] * r2[1]), P.delta_reproj, rho0, rho1);
            if (t == 0) chi_r = rho0;
            w = rho1 * P.info_reproj;
            const double pm = pfix ? 0.0 : 1.0, lm = rfix ? 0.0 : 1.0;
#pragma unroll
            for (int a = 0; a < NRR; ++a) {
                const int rr = T == 1 ? a : t;
                const double j0 = -(double)(rr ? Jf[3] : Jf[0]), j1 = -(double)(rr ? Jf[4] : Jf[1]), j2 = -(double)(rr ? Jf[5] : Jf[2]);
                const double r = rr ? r2[1] : r2[0];
                rres[a] = r;
                Jp[a][0] = pm * (-j1 * pz + j2 * py);
                Jp[a][1] = pm * (j0 * pz - j2 * px);
                Jp[a][2] = pm * (-j0 * py + j1 * px);
                Jp[a][3] = pm * j0; Jp[a][4] = pm * j1; Jp[a][5] = pm * j2;
                const double Jl0 = lm * (j0 * Rm[0] + j1 * Rm[3] + j2 * Rm[6]);
                const double Jl1 = lm * (j0 * Rm[1] + j1 * Rm[4] + j2 * Rm[7]);
                const double Jl2 = lm * (j0 * Rm[2] + j1 * Rm[5] + j2 * Rm[8]);
                D[0] += w * (Jl0 * Jl0); D[1] += w * (Jl0 * Jl1); D[2] += w * (Jl0 * Jl2);
                D[3] += w * (Jl1 * Jl1); D[4] += w * (Jl1 * Jl2); D[5] += w * (Jl2 * Jl2);
                bb[0] -= w * (Jl0 * r); bb[1] -= w * (Jl1 * r); bb[2] -= w * (Jl2 * r);
            }
            if (t == 0) {
                // factored form: the PCG kernels rebuild J_l, J_p from these 32 bytes
#pragma unroll
                for (int q = 0; q < 6; ++q) rc.J[q] = Jf[q];
                rc.w = lm * w;
            }
        }
        if (t == 0) P.rowrec[row] = rc;
        // H_pp (21 packed) / b_p (6) partials of the tile = the dense pose block of the normal equations,
        //   [H_pp | b_p] = sum_k (w_k Jp_k)^T [Jp_k | -r_k]     (k = the wave's 64 lanes: one residual component each)
        // a (6 x 64) x (64 x 7) product per wave: sixteen v_mfma_f64_16x16x4 instead of 27 wave reductions.  Every lane
        // leaves its seven values and its weight in LDS; for MFMA m lane (e = lane % 16, kk = lane / 16) supplies
        // A[e][kk] = w Jp[e] and B[kk][e] = Jp[e] | -r of lane 4 m + kk.  Result layout of the f64 form: register g of a lane
        // holds C[row = kk + 4 g][col = e].
        {
            typedef double v4d __attribute__((ext_vector_type(4)));
            __syncthreads();                                       // every wave is done with the staged positions
            double* mb = dyn + wave * 512;                         // [64 lanes][8], the wave's own
            const int e = lane & 15, kk = lane >> 4;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < NRR; ++a) {
                // (the buffer is the wave's own: its LDS operations execute in order, a wave-level fence is all it takes)
                if (a > 0) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
                for (int q = 0; q < 6; ++q) mb[lane * 8 + q] = Jp[a][q];
                mb[lane * 8 + 6] = -rres[a];
                mb[lane * 8 + 7] = w;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
                for (int m = 0; m < 16; ++m) {
                    const int kq = 4 * m + kk;
                    const double val = e < 7 ? mb[kq * 8 + e] : 0.0;
                    const double av = e < 6 ? mb[kq * 8 + 7] * val : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, val, acc, 0, 0, 0);
                }
            }
            auto put = [&](int row, double c) {
                if (row < 6) {
                    if (e >= row && e < 6) lds28[wave * 28 + row * 6 - (row * (row - 1)) / 2 + (e - row)] = c;
                    if (e == 6) lds28[wave * 28 + 21 + row] = c;
                }
            };
            put(kk, acc.x);
            put(4 + kk, acc.y);
        }
        {
            const double sm = wave_sum(chi_r);
            if (lane == 0) lds28[wave * 28 + 27] = sm;
        }
        __syncthreads();
        if (tid < 28) P.part_lin[(size_t)b * 32 + tid] = lds28[tid] + lds28[28 + tid] + lds28[56 + tid] + lds28[84 + tid];
    }
    double part[2];
    part[0] = chi;
    part[1] = 0;
    if (LIN) {
#pragma unroll
        for (int k = 0; k < 6; ++k) D[k] = sub_sum_t<T>(D[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) bb[k] = sub_sum_t<T>(bb[k]);
        if (t == 0) {
            double* Dr = P.D + 6 * (size_t)row;
            double dd[6];
            if (LDS) {                                      // the row's whole diagonal block and gradient: written once
#pragma unroll
                for (int k = 0; k < 6; ++k) { dd[k] = D[k]; Dr[k] = dd[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) P.bl[3 * row + k] = bb[k];
            } else {                                        // gather path: k_reproj<true> left the reprojection part
#pragma unroll
                for (int k = 0; k < 6; ++k) { dd[k] = Dr[k] + D[k]; Dr[k] = dd[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) P.bl[3 * row + k] += bb[k];
            }
            part[1] = fmax(fabs(dd[0]), fmax(fabs(dd[3]), fabs(dd[5])));
        }
    }
    double c = wave_sum(part[0]);
    double m = part[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (lane == 0) { lds[wave * 2] = c; lds[wave * 2 + 1] = m; }
    __syncthreads();
    if (tid == 0) {
        P.part_reg[2 * (size_t)b] = lds[0] + lds[2] + lds[4] + lds[6];
        P.part_reg[2 * (size_t)b + 1] = fmax(fmax(lds[1], lds[3]), fmax(lds[5], lds[7]));
    }
}

// =====================================================================================
// The same fused pass for PLAIN deformable-BA windows (LocalDeformableBundleAdjustment as the reference builds it,
// OPT:927-1137: nothing fixed, no masks, no offsets, every damper with its four vertices, springs without a robust
// kernel) on the LDS path -- the case the benchmark configurations C2-C5 are.  Same streams in, same outputs as
// k_reg<T, true, true> (per-incidence factors, row factors, D, b_l, tile partials); what differs is the instruction
// count: the incidence bodies are branch-free (padding slots run on safe operands and contribute zeros) and free of the
// mask / level / fixed-vertex logic, and the spring's 1/d0, 1/d, 1/sqrt(d) come from the hardware estimates plus
// Newton steps written out (positive normal operands by construction) instead of the library's division and sqrt
// sequences.  Reference arithmetic: position_regularizer.cc:32-61 (Jacobian as written), spatial_regularizer.cc:32-59,
// reprojection_error.cc:32-64, base_fixed_sized_edge.hpp:49-133.
// =====================================================================================
__device__ inline double fast_rcp_pos(double a) {                   // 1/a, a > 0 normal: v_rcp_f64 + two Newton steps (<= 1 ulp)
    double y = __builtin_amdgcn_rcp(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    return fma(y, e, y);
}
__device__ inline double fast_rsqrt_pos(double a) {                 // a^-1/2, a > 0 normal: v_rsq_f64 + one third-order step
    const double y = __builtin_amdgcn_rsq(a);
    const double e = fma(-a * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
__device__ inline double fast_sqrt_pos(double a) {                  // a^1/2, a > 0 normal: rsq estimate + two coupled Newton steps
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double e2 = fma(-g, g, a);
    return fma(e2, h, g);
}

// The two per-incidence factors of a plain BA window, in ONE place: k_lin_plain forms them for the diagonal blocks and (unless
// Dev::rc says otherwise) stores them for the operator; with Dev::rc the operator (k_spmv_f<.., RCS / RCD>) re-forms them from the
// staged linearisation point with these very operation sequences (explicit fma, contraction off: the same bits in either kernel).
//   spring (BA form, position_regularizer.cc:51-60, no kernel): qc = info (2 k / (d0 sqrt(d)))^2 = info (2 k / d0)^2 / d
//   damper (spatial_regularizer.cc:32-59 + Huber): s = rho'(e) info w^2, e = info |w g|^2
__device__ inline double sq3(double a, double b, double c) { return fma(c, c, fma(b, b, a * a)); }
__device__ inline double spring_qc(double d2, double d0, double ks, double ip) {
#pragma clang fp contract(off)
    const double c = 2.0 * ks * fast_rcp_pos(d0);
    return ip * (c * c) * fast_rsqrt_pos(d2);
}
__device__ inline double damper_s(double g0, double g1, double g2, double w, double isp, double dsp, double& rho0) {
#pragma clang fp contract(off)
    const double r0 = w * g0, r1 = w * g1, r2 = w * g2;
    const double e = isp * sq3(r0, r1, r2), dsqr = dsp * dsp;
    const bool act = dsp > 0 && e > dsqr;                          // Huber active: rho' = delta / sqrt(e), rho = 2 delta sqrt(e) - delta^2
    const double ea = act ? e : 1.0, y = fast_rsqrt_pos(ea);
    rho0 = act ? 2.0 * (ea * y) * dsp - dsqr : e;
    return (act ? dsp * y : 1.0) * isp * w * w;
}

template <int T, int OCC = 4, int CAM = -1, bool TPC = false, int EXP = 0, bool H4 = false, bool RCS = false, bool RCD = false, int NBT = 4, bool NT = false>   // NBT: slots per lane and request batch; NT: non-temporal stream accesses (probe variants); RCS / RCD: the spring / damper factors are not stored (Dev::rc: the operator re-forms them); EXP != 0: timing experiments with a piece removed (NRS_LIN_EXP; wrong results); H4: 4-byte damper headers (Dev::d_h4)
__global__ __launch_bounds__(BLK, OCC) void k_lin_plain(Dev P, const double* __restrict__ xl_g, int cls) {
    __shared__ double mfb[4 * 128];                                // pose-block operands: 1 KB per wave
    __shared__ double spose[8];                                    // the tile's pose (q, t): fetched during staging, read after the loops
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    int b = xcd_tile(blockIdx.x, P.sh_nt[cls]);
    if (b >= P.sh_nt[cls]) return;
    b = P.tile_list[(cls ? P.n_tiles_cls[0] : 0) + P.sh_t0[cls] + b];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: slice bounds and loop conditions are scalar)
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    auto stamp = [&](int k) { if (P.dbg_clk && lane == 0) P.dbg_clk[(size_t)slice * 8 + k] = wall_clock64(); };
    stamp(0);
    // this lane's share of its row's incidences (k = t, t + T, ...): slots beyond it are padding -- not requested, not stored
    const uint32_t rcn = P.row_cnt[row];
    const int my_s = ((int)(rcn & 0xFFFFu) + T - 1 - t) / T, my_d = ((int)(rcn >> 16) + T - 1 - t) / T;
    if (tid < 7) spose[tid] = reinterpret_cast<const double*>(P.lin_pose + P.grp_pose[row / ROW_ALIGN])[tid];   // (a tile never straddles keyframes)
    const int s_beg = P.ss_ptr[slice], s_end = P.ss_ptr[slice + 1];
    const int d_beg = P.sd_ptr[slice], d_end = EXP == 5 ? P.sd_ptr[slice] + (((P.sd_ptr[slice + 1] - P.sd_ptr[slice]) / 64 + 1) / 2) * 64 : P.sd_ptr[slice + 1];   // (EXP 5: half of the damper slots -- what a two-incidence form of the dampers would stream)
    // Record requests go out in BATCHES of NB slots per lane and stream: all NB requests of a batch are in flight together,
    // then the slots are consumed one by one -- a wave waits out one memory round trip per batch instead of one per pair of
    // slots (two-slot double buffering spent ~85 % of the loops waiting: 13.6 us of a 28.6 us wave at C4), and with four
    // waves per SIMD the others compute meanwhile.  Nothing is carried across loop iterations in registers: software
    // pipelining across the back edge makes the compiler park register copies behind a vmcnt(0) there.  The loads are
    // UNCONDITIONAL (slots past the slice's end read a clamped index and count as padding when consumed) and nothing
    // touches the loaded registers before consume(): with a predicated load the compiler sinks the unpacking behind the
    // load and the wave waits out the full latency of every single request (rounds 1 and 2).
    constexpr int NB = NBT;                                        // (6 or 8 slots per lane and batch: 111 / 238 spilled VGPRs at 4 waves per SIMD, round 5)
    uint32_t rs_om[NB], rs_d0[NB];
    auto req_s = [&](int q, int idx) {
        if (EXP == 3) { rs_om[q] = (uint32_t)(idx & 63) | 0x10000u; rs_d0[q] = 0x3F800000u; return; }
        rs_om[q] = 0xFFFFu; rs_d0[q] = 0x3F800000u;                  // padding: no neighbour, rest length 1
        if ((idx - s_beg - lane) / 64 < my_s) {
            if (NT) { rs_om[q] = __builtin_nontemporal_load(P.s_om + idx); rs_d0[q] = __float_as_uint(__builtin_nontemporal_load(P.s_d0 + idx)); }
            else { rs_om[q] = P.s_om[idx]; rs_d0[q] = __float_as_uint(P.s_d0[idx]); }
        }
    };
#pragma unroll
    for (int q = 0; q < NB; ++q) req_s(q, s_beg + lane + 64 * q);  // the first spring batch rides on the staging loads
    __builtin_amdgcn_sched_barrier(0);
    double* lx = dyn;
    if (P.fused) stage_rows(P, b, tid, xl_g, nullptr, lx);          // (the fused path keeps its own fixed-stride list)
    else stage_rows<true>(P, b, tid, xl_g, nullptr, lx);
    __syncthreads();
    stamp(1);
    const int self = row - b * P.tile_rows;
    const double xo0 = lx[3 * self], xo1 = lx[3 * self + 1], xo2 = lx[3 * self + 2];
    double D0 = 0, D1 = 0, D2 = 0, D3 = 0, D4 = 0, D5 = 0, bb0 = 0, bb1 = 0, bb2 = 0, chi = 0;
    // ---- springs: r = k (d - d0) / d0, J = cg (x_i - x_j)^T with cg = (k / d0) 2 / sqrt(d) as the reference writes it
    // (position_regularizer.cc:51-60) or k / (d0 d) (tracking form); information info_pos, no kernel
    const double ks = P.k_spring, ip = P.info_pos;
    // (a slot's raw words are unpacked BEFORE its successor is requested into the same registers -- the raw value is dead
    // by then, so the ring needs no register copies, which the compiler otherwise parks behind a vmcnt(0) at the loop's end)
    auto do_s = [&](int o16, bool live, bool count, float d0f, int idx) {
        const bool pad = !live || o16 == REC_NONE;
        const int o = EXP == 2 ? self : (pad ? self : o16) & 0xFFFF;   // (16 bits: the LDS offset is then one v_mad_u32_u24)
        const double v0 = xo0 - lx[3 * o], v1 = xo1 - lx[3 * o + 1], v2 = xo2 - lx[3 * o + 2];
        double d2 = sq3(v0, v1, v2);
        d2 = pad ? 1.0 : d2;
        const double d0 = pad ? 1.0 : (double)d0f;
        const double rs = fast_rsqrt_pos(d2), id0 = fast_rcp_pos(d0);
        const double r = ks * fma(d2, rs, -d0) * id0;
        const double cg = 2.0 * ks * id0 * fast_sqrt_pos(rs);       // (plain windows are BA windows: spring_form 0, checked by engine_create)
        const double qc = pad ? 0.0 : spring_qc(d2, d0, ks, ip);
        if (live && EXP != 1 && !RCS) { if (NT) __builtin_nontemporal_store(qc, P.s_qc + idx); else P.s_qc[idx] = qc; }                    // (whole lines: padding slots inside the slice are written too -- a lane-masked store leaves partial lines, which cost the memory side a read-modify-write)
        chi += count && !pad ? ip * r * r : 0.0;
        const double t0 = qc * v0, t1 = qc * v1, t2 = qc * v2;
        D0 = fma(t0, v0, D0); D1 = fma(t0, v1, D1); D2 = fma(t0, v2, D2);
        D3 = fma(t1, v1, D3); D4 = fma(t1, v2, D4); D5 = fma(t2, v2, D5);
        const double qr = ip * r * cg;                              // (v = 0 on padding slots)
        bb0 = fma(-qr, v0, bb0); bb1 = fma(-qr, v1, bb1); bb2 = fma(-qr, v2, bb2);
    };
    for (int ub = s_beg; ub < (EXP == 4 ? s_beg : s_end); ub += 64 * NB) {   // wave-uniform trip count
        const int base = ub + lane;
        if (ub != s_beg) {
#pragma unroll
            for (int q = 0; q < NB; ++q) req_s(q, base + 64 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t om = consume(rs_om[q]);
            const float d0f = __uint_as_float(consume(rs_d0[q]));
            do_s((int)(om & 0xFFFFu), base + 64 * q < s_end, (om & ((uint32_t)SR_COUNT << 16)) != 0, d0f, base + 64 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2);
    // ---- dampers: r = w ((x1n - x1c) - (x2n - x2c)); the records list the three others in canonical order
    // (engine_create), so that sg_i * sum_k sg_k x_k = (x_i - x[o1]) - (x[o0] - x[o2]) for every role
    // x_i - x_next(i) and x_i - x_prev(i): one of the two is the first half of every damper residual of this row (TPC: read
    // once here, behind the spring loop whose registers are free by now, instead of once per incidence -- two of the three gathers of a damper were the row's own partners)
    double en0 = 0, en1 = 0, en2 = 0, ep0 = 0, ep1 = 0, ep2 = 0;
    if (TPC) {
        const uint32_t tp = P.row_tp[row];                         // the row's temporal partners (tile-local ids); requested here for the same reason as the flag and the keypoint below
        const int tn = (int)(tp & 0xFFFFu) == REC_NONE ? self : (int)(tp & 0xFFFFu), tq = (int)(tp >> 16) == REC_NONE ? self : (int)(tp >> 16);
        en0 = xo0 - lx[3 * tn]; en1 = xo1 - lx[3 * tn + 1]; en2 = xo2 - lx[3 * tn + 2];
        ep0 = xo0 - lx[3 * tq]; ep1 = xo1 - lx[3 * tq + 1]; ep2 = xo2 - lx[3 * tq + 2];
    }
    const double isp = P.info_spatial, dsp = P.delta_spatial;
    uint32_t rd_w[NB];
    uint2 rd_h[NB];
    auto req_d = [&](int q, int idx) {
        if (EXP == 3) { rd_h[q] = make_uint2((uint32_t)(idx & 63) | ((uint32_t)((idx + 1) & 63) << 16), (uint32_t)((idx + 2) & 63) | 0x80000u); rd_w[q] = 0x3F800000u; return; }
        rd_h[q] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); rd_w[q] = 0u;
        if (H4) { if ((idx - d_beg - lane) / 64 < my_d) {
            if (NT) { rd_h[q].x = __builtin_nontemporal_load(P.d_h4 + idx); rd_w[q] = __float_as_uint(__builtin_nontemporal_load(P.d_w + idx)); }
            else { rd_h[q].x = P.d_h4[idx]; rd_w[q] = __float_as_uint(P.d_w[idx]); } } }
        else if ((idx - d_beg - lane) / 64 < my_d) { rd_h[q] = P.d_hdr[idx]; rd_w[q] = __float_as_uint(P.d_w[idx]); }
    };
    auto do_d = [&](uint32_t hx, uint32_t hy, float wf, int idx) {
        const uint32_t m16 = H4 ? (hx == 0xFFFFFFFFu ? (uint32_t)REC_NONE : hx >> 24) : hy >> 16;
        const bool live = idx < d_end;
        const bool pad = !live || m16 == REC_NONE;
        const int o0 = EXP == 2 ? self : (pad ? self : (int)(hx & (H4 ? 0xFFFu : 0xFFFFu))) & 0xFFFF, o1 = EXP == 2 || H4 ? self : (pad ? self : (int)(hx >> 16)) & 0xFFFF,
                  o2 = EXP == 2 ? self : (pad ? self : (int)(H4 ? (hx >> 12) & 0xFFFu : hy & 0xFFFFu)) & 0xFFFF;
        double g0, g1, g2;
        if (TPC) {
            const bool fwd = (m16 & 2u) == 0;                        // roles 1c / 2c: the partner is in the next keyframe
            g0 = (fwd ? en0 : ep0) - (lx[3 * o0] - lx[3 * o2]);
            g1 = (fwd ? en1 : ep1) - (lx[3 * o0 + 1] - lx[3 * o2 + 1]);
            g2 = (fwd ? en2 : ep2) - (lx[3 * o0 + 2] - lx[3 * o2 + 2]);
        } else {
            g0 = (xo0 - lx[3 * o1]) - (lx[3 * o0] - lx[3 * o2]);
            g1 = (xo1 - lx[3 * o1 + 1]) - (lx[3 * o0 + 1] - lx[3 * o2 + 1]);
            g2 = (xo2 - lx[3 * o1 + 2]) - (lx[3 * o0 + 2] - lx[3 * o2 + 2]);
        }
        const double w = pad ? 0.0 : (double)wf;
        double rho0;
        const double sfac = damper_s(g0, g1, g2, w, isp, dsp, rho0);
        chi += (m16 & DM_COUNT) ? rho0 : 0.0;                      // (padding: rho0 = 0)
        if (live && EXP != 1 && !RCD) { if (NT) __builtin_nontemporal_store(sfac, P.d_s + idx); else P.d_s[idx] = sfac; }
        D0 += sfac; D3 += sfac; D5 += sfac;
        bb0 = fma(-sfac, g0, bb0); bb1 = fma(-sfac, g1, bb1); bb2 = fma(-sfac, g2, bb2);
    };
    for (int ub = d_beg; ub < (EXP == 4 ? d_beg : d_end); ub += 64 * NB) {
        const int base = ub + lane;
#pragma unroll
        for (int q = 0; q < NB; ++q) req_d(q, base + 64 * q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const uint32_t hx = consume(rd_h[q].x), hy = H4 ? 0u : consume(rd_h[q].y);
            const float wf = __uint_as_float(consume(rd_w[q]));
            do_d(hx, hy, wf, base + 64 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(3);
    // ---- reprojection edge of the row (as in k_reg: the first two lanes of a row take one residual component each).
    // The pose is the tile's (a tile never straddles keyframes): scalar loads, nothing waits on a per-lane gather.
    constexpr int NRR = T == 1 ? 2 : 1;
    // (the row's flag and keypoint are requested HERE, not up front: held across the loops they are spilled at four waves per SIMD -- 20 bytes of
    // scratch written and read back per lane, which in the HBM regime is traffic; the other waves of the SIMD cover this round trip)
    const int rf = P.rflag[row];
    const float uvx = P.uv[2 * row], uvy = P.uv[2 * row + 1];
    const bool active = (rf & RF_OBS) && (rf & RF_REPROJ_ACTIVE);
    RowRec rc;
#pragma unroll
    for (int q = 0; q < 6; ++q) rc.J[q] = 0.f;
    rc.w = 0;
    double Jp[NRR][6], rres[NRR], w = 0, chi_r = 0;
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        rres[a] = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) Jp[a][q] = 0;
    }
    if (active && t < (T == 1 ? 1 : 2)) {
        Pose Tcw;
#pragma unroll
        for (int q = 0; q < 4; ++q) Tcw.q[q] = spose[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) Tcw.t[q] = spose[4 + q];
        double Rm[9];
        quat_to_R(Tcw.q, Rm);
        const double px = Rm[0] * xo0 + Rm[1] * xo1 + Rm[2] * xo2 + Tcw.t[0];
        const double py = Rm[3] * xo0 + Rm[4] * xo1 + Rm[5] * xo2 + Tcw.t[1];
        const double pz = Rm[6] * xo0 + Rm[7] * xo1 + Rm[8] * xo2 + Tcw.t[2];
        float u, v, Jf[6];
        Cam cam = P.cam;
        if (CAM >= 0) cam.model = CAM;                             // (one instance per camera model: the pinhole one carries no fisheye code or registers)
        project_f32(cam, (float)px, (float)py, (float)pz, u, v);
        projection_jacobian_f32(cam, (float)px, (float)py, (float)pz, Jf);
        const double r2[2] = {(double)uvx - (double)u, (double)uvy - (double)v};
        double rho0, rho1;
        huber(P.info_reproj * (r2[0] * r2[0] + r2[1] * r2[1]), P.delta_reproj, rho0, rho1);
        if (t == 0) chi_r = rho0;
        w = rho1 * P.info_reproj;
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            const int rr = T == 1 ? a : t;
            const double j0 = -(double)(rr ? Jf[3] : Jf[0]), j1 = -(double)(rr ? Jf[4] : Jf[1]), j2 = -(double)(rr ? Jf[5] : Jf[2]);
            const double r = rr ? r2[1] : r2[0];
            rres[a] = r;
            Jp[a][0] = -j1 * pz + j2 * py;
            Jp[a][1] = j0 * pz - j2 * px;
            Jp[a][2] = -j0 * py + j1 * px;
            Jp[a][3] = j0; Jp[a][4] = j1; Jp[a][5] = j2;
            const double Jl0 = j0 * Rm[0] + j1 * Rm[3] + j2 * Rm[6];
            const double Jl1 = j0 * Rm[1] + j1 * Rm[4] + j2 * Rm[7];
            const double Jl2 = j0 * Rm[2] + j1 * Rm[5] + j2 * Rm[8];
            D0 += w * (Jl0 * Jl0); D1 += w * (Jl0 * Jl1); D2 += w * (Jl0 * Jl2);
            D3 += w * (Jl1 * Jl1); D4 += w * (Jl1 * Jl2); D5 += w * (Jl2 * Jl2);
            bb0 -= w * (Jl0 * r); bb1 -= w * (Jl1 * r); bb2 -= w * (Jl2 * r);
        }
        if (t == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) rc.J[q] = Jf[q];
            rc.w = w;
        }
    }
    if (t == 0) P.rowrec[row] = rc;
    stamp(4);
    // [H_pp | b_p] partials of the SLICE on the matrix cores (the product of k_reg: sixteen v_mfma_f64_16x16x4 per wave),
    // operands through a 1 KB region of the wave's own, sixteen lanes at a time: no workgroup barrier anywhere behind the
    // staging one -- the four waves of a tile carry different incidence counts (rows are sorted by them) and finish apart.
    // Every wave leaves its own partial slot: [0..27) H_pp / b_p, [27] chi2 of the reprojection edges, [28] chi2 of the
    // regularisers, [29] max |diagonal| of its rows; k_pose_sums adds the slices of a pose up in a fixed order.
    double* slot = P.part_lin + (size_t)slice * 32;
    {
        typedef double v4d __attribute__((ext_vector_type(4)));
        double* mb = mfb + wave * 128;                             // [16 lanes][8]
        const int e = lane & 15, kk = lane >> 4;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();                   // (the previous group's reads are done)
                if (kk == c4) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) mb[e * 8 + q] = Jp[a][q];
                    mb[e * 8 + 6] = -rres[a];
                    mb[e * 8 + 7] = w;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int m = 0; m < 4; ++m) {                      // MFMA 4 c4 + m: k index 4 m + kk of this group of sixteen lanes
                    const int kq = 4 * m + kk;
                    const double val = e < 7 ? mb[kq * 8 + e] : 0.0;
                    const double av = e < 6 ? mb[kq * 8 + 7] * val : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, val, acc, 0, 0, 0);
                }
            }
        }
        auto put = [&](int prow, double c) {
            if (prow < 6) {
                if (e >= prow && e < 6) slot[prow * 6 - (prow * (prow - 1)) / 2 + (e - prow)] = c;
                if (e == 6) slot[21 + prow] = c;
            }
        };
        put(kk, acc.x);
        put(4 + kk, acc.y);
    }
    D0 = sub_sum_t<T>(D0); D1 = sub_sum_t<T>(D1); D2 = sub_sum_t<T>(D2);
    D3 = sub_sum_t<T>(D3); D4 = sub_sum_t<T>(D4); D5 = sub_sum_t<T>(D5);
    bb0 = sub_sum_t<T>(bb0); bb1 = sub_sum_t<T>(bb1); bb2 = sub_sum_t<T>(bb2);
    double md = 0;
    if (t == 0) {
        double* Dr = P.D + 6 * (size_t)row;
        Dr[0] = D0; Dr[1] = D1; Dr[2] = D2; Dr[3] = D3; Dr[4] = D4; Dr[5] = D5;
        P.bl[3 * row] = bb0; P.bl[3 * row + 1] = bb1; P.bl[3 * row + 2] = bb2;
        md = fmax(fabs(D0), fmax(fabs(D3), fabs(D5)));
    }
    const double cr = wave_sum(chi_r), cg2 = wave_sum(chi);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmax(md, __shfl_xor(md, off, 64));
    if (lane == 0) { slot[27] = cr; slot[28] = cg2; slot[29] = md; }
    stamp(5);
}

// =====================================================================================
// chi2 of a trial state, one thread per edge (BA windows: no masks, nothing fixed).  Same residuals and
// Huber as k_reg; only the order of the sum differs from the incidence-ordered one (relative 1e-16).
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_chi_edges(Dev P, const double* __restrict__ xl) {
    __shared__ double lds[4];
    // a fixed number of workgroups (<= 2048, so that the single-workgroup k_finalize has few partials to sum: on C4
    // 428k partials made it 760 us per trial), each over a contiguous range of edges in a fixed order
    const int tid = threadIdx.x, n = P.ec_nsp + P.ec_ndm;
    const int per = (int)((((int64_t)n + P.ec_nblk - 1) / P.ec_nblk + BLK - 1) / BLK) * BLK;
    const int64_t beg = (int64_t)blockIdx.x * per;
    const int end = (int)(beg + per < n ? beg + per : n);
    double rho[1] = {0};
    for (int i = (int)beg + tid; i < end; i += BLK) {
        double r0v, rho1;
        if (i < P.ec_nsp) {
            const EcSpring s = P.ec_sp[i];
            const double v0 = xl[3 * (size_t)s.a] - xl[3 * (size_t)s.b], v1 = xl[3 * (size_t)s.a + 1] - xl[3 * (size_t)s.b + 1],
                         v2 = xl[3 * (size_t)s.a + 2] - xl[3 * (size_t)s.b + 2];
            const double d = sqrt(v0 * v0 + v1 * v1 + v2 * v2), d0 = (double)s.d0;
            const double r = P.k_spring * (d - d0) / d0;
            huber(P.info_pos * r * r, P.delta_pos, r0v, rho1);
        } else {
            const EcDamper dd = P.ec_dm[i - P.ec_nsp];
            const double w = (double)P.ec_w[i - P.ec_nsp];
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (dd.r[k] >= 0) {
                    const double sg = damper_sign(k);
                    s0 += sg * xl[3 * (size_t)dd.r[k]]; s1 += sg * xl[3 * (size_t)dd.r[k] + 1]; s2 += sg * xl[3 * (size_t)dd.r[k] + 2];
                }
            }
            const double r0 = w * s0, r1 = w * s1, r2 = w * s2;
            huber(P.info_spatial * (r0 * r0 + r1 * r1 + r2 * r2), P.delta_spatial, r0v, rho1);
        }
        rho[0] += r0v;
    }
    block_sum<1>(rho, lds, tid & 63, tid >> 6);
    if (tid == 0) P.part_ec[blockIdx.x] = rho[0];
}

// =====================================================================================
// finalize: fixed-order sums of the partials.  LIN: H_pp, b_p, chi2, max diag.  else chi2, scale.
// =====================================================================================
// Publication to the host: everything goes into mapped host memory from ONE thread, then a system-scope
// fence, then the sequence number the host is polling for (h_flags[7]) -- a host round trip is then the
// latency of that word (measured 9.5 us per {two launches + wait} against 15.2 us with
// hipStreamSynchronize, tools/micro/sync_latency.hip).
__device__ inline void publish_flags(const Dev& P, int seq, int zero = 0) {   // zero: leave the device words cleared for the next solve (directly solved engines: no memset launch per LM trial)
#pragma unroll
    for (int k = 0; k < 7; ++k) { P.h_flags[k] = P.flags[k]; if (zero) P.flags[k] = 0; }
    __threadfence_system();
    *reinterpret_cast<volatile int*>(P.h_flags + 7) = seq;
}

template <bool LIN>
__global__ __launch_bounds__(BLK) void k_finalize(Dev P, int seq, int zero_flags) {
    __shared__ double lds[4 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double chi = 0, md = 0, sc = 0;
    if (!LIN)
        for (int b = tid; b < P.n_vecblk; b += BLK) sc += P.part_apply[b];
    if (LIN && P.plain) { for (int k = tid; k < P.K; k += BLK) chi += P.part_pchi[k]; }      // k_pose_sums: chi2 per pose (reprojection + regularisers)
    else if (LIN) for (int g = tid; g < P.n_groups * P.lin_rb; g += BLK) chi += P.part_lin[(size_t)g * 32 + 27];
    else for (int g = tid; g < P.n_groups; g += BLK) chi += P.part_rchi[g];
    if (!LIN && P.ec_on) {
        for (int b = tid; b < P.ec_nblk; b += BLK) chi += P.part_ec[b];     // trial states: edge-parallel chi2
    } else if (!(LIN && P.plain)) {
        for (int b = tid; b < P.n_regblk; b += BLK) {
            chi += P.part_reg[2 * (size_t)b];
            md = fmax(md, P.part_reg[2 * (size_t)b + 1]);
        }
    }
    if (LIN)
        for (int k = tid; k < P.K; k += BLK) md = fmax(md, P.red[3 + k]);      // k_pose_sums: max |diag H_pp| (plain: and of the rows' blocks)
    if (P.sk_n > 0) {                                              // embedded mode: the skinned observations' chi2 (and what they add to the diagonal)
        for (int b = tid; b < P.sk_nblk; b += BLK) chi += P.sk_part[(size_t)b * 32 + 27];
        if (LIN && tid == 0) md = fmax(md, *P.sk_maxdiag);
    }
    double c = wave_sum(chi);
    sc = wave_sum(sc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmax(md, __shfl_xor(md, off, 64));
    if (lane == 0) { lds[wave * 3] = c; lds[wave * 3 + 1] = md; lds[wave * 3 + 2] = sc; }
    __syncthreads();
    if (tid == 0) {
        P.scal[SC_CHI] = lds[0] + lds[3] + lds[6] + lds[9];
        if (LIN) P.scal[SC_MAXDIAG] = fmax(fmax(lds[1], lds[4]), fmax(lds[7], lds[10]));
        if (!LIN) P.scal[SC_SCALE] = lds[2] + lds[5] + lds[8] + lds[11];
        // written straight into mapped host memory (no copy kernels)
        P.h_scal[SC_CHI] = P.scal[SC_CHI];
        if (LIN) P.h_scal[SC_MAXDIAG] = P.scal[SC_MAXDIAG];
        if (!LIN) P.h_scal[SC_SCALE] = P.scal[SC_SCALE];
        publish_flags(P, seq, zero_flags);
    }
}

// Sharded evaluation (multi-GPU): the same fixed-order sums over this rank's partials (slots of other
// ranks' tiles are never written and stay zero) go into a packet that is all-reduced (sum) across the
// ranks; the maxima travel as one slot per rank.  LIN: the packet also carries H_pp and b_p of the
// rank's own poses, so that after the all-reduce every rank holds the pose blocks of all poses --
// the "all-reduce of the per-block normal equations" of SURVEY.md 8(e).
template <bool LIN>
__global__ __launch_bounds__(BLK) void k_finalize_pack(Dev P) {
    __shared__ double lds[4 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = 2 + P.sh_world + (LIN ? 27 * P.K : 0);
    for (int i = tid; i < n; i += BLK) P.pk_loc[i] = 0.0;
    double chi = 0, md = 0, sc = 0;
    if (!LIN)
        for (int b = tid; b < P.n_vecblk; b += BLK) sc += P.part_apply[b];
    if (LIN && P.plain) { for (int k = P.sh_k0 + tid; k < P.sh_k0 + P.sh_nk; k += BLK) chi += P.part_pchi[k]; }   // this rank's poses
    else if (LIN) for (int g = tid; g < P.n_groups * P.lin_rb; g += BLK) chi += P.part_lin[(size_t)g * 32 + 27];
    else for (int g = tid; g < P.n_groups; g += BLK) chi += P.part_rchi[g];
    if (!LIN && P.ec_on) {
        for (int b = tid; b < P.ec_nblk; b += BLK) chi += P.part_ec[b];     // trial states: edge-parallel chi2
    } else if (!(LIN && P.plain)) {
        for (int b = tid; b < P.n_regblk; b += BLK) {
            chi += P.part_reg[2 * (size_t)b];
            md = fmax(md, P.part_reg[2 * (size_t)b + 1]);
        }
    }
    if (LIN)
        for (int k = P.sh_k0 + tid; k < P.sh_k0 + P.sh_nk; k += BLK) md = fmax(md, P.red_loc[3 + k]);
    double c = wave_sum(chi);
    sc = wave_sum(sc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) md = fmax(md, __shfl_xor(md, off, 64));
    if (lane == 0) { lds[wave * 3] = c; lds[wave * 3 + 1] = md; lds[wave * 3 + 2] = sc; }
    __syncthreads();
    if (tid == 0) {
        P.pk_loc[0] = lds[0] + lds[3] + lds[6] + lds[9];
        P.pk_loc[1] = lds[2] + lds[5] + lds[8] + lds[11];
        P.pk_loc[2 + P.sh_rank] = fmax(fmax(lds[1], lds[4]), fmax(lds[7], lds[10]));
    }
    if (LIN) {
        double* q = P.pk_loc + 2 + P.sh_world;
        for (int i = tid; i < 27 * P.sh_nk; i += BLK) {
            const int k = P.sh_k0 + i / 27, cc = i % 27;
            q[27 * k + cc] = cc < 21 ? P.Hpp[21 * k + cc] : P.bp[6 * k + (cc - 21)];
        }
    }
}

template <bool LIN>
__global__ __launch_bounds__(BLK) void k_finalize_unpack(Dev P, int seq) {
    const int tid = threadIdx.x;
    if (LIN) {
        const double* q = P.pk + 2 + P.sh_world;
        for (int i = tid; i < 27 * P.K; i += BLK) {
            const int k = i / 27, cc = i % 27;
            if (cc < 21) P.Hpp[21 * k + cc] = q[i]; else P.bp[6 * k + (cc - 21)] = q[i];
        }
    }
    if (tid == 0) {
        P.scal[SC_CHI] = P.pk[0];
        P.h_scal[SC_CHI] = P.pk[0];
        if (LIN) {
            double md = 0;
            for (int r = 0; r < P.sh_world; ++r) md = fmax(md, P.pk[2 + r]);
            P.scal[SC_MAXDIAG] = md;
            P.h_scal[SC_MAXDIAG] = md;
        } else {
            P.scal[SC_SCALE] = P.pk[1];
            P.h_scal[SC_SCALE] = P.pk[1];
        }
        publish_flags(P, seq);
    }
}

// H_pp (21 packed) and b_p (6) of one pose per workgroup: fixed-order sums of the lineariser's partials
// (8 lanes per component, then the 8 in order); red[3 + k] = max |diagonal| for the LM lambda_0
__global__ __launch_bounds__(BLK) void k_pose_sums(Dev P) {
    __shared__ double lds[8][32];
    __shared__ double mdl[32];
    const int k = P.sh_k0 + blockIdx.x, tid = threadIdx.x, c = tid & 31, gl = tid >> 5;
    // plain windows (k_lin_plain): slots are per slice and also carry the regularisers' chi2 [28] and the rows' max |diagonal| [29]
    const int nsum = P.plain ? 29 : 27;
    double s = 0;
    if (c < nsum)
        for (int g = P.pose_grp_ptr[k] * P.lin_rb + gl; g < P.pose_grp_ptr[k + 1] * P.lin_rb; g += 8) s += P.part_lin[(size_t)g * 32 + c];
    if (P.plain && c == 29)
        for (int g = P.pose_grp_ptr[k] * P.lin_rb + gl; g < P.pose_grp_ptr[k + 1] * P.lin_rb; g += 8) s = fmax(s, P.part_lin[(size_t)g * 32 + c]);
    lds[gl][c] = s;
    __syncthreads();
    if (tid < 32) {
        double t = 0;
        if (c < nsum) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t += lds[q][c];
            if (c < 21) P.Hpp[k * 21 + c] = t; else if (c < 27) P.bp[k * 6 + (c - 21)] = t;
        }
        if (P.plain && c == 29) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t = fmax(t, lds[q][c]);
        }
        // diagonal entries of the packed upper triangle: 0,6,11,15,18,20
        mdl[c] = (c == 0 || c == 6 || c == 11 || c == 15 || c == 18 || c == 20 || (P.plain && c == 29)) ? fabs(t) : ((P.plain && (c == 27 || c == 28)) ? t : 0.0);
    }
    __syncthreads();
    if (tid == 0) {
        double m = 0;
        for (int q = 0; q < 21; ++q) m = fmax(m, mdl[q]);
        if (P.plain) { m = fmax(m, mdl[29]); P.part_pchi[k] = mdl[27] + mdl[28]; }
        (P.sh_on ? P.red_loc : P.red)[3 + k] = m;
    }
}

__global__ void k_publish(Dev P, int seq) {
    if (threadIdx.x == 0) publish_flags(P, seq);
}

}  // namespace nrs
