// PCG for small single-pose problems (a2 on frames of up to 1024 rows -- the reference's own scale, C1) inside ONE
// workgroup: a whole batch of iterations per launch, no launch boundary and no cross-workgroup hand-off between them.
//
// k_pcg_fused spends one launch per iteration, and a launch is ~15 us of dependent memory round trips plus ~4 us to the
// next one however small the frame is (543 points: 3069 iterations x 19.3 us = 59 ms, slower than the 1013-point frame).
// Here the preconditioned residual u, the positions of the linearisation point and the rows' reprojection factors live in
// LDS for ALL rows, the other PCG vectors (r, s, w, p, x) and the row's block-Jacobi inverse in the registers of the
// thread that owns the row (thread i <-> row i), the pose part in LDS, and an iteration is six workgroup barriers:
//     operator on the staged u (sixteen waves, slices of 8 rows x 8 lanes, records streamed from L2, two slices in flight)
//     -> one reduction of the nine partials (r.u, w.u, u_l.(H_pl^T u_p), H_pl u_l) -> scalars
//     -> vector update in registers -> tile sums -> coarse / tile-level corrections -> u back to LDS.
// The arithmetic is k_pcg_fused's (Chronopoulos-Gear recurrences, M^-1 = block-Jacobi + tile level + coarse level, the
// same convergence and milestone flags); the order of the sums differs, so iteration counts can differ by one or two.
// Between launches the state sits in the arrays k_trial_setup / k_apply use (half 0 of the ping-pong pairs).
#pragma once
#include "nrs_engine_pcg.hpp"

namespace nrs {

constexpr int WG1_THREADS = 1024;
constexpr int WG1_MAX_ROWS = 1024;
constexpr int WG1_MAX_CN = 3 * (WG1_MAX_ROWS / ROW_ALIGN) + 6;   // coarse unknowns of the largest frame it takes

// sums of N values over the 16 waves of the workgroup, every thread gets the totals (fixed order); two barriers
template <int N>
__device__ inline void wg1_sum(double* v, double* scratch /* 16 N */, int lane, int wave) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) scratch[wave * N + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += scratch[w * N + k];
        v[k] = t;
    }
    __syncthreads();
}

// dynamic LDS the kernel needs (host side of the same layout)
inline size_t wg1_shm_bytes(int n_rows, int n_tiles, int n_halo) {
    return sizeof(double) * (9 * (size_t)n_rows + 6 * (size_t)n_tiles) + sizeof(RowRec) * (size_t)n_rows + sizeof(int) * (4 * (size_t)(n_rows / 8) + n_tiles + 2) +
           sizeof(uint16_t) * ((size_t)n_halo + 8) + (size_t)n_rows + 64;
}

__global__ __launch_bounds__(WG1_THREADS) void k_pcg_wg(Dev P, double lam, int it0, int count, double tol2, double peek_tol2, int pub_seq) {
    extern __shared__ double dyn[];
    __shared__ double s_red[16 * 9];
    __shared__ double s_pose[6][6];                                // rows: r_p s_p p_p x_p u_p w_p
    __shared__ double s_Hpp[21], s_Hi[36];
    __shared__ double s_ycor[CO_MAX];
    __shared__ double s_cinv[WG1_MAX_CN * WG1_MAX_CN];             // A_c^-1 of this trial (<= 4 row groups + the pose)
    enum { RP = 0, SP = 1, PP = 2, XP = 3, UP = 4, WP = 5 };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_rows = P.n_rows, n_tiles = P.n_regblk, n_slices = n_rows / 8;
    const bool CO = P.coarse != 0;
    double* lu = dyn;                                              // u of every row
    double* lx = lu + 3 * (size_t)n_rows;                          // positions of the linearisation point
    double* lw = lx + 3 * (size_t)n_rows;                          // operator result, handed from the slice lanes to the row owners
    double* c_ts = lw + 3 * (size_t)n_rows;                        // n_tiles x 3: tile sums of the new residual
    double* c_yt = c_ts + 3 * (size_t)n_tiles;                     // n_tiles x 3: tile-level corrections
    RowRec* l_rc = reinterpret_cast<RowRec*>(c_yt + 3 * (size_t)n_tiles);   // reprojection factors of every row (16-byte aligned: the sizes above are even)
    int* l_ptr = reinterpret_cast<int*>(l_rc + n_rows);            // per slice {sbeg, send, dbeg, dend} (empty for fixed rows)
    int* l_hptr = l_ptr + 4 * (size_t)n_slices;                    // n_tiles + 1
    uint16_t* l_halo = reinterpret_cast<uint16_t*>(l_hptr + n_tiles + 2);   // every tile's halo list, as rows
    const int n_halo = P.halo_ptr[n_tiles];
    uint8_t* l_fix = reinterpret_cast<uint8_t*>(l_halo + ((n_halo + 7) & ~7));   // fixed rows carry no incidences

    // ---- entry: static data and the state of the previous launch (or of k_trial_setup)
    const bool own = tid < n_rows;
    const size_t orow = own ? (size_t)tid : 0;
    const bool o_free = own && !(P.rflag[orow] & RF_FIXED);
    double r[3] = {0, 0, 0}, s[3] = {0, 0, 0}, w[3] = {0, 0, 0}, p[3] = {0, 0, 0}, x[3] = {0, 0, 0}, u[3] = {0, 0, 0}, Di[6] = {0, 0, 0, 0, 0, 0};
    if (own) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r[k] = P.rv[3 * orow + k]; u[k] = P.uv3[3 * orow + k]; p[k] = P.pv[3 * orow + k]; s[k] = P.sv[3 * orow + k]; x[k] = P.xv[3 * orow + k];
            if (it0 > 0) w[k] = P.wv[3 * orow + k];
            lu[3 * orow + k] = u[k];
            lx[3 * orow + k] = P.lin_xl[3 * orow + k] + (P.X0 ? P.X0[3 * orow + k] : 0.0);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) Di[k] = P.Dinv[6 * orow + k];
        l_rc[orow] = P.rowrec[orow];
        l_fix[orow] = (P.rflag[orow] & RF_FIXED) ? 1 : 0;
    }
    for (int sl = tid; sl < n_slices; sl += WG1_THREADS) {
        // (a slice of 8 rows: fixed rows carry no incidences in k_pcg_fused either -- their lists are read as empty)
        const int sb = P.ss_ptr[sl], se = P.ss_ptr[sl + 1], db = P.sd_ptr[sl], de = P.sd_ptr[sl + 1];
        l_ptr[4 * sl] = sb; l_ptr[4 * sl + 1] = se; l_ptr[4 * sl + 2] = db; l_ptr[4 * sl + 3] = de;
    }
    for (int i = tid; i <= n_tiles; i += WG1_THREADS) l_hptr[i] = P.halo_ptr[i];
    for (int i = tid; i < n_halo; i += WG1_THREADS) l_halo[i] = (uint16_t)P.halo_rows[i];
    double bt_q[6] = {0, 0, 0, 0, 0, 0};                           // (B_t + lambda n_t)^-1 of tile `tid`
    if (CO && tid < n_tiles) {
#pragma unroll
        for (int q = 0; q < 6; ++q) bt_q[q] = P.co_bti[6 * (size_t)tid + q];
    }
    if (tid < 6) {
        s_pose[RP][tid] = P.rp[tid]; s_pose[SP][tid] = P.sp[tid]; s_pose[PP][tid] = P.pp[tid]; s_pose[XP][tid] = P.xp[tid]; s_pose[UP][tid] = P.up[tid];
        s_pose[WP][tid] = it0 > 0 ? P.wp[tid] : 0.0;
    }
    if (tid < 21) s_Hpp[tid] = P.Hpp[tid];
    if (tid < 36) s_Hi[tid] = P.Hppinv[tid];
    if (CO) for (int i = tid; i < P.co_n * P.co_n; i += WG1_THREADS) s_cinv[i] = P.co_inv[i];
    const Pose Tlin = P.lin_pose[0];
    const double pmask = P.pose_fixed[0] ? 0.0 : 1.0;
    const int cG = P.n_groups, cn = P.co_n;
    double gamma_prev = P.scal[SC_SLOT0], alpha_prev = P.scal[SC_SLOT0 + 1], gamma0 = P.scal[SC_GAMMA0];
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // totals of the previous operator pass (pose shares included in [0], [1])
    if (it0 > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) part[k] = P.part_spmv[k];
    }
    const bool already_done = P.flags[0] != 0;
    __syncthreads();
    int it = it0;
    for (; it < it0 + count && !already_done; ++it) {
        if (it > 0) {
            // ---- scalars of iteration it-1 from the totals of the previous pass
            const int ip = it - 1;
            const double gamma = part[0], delta = part[1];
            if (ip == 0) gamma0 = gamma;
            const bool bad = !isfinite(gamma) || !isfinite(delta);
            if ((gamma <= tol2 * gamma0) || bad || gamma == 0.0) {
                if (tid == 0) {
                    if (bad) P.flags[2] = 1;
                    P.flags[1] = ip;
                    P.flags[0] = 1;
                }
                break;
            }
            const double beta = ip == 0 ? 0.0 : gamma / gamma_prev;
            const double alpha = ip == 0 ? gamma / delta : gamma / (delta - beta * gamma / alpha_prev);
            gamma_prev = gamma; alpha_prev = alpha;
            if (tid == 0) {
                P.flags[1] = ip + 1;
                if (gamma <= peek_tol2 * gamma0) {
                    P.flags[3] = max(P.flags[3], (gamma <= 1e-6 * peek_tol2 * gamma0 ? 4 : gamma <= 1e-4 * peek_tol2 * gamma0 ? 3 : gamma <= 1e-2 * peek_tol2 * gamma0 ? 2 : 1));
                    if (P.flags[4] == 0) P.flags[4] = ip + 1;
                }
            }
            // ---- vector update in registers; tile sums of the new residual
            double ts[3] = {0, 0, 0};
            if (own) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = u[k] + beta * p[k];
                    s[k] = w[k] + beta * s[k];
                    x[k] += alpha * p[k];
                    r[k] -= alpha * s[k];
                    ts[k] = r[k];
                }
            }
            if (tid < 6) {
                const double pp = s_pose[UP][tid] + beta * s_pose[PP][tid];
                const double sp = s_pose[WP][tid] + beta * s_pose[SP][tid];
                s_pose[PP][tid] = pp; s_pose[SP][tid] = sp;
                s_pose[XP][tid] += alpha * pp;
                s_pose[RP][tid] -= alpha * sp;
            }
            if (CO) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) ts[k] += __shfl_xor(ts[k], off, 64);   // (32 rows = half a wave)
                }
                if ((lane & 31) == 0 && (tid >> 5) < n_tiles) {
                    c_ts[3 * (tid >> 5)] = ts[0]; c_ts[3 * (tid >> 5) + 1] = ts[1]; c_ts[3 * (tid >> 5) + 2] = ts[2];
                }
            }
            __syncthreads();
            if (CO) {
                // coarse level: y = A_c^-1 [group sums of r ; r_p]; tile level: y_t = B_t^-1 (tile sum)
                if (tid < cn) {
                    const int rb = ROW_ALIGN / P.tile_rows;
                    double y = 0;
                    for (int c = 0; c < cn; ++c) {
                        double z;
                        if (c < 3 * cG) {
                            const int g = c / 3, k = c % 3;
                            z = 0;
                            for (int j = 0; j < rb; ++j) z += c_ts[3 * (g * rb + j) + k];
                        } else z = s_pose[RP][c - 3 * cG];
                        y += s_cinv[c * cn + tid] * z;
                    }
                    s_ycor[tid] = y;
                }
                if (tid < n_tiles) {
                    double rc3[3] = {c_ts[3 * tid], c_ts[3 * tid + 1], c_ts[3 * tid + 2]}, yt[3];
                    tile_level(bt_q, rc3, yt);
                    c_yt[3 * tid] = yt[0]; c_yt[3 * tid + 1] = yt[1]; c_yt[3 * tid + 2] = yt[2];
                }
                __syncthreads();
            }
            // ---- u = M^-1 r
            if (own) {
                u[0] = Di[0] * r[0] + Di[1] * r[1] + Di[2] * r[2];
                u[1] = Di[1] * r[0] + Di[3] * r[1] + Di[4] * r[2];
                u[2] = Di[2] * r[0] + Di[4] * r[1] + Di[5] * r[2];
                if (CO && o_free) {
                    const int g = tid / ROW_ALIGN, tl = tid / P.tile_rows;
                    u[0] += s_ycor[3 * g] + c_yt[3 * tl]; u[1] += s_ycor[3 * g + 1] + c_yt[3 * tl + 1]; u[2] += s_ycor[3 * g + 2] + c_yt[3 * tl + 2];
                }
                lu[3 * orow] = u[0]; lu[3 * orow + 1] = u[1]; lu[3 * orow + 2] = u[2];
            }
            if (tid < 6) {
                double unew = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) unew += s_Hi[tid * 6 + c] * s_pose[RP][c];
                if (CO && pmask != 0.0) unew += s_ycor[3 * cG + tid];
                s_pose[UP][tid] = unew;
            }
            __syncthreads();
        }
        // ---- operator on the staged u: w = (H + lambda I) u, slice by slice (8 rows x 8 lanes per wave pass)
        double acc[7] = {0, 0, 0, 0, 0, 0, 0};                     // [0] u_l.(H_pl^T u_p), [1..6] H_pl u_l of this lane's rows
        double up6[6], rp6[6];                                     // (snapshots: the next update rewrites them without a barrier in between)
#pragma unroll
        for (int q = 0; q < 6; ++q) { up6[q] = s_pose[UP][q]; rp6[q] = s_pose[RP][q]; }
        for (int sl = wave; sl < n_slices; sl += 16) {
            const int row = sl * 8 + (lane >> 3), t = lane & 7;
            const int b = row / P.tile_rows, row0 = b * P.tile_rows, hb = l_hptr[b];
            const bool rfix = l_fix[row] != 0;
            const int sbeg = l_ptr[4 * sl], send = rfix ? sbeg : l_ptr[4 * sl + 1];
            const int dbeg = l_ptr[4 * sl + 2], dend = rfix ? dbeg : l_ptr[4 * sl + 3];
            auto grow = [&](int o) { return o < P.tile_rows ? row0 + o : (int)l_halo[hb + o - P.tile_rows]; };
            const double ul[3] = {lu[3 * row], lu[3 * row + 1], lu[3 * row + 2]};
            const double xs[3] = {lx[3 * row], lx[3 * row + 1], lx[3 * row + 2]};
            double a0 = 0, a1 = 0, a2 = 0;
            if (t == 0) {
                a0 = lam * ul[0]; a1 = lam * ul[1]; a2 = lam * ul[2];
                const RowRec rc = l_rc[row];
                if (rc.w != 0.0) {
                    double pr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    row_factored(rc, Tlin, xs, ul, up6, pmask, a0, a1, a2, pr);
#pragma unroll
                    for (int k = 0; k < 7; ++k) acc[k] += pr[2 + k];
                }
            }
            for (int idx = sbeg + lane; idx < send; idx += 64) {
                const SpringRec sr = load_spring(P, idx);
                if (sr.other == REC_NONE) continue;
                const int o = grow(sr.other);
                const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
                const double dot = sr.qc * (v0 * (ul[0] - lu[3 * o]) + v1 * (ul[1] - lu[3 * o + 1]) + v2 * (ul[2] - lu[3 * o + 2]));
                a0 += dot * v0; a1 += dot * v1; a2 += dot * v2;
            }
            for (int idx = dbeg + lane; idx < dend; idx += 64) {
                const DamperRec dr = load_damper(P, idx);
                if (dr.meta == REC_NONE) continue;
                if (dr.meta & DM_UNARY) {
                    a0 += dr.s * ul[0]; a1 += dr.s * ul[1]; a2 += dr.s * ul[2];
                    continue;
                }
                // canonical order of the others: a_i += s ((u_i - u[o1]) - (u[o0] - u[o2])), absent vertices are zeros
                const uint16_t o3[3] = {dr.o0, dr.o1, dr.o2};
                double v[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool has = o3[k] != REC_NONE;
                    const int o = has ? grow(o3[k]) : 0;
                    v[k][0] = has ? lu[3 * o] : 0.0; v[k][1] = has ? lu[3 * o + 1] : 0.0; v[k][2] = has ? lu[3 * o + 2] : 0.0;
                }
                a0 += dr.s * ((ul[0] - v[1][0]) - (v[0][0] - v[2][0]));
                a1 += dr.s * ((ul[1] - v[1][1]) - (v[0][1] - v[2][1]));
                a2 += dr.s * ((ul[2] - v[1][2]) - (v[0][2] - v[2][2]));
            }
            a0 = sub_sum_t<8>(a0); a1 = sub_sum_t<8>(a1); a2 = sub_sum_t<8>(a2);
            if (t == 0) { lw[3 * row] = a0; lw[3 * row + 1] = a1; lw[3 * row + 2] = a2; }
        }
        __syncthreads();
        // ---- the nine totals: r.u, w.u over the rows, the pose coupling sums; then the pose rows' share
        if (own) {
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] = lw[3 * orow + k];
        }
        part[0] = own ? r[0] * u[0] + r[1] * u[1] + r[2] * u[2] : 0.0;
        part[1] = own ? w[0] * u[0] + w[1] * u[1] + w[2] * u[2] : 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) part[2 + k] = acc[k];
        wg1_sum<9>(part, s_red, lane, wave);
        // pose rows: w_p = (H_pp + lambda) u_p + sum_l H_pl u_l ; every thread forms the same two pose sums in the same order
        {
            double g_p = 0, d_p = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double wa = lam * up6[a] + part[3 + a];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int lo = a < c ? a : c, hi = a < c ? c : a;
                    wa += s_Hpp[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)] * up6[c];
                }
                if (tid == a) s_pose[WP][a] = wa;                  // (read again only behind the next barrier)
                d_p += wa * up6[a];
                g_p += rp6[a] * up6[a];
            }
            part[0] += g_p;
            part[1] += d_p;
        }
    }
    __syncthreads();
    // ---- exit: the state goes back to the arrays the next launch (or k_apply) reads
    if (own) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            P.rv[3 * orow + k] = r[k]; P.uv3[3 * orow + k] = u[k]; P.pv[3 * orow + k] = p[k];
            P.sv[3 * orow + k] = s[k]; P.xv[3 * orow + k] = x[k]; P.wv[3 * orow + k] = w[k];
        }
    }
    if (tid < 6) {
        P.rp[tid] = s_pose[RP][tid]; P.sp[tid] = s_pose[SP][tid]; P.pp[tid] = s_pose[PP][tid]; P.xp[tid] = s_pose[XP][tid];
        P.up[tid] = s_pose[UP][tid]; P.wp[tid] = s_pose[WP][tid];
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) P.part_spmv[k] = part[k];
        P.scal[SC_SLOT0] = gamma_prev; P.scal[SC_SLOT0 + 1] = alpha_prev; P.scal[SC_GAMMA0] = gamma0;
        __threadfence();
        if (pub_seq != 0) publish_flags(P, pub_seq);
    }
}

}  // namespace nrs
