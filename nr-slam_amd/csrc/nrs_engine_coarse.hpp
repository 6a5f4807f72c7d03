// Two-level preconditioner of the fused path: coarse assembly, reduction, per-trial inverse.
// Part of nrs_engine.hip (one translation unit); see that file's header for the design.
#pragma once

namespace nrs {

// =====================================================================================
// Two-level preconditioner of the fused path (one pose, <= CO_MAX coarse unknowns).
//   k_coarse_tile   (per linearisation, one workgroup per tile): the tile's rows of Z^T H Z, i.e.
//                   sum of H_ij over i in the tile and j in each row group its incidences reach,
//                   its part of the landmark-pose coupling, sum of b, number of free rows;
//   k_coarse_reduce (one workgroup): C0 = Z^T H Z, N = Z^T Z, bc = Z^T b in fixed summation order;
//   k_coarse_invert (per trial, one workgroup): (C0 + lambda N)^-1 by in-place Gauss-Jordan in LDS.
// Coarse unknown 3g+c = translation c of every free row of group g; 3G+a = pose component a.
// =====================================================================================
template <int T>
__global__ __launch_bounds__(BLK) void k_coarse_tile(Dev P) {
    __shared__ double lds[4 * 9];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const int row0 = b * P.tile_rows;
    const int own_grp = row0 / ROW_ALIGN;
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb;
    double* lx = dyn;                                                      // positions, tile + halo
    unsigned char* lfix = reinterpret_cast<unsigned char*>(dyn + 3 * (size_t)(P.tile_rows + P.max_halo));
    unsigned short* lgrp = reinterpret_cast<unsigned short*>(lfix + P.tile_rows + P.max_halo + 8);
    stage_rows(P, b, tid, P.lin_xl, P.X0, lx);
    for (int i = tid; i < P.tile_rows + hn; i += BLK) {
        const int r = i < P.tile_rows ? row0 + i : P.halo_rows[hb + i - P.tile_rows];
        lfix[i] = (P.rflag[r] & RF_FIXED) ? 1 : 0;
        lgrp[i] = (unsigned short)(r / ROW_ALIGN);
    }
    __syncthreads();
    const int self = row - row0;
    const bool rfix = lfix[self] != 0;
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    // which groups do this tile's incidences reach?  (bit mask, order-independent OR)
    __shared__ unsigned int reach;
    if (tid == 0) reach = 1u << own_grp;
    __syncthreads();
    {
        unsigned int m = 0;
        for (int idx = sbeg + lane; idx < send; idx += 64) {
            const SpringRec rc = load_spring(P, idx);
            if (rc.other != REC_NONE && !lfix[rc.other]) m |= 1u << lgrp[rc.other];
        }
        for (int idx = dbeg + lane; idx < dend; idx += 64) {
            const DamperRec rc = load_damper(P, idx);
            if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
            const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (o[k] != REC_NONE && !lfix[o[k]]) m |= 1u << lgrp[o[k]];
        }
        if (m) atomicOr(&reach, m);
    }
    __syncthreads();
    const unsigned int reached = reach;
    for (int hg = 0; hg < P.n_groups; ++hg) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (!((reached >> hg) & 1u)) {                                 // uniform: nothing to sum
            if (tid < 6) P.co_ct[((size_t)b * P.n_groups + hg) * 6 + tid] = 0;
            continue;
        }
        {
            if (t == 0 && !rfix && hg == own_grp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k] = P.D[6 * (size_t)row + k];
            }
            for (int idx = sbeg + lane; idx < send; idx += 64) {
                const SpringRec rc = load_spring(P, idx);
                if (rc.other == REC_NONE || lfix[rc.other] || lgrp[rc.other] != hg) continue;
                const int o = rc.other;
                const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
                const double m = -rc.qc;                                   // H_ij = -qc v v^T
                acc[0] += m * v0 * v0; acc[1] += m * v0 * v1; acc[2] += m * v0 * v2;
                acc[3] += m * v1 * v1; acc[4] += m * v1 * v2; acc[5] += m * v2 * v2;
            }
            for (int idx = dbeg + lane; idx < dend; idx += 64) {
                const DamperRec rc = load_damper(P, idx);
                if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
                const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (o[k] == REC_NONE || lfix[o[k]] || lgrp[o[k]] != hg) continue;
                    const double c = (k == 2 ? 1.0 : -1.0) * rc.s;     // canonical order: a_i = s ((u_i - u[o1]) - (u[o0] - u[o2])) => H_ik = (-, -, +) s I
                    acc[0] += c; acc[3] += c; acc[5] += c;
                }
            }
        }
        block_sum_store<6>(acc, lds, tid, P.co_ct + ((size_t)b * P.n_groups + hg) * 6);
        __syncthreads();
    }
    {   // tile-level block: the same sum restricted to j inside the tile (second, finer level)
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (t == 0 && !rfix) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = P.D[6 * (size_t)row + k];
        }
        for (int idx = sbeg + lane; idx < send; idx += 64) {
            const SpringRec rc = load_spring(P, idx);
            if (rc.other == REC_NONE || rc.other >= P.tile_rows || lfix[rc.other]) continue;
            const int o = rc.other;
            const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
            const double m = -rc.qc;
            acc[0] += m * v0 * v0; acc[1] += m * v0 * v1; acc[2] += m * v0 * v2;
            acc[3] += m * v1 * v1; acc[4] += m * v1 * v2; acc[5] += m * v2 * v2;
        }
        for (int idx = dbeg + lane; idx < dend; idx += 64) {
            const DamperRec rc = load_damper(P, idx);
            if (rc.meta == REC_NONE || (rc.meta & DM_UNARY)) continue;
            const uint16_t o[3] = {rc.o0, rc.o1, rc.o2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (o[k] == REC_NONE || o[k] >= P.tile_rows || lfix[o[k]]) continue;
                const double c = (k == 2 ? 1.0 : -1.0) * rc.s;
                acc[0] += c; acc[3] += c; acc[5] += c;
            }
        }
        block_sum_store<6>(acc, lds, tid, P.co_bt + (size_t)b * 6);
        __syncthreads();
    }
    // landmark-pose coupling of the tile's rows: sum of H_lp = J_l^T w J_p (3x6), two halves of 9
    double cp[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) cp[k] = 0;
    double tb[4] = {0, 0, 0, 0};
    if (t == 0 && !rfix) {
        const RowRec rc = P.rowrec[row];
        if (rc.w != 0.0 && !P.pose_fixed[0]) {
            const Pose Tcw = P.lin_pose[0];
            double Rm[9];
            quat_to_R(Tcw.q, Rm);
            const double px = Rm[0] * xs[0] + Rm[1] * xs[1] + Rm[2] * xs[2] + Tcw.t[0];
            const double py = Rm[3] * xs[0] + Rm[4] * xs[1] + Rm[5] * xs[2] + Tcw.t[1];
            const double pz = Rm[6] * xs[0] + Rm[7] * xs[1] + Rm[8] * xs[2] + Tcw.t[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
                const double Jp[6] = {-j1 * pz + j2 * py, j0 * pz - j2 * px, -j0 * py + j1 * px, j0, j1, j2};
                const double Jl[3] = {j0 * Rm[0] + j1 * Rm[3] + j2 * Rm[6], j0 * Rm[1] + j1 * Rm[4] + j2 * Rm[7],
                                      j0 * Rm[2] + j1 * Rm[5] + j2 * Rm[8]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int a = 0; a < 6; ++a) cp[c * 6 + a] += rc.w * Jl[c] * Jp[a];
            }
        }
        tb[0] = P.bl[3 * row]; tb[1] = P.bl[3 * row + 1]; tb[2] = P.bl[3 * row + 2];
        tb[3] = 1.0;
    }
    block_sum_store<9>(cp, lds, tid, P.co_cp + (size_t)b * 18);
    __syncthreads();
    block_sum_store<9>(cp + 9, lds, tid, P.co_cp + (size_t)b * 18 + 9);
    __syncthreads();
    block_sum_store<4>(tb, lds, tid, P.co_tb + (size_t)b * 4);
}

__global__ __launch_bounds__(BLK) void k_coarse_reduce(Dev P) {
    const int tid = threadIdx.x;
    const int G = P.n_groups, n = P.co_n, rb = ROW_ALIGN / P.tile_rows;
    for (int i = tid; i < n * n; i += BLK) P.co_c0[i] = 0;
    __syncthreads();
    // group-group blocks: thread per (g, h), fixed order over the group's tiles and their slots
    for (int gh = tid; gh < G * G; gh += BLK) {
        const int g = gh / G, h = gh % G;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int tl = g * rb; tl < (g + 1) * rb; ++tl)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += P.co_ct[((size_t)tl * G + h) * 6 + k];
        const double m[9] = {acc[0], acc[1], acc[2], acc[1], acc[3], acc[4], acc[2], acc[4], acc[5]};
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) P.co_c0[(size_t)(3 * g + a) * n + 3 * h + c] = m[a * 3 + c];
    }
    // group sums of the coupling, of b and of the free-row counts
    for (int g = tid; g < G; g += BLK) {
        double cp[18], tb[4] = {0, 0, 0, 0};
        for (int k = 0; k < 18; ++k) cp[k] = 0;
        for (int tl = g * rb; tl < (g + 1) * rb; ++tl) {
            for (int k = 0; k < 18; ++k) cp[k] += P.co_cp[(size_t)tl * 18 + k];
            for (int k = 0; k < 4; ++k) tb[k] += P.co_tb[(size_t)tl * 4 + k];
        }
        for (int c = 0; c < 3; ++c) {
            for (int a = 0; a < 6; ++a) {
                P.co_c0[(size_t)(3 * g + c) * n + 3 * G + a] = cp[c * 6 + a];
                P.co_c0[(size_t)(3 * G + a) * n + 3 * g + c] = cp[c * 6 + a];
            }
            P.co_nn[3 * g + c] = tb[3];
            P.co_bc[3 * g + c] = tb[c];
        }
    }
    // pose block
    if (tid < 36) {
        const int a = tid / 6, c = tid % 6;
        const int lo = a < c ? a : c, hi = a < c ? c : a;
        const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
        const bool pfix = P.pose_fixed[0] != 0;
        P.co_c0[(size_t)(3 * G + a) * n + 3 * G + c] = pfix ? (a == c ? 1.0 : 0.0) : P.Hpp[pk];
        if (c == 0) { P.co_nn[3 * G + a] = pfix ? 0.0 : 1.0; P.co_bc[3 * G + a] = pfix ? 0.0 : P.bp[a]; }
    }
}

// 6x6 in-place inverse in registers (Gauss-Jordan without pivoting, every index static); false if a pivot is not positive
__device__ inline bool inv6_inplace(double (&m)[6][6]) {
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const double piv = m[p][p];
        ok = ok && piv > 0 && isfinite(piv);
        const double pinv = 1.0 / piv;
        double col[6], row[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { col[i] = m[i][p]; row[i] = m[p][i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                m[i][j] = (i == p) ? ((j == p) ? pinv : row[j] * pinv) : ((j == p) ? -col[i] * pinv : m[i][j] - col[i] * row[j] * pinv);
    }
    return ok;
}

__global__ __launch_bounds__(BLK) void k_coarse_invert(Dev P, double lam) {
    // In-place BLOCK Gauss-Jordan (SPD: no pivoting) with the matrix in REGISTERS: thread (br, bc) keeps a
    // 6x6 block.  Per block step the pivot block is inverted in its owner's registers, the pivot block row
    // is scaled by it, and every other block takes one 6x6x6 update; only the pivot block, row and column go
    // through LDS (double-buffered by step parity: two barriers per step, n/6 steps instead of n pivots).
    // The matrix is padded to a multiple of 6 with an identity block.  A non-positive pivot switches the
    // coarse level off for this trial.
    constexpr int BS = 6, NBMAX = (CO_MAX + BS - 1) / BS;
    __shared__ double colb[2][NBMAX][BS * BS], rowb[2][NBMAX][BS * BS], pivb[2][BS * BS];
    __shared__ int bad;
    const int tid = threadIdx.x, n = P.co_n;
    const int nb = (n + BS - 1) / BS;
    const bool act = tid < nb * nb;
    const int br = act ? tid / nb : 0, bc = act ? tid % nb : 0;
    if (tid == 0) bad = 0;
    for (int tl = tid; tl < P.n_regblk; tl += BLK) {                       // tile-level 3x3 blocks of this trial
        double Bi[6];
        const double nf = P.co_tb[4 * (size_t)tl + 3];
        const bool okb = nf > 0 && inv3_sym(P.co_bt + 6 * (size_t)tl, lam * nf, Bi) && Bi[0] > 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) P.co_bti[6 * (size_t)tl + q] = okb ? Bi[q] : 0.0;
    }
    double a[BS][BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
            const int r = br * BS + i, c = bc * BS + j;
            double v = r == c ? 1.0 : 0.0;
            if (act && r < n && c < n) {
                v = P.co_c0[(size_t)r * n + c];
                if (r == c) {
                    v += lam * P.co_nn[r];
                    if (P.co_nn[r] == 0.0 && v == 0.0) v = 1.0;            // empty group: keep the system regular
                }
            }
            a[i][j] = v;
        }
    __syncthreads();
    for (int pb = 0; pb < nb; ++pb) {
        const int buf = pb & 1;
        // phase A: the pivot block's owner inverts it; the pivot block column is published as it is
        if (act && bc == pb) {
            if (br == pb) {
                if (!inv6_inplace(a)) bad = 1;
#pragma unroll
                for (int i = 0; i < BS; ++i)
#pragma unroll
                    for (int j = 0; j < BS; ++j) pivb[buf][i * BS + j] = a[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < BS; ++i)
#pragma unroll
                    for (int j = 0; j < BS; ++j) colb[buf][br][i * BS + j] = a[i][j];
            }
        }
        __syncthreads();
        if (bad) break;                                                    // (uniform: read after the barrier)
        // phase B: pivot block row <- P^-1 * row, published; pivot block column <- -column * P^-1
        if (act && br == pb && bc != pb) {                                  // row: R = P^-1 * A_pj
            double t[BS][BS];
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    double sacc = 0;
#pragma unroll
                    for (int k = 0; k < BS; ++k) sacc += pivb[buf][i * BS + k] * a[k][j];
                    t[i][j] = sacc;
                }
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) { a[i][j] = t[i][j]; rowb[buf][bc][i * BS + j] = t[i][j]; }
        } else if (act && bc == pb && br != pb) {                           // column: -A_ip * P^-1
            double t[BS][BS];
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    double sacc = 0;
#pragma unroll
                    for (int k = 0; k < BS; ++k) sacc += a[i][k] * pivb[buf][k * BS + j];
                    t[i][j] = -sacc;
                }
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) a[i][j] = t[i][j];
        }
        __syncthreads();
        // phase C: every other block takes A_ij -= C_i * R_j
        if (act && br != pb && bc != pb) {
#pragma unroll
            for (int i = 0; i < BS; ++i)
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    double sacc = 0;
#pragma unroll
                    for (int k = 0; k < BS; ++k) sacc += colb[buf][br][i * BS + k] * rowb[buf][bc][k * BS + j];
                    a[i][j] -= sacc;
                }
        }
    }
    __syncthreads();
    const bool off = bad != 0;
    // result -> global; y0 = A^-1 (Z^T b): every thread multiplies its block with its slice of the vector, the
    // nb partial sums of a row are added in a fixed order
    __shared__ double ypart[NBMAX * NBMAX][BS];
    if (act) {
        double bcv[BS];
#pragma unroll
        for (int j = 0; j < BS; ++j) bcv[j] = bc * BS + j < n ? P.co_bc[bc * BS + j] : 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            double y = 0;
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                const int r = br * BS + i, c = bc * BS + j;
                const double v = off ? 0.0 : a[i][j];
                if (r < n && c < n) { P.co_inv[(size_t)r * n + c] = v; y += v * bcv[j]; }
            }
            ypart[br * nb + bc][i] = y;
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += BLK) {
        double y = 0;
        for (int q = 0; q < nb; ++q) y += ypart[(j / BS) * nb + q][j % BS];
        P.co_y0[j] = y;
    }
}

}  // namespace nrs
