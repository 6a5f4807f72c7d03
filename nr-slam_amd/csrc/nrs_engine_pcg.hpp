// PCG kernels: trial setup, operator (factored, LDS-staged / gather fallback), vector update, fused single-launch iteration, state update, edge taps.
// Part of nrs_engine.hip (one translation unit); see that file's header for the design.
#pragma once

namespace nrs {

// =====================================================================================
// per-trial setup: block-Jacobi preconditioner for (H + lambda I) and the PCG start vectors
//   x = 0, r = b, u = M^-1 r, p = s = 0.
// =====================================================================================
__device__ inline bool inv6_spd(const double* Hu, double lam, double* Ainv /*36*/) {
    double L[6][6];
    double A[6][6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; ++k; }
    for (int i = 0; i < 6; ++i) A[i][i] += lam;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[i][j] = 0;
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
        for (int q = 0; q < j; ++q) d -= L[j][q] * L[j][q];
        if (!(d > 0)) { ok = false; d = 1; }
        const double l = sqrt(d);
        L[j][j] = l;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i][j];
            for (int q = 0; q < j; ++q) s -= L[i][q] * L[j][q];
            L[i][j] = s / l;
        }
    }
    for (int c = 0; c < 6; ++c) {                       // solve A X = e_c
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int q = 0; q < i; ++q) s -= L[i][q] * y[q];
            y[i] = s / L[i][i];
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int q = i + 1; q < 6; ++q) s -= L[q][i] * x[q];
            x[i] = s / L[i][i];
        }
        for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = x[i];
    }
    return ok;
}

__global__ __launch_bounds__(BLK) void k_trial_setup(Dev P, double lam) {
    __shared__ double lds_ru[4];
    const int i = (P.sh_vb0 + blockIdx.x) * BLK + threadIdx.x;      // row (own range); poses below: every rank, all of them
    double ru[1] = {0};
    if (i < P.n_rows) {
        double Di[6];
        const bool ok = inv3_sym(P.D + 6 * (size_t)i, lam, Di);
        if (!ok || !isfinite(Di[0])) P.flags[2] = 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) P.Dinv[6 * (size_t)i + k] = Di[k];
        const double r0 = P.bl[3 * i], r1 = P.bl[3 * i + 1], r2 = P.bl[3 * i + 2];
        P.rv[3 * i] = r0; P.rv[3 * i + 1] = r1; P.rv[3 * i + 2] = r2;
        double y0 = 0, y1 = 0, y2 = 0;
        if (P.coarse && !(P.rflag[i] & RF_FIXED)) {
            const int g = i / ROW_ALIGN, tl = i / P.tile_rows;
            double yt[3];
            tile_level(P.co_bti + 6 * (size_t)tl, P.co_tb + 4 * (size_t)tl, yt);
            y0 = P.co_y0[3 * g] + yt[0]; y1 = P.co_y0[3 * g + 1] + yt[1]; y2 = P.co_y0[3 * g + 2] + yt[2];
        }
        const double u0 = Di[0] * r0 + Di[1] * r1 + Di[2] * r2 + y0;
        const double u1 = Di[1] * r0 + Di[3] * r1 + Di[4] * r2 + y1;
        const double u2 = Di[2] * r0 + Di[4] * r1 + Di[5] * r2 + y2;
        P.uv3[3 * i] = u0; P.uv3[3 * i + 1] = u1; P.uv3[3 * i + 2] = u2;
        ru[0] = r0 * u0 + r1 * u1 + r2 * u2;
#pragma unroll
        for (int k = 0; k < 3; ++k) { P.xv[3 * i + k] = 0; P.pv[3 * i + k] = 0; P.sv[3 * i + k] = 0; }
    }
    if (blockIdx.x * BLK + threadIdx.x < P.K) {
        const int i = blockIdx.x * BLK + threadIdx.x;
        double Ai[36];
        if (!inv6_spd(P.Hpp + 21 * i, lam, Ai)) P.flags[2] = 1;
        for (int k = 0; k < 36; ++k) P.Hppinv[36 * i + k] = Ai[k];
        for (int a = 0; a < 6; ++a) {
            double s = 0;
            for (int c = 0; c < 6; ++c) s += Ai[a * 6 + c] * P.bp[6 * i + c];
            if (P.coarse && !P.pose_fixed[i]) s += P.co_y0[3 * P.n_groups + a];
            P.up[6 * i + a] = s;
            P.rp[6 * i + a] = P.bp[6 * i + a];
            P.xp[6 * i + a] = 0; P.pp[6 * i + a] = 0; P.sp[6 * i + a] = 0;
        }
    }
    if (P.ecd) {                                                   // r.u of iteration 0, half 0 of the pair
        block_sum<1>(ru, lds_ru, threadIdx.x & 63, threadIdx.x >> 6);
        if (threadIdx.x == 0) P.part_ru[P.sh_vb0 + blockIdx.x] = ru[0];
    }
}

// =====================================================================================
// PCG kernel 1: w = (H + lambda I) u for the landmark rows, from the per-incidence factors,
// plus the per-block partials the update kernel needs:
//   [0] r.u  [1] w.u  [2] u_l.(H_pl^T u_p)  [3..8] H_pl u_l (pose rows)
// =====================================================================================
template <int T, bool LDS>
__global__ __launch_bounds__(BLK) void k_spmv(Dev P, double lam, int it) {
    static_assert(!LDS, "gather fallback only: the LDS-staged path is k_spmv_f");
    __shared__ double lds[4 * 9];
    constexpr int R = 64 / T;
    const int b = xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    const double* u = P.uv3;
    const int self = row;
    double a0 = 0, a1 = 0, a2 = 0;
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double ul0 = u[3 * self], ul1 = u[3 * self + 1], ul2 = u[3 * self + 2];
    {
        // row part: the T lanes of a row share the 6 pose components of H_pl
        const int kf = P.grp_pose[row / ROW_ALIGN];
        if (t == 0) {
            const double* D = (P.D_op ? P.D_op : P.D) + 6 * (size_t)row;
            a0 = (D[0] + lam) * ul0 + D[1] * ul1 + D[2] * ul2;
            a1 = D[1] * ul0 + (D[3] + lam) * ul1 + D[4] * ul2;
            a2 = D[2] * ul0 + D[4] * ul1 + (D[5] + lam) * ul2;
        }
        double h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
        for (int p = t; p < 6; p += T) {
            const double upk = ((it & 1) ? P.up2 : P.up)[6 * kf + p];
            const double e0 = P.Hpl[(size_t)(p * 3) * P.n_rows + row];
            const double e1 = P.Hpl[(size_t)(p * 3 + 1) * P.n_rows + row];
            const double e2 = P.Hpl[(size_t)(p * 3 + 2) * P.n_rows + row];
            h0 += e0 * upk; h1 += e1 * upk; h2 += e2 * upk;
            part[3 + p] = e0 * ul0 + e1 * ul1 + e2 * ul2;
        }
        a0 += h0; a1 += h1; a2 += h2;
        part[2] = ul0 * h0 + ul1 * h1 + ul2 * h2;
    }
    if (!rfix) {
        const int beg = P.ss_ptr[slice], end = P.ss_ptr[slice + 1];
        const size_t nz = (size_t)P.ss_nnz;
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int o = P.s_other[idx];
            if (o < 0) continue;
            const double g0 = P.s_g[idx], g1 = P.s_g[nz + idx], g2 = P.s_g[2 * nz + idx];
            const double dot = g0 * u[3 * o] + g1 * u[3 * o + 1] + g2 * u[3 * o + 2];
            a0 -= g0 * dot; a1 -= g1 * dot; a2 -= g2 * dot;
        }
    }
    if (!rfix) {
        const int beg = P.sd_ptr[slice], end = P.sd_ptr[slice + 1];
        for (int idx = beg + lane; idx < end; idx += 64) {
            const int meta = P.d_meta[idx];
            if (meta < 0 || (meta & DM_UNARY)) continue;
            const int o[3] = {P.d_o0[idx], P.d_o1[idx], P.d_o2[idx]};
            const int role = meta & 3;
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double sg = damper_sign(k + (k >= role ? 1 : 0));
                if (o[k] >= 0) { s0 += sg * u[3 * o[k]]; s1 += sg * u[3 * o[k] + 1]; s2 += sg * u[3 * o[k] + 2]; }
            }
            const double c = damper_sign(role) * P.d_s[idx];
            a0 += c * s0; a1 += c * s1; a2 += c * s2;
        }
    }
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        P.wv[3 * row] = a0; P.wv[3 * row + 1] = a1; P.wv[3 * row + 2] = a2;
        part[0] = P.rv[3 * row] * ul0 + P.rv[3 * row + 1] * ul1 + P.rv[3 * row + 2] * ul2;
        part[1] = a0 * ul0 + a1 * ul1 + a2 * ul2;
    }
    block_sum_store<9>(part, lds, tid, P.part_spmv + (size_t)b * NPART);
}

// =====================================================================================
// PCG kernel 1, LDS-staged path: the same operator in factored form.  The tile's u and the positions
// of the linearisation point are staged (own rows + halo); per incidence the kernel reads 12 bytes
// (nrs_engine_types.hpp): spring {4-byte header, qc}  a_i += qc (v . (u_i - u_j)) v,  v = x_i - x_j;
//                 damper {8-byte header, 4-byte weight}  a_i += sg_i s (sum_k sg_k u_k)  (all four vertices, the own one
//                 included), s re-formed from the weight unless the edge's Huber kernel is active;
// per row 32 bytes of reprojection factors instead of the 6x3 H_pl block and the 3x3 diagonal.
// =====================================================================================
template <int T, bool DF, bool TPC = false, bool H4 = false, bool RCS = false, bool RCD = false, bool NT = false>   // NT: non-temporal stream loads (Dev::nt); H4: 4-byte damper headers (Dev::d_h4; needs TPC); RCS / RCD: spring / damper factors re-formed from the staged linearisation point (Dev::rc)
__device__ __forceinline__ void spmv_f_body(const Dev& P, const double lam, const int cls, const int it, const double tol2, const int blk) {   // blk: the workgroup's index among the operator's (k_spmv_f_skin launches others behind them)
    static_assert(!(RCS || RCD) || (TPC && H4 && !DF), "factor recomputation: plain windows with cached partners and 4-byte headers");
    __shared__ double lds[4 * 9];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    constexpr int U = 2;                                           // records per lane and buffer (two buffers per stream)
    const int bi = xcd_tile(blk, P.sh_nt[cls] + P.sh_ntb[cls]);
    if (bi >= P.sh_nt[cls] + P.sh_ntb[cls]) return;
    const int b = P.tile_list[(cls ? P.n_tiles_cls[0] : 0) + (bi < P.sh_nt[cls] ? P.sh_t0[cls] + bi : P.sh_t0b[cls] + bi - P.sh_nt[cls])];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: slice bounds and loop conditions are scalar)
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    auto stamp = [&](int k) { if (H4 && P.dbg_clk && lane == 0) P.dbg_clk[(size_t)slice * 8 + k] = wall_clock64(); };   // (NRS_SPMV_DBG: phase clocks of one launch)
    stamp(0);
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    const uint32_t tp = TPC ? P.row_tp[row] : 0u;                  // plain BA windows: the row's temporal partners (tile-local ids)
    // ... and this lane's share of the row's incidences: slots beyond it are padding and are not requested
    const uint32_t rcn = TPC ? P.row_cnt[row] : 0u;
    const int my_s = ((int)(rcn & 0xFFFFu) + T - 1 - t) / T, my_d = ((int)(rcn >> 16) + T - 1 - t) / T;
    const int kf = P.grp_pose[row / ROW_ALIGN];
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    // DF (temporal-difference dampers): u, G^f, G^b for tile + halo, then the spring positions; else u, positions
    const size_t nst = (size_t)(P.tile_rows + P.cap_h[cls] + 1);
    double* lu = dyn;
    double* lgf = dyn + 3 * nst;
    double* lgb = dyn + 6 * nst;
    double* lx = dyn + (DF ? 9 : 3) * nst;
    // row ZROW of the arrays is zero: padding records and absent damper vertices point at it, so the
    // incidence loops are branch-free and the LDS reads of a whole chunk can be in flight together
    const int ZROW = P.tile_rows + P.cap_h[cls], ZROWX = RCD ? ZROW : P.tile_rows + P.cap_s[cls];   // (RCD: positions of the whole halo)
    if (tid < 3) {
        lu[3 * ZROW + tid] = 0; lx[3 * ZROWX + tid] = 0;
        if (DF) { lgf[3 * ZROW + tid] = 0; lgb[3 * ZROW + tid] = 0; }
    }
    // r.u of this iteration was left by the previous vector update (or the trial setup): its loads go
    // out with the staging loads, the sum rides on the staging barrier
    double gsum = 0;
    if (P.ecd) {
        const double* pr = P.part_ru + (size_t)(it & 1) * P.n_vecblk;
        if (P.hier) { if (tid == 0) gsum = P.red[3 + 6 * P.K]; }  // pre-reduced by k_reduce_ru
        else for (int i = tid; i < P.n_vecblk; i += BLK) gsum += pr[i];
        const double* upp = (it & 1) ? P.up2 : P.up;
        const double* rpp = (it & 1) ? P.rp2 : P.rp;
        for (int i = tid; i < 6 * P.K; i += BLK) gsum += rpp[i] * upp[i];
        gsum = wave_sum(gsum);
        if (lane == 0) lds[wave] = gsum;
    }
    const double gamma0 = P.scal[SC_GAMMA0];
    const int done_flag = P.flags[0];                              // launches behind a converged solve are no-ops
    if (DF) {
        stage_rows_d(P, b, tid, P.uv3, lu, lgf, lgb);
        const int row0 = b * P.tile_rows, hb = P.halo_ptr[b], ns = P.halo_ns[b];
        for (int i = tid; i < P.tile_rows + ns; i += BLK) {        // positions of the linearisation point: tile + spring halo
            const size_t r = 3 * (size_t)(i < P.tile_rows ? row0 + i : P.halo_rows[hb + i - P.tile_rows]);
            lx[3 * i] = P.lin_xl[r]; lx[3 * i + 1] = P.lin_xl[r + 1]; lx[3 * i + 2] = P.lin_xl[r + 2];
        }
    } else stage_rows2<TPC, RCD>(P, b, tid, P.uv3, P.lin_xl, P.X0, lu, lx);   // (TPC: plain windows, whose halo lists also sit at a fixed stride)
    // row factors and the first record chunks are requested while the staging loads are in flight
    RowRec rc;
    rc.w = 0;
    double rv0 = 0, rv1 = 0, rv2 = 0;
    if (t == 0) {
        rc = P.rowrec[row];
        if (!P.ecd) { rv0 = P.rv[3 * row]; rv1 = P.rv[3 * row + 1]; rv2 = P.rv[3 * row + 2]; }
    }
    // Records are double-buffered: chunk k+1 is requested before chunk k is consumed.  The loads are UNCONDITIONAL (slots
    // past the slice's end read a clamped index and count as padding when they are consumed) and the raw words are not
    // touched before the chunk is processed: with a predicated load the compiler sinks the unpacking into the predicated
    // block, right behind the load, and the wave waits out the full memory latency of every chunk it has just requested
    // (rounds 1 and 2: ~80 serialised round trips per tile).  sched_barrier keeps request and consumption apart.
    const int send_u = P.ss_ptr[slice + 1], dend_u = P.sd_ptr[slice + 1];   // (send / dend: empty for a fixed row)
    const int s_last = max(send_u - 1, 0), d_last = max(dend_u - 1, 0);
    uint32_t somA[U], somB[U], sd0A[U], sd0B[U];                   // (sd0: the rest length's bits, RCS)
    double sqcA[U], sqcB[U], dsA[U], dsB[U];
    uint2 dhA[U], dhB[U];
    auto load_springs = [&](uint32_t* om, double* qc, uint32_t* d0w, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (TPC) {
                om[q] = 0xFFFFu; qc[q] = 0.0;
                if (RCS) { d0w[q] = 0x3F800000u; if ((idx + 64 * q - sbeg - lane) / 64 < my_s) { om[q] = P.s_om[idx + 64 * q]; d0w[q] = __float_as_uint(P.s_d0[idx + 64 * q]); } continue; }
                if ((idx + 64 * q - sbeg - lane) / 64 < my_s) {
                    if (NT) { om[q] = __builtin_nontemporal_load(P.s_om + idx + 64 * q); qc[q] = __builtin_nontemporal_load(P.s_qc + idx + 64 * q); }
                    else { om[q] = P.s_om[idx + 64 * q]; qc[q] = P.s_qc[idx + 64 * q]; }
                }
                continue;
            }
            const int j = min(idx + 64 * q, s_last);
            om[q] = P.s_om[j];
            qc[q] = P.s_qc[j];
        }
    };
    auto load_dampers = [&](uint2* h, double* sv, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (TPC) {
                h[q] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu); sv[q] = 0.0;
                if (H4 && RCD) { if ((idx + 64 * q - dbeg - lane) / 64 < my_d) { h[q].x = P.d_h4[idx + 64 * q]; h[q].y = __float_as_uint(P.d_w[idx + 64 * q]); } }
                else if (H4) { if ((idx + 64 * q - dbeg - lane) / 64 < my_d) {
                    if (NT) { h[q].x = __builtin_nontemporal_load(P.d_h4 + idx + 64 * q); sv[q] = __builtin_nontemporal_load(P.d_s + idx + 64 * q); }
                    else { h[q].x = P.d_h4[idx + 64 * q]; sv[q] = P.d_s[idx + 64 * q]; } } }
                else if ((idx + 64 * q - dbeg - lane) / 64 < my_d) { h[q] = P.d_hdr[idx + 64 * q]; sv[q] = P.d_s[idx + 64 * q]; }
                continue;
            }
            const int j = min(idx + 64 * q, d_last);
            if (DF) h[q] = make_uint2(P.d_om[j], 0u);              // {partner | meta << 16} + s: 12 bytes
            else h[q] = P.d_hdr[j];
            sv[q] = P.d_s[j];                                      // (0 for padding slots and for edges at level != 0)
        }
    };
    load_springs(somA, sqcA, sd0A, sbeg + lane);
    load_dampers(dhA, dsA, dbeg + lane);
    __builtin_amdgcn_sched_barrier(0);
    // (one decision per workgroup: the flag can be raised -- by workgroup 0 of this very launch, or of k_pcg_update's -- between
    // the loads of two waves of a workgroup that starts late; see k_pcg_fused)
    const int done_wg = __syncthreads_or(done_flag);
    stamp(1);
    if (done_wg && !(H4 && P.dbg_clk)) return;
    if (P.ecd && it > 0) {
        const double gamma = lds[0] + lds[1] + lds[2] + lds[3];
        const bool bad = !isfinite(gamma);
        if (gamma <= tol2 * gamma0 || bad || gamma == 0.0) {       // converged: the operator is not applied again
            if (blk == 0 && tid == 0 && cls == 0) {
                if (bad) P.flags[2] = 1;
                P.flags[1] = it;
                __threadfence();
                P.flags[0] = 1;
            }
            return;
        }
        __syncthreads();                                           // lds is reused by the final reduction
    }
    const int self = row - b * P.tile_rows;
    const double ul[3] = {lu[3 * self], lu[3 * self + 1], lu[3 * self + 2]};
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    double a0 = 0, a1 = 0, a2 = 0;
    // u_i - u_next(i), u_i - u_prev(i): the first half of every damper term of this row (TPC: read once, not per incidence)
    double en[3] = {0, 0, 0}, ep[3] = {0, 0, 0}, exn[3] = {0, 0, 0}, exq[3] = {0, 0, 0};   // (exn / exq: the same differences of the linearisation point, RCD)
    const double ks = P.k_spring, ip = P.info_pos, isp = P.info_spatial;
    if (TPC) {
        const int tn = (int)(tp & 0xFFFFu) == REC_NONE ? ZROW : (int)(tp & 0xFFFFu), tq = (int)(tp >> 16) == REC_NONE ? ZROW : (int)(tp >> 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) { en[k] = ul[k] - lu[3 * tn + k]; ep[k] = ul[k] - lu[3 * tq + k]; }
        if (RCD) {
            const int xn = (int)(tp & 0xFFFFu) == REC_NONE ? self : (int)(tp & 0xFFFFu), xq = (int)(tp >> 16) == REC_NONE ? self : (int)(tp >> 16);
#pragma unroll
            for (int k = 0; k < 3; ++k) { exn[k] = xs[k] - lx[3 * xn + k]; exq[k] = xs[k] - lx[3 * xq + k]; }
        }
    }
    auto do_springs = [&](const uint32_t* om, const double* qcv, const uint32_t* d0w, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int o16 = (int)(consume(om[q]) & 0xFFFFu);
            const bool pad = idx + 64 * q >= send || o16 == REC_NONE;
            const int o = pad ? ZROW : o16, ox = pad ? ZROWX : o16;
            double qc = pad || RCS ? 0.0 : consume(qcv[q]);
            const double v0 = xs[0] - lx[3 * ox], v1 = xs[1] - lx[3 * ox + 1], v2 = xs[2] - lx[3 * ox + 2];
            if (RCS) { const double d2 = sq3(v0, v1, v2); qc = pad ? 0.0 : spring_qc(pad ? 1.0 : d2, (double)__uint_as_float(consume(d0w[q])), ks, ip); }
            const double dot = qc * (v0 * (ul[0] - lu[3 * o]) + v1 * (ul[1] - lu[3 * o + 1]) + v2 * (ul[2] - lu[3 * o + 2]));
            a0 += dot * v0; a1 += dot * v1; a2 += dot * v2;
        }
    };
    const double gfo[3] = {DF ? lgf[3 * self] : 0.0, DF ? lgf[3 * self + 1] : 0.0, DF ? lgf[3 * self + 2] : 0.0};
    const double gbo[3] = {DF ? lgb[3 * self] : 0.0, DF ? lgb[3 * self + 1] : 0.0, DF ? lgb[3 * self + 2] : 0.0};
    auto do_dampers = [&](const uint2* hd, const double* dsv, int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const bool live = idx + 64 * q < dend;
            if (DF) {
                // a_i += s (G^d_i - G^d_o): forward differences for roles 1c / 2c, backward for 1n / 2n
                const uint32_t hx = consume(hd[q].x);
                const int p16 = (int)(hx & 0xFFFFu), m16 = (int)(hx >> 16);
                const bool pad = !live || m16 == REC_NONE || p16 == REC_NONE;
                const int o = pad ? ZROW : p16;
                const double sv = pad ? 0.0 : consume(dsv[q]);
                const bool bw = (m16 & 2) != 0;
                const double* lg = bw ? lgb : lgf;
                const double g0 = (bw ? gbo[0] : gfo[0]) - lg[3 * o], g1 = (bw ? gbo[1] : gfo[1]) - lg[3 * o + 1], g2 = (bw ? gbo[2] : gfo[2]) - lg[3 * o + 2];
                a0 += sv * g0; a1 += sv * g1; a2 += sv * g2;
                continue;
            }
            // padding records carry s = 0; a unary damper (the other vertex is a value) and absent
            // vertices read the zero row, which leaves the diagonal term s u_i
            const uint32_t hx = consume(hd[q].x), hy = H4 ? 0u : consume(hd[q].y);
            const int r0 = (int)(hx & (H4 ? 0xFFFu : 0xFFFFu)), r1 = H4 ? 0 : (int)(hx >> 16), r2 = (int)(H4 ? (hx >> 12) & 0xFFFu : hy & 0xFFFFu),
                      m16 = H4 ? (hx == 0xFFFFFFFFu ? (int)REC_NONE : (int)(hx >> 24)) : (int)(hy >> 16);
            const bool pad = !live || m16 == REC_NONE;
            double sv = pad || RCD ? 0.0 : consume(dsv[q]);
            const bool un = pad || (m16 & DM_UNARY) != 0;
            const int o0 = H4 ? (pad ? ZROW : r0) : (un || r0 == REC_NONE) ? ZROW : r0;      // (H4: plain windows, every damper with its four vertices)
            const int o1 = (un || r1 == REC_NONE) ? ZROW : r1;
            const int o2 = H4 ? (pad ? ZROW : r2) : (un || r2 == REC_NONE) ? ZROW : r2;
            // the others come in canonical order (engine_create): a_i += s ((u_i - u[o1]) - (u[o0] - u[o2])) for every role
            double g0, g1, g2;
            if (TPC) {
                const bool fwd = (m16 & 2) == 0;                     // roles 1c / 2c: the partner is in the next keyframe
                g0 = (fwd ? en[0] : ep[0]) - (lu[3 * o0] - lu[3 * o2]);
                g1 = (fwd ? en[1] : ep[1]) - (lu[3 * o0 + 1] - lu[3 * o2 + 1]);
                g2 = (fwd ? en[2] : ep[2]) - (lu[3 * o0 + 2] - lu[3 * o2 + 2]);
                if (RCD) {                                           // the factor of the linearisation point, as k_lin_plain forms it
                    const double x0 = (fwd ? exn[0] : exq[0]) - (lx[3 * o0] - lx[3 * o2]), x1 = (fwd ? exn[1] : exq[1]) - (lx[3 * o0 + 1] - lx[3 * o2 + 1]),
                                 x2 = (fwd ? exn[2] : exq[2]) - (lx[3 * o0 + 2] - lx[3 * o2 + 2]);
                    double rho0;
                    sv = pad ? 0.0 : damper_s(x0, x1, x2, (double)__uint_as_float(consume(hd[q].y)), isp, P.delta_spatial, rho0);
                }
            } else {
                g0 = (ul[0] - lu[3 * o1]) - (lu[3 * o0] - lu[3 * o2]);
                g1 = (ul[1] - lu[3 * o1 + 1]) - (lu[3 * o0 + 1] - lu[3 * o2 + 1]);
                g2 = (ul[2] - lu[3 * o1 + 2]) - (lu[3 * o0 + 2] - lu[3 * o2 + 2]);
            }
            a0 += sv * g0; a1 += sv * g1; a2 += sv * g2;
        }
    };
    for (int base = sbeg; base < send_u; base += 128 * U) {        // wave-uniform trip count
        load_springs(somB, sqcB, sd0B, base + 64 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
        do_springs(somA, sqcA, sd0A, base + lane);
        __builtin_amdgcn_sched_barrier(0);
        load_springs(somA, sqcA, sd0A, base + 128 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
        do_springs(somB, sqcB, sd0B, base + 64 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2);
    for (int base = dbeg; base < dend_u; base += 128 * U) {
        load_dampers(dhB, dsB, base + 64 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
        do_dampers(dhA, dsA, base + lane);
        __builtin_amdgcn_sched_barrier(0);
        load_dampers(dhA, dsA, base + 128 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
        do_dampers(dhB, dsB, base + 64 * U + lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(3);
    // the row's own terms come last: their temporaries then never coexist with the record registers
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (t == 0) {
        a0 += lam * ul[0]; a1 += lam * ul[1]; a2 += lam * ul[2];
        if (rc.w != 0.0) {
            double up[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) up[p] = ((it & 1) ? P.up2 : P.up)[6 * kf + p];
            row_factored(rc, P.lin_pose[kf], xs, ul, up, P.pose_fixed[kf] ? 0.0 : 1.0, a0, a1, a2, part);
        }
    }
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        P.wv[3 * row] = a0; P.wv[3 * row + 1] = a1; P.wv[3 * row + 2] = a2;
        part[0] = P.ecd ? 0.0 : rv0 * ul[0] + rv1 * ul[1] + rv2 * ul[2];
        part[1] = a0 * ul[0] + a1 * ul[1] + a2 * ul[2];
    }
    stamp(4);
    block_sum_store<9>(part, lds, tid, P.part_spmv + (size_t)b * NPART);
    stamp(5);
}
template <int T, bool DF, bool TPC = false, bool H4 = false, bool RCS = false, bool RCD = false, bool NT = false>
__global__ __launch_bounds__(BLK, RCD ? 3 : 4) void k_spmv_f(Dev P, double lam, int cls, int it, double tol2) {
    spmv_f_body<T, DF, TPC, H4, RCS, RCD, NT>(P, lam, cls, it, tol2, (int)blockIdx.x);
}

// =====================================================================================
// large problems only: fixed-order reduction of the SpMV partials.  Workgroup 0: the three dot
// partials over all workgroups; workgroup 1 + k: the six pose sums of pose k.
// =====================================================================================
// large problems: fixed-order sum of the r.u partials the vector update (or the trial setup) left for
// iteration `it`, so that the operator kernel reads one number
__global__ __launch_bounds__(BLK) void k_reduce_ru(Dev P, int it) {
    __shared__ double lds[4];
    const int tid = threadIdx.x;
    const double* pr = P.part_ru + (size_t)(it & 1) * P.n_vecblk;
    double v[1] = {0};
    for (int i = tid; i < P.n_vecblk; i += BLK) v[0] += pr[i];
    block_sum<1>(v, lds, tid & 63, tid >> 6);
    if (tid == 0) P.red[3 + 6 * P.K] = v[0];
}

__global__ __launch_bounds__(BLK) void k_reduce_partials(Dev P) {
    __shared__ double lds[4 * 6];
    if (P.flags[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0) {
        double v[3] = {0, 0, 0};
        for (int b = tid; b < P.n_regblk; b += BLK) {
            v[0] += P.part_spmv[(size_t)b * NPART];
            v[1] += P.part_spmv[(size_t)b * NPART + 1];
            v[2] += P.part_spmv[(size_t)b * NPART + 2];
        }
        if (P.sk_pcg) {                                            // embedded BA window: what the skinned observations add to w.u (row pass) and to the cross term
            for (int b = tid; b < P.sk_nblk; b += BLK) { v[1] += P.sk_opart[(size_t)b * 8 + 7]; v[2] += P.sk_opart[(size_t)b * 8 + 6]; }
        }
        block_sum<3>(v, lds, lane, wave);
        if (tid == 0) { P.red[0] = v[0]; P.red[1] = v[1]; P.red[2] = v[2]; }
    } else {
        const int k = blockIdx.x - 1;
        const int rb = ROW_ALIGN / (BLK / P.T);
        const int g0 = P.pose_grp_ptr[k] * rb, g1 = P.pose_grp_ptr[k + 1] * rb;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int g = g0 + tid; g < g1; g += BLK) {
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] += P.part_spmv[(size_t)g * NPART + 3 + a];
        }
        if (P.sk_pcg)
            for (int b = P.sk_pose_blk[k] + tid; b < P.sk_pose_blk[k + 1]; b += BLK) {
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[a] += P.sk_opart[(size_t)b * 8 + a];
            }
        block_sum<6>(acc, lds, lane, wave);
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) P.red[3 + 6 * k + a] = acc[a];
        }
    }
}

// =====================================================================================
// PCG kernel 2 (Chronopoulos-Gear single-reduction CG): every workgroup re-derives the scalars
// from the partials in a fixed order, then updates its rows:
//   gamma = r.u, delta = w.u, beta = gamma/gamma_old, alpha = gamma/(delta - beta*gamma/alpha_old)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = M^-1 r
// Workgroups >= n_vec8 own the pose rows (w_p = (H_pp + lambda) u_p + sum_l H_pl u_l).
// =====================================================================================
// a node row's list, SK_RL lanes: a += om g_o over the entries this lane takes (the lanes are combined by sub_sum_t<SK_RL>)
__device__ __forceinline__ void skin_row_gather(const Dev& P, const int q0, const int q1, const int t, double a[3]) {
    int q = q0 + t;
    for (; q + SK_RL < q1; q += 2 * SK_RL) {                       // two entries in flight (added in list order: the same sums)
        const double om0 = P.sk_rl_om[q], om1 = P.sk_rl_om[q + SK_RL];
        const double* g0 = P.sk_g + 4 * (size_t)P.sk_rl_obs[q];
        const double* g1 = P.sk_g + 4 * (size_t)P.sk_rl_obs[q + SK_RL];
        const double x0 = g0[0], x1 = g0[1], x2 = g0[2], y0 = g1[0], y1 = g1[1], y2 = g1[2];
        a[0] += om0 * x0; a[1] += om0 * x1; a[2] += om0 * x2;
        a[0] += om1 * y0; a[1] += om1 * y1; a[2] += om1 * y2;
    }
    if (q < q1) {
        const double om = P.sk_rl_om[q];
        const double* g = P.sk_g + 4 * (size_t)P.sk_rl_obs[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] += om * g[k];
    }
}

// SKR (embedded BA window, nrs_engine_skin.hpp): a row workgroup owns SK_RPB rows with SK_RL lanes each and first completes their w
// with the observations' row pass (w_row += sum om g_o, the gather in flight while the scalars are derived); lane 0 of a row updates it.
template <bool SKR = false>
__global__ __launch_bounds__(BLK) void k_pcg_update(Dev P, double lam, int it, double tol2, double peek_tol2, int pub_seq) {
    __shared__ double lds[4 * 3];
    const int n_vecblk = P.sh_nvb;                                 // own row range (the whole problem when not sharded)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // u_p and r_p are read by every workgroup (for the scalars) and rewritten by the pose workgroups
    // of the same launch: they are a ping-pong pair (read half it&1, write the other)
    const double* up_in = (it & 1) ? P.up2 : P.up;   double* up_out = (it & 1) ? P.up : P.up2;
    const double* rp_in = (it & 1) ? P.rp2 : P.rp;   double* rp_out = (it & 1) ? P.rp : P.rp2;
    // Everything this launch reads is requested before the first dependent use (flag, scalars,
    // partials, the two rows of this thread): otherwise the launch is a chain of four round trips.
    const int done_flag = P.flags[0];
    const double sc_gamma0 = P.scal[SC_GAMMA0];
    const double sc_slot0 = P.scal[(it & 1) ? SC_SLOT1 : SC_SLOT0], sc_slot1 = P.scal[((it & 1) ? SC_SLOT1 : SC_SLOT0) + 1];
    const int n_vec2 = (n_vecblk + 1) >> 1;
    const int n_vec8 = SKR ? P.n_rows / SK_RPB : ((n_vec2 + 7) >> 3) << 3;          // (row workgroups of this launch)
    const bool row_wg = (int)blockIdx.x < n_vec8;
    const int pair = (row_wg && !SKR) ? P.sh_vb0 * (BLK / 2) + xcd_tile(blockIdx.x, n_vec2) * BLK + tid : 0;
    const int sk_r = (int)blockIdx.x * SK_RPB + tid / SK_RL, sk_t = tid % SK_RL;   // SKR: this lane's row and its place in the row's group
    const bool has_rows = SKR ? row_wg && sk_t == 0 : row_wg && 2 * pair < (P.sh_vb0 + n_vecblk) * BLK;
    const size_t o = SKR ? 3 * (size_t)sk_r : 6 * (size_t)pair;
    double uu[6], pp[6], ww[6], ss[6], rr[6], xx[6], Di[12];
    double ska[3] = {0, 0, 0};
    bool sk_add = false;
    if (SKR) {
        if (row_wg) {
            const int q0 = P.sk_row_q[2 * (size_t)sk_r], q1 = P.sk_row_q[2 * (size_t)sk_r + 1];
            sk_add = q1 > q0 && !(P.rflag[sk_r] & RF_FIXED);      // (a fixed row is an identity row: nothing is added to it)
            if (has_rows) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    uu[k] = P.uv3[o + k]; pp[k] = P.pv[o + k]; ww[k] = P.wv[o + k]; ss[k] = P.sv[o + k]; rr[k] = P.rv[o + k]; xx[k] = P.xv[o + k];
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) Di[k] = P.Dinv[2 * o + k];
            }
            if (sk_add) skin_row_gather(P, q0, q1, sk_t, ska);
#pragma unroll
            for (int k = 0; k < 3; ++k) ska[k] = sub_sum_t<SK_RL>(ska[k]);
        }
    } else if (has_rows) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double2 a = *reinterpret_cast<const double2*>(P.uv3 + o + 2 * k);
            const double2 b = *reinterpret_cast<const double2*>(P.pv + o + 2 * k);
            const double2 c = *reinterpret_cast<const double2*>(P.wv + o + 2 * k);
            const double2 d = *reinterpret_cast<const double2*>(P.sv + o + 2 * k);
            const double2 e = *reinterpret_cast<const double2*>(P.rv + o + 2 * k);
            const double2 f = *reinterpret_cast<const double2*>(P.xv + o + 2 * k);
            uu[2 * k] = a.x; uu[2 * k + 1] = a.y; pp[2 * k] = b.x; pp[2 * k + 1] = b.y;
            ww[2 * k] = c.x; ww[2 * k + 1] = c.y; ss[2 * k] = d.x; ss[2 * k + 1] = d.y;
            rr[2 * k] = e.x; rr[2 * k + 1] = e.y; xx[2 * k] = f.x; xx[2 * k + 1] = f.y;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double2 a = *reinterpret_cast<const double2*>(P.Dinv + 2 * o + 2 * k);
            Di[2 * k] = a.x; Di[2 * k + 1] = a.y;
        }
    }
    // pose workgroups (one wave per pose): their inputs are requested up front as well
    const int pk_pose = row_wg ? P.K : (int)(blockIdx.x - n_vec8) * 4 + wave;
    const bool has_pose = pk_pose < P.K;
    const int pa = lane < 6 ? lane : 0;
    double q_up[6], q_H[6], q_Hi[6], q_pp = 0, q_sp = 0, q_rp = 0, q_xp = 0, q_acc[6] = {0, 0, 0, 0, 0, 0};
    int pg0 = 0, pg1 = 0;
    if (has_pose) {
        const int rb = ROW_ALIGN / (BLK / P.T);       // reg-blocks per row group
        pg0 = P.pose_grp_ptr[pk_pose] * rb; pg1 = P.pose_grp_ptr[pk_pose + 1] * rb;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int lo = pa < c ? pa : c, hi = pa < c ? c : pa;
            const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
            q_up[c] = up_in[6 * pk_pose + c];
            q_H[c] = P.Hpp[21 * pk_pose + pk];
            q_Hi[c] = P.Hppinv[36 * pk_pose + pa * 6 + c];
        }
        q_pp = P.pp[6 * pk_pose + pa]; q_sp = P.sp[6 * pk_pose + pa]; q_rp = rp_in[6 * pk_pose + pa]; q_xp = P.xp[6 * pk_pose + pa];
        if (P.hier) {
#pragma unroll
            for (int a = 0; a < 6; ++a) q_acc[a] = P.red[3 + 6 * pk_pose + a];
        } else if (pg0 + lane < pg1) {
#pragma unroll
            for (int a = 0; a < 6; ++a) q_acc[a] += P.part_spmv[(size_t)(pg0 + lane) * NPART + 3 + a];
        }
    }
    double v[3] = {0, 0, 0};
    if (P.hier) {
        if (tid == 0) { v[0] = P.ecd ? P.red[3 + 6 * P.K] : P.red[0]; v[1] = P.red[1]; v[2] = P.red[2]; }
    } else {
        for (int b = tid; b < P.n_regblk; b += BLK) {
            if (!P.ecd) v[0] += P.part_spmv[(size_t)b * NPART];
            v[1] += P.part_spmv[(size_t)b * NPART + 1];
            v[2] += P.part_spmv[(size_t)b * NPART + 2];
        }
        if (P.ecd) {
            const double* pr = P.part_ru + (size_t)(it & 1) * P.n_vecblk;
            for (int b = tid; b < P.n_vecblk; b += BLK) v[0] += pr[b];
        }
        if (P.sk_pcg) {                                            // embedded BA window: the skinned observations' shares of w.u and of the cross term
            for (int b = tid; b < P.sk_nblk; b += BLK) { v[1] += P.sk_opart[(size_t)b * 8 + 7]; v[2] += P.sk_opart[(size_t)b * 8 + 6]; }
        }
    }
    // pose rows: gamma_p = r_p.u_p ; delta_p = u_p.(H_pp + lam)u_p + cross (cross is v[2])
    for (int i = tid; i < 6 * P.K; i += BLK) {
        const int k = i / 6, a = i % 6;
        const double ua = up_in[i];
        v[0] += rp_in[i] * ua;
        double s = lam * ua;
        for (int c = 0; c < 6; ++c) {
            const int lo = a < c ? a : c, hi = a < c ? c : a;
            const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);     // packed upper index
            s += P.Hpp[21 * k + pk] * up_in[6 * k + c];
        }
        v[1] += ua * s;
    }
    // pub_seq != 0: this is the last launch of a batch the host is waiting for -- the thread that owns the
    // flags publishes them (mapped host memory + sequence word) as soon as they are final
    const bool publisher = pub_seq != 0 && blockIdx.x == 0 && tid == 0;
    // (one decision per workgroup: workgroup 0 raises the flag in the launch that finds the solve converged, while workgroups of
    // that launch are still starting -- with a per-thread test the waves that loaded the word after the store left, the sums below
    // lost their shares and the remaining rows were updated with garbage scalars; found on the single-launch form, k_pcg_fused)
    if (__syncthreads_or(done_flag)) {
        if (publisher) publish_flags(P, pub_seq);
        return;
    }
    block_sum<3>(v, lds, lane, wave);
    const double gamma = v[0], delta = v[1] + v[2];
    double* nslot = P.scal + ((it & 1) ? SC_SLOT0 : SC_SLOT1);
    const double gamma0 = it == 0 ? gamma : sc_gamma0;
    const bool bad = !isfinite(gamma) || !isfinite(delta);
    const bool conv = (gamma <= tol2 * gamma0) || bad || gamma == 0.0;
    if (conv) {
        if (blockIdx.x == 0 && tid == 0) {
            if (bad) P.flags[2] = 1;
            P.flags[1] = it;
            __threadfence();
            P.flags[0] = 1;
            if (publisher) publish_flags(P, pub_seq);
        }
        return;
    }
    const double beta = it == 0 ? 0.0 : gamma / sc_slot0;
    const double alpha = it == 0 ? gamma / delta : gamma / (delta - beta * gamma / sc_slot1);
    if (blockIdx.x == 0 && tid == 0) {
        nslot[0] = gamma;
        nslot[1] = alpha;
        if (it == 0) P.scal[SC_GAMMA0] = gamma;
        P.flags[1] = it + 1;
        // "peek" milestones for early trial rejection: level 1 at peek_tol, level 2 at peek_tol/10
        if (gamma <= peek_tol2 * gamma0) {
            P.flags[3] = max(P.flags[3], (gamma <= 1e-6 * peek_tol2 * gamma0 ? 4 : gamma <= 1e-4 * peek_tol2 * gamma0 ? 3 : gamma <= 1e-2 * peek_tol2 * gamma0 ? 2 : 1));
            if (P.flags[4] == 0) P.flags[4] = it + 1;               // iterations the first milestone took (sizes the next first batch)
        }
        if (publisher) publish_flags(P, pub_seq);
    }
    // row workgroups: every thread updates TWO consecutive rows (6 doubles = three 16-byte
    // accesses per vector); n_rows is a multiple of 256, so pairs never straddle anything
    if (SKR) {
        if (has_rows) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (sk_add) ww[k] += ska[k];
                pp[k] = uu[k] + beta * pp[k];
                ss[k] = ww[k] + beta * ss[k];
                xx[k] += alpha * pp[k];
                rr[k] -= alpha * ss[k];
            }
            uu[0] = Di[0] * rr[0] + Di[1] * rr[1] + Di[2] * rr[2];
            uu[1] = Di[1] * rr[0] + Di[3] * rr[1] + Di[4] * rr[2];
            uu[2] = Di[2] * rr[0] + Di[4] * rr[1] + Di[5] * rr[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) { P.pv[o + k] = pp[k]; P.sv[o + k] = ss[k]; P.xv[o + k] = xx[k]; P.rv[o + k] = rr[k]; P.uv3[o + k] = uu[k]; }
        }
    }
    if (row_wg && !SKR) {
        if (has_rows) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                pp[k] = uu[k] + beta * pp[k];
                ss[k] = ww[k] + beta * ss[k];
                xx[k] += alpha * pp[k];
                rr[k] -= alpha * ss[k];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double* Dh = Di + 6 * h;
                const double r0 = rr[3 * h], r1 = rr[3 * h + 1], r2 = rr[3 * h + 2];
                uu[3 * h] = Dh[0] * r0 + Dh[1] * r1 + Dh[2] * r2;
                uu[3 * h + 1] = Dh[1] * r0 + Dh[3] * r1 + Dh[4] * r2;
                uu[3 * h + 2] = Dh[2] * r0 + Dh[4] * r1 + Dh[5] * r2;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                *reinterpret_cast<double2*>(P.pv + o + 2 * k) = make_double2(pp[2 * k], pp[2 * k + 1]);
                *reinterpret_cast<double2*>(P.sv + o + 2 * k) = make_double2(ss[2 * k], ss[2 * k + 1]);
                *reinterpret_cast<double2*>(P.xv + o + 2 * k) = make_double2(xx[2 * k], xx[2 * k + 1]);
                *reinterpret_cast<double2*>(P.rv + o + 2 * k) = make_double2(rr[2 * k], rr[2 * k + 1]);
                *reinterpret_cast<double2*>(P.uv3 + o + 2 * k) = make_double2(uu[2 * k], uu[2 * k + 1]);
            }
        }
        if (P.ecd) {                                               // r.u of the next iteration, other half of the pair
            double ru[1] = {0};
            if (has_rows) {
#pragma unroll
                for (int k = 0; k < 6; ++k) ru[0] += rr[k] * uu[k];
            }
            block_sum<1>(ru, lds, lane, wave);
            const int w2 = 2 * xcd_tile(blockIdx.x, n_vec2);
            double* pw = P.part_ru + (size_t)((it + 1) & 1) * P.n_vecblk;
            if (tid == 0 && w2 < P.n_vecblk) pw[w2] = ru[0];
            if (tid == 1 && w2 + 1 < P.n_vecblk) pw[w2 + 1] = 0.0;
        }
    } else if (!row_wg) {
        // pose workgroups: one wave per pose; its 64 lanes split the pose's SpMV partials
        if (has_pose) {
            const int k = pk_pose;
            double acc[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a] = q_acc[a];
            if (!P.hier) {
                for (int g = pg0 + lane + 64; g < pg1; g += 64) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) acc[a] += P.part_spmv[(size_t)g * NPART + 3 + a];
                }
                if (P.sk_pcg)                                      // ... and of the pose rows: sum of B s over the pose's observation blocks
                    for (int b = P.sk_pose_blk[k] + lane; b < P.sk_pose_blk[k + 1]; b += 64) {
#pragma unroll
                        for (int a = 0; a < 6; ++a) acc[a] += P.sk_opart[(size_t)b * 8 + a];
                    }
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[a] = wave_sum(acc[a]);
            }
            // lanes 0..5 own one pose component each
            const int a = pa;
            double hw = acc[0], ua = q_up[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) { hw = (a == q) ? acc[q] : hw; ua = (a == q) ? q_up[q] : ua; }
            const int i = 6 * k + a;
            double w = lam * ua + hw;
#pragma unroll
            for (int c = 0; c < 6; ++c) w += q_H[c] * q_up[c];
            const double p = ua + beta * q_pp;
            const double sN = w + beta * q_sp;
            const double rnew = q_rp - alpha * sN;
            double unew = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) unew += q_Hi[c] * __shfl(rnew, c, 64);
            if (lane < 6) {
                P.pp[i] = p;
                P.sp[i] = sN;
                P.xp[i] = q_xp + alpha * p;
                rp_out[i] = rnew;
                up_out[i] = unew;
            }
        }
    }
}

// =====================================================================================
// Fused PCG iteration for small problems (single-frame tracking, short BA windows): ONE launch per
// iteration.  F(it) = [vector update of iteration it-1] followed by [operator apply of iteration
// it].  Every workgroup re-derives the scalars from the previous launch's partials, updates its own
// rows, and RECOMPUTES the updated u of its halo rows from (r, s, w, M^-1) instead of waiting for
// their owners -- so there is no inter-workgroup hand-off inside a launch.  r, s, w, the pose
// vectors and the partials are ping-pong pairs (read half (it+1)&1, write half it&1): owners write
// the new values while neighbours still read the old ones.  Same arithmetic, in the same order, as
// k_pcg_update + k_spmv.
// =====================================================================================
template <int T, bool CO>
__global__ __launch_bounds__(BLK) void k_pcg_fused(Dev P, double lam, int it, double tol2, double peek_tol2, int pub_seq) {
    __shared__ double lds[4 * 9];
    __shared__ double s_up[6];
    extern __shared__ double dyn[];
    constexpr int R = 64 / T;
    constexpr int U = 4;
    if (P.one_xcd && (blockIdx.x & 7)) return;                     // (small frames: one workgroup in eight works, so that all of them sit on one XCD)
    const int b = P.one_xcd ? (int)(blockIdx.x >> 3) : xcd_tile(blockIdx.x, P.n_regblk);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // read half / write half of the ping-pong pairs.  Launch 0 only applies the operator: the state
    // written by k_trial_setup lives in half 0 and stays there.
    const int hin = it == 0 ? 0 : ((it + 1) & 1), hout = it & 1;
    const double* r_in = hin ? P.rv2 : P.rv;    double* r_out = hout ? P.rv2 : P.rv;
    const double* s_in = hin ? P.sv2 : P.sv;    double* s_out = hout ? P.sv2 : P.sv;
    const double* w_in = hin ? P.wv2 : P.wv;    double* w_out = hout ? P.wv2 : P.wv;
    const double* rp_in = hin ? P.rp2 : P.rp;   double* rp_out = hout ? P.rp2 : P.rp;
    const double* sp_in = hin ? P.sp2 : P.sp;   double* sp_out = hout ? P.sp2 : P.sp;
    const double* up_in = hin ? P.up2 : P.up;   double* up_out = hout ? P.up2 : P.up;
    const double* part_in = hin ? P.part_spmv2 : P.part_spmv;
    double* part_out = hout ? P.part_spmv2 : P.part_spmv;
    const double* ts_in = hin ? P.part_ts2 : P.part_ts;
    double* ts_out = hout ? P.part_ts2 : P.part_ts;
    const int slice = b * 4 + wave;
    const int row = slice * R + lane / T;
    const int t = lane % T;
    const int row0 = b * P.tile_rows;
    auto stamp = [&](int k) { if (P.dbg_clk && tid == 0) P.dbg_clk[(size_t)b * 8 + k] = wall_clock64(); };   // (NRS_PCG_DBG: phase clocks of one launch)
    stamp(0);
    // The launch is a chain of dependent memory round trips unless everything is requested at once:
    // level 1 = whatever is addressed by the tile index alone (tile descriptor, flags, scalars, all
    // partials, own rows, fixed-stride halo list, slice pointers), level 2 = what those address
    // (records, halo rows, the pose's partials and vectors).  Nothing is loaded after that.
    const int4 td = *reinterpret_cast<const int4*>(P.tile_desc + 8 * (size_t)b);
    const int kf = td.x, pg0 = td.y, pg1 = td.z, hb = td.w;       // a tile never straddles two poses
    const int hn = P.tile_desc[8 * (size_t)b + 4];
    const int done_flag = P.flags[0];
    const int ipq = it > 0 ? it - 1 : 0;                           // PCG iteration whose scalars this launch finishes
    const double sc_gamma0 = P.scal[SC_GAMMA0];
    const double sc_slot0 = P.scal[(ipq & 1) ? SC_SLOT1 : SC_SLOT0], sc_slot1 = P.scal[((ipq & 1) ? SC_SLOT1 : SC_SLOT0) + 1];
    const int self = row - row0;
    const bool rfix = (P.rflag[row] & RF_FIXED) != 0;
    double* lu = dyn;
    double* lx = dyn + 3 * (size_t)(P.tile_rows + P.max_halo);     // positions of the linearisation point

    // ================= phase 1: every global load this launch needs is requested up front (the
    // launch is a chain of dependent round trips otherwise: partials -> vectors -> records)
    double v[3] = {0, 0, 0};
    const bool coarse = CO && it > 0;                              // CO: two-level preconditioner compiled in
    double ts9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // thread q: sums of r, s, w of tile q (coarse level)
    if (it > 0) {
        for (int q = tid; q < P.n_regblk; q += BLK) {
            v[0] += part_in[(size_t)q * NPART];
            v[1] += part_in[(size_t)q * NPART + 1];
            v[2] += part_in[(size_t)q * NPART + 2];
        }
        if (coarse && tid < P.n_regblk) {
#pragma unroll
            for (int c = 0; c < 9; ++c) ts9[c] = ts_in[(size_t)c * P.n_regblk + tid];   // component-major: coalesced
        }
        for (int i = tid; i < 6 * P.K; i += BLK) {
            const int k = i / 6, a = i % 6;
            const double ua = up_in[i];
            v[0] += rp_in[i] * ua;
            double sacc = lam * ua;
            for (int c = 0; c < 6; ++c) {
                const int lo = a < c ? a : c, hi = a < c ? c : a;
                const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
                sacc += P.Hpp[21 * k + pk] * up_in[6 * k + c];
            }
            v[1] += ua * sacc;
        }
    }
    // own row (one thread per row) and first halo row of this thread
    const bool own = tid < P.tile_rows;
    const size_t orow = (size_t)(row0 + (own ? tid : 0));
    double o_u[3], o_p[3], o_w[3], o_s[3], o_r[3], o_x[3], o_D[6];
    const bool hh = tid < hn;
    const size_t hrow = (size_t)P.halo_fix[(size_t)b * BLK + tid];
    double h_w[3], h_s[3], h_r[3], h_D[6];
    double x_own[3] = {0, 0, 0}, x_h[3] = {0, 0, 0};
    const bool o_free = own && !(P.rflag[orow] & RF_FIXED);
    const bool h_free = hh && !(P.rflag[hrow] & RF_FIXED);
    const int cn = P.co_n, cG = P.n_groups;
    double bt_q[6];                                                  // (B_t + lambda n_t)^-1 of tile `tid`
    const int th_h = (int)(hrow / (size_t)P.tile_rows);              // tile of this thread's halo row
    // A_c^-1 y: thread (row r, part) owns a run of columns.  With <= 64 coarse unknowns (frames up to ~4.9k
    // points) four parts cover the workgroup and the <= 16 matrix entries of a thread are requested here,
    // with all the other loads of the launch; larger systems use two parts and load on the fly.
    constexpr int CO_PF = 16;
    const bool co4 = 4 * cn <= BLK;
    const int co_parts = co4 ? 4 : 2;
    const int co_r = tid % (cn > 0 ? cn : 1), co_part = tid / (cn > 0 ? cn : 1);
    const int co_len = (cn + co_parts - 1) / co_parts;
    const int co_c0 = co_part * co_len, co_c1 = min(cn, co_c0 + co_len);
    double c_m[CO_PF];
    if (coarse) {
#pragma unroll
        for (int q = 0; q < 6; ++q) bt_q[q] = tid < P.n_regblk ? P.co_bti[6 * (size_t)tid + q] : 0.0;
        if (co4) {
#pragma unroll
            for (int q = 0; q < CO_PF; ++q) {
                const int c = co_c0 + q;
                c_m[q] = (co_part < 4 && c < co_c1) ? P.co_inv[(size_t)c * cn + co_r] : 0.0;   // symmetric: column read, coalesced over r
            }
        }
    }
    if (own) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x_own[k] = P.lin_xl[3 * orow + k] + (P.X0 ? P.X0[3 * orow + k] : 0.0);
    }
    if (hh) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x_h[k] = P.lin_xl[3 * hrow + k] + (P.X0 ? P.X0[3 * hrow + k] : 0.0);
    }
    if (it > 0) {
        if (own) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o_u[k] = P.uv3[3 * orow + k]; o_p[k] = P.pv[3 * orow + k]; o_w[k] = w_in[3 * orow + k];
                o_s[k] = s_in[3 * orow + k]; o_r[k] = r_in[3 * orow + k]; o_x[k] = P.xv[3 * orow + k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) o_D[k] = P.Dinv[6 * orow + k];
        }
        if (hh) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { h_w[k] = w_in[3 * hrow + k]; h_s[k] = s_in[3 * hrow + k]; h_r[k] = r_in[3 * hrow + k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) h_D[k] = P.Dinv[6 * hrow + k];
        }
    } else {
        stage_rows(P, b, tid, P.uv3, nullptr, lu);
    }
    // wave 0: everything the pose-vector update of this tile's pose needs
    const int pa = lane < 6 ? lane : 0;
    double q_up[6], q_H[6], q_Hi[6], q_sp = 0, q_rp = 0, q_pp = 0, q_xp = 0, q_acc[6] = {0, 0, 0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) q_up[c] = up_in[6 * kf + c];
        if (it > 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int lo = pa < c ? pa : c, hi = pa < c ? c : pa;
                const int pk = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
                q_H[c] = P.Hpp[21 * kf + pk];
                q_Hi[c] = P.Hppinv[36 * kf + pa * 6 + c];
            }
            q_sp = sp_in[6 * kf + pa];
            q_rp = rp_in[6 * kf + pa];
            if (b == pg0) { q_pp = P.pp[6 * kf + pa]; q_xp = P.xp[6 * kf + pa]; }
            if (pg0 + lane < pg1) {
#pragma unroll
                for (int c = 0; c < 6; ++c) q_acc[c] += part_in[(size_t)(pg0 + lane) * NPART + 3 + c];
            }
        }
    }
    // row factors, the tile's pose at the linearisation point, first record chunks
    RowRec rc;
    rc.w = 0;
    if (t == 0) rc = P.rowrec[row];
    const Pose Tlin = P.lin_pose[kf];
    const double pmask = P.pose_fixed[kf] ? 0.0 : 1.0;
    const int sbeg = P.ss_ptr[slice], send = rfix ? sbeg : P.ss_ptr[slice + 1];
    const int dbeg = P.sd_ptr[slice], dend = rfix ? dbeg : P.sd_ptr[slice + 1];
    SpringRec sr[U];
    DamperRec dr[U];
    auto load_springs = [&](int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            sr[q].other = REC_NONE;
            if (j < send) sr[q] = load_spring(P, j);
        }
    };
    auto load_dampers = [&](int idx) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int j = idx + 64 * q;
            dr[q].meta = REC_NONE;
            if (j < dend) dr[q] = load_damper(P, j);
        }
    };
    load_springs(sbeg + lane);
    load_dampers(dbeg + lane);

    // The flag is raised by workgroup 0 of the launch that finds the solve converged -- while other workgroups of that very launch
    // are still starting.  The decision has to be ONE per workgroup: with a per-thread test, waves that loaded the word before
    // the store went on and waves that loaded it after left, the block sums below lost the leavers' shares, and the tile ran an
    // update with garbage scalars behind the converged solve (seen as run-to-run differences of 1e-14 in chi2: tools/flake_probe.py,
    // NRS_CHECK_FUSED=1).
    if (__syncthreads_or(done_flag)) {
        if (pub_seq != 0 && blockIdx.x == 0 && tid == 0) publish_flags(P, pub_seq);      // (see k_pcg_update)
        return;
    }
    // wave 0: w_p = (H_pp + lambda) u_p + sum_l H_pl u_l of the tile's pose (independent of alpha, beta)
    double w_pose = 0, ua_pose = 0;
    if (wave == 0) {
        ua_pose = q_up[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) ua_pose = (pa == q) ? q_up[q] : ua_pose;
        if (it > 0) {
            double acc[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] = q_acc[c];
            for (int g = pg0 + lane + 64; g < pg1; g += 64) {
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] += part_in[(size_t)g * NPART + 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] = wave_sum(acc[c]);
            double hw = acc[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) hw = (pa == q) ? acc[q] : hw;
            w_pose = lam * ua_pose + hw;
#pragma unroll
            for (int c = 0; c < 6; ++c) w_pose += q_H[c] * q_up[c];
        }
    }
    stamp(1);
    // ================= coarse level: y = A_c^-1 Z^T r_new with r_new = r - alpha w - alpha beta s, i.e.
    // y = yR - alpha yW - alpha beta yS; the three products are formed before alpha, beta are known
    double* c_ts = dyn + 6 * (size_t)(P.tile_rows + P.max_halo);  // n_regblk x 9 tile sums
    double* c_yt = c_ts + 9 * (size_t)P.n_regblk;                  // n_regblk x 3 tile-level corrections
    double* c_v = c_yt + 3 * (size_t)P.n_regblk;                   // 16 vectors of CO_MAX: Rc Sc Wc, (yR yS yW) x up to 4 column parts, y
    if (coarse) {
        if (tid < P.n_regblk) {
#pragma unroll
            for (int c = 0; c < 9; ++c) c_ts[9 * tid + c] = ts9[c];
        }
        if (wave == 0 && lane < 6) {
            c_v[0 * CO_MAX + 3 * cG + lane] = q_rp;
            c_v[1 * CO_MAX + 3 * cG + lane] = q_sp;
            c_v[2 * CO_MAX + 3 * cG + lane] = w_pose;
        }
        __syncthreads();
        const int rb = ROW_ALIGN / P.tile_rows;
        if (tid < 9 * cG) {                                        // group sums in fixed order
            const int g = tid / 9, c = tid % 9;
            double sum = 0;
            for (int j = 0; j < rb; ++j) sum += c_ts[9 * (g * rb + j) + c];
            c_v[(c / 3) * CO_MAX + 3 * g + c % 3] = sum;
        }
        __syncthreads();
        if (co_part < co_parts) {                                  // three partial dot products over this thread's columns
            double yr = 0, ys = 0, yw = 0;
            if (co4) {
#pragma unroll
                for (int q = 0; q < CO_PF; ++q) {
                    const int c = co_c0 + q;
                    if (c < co_c1) { yr += c_m[q] * c_v[c]; ys += c_m[q] * c_v[CO_MAX + c]; yw += c_m[q] * c_v[2 * CO_MAX + c]; }
                }
            } else {
#pragma unroll 4
                for (int c = co_c0; c < co_c1; ++c) {
                    const double m = P.co_inv[(size_t)c * cn + co_r];
                    yr += m * c_v[c]; ys += m * c_v[CO_MAX + c]; yw += m * c_v[2 * CO_MAX + c];
                }
            }
            double* dst = c_v + (3 + 3 * co_part) * CO_MAX;         // part p writes vectors 3+3p .. 5+3p
            dst[co_r] = yr; dst[CO_MAX + co_r] = ys; dst[2 * CO_MAX + co_r] = yw;
        }
    }
    stamp(2);
    // ================= phase 2: scalars of iteration it-1 (k_pcg_update prologue)
    double alpha = 0, beta = 0;
    if (it > 0) {
        block_sum<3>(v, lds, lane, wave);
        const double gamma = v[0], delta = v[1] + v[2];
        const int ip = it - 1;                                     // PCG iteration these scalars belong to
        double* nslot = P.scal + ((ip & 1) ? SC_SLOT0 : SC_SLOT1);
        const double gamma0 = ip == 0 ? gamma : sc_gamma0;
        const bool bad = !isfinite(gamma) || !isfinite(delta);
        const bool conv = (gamma <= tol2 * gamma0) || bad || gamma == 0.0;
        if (conv) {
            if (blockIdx.x == 0 && tid == 0) {
                if (bad) P.flags[2] = 1;
                P.flags[1] = ip;
                __threadfence();
                P.flags[0] = 1;
                if (pub_seq != 0) publish_flags(P, pub_seq);
            }
            return;
        }
        beta = ip == 0 ? 0.0 : gamma / sc_slot0;
        alpha = ip == 0 ? gamma / delta : gamma / (delta - beta * gamma / sc_slot1);
        if (blockIdx.x == 0 && tid == 0) {
            nslot[0] = gamma;
            nslot[1] = alpha;
            if (ip == 0) P.scal[SC_GAMMA0] = gamma;
            P.flags[1] = ip + 1;
            if (gamma <= peek_tol2 * gamma0) {
                P.flags[3] = max(P.flags[3], (gamma <= 1e-6 * peek_tol2 * gamma0 ? 4 : gamma <= 1e-4 * peek_tol2 * gamma0 ? 3 : gamma <= 1e-2 * peek_tol2 * gamma0 ? 2 : 1));
                if (P.flags[4] == 0) P.flags[4] = ip + 1;
            }
            if (pub_seq != 0) publish_flags(P, pub_seq);
        }
    } else if (pub_seq != 0 && blockIdx.x == 0 && tid == 0) publish_flags(P, pub_seq);   // F(0) finishes no iteration
    if (coarse) {                                                  // (the reduction above was a barrier: yR, yS, yW are visible)
        if (tid < cn) {
            double yr = 0, ys = 0, yw = 0;
            for (int p = 0; p < co_parts; ++p) {
                yr += c_v[(3 + 3 * p) * CO_MAX + tid]; ys += c_v[(4 + 3 * p) * CO_MAX + tid]; yw += c_v[(5 + 3 * p) * CO_MAX + tid];
            }
            c_v[15 * CO_MAX + tid] = yr - alpha * yw - alpha * beta * ys;
        }
        if (tid < P.n_regblk) {                                    // tile level: y_t = B_t^-1 (R - alpha W - alpha beta S)_t, every tile
            double rc3[3], yt[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) rc3[k] = c_ts[9 * tid + k] - alpha * c_ts[9 * tid + 6 + k] - alpha * beta * c_ts[9 * tid + 3 + k];
            tile_level(bt_q, rc3, yt);
            c_yt[3 * tid] = yt[0]; c_yt[3 * tid + 1] = yt[1]; c_yt[3 * tid + 2] = yt[2];
        }
        __syncthreads();
    }
    stamp(3);
    const double* ycor = c_v + 15 * CO_MAX;
    // ================= phase 3: pose vector of this tile's pose (wave 0; every workgroup recomputes
    // it, the first workgroup of the pose also stores the pose part of the state)
    if (wave == 0) {
        const int a = pa;
        const int i = 6 * kf + a;
        const double ua = ua_pose;
        double unew = ua;
        if (it > 0) {
            const double w = w_pose;
            const double sN = w + beta * q_sp;
            const double rnew = q_rp - alpha * sN;
            unew = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) unew += q_Hi[c] * __shfl(rnew, c, 64);
            if (coarse && pmask != 0.0) unew += ycor[3 * cG + a];
            if (lane < 6 && b == pg0) {
                const double p = ua + beta * q_pp;
                P.pp[i] = p;
                P.xp[i] = q_xp + alpha * p;
                sp_out[i] = sN;
                rp_out[i] = rnew;
                up_out[i] = unew;
            }
        }
        if (lane < 6) s_up[lane] = unew;
    }
    // ================= phase 4: u of the tile (own rows: full update, stored; halo rows:
    // recomputed from r, s, w, M^-1, LDS only)
    double dot_ru = 0;                                             // r.u of this thread's own row (after the update)
    double sum9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                  // this tile's sums of r, s, w for the next launch
    if (it > 0) {
        if (own) {
            double rn[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double p = o_u[k] + beta * o_p[k];
                const double sN = o_w[k] + beta * o_s[k];
                P.pv[3 * orow + k] = p;
                s_out[3 * orow + k] = sN;
                P.xv[3 * orow + k] = o_x[k] + alpha * p;
                rn[k] = o_r[k] - alpha * sN;
                r_out[3 * orow + k] = rn[k];
                sum9[k] = rn[k]; sum9[3 + k] = sN;
            }
            double u0 = o_D[0] * rn[0] + o_D[1] * rn[1] + o_D[2] * rn[2];
            double u1 = o_D[1] * rn[0] + o_D[3] * rn[1] + o_D[4] * rn[2];
            double u2 = o_D[2] * rn[0] + o_D[4] * rn[1] + o_D[5] * rn[2];
            if (coarse && o_free) {                                // group level + tile level
                const int g = row0 / ROW_ALIGN;
                u0 += ycor[3 * g] + c_yt[3 * b]; u1 += ycor[3 * g + 1] + c_yt[3 * b + 1]; u2 += ycor[3 * g + 2] + c_yt[3 * b + 2];
            }
            P.uv3[3 * orow] = u0; P.uv3[3 * orow + 1] = u1; P.uv3[3 * orow + 2] = u2;
            lu[3 * tid] = u0; lu[3 * tid + 1] = u1; lu[3 * tid + 2] = u2;
            dot_ru = rn[0] * u0 + rn[1] * u1 + rn[2] * u2;
        }
        for (int i = tid; i < hn; i += BLK) {
            if (i != tid) {                                        // beyond the prefetched one (large halos only)
                const size_t r2 = (size_t)P.halo_rows[hb + i];
#pragma unroll
                for (int k = 0; k < 3; ++k) { h_w[k] = w_in[3 * r2 + k]; h_s[k] = s_in[3 * r2 + k]; h_r[k] = r_in[3 * r2 + k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) h_D[k] = P.Dinv[6 * r2 + k];
            }
            double rn[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) rn[k] = h_r[k] - alpha * (h_w[k] + beta * h_s[k]);
            double* dst = lu + 3 * (size_t)(P.tile_rows + i);
            double y0 = 0, y1 = 0, y2 = 0;
            if (coarse) {
                const size_t r2 = i == tid ? hrow : (size_t)P.halo_rows[hb + i];
                const bool fr = i == tid ? h_free : !(P.rflag[r2] & RF_FIXED);
                if (fr) {
                    const int sl = (int)(r2 / ROW_ALIGN);
                    const int th = i == tid ? th_h : (int)(r2 / (size_t)P.tile_rows);
                    y0 = ycor[3 * sl] + c_yt[3 * th]; y1 = ycor[3 * sl + 1] + c_yt[3 * th + 1]; y2 = ycor[3 * sl + 2] + c_yt[3 * th + 2];
                }
            }
            dst[0] = h_D[0] * rn[0] + h_D[1] * rn[1] + h_D[2] * rn[2] + y0;
            dst[1] = h_D[1] * rn[0] + h_D[3] * rn[1] + h_D[4] * rn[2] + y1;
            dst[2] = h_D[2] * rn[0] + h_D[4] * rn[1] + h_D[5] * rn[2] + y2;
        }
    }
    if (own) { lx[3 * tid] = x_own[0]; lx[3 * tid + 1] = x_own[1]; lx[3 * tid + 2] = x_own[2]; }
    for (int i = tid; i < hn; i += BLK) {
        if (i != tid) {
            const size_t r2 = (size_t)P.halo_rows[hb + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) x_h[k] = P.lin_xl[3 * r2 + k] + (P.X0 ? P.X0[3 * r2 + k] : 0.0);
        }
        double* dst = lx + 3 * (size_t)(P.tile_rows + i);
        dst[0] = x_h[0]; dst[1] = x_h[1]; dst[2] = x_h[2];
    }
    __syncthreads();
    stamp(4);
    // ================= phase 5: operator apply on the staged u (k_spmv_f)
    double a0 = 0, a1 = 0, a2 = 0;
    double part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double ul[3] = {lu[3 * self], lu[3 * self + 1], lu[3 * self + 2]};
    const double xs[3] = {lx[3 * self], lx[3 * self + 1], lx[3 * self + 2]};
    if (t == 0) {
        a0 = lam * ul[0]; a1 = lam * ul[1]; a2 = lam * ul[2];
        if (rc.w != 0.0) {
            double up[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) up[p] = s_up[p];
            row_factored(rc, Tlin, xs, ul, up, pmask, a0, a1, a2, part);
        }
    }
    for (int idx = sbeg + lane; idx < send; idx += 64 * U) {
        if (idx != sbeg + lane) load_springs(idx);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int o = sr[q].other;
            if (o == REC_NONE) continue;
            const double v0 = xs[0] - lx[3 * o], v1 = xs[1] - lx[3 * o + 1], v2 = xs[2] - lx[3 * o + 2];
            const double dot = sr[q].qc * (v0 * (ul[0] - lu[3 * o]) + v1 * (ul[1] - lu[3 * o + 1]) + v2 * (ul[2] - lu[3 * o + 2]));
            a0 += dot * v0; a1 += dot * v1; a2 += dot * v2;
        }
    }
    for (int idx = dbeg + lane; idx < dend; idx += 64 * U) {
        if (idx != dbeg + lane) load_dampers(idx);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (dr[q].meta == REC_NONE) continue;
            if (dr[q].meta & DM_UNARY) {
                a0 += dr[q].s * ul[0]; a1 += dr[q].s * ul[1]; a2 += dr[q].s * ul[2];
                continue;
            }
            // canonical order of the others: a_i += s ((u_i - u[o1]) - (u[o0] - u[o2])), absent vertices are zeros
            const uint16_t o[3] = {dr[q].o0, dr[q].o1, dr[q].o2};
            double v[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool has = o[k] != REC_NONE;
                v[k][0] = has ? lu[3 * o[k]] : 0.0; v[k][1] = has ? lu[3 * o[k] + 1] : 0.0; v[k][2] = has ? lu[3 * o[k] + 2] : 0.0;
            }
            const double sv = dr[q].s;
            a0 += sv * ((ul[0] - v[1][0]) - (v[0][0] - v[2][0]));
            a1 += sv * ((ul[1] - v[1][1]) - (v[0][1] - v[2][1]));
            a2 += sv * ((ul[2] - v[1][2]) - (v[0][2] - v[2][2]));
        }
    }
    stamp(5);
    a0 = sub_sum_t<T>(a0); a1 = sub_sum_t<T>(a1); a2 = sub_sum_t<T>(a2);
    if (t == 0) {
        w_out[3 * row] = a0; w_out[3 * row + 1] = a1; w_out[3 * row + 2] = a2;
        if (it == 0) {
            const double r0 = r_out[3 * row], r1 = r_out[3 * row + 1], r2 = r_out[3 * row + 2];
            part[0] = r0 * ul[0] + r1 * ul[1] + r2 * ul[2];
            sum9[0] = r0; sum9[1] = r1; sum9[2] = r2;            // launch 0: r = b, s = 0
        }
        part[1] = a0 * ul[0] + a1 * ul[1] + a2 * ul[2];
        sum9[6] = a0; sum9[7] = a1; sum9[8] = a2;
    }
    part[0] += dot_ru;
    if (!CO) {
        block_sum_store<9>(part, lds, tid, part_out + (size_t)b * NPART);
    } else {
        // one reduction for the nine PCG partials and the nine tile sums (one barrier instead of three)
        double* l18 = c_v;                                         // the coarse vectors are dead by now: 4 x 18 doubles
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double a = wave_sum(part[k]), c = wave_sum(sum9[k]);
            if (lane == 0) { l18[wave * 18 + k] = a; l18[wave * 18 + 9 + k] = c; }
        }
        __syncthreads();
        if (tid < 18) {
            const double tot = l18[tid] + l18[18 + tid] + l18[36 + tid] + l18[54 + tid];
            if (tid < 9) part_out[(size_t)b * NPART + tid] = tot;
            else ts_out[(size_t)(tid - 9) * P.n_regblk + b] = tot;
        }
    }
    stamp(6);
}

// =====================================================================================
// trial state = state (+) x ;  partial of computeScale: sum_j x_j (lambda x_j + b_j)
// (levenberg.cpp:167-174; LandmarkVertex::oplusImpl landmark_vertex.cc:40-43)
// =====================================================================================
__global__ __launch_bounds__(BLK) void k_apply(Dev P, double lam, const Pose* __restrict__ pose_in,
                                               const double* __restrict__ xl_in, Pose* pose_out, double* xl_out) {
    __shared__ double lds[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = (P.sh_vb0 + blockIdx.x) * BLK + tid;             // row (own range); poses below: every rank, all of them
    double sc[1] = {0};
    if (i < P.n_rows) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t j = 3 * (size_t)i + k;
            const double x = P.xv[j];
            xl_out[j] = xl_in[j] + x;
            sc[0] += x * (lam * x + P.bl[j]);
        }
    }
    if ((int)(blockIdx.x * BLK + tid) < P.K) {
        const int i = blockIdx.x * BLK + tid;
        Pose Tcw = pose_in[i];
        if (!P.pose_fixed[i]) {
            double upd[6];
            for (int a = 0; a < 6; ++a) {
                upd[a] = P.xp[6 * i + a];
                if (P.sh_lead) sc[0] += upd[a] * (lam * upd[a] + P.bp[6 * i + a]);
            }
            pose_oplus(Tcw, upd);
        }
        pose_out[i] = Tcw;
    }
    block_sum<1>(sc, lds, lane, wave);
    if (tid == 0) P.part_apply[P.sh_vb0 + blockIdx.x] = sc[0];
}

// k_apply and k_reproj<false> in one launch for single-pose problems (a2's per-frame engines, where an LM trial is a handful of
// launch-bound kernels behind the direct solve): the trial state and, from it, the chi2 of the reprojection edges, with the
// expressions -- and so the bits -- of the two kernels it stands for.  One workgroup per row group, as both of them.
__global__ __launch_bounds__(BLK) void k_apply_reproj(Dev P, double lam, const Pose* __restrict__ pose_in, const double* __restrict__ xl_in,
                                                      Pose* pose_out, double* xl_out) {
    __shared__ double lds[4];
    const int gi = xcd_tile(blockIdx.x, P.sh_ng);
    if (gi >= P.sh_ng) return;
    const int g = P.sh_g0 + gi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = g * ROW_ALIGN + tid;
    Pose Tcw = pose_in[0];
    const bool pfix = P.pose_fixed[0] != 0;
    double sc[1] = {0};
    {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t j = 3 * (size_t)row + k;
            const double x = P.xv[j];
            xl_out[j] = xl_in[j] + x;
            sc[0] += x * (lam * x + P.bl[j]);
        }
    }
    if (!pfix) {                                                   // (every workgroup: the pose is six numbers)
        double upd[6];
        for (int a = 0; a < 6; ++a) {
            upd[a] = P.xp[a];
            if (g == 0 && tid == 0) sc[0] += upd[a] * (lam * upd[a] + P.bp[a]);
        }
        pose_oplus(Tcw, upd);
    }
    if (g == 0 && tid == 0) pose_out[0] = Tcw;
    block_sum<1>(sc, lds, lane, wave);
    if (tid == 0) P.part_apply[g] = sc[0];
    __syncthreads();                                               // (lds is reused)
    double R[9];
    quat_to_R(Tcw.q, R);
    const int rf = P.rflag[row];
    const bool rfix = (rf & RF_FIXED) != 0;
    const bool active = (rf & RF_OBS) && (rf & RF_REPROJ_ACTIVE) && !(pfix && rfix);
    double c1[1] = {0};
    if (active) {
        double x0 = xl_in[3 * row] + P.xv[3 * row], x1 = xl_in[3 * row + 1] + P.xv[3 * row + 1], x2 = xl_in[3 * row + 2] + P.xv[3 * row + 2];
        if (P.X0) { x0 += P.X0[3 * row]; x1 += P.X0[3 * row + 1]; x2 += P.X0[3 * row + 2]; }
        const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
        const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
        const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
        float u, v;
        project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
        const double r0 = (double)P.uv[2 * row] - (double)u, r1 = (double)P.uv[2 * row + 1] - (double)v;
        double rho0, rho1;
        huber(P.info_reproj * (r0 * r0 + r1 * r1), P.delta_reproj, rho0, rho1);
        c1[0] = rho0;
    }
    block_sum<1>(c1, lds, lane, wave);
    if (tid == 0) P.part_rchi[g] = c1[0];
}

// sharded download: a rank's own rows, zeros elsewhere (summed over the ranks afterwards)
__global__ void k_mask_rows(Dev P, const double* __restrict__ in, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= 3 * (size_t)P.n_rows) return;
    const size_t lo = 3 * (size_t)P.sh_vb0 * BLK, hi = 3 * (size_t)(P.sh_vb0 + P.sh_nvb) * BLK;
    out[i] = (i >= lo && i < hi) ? in[i] : 0.0;
}

// =====================================================================================
// edge taps (edge-parallel, not on the timed path): residuals of every edge at a given state
// =====================================================================================
__global__ void k_tap_residuals(Dev P, const Pose* poses, const double* xl, const uint8_t* rflag, const float* uvs, const int* vrow,   // (rflag / uvs: the engine's, or full-length copies on a rank that holds its own rows only)
                                const int* sp_ij, const float* sp_d0, const int* dm_idx, const float* dm_w,
                                double* r_reproj, double* r_spring, double* r_damper) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.M) {
        const int row = vrow[i];
        r_reproj[2 * i] = r_reproj[2 * i + 1] = 0;
        if (rflag[row] & RF_OBS) {
            const Pose Tcw = poses[P.grp_pose[row / ROW_ALIGN]];
            double R[9];
            quat_to_R(Tcw.q, R);
            double x0 = xl[3 * row], x1 = xl[3 * row + 1], x2 = xl[3 * row + 2];
            if (P.X0) { x0 += P.X0[3 * row]; x1 += P.X0[3 * row + 1]; x2 += P.X0[3 * row + 2]; }
            const double px = R[0] * x0 + R[1] * x1 + R[2] * x2 + Tcw.t[0];
            const double py = R[3] * x0 + R[4] * x1 + R[5] * x2 + Tcw.t[1];
            const double pz = R[6] * x0 + R[7] * x1 + R[8] * x2 + Tcw.t[2];
            float u, v;
            project_f32(P.cam, (float)px, (float)py, (float)pz, u, v);
            r_reproj[2 * i] = (double)uvs[2 * row] - (double)u;
            r_reproj[2 * i + 1] = (double)uvs[2 * row + 1] - (double)v;
        }
    }
    if (i < P.n_sp) {
        const int a = vrow[sp_ij[2 * i]], b = vrow[sp_ij[2 * i + 1]];
        double v[3];
        for (int k = 0; k < 3; ++k) {
            v[k] = xl[3 * a + k] - xl[3 * b + k];
            if (P.X0) v[k] = (xl[3 * a + k] + P.X0[3 * a + k]) - (xl[3 * b + k] + P.X0[3 * b + k]);
        }
        const double d = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), d0 = (double)sp_d0[i];
        r_spring[i] = P.k_spring * (d - d0) / d0;
    }
    if (i < P.n_dm) {
        const double w = (double)dm_w[i];
        for (int k = 0; k < 3; ++k) {
            double s = 0;
            for (int role = 0; role < 4; ++role) {
                const int v = dm_idx[4 * i + role];
                if (v >= 0) s += damper_sign(role) * xl[3 * vrow[v] + k];
            }
            r_damper[3 * i + k] = w * s;
        }
    }
}

}  // namespace nrs
