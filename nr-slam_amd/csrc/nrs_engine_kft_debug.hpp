// Parity tap of the keyframe-block factorisation (included at the end of nrs_engine.hip: it uses the engine's evaluation).
#pragma once

namespace nrs {

// Parity tap (include/nrs.h nrs_debug_kft): linearise at the current estimate, assemble / factorise the keyframe blocks at `lam`, then
//   what 0: info[0..6) = {on, K, ld, nb, m, factor bytes >> 20}, then nf_k, np_k per keyframe          (out_i, 6 + 2 K ints)
//   what 1: the ASSEMBLED diagonal block A_k (ld x ld, compact order: node copies by row, then the pose)  (out_d)
//   what 2: the coupling T_k as a dense ld x ld matrix (rows: keyframe k, columns: keyframe k + 1)       (out_d)
//   what 3: u = M^-1 r for r given in solver order (6 K pose entries, then 3 per vertex); u in the same order (in_d -> out_d)
//   what 4: per vertex its (keyframe, compact node) pair, -1 for a fixed vertex                          (out_i, 2 M ints)
int engine_kft_debug(nrs_ctx* c, Engine* e, double lam, int what, int k, const double* in_d, double* out_d, int32_t* out_i) {
    Dev& d = e->d;
    KftHost* H = e->kft;
    if (what == 0) {
        out_i[0] = H && H->on ? 1 : 0;
        if (!out_i[0]) return NRS_OK;
        out_i[1] = H->d.K; out_i[2] = H->d.ld; out_i[3] = H->d.nb; out_i[4] = H->d.m; out_i[5] = (int32_t)(H->bytes >> 20);
        NRS_HIP(c, hipMemcpy(out_i + 6, H->d.kf_nf, 4 * (size_t)H->d.K, hipMemcpyDeviceToHost));
        NRS_HIP(c, hipMemcpy(out_i + 6 + H->d.K, H->d.kf_np, 4 * (size_t)H->d.K, hipMemcpyDeviceToHost));
        return NRS_OK;
    }
    if (!H || !H->on) return c->fail(NRS_ERR_STATE, "no keyframe-block factorisation on this window");
    const KftDev& F = H->d;
    const size_t n2 = (size_t)F.ld * F.ld;
    if (what == 4) {
        std::vector<int> rc((size_t)d.n_rows);
        NRS_HIP(c, hipMemcpy(rc.data(), F.row_ci, 4 * rc.size(), hipMemcpyDeviceToHost));
        std::vector<int> gp((size_t)d.n_groups);
        NRS_HIP(c, hipMemcpy(gp.data(), d.grp_pose, 4 * gp.size(), hipMemcpyDeviceToHost));
        for (int v = 0; v < d.M; ++v) { out_i[2 * v] = gp[e->vrow[v] / ROW_ALIGN]; out_i[2 * v + 1] = rc[e->vrow[v]]; }
        return NRS_OK;
    }
    NRS_TRY(evaluate<true>(c, e, e->cur));
    NRS_TRY(read_scalars(c, e));
    if (what == 1 || what == 2) {
        if (k < 0 || k >= F.K) return c->fail(NRS_ERR_INVALID, "keyframe out of range");
        hipLaunchKernelGGL(k_kft_clear, dim3((unsigned)((n2 / 2 + 255) / 256), F.K), dim3(256), 0, c->stream, F);
        hipLaunchKernelGGL(k_kft_diag, dim3(d.n_rows / SK_RPB), dim3(BLK), 0, c->stream, d, F, lam);
        hipLaunchKernelGGL(k_kft_pose, dim3((36 * F.K + 255) / 256), dim3(256), 0, c->stream, d, F, lam);
        if (F.n_pp) hipLaunchKernelGGL(k_kft_pairs, dim3((unsigned)(((size_t)F.n_pp * KFT_PL + 255) / 256)), dim3(256), 0, c->stream, d, F);
        if (F.n_tp) hipLaunchKernelGGL(k_kft_tvals, dim3((F.n_tp + 255) / 256), dim3(256), 0, c->stream, d, F);
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        if (what == 1) { NRS_HIP(c, hipMemcpy(out_d, F.A + (size_t)k * n2, 8 * n2, hipMemcpyDeviceToHost)); return NRS_OK; }
        std::vector<int> ptr((size_t)F.K * (F.nfm + 1)), from((size_t)F.n_tp + 1), tp((size_t)F.n_tp + 1);
        std::vector<double> tv((size_t)F.n_tp + 1);
        NRS_HIP(c, hipMemcpy(ptr.data(), F.cl_ptr[1], 4 * ptr.size(), hipMemcpyDeviceToHost));
        NRS_HIP(c, hipMemcpy(from.data(), F.cl_from[1], 4 * (size_t)F.n_tp, hipMemcpyDeviceToHost));
        NRS_HIP(c, hipMemcpy(tp.data(), F.cl_tp[1], 4 * (size_t)F.n_tp, hipMemcpyDeviceToHost));
        NRS_HIP(c, hipMemcpy(tv.data(), F.tp_val, 8 * (size_t)F.n_tp, hipMemcpyDeviceToHost));
        std::fill(out_d, out_d + n2, 0.0);
        for (int a = 0; a < F.nfm; ++a)                            // dir 1 at keyframe k by its node a: the nodes b of keyframe k + 1
            for (int q = ptr[(size_t)k * (F.nfm + 1) + a]; q < ptr[(size_t)k * (F.nfm + 1) + a + 1]; ++q)
                for (int cc = 0; cc < 3; ++cc) out_d[(size_t)(3 * a + cc) * F.ld + 3 * from[q] + cc] = tv[tp[q]];
        return NRS_OK;
    }
    if (what == 3) {
        NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
        NRS_TRY(kft_factor(c, e, H, lam));
        std::vector<double> rv(3 * (size_t)d.n_rows, 0.0);
        for (int v = 0; v < d.M; ++v) for (int a = 0; a < 3; ++a) rv[3 * (size_t)e->vrow[v] + a] = in_d[6 * (size_t)d.K + 3 * (size_t)v + a];
        NRS_HIP(c, hipMemcpyAsync(d.rv, rv.data(), 8 * rv.size(), hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemcpyAsync(d.rp, in_d, 8 * 6 * (size_t)d.K, hipMemcpyHostToDevice, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.uv3, 0, 8 * 3 * (size_t)d.n_rows, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.up, 0, 8 * 6 * (size_t)d.K, c->stream));
        NRS_TRY(kft_apply(c, H, d.rv, d.rp, d.uv3, d.up, d.flags));
        NRS_HIP(c, hipGetLastError());
        NRS_HIP(c, hipMemcpyAsync(rv.data(), d.uv3, 8 * rv.size(), hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipMemcpyAsync(out_d, d.up, 8 * 6 * (size_t)d.K, hipMemcpyDeviceToHost, c->stream));
        int fl[8];
        NRS_HIP(c, hipMemcpyAsync(fl, d.flags, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        for (int v = 0; v < d.M; ++v) for (int a = 0; a < 3; ++a) out_d[6 * (size_t)d.K + 3 * (size_t)v + a] = rv[3 * (size_t)e->vrow[v] + a];
        if (fl[2]) return c->fail(NRS_ERR_NUMERIC, "keyframe-block factorisation: a pivot that is not positive");
        return NRS_OK;
    }
    return c->fail(NRS_ERR_INVALID, "nrs_debug_kft: unknown request");
}

}  // namespace nrs
