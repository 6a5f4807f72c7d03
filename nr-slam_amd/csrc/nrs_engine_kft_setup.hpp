// Set-up of the keyframe-block factorisation (nrs_engine_kft.hpp): host lists, one device buffer.  Part of nrs_engine.hip.
#pragma once

namespace nrs {

// ---- set-up (once per window): compact indices, the pair / coupling lists in a fixed order, one device buffer
struct KftEnt { uint64_t key; uint32_t src; double w; };
static int kft_setup(nrs_ctx* c, Engine* e, const EngineSpec& s, const std::vector<int>& pose_grp_ptr) {
    const Dev& d = e->d;
    const int K = d.K;
    const bool tm = c->env("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tm) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] kft_setup %-18s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (K < 1 || K > 255 || (e->sp_pos.empty() && s.n_sp > 0)) return NRS_OK;
    std::vector<int> kf_nf(K, 0), kf_np(K, 0), row_ci((size_t)d.n_rows, -1);
    int nf_max = 0;
    for (int k = 0; k < K; ++k) {
        int n = 0;
        for (int r = pose_grp_ptr[k] * ROW_ALIGN; r < pose_grp_ptr[k + 1] * ROW_ALIGN; ++r)
            if (!(e->h_rflag[r] & RF_FIXED)) row_ci[r] = n++;
        kf_nf[k] = n;
        kf_np[k] = (s.pose_fixed && s.pose_fixed[k]) ? 0 : 6;
        nf_max = std::max(nf_max, n);
    }
    if (nf_max >= 4096 || nf_max < 1) return NRS_OK;
    const int ld = ((3 * nf_max + 6 + KFT_B - 1) / KFT_B) * KFT_B, nb = ld / KFT_B, nfm = ld / 3;
    const size_t n2 = (size_t)ld * ld;
    if (c->opt.embedded_solver == 0) {
        // automatic choice by a cost model of the two solvers, fitted on tools/kft_probe.py runs (10 .. 40 keyframes, 100 .. 650 nodes per keyframe,
        // profiles/r06_kft_crossover.txt): the factorisation is ceil(K / 2) dependent inversions of nb + 1 launches, (16.5 + 0.72 nb) us a launch
        // (the panel workgroup's path: sweep, look-ahead, tile I/O), + 0.05 K ms per trial (assembly, Schur updates,
        // one pass over the chains); the block-Jacobi PCG took 14 ms (K / 20)^0.45 per trial whatever the node count.  Measured on 20 keyframes:
        // 7.8 x the PCG's rate at 100 nodes per keyframe, 4.8 x at 200, 3.3 x at 300, 2.3 x at 400, 1.6 x at 458 (C2: 118 against 73 LM
        // iterations / s); 2.3 x at 458 x 10 keyframes.  A heuristic of this scene family: nrs_options.embedded_solver = 1 / 2 decide.
        // Beyond ~550 nodes per keyframe the launch is bound by its nb^2 tiles, (9 + 0.045 nb^2) us, and the PCG's trial grows by 16 us a node:
        // 600 nodes x 20 keyframes 91.4 against 62.6 LM iterations / s, 700 nodes 60.4 against 57.8 (the crossover), 600 x 10 keyframes 165 against 134.
        const double launch_us = std::max(16.5 + 0.72 * nb, 9.0 + 0.045 * nb * nb);
        const double kft_ms = ((K + 1) / 2) * (nb + 1.0) * launch_us * 1e-3 + 0.05 * K, pcg_ms = (14.0 + 0.016 * std::max(0, nf_max - 500)) * std::pow(K / 20.0, 0.45);
        if (kft_ms > pcg_ms) return NRS_OK;
    }
    if ((size_t)K * n2 * sizeof(double) > ((size_t)6 << 30)) return NRS_OK;       // (the factor would not be worth its memory: the PCG stays block-Jacobi)
    std::vector<int> kf_row((size_t)K * nfm, -1);
    for (int k = 0; k < K; ++k)
        for (int r = pose_grp_ptr[k] * ROW_ALIGN; r < pose_grp_ptr[k + 1] * ROW_ALIGN; ++r)
            if (row_ci[r] >= 0) kf_row[(size_t)k * nfm + row_ci[r]] = r;
    auto vk = [&](int v) { return s.lm_pose[v]; };
    auto vc = [&](int v) { return row_ci[e->vrow[v]]; };
    // ---- same-keyframe pairs
    std::vector<KftEnt> pe;
    pe.reserve((size_t)s.n_sp + 2 * (size_t)s.n_dm);
    auto add_pair = [&](int k, int a, int b, uint32_t src, double w) {
        if (a < 0 || b < 0 || a == b) return;
        const int hi = std::max(a, b), lo = std::min(a, b);
        pe.push_back(KftEnt{((uint64_t)k << 24) | ((uint64_t)hi << 12) | (uint64_t)lo, src, w});
    };
    for (int q = 0; q < s.n_sp; ++q) {
        const int i = s.sp_ij[2 * (size_t)q], j = s.sp_ij[2 * (size_t)q + 1];
        if (vk(i) != vk(j)) return NRS_OK;                         // (a spring across keyframes: not a BA window's)
        const int pos = vc(i) >= 0 ? e->sp_pos[2 * (size_t)q] : e->sp_pos[2 * (size_t)q + 1];
        if (pos >= 0) add_pair(vk(i), vc(i), vc(j), (0u << 30) | (uint32_t)pos, 0.0);
    }
    std::vector<KftEnt> te;
    te.reserve(4 * (size_t)s.n_dm);
    for (int q = 0; q < s.n_dm; ++q) {
        const int* v = s.dm_idx + 4 * (size_t)q;                   // (1c, 2c, 1n, 2n), -1: absent
        int kk[4], cc[4], pos = -1;
        for (int r = 0; r < 4; ++r) {
            kk[r] = v[r] >= 0 ? vk(v[r]) : -1; cc[r] = v[r] >= 0 ? vc(v[r]) : -1;
            if (pos < 0 && cc[r] >= 0) pos = e->dm_pos[4 * (size_t)q + r];
        }
        if (pos < 0) continue;
        // J = (-w, +w, +w, -w) I on (1c, 2c, 1n, 2n): block (p, q) = sg_p sg_q s I
        if (v[0] >= 0 && v[1] >= 0) { if (kk[0] != kk[1]) return NRS_OK; add_pair(kk[0], cc[0], cc[1], (1u << 30) | (uint32_t)pos, 0.0); }
        if (v[2] >= 0 && v[3] >= 0) { if (kk[2] != kk[3]) return NRS_OK; add_pair(kk[2], cc[2], cc[3], (1u << 30) | (uint32_t)pos, 0.0); }
        static const int cur[2] = {0, 1}, nxt[2] = {2, 3};
        for (int x = 0; x < 2; ++x)
            for (int y = 0; y < 2; ++y) {
                const int a = cur[x], b = nxt[y];
                if (v[a] < 0 || v[b] < 0 || cc[a] < 0 || cc[b] < 0) continue;
                if (kk[b] != kk[a] + 1) return NRS_OK;             // (a damper that does not join consecutive keyframes)
                const bool minus = (x == y);                       // (1c,1n), (2c,2n): - s ; (1c,2n), (2c,1n): + s
                te.push_back(KftEnt{((uint64_t)kk[a] << 24) | ((uint64_t)cc[a] << 12) | (uint64_t)cc[b], (minus ? 0x80000000u : 0u) | (uint32_t)pos, 0.0});
            }
    }
    mark("edge entries");
    // The skinned observations' pairs (55 per observation: 5 M at C2) are never materialised: per keyframe -- in parallel -- a counting pass
    // over (hi, lo) < nf^2, then a placing pass straight into the keyframe's sorted source / weight arrays.  Order inside a pair's list:
    // springs and dampers in edge order, then the observations in caller order (the summation order of k_kft_pairs).
    std::vector<size_t> small_ptr(K + 1, 0);
    std::vector<KftEnt> small(pe.size());
    {
        for (const KftEnt& x : pe) small_ptr[(x.key >> 24) + 1]++;
        for (int k = 0; k < K; ++k) small_ptr[k + 1] += small_ptr[k];
        std::vector<size_t> fill(small_ptr.begin(), small_ptr.end() - 1);
        for (const KftEnt& x : pe) small[fill[x.key >> 24]++] = x;
    }
    std::vector<int> sk_ptr(K + 1, 0), sk_of((size_t)s.n_skin);
    for (int i = 0; i < s.n_skin; ++i) sk_ptr[s.sk_pose[i] + 1]++;
    for (int k = 0; k < K; ++k) sk_ptr[k + 1] += sk_ptr[k];
    {
        std::vector<int> fill(sk_ptr.begin(), sk_ptr.end() - 1);
        for (int i = 0; i < s.n_skin; ++i) sk_of[fill[s.sk_pose[i]]++] = i;
    }
    struct KfLists { std::vector<uint32_t> pp_id, pe_src; std::vector<int> pp_ptr; std::vector<double> pe_w; };
    std::vector<KfLists> kl(K);
    {
        const int nt = host_threads(c, (size_t)s.n_sp + 55 * (size_t)s.n_skin);
        parallel_for(std::min(nt, K), [&](int ti, int n) {
            int64_t ka, kb;
            chunk(K, ti, n, ka, kb);
            std::vector<int> cnt;
            std::vector<int> cn((size_t)SK_MAX);
            for (int k = (int)ka; k < (int)kb; ++k) {
                const size_t nk = (size_t)kf_nf[k], nkey = nk * nk;
                cnt.assign(nkey + 1, 0);
                auto each_skin_pair = [&](auto&& fn) {
                    for (int q = sk_ptr[k]; q < sk_ptr[k + 1]; ++q) {
                        const int i = sk_of[q];
                        for (int a2 = 0; a2 < SK_MAX; ++a2) { const int v = s.sk_node[SK_MAX * (size_t)i + a2]; cn[a2] = v >= 0 ? vc(v) : -1; }
                        for (int a2 = 0; a2 < SK_MAX; ++a2)
                            for (int b2 = a2 + 1; b2 < SK_MAX; ++b2) {
                                if (cn[a2] < 0 || cn[b2] < 0 || cn[a2] == cn[b2]) continue;
                                const int hi = std::max(cn[a2], cn[b2]), lo = std::min(cn[a2], cn[b2]);
                                fn((size_t)hi * nk + lo, i, a2, b2);
                            }
                    }
                };
                for (size_t q = small_ptr[k]; q < small_ptr[k + 1]; ++q) cnt[((small[q].key >> 12) & 0xFFF) * nk + (small[q].key & 0xFFF) + 1]++;
                each_skin_pair([&](size_t key, int, int, int) { cnt[key + 1]++; });
                KfLists& o = kl[k];
                for (size_t key = 0; key < nkey; ++key) {
                    if (cnt[key + 1] > 0) { o.pp_id.push_back(((uint32_t)k << 24) | ((uint32_t)(key / nk) << 12) | (uint32_t)(key % nk)); o.pp_ptr.push_back(cnt[key]); }
                    cnt[key + 1] += cnt[key];
                }
                const size_t ne = (size_t)cnt[nkey];
                o.pp_ptr.push_back((int)ne);
                o.pe_src.resize(ne); o.pe_w.resize(ne);
                for (size_t q = small_ptr[k]; q < small_ptr[k + 1]; ++q) {
                    const int at = cnt[((small[q].key >> 12) & 0xFFF) * nk + (small[q].key & 0xFFF)]++;
                    o.pe_src[at] = small[q].src; o.pe_w[at] = small[q].w;
                }
                each_skin_pair([&](size_t key, int i, int a2, int b2) {
                    const int at = cnt[key]++;
                    o.pe_src[at] = (2u << 30) | (uint32_t)e->sk_slot[i];
                    o.pe_w[at] = s.sk_om[SK_MAX * (size_t)i + a2] * s.sk_om[SK_MAX * (size_t)i + b2];
                });
            }
        });
    }
    mark("pair lists");
    auto by_key = [](const KftEnt& a, const KftEnt& b) { return a.key < b.key; };
    std::stable_sort(te.begin(), te.end(), by_key);
    std::vector<uint32_t> pp_id, te_src(te.size());
    std::vector<int> pp_ptr, tp_ptr;
    std::vector<size_t> ent_base(K + 1, 0);                          // (the entries themselves go up keyframe by keyframe: no concatenated host copy)
    {
        size_t n_pairs = 0;
        for (int k = 0; k < K; ++k) { n_pairs += kl[k].pp_id.size(); ent_base[k + 1] = ent_base[k] + kl[k].pe_src.size(); }
        if (ent_base[K] >= ((size_t)1 << 31)) return NRS_OK;
        pp_id.reserve(n_pairs); pp_ptr.reserve(n_pairs + 1);
        for (int k = 0; k < K; ++k) {
            pp_id.insert(pp_id.end(), kl[k].pp_id.begin(), kl[k].pp_id.end());
            for (size_t q = 0; q + 1 < kl[k].pp_ptr.size(); ++q) pp_ptr.push_back((int)ent_base[k] + kl[k].pp_ptr[q]);
        }
        pp_ptr.push_back((int)ent_base[K]);
    }
    const size_t pe_size = ent_base[K];
    std::vector<uint64_t> tp_key;
    for (size_t i = 0; i < te.size(); ++i) {
        if (i == 0 || te[i].key != te[i - 1].key) { tp_key.push_back(te[i].key); tp_ptr.push_back((int)i); }
        te_src[i] = te[i].src;
    }
    tp_ptr.push_back((int)te.size());
    const size_t n_pp = pp_id.size(), n_tp = tp_key.size();
    // coupling lists by the 'to' node: dir 0 at keyframe k + 1 by b (from a of k), dir 1 at keyframe k by a (from b of k + 1)
    std::vector<int> cl_ptr[2], cl_from[2], cl_tp[2];
    for (int dir = 0; dir < 2; ++dir) {
        cl_ptr[dir].assign((size_t)K * (nfm + 1), 0);
        std::vector<int> cnt((size_t)K * nfm, 0);
        for (size_t i = 0; i < n_tp; ++i) {
            const int k = (int)(tp_key[i] >> 24), a = (int)((tp_key[i] >> 12) & 0xFFF), b = (int)(tp_key[i] & 0xFFF);
            cnt[dir == 0 ? (size_t)(k + 1) * nfm + b : (size_t)k * nfm + a]++;
        }
        int run = 0;
        for (int k = 0; k < K; ++k) {
            for (int n = 0; n < nfm; ++n) { cl_ptr[dir][(size_t)k * (nfm + 1) + n] = run; run += cnt[(size_t)k * nfm + n]; }
            cl_ptr[dir][(size_t)k * (nfm + 1) + nfm] = run;
        }
        cl_from[dir].assign(n_tp + 1, 0); cl_tp[dir].assign(n_tp + 1, 0);
        std::vector<int> fill((size_t)K * nfm);
        for (int k = 0; k < K; ++k) for (int n = 0; n < nfm; ++n) fill[(size_t)k * nfm + n] = cl_ptr[dir][(size_t)k * (nfm + 1) + n];
        for (size_t i = 0; i < n_tp; ++i) {                        // (tp order = ascending (k, a, b): every list in a fixed order)
            const int k = (int)(tp_key[i] >> 24), a = (int)((tp_key[i] >> 12) & 0xFFF), b = (int)(tp_key[i] & 0xFFF);
            const int q = fill[dir == 0 ? (size_t)(k + 1) * nfm + b : (size_t)k * nfm + a]++;
            cl_from[dir][q] = dir == 0 ? a : b; cl_tp[dir][q] = (int)i;
        }
    }
    mark("coupling lists");
    // ---- one device buffer
    auto al = [](size_t b2) { return (b2 + 255) & ~(size_t)255; };
    const size_t tile = (size_t)KFT_B * KFT_B;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_A = take(8 * (size_t)K * n2), o_YT = take(8 * 2 * n2), o_Bb = take(8 * 4 * nb * tile), o_Cb = take(8 * 4 * nb * tile), o_Pv = take(8 * 4 * tile),
                 o_z = take(8 * (size_t)K * ld), o_xs = take(8 * (size_t)K * ld), o_vb = take(8 * 2 * (size_t)ld), o_nf = take(4 * (size_t)K), o_np = take(4 * (size_t)K), o_kr = take(4 * kf_row.size()),
                 o_rc = take(4 * row_ci.size()), o_ppid = take(4 * (n_pp + 1)), o_ppp = take(4 * (n_pp + 1)), o_pes = take(4 * (pe_size + 1)), o_pew = take(8 * (pe_size + 1)),
                 o_tpp = take(4 * (n_tp + 1)), o_tes = take(4 * (te.size() + 1)), o_tpv = take(8 * (n_tp + 1)),
                 o_cp0 = take(4 * cl_ptr[0].size()), o_cp1 = take(4 * cl_ptr[1].size()), o_cf0 = take(4 * (n_tp + 1)), o_cf1 = take(4 * (n_tp + 1)),
                 o_ct0 = take(4 * (n_tp + 1)), o_ct1 = take(4 * (n_tp + 1)), o_cv0 = take(8 * (n_tp + 1)), o_cv1 = take(8 * (n_tp + 1));
    if (c->ensure(c->dba_kft, off) != NRS_OK) return NRS_OK;      // (no memory for the factor: block-Jacobi PCG)
    mark("device buffer");
    char* base = c->dba_kft.as<char>();
    auto up = [&](size_t o, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess; };
    NRS_HIP(c, up(o_nf, kf_nf.data(), 4 * (size_t)K)); NRS_HIP(c, up(o_np, kf_np.data(), 4 * (size_t)K)); NRS_HIP(c, up(o_kr, kf_row.data(), 4 * kf_row.size()));
    NRS_HIP(c, up(o_rc, row_ci.data(), 4 * row_ci.size())); NRS_HIP(c, up(o_ppid, pp_id.data(), 4 * n_pp)); NRS_HIP(c, up(o_ppp, pp_ptr.data(), 4 * (n_pp + 1)));
    for (int k = 0; k < K; ++k) { NRS_HIP(c, up(o_pes + 4 * ent_base[k], kl[k].pe_src.data(), 4 * kl[k].pe_src.size())); NRS_HIP(c, up(o_pew + 8 * ent_base[k], kl[k].pe_w.data(), 8 * kl[k].pe_w.size())); }
    NRS_HIP(c, up(o_tpp, tp_ptr.data(), 4 * (n_tp + 1))); NRS_HIP(c, up(o_tes, te_src.data(), 4 * te.size()));
    NRS_HIP(c, up(o_cp0, cl_ptr[0].data(), 4 * cl_ptr[0].size())); NRS_HIP(c, up(o_cp1, cl_ptr[1].data(), 4 * cl_ptr[1].size()));
    NRS_HIP(c, up(o_cf0, cl_from[0].data(), 4 * n_tp)); NRS_HIP(c, up(o_cf1, cl_from[1].data(), 4 * n_tp));
    NRS_HIP(c, up(o_ct0, cl_tp[0].data(), 4 * n_tp)); NRS_HIP(c, up(o_ct1, cl_tp[1].data(), 4 * n_tp));
    NRS_HIP(c, hipMemsetAsync(base + o_z, 0, o_nf - o_z, c->stream));     // (z and xs)
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    mark("uploads");
    KftHost* H = new (std::nothrow) KftHost();
    if (!H) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    KftDev& F = H->d;
    memset(&F, 0, sizeof(F));
    F.K = K; F.ld = ld; F.nb = nb; F.m = K / 2; F.nfm = nfm;
    F.A = reinterpret_cast<double*>(base + o_A); F.YT = reinterpret_cast<double*>(base + o_YT);
    F.Bb = reinterpret_cast<double*>(base + o_Bb); F.Cb = reinterpret_cast<double*>(base + o_Cb); F.Pv = reinterpret_cast<double*>(base + o_Pv);
    F.z = reinterpret_cast<double*>(base + o_z); F.xs = reinterpret_cast<double*>(base + o_xs); F.vb = reinterpret_cast<double*>(base + o_vb);
    F.kf_nf = reinterpret_cast<const int*>(base + o_nf); F.kf_np = reinterpret_cast<const int*>(base + o_np);
    F.kf_row = reinterpret_cast<const int*>(base + o_kr); F.row_ci = reinterpret_cast<const int*>(base + o_rc);
    F.n_pp = (int)n_pp; F.pp_id = reinterpret_cast<const uint32_t*>(base + o_ppid); F.pp_ptr = reinterpret_cast<const int*>(base + o_ppp);
    F.pe_src = reinterpret_cast<const uint32_t*>(base + o_pes); F.pe_w = reinterpret_cast<const double*>(base + o_pew);
    F.n_tp = (int)n_tp; F.tp_ptr = reinterpret_cast<const int*>(base + o_tpp); F.te_src = reinterpret_cast<const uint32_t*>(base + o_tes);
    F.tp_val = reinterpret_cast<double*>(base + o_tpv);
    F.cl_ptr[0] = reinterpret_cast<const int*>(base + o_cp0); F.cl_ptr[1] = reinterpret_cast<const int*>(base + o_cp1);
    F.cl_from[0] = reinterpret_cast<const int*>(base + o_cf0); F.cl_from[1] = reinterpret_cast<const int*>(base + o_cf1);
    F.cl_tp[0] = reinterpret_cast<const int*>(base + o_ct0); F.cl_tp[1] = reinterpret_cast<const int*>(base + o_ct1);
    F.cl_val[0] = reinterpret_cast<double*>(base + o_cv0); F.cl_val[1] = reinterpret_cast<double*>(base + o_cv1);
    static bool attr_done = false;
    if (!attr_done) {
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_kft_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KFT_PANEL_LDS));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_kft_step<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KFT_STEP_LDS));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_kft_step<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KFT_STEP_LDS));
        NRS_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_kft_step<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KFT_STEP_LDS));
        attr_done = true;
    }
    H->bytes = off;
    H->kf_nb.resize(K);
    for (int k = 0; k < K; ++k) H->kf_nb[k] = std::max(1, (3 * kf_nf[k] + kf_np[k] + KFT_B - 1) / KFT_B);
    H->on = true;
    e->kft = H;
    return NRS_OK;
}

}  // namespace nrs
