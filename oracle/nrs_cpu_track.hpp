// CPU restatement of the reference's per-frame solves in C++ -- TEST INFRASTRUCTURE ONLY (part of oracle/nrs_cpu.cpp: it is
// included inside that file's anonymous namespace and uses its camera models, SE(3) algebra, block matrix, AMD ordering
// and block Cholesky).  It is the compiled CPU side of the "tracked fps" half of the metric (bench.py cpu_baseline leg)
// and a second checker next to oracle/nrs_oracle.py for
//   a1  CameraPoseOptimization                       modules/optimization/g2o_optimization.cc:50-146
//   a2  CameraPoseAndDeformationOptimization         modules/optimization/g2o_optimization.cc:148-557
// with the edge types
//   ReprojectionErrorOnlyPose / ...WithDeformation   reprojection_error_only_pose.cc:50-75, reprojection_error_with_deformation.cc:37-68
//   SpatialRegularizerWithDeformation                spatial_regularizer_with_deformation.cc:36-49
//   PositionRegularizerWithDeformation               position_regularizer_with_deformation.cc:31-57
//   SpatialRegularizerFixed                          spatial_regularizer_fixed.cc:32-43
// g2o's machinery as in nrs_cpu.cpp: active set = level-0 edges with a non-fixed vertex (sparse_optimizer.cpp:203-285),
// quadratic form with Huber and without rho'' (base_fixed_sized_edge.hpp:49-63), Levenberg-Marquardt
// (optimization_algorithm_levenberg.cpp:57-174), and the linear solve the reference runs: a full sparse Cholesky of
// H + lambda I per LM trial with a fill-reducing ordering on the block pattern, symbolic step once per optimize()
// (linear_solver_eigen.h:92-173) -- here the AMD ordering / up-looking block Cholesky of nrs_cpu.cpp, pose blocks last.
// RegularizationGraph::GetEdges / UpdateVertex on the flat graph: regularization_graph.cc:61-146.
// Parity pinning: as the rest of oracle/ -- "parity unpinned" beyond g2o's 36x36 system; held to the NumPy oracle on the
// committed goldens (tests/test_oracle_cpp_track_cpu.py).
#pragma once

struct TStats { double t_total, t_graph, t_structure, t_factor, t_solve, t_linearize; double chol_flops; int32_t n_factor, n_trials, n_iters, unknowns_max; };

struct TGraph {
    int model = 0;
    float prm[8];
    Pose pose, pose_bak;
    bool pose_fixed = false;
    int N = 0;                                   // point vertices (3-dof, additive: LandmarkVertex, landmark_vertex.cc:36-43)
    vector<double> x, x_bak;                     // estimates (deformation delta)
    vector<uint8_t> pt_fixed;
    // reprojection edges: edge i observes point vertex i (deform) or the constant X0[i] (pose-only)
    int n_rep = 0;
    bool rep_pt = true;
    vector<double> X0, uv, rep_err;              // 3 n, 2 n, 2 n (stored _error)
    vector<int> rep_level;
    double info_rep = 1, delta_rep = 0;
    // pair edges (damper 3-d, spring 1-d) on the same (i, j)
    vector<int> ei, ej, dm_level;
    vector<double> ew, ed0, dm_err;              // dm_err 3 E
    double info_dm = 1, delta_dm = 0, info_sp = 1, delta_sp = 0, k_spring = 1.1;
    // unary dampers against a vertex read live (SpatialRegularizerFixed: raw pointer to the other vertex' estimate)
    vector<int> ui, uj;
    vector<double> uw;
    // embedded mode (oracle/embedded_oracle.py SkinnedReprojEdges): observations of optimised points WITHOUT a vertex -- the point sits at
    // X0 + sum_k om_k x[node_k] over <= 11 vertices (pads: vertex 0 with weight 0, as the oracle's arrays), Jacobian om_k J_l per node
    static constexpr int SKN = 11, SKV = SKN + 2;
    int n_sk = 0;
    vector<int> sk_node, sk_level;
    vector<double> sk_om, sk_X0, sk_uv, sk_err;
    vector<uint8_t> a_sk;
    vector<int64_t> s_sk;                        // (v1, v2) row-major over (pose omega, pose upsilon, the 11 nodes), -1: not in the system
    // ---- per optimize(): active sets, index mapping, structure
    vector<uint8_t> a_rep, a_dm, a_sp, a_un;
    vector<int> blk;                             // point vertex -> block (permuted), -1 = not in the system
    int pb[2] = {-1, -1};                        // pose blocks (omega, upsilon)
    int nb = 0;
    BlockMat H;
    vector<double> b;
    BlockChol chol;
    vector<int64_t> s_rep, s_pair, s_un;         // slots into H.val
    TStats* st = nullptr;

    int E() const { return (int)ei.size(); }

    void cam_point(int i, double* xw) const {
        for (int a = 0; a < 3; ++a) xw[a] = X0[3 * (size_t)i + a] + (rep_pt ? x[3 * (size_t)i + a] : 0.0);
    }
    void rep_residual(int i, double* r, double* pc_out = nullptr) const {
        double xw[3], pc[3];
        cam_point(i, xw);
        quat_rotate(pose.q, xw, pc);
        for (int a = 0; a < 3; ++a) pc[a] += pose.t[a];
        float u, v;
        project_f32(model, prm, (float)pc[0], (float)pc[1], (float)pc[2], u, v);
        r[0] = uv[2 * (size_t)i] - (double)u; r[1] = uv[2 * (size_t)i + 1] - (double)v;
        if (pc_out) { pc_out[0] = pc[0]; pc_out[1] = pc[1]; pc_out[2] = pc[2]; }
    }
    void sk_deformation(int i, double* d) const {                    // (sequential over the nodes, as SkinnedReprojEdges.deformation)
        d[0] = d[1] = d[2] = 0;
        for (int k = 0; k < SKN; ++k) {
            const double om = sk_om[SKN * (size_t)i + k];
            const double* xn = &x[3 * (size_t)sk_node[SKN * (size_t)i + k]];
            for (int a = 0; a < 3; ++a) d[a] += om * xn[a];
        }
    }
    void sk_residual(int i, double* r, double* pc_out = nullptr) const {
        double d[3], xw[3], pc[3];
        sk_deformation(i, d);
        for (int a = 0; a < 3; ++a) xw[a] = sk_X0[3 * (size_t)i + a] + d[a];
        quat_rotate(pose.q, xw, pc);
        for (int a = 0; a < 3; ++a) pc[a] += pose.t[a];
        float u, v;
        project_f32(model, prm, (float)pc[0], (float)pc[1], (float)pc[2], u, v);
        r[0] = sk_uv[2 * (size_t)i] - (double)u; r[1] = sk_uv[2 * (size_t)i + 1] - (double)v;
        if (pc_out) { pc_out[0] = pc[0]; pc_out[1] = pc[1]; pc_out[2] = pc[2]; }
    }
    void dm_residual(int k, double* r) const {
        for (int a = 0; a < 3; ++a) r[a] = ew[k] * (x[3 * (size_t)ei[k] + a] - x[3 * (size_t)ej[k] + a]);
    }
    double sp_residual(int k, double* v, double& d) const {
        for (int a = 0; a < 3; ++a) v[a] = (X0[3 * (size_t)ei[k] + a] + x[3 * (size_t)ei[k] + a]) - (X0[3 * (size_t)ej[k] + a] + x[3 * (size_t)ej[k] + a]);
        d = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        return k_spring * (d - ed0[k]) / ed0[k];
    }
    void un_residual(int k, double* r) const {
        for (int a = 0; a < 3; ++a) r[a] = uw[k] * (x[3 * (size_t)ui[k] + a] - x[3 * (size_t)uj[k] + a]);
    }

    // SparseOptimizer::initializeOptimization(0) + buildIndexMapping + BlockSolver::buildStructure + the symbolic step
    bool initialize() {
        const double t0 = now_s();
        const int Ec = E(), U = (int)ui.size();
        a_rep.assign(n_rep, 0); a_dm.assign(Ec, 0); a_sp.assign(Ec, 0); a_un.assign(U, 0);
        vector<uint8_t> used(N, 0);
        bool pose_used = false;
        for (int i = 0; i < n_rep; ++i) {
            const bool allfix = pose_fixed && (!rep_pt || pt_fixed[i]);
            a_rep[i] = rep_level[i] == 0 && !allfix;
            if (a_rep[i]) { pose_used = true; if (rep_pt) used[i] = 1; }
        }
        for (int k = 0; k < Ec; ++k) {
            const bool allfix = pt_fixed[ei[k]] && pt_fixed[ej[k]];
            a_dm[k] = dm_level[k] == 0 && !allfix;
            a_sp[k] = !allfix;                                       // position regularisers stay at level 0 (OPT:368-393 never touches them)
            if (a_dm[k] || a_sp[k]) { used[ei[k]] = 1; used[ej[k]] = 1; }
        }
        for (int k = 0; k < U; ++k) { a_un[k] = !pt_fixed[ui[k]]; if (a_un[k]) used[ui[k]] = 1; }
        a_sk.assign(n_sk, 0);
        for (int i = 0; i < n_sk; ++i) {
            bool allfix = pose_fixed;
            for (int k = 0; k < SKN; ++k) allfix = allfix && pt_fixed[sk_node[SKN * (size_t)i + k]];
            a_sk[i] = sk_level[i] == 0 && !allfix;
            if (a_sk[i]) { pose_used = true; for (int k = 0; k < SKN; ++k) used[sk_node[SKN * (size_t)i + k]] = 1; }
        }
        // natural order: pose blocks first, then the active non-fixed points by id
        vector<int> nat(N, -1);
        int n = 0;
        const bool pose_in = pose_used && !pose_fixed;
        if (pose_in) n = 2;
        for (int i = 0; i < N; ++i) if (used[i] && !pt_fixed[i]) nat[i] = n++;
        nb = n;
        if (nb == 0) return false;
        // block pattern in natural order, AMD on it (pose blocks last), then the final structure
        auto pattern = [&](const vector<int>& pinv, BlockMat& M) {
            vector<std::pair<int, int>> pr;
            auto P = [&](int v) { return pinv[v]; };
            if (pose_in) { const int a = P(0), c = P(1); pr.emplace_back(a, a); pr.emplace_back(a, c); pr.emplace_back(c, a); pr.emplace_back(c, c); }
            for (int i = 0; i < n_rep; ++i) {
                if (!a_rep[i] || !rep_pt || nat[i] < 0) continue;
                const int l = P(nat[i]);
                pr.emplace_back(l, l);
                if (pose_in) { const int a = P(0), c = P(1); pr.emplace_back(a, l); pr.emplace_back(l, a); pr.emplace_back(c, l); pr.emplace_back(l, c); }
            }
            for (int k = 0; k < Ec; ++k) {
                if (!a_dm[k] && !a_sp[k]) continue;
                const int i = nat[ei[k]], j = nat[ej[k]];
                if (i >= 0) pr.emplace_back(P(i), P(i));
                if (j >= 0) pr.emplace_back(P(j), P(j));
                if (i >= 0 && j >= 0) { pr.emplace_back(P(i), P(j)); pr.emplace_back(P(j), P(i)); }
            }
            for (int k = 0; k < U; ++k) if (a_un[k]) { const int i = P(nat[ui[k]]); pr.emplace_back(i, i); }
            for (int i = 0; i < n_sk; ++i) {
                if (!a_sk[i]) continue;
                int v[SKV];
                v[0] = pose_in ? P(0) : -1; v[1] = pose_in ? P(1) : -1;
                for (int k = 0; k < SKN; ++k) { const int nt = nat[sk_node[SKN * (size_t)i + k]]; v[2 + k] = nt >= 0 ? P(nt) : -1; }
                for (int p = 0; p < SKV; ++p) for (int q = 0; q < SKV; ++q) if (v[p] >= 0 && v[q] >= 0) pr.emplace_back(v[p], v[q]);
            }
            std::sort(pr.begin(), pr.end());
            pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
            M.n = nb;
            M.ptr.assign(nb + 1, 0);
            M.col.resize(pr.size());
            for (size_t q = 0; q < pr.size(); ++q) { M.ptr[pr[q].first + 1]++; M.col[q] = pr[q].second; }
            for (int q = 0; q < nb; ++q) M.ptr[q + 1] += M.ptr[q];
            M.val.assign(9 * pr.size(), 0.0);
            M.diag.resize(nb);
            for (int q = 0; q < nb; ++q) M.diag[q] = M.find(q, q);
        };
        vector<int> pinv(nb);
        for (int q = 0; q < nb; ++q) pinv[q] = q;
        {
            BlockMat H0;
            pattern(pinv, H0);
            vector<uint8_t> last(nb, 0);
            if (pose_in) last[0] = last[1] = 1;
            const vector<int> perm = amd_order(H0, last);
            for (int q = 0; q < nb; ++q) pinv[perm[q]] = q;
        }
        pattern(pinv, H);
        pb[0] = pose_in ? pinv[0] : -1; pb[1] = pose_in ? pinv[1] : -1;
        blk.assign(N, -1);
        for (int i = 0; i < N; ++i) if (nat[i] >= 0) blk[i] = pinv[nat[i]];
        // slots
        s_rep.assign(9 * (size_t)n_rep, -1);
        for (int i = 0; i < n_rep; ++i) {
            if (!a_rep[i]) continue;
            int64_t* s = &s_rep[9 * (size_t)i];
            const int a = pb[0], c = pb[1], l = rep_pt ? blk[i] : -1;
            if (a >= 0) { s[0] = H.find(a, a); s[1] = H.find(a, c); s[2] = H.find(c, a); s[3] = H.find(c, c); }
            if (l >= 0) {
                s[8] = H.find(l, l);
                if (a >= 0) { s[4] = H.find(a, l); s[5] = H.find(l, a); s[6] = H.find(c, l); s[7] = H.find(l, c); }
            }
        }
        s_pair.assign(4 * (size_t)Ec, -1);
        for (int k = 0; k < Ec; ++k) {
            if (!a_dm[k] && !a_sp[k]) continue;
            const int i = blk[ei[k]], j = blk[ej[k]];
            int64_t* s = &s_pair[4 * (size_t)k];
            if (i >= 0) s[0] = H.find(i, i);
            if (j >= 0) s[3] = H.find(j, j);
            if (i >= 0 && j >= 0) { s[1] = H.find(i, j); s[2] = H.find(j, i); }
        }
        s_un.assign(U, -1);
        for (int k = 0; k < U; ++k) if (a_un[k]) s_un[k] = H.find(blk[ui[k]], blk[ui[k]]);
        s_sk.assign((size_t)SKV * SKV * n_sk, -1);
        for (int i = 0; i < n_sk; ++i) {
            if (!a_sk[i]) continue;
            int v[SKV];
            v[0] = pb[0]; v[1] = pb[1];
            for (int k = 0; k < SKN; ++k) v[2 + k] = blk[sk_node[SKN * (size_t)i + k]];
            for (int p = 0; p < SKV; ++p) for (int q = 0; q < SKV; ++q)
                if (v[p] >= 0 && v[q] >= 0) s_sk[((size_t)i * SKV + p) * SKV + q] = H.find(v[p], v[q]);
        }
        b.assign(3 * (size_t)nb, 0.0);
        chol = BlockChol();
        chol.analyze(H);
        if (st) { st->t_structure += now_s() - t0; st->chol_flops = std::max(st->chol_flops, chol.flops); st->unknowns_max = std::max(st->unknowns_max, 3 * nb); }
        return true;
    }

    // computeActiveErrors: the active edges' stored errors; returns the robustified chi2 (sparse_optimizer.cpp:101-114)
    double active_chi2() {
        double chi = 0, rho0, rho1;
        for (int i = 0; i < n_rep; ++i) {
            if (!a_rep[i]) continue;
            double* r = &rep_err[2 * (size_t)i];
            rep_residual(i, r);
            huber(info_rep * (r[0] * r[0] + r[1] * r[1]), delta_rep, rho0, rho1);
            chi += rho0;
        }
        for (int k = 0; k < E(); ++k) {
            if (a_dm[k]) {
                double* r = &dm_err[3 * (size_t)k];
                dm_residual(k, r);
                huber(info_dm * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), delta_dm, rho0, rho1);
                chi += rho0;
            }
        }
        for (int k = 0; k < E(); ++k) {
            if (a_sp[k]) {
                double v[3], d;
                const double r = sp_residual(k, v, d);
                huber(info_sp * r * r, delta_sp, rho0, rho1);
                chi += rho0;
            }
        }
        for (int i = 0; i < n_sk; ++i) {
            if (!a_sk[i]) continue;
            double* r = &sk_err[2 * (size_t)i];
            sk_residual(i, r);
            huber(info_rep * (r[0] * r[0] + r[1] * r[1]), delta_rep, rho0, rho1);
            chi += rho0;
        }
        for (size_t k = 0; k < ui.size(); ++k) {
            if (!a_un[k]) continue;
            double r[3];
            un_residual((int)k, r);
            huber(info_dm * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), delta_dm, rho0, rho1);
            chi += rho0;
        }
        return chi;
    }

    static void add3(double* blk_, const double* Ja, const double* Jc, int rows, double w) {    // blk += w Ja^T Jc (rows x 3 each)
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) {
                double s = 0;
                for (int r = 0; r < rows; ++r) s += Ja[3 * r + p] * Jc[3 * r + q];
                blk_[3 * p + q] += w * s;
            }
    }

    // linearizeOplus + constructQuadraticForm over the active edges at the stored errors' state (errors are current)
    void build_system() {
        std::fill(H.val.begin(), H.val.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0);
        double R[9], rho0, rho1;
        quat_to_R(pose.q, R);
        for (int i = 0; i < n_rep; ++i) {
            if (!a_rep[i]) continue;
            double r[2], pc[3];
            rep_residual(i, r, pc);
            float Jf[6];
            projjac_f32(model, prm, (float)pc[0], (float)pc[1], (float)pc[2], Jf);
            huber(info_rep * (r[0] * r[0] + r[1] * r[1]), delta_rep, rho0, rho1);
            const double w = rho1 * info_rep;
            double Ja[6], Jc[6], Jl[6];
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                Ja[3 * rr] = -j1 * pc[2] + j2 * pc[1]; Ja[3 * rr + 1] = j0 * pc[2] - j2 * pc[0]; Ja[3 * rr + 2] = -j0 * pc[1] + j1 * pc[0];
                Jc[3 * rr] = j0; Jc[3 * rr + 1] = j1; Jc[3 * rr + 2] = j2;
                Jl[3 * rr] = j0 * R[0] + j1 * R[3] + j2 * R[6];
                Jl[3 * rr + 1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
                Jl[3 * rr + 2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
            }
            const int64_t* s = &s_rep[9 * (size_t)i];
            const int l = rep_pt ? blk[i] : -1;
            if (pb[0] >= 0) {
                add3(&H.val[9 * s[0]], Ja, Ja, 2, w); add3(&H.val[9 * s[1]], Ja, Jc, 2, w);
                add3(&H.val[9 * s[2]], Jc, Ja, 2, w); add3(&H.val[9 * s[3]], Jc, Jc, 2, w);
                for (int p = 0; p < 3; ++p) {
                    b[3 * (size_t)pb[0] + p] -= w * (Ja[p] * r[0] + Ja[3 + p] * r[1]);
                    b[3 * (size_t)pb[1] + p] -= w * (Jc[p] * r[0] + Jc[3 + p] * r[1]);
                }
            }
            if (l >= 0) {
                add3(&H.val[9 * s[8]], Jl, Jl, 2, w);
                for (int p = 0; p < 3; ++p) b[3 * (size_t)l + p] -= w * (Jl[p] * r[0] + Jl[3 + p] * r[1]);
                if (pb[0] >= 0) {
                    add3(&H.val[9 * s[4]], Ja, Jl, 2, w); add3(&H.val[9 * s[5]], Jl, Ja, 2, w);
                    add3(&H.val[9 * s[6]], Jc, Jl, 2, w); add3(&H.val[9 * s[7]], Jl, Jc, 2, w);
                }
            }
        }
        for (int k = 0; k < E(); ++k) {
            const int64_t* s = &s_pair[4 * (size_t)k];
            const int bi = blk[ei[k]], bj = blk[ej[k]];
            if (a_dm[k]) {                                           // r = w (d_i - d_j), J = (+w I, -w I)
                double r[3];
                dm_residual(k, r);
                huber(info_dm * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), delta_dm, rho0, rho1);
                const double wi = rho1 * info_dm, sfac = wi * ew[k] * ew[k];
                for (int d = 0; d < 9; d += 4) {
                    if (bi >= 0) H.val[9 * s[0] + d] += sfac;
                    if (bj >= 0) H.val[9 * s[3] + d] += sfac;
                    if (bi >= 0 && bj >= 0) { H.val[9 * s[1] + d] -= sfac; H.val[9 * s[2] + d] -= sfac; }
                }
                for (int p = 0; p < 3; ++p) {
                    if (bi >= 0) b[3 * (size_t)bi + p] -= wi * ew[k] * r[p];
                    if (bj >= 0) b[3 * (size_t)bj + p] += wi * ew[k] * r[p];
                }
            }
            if (a_sp[k]) {                                           // r = k (d - d0) / d0, J_i = k / (2 d0 d) 2 v^T
                double v[3], d;
                const double r = sp_residual(k, v, d);
                huber(info_sp * r * r, delta_sp, rho0, rho1);
                const double wi = rho1 * info_sp, a = k_spring / (2 * ed0[k] * d);
                const double g[3] = {a * (2 * v[0]), a * (2 * v[1]), a * (2 * v[2])};
                for (int p = 0; p < 3; ++p)
                    for (int q = 0; q < 3; ++q) {
                        const double t = wi * g[p] * g[q];
                        if (bi >= 0) H.val[9 * s[0] + 3 * p + q] += t;
                        if (bj >= 0) H.val[9 * s[3] + 3 * p + q] += t;
                        if (bi >= 0 && bj >= 0) { H.val[9 * s[1] + 3 * p + q] -= t; H.val[9 * s[2] + 3 * p + q] -= t; }
                    }
                for (int p = 0; p < 3; ++p) {
                    if (bi >= 0) b[3 * (size_t)bi + p] -= wi * r * g[p];
                    if (bj >= 0) b[3 * (size_t)bj + p] += wi * r * g[p];
                }
            }
        }
        for (int i = 0; i < n_sk; ++i) {                                // skinned observations: J_pose as the reprojection edges', J_node_k = om_k J_l
            if (!a_sk[i]) continue;
            double r[2], pc[3];
            sk_residual(i, r, pc);
            float Jf[6];
            projjac_f32(model, prm, (float)pc[0], (float)pc[1], (float)pc[2], Jf);
            huber(info_rep * (r[0] * r[0] + r[1] * r[1]), delta_rep, rho0, rho1);
            const double w = rho1 * info_rep;
            double J[SKV][6], Jl[6];
            for (int rr = 0; rr < 2; ++rr) {
                const double j0 = -(double)Jf[3 * rr], j1 = -(double)Jf[3 * rr + 1], j2 = -(double)Jf[3 * rr + 2];
                J[0][3 * rr] = -j1 * pc[2] + j2 * pc[1]; J[0][3 * rr + 1] = j0 * pc[2] - j2 * pc[0]; J[0][3 * rr + 2] = -j0 * pc[1] + j1 * pc[0];
                J[1][3 * rr] = j0; J[1][3 * rr + 1] = j1; J[1][3 * rr + 2] = j2;
                Jl[3 * rr] = j0 * R[0] + j1 * R[3] + j2 * R[6];
                Jl[3 * rr + 1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
                Jl[3 * rr + 2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
            }
            int vb[SKV];
            vb[0] = pb[0]; vb[1] = pb[1];
            for (int k = 0; k < SKN; ++k) {
                vb[2 + k] = blk[sk_node[SKN * (size_t)i + k]];
                const double om = sk_om[SKN * (size_t)i + k];
                for (int cc = 0; cc < 6; ++cc) J[2 + k][cc] = om * Jl[cc];
            }
            const int64_t* sl = &s_sk[(size_t)i * SKV * SKV];
            for (int p = 0; p < SKV; ++p) {
                if (vb[p] < 0) continue;
                for (int q = 0; q < SKV; ++q)
                    if (vb[q] >= 0) add3(&H.val[9 * sl[p * SKV + q]], J[p], J[q], 2, w);
                for (int cc = 0; cc < 3; ++cc) b[3 * (size_t)vb[p] + cc] -= w * (J[p][cc] * r[0] + J[p][3 + cc] * r[1]);
            }
        }
        for (size_t k = 0; k < ui.size(); ++k) {
            if (!a_un[k]) continue;
            double r[3];
            un_residual((int)k, r);
            huber(info_dm * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]), delta_dm, rho0, rho1);
            const double wi = rho1 * info_dm, sfac = wi * uw[k] * uw[k];
            const int bi = blk[ui[k]];
            for (int d = 0; d < 9; d += 4) H.val[9 * s_un[k] + d] += sfac;
            for (int p = 0; p < 3; ++p) b[3 * (size_t)bi + p] -= wi * uw[k] * r[p];
        }
    }

    void push() { pose_bak = pose; x_bak = x; }
    void pop() { pose = pose_bak; x = x_bak; }
    void update(const double* dx) {
        if (pb[0] >= 0) {
            double upd[6];
            for (int a = 0; a < 3; ++a) { upd[a] = dx[3 * (size_t)pb[0] + a]; upd[3 + a] = dx[3 * (size_t)pb[1] + a]; }
            pose_oplus(pose, upd);
        }
        for (int i = 0; i < N; ++i)
            if (blk[i] >= 0)
                for (int a = 0; a < 3; ++a) x[3 * (size_t)i + a] += dx[3 * (size_t)blk[i] + a];
    }

    // SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
    int optimize(int iterations, int round, Trial* trace, int trace_cap, int* n_tr) {
        if (!initialize()) return -1;
        vector<double> dx(3 * (size_t)nb, 0.0);
        double lam = -1, ni = 2;
        int done = 0;
        for (int it = 0; it < iterations; ++it) {
            double t0 = now_s();
            double chi = active_chi2();
            build_system();
            if (st) st->t_linearize += now_s() - t0;
            if (it == 0) {
                double md = 0;
                for (int q = 0; q < nb; ++q) { const double* d = &H.val[9 * H.diag[q]]; md = std::max(md, std::max(std::fabs(d[0]), std::max(std::fabs(d[4]), std::fabs(d[8])))); }
                lam = 1e-5 * md;
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                push();
                t0 = now_s();
                const bool ok = chol.factor(H, lam);
                if (st) { st->t_factor += now_s() - t0; st->n_factor++; }
                t0 = now_s();
                if (ok) chol.solve(b.data(), dx.data());              // (not positive definite: the stale x is applied, as in the reference)
                if (st) st->t_solve += now_s() - t0;
                update(dx.data());
                const double temp = ok ? active_chi2() : std::numeric_limits<double>::max();
                if (!ok) (void)active_chi2();
                double scale_lm = 1e-3;
                for (size_t q = 0; q < dx.size(); ++q) scale_lm += dx[q] * (lam * dx[q] + b[q]);
                rho = (chi - temp) / scale_lm;
                const bool accepted = rho > 0 && std::isfinite(temp);
                if (trace && *n_tr < trace_cap) trace[*n_tr] = Trial{it + 100 * round, qmax, accepted, ok, 0, lam, chi, temp, rho};
                ++*n_tr;
                if (st) st->n_trials++;
                if (accepted) {
                    double alpha = 1.0 - std::pow(2 * rho - 1, 3);
                    alpha = std::min(alpha, 2.0 / 3.0);
                    lam *= std::max(1.0 / 3.0, alpha);
                    ni = 2;
                    chi = temp;
                } else {
                    lam *= ni;
                    ni *= 2;
                    pop();
                    if (!std::isfinite(lam)) break;
                }
                ++qmax;
            } while (rho < 0 && qmax < 10);
            ++done;
            if (st) st->n_iters++;
            if (qmax == 10 || rho == 0 || !std::isfinite(lam)) break;
        }
        return done;
    }
};

// flat RegularizationGraph (include/nrs.h nrs_graph): GetEdges (regularization_graph.cc:61-87), UpdateVertex (:89-146)
struct FlatGraph {
    int n_points;
    const int32_t *rowptr, *col, *eid;
    float *e_w, *e_max, *e_min;
    const float* e_d0;
    int32_t* e_status;
    float sigma, stretch_th, min_w;
    __attribute__((optimize("fp-contract=off"))) static float weight(float d, float sigma) {                      // InterpolationWeight: float argument, exp in double, rounded to float
        const float arg = -(d * d) / (2.0f * sigma * sigma);
        return (float)std::exp((double)arg);
    }
    // positions (into the row's CSR range) in the reference's order: status asc, weight desc, id asc; cut at the first weight < min_weight
    void get_edges(int p, vector<int>& out) const {
        out.clear();
        for (int a = rowptr[p]; a < rowptr[p + 1]; ++a) out.push_back(a);
        std::stable_sort(out.begin(), out.end(), [&](int a, int c) {
            const int sa = e_status[eid[a]], sc = e_status[eid[c]];
            if (sa != sc) return sa < sc;
            return e_w[eid[a]] > e_w[eid[c]];
        });
        size_t k = 0;
        while (k < out.size() && !(e_w[eid[out[k]]] < min_w)) ++k;
        out.resize(k);
    }
    __attribute__((optimize("fp-contract=off"))) int update_vertex(int p, const float* pos) {
        int good = 0;
        for (int a = rowptr[p]; a < rowptr[p + 1]; ++a) {
            const int e = eid[a], o = col[a];
            const float dx = pos[3 * p] - pos[3 * o], dy = pos[3 * p + 1] - pos[3 * o + 1], dz = pos[3 * p + 2] - pos[3 * o + 2];
            float s = dx * dx;
            s += dy * dy;
            s += dz * dz;
            const float d = std::sqrt(s);
            e_max[e] = std::max(e_max[e], d);
            e_min[e] = std::min(e_min[e], d);
            e_w[e] = weight(e_max[e], sigma);
            if (std::fabs((e_max[e] - e_min[e]) / e_min[e]) > stretch_th) e_status[e] = 3;
            else ++good;
        }
        return good;
    }
};
