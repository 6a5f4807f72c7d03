// Host reference of the numeric phase of the direct solve (TEST INFRASTRUCTURE ONLY): dense fronts, plain loops, on exactly
// the arrays nr-slam_amd/csrc/nrs_nd_plan.hpp lays out -- what the device kernels k_nd_level / k_nd_back compute.  It checks
// the plan (child maps, entry lists, offsets) without a GPU (tests/test_nd_cpu.py) and is the yardstick of the device
// kernels (tests/test_gpu_nd.py).  The algorithm it stands for on the reference side: LinearSolverEigen::solve
// (third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-136), a sparse Cholesky of (H + lambda I).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include "../nr-slam_amd/csrc/nrs_nd_plan.hpp"

namespace nrs {
// Host reference of the numeric phase on exactly these arrays (dense fronts, plain loops): what the device kernels compute.
//   Dn: 9 doubles per node (its diagonal block, row-major), Vp: 9 per pair (rows = pair_hi's components), bn: 3 per node.
// Returns false if a pivot is not positive (the reference's LinearSolverEigen::solve then reports failure, linear_solver_eigen.h:124-136).
inline bool nd_host_solve(const NdPlan& P, const double* Dn, const double* Vp, const double* bn, double lam, double* x) {
    std::vector<double> Lp(P.L_doubles, 0.0), U(P.U_doubles, 0.0), Fm;
    bool ok = true;
    for (int l = 0; l < P.n_levels; ++l)
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {
            const NdFrontD& D = P.fr[P.lvl_fronts[i]];
            const int s = D.s, m = D.s + D.b, n = m + 1, mn = m / 3;
            Fm.assign((size_t)n * n, 0.0);
            auto at = [&](int r, int c) -> double& { return Fm[(size_t)r * n + c]; };
            for (int e = 0; e < D.n_ent; ++e) {
                const NdEnt& E = P.ent[D.ent_off + e];
                const uint32_t kind = E.src >> ND_KIND_SHIFT, src = E.src & ND_SRC_MASK;
                if (kind == 2) { for (int j = 0; j < 3; ++j) at(m, 3 * E.c + j) += bn[3 * (size_t)src + j]; continue; }
                const double* v = kind == 0 ? Dn + 9 * (size_t)src : Vp + 9 * (size_t)src;
                for (int a = 0; a < 3; ++a)
                    for (int j = 0; j < 3; ++j) at(3 * E.r + a, 3 * E.c + j) += v[3 * a + j] + (kind == 0 && a == j ? lam : 0.0);
            }
            for (int k = 0; k < D.n_ch; ++k) {
                const NdFrontD& C = P.fr[P.child[D.ch_off + k]];
                const int16_t* cm = &P.cmap[D.cmap_off + (size_t)k * (mn + 1)];
                const double* Uc = &U[C.U_off];
                auto crow = [&](int r) { const int a = cm[r / 3]; return a < 0 ? -1 : 3 * a + r % 3; };   // (r == m: node slot mn, component 0)
                for (int r = 0; r < n; ++r) {
                    const int a = crow(r);
                    if (a < 0) continue;
                    for (int c2 = 0; c2 <= r && c2 < m; ++c2) {
                        const int b2 = crow(c2);
                        if (b2 >= 0) at(r, c2) += Uc[(size_t)a * C.ldU + b2];
                    }
                }
            }
            // partial Cholesky of the first s columns over all n rows (lower triangle)
            for (int j = 0; j < s; ++j) {
                double d = at(j, j);
                if (!(d > 0)) { ok = false; d = 1; }
                const double l = std::sqrt(d);
                at(j, j) = l;
                for (int r = j + 1; r < n; ++r) at(r, j) /= l;
                for (int c2 = j + 1; c2 < m; ++c2) {
                    const double v = at(c2, j);
                    if (v == 0) continue;
                    for (int r = c2; r < n; ++r) at(r, c2) -= at(r, j) * v;
                }
            }
            double* L = &Lp[D.L_off];
            for (int r = 0; r < n; ++r)
                for (int c2 = 0; c2 < s; ++c2) L[(size_t)r * s + c2] = c2 <= r ? at(r, c2) : 0.0;
            double* Uf = &U[D.U_off];
            for (int r = 0; r <= D.b; ++r)
                for (int c2 = 0; c2 <= r && c2 < D.b; ++c2) { Uf[(size_t)r * D.ldU + c2] = at(s + r, s + c2); if (r < D.b) Uf[(size_t)c2 * D.ldU + r] = at(s + r, s + c2); }
        }
    // back substitution, root first: L11^T x_own = y - L21^T x_bnd
    std::vector<double> t;
    for (int l = P.n_levels - 1; l >= 0; --l)
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {
            const NdFrontD& D = P.fr[P.lvl_fronts[i]];
            const int s = D.s, m = D.s + D.b;
            const double* L = &Lp[D.L_off];
            t.assign(s, 0.0);
            for (int q = 0; q < s; ++q) t[q] = L[(size_t)m * s + q];
            for (int r = 0; r < D.b; ++r) {
                const double xb = x[3 * (size_t)P.bnd[D.bnd_off + r / 3] + r % 3];
                for (int q = 0; q < s; ++q) t[q] -= L[(size_t)(s + r) * s + q] * xb;
            }
            for (int p = s - 1; p >= 0; --p) {
                const double xp = t[p] / L[(size_t)p * s + p];
                x[3 * (size_t)P.own[D.own_off + p / 3] + p % 3] = xp;
                for (int q = 0; q < p; ++q) t[q] -= L[(size_t)p * s + q] * xp;
            }
        }
    return ok;
}

}  // namespace nrs

extern "C" int nrs_cpu_nd_solve(int32_t n_nodes, const double* pos, const uint8_t* last, int32_t n_pairs, const int32_t* pairs,
                                const double* Dn, const double* Vp, const double* bn, double lambda, double* x, int64_t* stats) {
    nrs::NdPlan P;
    std::string err;
    if (n_nodes <= 0 || n_pairs < 0 || !nrs::nd_build_plan(n_nodes, pos, last, n_pairs, pairs, P, &err)) {
        if (getenv("NRS_ND_ERR")) fprintf(stderr, "[nd_host] no plan: %s\n", err.c_str());
        return -1;
    }
    if (stats) {
        stats[0] = P.n_fronts; stats[1] = P.n_levels; stats[2] = P.max_s; stats[3] = P.max_b;
        stats[4] = (int64_t)P.L_doubles; stats[5] = (int64_t)P.U_doubles; stats[6] = (int64_t)P.flops; stats[7] = (int64_t)P.wg.size() / 3;
    }
    std::vector<double> V(9 * (size_t)n_pairs);                    // caller: rows = first node of the pair; plan: rows = the later-eliminated node
    for (int q = 0; q < n_pairs; ++q) {
        const bool flip = P.pair_hi[q] != pairs[2 * q];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) V[9 * (size_t)q + 3 * a + b] = flip ? Vp[9 * (size_t)q + 3 * b + a] : Vp[9 * (size_t)q + 3 * a + b];
    }
    return nrs::nd_host_solve(P, Dn, V.data(), bn, lambda, x) ? 0 : -6;
}

// Structural invariants of a plan the device kernels rely on (tests/test_nd_cpu.py): returns 0, or the number of the first
// violated check.  (1) every node is owned by exactly one front; (2) both ends of every pair sit in the front that owns the
// earlier-eliminated one (own or boundary): the separators separate; (3) a front's boundary is sorted by elimination position and
// all of it lies in the parent's front; (4) the boundary's owner segments (NdFrontD::seg_off) tile it in order, parent side first,
// every owner a proper ancestor, each ancestor at most once; (5) a front owns at most ND_SMAXN nodes; (6) workgroups: every
// (I >= J) pair of row blocks once, one inverse workgroup per front, levels ascending.
extern "C" int nrs_cpu_nd_plan_check(int32_t n_nodes, const double* pos, const uint8_t* last, int32_t n_pairs, const int32_t* pairs) {
    nrs::NdPlan P;
    std::string err;
    if (n_nodes <= 0 || n_pairs < 0 || !nrs::nd_build_plan(n_nodes, pos, last, n_pairs, pairs, P, &err)) return -1;
    const int nf = P.n_fronts;
    std::vector<int> owner(n_nodes, -1);
    for (int f = 0; f < nf; ++f) {
        const nrs::NdFrontD& D = P.fr[f];
        if (D.s / 3 > nrs::ND_SMAXN || D.s % 3 || D.b % 3) return 5;
        for (int i = 0; i < D.s / 3; ++i) { const int u = P.own[D.own_off + i]; if (owner[u] >= 0) return 1; owner[u] = f; }
    }
    for (int u = 0; u < n_nodes; ++u) if (owner[u] < 0) return 1;
    auto in_front = [&](int f, int u) {
        const nrs::NdFrontD& D = P.fr[f];
        for (int i = 0; i < D.s / 3; ++i) if (P.own[D.own_off + i] == u) return true;
        for (int i = 0; i < D.b / 3; ++i) if (P.bnd[D.bnd_off + i] == u) return true;
        return false;
    };
    for (int q = 0; q < n_pairs; ++q) {
        const int a = pairs[2 * q], b = pairs[2 * q + 1];
        const int lo = P.elim[a] < P.elim[b] ? a : b, hi = lo == a ? b : a;
        if (!in_front(owner[lo], hi)) return 2;
    }
    for (int f = 0; f < nf; ++f) {
        const nrs::NdFrontD& D = P.fr[f];
        for (int i = 0; i < D.b / 3; ++i) {
            const int u = P.bnd[D.bnd_off + i];
            if (i > 0 && P.elim[u] <= P.elim[P.bnd[D.bnd_off + i - 1]]) return 3;
            if (D.par < 0 || !in_front(D.par, u)) return 3;
        }
        if ((D.par < 0) != (D.b == 0 && D.par < 0)) return 3;
        int row = 0, prev_owner = f;
        std::vector<int> seen;
        for (int k = 0; k < D.n_seg; ++k) {
            const int o = P.seg[2 * (size_t)(D.seg_off + k)], end = P.seg[2 * (size_t)(D.seg_off + k) + 1];
            if (end <= row || end % 3) return 4;
            bool anc = false;                                     // o is a proper ancestor of f, above the previous owner
            for (int g = P.fr[prev_owner].par; g >= 0; g = P.fr[g].par) if (g == o) { anc = true; break; }
            if (!anc) return 4;
            for (int i = row / 3; i < end / 3; ++i) if (owner[P.bnd[D.bnd_off + i]] != o) return 4;
            for (int s2 : seen) if (s2 == o) return 4;
            seen.push_back(o);
            prev_owner = o;
            row = end;
        }
        if (row != D.b) return 4;
    }
    std::vector<int> n_inv(nf, 0), n_reg(nf, 0);
    for (int l = 0; l < P.n_levels; ++l)
        for (int w = P.lvl_wg_ptr[l]; w < P.lvl_wg_ptr[l + 1]; ++w) {
            const int f = P.wg[3 * w], I = P.wg[3 * w + 1], J = P.wg[3 * w + 2];
            if (P.fr[f].level != l) return 6;
            if (I < 0) { if (J >= 0) return 6; n_inv[f]++; }
            else { if (J < 0 || J > I || I >= P.fr[f].nR) return 6; n_reg[f]++; }
        }
    for (int f = 0; f < nf; ++f) {
        const int nR = P.fr[f].nR;
        if (n_reg[f] != nR * (nR + 1) / 2 || n_inv[f] != 1) return 6;
        if (P.fr[f].par >= 0 && P.fr[P.fr[f].par].level <= P.fr[f].level) return 6;
    }
    return 0;
}
