// Symbolic phase of the direct solve of a2's single-frame system (H + lambda I) x = b: nested dissection of the point
// graph, the separator tree as a set of dense FRONTS, everything the numeric kernels (nrs_engine_nd.hpp) index with.
//
// What it replaces: g2o's LinearSolverEigen (reference third_party/g2o/g2o/solvers/eigen/linear_solver_eigen.h:92-173:
// AMD ordering on the block pattern + symbolic factorisation once per optimize(), numeric SimplicialLLT per LM trial) for
// CameraPoseAndDeformationOptimization (modules/optimization/g2o_optimization.cc:148-557): one pose block + one 3-dof
// block per point, point-point coupling through the <= 11 regularisers a point initiates (OPT:255-335), every point
// coupled to the pose.  The points lie on a surface, so the graph is planar-like and recursive bisection with vertex
// separators gives a tree whose fronts are small dense matrices: the work the matrix cores are for.
//
// Unknowns are NODES of 3 scalars (a point's deformation; half a pose block).  A front owns <= ND_SMAXN nodes (its
// separator chunk: the columns it eliminates) and carries a BOUNDARY: the later-eliminated nodes its columns couple to,
// sorted by elimination order, plus one extra row for the right-hand side (forward substitution rides along with the
// factorisation).  Separators larger than ND_SMAXN nodes are split into a chain of fronts.  Plain C++ (no HIP): the
// same arrays drive the device kernels and the host reference solve the tests hold them to (oracle/nd_host.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <exception>
#include <functional>
#include <system_error>
#include <thread>
#include <vector>

namespace nrs {

constexpr int ND_SMAXN = 32;        // own nodes of a front (96 unknowns: its diagonal block lives in LDS)
constexpr int ND_LEAFN = 32;        // leaves of the dissection
constexpr int ND_TB = 48;           // boundary rows per row block (16 nodes)

struct NdFrontD {                   // one front, as the kernels read it
    int s, b;                       // own / boundary unknowns (b without the right-hand-side row)
    int L_off;                      // doubles: panel [(s + b + 2 + s) x s]: L11, L21, y^T, 1 / diag(L11), (L11^-1)^T (device solver: its back pass multiplies)
    int U_off, ldU;                 // doubles: Schur complement [(b + 1) x ldU] handed to the parent (last row: rhs)
    int own_off, bnd_off;           // node lists (own: s / 3 entries, bnd: b / 3)
    int ent_off, n_ent;             // original entries of the own columns
    int ch_off, n_ch;               // children: front ids at child[ch_off ..]; their maps at cmap[cmap_off + k * (m / 3 + 1) ..]
    int cmap_off;
    int nR;                         // boundary row blocks: ceil((b + 1) / ND_TB)
    int level;
    // device data flow: a front's Schur complement is WRITTEN INTO ITS PARENT'S INDEX SPACE (slot `pslot` of the parent's
    // assembly area, through pmap), so that a parent assembles by dense, contiguous reads -- no index maps on the reading side
    int par, pslot;                 // parent front (-1: a root), which of its assembly slots this front writes
    int pmap_off;                   // pmap[pmap_off + i]: position of boundary node i in the parent's node list; [.. + b / 3]: the parent's rhs slot
    int A_off, ldA;                 // doubles: assembly area, n_ch slots of [(s + b + 1) x ldA] (zero where no child writes: zeroed once at upload)
    int pA_off, pldA;               // this front's slot in the parent's assembly area (doubles) and the parent's ldA
    // the boundary by owner: seg[seg_off + i] = (ancestor front, end row of its unknowns in the boundary), i = 0 the parent, then
    // its parent, ...: the boundary is sorted by elimination position and a front's own nodes are contiguous there
    int seg_off, n_seg;
};
// an original entry: 3 x 3 block (kind 0: diagonal block of node `src`, + lambda I; kind 1: pair `src`, rows = the later node)
// or 1 x 3 (kind 2: right-hand side of node `src`) at node positions (r, c) of the front; c is an own column
struct NdEnt { uint16_t r, c; uint32_t src; };
constexpr uint32_t ND_KIND_SHIFT = 30, ND_SRC_MASK = (1u << 30) - 1;

struct NdPlan {
    int n_nodes = 0, n_pairs = 0, n_fronts = 0, n_levels = 0;
    std::vector<NdFrontD> fr;
    std::vector<int> own, bnd, child;
    std::vector<int16_t> cmap;      // per (front, child): front node position (and the rhs slot) -> child boundary position, -1 (host reference)
    std::vector<int16_t> pmap;      // per front: boundary node (and the rhs slot) -> node position in the parent front (device)
    std::vector<NdEnt> ent;
    std::vector<int> lvl_ptr, lvl_fronts;       // fronts of every level (leaves first)
    std::vector<int> seg;                       // (front, end row) pairs: NdFrontD::seg_off
    std::vector<int> lvl_wg_ptr, wg;            // workgroups of every level: (front, row block I, row block J <= I), and (front, -1, -1) for every front
    std::vector<int> lvl_wg_split;              // per level: where its off-diagonal workgroups (I > J) start -- the diagonal (I, I) and inverse
                                                // ones come first, so a crowded level can run as two launches (nrs_engine_nd.hpp k_nd_tile)
    std::vector<int> pair_hi, pair_lo;          // every pair oriented by elimination order (block rows = hi)
    std::vector<int> elim;                      // node -> elimination position
    size_t L_doubles = 0, U_doubles = 0, A_doubles = 0;
    int max_s = 0, max_b = 0, max_ch = 0;
    double flops = 0, flops_crit = 0;
};

// n_nodes nodes at pos (geometry of the dissection), `last` nodes (the pose halves) are eliminated at the root whatever
// their position; pairs: unique unordered couplings (a, b), a != b.  Returns false (err set) if the plan cannot be built.
inline bool nd_build_plan(int n_nodes, const double* pos, const uint8_t* last, int n_pairs, const int* pairs, NdPlan& P, std::string* err,
                          int leaf_n = ND_LEAFN, int smax_n = ND_SMAXN, bool with_cmap = true, int par_min = 0, bool vertex_cover = true) {
    auto fail = [&](const char* m) { if (err) *err = m; return false; };
    P = NdPlan();
    P.n_nodes = n_nodes; P.n_pairs = n_pairs;
    if (n_nodes <= 0) return fail("no nodes");
    if (leaf_n > smax_n || smax_n > ND_SMAXN || leaf_n < 1) return fail("leaf / chunk sizes");
    // ---- adjacency (CSR, with the pair id of every entry)
    std::vector<int> ap(n_nodes + 1, 0), an(2 * (size_t)n_pairs), aq(2 * (size_t)n_pairs);
    for (int q = 0; q < n_pairs; ++q) {
        const int a = pairs[2 * q], b = pairs[2 * q + 1];
        if (a < 0 || b < 0 || a >= n_nodes || b >= n_nodes || a == b) return fail("pair out of range");
        ap[a + 1]++; ap[b + 1]++;
    }
    for (int i = 0; i < n_nodes; ++i) ap[i + 1] += ap[i];
    {
        std::vector<int> fill(ap.begin(), ap.end() - 1);
        for (int q = 0; q < n_pairs; ++q) {
            const int a = pairs[2 * q], b = pairs[2 * q + 1];
            an[fill[a]] = b; aq[fill[a]++] = q;
            an[fill[b]] = a; aq[fill[b]++] = q;
        }
    }
    // ---- separator tree by recursive coordinate bisection
    // The two halves of a bisection are independent problems: with par_min > 0 the first two levels of the recursion hand one half each to a
    // thread of its own when a half has that many nodes (a worker keeps its fronts in a list of its own; the lists are joined in the order a
    // single thread creates the fronts in -- left subtree, right subtree, separator -- so the plan is the same, index for index)
    struct FH { std::vector<int> own, ch; };
    struct Rec {
        const double* pos; const std::vector<int>&ap, &an; int leaf_n, smax_n; bool vertex_cover; int par_min;
        std::vector<FH> F;                                         // this worker's fronts (children: indices into this list)
        std::vector<int> side, slot, mark;
        int stamp = 0;
        std::vector<std::pair<double, int>> keyed;
        int new_front(std::vector<int> own, std::vector<int> ch) { F.push_back(FH{std::move(own), std::move(ch)}); return (int)F.size() - 1; }
        // orders v along its longest axis (not skip_axis): completely (split < 0), or only so far that the `split` smallest come first
        int asort(std::vector<int>& v, int skip_axis, int split = -1) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int u : v)
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos[3 * (size_t)u + a]); hi[a] = std::max(hi[a], pos[3 * (size_t)u + a]); }
            int ax = -1;
            for (int a = 0; a < 3; ++a)
                if (a != skip_axis && (ax < 0 || hi[a] - lo[a] > hi[ax] - lo[ax])) ax = a;
            keyed.resize(v.size());                                    // (coordinate, id) pairs: the sort touches no indirect memory
            for (size_t i = 0; i < v.size(); ++i) keyed[i] = {pos[3 * (size_t)v[i] + ax], v[i]};
            if (split < 0) {
                std::sort(keyed.begin(), keyed.end());
                for (size_t i = 0; i < v.size(); ++i) v[i] = keyed[i].second;
            } else std::nth_element(keyed.begin(), keyed.begin() + split, keyed.end());     // (pairs are distinct: the two halves are determined; v stays as it is)
            return ax;
        }
        // chain of fronts over one separator (or leaf) that is longer than a front may own
        int mk_chain(std::vector<int>& nodes, std::vector<int> ch) {
            const int n = (int)nodes.size(), nc = (n + smax_n - 1) / smax_n;
            int prev = -1;
            for (int k = 0; k < nc; ++k) {
                const int a = (int)((int64_t)n * k / nc), b = (int)((int64_t)n * (k + 1) / nc);
                std::vector<int> own(nodes.begin() + a, nodes.begin() + b);
                prev = new_front(std::move(own), k == 0 ? std::move(ch) : std::vector<int>{prev});
            }
            return prev;
        }
        // a worker for one half: the marks as they stand (what it reads of nodes outside its half never equals a stamp it hands out)
        Rec fork() const { Rec r{pos, ap, an, leaf_n, smax_n, vertex_cover, par_min, {}, side, std::vector<int>(slot.size(), 0), mark, stamp, {}}; return r; }
        // its fronts behind this worker's, its roots in this worker's numbering
        std::vector<int> adopt(Rec& w, std::vector<int> roots) {
            const int off = (int)F.size();
            for (FH& f : w.F) { for (int& c : f.ch) c += off; F.push_back(std::move(f)); }
            for (int& r : roots) r += off;
            return roots;
        }
        std::vector<int> run(std::vector<int> v, int depth) {        // (v ascending by node index, here and in every call below)
            if (v.empty()) return {};
            if ((int)v.size() <= leaf_n) { asort(v, -1); return {mk_chain(v, {})}; }
            const size_t half = v.size() / 2;
            const int ax = asort(v, -1, (int)half);
            // the halves in a defined order (node index): v is, and is read in that order
            const int sl = ++stamp, sr = ++stamp;
            for (size_t i = 0; i < v.size(); ++i) side[keyed[i].second] = i < half ? sl : sr;
            std::vector<int> L, R;
            L.reserve(half); R.reserve(v.size() - half);
            for (int u : v) (side[u] == sl ? L : R).push_back(u);
            // the thinner of the two candidate separators: the L nodes that touch R, or the R nodes that touch L
            // (one pass over L's connections finds both: the graph is symmetric)
            std::vector<int> sepL, sepR;
            const int sm = ++stamp;
            for (int u : L) {
                bool t = false;
                for (int e = ap[u]; e < ap[u + 1]; ++e)
                    if (side[an[e]] == sr) { t = true; mark[an[e]] = sm; }
                if (t) sepL.push_back(u);
            }
            for (int u : R) if (mark[u] == sm) sepR.push_back(u);
            // the smallest set of nodes that covers every edge of the cut: a minimum vertex cover of the bipartite graph (sepL, sepR, cut
            // edges), from a maximum matching (Koenig).  Never larger than the thinner side; on kNN graphs 10-25 % smaller, which takes
            // chain links -- whole levels -- off the top of the tree
            std::vector<int> sep;
            if (!vertex_cover) sep = sepL.size() <= sepR.size() ? sepL : sepR;
            else {
                const int nl = (int)sepL.size(), nr = (int)sepR.size();
                for (int j = 0; j < nr; ++j) slot[sepR[j]] = j;
                std::vector<int> eptr(nl + 1, 0), eadj;
                for (int i = 0; i < nl; ++i) {
                    const int u = sepL[i];
                    for (int e = ap[u]; e < ap[u + 1]; ++e) if (side[an[e]] == sr) eadj.push_back(slot[an[e]]);
                    eptr[i + 1] = (int)eadj.size();
                }
                std::vector<int> ml(nl, -1), mr(nr, -1), seen(nr, -1);
                // Kuhn's augmenting paths (a few dozen to a few hundred nodes a side)
                int cur = 0;
                std::function<bool(int)> augment = [&](int i) {
                    for (int e = eptr[i]; e < eptr[i + 1]; ++e) {
                        const int j = eadj[e];
                        if (seen[j] == cur) continue;
                        seen[j] = cur;
                        if (mr[j] < 0 || augment(mr[j])) { ml[i] = j; mr[j] = i; return true; }
                    }
                    return false;
                };
                for (int i = 0; i < nl; ++i) { cur = i; augment(i); }
                std::vector<int> stk;
                // Z: reachable from the unmatched left nodes along alternating paths; cover = (left \ Z) + (right in Z)
                std::vector<char> zl(nl, 0), zr(nr, 0);
                stk.clear();
                for (int i = 0; i < nl; ++i) if (ml[i] < 0) { zl[i] = 1; stk.push_back(i); }
                while (!stk.empty()) {
                    const int i = stk.back(); stk.pop_back();
                    for (int e = eptr[i]; e < eptr[i + 1]; ++e) {
                        const int j = eadj[e];
                        if (zr[j] || ml[i] == j) continue;
                        zr[j] = 1;
                        if (mr[j] >= 0 && !zl[mr[j]]) { zl[mr[j]] = 1; stk.push_back(mr[j]); }
                    }
                }
                for (int i = 0; i < nl; ++i) if (!zl[i]) sep.push_back(sepL[i]);
                for (int j = 0; j < nr; ++j) if (zr[j]) sep.push_back(sepR[j]);
            }
            const int ss = ++stamp;
            for (int u : sep) side[u] = ss;
            std::vector<int> A, B;
            for (int u : L) if (side[u] != ss) A.push_back(u);
            for (int u : R) if (side[u] != ss) B.push_back(u);
            std::vector<int> ra, rb;
            if (par_min > 0 && depth < 2 && (int)std::min(A.size(), B.size()) >= par_min) {
                Rec wa = fork(), wb = fork();
                std::exception_ptr ex;
                bool threaded = true;
                std::thread th;
                try { th = std::thread([&] { try { ra = wa.run(std::move(A), depth + 1); } catch (...) { ex = std::current_exception(); } }); }
                catch (const std::system_error&) { threaded = false; }
                if (!threaded) ra = wa.run(std::move(A), depth + 1);
                try { rb = wb.run(std::move(B), depth + 1); }
                catch (...) { if (th.joinable()) th.join(); throw; }
                if (th.joinable()) th.join();
                if (ex) std::rethrow_exception(ex);
                ra = adopt(wa, std::move(ra));
                rb = adopt(wb, std::move(rb));
            } else {
                ra = run(std::move(A), depth + 1);
                rb = run(std::move(B), depth + 1);
            }
            ra.insert(ra.end(), rb.begin(), rb.end());
            if (sep.empty()) return ra;                            // the halves do not touch: two independent subtrees
            asort(sep, ax);                                        // along the cut, so that the chunks of a long separator are contiguous
            return {mk_chain(sep, std::move(ra))};
        }
    } rec{pos, ap, an, leaf_n, smax_n, vertex_cover, par_min, {}, std::vector<int>(n_nodes, 0), std::vector<int>(n_nodes, 0), std::vector<int>(n_nodes, 0), 0, {}};
    std::vector<int> regular, tail;
    for (int i = 0; i < n_nodes; ++i) (last && last[i] ? tail : regular).push_back(i);
    std::vector<int> roots = rec.run(std::move(regular), 0);
    std::vector<FH>& F = rec.F;
    if (!tail.empty()) {
        if (roots.size() == 1 && F[roots[0]].own.size() + tail.size() <= (size_t)smax_n) F[roots[0]].own.insert(F[roots[0]].own.end(), tail.begin(), tail.end());
        else { const int r = rec.mk_chain(tail, std::move(roots)); roots = {r}; }
    }
    const int nf = (int)F.size();
    if (nf == 0) return fail("empty tree");
    // ---- elimination order = front creation order (children are created before their parents), own order inside
    P.elim.assign(n_nodes, -1);
    {
        int k = 0;
        for (auto& f : F) for (int u : f.own) { if (P.elim[u] >= 0) return fail("node owned twice"); P.elim[u] = k++; }
        if (k != n_nodes) return fail("node not owned");
    }
    const std::vector<int>& elim = P.elim;
    std::vector<int> node_at(n_nodes);
    for (int u = 0; u < n_nodes; ++u) node_at[elim[u]] = u;
    P.pair_hi.resize(n_pairs); P.pair_lo.resize(n_pairs);
    for (int q = 0; q < n_pairs; ++q) {
        const int a = pairs[2 * q], b = pairs[2 * q + 1];
        if (elim[a] > elim[b]) { P.pair_hi[q] = a; P.pair_lo[q] = b; } else { P.pair_hi[q] = b; P.pair_lo[q] = a; }
    }
    // ---- boundaries (bottom-up), levels
    std::vector<std::vector<int>> bnd(nf);
    std::vector<int> level(nf, 0), parent(nf, -1), seen(n_nodes, -1);
    for (int f = 0; f < nf; ++f) {
        const int pmin = elim[F[f].own.front()], pmax = elim[F[f].own.back()];
        std::vector<int>& c = bnd[f];
        auto take = [&](int u) { if (seen[u] != f) { seen[u] = f; c.push_back(elim[u]); } };   // (each node once: elimination positions, plain integer keys)
        for (int u : F[f].own)
            for (int e = ap[u]; e < ap[u + 1]; ++e) if (elim[an[e]] > pmax) take(an[e]);
        for (int ch : F[f].ch) {
            if (ch >= f) return fail("child created after its parent");
            parent[ch] = f;
            level[f] = std::max(level[f], level[ch] + 1);
            for (int u : bnd[ch]) {
                if (elim[u] > pmax) take(u);
                else if (elim[u] < pmin) return fail("a child's boundary node is not in its parent");
            }
        }
        std::sort(c.begin(), c.end());
        for (int& u : c) u = node_at[u];
        if (c.size() * 3 + 1 > 30000 || F[f].own.size() > (size_t)smax_n) return fail("front too large");
    }
    for (int f = 0; f < nf; ++f)
        if (parent[f] < 0 && !bnd[f].empty()) return fail("a root with a boundary");
    // ---- flat arrays
    P.n_fronts = nf;
    P.fr.resize(nf);
    P.ent.reserve(2 * (size_t)n_nodes + (size_t)n_pairs);
    P.own.reserve(n_nodes);
    std::vector<int> where(n_nodes, -1);                           // node -> position in the front being laid out
    std::vector<int> owner(n_nodes, -1);                           // node -> the front that eliminates it
    for (int f = 0; f < nf; ++f) for (int u : F[f].own) owner[u] = f;
    for (int f = 0; f < nf; ++f) {
        NdFrontD& D = P.fr[f];
        const int ns = (int)F[f].own.size(), nbn = (int)bnd[f].size(), mn = ns + nbn;
        D.s = 3 * ns; D.b = 3 * nbn; D.level = level[f];
        D.own_off = (int)P.own.size(); D.bnd_off = (int)P.bnd.size();
        P.own.insert(P.own.end(), F[f].own.begin(), F[f].own.end());
        P.bnd.insert(P.bnd.end(), bnd[f].begin(), bnd[f].end());
        D.L_off = (int)P.L_doubles;
        P.L_doubles += (size_t)(D.s + D.b + 2 + D.s) * D.s;
        D.ldU = (D.b + 1 + 3) & ~3;
        D.U_off = (int)P.U_doubles;
        P.U_doubles += (size_t)(D.b + 1) * D.ldU;
        if (P.L_doubles > (size_t)1 << 30 || P.U_doubles > (size_t)1 << 30) return fail("factor too large");
        D.nR = (D.b + 1 + ND_TB - 1) / ND_TB;
        D.par = parent[f]; D.pslot = 0;
        D.pmap_off = (int)P.pmap.size();
        P.pmap.resize(P.pmap.size() + nbn + 1, (int16_t)-1);
        D.ldA = (D.s + D.b + 2) & ~1;
        D.A_off = (int)P.A_doubles;
        P.A_doubles += (size_t)F[f].ch.size() * (D.s + D.b + 1) * D.ldA;
        if (P.A_doubles > (size_t)1 << 30) return fail("assembly areas too large");
        D.seg_off = (int)P.seg.size() / 2; D.n_seg = 0;
        for (int i = 0; i < nbn; ++i) {
            const int o = owner[bnd[f][i]];
            if (i == 0 || o != owner[bnd[f][i - 1]]) { P.seg.push_back(o); P.seg.push_back(0); ++D.n_seg; }
            P.seg.back() = 3 * (i + 1);
        }
        D.ch_off = (int)P.child.size(); D.n_ch = (int)F[f].ch.size();
        P.child.insert(P.child.end(), F[f].ch.begin(), F[f].ch.end());
        for (int i = 0; i < ns; ++i) where[F[f].own[i]] = i;
        for (int i = 0; i < nbn; ++i) where[bnd[f][i]] = ns + i;
        // child maps: front node position (then the rhs slot) -> position in the child's boundary
        D.cmap_off = (int)P.cmap.size();
        for (size_t k = 0; k < F[f].ch.size(); ++k) {
            const int ch = F[f].ch[k];
            const size_t base = P.cmap.size();
            if (with_cmap) P.cmap.resize(base + mn + 1, (int16_t)-1);   // (the host reference's gather maps: the device writes through pmap)
            P.fr[ch].pslot = (int)k;                               // (children are laid out before their parents)
            for (size_t i = 0; i < bnd[ch].size(); ++i) {
                const int w = where[bnd[ch][i]];
                if (w < 0) return fail("a child's boundary node is missing from the parent front");
                if (with_cmap) P.cmap[base + w] = (int16_t)i;
                P.pmap[P.fr[ch].pmap_off + i] = (int16_t)w;
            }
            if (with_cmap) P.cmap[base + mn] = (int16_t)bnd[ch].size();
            P.pmap[P.fr[ch].pmap_off + bnd[ch].size()] = (int16_t)mn;
        }
        // original entries of the own columns
        D.ent_off = (int)P.ent.size();
        for (int i = 0; i < ns; ++i) {
            const int u = F[f].own[i];
            P.ent.push_back(NdEnt{(uint16_t)i, (uint16_t)i, (0u << ND_KIND_SHIFT) | (uint32_t)u});
            P.ent.push_back(NdEnt{(uint16_t)mn, (uint16_t)i, (2u << ND_KIND_SHIFT) | (uint32_t)u});
            for (int e = ap[u]; e < ap[u + 1]; ++e) {
                const int v = an[e];
                if (elim[v] < elim[u]) continue;
                if (where[v] < 0) return fail("a neighbour is missing from the front");
                P.ent.push_back(NdEnt{(uint16_t)where[v], (uint16_t)i, (1u << ND_KIND_SHIFT) | (uint32_t)aq[e]});
            }
        }
        D.n_ent = (int)P.ent.size() - D.ent_off;
        for (int u : F[f].own) where[u] = -1;
        for (int u : bnd[f]) where[u] = -1;
        P.max_s = std::max(P.max_s, D.s); P.max_b = std::max(P.max_b, D.b); P.max_ch = std::max(P.max_ch, D.n_ch);
        const double s = D.s, b = D.b + 1;
        P.flops += s * s * s / 3 + s * s * b + s * b * b;
    }
    for (int f = 0; f < nf; ++f) {
        NdFrontD& D = P.fr[f];
        D.pA_off = 0; D.pldA = 0;
        if (D.par >= 0) { const NdFrontD& Q = P.fr[D.par]; D.pldA = Q.ldA; D.pA_off = Q.A_off + D.pslot * (Q.s + Q.b + 1) * Q.ldA; }
    }
    // ---- levels and their workgroups
    P.n_levels = 1 + *std::max_element(level.begin(), level.end());
    P.lvl_ptr.assign(P.n_levels + 1, 0);
    for (int f = 0; f < nf; ++f) P.lvl_ptr[level[f] + 1]++;
    for (int l = 0; l < P.n_levels; ++l) P.lvl_ptr[l + 1] += P.lvl_ptr[l];
    P.lvl_fronts.resize(nf);
    {
        std::vector<int> fill(P.lvl_ptr.begin(), P.lvl_ptr.end() - 1);
        for (int f = 0; f < nf; ++f) P.lvl_fronts[fill[level[f]]++] = f;
    }
    P.lvl_wg_ptr.assign(P.n_levels + 1, 0);
    P.lvl_wg_split.assign(P.n_levels, 0);
    for (int l = 0; l < P.n_levels; ++l) {
        double worst = 0;
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {                                     // the workgroups that factorise a panel of their own
            const int f = P.lvl_fronts[i];
            for (int I = 0; I < P.fr[f].nR; ++I) { P.wg.push_back(f); P.wg.push_back(I); P.wg.push_back(I); }
            P.wg.push_back(f); P.wg.push_back(-1); P.wg.push_back(-1);                               // the inverse of L11 (device back pass)
            const double s = P.fr[f].s, tb = std::min(ND_TB, P.fr[f].b + 1);
            worst = std::max(worst, s * s * s / 3 + 2 * s * s * tb + s * tb * tb);
        }
        P.lvl_wg_split[l] = (int)P.wg.size() / 3;
        for (int i = P.lvl_ptr[l]; i < P.lvl_ptr[l + 1]; ++i) {                                     // ... and the off-diagonal Schur tiles
            const int f = P.lvl_fronts[i];
            for (int I = 0; I < P.fr[f].nR; ++I)
                for (int J = 0; J < I; ++J) { P.wg.push_back(f); P.wg.push_back(I); P.wg.push_back(J); }
        }
        P.flops_crit += worst;
        P.lvl_wg_ptr[l + 1] = (int)P.wg.size() / 3;
    }
    return true;
}

}  // namespace nrs
