// Host-side problem construction for the deformable BA (plain C++, no device code).
//
// Restates the edge-construction loops of the reference LocalDeformableBundleAdjustment
// (modules/optimization/g2o_optimization.cc:927-1137) on flat arrays: one landmark per
// (keyframe, map point), springs between graph neighbours inside a keyframe, dampers between a
// pair of neighbours in two consecutive keyframes.  The order of the emitted edges is the
// reference's insertion order, the counting rule (`n_regularizers > 10`, duplicates count) and the
// de-duplication keys (SpatialPoint / TemporalPoint, OPT:816-878) are kept, so the output is
// index-for-index what g2o would have been given.
#include <climits>
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <cstdlib>
#include <system_error>
#include <thread>
#include <vector>
#include "../../include/nrs.h"
namespace nrs { const char* process_debug_option(const char* name); }   // nrs_pose_only.hip: the process-wide snapshot of the NRS_* switches

namespace {
constexpr int kRegularizersPerPoint = 10;      // OPT:958

inline uint64_t pair_key(int32_t a, int32_t b) {
    const uint32_t lo = (uint32_t)(a < b ? a : b), hi = (uint32_t)(a < b ? b : a);
    return ((uint64_t)lo << 32) | hi;
}

// insert-only set of pair keys: open addressing, linear probing, cleared per keyframe.  (The
// node-based std::unordered_set was two thirds of the build time.)
struct PairSet {
    std::vector<uint64_t> slot;          // key + 1, 0 = empty
    uint64_t mask = 0;
    void reset(std::size_t expected) {
        std::size_t cap = 64;
        while (cap < 2 * expected) cap <<= 1;
        if (slot.size() != cap) slot.assign(cap, 0); else std::fill(slot.begin(), slot.end(), 0);
        mask = cap - 1;
    }
    bool insert(uint64_t key) {          // true if the key was not present
        const uint64_t v = key + 1;
        uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 17;
        for (;;) {
            uint64_t& s = slot[h & mask];
            if (s == 0) { s = v; return true; }
            if (s == v) return false;
            ++h;
        }
    }
};
}  // namespace

extern "C" int nrs_dba_build_edges(int32_t n_kf, const int32_t* kf_rowptr, const int32_t* kf_pt,
                                   int32_t n_points, const int32_t* nbr_rowptr, const int32_t* nbr_col,
                                   const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status,
                                   int32_t* n_spring, int32_t* sp_ij, float* sp_d0,
                                   int32_t* n_damper, int32_t* dm_idx, float* dm_w) {
    if (n_kf < 0 || n_points < 0 || !kf_rowptr || !nbr_rowptr || !n_spring || !n_damper) return NRS_ERR_INVALID;
    if (n_kf > 0 && kf_rowptr[n_kf] > 0 && !kf_pt) return NRS_ERR_INVALID;
    if (n_points > 0 && nbr_rowptr[n_points] > 0 && (!nbr_col || !nbr_w || !nbr_d0 || !nbr_status)) return NRS_ERR_INVALID;
    const bool fill = sp_ij != nullptr;
    if (fill && (!sp_d0 || !dm_idx || !dm_w)) return NRS_ERR_INVALID;
    const int32_t cap_s = *n_spring, cap_d = *n_damper;
    for (int32_t i = 0; i < kf_rowptr[n_kf]; ++i)
        if (kf_pt[i] < 0 || kf_pt[i] >= n_points) return NRS_ERR_INVALID;
    for (int32_t i = 0; i < (n_points > 0 ? nbr_rowptr[n_points] : 0); ++i)    // both edge loops index cur[] / nxt[] with these
        if (nbr_col[i] < 0 || nbr_col[i] >= n_points) return NRS_ERR_INVALID;

    // A keyframe's edges depend on its own landmarks and the next keyframe's only, so keyframes are independent: a few host
    // threads take contiguous ranges of keyframes.  Pass 1 counts per keyframe, pass 2 (fill) writes at the prefix offsets:
    // the output is index-for-index the sequential one for any thread count.
    struct Work {
        std::vector<int32_t> cur, nxt;       // inserted_landmarks[kf][mappoint] -> landmark index (OPT:927-952): two rows suffice
        PairSet spring_seen, damper_seen;
        bool overflow = false;               // an edge did not fit the caller's arrays
    };
    auto keyframe = [&](Work& w, int k, bool emit, int64_t ns0, int64_t nd0, int64_t& ns_out, int64_t& nd_out) {
        auto load = [&](std::vector<int32_t>& row, int kk) {
            for (int32_t i = kf_rowptr[kk]; i < kf_rowptr[kk + 1]; ++i) row[kf_pt[i]] = i;
        };
        auto clear = [&](std::vector<int32_t>& row, int kk) {
            for (int32_t i = kf_rowptr[kk]; i < kf_rowptr[kk + 1]; ++i) row[kf_pt[i]] = -1;
        };
        std::vector<int32_t>&cur = w.cur, &nxt = w.nxt;
        const bool has_next = k + 1 < n_kf;
        load(cur, k);
        if (has_next) load(nxt, k + 1);
        int64_t ns = ns0, nd = nd0;
        // keys carry the keyframe id: a per-keyframe set is equivalent (<= 11 insertions per point)
        w.spring_seen.reset((std::size_t)(kf_rowptr[k + 1] - kf_rowptr[k]) * 12);
        w.damper_seen.reset((std::size_t)(kf_rowptr[k + 1] - kf_rowptr[k]) * 12);
        for (int32_t l = kf_rowptr[k]; l < kf_rowptr[k + 1]; ++l) {
            const int32_t p = kf_pt[l];
            const int32_t lo = nbr_rowptr[p], hi = nbr_rowptr[p + 1];
            int n_reg = 0;
            for (int32_t e = lo; e < hi; ++e) {                                     // OPT:1033-1074
                if (n_reg > kRegularizersPerPoint || nbr_status[e] == NRS_GRAPH_BAD) break;
                const int32_t o = nbr_col[e];
                if (cur[o] < 0) continue;
                if (!w.spring_seen.insert(pair_key(p, o))) { ++n_reg; continue; }
                if (emit && ns >= cap_s) w.overflow = true;
                else if (emit) {
                    sp_ij[2 * ns] = l;
                    sp_ij[2 * ns + 1] = cur[o];
                    sp_d0[ns] = nbr_d0[e];
                }
                ++ns;
                ++n_reg;
            }
            if (has_next) {                                                          // OPT:1076-1136
                if (nxt[p] < 0) continue;
                const int32_t ln = nxt[p];
                n_reg = 0;
                for (int32_t e = lo; e < hi; ++e) {
                    if (n_reg > kRegularizersPerPoint || nbr_status[e] == NRS_GRAPH_BAD) break;
                    const int32_t o = nbr_col[e];
                    if (cur[o] < 0 || nxt[o] < 0) continue;
                    if (!w.damper_seen.insert(pair_key(p, o))) { ++n_reg; continue; }
                    if (emit && nd >= cap_d) w.overflow = true;
                    else if (emit) {
                        dm_idx[4 * nd] = l;
                        dm_idx[4 * nd + 1] = cur[o];
                        dm_idx[4 * nd + 2] = ln;
                        dm_idx[4 * nd + 3] = nxt[o];
                        dm_w[nd] = nbr_w[e];
                    }
                    ++nd;
                    ++n_reg;
                }
            }
        }
        clear(cur, k);
        if (has_next) clear(nxt, k + 1);
        ns_out = ns - ns0; nd_out = nd - nd0;
    };
    int nt = 1;
    if (n_kf > 1 && kf_rowptr[n_kf] >= 20000) {
        nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* ev = nrs::process_debug_option("NRS_HOST_THREADS")) nt = std::max(1, std::min(64, atoi(ev)));
        nt = std::min(nt, n_kf);
    }
    std::vector<int64_t> cs((size_t)n_kf + 1, 0), cd((size_t)n_kf + 1, 0);
    bool overflow = false;
    auto run = [&](bool emit, bool offsets_known) {
        auto body = [&](int ti) {
            Work w;
            w.cur.assign(n_points, -1); w.nxt.assign(n_points, -1);
            const int k0 = (int)((int64_t)n_kf * ti / nt), k1 = (int)((int64_t)n_kf * (ti + 1) / nt);
            for (int k = k0; k < k1; ++k) {
                int64_t a = 0, b2 = 0;
                keyframe(w, k, emit, cs[k], cd[k], a, b2);
                if (!offsets_known) { cs[k + 1] = cs[k] + a; cd[k + 1] = cd[k] + b2; }      // (one thread: running offsets)
            }
            if (w.overflow) overflow = true;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) {
            try { th.emplace_back(body, t); } catch (const std::system_error&) { body(t); }      // no thread to be had: this share runs here
        }
        body(0);
        for (auto& x : th) x.join();
    };
    if (nt == 1) run(fill, false);                                  // one pass, offsets accumulate as it goes
    else {
        // counts per keyframe (every thread starts its range at 0: cs[k + 1] - cs[k] is what matters), prefix, fill
        auto count_body = [&](int ti) {
            Work w;
            w.cur.assign(n_points, -1); w.nxt.assign(n_points, -1);
            const int k0 = (int)((int64_t)n_kf * ti / nt), k1 = (int)((int64_t)n_kf * (ti + 1) / nt);
            for (int k = k0; k < k1; ++k) keyframe(w, k, false, 0, 0, cs[k + 1], cd[k + 1]);
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) {
                try { th.emplace_back(count_body, t); } catch (const std::system_error&) { count_body(t); }
            }
            count_body(0);
            for (auto& x : th) x.join();
        }
        for (int k = 0; k < n_kf; ++k) { cs[k + 1] += cs[k]; cd[k + 1] += cd[k]; }
        if (fill) run(true, true);
    }
    if (overflow) return NRS_ERR_INVALID;
    const int64_t ns = cs[n_kf], nd = cd[n_kf];
    if (ns > INT32_MAX || nd > INT32_MAX) return NRS_ERR_INVALID;
    *n_spring = (int32_t)ns;
    *n_damper = (int32_t)nd;
    return NRS_OK;
}

// ---- N2b: the embedded form of the window (include/nrs.h).  Vertices = the keyframe copies of the NODES; the walks of OPT:1033-1136
// run between node copies only (an observed point that is not a node is passed over like one the keyframe does not observe), and
// every other observed point is bound, by the same walk, to the <= 11 node copies of its keyframe it accepts, with the connection
// weights normalised (float32 weights widened, summed and divided in float64, in walk order).  oracle/embedded_oracle.py
// dba_build_embedded states it; with every point a node the lists are nrs_dba_build_edges', index for index.
extern "C" int nrs_dba_build_edges_embedded(int32_t n_kf, const int32_t* kf_rowptr, const int32_t* kf_pt, int32_t n_points, const uint8_t* is_node,
                                            const int32_t* nbr_rowptr, const int32_t* nbr_col, const float* nbr_w, const float* nbr_d0, const int32_t* nbr_status,
                                            int32_t* n_lm, int32_t* lm_obs, int32_t* n_spring, int32_t* sp_ij, float* sp_d0,
                                            int32_t* n_damper, int32_t* dm_idx, float* dm_w,
                                            int32_t* n_skin, int32_t* sk_obs, int32_t* sk_node, double* sk_omega) {
    if (n_kf < 0 || n_points < 0 || !kf_rowptr || !nbr_rowptr || !is_node || !n_lm || !n_spring || !n_damper || !n_skin) return NRS_ERR_INVALID;
    if (n_kf > 0 && kf_rowptr[n_kf] > 0 && !kf_pt) return NRS_ERR_INVALID;
    if (n_points > 0 && nbr_rowptr[n_points] > 0 && (!nbr_col || !nbr_w || !nbr_d0 || !nbr_status)) return NRS_ERR_INVALID;
    const bool fill = lm_obs != nullptr;
    if (fill && (!sp_ij || !sp_d0 || !dm_idx || !dm_w || !sk_obs || !sk_node || !sk_omega)) return NRS_ERR_INVALID;
    const int64_t cap_l = *n_lm, cap_s = *n_spring, cap_d = *n_damper, cap_k = *n_skin;
    for (int32_t i = 0; i < kf_rowptr[n_kf]; ++i)
        if (kf_pt[i] < 0 || kf_pt[i] >= n_points) return NRS_ERR_INVALID;
    for (int32_t i = 0; i < (n_points > 0 ? nbr_rowptr[n_points] : 0); ++i)
        if (nbr_col[i] < 0 || nbr_col[i] >= n_points) return NRS_ERR_INVALID;
    // node copies are numbered keyframe by keyframe in observation order: first offsets per keyframe
    std::vector<int32_t> first((size_t)n_kf + 1, 0);
    for (int k = 0; k < n_kf; ++k) {
        int32_t m = 0;
        for (int32_t i = kf_rowptr[k]; i < kf_rowptr[k + 1]; ++i) m += is_node[kf_pt[i]] ? 1 : 0;
        first[k + 1] = first[k] + m;
    }
    std::vector<int32_t> cur(n_points, -1), nxt(n_points, -1);
    auto load = [&](std::vector<int32_t>& row, int kk) {
        int32_t l = first[kk];
        for (int32_t i = kf_rowptr[kk]; i < kf_rowptr[kk + 1]; ++i) if (is_node[kf_pt[i]]) row[kf_pt[i]] = l++;
    };
    auto clear = [&](std::vector<int32_t>& row, int kk) {
        for (int32_t i = kf_rowptr[kk]; i < kf_rowptr[kk + 1]; ++i) row[kf_pt[i]] = -1;
    };
    PairSet spring_seen, damper_seen;
    int64_t nl = 0, ns = 0, nd = 0, nk = 0;
    bool overflow = false;
    for (int k = 0; k < n_kf; ++k) {
        const bool has_next = k + 1 < n_kf;
        load(cur, k);
        if (has_next) load(nxt, k + 1);
        spring_seen.reset((std::size_t)(kf_rowptr[k + 1] - kf_rowptr[k]) * 12);
        damper_seen.reset((std::size_t)(kf_rowptr[k + 1] - kf_rowptr[k]) * 12);
        for (int32_t ob = kf_rowptr[k]; ob < kf_rowptr[k + 1]; ++ob) {
            const int32_t p = kf_pt[ob];
            const int32_t lo = nbr_rowptr[p], hi = nbr_rowptr[p + 1];
            if (!is_node[p]) {                                                       // a skinned observation: its walk accepts node copies
                int n_reg = 0;
                int32_t nodes[11];
                double w[11], tot = 0.0;
                for (int32_t e = lo; e < hi; ++e) {
                    if (n_reg > kRegularizersPerPoint || nbr_status[e] == NRS_GRAPH_BAD) break;
                    const int32_t o = nbr_col[e];
                    if (cur[o] < 0) continue;
                    nodes[n_reg] = cur[o]; w[n_reg] = (double)nbr_w[e];
                    ++n_reg;
                }
                if (n_reg == 0) continue;                                            // (no node copy within reach: the observation constrains nothing)
                for (int q = 0; q < n_reg; ++q) tot += w[q];
                if (fill && nk >= cap_k) overflow = true;
                else if (fill) {
                    sk_obs[nk] = ob;
                    for (int q = 0; q < 11; ++q) { sk_node[11 * nk + q] = q < n_reg ? nodes[q] : -1; sk_omega[11 * nk + q] = q < n_reg ? w[q] / tot : 0.0; }
                }
                ++nk;
                continue;
            }
            const int32_t l = cur[p];
            if (fill && nl >= cap_l) overflow = true; else if (fill) lm_obs[nl] = ob;
            ++nl;
            int n_reg = 0;
            for (int32_t e = lo; e < hi; ++e) {                                     // OPT:1033-1074 between node copies
                if (n_reg > kRegularizersPerPoint || nbr_status[e] == NRS_GRAPH_BAD) break;
                const int32_t o = nbr_col[e];
                if (cur[o] < 0) continue;
                if (!spring_seen.insert(pair_key(p, o))) { ++n_reg; continue; }
                if (fill && ns >= cap_s) overflow = true;
                else if (fill) { sp_ij[2 * ns] = l; sp_ij[2 * ns + 1] = cur[o]; sp_d0[ns] = nbr_d0[e]; }
                ++ns; ++n_reg;
            }
            if (has_next && nxt[p] >= 0) {                                           // OPT:1076-1136
                const int32_t ln = nxt[p];
                n_reg = 0;
                for (int32_t e = lo; e < hi; ++e) {
                    if (n_reg > kRegularizersPerPoint || nbr_status[e] == NRS_GRAPH_BAD) break;
                    const int32_t o = nbr_col[e];
                    if (cur[o] < 0 || nxt[o] < 0) continue;
                    if (!damper_seen.insert(pair_key(p, o))) { ++n_reg; continue; }
                    if (fill && nd >= cap_d) overflow = true;
                    else if (fill) { dm_idx[4 * nd] = l; dm_idx[4 * nd + 1] = cur[o]; dm_idx[4 * nd + 2] = ln; dm_idx[4 * nd + 3] = nxt[o]; dm_w[nd] = nbr_w[e]; }
                    ++nd; ++n_reg;
                }
            }
        }
        clear(cur, k);
        if (has_next) clear(nxt, k + 1);
    }
    if (overflow || nl > INT32_MAX || ns > INT32_MAX || nd > INT32_MAX || nk > INT32_MAX) return NRS_ERR_INVALID;
    *n_lm = (int32_t)nl; *n_spring = (int32_t)ns; *n_damper = (int32_t)nd; *n_skin = (int32_t)nk;
    return NRS_OK;
}

// ---- N1 (include/nrs.h nrs_debug_nd_solve): helpers of the nested-dissection plan shared with the engine
#include "nrs_nd_plan.hpp"

namespace nrs {
// pair blocks as the caller gives them (rows = first node of the pair) -> the plan's orientation (rows = the later-eliminated node)
void nd_orient_pairs(const NdPlan& P, const int32_t* pairs, const double* Vp, std::vector<double>& out) {
    out.resize(9 * (size_t)P.n_pairs);
    for (int q = 0; q < P.n_pairs; ++q) {
        const bool flip = P.pair_hi[q] != pairs[2 * q];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) out[9 * (size_t)q + 3 * a + b] = flip ? Vp[9 * (size_t)q + 3 * b + a] : Vp[9 * (size_t)q + 3 * a + b];
    }
}
void nd_stats(const NdPlan& P, int64_t* stats) {
    if (!stats) return;
    stats[0] = P.n_fronts; stats[1] = P.n_levels; stats[2] = P.max_s; stats[3] = P.max_b;
    stats[4] = (int64_t)P.L_doubles; stats[5] = (int64_t)P.U_doubles; stats[6] = (int64_t)P.flops; stats[7] = (int64_t)P.wg.size() / 3;
}
}  // namespace nrs

