// Types, constants and device helpers of the graph-LM engine (included by nrs_engine.hip only).
// Part of nrs_engine.hip (one translation unit); see that file's header for the design.
#pragma once

namespace nrs {

constexpr int ROW_ALIGN = 256;       // pose row padding; also rows per k_reproj workgroup
constexpr int BLK = 256;             // threads per workgroup everywhere
constexpr int NPART = 12;            // per-block partial slots of the SpMV kernel: [0..2] dots, [3..8] pose sums
constexpr int SK_MAX = 11;           // embedded mode (nrs_engine_skin.hpp): nodes per skinned observation (the walk of OPT:255-279 accepts 11)
constexpr int SK_RL = 16;            // ... lanes per node-row list of the PCG form, lists per workgroup
constexpr int SK_RPB = BLK / SK_RL;
constexpr int CO_MAX = 84;           // largest coarse system of the two-level preconditioner (fits one workgroup's LDS)
constexpr int CO_GMAX = (CO_MAX - 6) / 3;   // row groups of the coarse level
static_assert(CO_GMAX <= 32, "k_coarse_tile keeps the reached groups in a 32-bit mask");

// incidence meta bits
constexpr int SM_COUNT = 1 << 30;    // spring: this incidence adds the edge's rho to chi2
constexpr int SM_ACTIVE = 1 << 29;   // spring: edge is at level 0
constexpr int DM_COUNT = 1 << 2;     // damper: bits 0-1 role
constexpr int DM_ACTIVE = 1 << 3;
constexpr int DM_UNARY = 1 << 4;     // damper: the other vertex is a value, not a variable

// Incidence data of the LDS-staged path: structure-of-arrays streams in sliced-ELL order, every load / store of a
// wave one contiguous run; neighbour ids are tile-local (own rows, then halo).
//   springs : s_om  u32 {other u16 | meta u16 << 16}      static
//             s_d0  f32 rest length                          static, lineariser only
//             s_qc  f64 rho' Omega cg^2                      written by the lineariser (full-width stores), read by the operator
//   dampers : d_hdr 2 x u32 {o0 | o1 << 16, o2 | meta << 16} static
//             d_w   f32 edge weight                          static, lineariser only
//             d_s   f64 rho' Omega w^2                       written by the lineariser (full-width stores), read by the operator
// Every load is unconditional, so a chunk's loads are all in flight before anything waits on them.  (Tried and
// dropped: re-forming s = Omega w w in the operator from the 4-byte weight and storing s only for Huber-active edges
// -- 25 % less damper traffic, but the conditional load needs the weight first, which serialises the record
// prefetch: operator 25 -> 32 us on C2.)
// The operator is applied in factored form: a spring block is qc * v v^T with v = x_i - x_j re-formed from the staged
// linearisation point, a damper block is +-s I3, a reprojection block is J^T w J with J rebuilt from the fp32
// projection Jacobian kept per row.  SpringRec / DamperRec are the register views the kernels work on.
struct SpringRec { double qc; uint16_t other, meta; };
struct __attribute__((aligned(16))) RowRec { float J[6]; double w; };                           // 32 B
struct DamperRec { uint16_t o0, o1, o2, meta; double s; };
constexpr uint16_t REC_NONE = 0xFFFF;
constexpr uint16_t SR_ACTIVE = 1, SR_COUNT = 2;

// edge lists of the chi2-only evaluation (BA form): every edge once, rows instead of vertex ids
struct __attribute__((aligned(16))) EcSpring { int a, b; float d0; int pad; };
struct __attribute__((aligned(16))) EcDamper { int r[4]; };         // roles 1c 2c 1n 2n, -1 = absent

struct Dev {
    int K, M, n_rows, n_groups;      // poses, vertices, padded rows, ROW_ALIGN groups
    int row_lo, row_hi;              // the rows whose per-row arrays this engine holds ([0, n_rows) on one GPU; a rank of a sharded window: its own keyframes
                                     // and one ghost keyframe either side -- the arrays are addressed by the global row index all the same, see ArenaPlan::get_rows)
    int T;                           // lanes per row
    int n_sp, n_dm, n_un;
    int n_regblk, n_vecblk;
    Cam cam;
    double info_reproj, delta_reproj, info_pos, delta_pos, info_spatial, delta_spatial, k_spring;
    int spring_form;
    // rows
    int* grp_pose;                   // n_groups -> pose index
    int* pose_grp_ptr;               // K+1 -> group ranges
    uint8_t* rflag;                  // n_rows
    uint8_t* pose_fixed;             // K
    float* uv;                       // n_rows x 2
    double* X0;                      // n_rows x 3 or null
    // incidences (sliced ELL)
    int* ss_ptr; int ss_nnz; int* sd_ptr; int sd_nnz;
    int* s_other; float* s_d0; int* s_meta;
    int* d_o0; int* d_o1; int* d_o2; float* d_w; int* d_meta;
    // LDS staging: neighbour ids above are LOCAL to the workgroup's tile: [0, tile_rows) = own rows,
    // tile_rows + i = halo_rows[halo_ptr[b] + i]
    int use_lds, tile_rows, max_halo;
    int max_halo_s;                  // halo lists start with the spring (same-keyframe) neighbours: at most this many
    // tiles come in two classes so that a few tiles with very large halos do not set the LDS size
    // (= occupancy) of all: tile_list = class-0 tiles, then class-1 tiles; caps per class
    int* tile_list; int n_tiles_cls[2]; int cap_h[2], cap_s[2];
    int* halo_ptr; int* halo_rows; int* halo_ns;   // halo_ns[b] = number of spring-halo rows of tile b
    uint32_t* s_om; double* s_qc; uint2* d_hdr;      // (s_d0, d_w, d_s: above; sized by the incidence count in both paths)
    // Temporal-difference form of the dampers (BA windows on the two-kernel path).  A reference damper joins the SAME two
    // map points in two consecutive keyframes, (1c, 2c, 1n, 2n) with 1n = next(1c), 2n = next(2c) (OPT:1076-1136), so
    // with G^f_r = v_r - v_next(r) and G^b_r = v_r - v_prev(r) every incidence is
    //     a_i += s (G^d_i - G^d_o),  o = the partner row in the same keyframe, d = forward for roles 1c / 2c, backward for 1n / 2n.
    // The kernels stage v, G^f, G^b for the tile and its same-keyframe halo (the temporal partners are read from global
    // memory once per staged row): one LDS vector per damper incidence instead of three, a 4-byte header instead of 8,
    // and a halo of ~190 rows instead of ~520 (the rows of the two adjacent keyframes are no longer part of it).
    int dform;
    uint32_t* d_om;                  // dform: {partner u16 | meta u16 << 16} per damper incidence
    int* nxt_row; int* prv_row;      // dform: row of the same map point in the next / previous keyframe (-1: none)
    int* halo_nxt; int* halo_prv;    // dform: the same for the halo rows (indexed like halo_rows)
    // plain BA windows: a damper's second vertex in canonical order (o1) is the row's OWN temporal partner -- the same map point in
    // the next (roles 1c / 2c) or previous (1n / 2n) keyframe, OPT:1076-1136 -- hence the same for all dampers of a row and
    // direction: row_tp[row] = {tile-local id of next | of prev << 16} (REC_NONE: none) lets the kernels read it once per row
    uint32_t* row_tp; int tp_ok;
    int rc;                          // plain two-kernel path with h4: bit 0 / bit 1 = the spring / damper factors are not stored -- the operator re-forms them from the
                                     // staged linearisation point (spring_qc / damper_s, nrs_engine_linearize.hpp): 8 instead of 12 bytes per incidence there, no factor stores in the lineariser
    int nt;                          // h4 path: the incidence streams (12 bytes per slot) exceed the Infinity Cache -- read / written once per launch with the non-temporal hint
                                     // (k_lin_plain C4 2.19 -> 2.10 ms; NRS_NT=0 / 1 overrides)
    uint32_t* d_h4; int h4;          // plain two-kernel path with cached temporal partners: 4-byte damper headers {o0 : 12 | o2 : 12 | meta : 8} derived from d_hdr
                                     // (the partner o1 is the row's own and tile-local ids stay below 4096): 8 instead of 12 bytes per damper incidence in both kernels
    uint32_t* row_cnt;               // plain windows: {spring incidences | damper incidences << 16} of every row: a lane's slots beyond its share are padding
                                     // and are neither requested nor stored (a fifth of all slots on the benchmark windows)
    RowRec* rowrec;                  // n_rows (LDS path): reprojection factors of the linearisation point
    Pose* lin_pose; double* lin_xl;  // the linearisation point itself (= pose[cur], xl[cur])
    // state (two copies: current / trial, swapped on accept)
    Pose* pose[2]; double* xl[2];
    Pose* pose_init; double* xl_init;
    // linearisation
    double* D;                       // n_rows x 6   (xx xy xz yy yz zz)
    double* Hpl;                     // 18 x n_rows  (component-major; gather fallback path only)
    double* s_g;                     // 3 x nnz_s
    double* d_s;                     // nnz_d
    double* Hpp;                     // K x 21
    double* bp; double* bl;          // 6K, 3 n_rows
    double* Dinv; double* Hppinv;    // n_rows x 6, K x 36
    // PCG vectors: pose part [6K] and row part [3 n_rows]
    double *xp, *rp, *up, *pp, *sp, *wp;
    double *xv, *rv, *uv3, *pv, *sv, *wv;
    // second halves of the ping-pong pairs used by the fused small-problem iteration
    double *rp2, *sp2, *up2, *rv2, *sv2, *wv2, *part_spmv2;
    int fused;
    // two-level preconditioner of the fused path (single pose): coarse unknowns = one translation per
    // 256-row group + the pose; M^-1 = block-Jacobi + Z (Z^T H Z + lambda Z^T Z)^-1 Z^T
    int coarse, co_n;                // enabled, number of coarse unknowns (3 n_groups + 6)
    double* co_ct;                   // n_regblk x n_groups x 6: sum of H_ij over i in tile, j in each row group
    double* co_cp;                   // n_regblk x 18: sum of H_lp over the tile's rows (3x6)
    double* co_tb;                   // n_regblk x 4: sum of b (3) and number of free rows
    double* part_ts; double* part_ts2;   // 9 x n_regblk (component-major, ping-pong): tile sums of r, s, w
    double* co_bt;                   // n_regblk x 6: sum of H_ij over i, j in the tile (tile-level diagonal block)
    double* co_bti;                  // n_regblk x 6: (B_t + lambda n_t I)^-1 of the current trial (0 if not positive)
    double* co_c0;                   // co_n x co_n: Z^T H Z ; co_nn: Z^T Z diagonal ; co_bc: Z^T b
    double* co_nn; double* co_bc;
    double* co_inv;                  // co_n x co_n: (C0 + lambda N)^-1 of the current trial
    double* co_y0;                   // co_n: its product with Z^T b (start vector of the trial)
    int* tile_desc;                  // fused path: 8 ints per tile {pose, first tile of pose, end tile of pose, halo begin, halo count, 0,0,0}
    int* halo_fix;                   // fused path: BLK ints per tile = the first BLK halo rows (fixed stride: no pointer chase);
                                     // plain two-kernel path: HALO_FIX ints per tile, -1 behind the list's end (stage_rows<true>)
    // large problems: the SpMV partials are pre-reduced by k_reduce_partials (one launch) instead of
    // being re-summed by every workgroup of the update kernel (which is O(workgroups^2) reads)
    int hier;
    // two-kernel path: r.u of the next iteration is produced by the vector update (and by the trial
    // setup), so the operator kernel can see convergence BEFORE applying the operator once more
    int ecd; double* part_ru;        // 2 x n_vecblk (ping-pong by iteration parity)
    double* red;                     // [0..2] r.u, w.u, cross ; [3 + 6k + a] pose sums
    // partials / scalars
    double* part_lin;                // lin_slots x 32  (lineariser: 27 pose sums + chi; per tile on the LDS path, else per group)
    double* part_rchi;               // n_groups        (chi2-only reprojection pass)
    double* part_pchi;               // K               (plain windows: chi2 of a pose's slices, written by k_pose_sums)
    int lin_rb;                      // part_lin slots per ROW_ALIGN group (tiles per group on the LDS path, else 1)
    double* part_reg;                // n_regblk x 2    (chi, maxdiag)
    double* part_spmv;               // n_regblk x NPART
    double* part_apply;              // n_vecblk
    double* scal;
    // chi2 of a trial state, edge-parallel (BA windows without masks: every spring / damper is evaluated
    // once from these lists instead of from the incidence records of the row that counts it)
    int ec_on, ec_nsp, ec_ndm, ec_nblk;
    long long* dbg_clk; int one_xcd;   // (one_xcd: every tile of a small fused problem on the workgroups of one XCD: pcg_enqueue_batch)              // NRS_LIN_DBG: per-wave phase clocks of one k_lin_plain launch (8 per slice), else null
    int plain;                       // plain BA window on the LDS path: the lineariser is k_lin_plain (nrs_engine_linearize.hpp)
    EcSpring* ec_sp; EcDamper* ec_dm; float* ec_w; double* part_ec;
    double* h_scal; int* h_flags;     // host-mapped mirrors, written by k_finalize / k_publish (no copy kernels)
    int* flags;                      // [0] pcg done, [1] pcg iterations, [2] nan flag
    // shard window (multi-GPU): the layout is the whole problem on every rank, a rank
    // launches the row kernels over its own contiguous range of poses only.  Unsharded: the full ranges.
    int sh_on, sh_rank, sh_world;
    int sh_lead;                     // this rank counts the pose-level terms of replicated sums (rank 0)
    int sh_k0, sh_nk;                // poses
    int sh_g0, sh_ng;                // ROW_ALIGN groups
    int sh_vb0, sh_nvb;              // BLK-row vector blocks
    int sh_t0[2], sh_nt[2];          // per tile class: first entry (relative to the class) and count in tile_list
    // a launch of the operator may cover two pieces of a class list: [sh_t0, sh_t0 + sh_nt) then [sh_t0b, sh_t0b + sh_ntb)
    // (the boundary tiles at both ends of a rank's range); sh_ntb = 0 otherwise
    int sh_t0b[2], sh_ntb[2];
    // of the rank's own tiles in each class: the first sh_front and the last sh_back read rows of other ranks
    int sh_front[2], sh_back[2];
    double* red_loc;                 // this rank's SpMV sums before the all-reduce into red
    // embedded-deformation mode (N2, nrs_engine_skin.hpp): sk_n SKINNED observations -- points without a vertex whose position is
    // X0 + sum_k om[k] x[row[k]] over <= 11 node rows; their reprojection edges constrain those rows and the pose (direct solver only)
    int sk_n, sk_nblk;
    const float* sk_uv; const double* sk_X0; const int* sk_row; const double* sk_om; const uint8_t* sk_active;   // (sk_row / sk_om: 11 x sk_n, node-slot-major)
    double* sk_rec;                  // sk_n x 27: per observation J_l^T w J_l (6), -J_l^T w r (3), J_p^T w J_l (18) of the linearisation point
    double* sk_part;                 // sk_nblk x 32: H_pp (21), b_p (6), chi2 partials of the observations' workgroups
    double* sk_chi;                  // sk_n: r^T Omega r at the evaluated state (the drivers' inlier classification)
    double* sk_maxdiag;              // largest diagonal entry of the blocks the observations add to (joins SC_MAXDIAG)
    // embedded BA windows (N2b: K poses, PCG path -- k_skin_rows / k_skin_pose / k_skin_op / k_skin_op_rows).  Observations are grouped by
    // pose, every pose's padded to BLK slots (a workgroup of k_skin / k_skin_op serves ONE pose); per node row the observations that
    // reach it in slot order with their weights (fixed-order gathers: no atomics on values, bit-reproducible)
    int sk_pcg;                      // 1: the observations act through the PCG operator (not through the direct solver's blocks)
    const int* sk_blk_pose;          // sk_nblk: pose of a block's observations (null: pose 0 -- the single-frame engines)
    const int* sk_pose_blk;          // K + 1: block range of every pose
    const double* sk_base;           // BA form: a skinned point sits at X0 + sum om (x - x_start), x_start = xl_init (null: X0 + sum om x)
    int sk_nrl;                      // node rows with a list
    const int* sk_rl_row; const int* sk_rl_ptr; const int* sk_rl_obs; const double* sk_rl_om;   // list j: row, entries [ptr[j], ptr[j+1]): observation slot, weight
    double* sk_recT;                 // 24 x sk_n: the operator's part of sk_rec (A: 6, B: 18), value-major (written when sk_pcg)
    double* sk_g;                    // sk_n x 4: per observation A s + B^T u_p (3 values in a 32-byte slot) of the current PCG direction u
    double* sk_opart;                // sk_nblk x 8: sums over a block's observations of B s (6: the pose rows of H u), of (B^T u_p).s (the cross term u_l.(H_lp u_p)) and of g.s (their share of w.u)
    const int* sk_row_q;             // n_rows x 2: a row's range in sk_rl_obs / sk_rl_om (empty: no observation reaches it)
    double* D_op;                    // gather path (use_lds = 0) of an embedded problem: the rows' diagonal blocks as the lineariser left them, WITHOUT the observations'
                                     // share k_skin_rows adds to D for the preconditioner -- k_spmv applies these (k_skin_op / the row pass apply A_o whole); null: D
    double* pk; double* pk_loc;      // evaluation packet: [0] chi2 [1] scale [2..2+world) max diag per rank, then K x 27 (H_pp, b_p)
};

// factor recomputation of a tile class (Dev::rc): with the damper part on, the operator stages the positions of the whole halo,
// which has to fit a workgroup's 64 KB -- a class of outlier halos keeps its dampers' factors stored
inline int rc_of(const Dev& d, int cls) {
    if (!(d.rc & 2)) return d.rc;
    return sizeof(double) * 3 * (size_t)(2 * (d.tile_rows + d.cap_h[cls]) + 2) > 64 * 1024 - 512 ? (d.rc & 1) : d.rc;
}

enum { SC_CHI = 0, SC_MAXDIAG = 1, SC_SCALE = 2, SC_GAMMA0 = 3, SC_SLOT0 = 4, SC_SLOT1 = 6, SC_N = 16 };

// Speculative LM trials of a directly solved single-frame engine (a2; engine_optimize).  Rejections come in runs there -- the damping
// grows x 2, x 4, x 8 ... until a step is accepted, four trials in a row as a rule -- and one trial is a chain of a dozen launches that
// keeps a few dozen workgroups busy.  After a rejected trial the next trials of the run (the same damping sequence g2o would walk)
// are factorised, solved and evaluated SIDE BY SIDE on streams of their own, each on a shadow set of everything a trial writes: step
// vectors, trial state, partial sums, scalars, status words and their host mirrors (here) and the solver's factor storage, assembly
// areas and unknowns (NdSolver).  The host then reads the results in order; the first accepted one becomes the state (pointer swap),
// the rest are discarded.  Same trials, same arithmetic, same bits as one at a time.
constexpr int SPEC_MAX = 3;          // shadow sets: up to 1 + SPEC_MAX trials in flight
struct SpecSet {
    double *xv, *xp, *part_apply, *part_rchi, *part_reg, *scal;
    double *sk_part, *sk_chi;        // embedded mode (the context's skin buffer)
    int* flags;
    int* abort;                      // id of the set's solve that is not needed any more (NdDev::abort)
    Pose* pose; double* xl;          // the trial state (swapped with the engine's on acceptance)
    double* h_scal; int* h_flags;    // mapped host mirrors of its own (nrs_ctx::pin_spec_*)
};

struct NdEngine;                     // nrs_engine_nd.hpp: the direct solver of a single-frame engine
struct KftHost;                      // nrs_engine_kft.hpp: the keyframe-block factorisation of an embedded BA window
struct Engine {
    Dev d;
    NdEngine* nd = nullptr;
    KftHost* kft = nullptr;
    Arena* arena = nullptr;
    int cur = 0;
    int n_spec = 0;                  // shadow sets carved for speculative trials (single-frame engines; 0: none)
    SpecSet spec[SPEC_MAX];
    int pred_iters = 0;              // inner iterations of the last fully solved LM trial (sizes later batches)
    int pack_rows = 0;               // rows whose incidence records this engine holds (sharded: the rank's keyframe range)
    size_t arena_bytes = 0;          // device bytes carved for it
    int pred_peek = 0;               // iterations the last trial needed to reach the first peek milestone
    bool first_trial_accepted = false;   // outcome of the first trial of the previous LM iteration
    double* h_scal = nullptr;        // pinned host mirrors
    int* h_flags = nullptr;
    HaloPlan halo;                   // sharded: boundary-keyframe rows exchanged with rank-1 / rank+1 (offsets in doubles)
    std::vector<int> vrow;           // vertex -> row
    // host copies needed to rewrite masks and to run the edge taps
    std::vector<int> sp_ij, dm_idx, un_ij;
    std::vector<float> sp_d0, dm_w, un_w;
    std::vector<int> sp_pos, dm_pos, un_pos;     // SELL positions of every incidence (2 / 4 / 1 per edge)
    std::vector<int> h_s_meta, h_d_meta;
    std::vector<uint32_t> h_s_om;
    std::vector<uint2> h_d_hdr;
    std::vector<uint32_t> h_d_om;
    std::vector<uint8_t> h_rflag, h_pose_fixed;
    std::vector<float> h_uv;         // observations by row, kept by a rank that holds its own rows only (residual taps)
    std::vector<int> sk_slot;        // embedded BA windows: observation (caller order) -> slot (pose-grouped, padded)
    std::vector<int> sk_vert;        // embedded mode: the skinned observations' node vertices (n_skin x 11, -1 pads) and weights
    std::vector<double> sk_om, sk_X0;
    unsigned long long serial = 0;   // identifies this engine to the context's tap buffer (nrs_ctx::tap)
    // device-packed engines (nrs_engine_devpack.hpp) keep no host copies of the edges: the taps read the raw device copies
    bool dev_edges = false;
    const int *raw_sp = nullptr, *raw_dm = nullptr;
    const float *raw_d0 = nullptr, *raw_w = nullptr;
};

// =====================================================================================
// device helpers
// =====================================================================================
__device__ inline int xcd_tile(int b, int nb) {
    const int nb8 = (nb + 7) >> 3;
    return (b & 7) * nb8 + (b >> 3);
}

template <int N>
__device__ inline void block_sum(double* v, double* lds /* 4*N */, int lane, int wave) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * N + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = lds[k] + lds[N + k] + lds[2 * N + k] + lds[3 * N + k];
    __syncthreads();
}

// same reduction, totals written to out[0..N) by the first N threads
template <int N>
__device__ inline void block_sum_store(const double* v, double* lds /* 4*N */, int tid, double* out) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) lds[wave * N + k] = s;
    }
    __syncthreads();
    if (tid < N) out[tid] = lds[tid] + lds[N + tid] + lds[2 * N + tid] + lds[3 * N + tid];
}

template <int T>
__device__ inline double sub_sum_t(double v) { return group_sum<T>(v); }   // reduce over the T lanes of a row

__device__ inline bool inv3_sym(const double* d /*xx xy xz yy yz zz*/, double lam, double* o) {
    const double a = d[0] + lam, b = d[1], c = d[2], e = d[3] + lam, f = d[4], g = d[5] + lam;
    const double c00 = e * g - f * f, c01 = c * f - b * g, c02 = b * f - c * e;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * g - c * c) * id; o[4] = (b * c - a * f) * id; o[5] = (a * e - b * b) * id;
    return det > 0;
}

// tile-level coarse correction: y = Bi rc with Bi = (B_t + lambda n_t I)^-1 prepared per trial by k_coarse_invert
__device__ inline void tile_level(const double* Bi /*6*/, const double* rc, double* y) {
    y[0] = Bi[0] * rc[0] + Bi[1] * rc[1] + Bi[2] * rc[2];
    y[1] = Bi[1] * rc[0] + Bi[3] * rc[1] + Bi[4] * rc[2];
    y[2] = Bi[2] * rc[0] + Bi[4] * rc[1] + Bi[5] * rc[2];
}

// stage 3-vectors of the tile's own rows and of its halo rows into LDS (optionally adding X0).
// The halo is a gather through an index list: all indices of a thread are requested first, then all
// rows, so that a thread has its 2-4 gathers in flight together instead of one dependent pair at a time.
// FIX (plain BA windows): the first HALO_FIX halo rows of every tile also sit at a fixed stride (halo_fix, -1 behind the
// list's end), so their index loads depend on the tile number only and go out TOGETHER with the halo_ptr loads instead of
// behind them -- one memory round trip less on the chain tile -> list bounds -> indices -> rows -> LDS that every wave
// of the tile waits out at the staging barrier (C4: the chain was 9.7 us of a 29.5 us wave of the lineariser).
constexpr int STAGE_K = 4;
constexpr int HALO_FIX = 2 * BLK;
template <bool FIX = false>
__device__ inline void stage_rows(const Dev& P, int b, int tid, const double* __restrict__ v, const double* __restrict__ add,
                                  double* lds) {
    const int row0 = b * P.tile_rows;
    int idx[STAGE_K];
    if (FIX) {
#pragma unroll
        for (int k = 0; k < HALO_FIX / BLK; ++k) idx[k] = P.halo_fix[(size_t)b * HALO_FIX + k * BLK + tid];
    }
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb;
#pragma unroll
    for (int k = FIX ? HALO_FIX / BLK : 0; k < STAGE_K; ++k) { const int i = tid + k * BLK; idx[k] = i < hn ? P.halo_rows[hb + i] : -1; }
    for (int i = tid; i < 3 * P.tile_rows; i += BLK) lds[i] = v[3 * (size_t)row0 + i] + (add ? add[3 * (size_t)row0 + i] : 0.0);
    double val[STAGE_K][3];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            const size_t r = (size_t)idx[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) val[k][c] = v[3 * r + c] + (add ? add[3 * r + c] : 0.0);
        }
    }
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            double* d = lds + 3 * (size_t)(P.tile_rows + tid + k * BLK);
            d[0] = val[k][0]; d[1] = val[k][1]; d[2] = val[k][2];
        }
    }
    for (int i = tid + STAGE_K * BLK; i < hn; i += BLK) {         // very large halos
        const size_t r = (size_t)P.halo_rows[hb + i];
        double* d = lds + 3 * (size_t)(P.tile_rows + i);
        d[0] = v[3 * r] + (add ? add[3 * r] : 0.0);
        d[1] = v[3 * r + 1] + (add ? add[3 * r + 1] : 0.0);
        d[2] = v[3 * r + 2] + (add ? add[3 * r + 2] : 0.0);
    }
}

// temporal-difference form: v and its forward / backward differences for the tile's rows and its (same-keyframe) halo
__device__ inline void stage_rows_d(const Dev& P, int b, int tid, const double* __restrict__ v, double* lv, double* lf, double* lb) {
    const int row0 = b * P.tile_rows;
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb;
    for (int i = tid; i < P.tile_rows + hn; i += BLK) {
        int r, rn, rp;
        if (i < P.tile_rows) { r = row0 + i; rn = P.nxt_row[r]; rp = P.prv_row[r]; }
        else { const int j = hb + i - P.tile_rows; r = P.halo_rows[j]; rn = P.halo_nxt[j]; rp = P.halo_prv[j]; }
        const size_t a = 3 * (size_t)r, an = 3 * (size_t)(rn >= 0 ? rn : r), ap = 3 * (size_t)(rp >= 0 ? rp : r);
        const double v0 = v[a], v1 = v[a + 1], v2 = v[a + 2];
        const double n0 = v[an], n1 = v[an + 1], n2 = v[an + 2];
        const double p0 = v[ap], p1 = v[ap + 1], p2 = v[ap + 2];
        lv[3 * i] = v0; lv[3 * i + 1] = v1; lv[3 * i + 2] = v2;
        lf[3 * i] = v0 - n0; lf[3 * i + 1] = v1 - n1; lf[3 * i + 2] = v2 - n2;
        lb[3 * i] = v0 - p0; lb[3 * i + 1] = v1 - p1; lb[3 * i + 2] = v2 - p2;
    }
}

// u and the (spring) positions of the linearisation point, one pass over the halo list
template <bool FIX = false, bool ALLX = false>                     // ALLX: positions of every halo row (the operator re-forms the damper factors, Dev::rc)
__device__ inline void stage_rows2(const Dev& P, int b, int tid, const double* __restrict__ u, const double* __restrict__ x,
                                   const double* __restrict__ add, double* lu, double* lx) {
    const int row0 = b * P.tile_rows;
    int idx[STAGE_K];
    if (FIX) {
#pragma unroll
        for (int k = 0; k < HALO_FIX / BLK; ++k) idx[k] = P.halo_fix[(size_t)b * HALO_FIX + k * BLK + tid];
    }
    const int hb = P.halo_ptr[b], hn = P.halo_ptr[b + 1] - hb, ns = ALLX ? hn : P.halo_ns[b];
#pragma unroll
    for (int k = FIX ? HALO_FIX / BLK : 0; k < STAGE_K; ++k) { const int i = tid + k * BLK; idx[k] = i < hn ? P.halo_rows[hb + i] : -1; }
    for (int i = tid; i < 3 * P.tile_rows; i += BLK) {
        lu[i] = u[3 * (size_t)row0 + i];
        lx[i] = x[3 * (size_t)row0 + i] + (add ? add[3 * (size_t)row0 + i] : 0.0);
    }
    double uu[STAGE_K][3], xx[STAGE_K][3];
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        if (idx[k] >= 0) {
            const size_t r = (size_t)idx[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) uu[k][c] = u[3 * r + c];
            if (tid + k * BLK < ns) {                              // positions: spring neighbours only
#pragma unroll
                for (int c = 0; c < 3; ++c) xx[k][c] = x[3 * r + c] + (add ? add[3 * r + c] : 0.0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < STAGE_K; ++k) {
        const int i = tid + k * BLK;
        if (idx[k] >= 0) {
            double* d = lu + 3 * (size_t)(P.tile_rows + i);
            d[0] = uu[k][0]; d[1] = uu[k][1]; d[2] = uu[k][2];
            if (i < ns) {
                double* e = lx + 3 * (size_t)(P.tile_rows + i);
                e[0] = xx[k][0]; e[1] = xx[k][1]; e[2] = xx[k][2];
            }
        }
    }
    for (int i = tid + STAGE_K * BLK; i < hn; i += BLK) {         // very large halos
        const size_t r = (size_t)P.halo_rows[hb + i];
        double* d = lu + 3 * (size_t)(P.tile_rows + i);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = u[3 * r + k];
        if (i < ns) {
            double* e = lx + 3 * (size_t)(P.tile_rows + i);
#pragma unroll
            for (int k = 0; k < 3; ++k) e[k] = x[3 * r + k] + (add ? add[3 * r + k] : 0.0);
        }
    }
}

// reprojection block of one row in factored form (same expressions as k_reproj):
//   a += J_l^T w (J_l u_l + J_p u_p) ; pose partial = J_p^T w J_l u_l ; cross = u_l . H_lp u_p
__device__ inline void row_factored(const RowRec& rc, const Pose& Tcw, const double* xs, const double* ul, const double* up,
                                    double pm, double& a0, double& a1, double& a2, double* part /*9*/) {
    double R[9];
    quat_to_R(Tcw.q, R);
    const double px = R[0] * xs[0] + R[1] * xs[1] + R[2] * xs[2] + Tcw.t[0];
    const double py = R[3] * xs[0] + R[4] * xs[1] + R[5] * xs[2] + Tcw.t[1];
    const double pz = R[6] * xs[0] + R[7] * xs[1] + R[8] * xs[2] + Tcw.t[2];
    double tl[2], tp[2], Jl[2][3], Jp[2][6];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const double j0 = -(double)rc.J[3 * rr], j1 = -(double)rc.J[3 * rr + 1], j2 = -(double)rc.J[3 * rr + 2];
        Jp[rr][0] = pm * (-j1 * pz + j2 * py);
        Jp[rr][1] = pm * (j0 * pz - j2 * px);
        Jp[rr][2] = pm * (-j0 * py + j1 * px);
        Jp[rr][3] = pm * j0; Jp[rr][4] = pm * j1; Jp[rr][5] = pm * j2;
        Jl[rr][0] = j0 * R[0] + j1 * R[3] + j2 * R[6];
        Jl[rr][1] = j0 * R[1] + j1 * R[4] + j2 * R[7];
        Jl[rr][2] = j0 * R[2] + j1 * R[5] + j2 * R[8];
        tl[rr] = Jl[rr][0] * ul[0] + Jl[rr][1] * ul[1] + Jl[rr][2] * ul[2];
        double s = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) s += Jp[rr][p] * up[p];
        tp[rr] = s;
    }
    const double w = rc.w;
    const double c0 = w * (tl[0] + tp[0]), c1 = w * (tl[1] + tp[1]);
    a0 += Jl[0][0] * c0 + Jl[1][0] * c1;
    a1 += Jl[0][1] * c0 + Jl[1][1] * c1;
    a2 += Jl[0][2] * c0 + Jl[1][2] * c1;
    part[2] = w * (tl[0] * tp[0] + tl[1] * tp[1]);
#pragma unroll
    for (int p = 0; p < 6; ++p) part[3 + p] = w * (Jp[0][p] * tl[0] + Jp[1][p] * tl[1]);
}

// A prefetched record word is consumed through one of these: the empty asm defines a new value at the point of
// consumption, so no pass can move the unpacking (mask, shift, conversion) back up behind the load that produced the word
// -- where it would make the wave wait for a chunk it has only just requested.
__device__ inline uint32_t consume(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ inline double consume(double v) { asm volatile("" : "+v"(v)); return v; }

// register views of one incidence (see the stream layout at the top)
__device__ inline SpringRec load_spring(const Dev& P, int j) {
    const uint32_t om = P.s_om[j];
    SpringRec r;
    r.qc = P.s_qc[j];
    r.other = (uint16_t)(om & 0xFFFFu); r.meta = (uint16_t)(om >> 16);
    return r;
}
__device__ inline DamperRec load_damper(const Dev& P, int j) {
    const uint2 h = P.d_hdr[j];
    DamperRec r;
    r.s = P.d_s[j];                                                // (0 for padding slots and for edges at level != 0)
    r.o0 = (uint16_t)(h.x & 0xFFFFu); r.o1 = (uint16_t)(h.x >> 16); r.o2 = (uint16_t)(h.y & 0xFFFFu); r.meta = (uint16_t)(h.y >> 16);
    return r;
}

__device__ inline double damper_sign(int role) { return (role == 0 || role == 3) ? -1.0 : 1.0; }

}  // namespace nrs
