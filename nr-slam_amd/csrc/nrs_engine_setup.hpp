// Host side of the engine: arena, row layout, sliced-ELL packing, halo lists, tile classes, uploads (engine_create and friends).
// Part of nrs_engine.hip (one translation unit); see that file's header for the design.
#pragma once

namespace nrs {

// =====================================================================================
// host side
// =====================================================================================
int engine_num_poses(const Engine* e) { return e->d.K; }
void engine_edge_counts(const Engine* e, int* n_sp, int* n_dm) { *n_sp = e->d.n_sp; *n_dm = e->d.n_dm; }

void arena_release(Arena* a) {
    if (a->base) (void)hipFree(a->base);
    a->base = nullptr;
    a->cap = a->off = 0;
}

struct ArenaPlan {                   // two passes: size, then carve
    Arena* a;
    bool dry;
    size_t off = 0;
    size_t row_lo = 0, row_n = 0;    // the rows this engine holds of every per-row array (a rank of a sharded window: its keyframes and one ghost keyframe either side)
    template <class Tp> Tp* get(size_t n) {
        const size_t bytes = ((n * sizeof(Tp) + 255) / 256) * 256 + 256;
        Tp* p = dry ? nullptr : reinterpret_cast<Tp*>(a->base + off);
        off += bytes;
        return p;
    }
    // a per-row array (per_row elements a row): storage for rows [row_lo, row_lo + row_n) only, addressed by the GLOBAL row index --
    // the pointer handed out is the storage's start minus row_lo rows, so every kernel and every exchange indexes as on one GPU and
    // nothing outside the held rows is ever touched (engine_create: the launches of a rank cover its own tiles, whose halos end one
    // keyframe away)
    template <class Tp> Tp* get_rows(size_t per_row) {
        Tp* p = get<Tp>(row_n * per_row);
        return dry ? nullptr : p - row_lo * per_row;
    }
};

// the shadow sets' arrays as the engine's own start out (zero partials, scalars and status words), their host mirrors, and the
// context's streams and events for them (created on first use)
static int spec_prepare(nrs_ctx* c, Engine* e) {
    const Dev& d = e->d;
    if (!c->pin_spec_scal) {
        NRS_HIP(c, hipHostMalloc((void**)&c->pin_spec_scal, sizeof(double) * SC_N * SPEC_MAX, hipHostMallocMapped | hipHostMallocCoherent));
        NRS_HIP(c, hipHostMalloc((void**)&c->pin_spec_flags, sizeof(int) * 8 * SPEC_MAX, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->pin_spec_flags, 0, sizeof(int) * 8 * SPEC_MAX);
        // (measured and dropped: shadow streams at the lowest priority so that they yield to the context's stream -- a level of the
        // factorisation on such a stream takes 57 us instead of 28 even with the device to itself)
        for (int j = 0; j < SPEC_MAX; ++j) {
            NRS_HIP(c, hipStreamCreateWithFlags(&c->spec_stream[j], hipStreamNonBlocking));
            NRS_HIP(c, hipEventCreateWithFlags(&c->spec_join[j], hipEventDisableTiming));
        }
        NRS_HIP(c, hipEventCreateWithFlags(&c->spec_fork, hipEventDisableTiming));
        for (int j = 0; j < 1 + SPEC_MAX; ++j) NRS_HIP(c, hipEventCreateWithFlags(&c->spec_back[j], hipEventDisableTiming));
    }
    for (int j = 0; j < e->n_spec; ++j) {
        SpecSet& q = e->spec[j];
        q.h_scal = c->pin_spec_scal + SC_N * j; q.h_flags = c->pin_spec_flags + 8 * j;
        NRS_HIP(c, hipMemsetAsync(q.part_apply, 0, sizeof(double) * (size_t)d.n_vecblk, c->stream));
        NRS_HIP(c, hipMemsetAsync(q.part_rchi, 0, sizeof(double) * (size_t)d.n_groups, c->stream));
        NRS_HIP(c, hipMemsetAsync(q.part_reg, 0, sizeof(double) * 2 * (size_t)d.n_regblk, c->stream));
        NRS_HIP(c, hipMemsetAsync(q.scal, 0, sizeof(double) * SC_N, c->stream));
        NRS_HIP(c, hipMemsetAsync(q.flags, 0, sizeof(int) * 8, c->stream));
        NRS_HIP(c, hipMemsetAsync(q.abort, 0, sizeof(int), c->stream));
    }
    return NRS_OK;
}

// shadow sets of a single-frame engine (speculative LM trials, nrs_engine_types.hpp): NRS_SPEC_TRIALS=<0..3> (0: one trial at a time)
static int spec_sets(const nrs_ctx* c) {
    if (c->opt.profile) return 0;                                  // (a profiling context times its launches one by one)
    if (const char* v = c->env("NRS_SPEC_TRIALS")) return std::max(0, std::min(SPEC_MAX, atoi(v)));
    // two shadow sets = three trials in flight: a fourth stream's launches are serialised behind another stream's on this runtime (its result
    // arrives a whole trial after the third's: 240 / 266 / 298 / 516 us at 1k points), so a third set only adds work that may be discarded
    return 2;
}

static void carve(ArenaPlan& A, Dev& d, bool has_X0, size_t nnz_s, size_t nnz_d, size_t n_slices, size_t n_halo, Engine* e) {
    const size_t nr = (size_t)d.n_rows, K = (size_t)d.K;
    if (d.row_hi <= 0) { d.row_lo = 0; d.row_hi = d.n_rows; }     // (callers that never shard leave the range unset: every row)
    A.row_lo = (size_t)d.row_lo; A.row_n = (size_t)(d.row_hi - d.row_lo);
    d.grp_pose = A.get<int>(d.n_groups);
    d.pose_grp_ptr = A.get<int>(K + 1);
    d.rflag = A.get_rows<uint8_t>(1);
    d.pose_fixed = A.get<uint8_t>(K);
    d.uv = A.get_rows<float>(2);
    double* X0 = has_X0 ? A.get_rows<double>(3) : A.get<double>(1);
    d.X0 = has_X0 ? X0 : nullptr;
    d.ss_ptr = A.get<int>(n_slices + 1);
    d.sd_ptr = A.get<int>(n_slices + 1);
    d.halo_ptr = A.get<int>((size_t)d.n_regblk + 1);
    d.halo_rows = A.get<int>(n_halo);
    d.halo_ns = A.get<int>((size_t)d.n_regblk);
    d.tile_list = A.get<int>((size_t)d.n_regblk);
    d.s_om = A.get<uint32_t>(d.use_lds ? nnz_s : 1);
    d.s_qc = A.get<double>(d.use_lds ? nnz_s : 1);
    d.d_hdr = A.get<uint2>(d.use_lds && !d.dform ? nnz_d : 1);
    d.d_om = A.get<uint32_t>(d.use_lds && d.dform ? nnz_d : 1);
    d.nxt_row = A.get<int>(d.dform ? nr : 1); d.prv_row = A.get<int>(d.dform ? nr : 1);
    d.halo_nxt = A.get<int>(d.dform ? std::max<size_t>(1, n_halo) : 1); d.halo_prv = A.get<int>(d.dform ? std::max<size_t>(1, n_halo) : 1);
    const size_t us = d.use_lds ? 1 : nnz_s, ud = d.use_lds ? 1 : nnz_d;     // unpacked arrays: fallback path only
    d.s_other = A.get<int>(us); d.s_d0 = A.get<float>(nnz_s); d.s_meta = A.get<int>(us);
    d.d_o0 = A.get<int>(ud); d.d_o1 = A.get<int>(ud); d.d_o2 = A.get<int>(ud);
    d.d_w = A.get<float>(nnz_d); d.d_meta = A.get<int>(ud);
    for (int s = 0; s < 2; ++s) { d.pose[s] = A.get<Pose>(K); d.xl[s] = A.get_rows<double>(3); }
    d.pose_init = A.get<Pose>(K);
    d.xl_init = A.get_rows<double>(3);
    d.D = A.get_rows<double>(6);
    d.Hpl = A.get<double>(d.use_lds ? 1 : 18 * nr);
    d.rowrec = d.use_lds ? A.get_rows<RowRec>(1) : A.get<RowRec>(1);
    d.row_tp = d.plain ? A.get_rows<uint32_t>(1) : A.get<uint32_t>(1);
    d.row_cnt = d.plain ? A.get_rows<uint32_t>(1) : A.get<uint32_t>(1);
    d.d_h4 = A.get<uint32_t>(d.plain && d.use_lds && !d.fused ? nnz_d : 1);
    d.s_g = A.get<double>(3 * us);
    d.d_s = A.get<double>(nnz_d);
    d.Hpp = A.get<double>(21 * K);
    d.bp = A.get<double>(6 * K);
    d.bl = A.get_rows<double>(3);
    d.Dinv = A.get_rows<double>(6);
    d.Hppinv = A.get<double>(36 * K);
    double** pv[] = {&d.xp, &d.rp, &d.up, &d.pp, &d.sp, &d.wp};
    for (auto p : pv) *p = A.get<double>(6 * K);
    double** rvv[] = {&d.xv, &d.rv, &d.uv3, &d.pv, &d.sv, &d.wv};
    for (auto p : rvv) *p = A.get_rows<double>(3);
    d.rp2 = A.get<double>(6 * K); d.sp2 = A.get<double>(6 * K); d.up2 = A.get<double>(6 * K);
    d.rv2 = A.get<double>(d.fused ? 3 * nr : 1); d.sv2 = A.get<double>(d.fused ? 3 * nr : 1); d.wv2 = A.get<double>(d.fused ? 3 * nr : 1);
    d.part_spmv2 = A.get<double>(d.fused ? NPART * (size_t)d.n_regblk : 1);
    {
        const size_t nb = d.coarse ? (size_t)d.n_regblk : 1, nc = d.coarse ? (size_t)d.co_n : 1;
        d.co_ct = A.get<double>(nb * (d.coarse ? (size_t)d.n_groups : 1) * 6);
        d.co_cp = A.get<double>(nb * 18);
        d.co_tb = A.get<double>(nb * 4);
        d.co_bt = A.get<double>(nb * 6);
        d.co_bti = A.get<double>(nb * 6);
        d.part_ts = A.get<double>(nb * 9); d.part_ts2 = A.get<double>(nb * 9);
        d.co_c0 = A.get<double>(nc * nc); d.co_nn = A.get<double>(nc); d.co_bc = A.get<double>(nc);
        d.co_inv = A.get<double>(nc * nc); d.co_y0 = A.get<double>(nc);
    }
    d.tile_desc = A.get<int>(d.fused ? 8 * (size_t)d.n_regblk : 4);
    d.halo_fix = A.get<int>(d.fused ? BLK * (size_t)d.n_regblk : d.plain && d.use_lds ? HALO_FIX * (size_t)d.n_regblk : 4);
    d.red = A.get<double>(4 + 6 * K);
    d.red_loc = A.get<double>(4 + 6 * K);
    d.pk = A.get<double>(2 + 8 + 27 * K);
    d.pk_loc = A.get<double>(2 + 8 + 27 * K);
    d.part_ru = A.get<double>(d.ecd ? 2 * (size_t)d.n_vecblk : 1);
    d.part_lin = A.get<double>(32 * (size_t)d.n_groups * (size_t)d.lin_rb);
    d.part_rchi = A.get<double>((size_t)d.n_groups);
    d.part_pchi = A.get<double>(K);
    d.part_reg = A.get<double>(2 * (size_t)d.n_regblk);
    d.part_spmv = A.get<double>(NPART * (size_t)d.n_regblk);
    d.part_apply = A.get<double>((size_t)d.n_vecblk);
    d.scal = A.get<double>(SC_N);
    d.flags = A.get<int>(8);
    d.ec_sp = A.get<EcSpring>(d.ec_on ? std::max(1, d.ec_nsp) : 1);
    d.ec_dm = A.get<EcDamper>(d.ec_on ? std::max(1, d.ec_ndm) : 1);
    d.ec_w = A.get<float>(d.ec_on ? std::max(1, d.ec_ndm) : 1);
    d.part_ec = A.get<double>(d.ec_on ? std::max(1, d.ec_nblk) : 1);
    for (int j = 0; j < e->n_spec; ++j) {                          // shadow sets of what an LM trial writes (speculative trials: nrs_engine_types.hpp)
        SpecSet& q = e->spec[j];
        q.xv = A.get_rows<double>(3); q.xp = A.get<double>(6 * K);
        q.pose = A.get<Pose>(K); q.xl = A.get_rows<double>(3);
        q.part_apply = A.get<double>((size_t)d.n_vecblk); q.part_rchi = A.get<double>((size_t)d.n_groups); q.part_reg = A.get<double>(2 * (size_t)d.n_regblk);
        q.scal = A.get<double>(SC_N); q.flags = A.get<int>(8); q.abort = A.get<int>(1);
        q.sk_part = q.sk_chi = nullptr;
    }
}

template <class Tp>
static int h2d(nrs_ctx* c, Tp* dst, const std::vector<Tp>& src) {
    if (!src.empty()) NRS_HIP(c, hipMemcpyAsync(dst, src.data(), sizeof(Tp) * src.size(), hipMemcpyHostToDevice, c->stream));
    return NRS_OK;
}

// a full-length host image of a per-row array (per_row elements a row) into the rows this engine holds of it
template <class Tp>
static int h2d_rows(nrs_ctx* c, const Dev& d, Tp* dst, const std::vector<Tp>& src, size_t per_row) {
    const size_t o = (size_t)d.row_lo * per_row, n = (size_t)(d.row_hi - d.row_lo) * per_row;
    if (n) NRS_HIP(c, hipMemcpyAsync(dst + o, src.data() + o, sizeof(Tp) * n, hipMemcpyHostToDevice, c->stream));
    return NRS_OK;
}

// Host-side set-up work split over a few threads.  Every use below is order-free (disjoint outputs, or integer counts) or
// reproduces the sequential order (a thread owns a range of ROWS and scans the edges in edge order): the packed problem is
// the same bits for any thread count (tests/test_gpu_scale.py).
static int host_threads(const nrs_ctx* c, size_t work) {
    if (work < 600000) {
        // single-frame problems (a 4.5k-point frame: 0.3 M incidences): sixteen threads per stage cost more than they save (8 ms per frame,
        // round 2); NRS_HOST_THREADS_SMALL=<n> tries a few (round 5: see profiles/README.md)
        if (const char* ev = c->env("NRS_HOST_THREADS_SMALL")) if (work >= 100000) return std::max(1, std::min(8, atoi(ev)));
        return 1;
    }
    int n = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* ev = c->env("NRS_HOST_THREADS")) n = std::max(1, std::min(64, atoi(ev)));
    return n;
}
// plain two-kernel windows whose rows all know their temporal partners: the dampers' 8-byte headers {o0, o1, o2, meta} shrink
// to 4 bytes {o0 : 12, o2 : 12, meta : 8} -- o1 is the row's own partner (row_tp) and tile-local ids stay below 4096.  Derived
// on the device from d_hdr, whichever packer built that.
__global__ void k_compact_headers(size_t n, const uint2* __restrict__ hdr, uint32_t* __restrict__ h4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 h = hdr[i];
    const uint32_t m16 = h.y >> 16;
    h4[i] = m16 == REC_NONE ? 0xFFFFFFFFu : ((h.x & 0xFFFu) | ((h.y & 0xFFFu) << 12) | ((m16 & 0xFFu) << 24));
}
static void engine_compact_headers(nrs_ctx* c, Engine* e) {
    Dev& d = e->d;
    d.h4 = 0; d.rc = 0; d.nt = 0;
    if (!(d.plain && d.tp_ok && d.use_lds && !d.fused && d.T == 2) || c->env("NRS_NO_H4")) return;
    if (d.tile_rows + std::max(d.cap_h[0], d.cap_h[1]) + 2 >= 4096 || d.sd_nnz <= 0) return;
    hipLaunchKernelGGL(k_compact_headers, dim3((unsigned)(((size_t)d.sd_nnz + 255) / 256)), dim3(256), 0, c->stream, (size_t)d.sd_nnz, d.d_hdr, d.d_h4);
    d.h4 = 1;
    if (const char* v = c->env("NRS_RC")) d.rc = atoi(v) & 3;
    d.nt = 12 * ((size_t)d.ss_nnz + (size_t)d.sd_nnz) > ((size_t)128 << 20);   // the streams (with the vectors next to them) do not stay in the 256 MB Infinity Cache between launches
    if (const char* v = c->env("NRS_NT")) d.nt = atoi(v) != 0;
}

template <class F>
static void parallel_for(int nt, F&& fn) {                         // fn(thread index, thread count)
    if (nt <= 1) { fn(0, 1); return; }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) {
        try { th.emplace_back([&fn, t, nt] { fn(t, nt); }); }
        catch (const std::system_error&) { fn(t, nt); }            // no thread to be had: this share runs here
    }
    fn(0, nt);
    for (auto& x : th) x.join();
}
static inline void chunk(int64_t n, int t, int nt, int64_t& a, int64_t& b) { a = n * t / nt; b = n * (t + 1) / nt; }
// count into a shared array: a plain increment when the stage runs on one thread (a locked add costs ~2 ns even uncontended:
// 2.5 ms per tracked frame over its two engines)
static inline void count_up(int* p, int nt) { if (nt == 1) ++*p; else __atomic_fetch_add(p, 1, __ATOMIC_RELAXED); }

// per-edge masks -> per-incidence meta words and per-row flags (host), then upload
static int push_masks(nrs_ctx* c, Engine* e, const uint8_t* sp_active, const uint8_t* dm_active) {
    Dev& d = e->d;
    auto vfixed = [&](int v) { return (e->h_rflag[e->vrow[v]] & RF_FIXED) != 0; };
    const int nt = host_threads(c, (size_t)d.n_sp + (size_t)d.n_dm);
    parallel_for(nt, [&](int ti, int n_thr) {                     // every edge writes its own incidence slots
    int64_t q0, q1;
    chunk(d.n_sp, ti, n_thr, q0, q1);
    for (int s = (int)q0; s < (int)q1; ++s) {
        const int i = e->sp_ij[2 * s], j = e->sp_ij[2 * s + 1];
        const bool act = (!sp_active || sp_active[s]) && !(vfixed(i) && vfixed(j));
        const int m = act ? SM_ACTIVE : 0;
        if (e->sp_pos[2 * s] >= 0) e->h_s_meta[e->sp_pos[2 * s]] = m | (act ? SM_COUNT : 0);      // (-1: the row belongs to another rank)
        if (e->sp_pos[2 * s + 1] >= 0) e->h_s_meta[e->sp_pos[2 * s + 1]] = m;
    }
    chunk(d.n_dm, ti, n_thr, q0, q1);
    for (int s = (int)q0; s < (int)q1; ++s) {
        bool allfix = true;
        int first = -1;
        for (int r = 0; r < 4; ++r) {
            const int v = e->dm_idx[4 * s + r];
            if (v >= 0) { if (first < 0) first = r; allfix = allfix && vfixed(v); }
        }
        const bool act = (!dm_active || dm_active[s]) && !allfix;
        for (int r = 0; r < 4; ++r) {
            const int p = e->dm_pos[4 * s + r];
            if (p < 0) continue;
            e->h_d_meta[p] = r | (act ? DM_ACTIVE : 0) | ((act && r == first) ? DM_COUNT : 0);
        }
    }
    chunk(d.n_un, ti, n_thr, q0, q1);
    for (int s = (int)q0; s < (int)q1; ++s) {
        const bool act = !vfixed(e->un_ij[2 * s]);
        if (e->un_pos[s] >= 0) e->h_d_meta[e->un_pos[s]] = 2 | DM_UNARY | (act ? (DM_ACTIVE | DM_COUNT) : 0);
    }
    });
    if (d.use_lds) {
        // the meta half-words of the static header streams (the factor streams are not touched)
        parallel_for(nt, [&](int ti, int n_thr) {
            int64_t a, b;
            chunk((int64_t)e->h_s_om.size(), ti, n_thr, a, b);
            for (int64_t i = a; i < b; ++i) {
                const int m = e->h_s_meta[i];
                const uint32_t m16 = (uint32_t)(((m & SM_ACTIVE) ? SR_ACTIVE : 0) | ((m & SM_COUNT) ? SR_COUNT : 0));
                e->h_s_om[i] = (e->h_s_om[i] & 0xFFFFu) | (m16 << 16);
            }
            chunk((int64_t)e->h_d_hdr.size(), ti, n_thr, a, b);
            for (int64_t i = a; i < b; ++i) {
                const uint32_t m16 = e->h_d_meta[i] < 0 ? (uint32_t)REC_NONE : (uint32_t)(e->h_d_meta[i] & 0xFFFF);
                e->h_d_hdr[i].y = (e->h_d_hdr[i].y & 0xFFFFu) | (m16 << 16);
            }
        });
        for (size_t i = 0; i < e->h_d_om.size(); ++i) {
            const uint32_t m16 = e->h_d_meta[i] < 0 ? (uint32_t)REC_NONE : (uint32_t)(e->h_d_meta[i] & 0xFFFF);
            e->h_d_om[i] = (e->h_d_om[i] & 0xFFFFu) | (m16 << 16);
        }
        NRS_TRY(h2d(c, d.s_om, e->h_s_om));
        if (d.dform) NRS_TRY(h2d(c, d.d_om, e->h_d_om));
        else NRS_TRY(h2d(c, d.d_hdr, e->h_d_hdr));
    } else {
        NRS_TRY(h2d(c, d.s_meta, e->h_s_meta));
        NRS_TRY(h2d(c, d.d_meta, e->h_d_meta));
    }
    NRS_TRY(h2d_rows(c, d, d.rflag, e->h_rflag, 1));
    NRS_TRY(h2d(c, d.pose_fixed, e->h_pose_fixed));
    return NRS_OK;
}

// Contiguous keyframe ranges for `world` ranks, balanced by padded rows, every rank at least one
// keyframe: kb[r] .. kb[r+1] are rank r's keyframes.  grp_ptr[k] = first ROW_ALIGN group of keyframe k.
void shard_plan(int K, const int* grp_ptr, int world, int* kb) {
    const int total = grp_ptr[K];
    kb[0] = 0;
    for (int r = 1; r < world; ++r) {
        const int64_t want = (int64_t)total * r / world;
        int k = kb[r - 1] + 1;                                     // at least one keyframe for rank r-1 ...
        while (k < K - (world - r) && grp_ptr[k] < want) ++k;      // ... and for every rank that follows
        // the boundary closest to the ideal split
        if (k - 1 > kb[r - 1] && want - grp_ptr[k - 1] < grp_ptr[k] - want) --k;
        kb[r] = k;
    }
    kb[world] = K;
}

static bool devpack_eligible(nrs_ctx* c, const EngineSpec& s, int n_pad_rows);
static int engine_create_device(nrs_ctx* c, const EngineSpec& s, Arena* arena, Engine* e, bool* done);

int engine_create(nrs_ctx* c, const EngineSpec& s, Arena* arena, Engine** out) {
    *out = nullptr;
    // failures every rank of a sharded upload sees alike (argument validation on identical inputs) are reported
    // without a collective; everything else is rank-local and is agreed on by the caller (nrs_dba_upload)
    c->err_local = false;
    if (s.K <= 0 || s.M <= 0 || !s.poses || !s.x || !s.lm_pose || !s.uv || !s.rflag || s.n_sp < 0 || s.n_dm < 0 || s.n_un < 0)
        return c->fail(NRS_ERR_INVALID, "engine: bad specification");
    for (int i = 0; i < s.M; ++i)
        if (s.lm_pose[i] < 0 || s.lm_pose[i] >= s.K || (i > 0 && s.lm_pose[i] < s.lm_pose[i - 1]))
            return c->fail(NRS_ERR_INVALID, "vertex pose index must be non-decreasing and in [0, n_poses)");
    if (!s.edges_on_device) {                                      // (device-built edge lists are valid by construction)
        for (int64_t i = 0; i < 2 * (int64_t)s.n_sp; ++i)
            if (s.sp_ij[i] < 0 || s.sp_ij[i] >= s.M) return c->fail(NRS_ERR_INVALID, "spring index out of range");
        for (int64_t i = 0; i < 4 * (int64_t)s.n_dm; ++i)
            if (s.dm_idx[i] < -1 || s.dm_idx[i] >= s.M) return c->fail(NRS_ERR_INVALID, "damper index out of range");
    }
    for (int64_t i = 0; i < 2 * (int64_t)s.n_un; ++i)
        if (s.un_ij[i] < 0 || s.un_ij[i] >= s.M) return c->fail(NRS_ERR_INVALID, "unary damper index out of range");
    if (c->comm && s.shard) {                                      // the same on every rank: no collective follows these returns
        if (c->comm->world > 8) return c->fail(NRS_ERR_INVALID, "sharded solve: at most 8 ranks");
        if (s.K < c->comm->world) return c->fail(NRS_ERR_INVALID, "sharded solve: %d keyframes cannot be split over %d ranks", s.K, c->comm->world);
    }
    c->err_local = true;                                           // from here on a failure may be this rank's alone: the caller lets the ranks agree
    const bool tm = c->env("NRS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!tm) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nrs] engine_create %-18s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    NRS_HIP(c, hipSetDevice(c->device));
    Engine* e = new (std::nothrow) Engine();
    if (!e) return c->fail(NRS_ERR_ALLOC, "out of host memory");
    struct Guard { nrs_ctx* c; Engine* e; bool keep = false; ~Guard() { if (!keep) engine_destroy(c, e); } } guard{c, e};
    e->arena = arena;
    Dev& d = e->d;
    memset(&d, 0, sizeof(d));
    // lanes per row: 2 measured best on C2 (92k rows), 8 on single-frame problems (4.5k rows), where
    // the kernels are bound by per-lane latency chains rather than by traffic (profiles/README.md)
    int n_pad_rows = 0;
    {
        std::vector<int> cnt(s.K, 0);
        for (int i = 0; i < s.M; ++i) cnt[s.lm_pose[i]]++;
        for (int k = 0; k < s.K; ++k) n_pad_rows += std::max(1, (cnt[k] + ROW_ALIGN - 1) / ROW_ALIGN) * ROW_ALIGN;
    }
    if (devpack_eligible(c, s, n_pad_rows)) {                      // plain BA window on the two-kernel path: built on the device
        bool done = false;
        NRS_TRY(engine_create_device(c, s, arena, e, &done));
        if (done) { guard.keep = true; *out = e; return NRS_OK; }
        *e = Engine();                                             // (did not qualify after all: the host path, from scratch)
        e->arena = arena;
        memset(&d, 0, sizeof(d));
    }
    if (s.edges_on_device) return c->fail(NRS_ERR_STATE, "device-built edge lists need the device-side construction, which this window does not qualify for");
    // a2's single-frame engines: the direct solver's symbolic phase needs the structure only and runs next to the packing below
    if (s.n_skin > 0) {                                            // (checked HERE: the plan thread below indexes by these)
        if ((!(arena == &c->arena_trk && s.K == 1) && !s.sk_pose) || !s.sk_uv || !s.sk_X0 || !s.sk_node || !s.sk_om)
            return c->fail(NRS_ERR_INVALID, "skinned observations: single-frame tracking engines, or BA windows with a pose per observation");
        for (size_t q = 0; q < (size_t)SK_MAX * s.n_skin; ++q)
            if (s.sk_node[q] >= s.M || s.sk_node[q] < -1) return c->fail(NRS_ERR_INVALID, "skinned observation: node index out of range");
    }
    NdPrep nd_prep;                                                // (declared after `guard`: joined before the engine can go away)
    NdIn nd_in;
    if (arena == &c->arena_trk && s.K == 1) {
        e->nd = new (std::nothrow) NdEngine();
        if (!e->nd) return c->fail(NRS_ERR_ALLOC, "out of host memory");
        e->n_spec = spec_sets(c);                                  // shadow sets for speculative LM trials (carved with the arena below)
        e->nd->pos.resize(3 * (size_t)s.M);
        for (size_t i = 0; i < 3 * (size_t)s.M; ++i) e->nd->pos[i] = s.x[i] + (s.X0 ? s.X0[i] : 0.0);
        nd_in.M = s.M; nd_in.rflag = s.rflag; nd_in.pose_fixed = s.pose_fixed && s.pose_fixed[0];
        nd_in.n_sp = s.n_sp; nd_in.sp_ij = s.sp_ij; nd_in.n_dm = s.n_dm; nd_in.dm_idx = s.dm_idx;
        nd_in.n_skin = s.n_skin; nd_in.sk_vert = s.sk_node; nd_in.sk_om = s.sk_om;
        nd_in.vpos = e->nd->pos.data();
        bool inline_run = c->env("NRS_HOST_THREADS") && atoi(c->env("NRS_HOST_THREADS")) <= 1;
        if (!inline_run) {
            PlanWorker* pw = static_cast<PlanWorker*>(c->plan_worker);
            if (!pw) {
                pw = new (std::nothrow) PlanWorker();
                if (pw && !pw->start()) { delete pw; pw = nullptr; }
                c->plan_worker = pw;
            }
            if (pw) { nd_prep.worker = pw; pw->submit([c, &nd_in, &nd_prep] { nd_prep_run(c, nd_in, nd_prep); }); }
            else inline_run = true;
        }
        if (inline_run) nd_prep_run(c, nd_in, nd_prep);
    }
    int T = n_pad_rows >= 32768 ? 2 : 8;
    if (const char* ev = c->env("NRS_SELL_T")) {
        const int v = atoi(ev);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) T = v;
    }
    d.T = T;
    d.K = s.K; d.M = s.M; d.n_sp = s.n_sp; d.n_dm = s.n_dm; d.n_un = s.n_un;
    d.cam = s.cam;
    d.info_reproj = s.info_reproj; d.delta_reproj = s.delta_reproj;
    d.info_pos = s.info_pos; d.delta_pos = s.delta_pos;
    d.info_spatial = s.info_spatial; d.delta_spatial = s.delta_spatial;
    d.k_spring = s.k_spring; d.spring_form = s.spring_form;

    const int nt_all = host_threads(c, 2 * (size_t)s.n_sp + 4 * (size_t)s.n_dm + (size_t)s.n_un);   // one decision for every set-up stage
    // ---- row layout: pose-major, each pose padded to ROW_ALIGN rows, Morton order inside
    std::vector<int> pose_ptr(s.K + 1, 0);
    for (int i = 0; i < s.M; ++i) pose_ptr[s.lm_pose[i] + 1]++;
    for (int k = 0; k < s.K; ++k) pose_ptr[k + 1] += pose_ptr[k];
    std::vector<int> pose_grp_ptr(s.K + 1, 0), grp_pose;
    for (int k = 0; k < s.K; ++k) {
        const int n = pose_ptr[k + 1] - pose_ptr[k];
        const int ng = std::max(1, (n + ROW_ALIGN - 1) / ROW_ALIGN);
        pose_grp_ptr[k + 1] = pose_grp_ptr[k] + ng;
        for (int g = 0; g < ng; ++g) grp_pose.push_back(k);
    }
    d.n_groups = pose_grp_ptr[s.K];
    d.n_rows = d.n_groups * ROW_ALIGN;
    d.n_regblk = d.n_rows / (BLK / T);
    d.n_vecblk = d.n_rows / BLK;
    e->vrow.resize(s.M);
    {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        auto pos = [&](int v, int a) { return s.x[3 * (size_t)v + a] + (s.X0 ? s.X0[3 * (size_t)v + a] : 0.0); };
        for (int v = 0; v < s.M; ++v)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], pos(v, a)); hi[a] = std::max(hi[a], pos(v, a)); }
        auto spread = [](uint64_t v) {            // 21 bits -> every third bit
            v &= 0x1fffff;
            v = (v | v << 32) & 0x1f00000000ffffULL;
            v = (v | v << 16) & 0x1f0000ff0000ffULL;
            v = (v | v << 8) & 0x100f00f00f00f00fULL;
            v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
            v = (v | v << 2) & 0x1249249249249249ULL;
            return v;
        };
        const bool morton = c->env("NRS_NO_MORTON") == nullptr;
        const int nt_rows = nt_all;
        parallel_for(std::min(nt_rows, s.K), [&](int ti, int nt) {
        std::vector<std::pair<uint64_t, int>> keys;
        int64_t k0, k1;
        chunk(s.K, ti, nt, k0, k1);
        for (int k = (int)k0; k < (int)k1; ++k) {
            keys.clear();
            for (int v = pose_ptr[k]; v < pose_ptr[k + 1]; ++v) {
                uint64_t code = 0;
                if (morton)
                    for (int a = 0; a < 3; ++a) {
                        const double ext = hi[a] - lo[a];
                        const double f = ext > 0 ? (pos(v, a) - lo[a]) / ext : 0.0;
                        code |= spread((uint64_t)(f * 2097151.0)) << a;
                    }
                keys.emplace_back(code, v);
            }
            std::stable_sort(keys.begin(), keys.end());
            for (size_t i = 0; i < keys.size(); ++i) e->vrow[keys[i].second] = pose_grp_ptr[k] * ROW_ALIGN + (int)i;
        }
        });
    }
    // Inside a tile the row order is free (neighbour ids are tile-local, the vector kernels stream): rows are sorted
    // by their incidence counts, so that the rows of a slice (64 / T consecutive rows share the slice's width) have
    // similar counts.  C2: sliced-ELL padding 1.34x / 1.37x (springs / dampers) -> 1.22x / 1.15x of the incidences.
    if (!c->env("NRS_NO_TILE_SORT")) {
        std::vector<int> cs(s.M, 0), cd(s.M, 0);
        const int nt = nt_all;
        parallel_for(nt, [&](int ti, int n) {                      // integer counts: order-free
            int64_t a, b;
            chunk(2 * (int64_t)s.n_sp, ti, n, a, b);
            for (int64_t q = a; q < b; ++q) count_up(&cs[s.sp_ij[q]], n);
            chunk(4 * (int64_t)s.n_dm, ti, n, a, b);
            for (int64_t q = a; q < b; ++q)
                if (s.dm_idx[q] >= 0) count_up(&cd[s.dm_idx[q]], n);
            chunk(s.n_un, ti, n, a, b);
            for (int64_t q = a; q < b; ++q) count_up(&cd[s.un_ij[2 * q]], n);
        });
        const int tile = BLK / T;
        std::vector<int> row_v((size_t)d.n_rows, -1);
        for (int v = 0; v < s.M; ++v) row_v[e->vrow[v]] = v;
        parallel_for(nt, [&](int ti, int n) {                      // tiles are independent
            std::vector<int> seg;
            int64_t t0, t1;
            chunk(d.n_rows / tile, ti, n, t0, t1);
            for (int64_t tl = t0; tl < t1; ++tl) {
                const int r0 = (int)tl * tile;
                seg.clear();
                for (int r = r0; r < r0 + tile; ++r) if (row_v[r] >= 0) seg.push_back(row_v[r]);
                std::stable_sort(seg.begin(), seg.end(), [&](int a, int b2) { return cd[a] != cd[b2] ? cd[a] > cd[b2] : cs[a] > cs[b2]; });
                for (size_t i = 0; i < seg.size(); ++i) e->vrow[seg[i]] = r0 + (int)i;
            }
        });
    }
    mark("row layout");
    // ---- incidence lists -> sliced ELL, built with two counting passes (no per-row containers)
    const int Rw = 64 / T;
    const int n_slices = d.n_rows / Rw;
    const int dm_slots = 4 * s.n_dm;
    e->sp_pos.assign(2 * (size_t)s.n_sp, -1);
    e->dm_pos.assign(4 * (size_t)s.n_dm, -1);
    e->un_pos.assign((size_t)s.n_un, -1);
    // Sharded: a rank packs (and later stores) the incidence records of ITS rows only -- the rows of its keyframe range.
    // Rows of other ranks keep empty lists: their tiles never run here, and what this rank's tiles read of them are
    // vector rows (the boundary-keyframe exchange), not records.  Packing time and record memory scale with 1 / ranks.
    int pack_lo = 0, pack_hi = d.n_rows;
    if (c->comm && s.shard && c->comm->world <= 8 && s.K >= c->comm->world && !c->env("NRS_SHARD_PACK_ALL")) {
        std::vector<int> kb0(c->comm->world + 1);
        shard_plan(s.K, pose_grp_ptr.data(), c->comm->world, kb0.data());
        pack_lo = pose_grp_ptr[kb0[c->comm->rank]] * ROW_ALIGN;
        pack_hi = pose_grp_ptr[kb0[c->comm->rank + 1]] * ROW_ALIGN;
    }
    auto mine = [&](int row) { return row >= pack_lo && row < pack_hi; };
    e->pack_rows = pack_hi - pack_lo;
    // the row of every incidence, once (the passes below scan these flat arrays instead of chasing vrow)
    const int nt_pack = nt_all;
    std::vector<int> sp_row(2 * (size_t)s.n_sp), dm_row(4 * (size_t)s.n_dm), un_row((size_t)s.n_un);
    std::vector<int> cnt_s(d.n_rows, 0), cnt_d(d.n_rows, 0);
    parallel_for(nt_pack, [&](int ti, int n) {
        int64_t a, b;
        chunk(2 * (int64_t)s.n_sp, ti, n, a, b);
        for (int64_t q = a; q < b; ++q) {
            const int r = e->vrow[s.sp_ij[q]];
            sp_row[q] = r;
            if (mine(r)) count_up(&cnt_s[r], n);
        }
        chunk(4 * (int64_t)s.n_dm, ti, n, a, b);
        for (int64_t q = a; q < b; ++q) {
            const int r = s.dm_idx[q] >= 0 ? e->vrow[s.dm_idx[q]] : -1;
            dm_row[q] = r;
            if (r >= 0 && mine(r)) count_up(&cnt_d[r], n);
        }
        chunk(s.n_un, ti, n, a, b);
        for (int64_t q = a; q < b; ++q) {
            const int r = e->vrow[s.un_ij[2 * q]];
            un_row[q] = r;
            if (mine(r)) count_up(&cnt_d[r], n);
        }
    });
    mark("incidence rows");
    std::vector<int> ss_ptr(n_slices + 1, 0), sd_ptr(n_slices + 1, 0);
    for (int sl = 0; sl < n_slices; ++sl) {
        int ws = 0, wd = 0;
        for (int r = 0; r < Rw; ++r) {
            ws = std::max(ws, (cnt_s[sl * Rw + r] + T - 1) / T);
            wd = std::max(wd, (cnt_d[sl * Rw + r] + T - 1) / T);
        }
        ss_ptr[sl + 1] = ss_ptr[sl] + ws * 64;
        sd_ptr[sl + 1] = sd_ptr[sl] + wd * 64;
    }
    const size_t nnz_s = (size_t)ss_ptr[n_slices], nnz_d = (size_t)sd_ptr[n_slices];
    d.ss_nnz = (int)nnz_s;
    d.sd_nnz = (int)nnz_d;
    // packed position of the k-th incidence of a row
    auto pos_of = [&](const std::vector<int>& ptr, int row, int k) {
        const int sl = row / Rw, r = row - sl * Rw;
        return (size_t)ptr[sl] + (size_t)(k / T) * 64 + (size_t)r * T + (size_t)(k % T);
    };
    std::vector<int> S_other(nnz_s, -1), D_o(3 * nnz_d, -1), D_role(nnz_d, -1);
    std::vector<float> S_d0(nnz_s, 0.f), D_w(nnz_d, 0.f);
    std::fill(cnt_s.begin(), cnt_s.end(), 0);
    std::fill(cnt_d.begin(), cnt_d.end(), 0);
    mark("sell arrays");
    // fill: a thread owns a contiguous range of rows (balanced by slots) and scans ALL incidences in edge order, taking the
    // ones of its rows -- the k-th incidence of a row is the k-th in edge order, as in a sequential pass
    std::vector<int> row_cut(nt_pack + 1, pack_hi);
    row_cut[0] = pack_lo;
    {
        const int sl_lo = pack_lo / Rw, sl_hi = pack_hi / Rw;
        const int64_t tot = ((int64_t)ss_ptr[sl_hi] - ss_ptr[sl_lo]) + ((int64_t)sd_ptr[sl_hi] - sd_ptr[sl_lo]);
        int sl = sl_lo;
        for (int t = 1; t < nt_pack; ++t) {
            const int64_t want = tot * t / nt_pack;
            while (sl < sl_hi && ((int64_t)ss_ptr[sl] - ss_ptr[sl_lo]) + ((int64_t)sd_ptr[sl] - sd_ptr[sl_lo]) < want) ++sl;
            row_cut[t] = sl * Rw;
        }
    }
    parallel_for(nt_pack, [&](int ti, int) {
        const int lo = row_cut[ti], hi = row_cut[ti + 1];
        if (lo >= hi) return;
        auto own = [&](int r) { return r >= lo && r < hi; };
        for (int q = 0; q < s.n_sp; ++q) {
            const int a = sp_row[2 * (size_t)q], b = sp_row[2 * (size_t)q + 1];
            if (own(a)) {
                const size_t pa = pos_of(ss_ptr, a, cnt_s[a]++);
                S_other[pa] = b; S_d0[pa] = s.sp_d0[q]; e->sp_pos[2 * (size_t)q] = (int)pa;
            }
            if (own(b)) {
                const size_t pb = pos_of(ss_ptr, b, cnt_s[b]++);
                S_other[pb] = a; S_d0[pb] = s.sp_d0[q]; e->sp_pos[2 * (size_t)q + 1] = (int)pb;
            }
        }
        for (int q = 0; q < s.n_dm; ++q) {
            const int* r4 = &dm_row[4 * (size_t)q];
            if (!(own(r4[0]) || own(r4[1]) || own(r4[2]) || own(r4[3]))) continue;
            for (int role = 0; role < 4; ++role) {
                if (r4[role] < 0 || !own(r4[role])) continue;
                const size_t pz = pos_of(sd_ptr, r4[role], cnt_d[r4[role]]++);
                int z = 0;
                for (int k = 0; k < 4; ++k)
                    if (k != role) D_o[3 * pz + z++] = r4[k];
                D_w[pz] = s.dm_w[q];
                D_role[pz] = role;
                e->dm_pos[4 * (size_t)q + role] = (int)pz;
            }
        }
        for (int q = 0; q < s.n_un; ++q) {       // own role 2 (1n, +), value-only other in role 3 (2n, -)
            const int row = un_row[q];
            if (!own(row)) continue;
            const size_t pz = pos_of(sd_ptr, row, cnt_d[row]++);
            D_o[3 * pz + 2] = e->vrow[s.un_ij[2 * q + 1]];
            D_w[pz] = s.un_w[q];
            D_role[pz] = 2;
            e->un_pos[q] = (int)pz;
        }
    });
    (void)dm_slots;
    mark("sell pack");
    mark("sell vectors");
    // ---- temporal-difference form of the dampers (nrs_engine_types.hpp), OPT-IN (NRS_DFORM=1).  Measured: C2 (cache
    // resident) operator 24.4 -> 24.9 us, lineariser 42 -> 43 us: neutral; C4 (HBM regime) operator 1.77 -> 4.15 ms,
    // lineariser 3.08 -> 5.69 ms: the three gathers per staged row (v, v[next], v[prev]; 954 row reads per tile against
    // 648 for the generic halo) cost more than the LDS reads and instructions they save.  Kept because it is parity-green
    // and the natural form once G^f / G^b come from a streaming pre-pass.  Needs every damper to join the same two
    // vertices' successors -- next(1c) = 1n, next(2c) = 2n, consistently over all dampers -- a BA window without masks,
    // offsets or unary dampers, and the two-kernel path (the fused single-launch iteration re-derives u for halo rows
    // and keeps the generic four-vertex records)
    std::vector<int> nxt_row, prv_row;
    {
        const int fused_max0 = c->env("NRS_FUSED_MAX_ROWS") ? atoi(c->env("NRS_FUSED_MAX_ROWS")) : 32768;
        bool plain = !s.X0 && s.n_un == 0 && s.n_dm > 0 && !s.sp_active && !s.dm_active && !s.pose_fixed && !c->env("NRS_NO_EDGE_CHI") && c->env("NRS_DFORM") &&
                     !c->env("NRS_NO_LDS") && !s.force_gather;
        for (int v = 0; v < s.M && plain; ++v) plain = !(s.rflag[v] & RF_FIXED);
        const bool two_kernel = d.n_rows >= fused_max0 || c->env("NRS_NO_FUSED") || (c->comm && s.shard);
        if (plain && two_kernel) {
            nxt_row.assign(d.n_rows, -1); prv_row.assign(d.n_rows, -1);
            for (int q = 0; q < s.n_dm && plain; ++q) {
                int r4[4];
                for (int k = 0; k < 4; ++k) { r4[k] = s.dm_idx[4 * (size_t)q + k] >= 0 ? e->vrow[s.dm_idx[4 * (size_t)q + k]] : -1; plain = plain && r4[k] >= 0; }
                if (!plain) break;
                for (int h = 0; h < 2 && plain; ++h) {
                    const int cur = r4[h], nx = r4[2 + h];
                    if ((nxt_row[cur] >= 0 && nxt_row[cur] != nx) || (prv_row[nx] >= 0 && prv_row[nx] != cur) || cur == nx) plain = false;
                    nxt_row[cur] = nx; prv_row[nx] = cur;
                }
            }
            d.dform = plain ? 1 : 0;
        }
        if (!d.dform) { nxt_row.clear(); prv_row.clear(); }
    }
    // ---- LDS staging: per workgroup (= 4 slices = BLK/T rows) the sorted list of rows referenced
    // outside the tile; neighbour ids become tile-local
    d.tile_rows = BLK / T;
    std::vector<int> halo_ptr(d.n_regblk + 1, 0), halo_rows, halo_ns(d.n_regblk, 0);
    d.max_halo_s = 0;
    std::vector<int> L_s(nnz_s, -1), L_d(3 * nnz_d, -1);      // tile-local ids
    {
        // tiles are independent: a few host threads each take a contiguous range of tiles
        int nt = std::max(1, std::min({8, (int)std::thread::hardware_concurrency(), d.n_regblk / 32}));   // (a 4.5k-point frame: 4 threads, 1.5 -> 0.5 ms)
        if (const char* ev = c->env("NRS_HOST_THREADS")) nt = std::max(1, std::min({64, atoi(ev), std::max(1, d.n_regblk)}));
        std::vector<std::vector<int>> part(nt);
        std::vector<int> cnt(d.n_regblk, 0);
        auto work = [&](int ti) {
            const int b0 = (int)((int64_t)d.n_regblk * ti / nt), b1 = (int)((int64_t)d.n_regblk * (ti + 1) / nt);
            std::vector<int> stamp(d.n_rows, -1), local(d.n_rows, 0), ext;
            for (int b = b0; b < b1; ++b) {
                const int row0 = b * d.tile_rows, row1 = row0 + d.tile_rows;
                ext.clear();
                const size_t s0 = (size_t)ss_ptr[b * 4], s1 = (size_t)ss_ptr[b * 4 + 4];
                const size_t d0 = (size_t)sd_ptr[b * 4], d1 = (size_t)sd_ptr[b * 4 + 4];
                auto see = [&](int o) {
                    if (o >= 0 && (o < row0 || o >= row1) && stamp[o] != b) { stamp[o] = b; ext.push_back(o); }
                };
                // spring neighbours first (the SpMV stages positions for them only), then damper-only rows
                for (size_t p2 = s0; p2 < s1; ++p2) see(S_other[p2]);
                const size_t ns = ext.size();
                if (d.dform) {                                     // the partner in the same keyframe: 2c / 1c (slot 0) or 2n / 1n (slot 2)
                    for (size_t p2 = d0; p2 < d1; ++p2)
                        if (D_role[p2] >= 0) see(D_o[3 * p2 + (D_role[p2] < 2 ? 0 : 2)]);
                } else
                    for (size_t p2 = 3 * d0; p2 < 3 * d1; ++p2) see(D_o[p2]);
                std::sort(ext.begin(), ext.begin() + ns);
                std::sort(ext.begin() + ns, ext.end());
                halo_ns[b] = (int)ns;
                for (size_t i = 0; i < ext.size(); ++i) local[ext[i]] = d.tile_rows + (int)i;
                auto loc = [&](int o) { return o < 0 ? -1 : (o >= row0 && o < row1) ? o - row0 : local[o]; };
                for (size_t p2 = s0; p2 < s1; ++p2) L_s[p2] = loc(S_other[p2]);
                for (size_t p2 = 3 * d0; p2 < 3 * d1; ++p2) L_d[p2] = loc(D_o[p2]);
                part[ti].insert(part[ti].end(), ext.begin(), ext.end());
                cnt[b] = (int)ext.size();
            }
        };
        mark("halo prep");
        parallel_for(nt, [&](int ti, int) { work(ti); });          // (a share runs inline when no thread can be created)
        mark("halo work");
        for (int b = 0; b < d.n_regblk; ++b) {
            halo_ptr[b + 1] = halo_ptr[b] + cnt[b];
            d.max_halo = std::max(d.max_halo, cnt[b]);
            d.max_halo_s = std::max(d.max_halo_s, halo_ns[b]);
        }
        halo_rows.reserve((size_t)halo_ptr[d.n_regblk]);
        for (int ti = 0; ti < nt; ++ti) halo_rows.insert(halo_rows.end(), part[ti].begin(), part[ti].end());
    }
    mark("halo lists");
    // ---- tile classes: if a few tiles have much larger halos than the rest they get their own
    // launch (class 1) with their own LDS size, and the bulk (class 0) keeps its occupancy
    std::vector<int> tile_list(d.n_regblk);
    {
        std::vector<int> hs(d.n_regblk);
        for (int b = 0; b < d.n_regblk; ++b) hs[b] = halo_ptr[b + 1] - halo_ptr[b];
        std::vector<int> sorted = hs;
        std::sort(sorted.begin(), sorted.end());
        int cut = d.max_halo;
        if (d.n_regblk >= 1024) {                                  // small problems are latency-bound: one launch
            // (the second launch has to fill the chip by itself: >= 4 workgroups per CU, or be needed
            // for the bulk to fit the LDS budget at all)
            const int p97 = sorted[(size_t)(0.97 * (d.n_regblk - 1))];
            const bool fits = sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.max_halo + d.max_halo_s + 2) <= 48 * 1024;
            if (4 * d.max_halo > 5 * p97 && (d.n_regblk - (int)(0.97 * d.n_regblk) >= 1024 || !fits) && !c->env("NRS_ONE_CLASS")) cut = p97;
        }
        if (c->env("NRS_TILE_CUT_PCT")) {                          // test switch: force a split at a percentile
            const double pct = atof(c->env("NRS_TILE_CUT_PCT")) / 100.0;
            cut = sorted[(size_t)(pct * (d.n_regblk - 1))];
        }
        int n0 = 0;
        for (int b = 0; b < d.n_regblk; ++b) if (hs[b] <= cut) tile_list[n0++] = b;
        int n1 = n0;
        for (int b = 0; b < d.n_regblk; ++b) if (hs[b] > cut) tile_list[n1++] = b;
        d.n_tiles_cls[0] = n0; d.n_tiles_cls[1] = d.n_regblk - n0;
        d.cap_h[0] = d.cap_h[1] = d.cap_s[0] = d.cap_s[1] = 0;
        for (int b = 0; b < d.n_regblk; ++b) {
            const int cls = hs[b] <= cut ? 0 : 1;
            d.cap_h[cls] = std::max(d.cap_h[cls], hs[b]);
            d.cap_s[cls] = std::max(d.cap_s[cls], halo_ns[b]);
        }
    }
    d.use_lds = 1;
    size_t lds_need = 0;
    for (int cls = 0; cls < 2; ++cls) {
        if (!d.n_tiles_cls[cls]) continue;
        if (d.dform) {
            lds_need = std::max(lds_need, sizeof(double) * 9 * (size_t)(d.tile_rows + d.cap_h[cls] + 1));                                    // linearise: x, G^f, G^b
            lds_need = std::max(lds_need, sizeof(double) * 3 * (3 * (size_t)(d.tile_rows + d.cap_h[cls] + 1) + d.tile_rows + d.cap_s[cls] + 1));  // operator: u, G^f, G^b + positions
            continue;
        }
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(d.tile_rows + d.cap_h[cls]) * (s.X0 ? 2 : 1));                          // linearise
        lds_need = std::max(lds_need, sizeof(double) * 3 * (size_t)(2 * d.tile_rows + d.cap_h[cls] + d.cap_s[cls] + 2));                    // operator: u + positions
    }
    if (c->env("NRS_NO_LDS") || s.force_gather || lds_need > 64 * 1024 - 512 || d.tile_rows + d.max_halo >= 65535) d.use_lds = 0;   // irregular graph / A-B switch
    if (!d.use_lds) d.dform = 0;
    d.lin_rb = d.use_lds ? ROW_ALIGN / d.tile_rows : 1;              // lineariser partials: per tile (LDS path) or per group
    // single-launch PCG iteration for problems that are bound by launch latency, not by traffic
    const int fused_max = c->env("NRS_FUSED_MAX_ROWS") ? atoi(c->env("NRS_FUSED_MAX_ROWS")) : 32768;
    d.fused = (d.use_lds && d.n_rows < fused_max && !c->env("NRS_NO_FUSED")) ? 1 : 0;
    if (s.n_skin > 0) d.fused = 0;                                 // embedded mode: the skinned observations' operator kernels sit between the two launches of an iteration
    d.hier = (d.n_regblk > 4096 || c->env("NRS_HIER")) ? 1 : 0;
    // (a profiling context times full operator launches only: no convergence-detecting early exits)
    d.ecd = (d.use_lds && !d.fused && !c->opt.profile && !c->env("NRS_NO_ECD")) ? 1 : 0;
    // two-level preconditioner: fused path, one pose, small enough coarse system
    d.co_n = 3 * d.n_groups + 6;
    // (worth its per-iteration cost on the pose + deformation problems; the lost-point stage, pose
    // fixed and few free rows, converges in a few dozen block-Jacobi iterations anyway)
    const bool pose_free = !(s.pose_fixed && s.pose_fixed[0]);
    const size_t fused_shm = sizeof(double) * (6 * (size_t)(d.tile_rows + d.max_halo) + 12 * (size_t)d.n_regblk + 16 * CO_MAX);
    // (and only from ~1.5k rows on: below, its per-iteration cost outweighs the iterations it saves -- 1013 points 31.3 ms with it,
    // 29.2 without; 2220 points 47.4 / 51.8; 4525 points 76.7 / 94.1, tools/small_frame_probe.py)
    const int co_min_tiles = c->env("NRS_COARSE_MIN_TILES") ? atoi(c->env("NRS_COARSE_MIN_TILES")) : 48;
    d.coarse = (d.fused && s.K == 1 && pose_free && d.co_n <= CO_MAX && d.n_regblk <= BLK && d.n_regblk >= co_min_tiles && fused_shm <= 63 * 1024 &&
                !c->env("NRS_NO_COARSE")) ? 1 : 0;
    // ---- shard window: the whole problem, or this rank's contiguous range of poses (balanced by rows)
    d.sh_on = 0; d.sh_rank = 0; d.sh_world = 1; d.sh_lead = 1;
    d.sh_k0 = 0; d.sh_nk = s.K; d.sh_g0 = 0; d.sh_ng = d.n_groups; d.sh_vb0 = 0; d.sh_nvb = d.n_vecblk;
    for (int cls = 0; cls < 2; ++cls) {
        d.sh_t0[cls] = 0; d.sh_nt[cls] = d.n_tiles_cls[cls];
        d.sh_t0b[cls] = d.sh_ntb[cls] = d.sh_front[cls] = d.sh_back[cls] = 0;
    }
    if (c->comm && s.shard) {
        const int W = c->comm->world, rk = c->comm->rank;
        // (a rank packs its own keyframe range only, so the halo sizes -- and with them this decision -- are rank-local)
        if (!d.use_lds) return c->fail(NRS_ERR_INVALID, "sharded solve: the graph's halo does not fit the LDS-staged path");
        std::vector<int> kb(W + 1);
        shard_plan(s.K, pose_grp_ptr.data(), W, kb.data());
        d.sh_on = 1; d.sh_rank = rk; d.sh_world = W; d.sh_lead = rk == 0;
        d.sh_k0 = kb[rk]; d.sh_nk = kb[rk + 1] - kb[rk];
        d.sh_g0 = pose_grp_ptr[kb[rk]]; d.sh_ng = pose_grp_ptr[kb[rk + 1]] - d.sh_g0;
        d.sh_vb0 = d.sh_g0 * (ROW_ALIGN / BLK); d.sh_nvb = d.sh_ng * (ROW_ALIGN / BLK);
        if ((int64_t)d.sh_nvb * BLK < s.K) return c->fail(NRS_ERR_INVALID, "sharded solve: shard smaller than the pose count");
        const int tb0 = d.sh_g0 * (ROW_ALIGN / d.tile_rows), tb1 = (d.sh_g0 + d.sh_ng) * (ROW_ALIGN / d.tile_rows);
        for (int cls = 0; cls < 2; ++cls) {                       // tile_list is ascending inside a class
            const int* tl = tile_list.data() + (cls ? d.n_tiles_cls[0] : 0);
            const int n = d.n_tiles_cls[cls];
            const int a = (int)(std::lower_bound(tl, tl + n, tb0) - tl), b2 = (int)(std::lower_bound(tl, tl + n, tb1) - tl);
            d.sh_t0[cls] = a; d.sh_nt[cls] = b2 - a;
        }
        // everything the own tiles reference must be owned or lie in the keyframe next to the range
        const int r_lo = (kb[rk] > 0 ? pose_grp_ptr[kb[rk] - 1] : d.sh_g0) * ROW_ALIGN;
        const int r_hi = (kb[rk + 1] < s.K ? pose_grp_ptr[kb[rk + 1] + 1] : d.sh_g0 + d.sh_ng) * ROW_ALIGN;
        for (int b = tb0; b < tb1; ++b)
            for (int i = halo_ptr[b]; i < halo_ptr[b + 1]; ++i)
                if (halo_rows[i] < r_lo || halo_rows[i] >= r_hi)
                    return c->fail(NRS_ERR_INVALID, "sharded solve: an edge of keyframe range [%d, %d) reaches beyond the adjacent keyframes", kb[rk], kb[rk + 1]);
        // ... so the rank holds the per-row arrays (state, vectors, diagonal blocks: ~410 bytes a row) of its own keyframes and of ONE ghost
        // keyframe either side only: its tiles' halos end there (just checked), the boundary exchange fills the ghosts, and no launch of
        // this rank touches a row beyond them (ArenaPlan::get_rows).  NRS_SHARD_FULL_VECTORS=1: every row, the round-1..4 form.
        if (W > 1 && !d.dform && !c->env("NRS_SHARD_FULL_VECTORS")) { d.row_lo = r_lo; d.row_hi = r_hi; }
        // boundary tiles (their halo holds rows of another rank) sit at the two ends of the rank's tile range:
        // they run after the interior tiles, once the neighbours' rows have arrived
        const int own_lo = d.sh_g0 * ROW_ALIGN, own_hi = (d.sh_g0 + d.sh_ng) * ROW_ALIGN;
        auto outside = [&](int r) { return r >= 0 && (r < own_lo || r >= own_hi); };
        auto foreign = [&](int b) {
            for (int i = halo_ptr[b]; i < halo_ptr[b + 1]; ++i) {
                if (outside(halo_rows[i])) return true;
                if (d.dform && (outside(nxt_row[halo_rows[i]]) || outside(prv_row[halo_rows[i]]))) return true;
            }
            if (d.dform)
                for (int r = b * d.tile_rows; r < (b + 1) * d.tile_rows; ++r)
                    if (outside(nxt_row[r]) || outside(prv_row[r])) return true;
            return false;
        };
        for (int cls = 0; cls < 2; ++cls) {
            const int* tl = tile_list.data() + (cls ? d.n_tiles_cls[0] : 0) + d.sh_t0[cls];
            const int n = d.sh_nt[cls];
            int first = n, last = -1;                              // first / last own tile of the class that is interior
            for (int i = 0; i < n; ++i) if (!foreign(tl[i])) { first = i; break; }
            for (int i = n - 1; i >= 0; --i) if (!foreign(tl[i])) { last = i; break; }
            if (last < first) { d.sh_front[cls] = n; d.sh_back[cls] = 0; continue; }          // no interior tile at all
            bool clean = true;                                     // (dampers reach one keyframe: the middle is interior)
            for (int i = first; i <= last && clean; ++i) clean = !foreign(tl[i]);
            if (!clean) { d.sh_front[cls] = n; d.sh_back[cls] = 0; continue; }
            d.sh_front[cls] = first; d.sh_back[cls] = n - 1 - last;
        }
        HaloPlan& h = e->halo;
        auto rows_of = [&](int k, size_t& off, size_t& n) { off = 3 * (size_t)pose_grp_ptr[k] * ROW_ALIGN; n = 3 * (size_t)(pose_grp_ptr[k + 1] - pose_grp_ptr[k]) * ROW_ALIGN; };
        if (rk > 0) { rows_of(kb[rk], h.lo_send, h.lo_send_n); rows_of(kb[rk] - 1, h.lo_recv, h.lo_recv_n); }
        if (rk < W - 1) { rows_of(kb[rk + 1] - 1, h.hi_send, h.hi_send_n); rows_of(kb[rk + 1], h.hi_recv, h.hi_recv_n); }
        d.fused = 0; d.coarse = 0; d.ecd = 0; d.hier = 1;
    }
    mark("halo");
    if (tm) fprintf(stderr, "[nrs] tiles %d x %d rows (T=%d), halo rows: max %d, mean %.1f, spring part max %d, classes %d (cap %d/%d) + %d (cap %d/%d), lds %d, fused %d\n", d.n_regblk, d.tile_rows, T, d.max_halo, (double)halo_rows.size() / d.n_regblk, d.max_halo_s, d.n_tiles_cls[0], d.cap_h[0], d.cap_s[0], d.n_tiles_cls[1], d.cap_h[1], d.cap_s[1], d.use_lds, d.fused);
    if (tm) fprintf(stderr, "[nrs] coarse level: wanted %d (fused %d, K %d, unknowns %d <= %d), enabled %d\n", d.fused && s.K == 1, d.fused, s.K, 3 * d.n_groups + 6, CO_MAX, d.coarse);
    // ---- edge lists for the chi2-only evaluation of trial states (BA form, nothing masked or fixed): each
    // edge once, ordered by the row that counts it (locality of the gathers); a rank keeps the edges it counts
    std::vector<uint32_t> row_tp;
    std::vector<EcSpring> ec_sp;
    std::vector<EcDamper> ec_dm;
    std::vector<float> ec_w;
    {
        bool plain = !s.X0 && s.n_un == 0 && !s.sp_active && !s.dm_active && !s.pose_fixed && !c->env("NRS_NO_EDGE_CHI");
        for (int v = 0; v < s.M && plain; ++v) plain = !(s.rflag[v] & RF_FIXED);
        d.ec_on = plain ? 1 : 0;
        {   // the specialised lineariser additionally wants every damper with its four vertices and springs without a kernel
            bool p4 = plain && d.use_lds && !d.dform && !(s.delta_pos > 0) && s.spring_form == 0 && !c->env("NRS_NO_PLAIN");
            for (int64_t q = 0; q < 4 * (int64_t)s.n_dm && p4; ++q) p4 = s.dm_idx[q] >= 0;
            d.plain = p4 ? 1 : 0;
            if (d.plain) d.lin_rb = ROW_ALIGN / (64 / T);          // k_lin_plain leaves one partial slot per SLICE (no workgroup barrier behind its loops)
        }
        if (plain) {
            const int own_lo = d.sh_g0 * ROW_ALIGN, own_hi = (d.sh_g0 + d.sh_ng) * ROW_ALIGN;
            // counting sort by the counting row (stable: edges of a row keep their order); threads: keys and counts are
            // order-free, the scatter gives every thread a range of rows and scans the keys in edge order
            std::vector<int> key, pos(d.n_rows + 1);
            const int nt_ec = nt_all;
            auto order_by_row = [&](int n_edges, auto row_of) {
                key.assign(n_edges, -1);
                std::fill(pos.begin(), pos.end(), 0);
                parallel_for(nt_ec, [&](int ti, int n) {
                    int64_t a, b;
                    chunk(n_edges, ti, n, a, b);
                    for (int64_t q = a; q < b; ++q) {
                        const int r = row_of((int)q);
                        if (r >= own_lo && r < own_hi) { key[q] = r; count_up(&pos[r + 1], n); }
                    }
                });
                for (int r = 0; r < d.n_rows; ++r) pos[r + 1] += pos[r];
                std::vector<int> out(pos[d.n_rows]);
                std::vector<int> cut(nt_ec + 1, d.n_rows);
                cut[0] = 0;
                for (int t = 1, r = 0; t < nt_ec; ++t) {
                    const int64_t want = (int64_t)out.size() * t / nt_ec;
                    while (r < d.n_rows && pos[r] < want) ++r;
                    cut[t] = r;
                }
                parallel_for(nt_ec, [&](int ti, int) {
                    const int lo = cut[ti], hi = cut[ti + 1];
                    if (lo >= hi) return;
                    for (int q = 0; q < n_edges; ++q)
                        if (key[q] >= lo && key[q] < hi) out[pos[key[q]]++] = q;
                });
                return out;
            };
            const std::vector<int> so = order_by_row(s.n_sp, [&](int q) { return sp_row[2 * (size_t)q]; });
            ec_sp.resize(so.size());
            parallel_for(nt_ec, [&](int ti, int n) {
                int64_t a, b;
                chunk((int64_t)so.size(), ti, n, a, b);
                for (int64_t i = a; i < b; ++i) { const int q = so[i]; ec_sp[i] = EcSpring{sp_row[2 * (size_t)q], sp_row[2 * (size_t)q + 1], s.sp_d0[q], 0}; }
            });
            const std::vector<int> dord = order_by_row(s.n_dm, [&](int q) {
                for (int k = 0; k < 4; ++k) if (dm_row[4 * (size_t)q + k] >= 0) return dm_row[4 * (size_t)q + k];
                return -1;
            });
            ec_dm.resize(dord.size()); ec_w.resize(dord.size());
            parallel_for(nt_ec, [&](int ti, int n) {
                int64_t a, b;
                chunk((int64_t)dord.size(), ti, n, a, b);
                for (int64_t i = a; i < b; ++i) {
                    const int q = dord[i];
                    for (int k = 0; k < 4; ++k) ec_dm[i].r[k] = dm_row[4 * (size_t)q + k];
                    ec_w[i] = s.dm_w[q];
                }
            });
        }
        // temporal partners of every row (plain windows): from the dampers' canonical second vertex; all dampers of a row and
    // direction must agree (they do for the reference's BA dampers), else the kernels read it per incidence as before
    d.tp_ok = 0; d.h4 = 0;
    if (d.plain) {
        static const int perm1[4] = {1, 2, 0, 1};                  // (perm[role][1] of the canonical order below)
        row_tp.assign((size_t)d.n_rows, 0xFFFFFFFFu);
        bool ok = true;
        for (int r = 0; r < d.n_rows && ok; ++r)
            for (int k = 0; k < cnt_d[r] && ok; ++k) {
                const size_t pz = pos_of(sd_ptr, r, k);
                const int role = D_role[pz];
                const uint32_t l = (uint32_t)(L_d[3 * pz + perm1[role]] & 0xFFFF);
                const int sh = role < 2 ? 0 : 16;
                const uint32_t cur = (row_tp[r] >> sh) & 0xFFFFu;
                if (cur != 0xFFFFu && cur != l) ok = false;
                row_tp[r] = (row_tp[r] & ~(0xFFFFu << sh)) | (l << sh);
            }
        d.tp_ok = ok ? 1 : 0;
    }
    d.ec_nsp = (int)ec_sp.size(); d.ec_ndm = (int)ec_dm.size();
        d.ec_nblk = std::min((d.ec_nsp + d.ec_ndm + BLK - 1) / BLK, 2048);
    }
    mark("edge lists");
    // ---- device memory: one arena allocation, reused across calls when large enough
    ArenaPlan dry{arena, true};
    {
        Dev tmp = d;
        Engine te;
        te.n_spec = e->n_spec;
        carve(dry, tmp, s.X0 != nullptr, nnz_s, nnz_d, ss_ptr.size() - 1, halo_rows.size(), &te);
    }
    if (dry.off > arena->cap) {
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        arena_release(arena);
        const size_t want = dry.off + dry.off / 8;
        hipError_t he = hipMalloc((void**)&arena->base, want);
        if (he != hipSuccess) return c->fail(NRS_ERR_ALLOC, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(he));
        arena->cap = want;
        if (c->env("NRS_POISON")) { (void)hipMemset(arena->base, 0xFF, want); (void)hipDeviceSynchronize(); }    // (debug: a read of memory nobody wrote shows up as NaN)
    }
    ArenaPlan real{arena, false};
    carve(real, d, s.X0 != nullptr, nnz_s, nnz_d, ss_ptr.size() - 1, halo_rows.size(), e);
    e->arena_bytes = real.off;

    mark("arena");
    // ---- host mirrors + uploads
    e->sp_ij.assign(s.sp_ij, s.sp_ij + 2 * (size_t)s.n_sp);
    e->sp_d0.assign(s.sp_d0, s.sp_d0 + (size_t)s.n_sp);
    e->dm_idx.assign(s.dm_idx, s.dm_idx + 4 * (size_t)s.n_dm);
    e->dm_w.assign(s.dm_w, s.dm_w + (size_t)s.n_dm);
    e->un_ij.assign(s.un_ij, s.un_ij + 2 * (size_t)s.n_un);
    e->un_w.assign(s.un_w, s.un_w + (size_t)s.n_un);
    e->h_rflag.assign(d.n_rows, RF_FIXED);            // padding rows: no edges, never move
    for (int v = 0; v < s.M; ++v) e->h_rflag[e->vrow[v]] = s.rflag[v];
    e->h_pose_fixed.assign(s.K, 0);
    if (s.pose_fixed) e->h_pose_fixed.assign(s.pose_fixed, s.pose_fixed + s.K);
    std::vector<float> uv((size_t)d.n_rows * 2, 0.f);
    std::vector<double> xl((size_t)d.n_rows * 3, 0.0), X0;
    if (s.X0) X0.assign((size_t)d.n_rows * 3, 0.0);
    for (int v = 0; v < s.M; ++v) {
        const size_t row = (size_t)e->vrow[v];
        uv[2 * row] = s.uv[2 * v];
        uv[2 * row + 1] = s.uv[2 * v + 1];
        for (int k = 0; k < 3; ++k) {
            xl[3 * row + k] = s.x[3 * (size_t)v + k];
            if (s.X0) X0[3 * row + k] = s.X0[3 * (size_t)v + k];
        }
    }
    e->h_s_meta.assign(nnz_s, 0);
    e->h_d_meta.assign(nnz_d, -1);
    const int nt_mir = nt_all;
    parallel_for(nt_mir, [&](int ti, int n) {
        int64_t a, b;
        chunk((int64_t)nnz_d, ti, n, a, b);
        for (int64_t i = a; i < b; ++i)
            if (D_role[i] >= 0) e->h_d_meta[i] = D_role[i];
    });
    std::vector<int> d_o0, d_o1, d_o2;
    if (d.use_lds) {
        auto u16 = [](int v) { return v < 0 ? (uint32_t)REC_NONE : (uint32_t)(v & 0xFFFF); };
        e->h_s_om.resize(nnz_s);
        parallel_for(nt_mir, [&](int ti, int n) {
            int64_t a, b;
            chunk((int64_t)nnz_s, ti, n, a, b);
            for (int64_t i = a; i < b; ++i) e->h_s_om[i] = u16(L_s[i]);                        // meta: push_masks
        });
        if (d.dform) {
            e->h_d_om.resize(nnz_d);
            for (size_t i = 0; i < nnz_d; ++i) e->h_d_om[i] = D_role[i] < 0 ? (uint32_t)REC_NONE : u16(L_d[3 * i + (D_role[i] < 2 ? 0 : 2)]);
        } else {
            // canonical order of the three other vertices, so that every role evaluates the same expression
            //   g = (v_i - v[o1]) - (v[o0] - v[o2])   (= sg_i * sum_k sg_k v_k; absent vertices read zeros):
            // o0 = the partner in the same keyframe, o1 = the own temporal partner, o2 = the partner's temporal partner
            static const int perm[4][3] = {{0, 1, 2}, {0, 2, 1}, {2, 0, 1}, {2, 1, 0}};    // indices into the others in ascending role order
            e->h_d_hdr.resize(nnz_d);
            parallel_for(nt_mir, [&](int ti, int n) {
                int64_t a, b;
                chunk((int64_t)nnz_d, ti, n, a, b);
                for (int64_t i = a; i < b; ++i) {
                    const int* pm = perm[D_role[i] < 0 ? 0 : D_role[i]];
                    e->h_d_hdr[i] = make_uint2(u16(L_d[3 * i + pm[0]]) | (u16(L_d[3 * i + pm[1]]) << 16), u16(L_d[3 * i + pm[2]]) | ((uint32_t)REC_NONE << 16));
                }
            });
        }
    } else {
        d_o0.resize(nnz_d); d_o1.resize(nnz_d); d_o2.resize(nnz_d);
        for (size_t i = 0; i < nnz_d; ++i) { d_o0[i] = D_o[3 * i]; d_o1[i] = D_o[3 * i + 1]; d_o2[i] = D_o[3 * i + 2]; }
    }
    const std::vector<int>& s_other = S_other;
    const std::vector<float>& s_d0 = S_d0;
    const std::vector<float>& d_w = D_w;
    mark("host mirrors");
    std::vector<Pose> poses(s.poses, s.poses + s.K);
    NRS_TRY(h2d(c, d.grp_pose, grp_pose));
    NRS_TRY(h2d(c, d.pose_grp_ptr, pose_grp_ptr));
    NRS_TRY(h2d_rows(c, d, d.uv, uv, 2));
    if (d.row_hi - d.row_lo < d.n_rows) e->h_uv = uv;              // (a row-limited rank: the residual taps stage the observations of every row from here)
    NRS_TRY(h2d_rows(c, d, d.xl_init, xl, 3));
    if (s.X0) NRS_TRY(h2d_rows(c, d, d.X0, X0, 3));
    NRS_TRY(h2d(c, d.pose_init, poses));
    NRS_TRY(h2d(c, d.ss_ptr, ss_ptr));
    NRS_TRY(h2d(c, d.sd_ptr, sd_ptr));
    NRS_TRY(h2d(c, d.halo_ptr, halo_ptr));
    NRS_TRY(h2d(c, d.halo_rows, halo_rows));
    NRS_TRY(h2d(c, d.halo_ns, halo_ns));
    NRS_TRY(h2d(c, d.tile_list, tile_list));
    if (d.dform) {
        std::vector<int> hn(halo_rows.size()), hp(halo_rows.size());
        for (size_t i = 0; i < halo_rows.size(); ++i) { hn[i] = nxt_row[halo_rows[i]]; hp[i] = prv_row[halo_rows[i]]; }
        NRS_TRY(h2d(c, d.nxt_row, nxt_row));
        NRS_TRY(h2d(c, d.prv_row, prv_row));
        NRS_TRY(h2d(c, d.halo_nxt, hn));
        NRS_TRY(h2d(c, d.halo_prv, hp));
        NRS_HIP(c, hipStreamSynchronize(c->stream));               // hn / hp die here
    }
    if (d.fused) {
        std::vector<int> tile_desc(8 * (size_t)d.n_regblk, 0), halo_fix((size_t)BLK * d.n_regblk, 0);
        const int rb = ROW_ALIGN / d.tile_rows;
        for (int b = 0; b < d.n_regblk; ++b) {
            const int kf = grp_pose[(size_t)b * d.tile_rows / ROW_ALIGN];
            int* td = &tile_desc[8 * (size_t)b];
            td[0] = kf; td[1] = pose_grp_ptr[kf] * rb; td[2] = pose_grp_ptr[kf + 1] * rb;
            td[3] = halo_ptr[b]; td[4] = halo_ptr[b + 1] - halo_ptr[b];
            for (int i = 0; i < td[4] && i < BLK; ++i) halo_fix[(size_t)b * BLK + i] = halo_rows[td[3] + i];
        }
        NRS_TRY(h2d(c, d.tile_desc, tile_desc));
        NRS_TRY(h2d(c, d.halo_fix, halo_fix));
    } else if (d.plain && d.use_lds) {                             // stage_rows<true>: the first HALO_FIX halo rows at a fixed stride
        std::vector<int> halo_fix((size_t)HALO_FIX * d.n_regblk, -1);
        for (int b = 0; b < d.n_regblk; ++b) {
            const int hn = std::min(halo_ptr[b + 1] - halo_ptr[b], HALO_FIX);
            for (int i = 0; i < hn; ++i) halo_fix[(size_t)b * HALO_FIX + i] = halo_rows[halo_ptr[b] + i];
        }
        NRS_TRY(h2d(c, d.halo_fix, halo_fix));
    }
    NRS_TRY(h2d(c, d.s_d0, s_d0));
    if (d.use_lds) {                                               // padding slots stay zero
        NRS_HIP(c, hipMemsetAsync(d.s_qc, 0, sizeof(double) * nnz_s, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.d_s, 0, sizeof(double) * nnz_d, c->stream));
    }
    if (!d.use_lds) {
        NRS_TRY(h2d(c, d.s_other, s_other));
        NRS_TRY(h2d(c, d.d_o0, d_o0));
        NRS_TRY(h2d(c, d.d_o1, d_o1));
        NRS_TRY(h2d(c, d.d_o2, d_o2));
    }
    NRS_TRY(h2d(c, d.d_w, d_w));
    if (d.plain) NRS_TRY(h2d_rows(c, d, d.row_tp, row_tp, 1));
    if (d.plain) {
        std::vector<uint32_t> rc((size_t)d.n_rows);
        for (int r = 0; r < d.n_rows; ++r) rc[r] = (uint32_t)cnt_s[r] | ((uint32_t)cnt_d[r] << 16);
        NRS_TRY(h2d_rows(c, d, d.row_cnt, rc, 1));
        NRS_HIP(c, hipStreamSynchronize(c->stream));               // (rc dies here)
    }
    if (d.ec_on) {
        NRS_TRY(h2d(c, d.ec_sp, ec_sp));
        NRS_TRY(h2d(c, d.ec_dm, ec_dm));
        NRS_TRY(h2d(c, d.ec_w, ec_w));
    }
    NRS_TRY(push_masks(c, e, s.sp_active, s.dm_active));
    e->serial = ++c->engine_serial;                                // (the residual taps are staged on first use: engine_residuals)
    NRS_HIP(c, hipMemsetAsync(d.part_apply, 0, sizeof(double) * (size_t)d.n_vecblk, c->stream));
    if (d.sh_on) {                                                // slots of other ranks' tiles are never written: zero for good
        NRS_HIP(c, hipMemsetAsync(d.part_lin, 0, sizeof(double) * 32 * (size_t)d.n_groups * (size_t)d.lin_rb, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.part_rchi, 0, sizeof(double) * (size_t)d.n_groups, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.part_reg, 0, sizeof(double) * 2 * (size_t)d.n_regblk, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.part_spmv, 0, sizeof(double) * NPART * (size_t)d.n_regblk, c->stream));
        NRS_HIP(c, hipMemsetAsync(d.red, 0, sizeof(double) * (4 + 6 * (size_t)d.K), c->stream));
        NRS_HIP(c, hipMemsetAsync(d.red_loc, 0, sizeof(double) * (4 + 6 * (size_t)d.K), c->stream));
    }
    NRS_HIP(c, hipMemsetAsync(d.scal, 0, sizeof(double) * SC_N, c->stream));
    NRS_HIP(c, hipMemsetAsync(d.flags, 0, sizeof(int) * 8, c->stream));
    mark("uploads enqueued");
    if (!c->pin_scal) NRS_HIP(c, hipHostMalloc((void**)&c->pin_scal, sizeof(double) * SC_N, hipHostMallocMapped | hipHostMallocCoherent));      // pinned mirrors live in
    if (!c->pin_flags) {                                                                                              // the context (reused)
        NRS_HIP(c, hipHostMalloc((void**)&c->pin_flags, sizeof(int) * 8, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->pin_flags, 0, sizeof(int) * 8);                  // [7] is the publication sequence word the host polls
    }
    e->h_scal = c->pin_scal;
    e->h_flags = c->pin_flags;
    e->d.h_scal = c->pin_scal;          // hipHostMalloc memory is mapped: same pointer on the device
    e->d.h_flags = c->pin_flags;
    if (e->n_spec > 0) NRS_TRY(spec_prepare(c, e));
    engine_compact_headers(c, e);
    NRS_HIP(c, hipStreamSynchronize(c->stream));       // host staging vectors die here
    mark("pinned+sync");
    if (s.n_skin > 0) {
        // ---- embedded mode: the skinned observations (nrs_engine_skin.hpp; device arrays in a buffer of the context).  A BA window (N2b:
        // sk_pose given, K poses) solves by the PCG with the observations applied as hyper-edges; a single-frame engine (N2a) by the direct
        // solver when it takes the frame (k_nd_values folds them into its blocks) and by the same PCG form when it does not.
        // Slots: observations grouped by pose (caller order inside a pose: K = 1 keeps the caller's order), every pose's padded to BLK;
        // per node row the list of the observations that reach it, in slot order.
        const bool ba_form = s.sk_pose != nullptr;
        if (d.fused || d.sh_on || !s.sk_uv || !s.sk_X0 || !s.sk_node || !s.sk_om) return c->fail(NRS_ERR_INVALID, "skinned observations: two-kernel PCG path, one GPU");
        if (!ba_form && !(arena == &c->arena_trk && s.K == 1)) return c->fail(NRS_ERR_INVALID, "skinned observations without a pose index: single-frame tracking engines only");
        const size_t n_in = (size_t)s.n_skin;
        std::vector<int> pose0;
        if (!ba_form) pose0.assign(n_in, 0);
        const int* sk_pose = ba_form ? s.sk_pose : pose0.data();
        std::vector<int> cnt(s.K + 1, 0), pose_blk(s.K + 1, 0);
        for (size_t i = 0; i < n_in; ++i) {
            if (sk_pose[i] < 0 || sk_pose[i] >= s.K) return c->fail(NRS_ERR_INVALID, "skinned observation: pose index out of range");
            cnt[sk_pose[i] + 1]++;
        }
        for (int k = 0; k < s.K; ++k) pose_blk[k + 1] = pose_blk[k] + (cnt[k + 1] + BLK - 1) / BLK;
        const size_t nblk = (size_t)pose_blk[s.K], n = nblk * BLK;
        std::vector<int> next(s.K), blk_pose(nblk);
        for (int k = 0; k < s.K; ++k) { next[k] = pose_blk[k] * BLK; for (int b2 = pose_blk[k]; b2 < pose_blk[k + 1]; ++b2) blk_pose[b2] = k; }
        e->sk_slot.resize(n_in);
        std::vector<float> uv(2 * n, 0.f);
        std::vector<double> X0(3 * n, 0.0), om(SK_MAX * n, 0.0);
        std::vector<int> rows(SK_MAX * n, -1);
        std::vector<uint8_t> act(n, 0);
        std::vector<int> rl_cnt(d.n_rows + 1, 0);
        for (size_t i = 0; i < n_in; ++i) {
            const size_t sl = (size_t)next[sk_pose[i]]++;
            e->sk_slot[i] = (int)sl;
            uv[2 * sl] = s.sk_uv[2 * i]; uv[2 * sl + 1] = s.sk_uv[2 * i + 1];
            for (int k = 0; k < 3; ++k) X0[3 * sl + k] = s.sk_X0[3 * i + k];
            act[sl] = 1;
            for (int k = 0; k < SK_MAX; ++k) {
                const int v = s.sk_node[SK_MAX * i + k];
                if (v < 0) continue;
                if (s.lm_pose[v] != sk_pose[i]) return c->fail(NRS_ERR_INVALID, "skinned observation: a node copy of another keyframe");
                rows[(size_t)k * n + sl] = e->vrow[v];              // (11 x n, node-slot-major: the kernels read them coalesced)
                om[(size_t)k * n + sl] = s.sk_om[SK_MAX * i + k];
                rl_cnt[e->vrow[v] + 1]++;
            }
        }
        // row lists (CSR over the rows that are reached), entries in slot order
        std::vector<int> rl_row, rl_ptr(1, 0), row_list(d.n_rows, -1);
        for (int r = 0; r < d.n_rows; ++r)
            if (rl_cnt[r + 1] > 0) { row_list[r] = (int)rl_row.size(); rl_row.push_back(r); rl_ptr.push_back(rl_ptr.back() + rl_cnt[r + 1]); }
        const size_t n_ent = (size_t)rl_ptr.back(), nrl = rl_row.size();
        std::vector<int> rl_obs(n_ent + 1), fill(rl_ptr.begin(), rl_ptr.end() - 1);
        std::vector<double> rl_om(n_ent + 1);
        for (size_t sl = 0; sl < n; ++sl)
            for (int k = 0; k < SK_MAX; ++k) {
                const int r = rows[(size_t)k * n + sl];
                if (r < 0) continue;
                const int q = fill[row_list[r]]++;
                rl_obs[q] = (int)sl; rl_om[q] = om[(size_t)k * n + sl];
            }
        auto al = [](size_t b2) { return (b2 + 255) & ~(size_t)255; };
        const size_t o_uv = 0, o_X0 = o_uv + al(8 * n), o_row = o_X0 + al(24 * n), o_om = o_row + al(4 * SK_MAX * n), o_act = o_om + al(8 * SK_MAX * n),
                     o_bp = o_act + al(n), o_pb = o_bp + al(4 * nblk), o_rr = o_pb + al(4 * (s.K + 1)), o_rp = o_rr + al(4 * (nrl + 1)), o_ro = o_rp + al(4 * (nrl + 1)),
                     o_rw = o_ro + al(4 * (n_ent + 1)), o_rec = o_rw + al(8 * (n_ent + 1)), o_part = o_rec + al(8 * 27 * n), o_chi = o_part + al(8 * 32 * nblk),
                     o_md = o_chi + al(8 * n), o_g = o_md + 256, o_op = o_g + al(8 * 4 * n), o_rq = o_op + al(8 * 8 * nblk), o_recT = o_rq + al(8 * (size_t)d.n_rows),
                     o_dop = o_recT + al(8 * 24 * n), o_spec = o_dop + (d.use_lds ? 0 : al(8 * 6 * (size_t)d.n_rows)),
                     spec_stride = al(8 * 32 * nblk) + al(8 * n), total = o_spec + (size_t)e->n_spec * spec_stride;   // (shadow sets of sk_part / sk_chi: speculative trials)
        DevBuf& buf = arena == &c->arena_trk ? c->nd_skin : c->dba_skin;
        NRS_TRY(c->ensure(buf, total));
        char* sb = buf.as<char>();
        auto up = [&](size_t off, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(sb + off, src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess; };
        NRS_HIP(c, up(o_uv, uv.data(), 8 * n)); NRS_HIP(c, up(o_X0, X0.data(), 24 * n)); NRS_HIP(c, up(o_row, rows.data(), 4 * SK_MAX * n));
        NRS_HIP(c, up(o_om, om.data(), 8 * SK_MAX * n)); NRS_HIP(c, up(o_act, act.data(), n)); NRS_HIP(c, up(o_bp, blk_pose.data(), 4 * nblk));
        NRS_HIP(c, up(o_pb, pose_blk.data(), 4 * (size_t)(s.K + 1))); NRS_HIP(c, up(o_rr, rl_row.data(), 4 * nrl)); NRS_HIP(c, up(o_rp, rl_ptr.data(), 4 * (nrl + 1)));
        NRS_HIP(c, up(o_ro, rl_obs.data(), 4 * n_ent)); NRS_HIP(c, up(o_rw, rl_om.data(), 8 * n_ent));
        NRS_HIP(c, hipMemsetAsync(sb + o_rec, 0, total - o_rec, c->stream));
        std::vector<int> row_q(2 * (size_t)d.n_rows, 0);           // per row: its list's range (k_pcg_update<true> / k_skin_op_rows go by rows)
        for (size_t l = 0; l < nrl; ++l) { row_q[2 * (size_t)rl_row[l]] = rl_ptr[l]; row_q[2 * (size_t)rl_row[l] + 1] = rl_ptr[l + 1]; }
        NRS_HIP(c, up(o_rq, row_q.data(), 8 * (size_t)d.n_rows));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        d.sk_n = (int)n; d.sk_nblk = (int)nblk; d.sk_pcg = 1;      // (a single-frame engine on the direct solver switches sk_pcg off below)
        d.sk_uv = reinterpret_cast<const float*>(sb + o_uv); d.sk_X0 = reinterpret_cast<const double*>(sb + o_X0);
        d.sk_row = reinterpret_cast<const int*>(sb + o_row); d.sk_om = reinterpret_cast<const double*>(sb + o_om);
        d.sk_active = reinterpret_cast<const uint8_t*>(sb + o_act);
        d.sk_blk_pose = reinterpret_cast<const int*>(sb + o_bp); d.sk_pose_blk = reinterpret_cast<const int*>(sb + o_pb);
        d.sk_nrl = (int)nrl; d.sk_rl_row = reinterpret_cast<const int*>(sb + o_rr); d.sk_rl_ptr = reinterpret_cast<const int*>(sb + o_rp);
        d.sk_rl_obs = reinterpret_cast<const int*>(sb + o_ro); d.sk_rl_om = reinterpret_cast<const double*>(sb + o_rw);
        d.sk_rec = reinterpret_cast<double*>(sb + o_rec); d.sk_part = reinterpret_cast<double*>(sb + o_part);
        d.sk_chi = reinterpret_cast<double*>(sb + o_chi); d.sk_maxdiag = reinterpret_cast<double*>(sb + o_md);
        d.sk_g = reinterpret_cast<double*>(sb + o_g); d.sk_opart = reinterpret_cast<double*>(sb + o_op); d.sk_row_q = reinterpret_cast<const int*>(sb + o_rq);
        d.sk_recT = reinterpret_cast<double*>(sb + o_recT);
        d.D_op = d.use_lds ? nullptr : reinterpret_cast<double*>(sb + o_dop);
        for (int j = 0; j < e->n_spec; ++j) {
            e->spec[j].sk_part = reinterpret_cast<double*>(sb + o_spec + (size_t)j * spec_stride);
            e->spec[j].sk_chi = reinterpret_cast<double*>(sb + o_spec + (size_t)j * spec_stride + al(8 * 32 * nblk));
        }
        d.sk_base = ba_form ? d.xl_init : nullptr;                 // (tracking form: the rows ARE the deformations, X0 + sum om x)
        e->sk_vert.assign(s.sk_node, s.sk_node + SK_MAX * n_in);
        e->sk_om.assign(s.sk_om, s.sk_om + SK_MAX * n_in);
        e->sk_X0.assign(s.sk_X0, s.sk_X0 + 3 * n_in);
    }
    if (e->nd) {                                                   // direct solve when the frame is small enough to gain from it
        NRS_TRY(nd_engine_finish(c, e, e->nd, nd_prep));
        mark("direct solve plan");
    }
    d.sk_pcg = (d.sk_n > 0 && !(e->nd && e->nd->on)) ? 1 : 0;      // (on the direct solver k_nd_values folds the observations into its blocks)
    if (d.sk_pcg) d.ecd = 0;                                       // (k_pcg_update<true> owns 16 rows a workgroup: no r.u partials per 256 rows for the operator's early test -- one launch in hundreds)
    if (d.sk_pcg && s.sk_pose && d.use_lds && !d.sh_on && !d.hier && c->opt.embedded_solver != 2) {   // embedded BA window: the keyframe-block factorisation as the PCG's preconditioner
        NRS_TRY(kft_setup(c, e, s, pose_grp_ptr));
        mark("keyframe-block factorisation plan");
    }
    NRS_TRY(engine_reset(c, e));
    guard.keep = true;
    *out = e;
    return NRS_OK;
}

void engine_stats(const Engine* e, int64_t stats[5]) {
    stats[0] = e->d.n_rows; stats[1] = e->pack_rows; stats[2] = e->d.ss_nnz; stats[3] = e->d.sd_nnz; stats[4] = (int64_t)e->arena_bytes;
}

void engine_destroy(nrs_ctx* c, Engine* e) {
    if (!e) return;
    (void)hipStreamSynchronize(c->stream);
    nd_engine_free(c, e->nd);
    delete e->kft;
    delete e;
}

int engine_update_flags(nrs_ctx* c, Engine* e, const uint8_t* rflag, const uint8_t* pose_fixed,
                        const uint8_t* sp_active, const uint8_t* dm_active) {
    // (checked before anything is touched: a rejected call leaves the engine as it was)
    if (e->d.plain) return c->fail(NRS_ERR_STATE, "masks on a plain BA window: not supported (its partial slots are per slice)");
    if (rflag)
        for (int v = 0; v < e->d.M; ++v) e->h_rflag[e->vrow[v]] = rflag[v];
    if (pose_fixed) e->h_pose_fixed.assign(pose_fixed, pose_fixed + e->d.K);
    e->d.ec_on = 0;                                                // masks / fixed vertices: chi2 comes from the incidence records
    NRS_TRY(push_masks(c, e, sp_active, dm_active));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    if (e->nd) {                                                   // the direct solver's plan is built on the free rows: a changed fixed set needs a new one
        bool same = e->nd->sig.size() == (size_t)e->d.M + 1 && e->nd->sig[e->d.M] == e->h_pose_fixed[0];
        for (int v = 0; v < e->d.M && same; ++v) same = e->nd->sig[v] == (e->h_rflag[e->vrow[v]] & RF_FIXED);
        if (!same) NRS_TRY(nd_engine_setup(c, e, e->nd));
        e->d.sk_pcg = (e->d.sk_n > 0 && !e->nd->on) ? 1 : 0;
        if (e->d.sk_pcg) e->d.ecd = 0;
    }
    return NRS_OK;
}

int engine_reset(nrs_ctx* c, Engine* e) {
    Dev& d = e->d;
    e->cur = 0;
    e->pred_iters = 0; e->pred_peek = 0; e->first_trial_accepted = false;   // batch-size predictors start fresh, as in a new engine
    NRS_HIP(c, hipMemcpyAsync(d.pose[0], d.pose_init, sizeof(Pose) * d.K, hipMemcpyDeviceToDevice, c->stream));
    const size_t o = 3 * (size_t)d.row_lo, n = 3 * (size_t)(d.row_hi - d.row_lo);                  // (the rows this engine holds: all of them on one GPU)
    NRS_HIP(c, hipMemcpyAsync(d.xl[0] + o, d.xl_init + o, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    NRS_HIP(c, hipMemcpyAsync(d.xl[1] + o, d.xl_init + o, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    return NRS_OK;
}

}  // namespace nrs

namespace nrs {
int engine_skin_set_active(nrs_ctx* c, Engine* e, const uint8_t* active) {
    if (e->d.sk_n <= 0) return NRS_OK;
    std::vector<uint8_t> act((size_t)e->d.sk_n, 0);                // (slots are pose-grouped and padded: padding stays inactive)
    for (size_t i = 0; i < e->sk_slot.size(); ++i) act[e->sk_slot[i]] = active[i];
    NRS_HIP(c, hipMemcpyAsync(const_cast<uint8_t*>(e->d.sk_active), act.data(), act.size(), hipMemcpyHostToDevice, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    return NRS_OK;
}
// embedded BA window: the skinned points at the current estimate, X0 + sum_k om_k (x_{n_k} - x_start_{n_k}), summed over k in order
// (caller order of the observations; host arithmetic on the downloaded rows -- the same expression k_skin evaluates)
int engine_skin_positions(nrs_ctx* c, Engine* e, double* xyz) {
    const Dev& d = e->d;
    if (!d.sk_pcg) return c->fail(NRS_ERR_STATE, "no skinned observations on this window");
    const size_t nr = 3 * (size_t)d.n_rows, n = e->sk_slot.size();
    std::vector<double> cur(nr), ini(nr);
    NRS_HIP(c, hipMemcpyAsync(cur.data(), d.xl[e->cur], 8 * nr, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipMemcpyAsync(ini.data(), d.xl_init, 8 * nr, hipMemcpyDeviceToHost, c->stream));
    NRS_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) {
        double x[3] = {e->sk_X0[3 * i], e->sk_X0[3 * i + 1], e->sk_X0[3 * i + 2]};
        for (int k = 0; k < SK_MAX; ++k) {
            const int v = e->sk_vert[SK_MAX * i + k];
            if (v < 0) continue;
            const size_t r = 3 * (size_t)e->vrow[v];
            const double om = e->sk_om[SK_MAX * i + k];
            for (int a = 0; a < 3; ++a) x[a] += om * (cur[r + a] - ini[r + a]);
        }
        xyz[3 * i] = x[0]; xyz[3 * i + 1] = x[1]; xyz[3 * i + 2] = x[2];
    }
    return NRS_OK;
}
int engine_skin_chi2(nrs_ctx* c, Engine* e, double* chi) {
    const Dev& d = e->d;
    if (d.sk_n <= 0) return NRS_OK;
    hipLaunchKernelGGL((k_skin<false>), dim3(d.sk_nblk), dim3(BLK), 0, c->stream, d, d.pose[e->cur], d.xl[e->cur]);
    NRS_HIP(c, hipGetLastError());
    {                                                              // (slots are pose-grouped and padded: back to the caller's order)
        std::vector<double> h((size_t)d.sk_n);
        NRS_HIP(c, hipMemcpyAsync(h.data(), d.sk_chi, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
        NRS_HIP(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < e->sk_slot.size(); ++i) chi[i] = h[e->sk_slot[i]];
    }
    return NRS_OK;
}
}  // namespace nrs
