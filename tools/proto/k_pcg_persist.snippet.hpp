// Prototype, not part of the build (round 3): the fused PCG iteration in a persistent loop with a counter hand-off between the
// workgroups of one XCD (pcg_fused_body = the body of k_pcg_fused as a device function returning 'solve finished').
// Measured on 543 / 1013-point frames: identical iteration counts and parity tests, 22.9 / 27.9 us per iteration against
// 18.2 / 21.7 us with one launch per iteration -- the drain of the stores, the agent-scope atomic and the spin cost more than
// the launch boundary they replace.
// Frames of <= 32 tiles (one workgroup per CU of one XCD): a whole batch of iterations in ONE launch.  Between two iterations
// the workgroups hand their rows over through the XCD's L2: every workgroup waits for its own stores (vmcnt(0)), counts
// itself in, spins on the counter, and drops its L1 / scalar cache before it reads what the others wrote.  All workgroups must
// be resident at once; a workgroup that waits longer than ~20 ms raises flags[6] and leaves (the host then repeats the trial
// with one launch per iteration and keeps to that).
template <int T, bool CO>
__global__ __launch_bounds__(BLK) void k_pcg_persist(Dev P, double lam, int it0, int count, double tol2, double peek_tol2, int pub_seq, unsigned* bar) {
    __shared__ double lds[4 * 9];
    __shared__ double s_up[6];
    __shared__ int s_abort;
    extern __shared__ double dyn[];
    if (blockIdx.x & 7) return;                                    // (one workgroup in eight works: all of them on one XCD)
    const int b = (int)(blockIdx.x >> 3);
    if (b >= P.n_regblk) return;
    const int tid = threadIdx.x;
    const bool lead = b == 0;
    bool fin = false, published = false;
    unsigned target = 0;
    for (int it = it0; it < it0 + count; ++it) {
        const bool last = it + 1 == it0 + count;
        fin = pcg_fused_body<T, CO>(P, b, lead, lam, it, tol2, peek_tol2, last ? pub_seq : 0, lds, s_up, dyn);
        published = last;
        if (fin || last) break;
        // ---- hand-off
        __builtin_amdgcn_s_waitcnt(0);                             // (vmcnt / lgkmcnt 0: this wave's stores have reached the L2)
        __syncthreads();
        target += (unsigned)P.n_regblk;
        if (tid == 0) {
            // (relaxed: the stores are in the L2 already, and an agent-scope release would write the whole L2 back)
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            int ab = 0, spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if ((++spins & 255) == 0 && wall_clock64() - t0 > 2000000) { ab = 1; break; }   // 20 ms of a 100 MHz clock
            }
            s_abort = ab;
        }
        __syncthreads();
        if (s_abort) {
            if (tid == 0) P.flags[6] = 1;
            fin = true; published = false;
            break;
        }
        asm volatile("buffer_inv sc1" ::: "memory");               // L1 dropped: the next loads see what the other workgroups left in the L2
        __builtin_amdgcn_s_dcache_inv();
    }
    if (!published && pub_seq != 0 && lead && tid == 0) { __threadfence(); publish_flags(P, pub_seq); }
}

