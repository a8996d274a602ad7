#pragma once
// rt_fused.hip -- the whole path in ONE launch: march and frequency pass of a run as two phases of the same
// persistent waves (what the reference's own GPU back-end does in one thread per ray,
// src/RayTraceImageCuda.cu:66-127, in behaviour only).
//
// Why: the two-kernel run ends twice.  The march ends when its slowest lane has finished its last ray -- one ray
// of 433 loop iterations takes 0.4 ms whatever else the launch holds, and for the last tenth of the launch most
// waves have nothing left to fetch (tools/wave_times.py: 9 % mean idle on the 6.4 M-ray stand-in, 37 % on its
// 8-rank shard) -- and only then may the frequency kernel start, which has a ragged end of its own.  Here a wave
// whose rays have run out turns to the frequency pass AT ONCE, on tiles the waves of its own work-group have
// finished marching, while the waves that hold the long rays are still marching: the march's tail is filled with
// frequency work, and one launch boundary goes.
//
// How:
//   * rays are reserved in chunks of whole 64-ray tiles; the wave that reserved a chunk marches all of it, so every
//     record of a tile is written by ONE wave.  Every tile in flight has a counter of rays not yet retired in LDS;
//     the lane that takes it to zero has finished the tile: the wave waits for its stores (s_waitcnt vmcnt(0)) and
//     pushes the tile onto the work-group's list (rt_march.hip: TileList);
//   * the consumer is a wave of the SAME work-group, i.e. of the same CU: it shares the producer's vector L1 and L2,
//     so the records need no agent-scope release / acquire (MI355X_MICROARCH.md: what costs microseconds is a
//     hand-off between CUs) -- and the balance between work-groups is the march's: every work-group fetches chunks
//     from the same counters until they run dry, so each ends up with its share of the tiles;
//   * LDS: the march tables (~100 KB for the shipped grids) stay where rt_march_kernel has them; the frequency pass
//     adds its exponent tables, the I_ang histogram and one transposition buffer per wave.  For the shipped sizes
//     sixteen of those do not fit beside the tables: the first `n_free` waves that turn to the frequency pass take
//     the buffers beside the tables, the last ones take buffers that overlay the tables and wait for the last marching
//     wave of the work-group before they touch them (they are the last to run dry anyway).
// The frequency pass itself is freq_tile of rt_freq.hip, unchanged; the stand-alone frequency kernel stays for
// everything this kernel does not take (rt_launch.hip: seeded mode, the exclusive mode, probes, the path tracer,
// the checking repeat of a failing run, tables that do not fit LDS).
#include "rt_freq.hip"

namespace rt {

// LDS layout of the fused kernel, byte offsets from the start of dynamic LDS (set by the host, rt_launch.hip):
//   [0, blob_bytes)            march tables; after the last marching wave: transposition buffers n_free .. n_waves-1
//   off_exp                    [2][EXP_TAB] doubles
//   off_iang                   I_ang histogram, n_ang doubles rounded up to even
//   off_ctl                    4 words: list head | marching waves | next buffer | list nodes handed out
//   off_rem                    [n_waves][32] words: rays in flight of every wave's open tiles (rt_march.hip, march_wave)
//   off_nodes                  [node_cap][2] words: the nodes of the tile list (rt_march.hip, TileList)
//   off_buf                    transposition buffers 0 .. n_free-1, per_wave doubles each
struct FusedLay {
    unsigned off_exp, off_iang, off_ctl, off_rem, off_nodes, off_buf;
    unsigned node_cap;
    unsigned n_free, per_wave; // per_wave in doubles
    unsigned split, k_part;    // TileList::split, TileList::k_part
    // the last n_consumers waves of a work-group never march: they run the frequency pass of finished tiles from the
    // start of the launch, so that the frequency work overlaps the march instead of piling up behind it (0: none)
    unsigned n_consumers, consumers_first;
};
struct FusedKArg {
    DevParams P;
    FreqKArg F;
    unsigned *tile_next; // [4 n_tiles] links of the work-group tile lists
    FusedLay lay;
};

// MAXQ: pixel runs per tile of the few-runs deposit (2 when a pixel has at least 64 rays: a transposition buffer is
// then 3.1 KB instead of 3.6 and fifteen of sixteen fit beside the tables of the shipped grids, so that only the last
// wave of a work-group to leave the march -- which waits for nobody -- takes an overlaid one)
constexpr int fused_wave_doubles(int maxq) { return 4 * XP_ROW + maxq * WAVE; }
// EMIS = false: the gain-only (seeded) frequency pass.  Its tiles span half a dozen pixels, so a wave that runs it needs a
// row cache behind its transposition rows (lay.per_wave says how much, H.nslot how many rows): only a handful of such
// buffers fit beside the march tables -- they go to the consumers, which run the frequency pass during the march; a wave
// that leaves the march takes a buffer over the tables once the last marcher is done.
template <bool BOUNDED, int SF, int MAXQ, bool EMIS = true>
__global__ void __launch_bounds__(1024) rt_fused_kernel(const FusedKArg A)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
#ifdef RT_WAVETIMES
    if (threadIdx.x == 0)
        atomicMin(&g_wt[2], __builtin_amdgcn_s_memrealtime()); // the first work-group to start: t = 0 of the launch
#endif
    const FreqHot &H   = A.F.hot;
    const int n_ang    = H.n_ang;
    double *exp2_tab   = reinterpret_cast<double *>(lds_raw + A.lay.off_exp);
    double *lds_iang   = reinterpret_cast<double *>(lds_raw + A.lay.off_iang);
    unsigned *ctl      = reinterpret_cast<unsigned *>(lds_raw + A.lay.off_ctl);
    double *buf_free   = reinterpret_cast<double *>(lds_raw + A.lay.off_buf);
    const unsigned n_waves  = blockDim.x >> 6;
    const unsigned n_march  = n_waves - A.lay.n_consumers; // waves 0 .. n_march-1 march, the others only consume
    // (which waves: the instruction arbiter of a SIMD serves its OLDEST waves first -- the lowest wave numbers of the
    // work-group -- whatever s_setprio says; consumers_first makes the consumers those waves)
    // (the wave's number through readfirstlane: wave-uniform for the compiler too, not only in fact)
    const unsigned wave_id  = (unsigned) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const bool consumer     = A.lay.consumers_first ? wave_id < A.lay.n_consumers : wave_id >= n_march;
    // (everything below lies behind the march tables: the copy of the tables at the head of march_wave ends in the
    // barrier that also publishes these)
    for (unsigned c = threadIdx.x; c < A.lay.n_free * A.lay.per_wave; c += blockDim.x)
        buf_free[c] = 0.0;
    for (int c = (int) threadIdx.x; c < EXP_TAB; c += (int) blockDim.x) {
        const double e        = exp2((double) c * (1.0 / EXP_TAB));
        exp2_tab[c]           = e;
        exp2_tab[EXP_TAB + c] = __hiloint2double(__double2hiint(e) - (c << 12), __double2loint(e));
    }
    for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x)
        lds_iang[c] = 0.0;
    // (the tile counters of every wave start at zero = free, rt_march.hip)
    for (unsigned c = threadIdx.x; c < n_waves * 32u; c += blockDim.x)
        reinterpret_cast<unsigned *>(lds_raw + A.lay.off_rem)[c] = 0u;
    if (threadIdx.x == 0) {
        ctl[0] = TILE_NONE;
        ctl[1] = n_march;
        ctl[2] = 0u;
        ctl[3] = 0u;
    }
    const TileList list{ &ctl[0], A.tile_next, reinterpret_cast<unsigned *>(lds_raw + A.lay.off_nodes), &ctl[3], A.lay.node_cap,
                         reinterpret_cast<unsigned *>(lds_raw + A.lay.off_rem) + wave_id * 32u,
                         &ctl[1], n_march, A.lay.split, A.lay.k_part };

    // ---- phase 1: the march (rt_march.hip), one tile per chunk, finished tiles pushed onto the list ----
    // (marching waves given a higher wave priority than the waves of their SIMD that have turned to the frequency pass
    // -- so that the stragglers with the long rays finish sooner: measured, s_setprio 1 and 3, nothing: 2.66 / 2.65
    // against 2.65 ms, 8-rank shard 0.537 / 0.536 against 0.534; profiles/r04_prio_ab.txt)
    march_load_tables<true>(A.P, lds_raw);
    if (!consumer)
        march_wave<true, BOUNDED, true, EMIS ? 1 : 0>(A.P, lds_raw, list); // (emission runs are backward runs: rt_launch.hip asks for it)

    // ---- phase 2: this wave's rays have run out; frequency pass on the work-group's finished tiles ----
    const int lane = lane_id();
#ifdef RT_WAVETIMES // diagnostic build: {left the march, has a buffer, first tile done, end, where, tiles} per wave in g_ft
    const unsigned long long fu_left = __builtin_amdgcn_s_memrealtime();
    unsigned long long fu_first = 0, fu_tiles = 0;
#endif
    unsigned slot  = 0;
    if (lane == 0) {
        if (!consumer)
            __hip_atomic_fetch_add(&ctl[1], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // one marching wave less
        slot = __hip_atomic_fetch_add(&ctl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    slot = (unsigned) __builtin_amdgcn_readfirstlane((int) slot);
    auto marching = [&]() { return __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    double *mine;
    if (slot < A.lay.n_free) {
        mine = buf_free + (size_t) slot * A.lay.per_wave;
    } else {
        // a buffer over the march tables: not before the last wave of the work-group has stopped reading them
        while (marching() != 0u)
            __builtin_amdgcn_s_sleep(32);
        mine = reinterpret_cast<double *>(lds_raw) + (size_t) (slot - A.lay.n_free) * A.lay.per_wave;
        for (unsigned c = (unsigned) lane; c < A.lay.per_wave; c += WAVE)
            mine[c] = 0.0;
        __builtin_amdgcn_wave_barrier();
    }
#ifdef RT_ABL_FUSED_MARCH_ONLY // profiling only: what the march phase costs inside this kernel (no frequency phase)
    return;
#endif
    double *xpose = mine;
    double *cache = mine + fused_wave_doubles(MAXQ); // (row cache of the gain-only instance; the emission instance has none: nslot = 0)
#ifdef RT_WAVETIMES
    const unsigned long long fu_buf = __builtin_amdgcn_s_memrealtime();
#endif
    for (;;) {
        unsigned tile = TILE_NONE;
        if (lane == 0)
            tile = tile_pop(list);
        tile = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
        if (tile == TILE_NONE) {
            // nothing finished right now.  Tiles are pushed by marching waves only: once none is left the list can
            // only shrink, and an empty list then is the end (the pushes of a wave precede its leaving the march).
            if (marching() == 0u) {
                if (lane == 0)
                    tile = tile_pop(list);
                tile = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
                if (tile == TILE_NONE)
                    break;
            } else {
                __builtin_amdgcn_s_sleep(64);
                continue;
            }
        }
        // (as in rt_freq_kernel: the cold half of the argument block is addressed inside the kernarg segment and made
        // opaque per tile, likewise the flag word and the lane number)
        ColdPtr C = (ColdPtr) ((const RT_CONST_AS char *) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(FusedKArg, F) +
                               offsetof(FreqKArg, cold));
        asm volatile("" : "+s"(C));
        unsigned hflags = H.flags;
        int lane_t      = lane;
        asm volatile("" : "+s"(hflags), "+v"(lane_t));
        // a whole tile, or one of the four parts of its frequency range (rt_march.hip: tile_publish)
        const unsigned part = (tile >> TILE_PART_SHIFT) & 3u;
        const int k0 = (tile & TILE_PART_FLAG) ? (int) (part * A.lay.k_part) : 0;
        const int k1 = (tile & TILE_PART_FLAG) && part < 3u ? k0 + (int) A.lay.k_part : 0x7fffffff;
        if (k0 < H.K)
            freq_tile<SF, EMIS, MAXQ>(H, hflags, C, lds_iang, exp2_tab, xpose, cache, tile & TILE_ID_MASK, lane_t, k0, k1);
#ifdef RT_WAVETIMES
        if (!fu_first)
            fu_first = __builtin_amdgcn_s_memrealtime();
        fu_tiles++;
#endif
    }
#ifdef RT_WAVETIMES
    if (lane == 0) {
        const unsigned long long fu_end = __builtin_amdgcn_s_memrealtime();
        const unsigned w = atomicAdd(&g_ft_n, 1u);
        if (w < 8192) {
            g_ft[0][w] = fu_left;
            g_ft[1][w] = fu_buf;
            g_ft[2][w] = fu_first ? fu_first : fu_end;
            g_ft[3][w] = fu_end;
            g_ft[4][w] = (unsigned long long) blockIdx.x | ((unsigned long long) (threadIdx.x >> 6) << 16) | ((unsigned long long) slot << 24);
            g_ft[5][w] = fu_tiles;
        }
    }
#endif
    __syncthreads();
    if (!(H.flags & FQ_DBG_NOFLUSH)) {
        for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x) {
            const double v = lds_iang[c];
            if (v != 0.0)
                unsafeAtomicAdd(&H.iang[c], v);
        }
    }
}

} // namespace rt
