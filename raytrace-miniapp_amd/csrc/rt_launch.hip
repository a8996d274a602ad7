// rt_launch.hip -- the kernels of the path and how a run puts them on a queue.
//
// Replaces the launch half of RayTraceImageCudaLoop (src/RayTraceImageCuda.cu:198-203: one thread-per-ray
// launch): a run is the march kernel (persistent lanes over LDS-resident tables, rt_march.hip) -> one
// 96-byte record per ray -> the frequency / deposit kernel (rt_freq.hip), back to back on one queue, or the
// path tracer (rt_path.hip) in place of the frequency kernel.  This is the only translation unit with the
// kernels of the path in it; the rest of the library reaches them through the functions declared in
// rt_runtime.h.
#include "rt_path.hip" // debug path tracer (before rt_freq.hip: no FMA contraction there)
#include "rt_freq.hip" // kernel B (includes rt_march.hip, kernel A)
#include "rt_fused.hip" // both as two phases of one launch

#include "rt_runtime.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

using namespace rtr;

namespace {

// Dynamic LDS above 64 KB has to be allowed per kernel and per device; asked for once per (device, kernel), and
// again only when a launch needs more than the attribute stands at.
int allow_lds(const void *kernel, int device, size_t bytes, size_t limit)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> allowed; // the size the attribute stands at
    if (bytes <= 64 * 1024)
        return RT_OK;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = allowed[{ device, kernel }];
    if (bytes <= have)
        return RT_OK;
    const size_t want = limit > bytes ? limit : bytes;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) want));
    have = want;
    return RT_OK;
}

// the frequency pass's own argument block (rt_device.h): hot = what the frequency loop reads, cold = what the
// per-ray preamble of a tile reads
rt::FreqKArg freq_args(const rt_hip_plan *p, bool iang_in_lds, int nslot, unsigned long long grid_waves)
{
    const rt::DevParams &P = p->P;
    rt::FreqKArg a;
    memset(&a, 0, sizeof(a));
    a.hot.gv0        = p->gv_dev.size() > 1 ? p->gv_dev[1] : nullptr;
    a.hot.gv1        = p->gv_dev.size() > 2 ? p->gv_dev[2] : nullptr;
    a.hot.gain       = P.gain;
    a.hot.rec        = P.rec;
    a.hot.image      = P.image;
    a.hot.iang       = P.iang;
    a.hot.ctl        = P.ctl;
    a.hot.dv2        = p->dv2_dev;
    a.hot.seed_fk    = P.has_seed ? P.seed.f[4] : nullptr;
    a.hot.bad        = P.bad;
    a.hot.scale      = P.scale;
    a.hot.gs_cap     = P.gs_cap;
    a.hot.K          = P.K;
    a.hot.Kp         = P.Kp;
    a.hot.L          = P.L;
    a.hot.method     = P.method;
    a.hot.rec_stride = P.rec_stride;
    a.hot.n_rays     = (unsigned) P.rays.count;
    a.hot.tile_begin = P.tile_begin;
    a.hot.tile_end   = P.tile_end;
    a.hot.freq_id    = P.freq_id;
    {
        unsigned sh = 0;
        while ((1ull << sh) < 2ull * grid_waves) // 2 x waves
            sh++;
        a.hot.fetch_shift = sh;
    }
    a.hot.nslot      = nslot;
    a.hot.nx         = P.beam.nx;
    a.hot.ny         = P.beam.ny;
    a.hot.n_ang      = P.beam.na * P.beam.nb;
    a.hot.flags      = (P.exclusive ? rt::FQ_EXCLUSIVE : 0u) | (P.safe == 1 ? rt::FQ_SAFE_CHECK : 0u) |
                  (P.safe == 2 ? rt::FQ_SAFE_SKIP : 0u) | (P.exact_emis ? rt::FQ_EXACT_EMIS : 0u) |
                  (P.has_seed ? rt::FQ_HAS_SEED : 0u) | (P.probe_on ? rt::FQ_PROBE : 0u) |
                  (p->gv_has_nan ? rt::FQ_GV_NAN : 0u) | (iang_in_lds ? rt::FQ_IANG_LDS : 0u) |
                  ((P.method != 1 || P.has_seed || P.probe_on) ? rt::FQ_NEED_EXIT : 0u) |
                  (P.own_cells ? rt::FQ_OWN_CELLS : 0u) | ((P.debug & 4u) ? rt::FQ_DBG_NOFLUSH : 0u);
    a.cold.beam  = P.beam;
    a.cold.seed  = P.seed;
    a.cold.rays  = P.rays;
    a.cold.probe = P.probe;
    return a;
}

// frequency kernel variants: SF = compile-time number of sub-segments (6 <=> N = 3,
// the shipped inputs; 0 = any N)
template <int SF, bool EMIS, bool EXCL> int launch_freq(rt_hip_plan *p, hipStream_t stream, unsigned cap_blocks)
{
    const size_t ang_bytes = p->n_iang * sizeof(double);
    // (one global atomic per ray on na*nb addresses serialises badly: the histogram stays in LDS)
    const int in_lds       = ang_bytes <= 32 * 1024;
    // Work-groups of FREQ_WG_WAVES waves; the register budget allows `waves` per SIMD, i.e. wg_per_cu work-groups.
    // Per-wave row cache for tiles with several pixel runs (seeded): up to 16 rows of Kp doubles, as many as fit
    // into the work-group's share of the 160 KB beside the exponent tables, the I_ang histogram and the per-wave
    // transposition rows (rt_freq.hip: freq_lds_doubles); fewer than 4 rows is not worth having.
    const int waves       = EMIS ? RT_FREQ_WAVES : RT_FREQ_WAVES_SEED;
    constexpr bool excl   = EXCL; // (= p->P.exclusive: launch_freq_any picks the instance)
    // (exclusive mode: while its flush wrote 8 bytes per lane with a pixel look-up per store, 12 waves per CU ran 3.6 %
    // faster than 16; with the regular-tile flush of 16-byte stores 16 waves win -- 22.15 against 23.0 ms on the
    // 4096^2 x 512 image, tools/config5_waves.py)
    int wg_waves          = (int) env_unsigned("RT_HIP_FREQ_WG_WAVES", (unsigned) rt::FREQ_WG_WAVES, 1, (unsigned) rt::FREQ_WG_WAVES);
    int wg_per_cu         = waves * 4 / wg_waves;
    wg_per_cu             = wg_per_cu < 1 ? 1 : wg_per_cu;
    auto lds_of           = [&](int rows) { return rt::freq_lds_doubles(in_lds != 0, (int) p->n_iang, excl, rows, p->P.Kp, wg_waves) * sizeof(double); };
    auto rows_that_fit    = [&](size_t budget) {
        int rows = 0;
        while (rows < 16 && lds_of(rows + 1) + 1024 <= budget)
            rows++;
        return rows;
    };
    int nslot = 0;
    if (!excl) { // (exclusive mode: no reduction at all; the space holds the store staging rows instead)
        nslot = rows_that_fit(p->lds_limit / (size_t) wg_per_cu);
        const int min_rows = (int) env_unsigned("RT_HIP_FREQ_MIN_ROWS", 7, 0, 16);
        if (!EMIS && nslot < min_rows && wg_per_cu > 1) { // seeded tiles hold ~7 pixels: rather one work-group less per CU than no row for them
            wg_per_cu--;
            nslot = rows_that_fit(p->lds_limit / (size_t) wg_per_cu);
        }
        nslot = nslot < 4 ? 0 : nslot;
    }
    const size_t lds = lds_of(nslot);
    // persistent grid: as many work-groups per CU as LDS (160 KB) and the wave slots allow; the
    // occupancy API under-reports large-LDS kernels, and an over-sized grid is harmless here
    // (surplus work-groups find the tile counter exhausted and leave)
    int per_cu = (int) (p->lds_limit / (lds + 512));
    per_cu     = per_cu > wg_per_cu ? wg_per_cu : (per_cu < 1 ? 1 : per_cu);
    per_cu = (int) env_unsigned("RT_HIP_FREQ_WGS", (unsigned) per_cu, 1, 16); // tuning override
    unsigned long long want = ((unsigned long long) (p->P.tile_end - p->P.tile_begin) + (unsigned) wg_waves - 1) / (unsigned) wg_waves;
    unsigned long long cap  = (unsigned long long) p->cu_count * (unsigned) per_cu;
    if (cap_blocks && cap > cap_blocks)
        cap = cap_blocks;
    const unsigned grid = (unsigned) (want < cap ? want : cap);
    if (grid > 0) {
        const rt::FreqKArg a = freq_args(p, in_lds != 0, nslot, (unsigned long long) grid * (unsigned) wg_waves);
        if (lds > p->lds_limit) {
            char msg[256];
            snprintf(msg, sizeof(msg), "frequency kernel: %zu bytes of LDS per work-group (I_ang histogram of %zu cells, %d waves) "
                     "exceed the device's %zu", lds, p->n_iang, wg_waves, p->lds_limit);
            return fail_arg(msg);
        }
        {
            const int rc = allow_lds(reinterpret_cast<const void *>(&rt::rt_freq_kernel<SF, EMIS, EXCL>), p->device, lds, p->lds_limit);
            if (rc != RT_OK)
                return rc;
        }
        hipLaunchKernelGGL((rt::rt_freq_kernel<SF, EMIS, EXCL>), dim3(grid), dim3((unsigned) wg_waves * 64), lds, stream, a);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

int launch_freq_any(rt_hip_plan *p, hipStream_t stream, unsigned tile_begin = 0, unsigned tile_end = ~0u,
                           unsigned freq_id = 0)
{
    p->P.tile_begin = tile_begin;
    p->P.tile_end   = tile_end < p->P.n_tiles ? tile_end : p->P.n_tiles;
    p->P.freq_id    = freq_id;
    const int S = p->P.L * RT_N_SUB;
    if (p->P.use_emis && p->P.exclusive)
        return (S == 6) ? launch_freq<6, true, true>(p, stream, 0) : launch_freq<0, true, true>(p, stream, 0);
    if (p->P.use_emis)
        return (S == 6) ? launch_freq<6, true, false>(p, stream, 0) : launch_freq<0, true, false>(p, stream, 0);
    // (rt_hip_plan_set_ray_grid grants the exclusive mode with emission only)
    return (S == 6) ? launch_freq<6, false, false>(p, stream, 0) : launch_freq<0, false, false>(p, stream, 0);
}

} // namespace

namespace rt {
// the outputs and the control block of a run zeroed by ONE launch (three hipMemsetAsync are three launches with a
// dependency gap behind each: ~1.5 us apiece, MI355X_MICROARCH.md "boundary" -- a percent of an 8-rank step)
extern "C" __global__ void __launch_bounds__(256) rt_zero_kernel(unsigned long long *a, unsigned long long na, unsigned long long *b,
                                                                unsigned long long nb, unsigned long long *c, unsigned long long nc)
{
    const unsigned long long step = (unsigned long long) gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < na + nb + nc; i += step) {
        unsigned long long *q = i < na ? a + i : (i < na + nb ? b + (i - na) : c + (i - na - nb));
        *q                    = 0ull;
    }
}
} // namespace rt

namespace rtr {

int launch_zero3(hipStream_t stream, void *a, size_t a_bytes, void *b, size_t b_bytes, void *c, size_t c_bytes)
{
    // (8-byte words: the image and I_ang are doubles, DevCtl is 8-byte aligned and sized)
    const unsigned long long na = a ? a_bytes / 8 : 0, nb = b ? b_bytes / 8 : 0, nc = c ? c_bytes / 8 : 0;
    if (na + nb + nc == 0)
        return RT_OK;
    unsigned long long blocks = (na + nb + nc + 255) / 256;
    blocks                    = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(rt::rt_zero_kernel, dim3((unsigned) blocks), dim3(256), 0, stream, (unsigned long long *) a, na,
                       (unsigned long long *) b, nb, (unsigned long long *) c, nc);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int launch_tan(const rt_ray *rays_dev, unsigned long long n, float *sxy_dev, hipStream_t stream)
{
    if (n == 0)
        return RT_OK;
    hipLaunchKernelGGL(rt::rt_tan_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, rays_dev, n, sxy_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int launch_seed_tab(const rt::DevSeed &sd, const rt::DevRays &R, size_t n_points, double *sf, unsigned char *sin)
{
    hipLaunchKernelGGL(rt::rt_seed_tab_kernel, dim3((unsigned) ((n_points + 255) / 256)), dim3(256), 0, nullptr, sd, R, sf, sin);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int launch_selftest(unsigned long long *counts_dev)
{
    hipLaunchKernelGGL(rt::rt_selftest_kernel, dim3(1024), dim3(256), 0, nullptr, counts_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// A run whose frequency pass reported failing rays (error -2 / -3) has deposited them: repeat the pass
// over the same march records, first integrating without depositing to mark the failing rays, then
// depositing all others (DevParams::safe).  Leaves image / I_ang as the CPU loop leaves them
// (RayTraceImageCPU.cpp:29-36) and the failure report as the first pass of the repeat gives it.
int plan_repeat_checked(rt_hip_plan *p)
{
    hipStream_t stream = p->last_stream;
    if (p->bad_rays < (size_t) p->n_rays || !p->bad_dev) {
        (void) hipFree(p->bad_dev);
        p->bad_dev = nullptr;
        HIP_TRY(dev_malloc((void **) &p->bad_dev, (size_t) p->n_rays + 16));
        p->bad_rays = (size_t) p->n_rays;
    }
    HIP_TRY(hipMemsetAsync(p->bad_dev, 0, (size_t) p->n_rays, stream));
    HIP_TRY(hipMemsetAsync(&p->ctl->failure_code, 0, sizeof(unsigned), stream));
    HIP_TRY(hipMemsetAsync(&p->ctl->n_failed, 0, sizeof(unsigned), stream));
    HIP_TRY(hipMemsetAsync(p->ctl->next_tile_f, 0, sizeof(p->ctl->next_tile_f), stream));
    p->P.bad  = p->bad_dev;
    p->P.safe = 1;
    int rc    = launch_freq_any(p, stream);
    if (rc == RT_OK) {
        if (!p->P.exclusive)
            HIP_TRY(hipMemsetAsync(p->last_image, 0, p->n_image * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(p->last_iang, 0, p->n_iang * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(p->ctl->next_tile_f, 0, sizeof(p->ctl->next_tile_f), stream));
        p->P.safe = 2;
        rc        = launch_freq_any(p, stream);
    }
    p->P.safe = 0;
    p->P.bad  = nullptr;
    if (rc != RT_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    return RT_OK;
}

// The chunks at the end of the ray list that only the oldest wave of every SIMD takes (rt_march.hip, "The end of a
// launch"): about as many rays as those waves march in one drain period, RT_HIP_LATE_X10 tenths of a ray per lane of
// theirs (0: no such zone), at most RT_HIP_LATE_CAP per cent of the launch (20: swept 8 ... 35 on the 8- and 16-rank shards and
// ASE_small.dat, profiles/r05_fused_end.txt).
static void late_zone(rt_hip_plan *p, unsigned grid, unsigned waves_per_wg, unsigned first_marching_wave, const char *env = "RT_HIP_LATE_X10",
                      unsigned def_x10 = 32)
{
    // (the waves that take the late chunks must be waves that march: waves_per_wg counts the marching waves,
    // first_marching_wave is where they start inside the work-group)
    p->P.late_first = first_marching_wave;
    const unsigned x10   = env_unsigned(env, def_x10, 0, 1000);
    const unsigned waves = env_unsigned("RT_HIP_LATE_WAVES", 4, 0, 16);
    p->P.late_waves  = waves < waves_per_wg ? waves : waves_per_wg;
    p->P.late_chunks = 0;
    if (x10 == 0 || p->P.late_waves == 0 || p->P.chunk == 0)
        return;
    unsigned long long rays = (unsigned long long) grid * p->P.late_waves * 64ull * x10 / 10ull;
    const unsigned long long cap = p->n_rays * env_unsigned("RT_HIP_LATE_CAP", 20, 0, 100) / 100ull; // (per cent of the launch)
    rays                    = rays > cap ? cap : rays;
    p->P.late_chunks        = (unsigned) (rays / p->P.chunk);
}

// One run on a queue: the march (persistent lanes) -> one record per ray -> the frequency pass, as ONE launch
// (rt_fused.hip) where that applies, as two kernels otherwise (or the path tracer in place of the frequency kernel).
int plan_launch_run(rt_hip_plan *p, hipStream_t stream)
{
    // the kernels index rays with 32 bits and round the ray count up to whole chunks of at most 4096 rays
    if (p->n_rays > (unsigned long long) MAX_LIST_RAYS)
        return fail_arg("more than 2^32 - 4096 rays in one run");
    const size_t need = rt::rec_bytes(p->n_rays, p->P.rec_stride); // (whole 64-ray tiles: the records are tile-wise)
    if (need > p->rec_bytes || !p->rec) {
        plan_quiesce(p);
        pool_free(p->device, p->rec);
        p->rec = nullptr;
        HIP_TRY(pool_alloc(p->device, (void **) &p->rec, need ? need : 16));
        p->rec_bytes = need;
    }
    p->P.rec = p->rec;
    // march, LDS variant: the whole march blob in LDS, one work-group of up to 1024 threads per CU;
    // global variant (persistent 256-thread work-groups) when the blob does not fit (RT_HIP_MARCH=global forces it)
    const char *force   = getenv("RT_HIP_MARCH");
    const bool lds_tab  = p->P.blob_bytes + 8 * 1024 <= p->lds_limit && !(force && strcmp(force, "global") == 0);
    // A run is one march launch -- or three, when the ray list is still on the host
    // (rt_hip_image_loop): the list crosses PCIe in slices, each with a synchronous copy (the fast
    // pageable path, ~35 GB/s; asynchronous copies of pageable memory reach a third of that), and
    // the march of a slice runs on image_loop's non-blocking queue while the host copies the next
    // one (16 B/ray: 102 MB, ~3 ms for the 6.4 M-ray case; swept: 3 slices 5.8 ms, 1 slice 6.9, 8 slices 7.3).
    unsigned n_launch = (p->host_rays && p->n_rays >= (2ull << 20)) ? 3u : 1u;
    if (p->host_rays)
        n_launch = env_unsigned("RT_HIP_UPLOAD_SLICES", n_launch, 1, 8); // tuning
    n_launch = n_launch < 1 ? 1 : (n_launch > 8 ? 8 : n_launch);
    // ---- the whole path in ONE launch (rt_fused.hip) where it applies: emission mode on the beam's own ray grid
    // with at least 32 rays per pixel (a 64-ray tile then spans at most three pixels: the few-runs deposit, which
    // needs no row cache), tables in LDS, nothing that wants the march records to itself (probe, path tracer,
    // the checking repeat, profiling switches), and room in LDS for the frequency pass beside the tables
    // (The gain-only mode -- a seed, forward method -- on a ray grid can run as one launch as well: its frequency pass needs a
    // row cache per wave, so only the handful of waves whose buffers fit beside the march tables run it during the march
    // (the layout below).  Built and measured in round 5, profiles/r05_seed_fused_ab.txt: seed_small.dat 3.28 against
    // 3.33 ms with four such waves, twice the rays 5.97 against 5.93 ms -- a wash, because what the one launch buys is the
    // idle end of the march, which is a fifth of a 0.5 ms launch and a hundredth of a 6 ms one, and what it costs is five
    // of sixteen waves marching less.  Two kernels stay the rule for this mode; RT_HIP_FUSED_SEED=1 takes the one launch.)
    const bool fused_emis = p->P.use_emis && p->P.method == 1 && p->P.own_cells && p->P.rays.nga * p->P.rays.ngb >= 32;
    const bool fused_gain = !p->P.use_emis && p->P.rays.list == nullptr && env_unsigned("RT_HIP_FUSED_SEED", 0, 0, 1) == 1;
    const bool fused_cand = lds_tab && n_launch == 1 && p->n_rays > 0 && !p->path_on && !p->probe_on && p->P.debug == 0 &&
                            (fused_emis || fused_gain) && !p->P.exclusive && p->P.safe == 0 &&
                            p->n_iang * sizeof(double) <= 32 * 1024 && env_unsigned("RT_HIP_FUSED", 1, 1, 2) == 1;
    unsigned bthr = lds_tab ? 1024u : 256u;
    if (lds_tab && !fused_cand) {
        // Few rays per lane leave the persistent lanes waiting for the longest ray of a short
        // queue: below about three rays per lane, fewer and busier lanes win (ASE_small, 399 000
        // rays on 256 CUs: 0.65 ms with 1024 threads per CU, 0.44 ms with 512; 8 waves per CU is
        // the least that still hides latency).
        const unsigned long long per_cu_rays = p->cu_count ? p->n_rays / (unsigned long long) p->cu_count : 0;
        // (tools/shard_threads.py on pixel-column shards of the stand-in: 3117 rays per CU 0.461 ms with 768 threads,
        // 0.472 with 1024; 4156 per CU: equal; 1558 per CU: 0.376 ms with 512, 0.432 with 1024)
        bthr = per_cu_rays >= 4ull * 1024 ? 1024u : (per_cu_rays >= 2560ull ? 768u : 512u);
    }
    // (the one-launch run always takes sixteen waves per CU: a quarter of them run the frequency pass from the start and
    // the end of the ray list is kept for the oldest wave of every SIMD, see below -- with those two the full
    // work-group wins at every size measured, 399 K rays ... 6.4 M, profiles/r05_fused_end.txt)
    bthr = env_unsigned("RT_HIP_MARCH_THREADS", bthr, 64, lds_tab ? 1024 : 256) / 64 * 64; // occupancy experiments
    const size_t mlds   = lds_tab ? (size_t) p->P.blob_bytes : 0;
    int per_cu          = 0;
    // the integrator's divisions without range bookkeeping where the tables and the step factor allow it
    // (rt_math.h, fdiv_nr; RT_HIP_MARCH_IEEE=1 forces the full IEEE sequences)
    const bool force_ieee = getenv("RT_HIP_MARCH_IEEE") != nullptr;
    const bool bounded = p->tables_bounded && p->P.c_h3 >= 1e-8f && !force_ieee;
    using march_fn = void (*)(const rt::DevParams);
    // (the instance with method and emission switch fixed at compile time for the emission / backward pair, rt_march.hip
    // MODE: the one-launch run -2.1 % with it; profiles/r05_loop_head.txt)
    int mode = (p->P.use_emis && p->P.method == 1 && !p->path_on) ? 1 : 0;
    // (gain-only, forward: only the method at compile time -- MODE 3, -0.7 %; with the emission switch fixed as well, or
    // alone, the same source compiles to a march that is 5 - 9 % SLOWER: RT_HIP_MARCH_MODE = 2 / 4 / 0 to see it)
    if (!p->P.use_emis && p->P.method == 2 && !p->path_on)
        mode = (int) env_unsigned("RT_HIP_MARCH_MODE", 3, 0, 4);
    const march_fn kernel =
        lds_tab ? (bounded ? (mode == 1 ? rt::rt_march_kernel<true, true, 1> : mode == 2 ? rt::rt_march_kernel<true, true, 2> : mode == 3 ? rt::rt_march_kernel<true, true, 3> : mode == 4 ? rt::rt_march_kernel<true, true, 4> : rt::rt_march_kernel<true, true, 0>)
                           : (mode == 1 ? rt::rt_march_kernel<true, false, 1> : rt::rt_march_kernel<true, false, 0>))
                : (bounded ? rt::rt_march_kernel<false, true, 0> : rt::rt_march_kernel<false, false, 0>);
    if (lds_tab) {
        const int rc = allow_lds(reinterpret_cast<const void *>(kernel), p->device, mlds, p->lds_limit);
        if (rc != RT_OK)
            return rc;
        per_cu = 1; // one work-group per CU: the tables take more than half of the LDS... or the work-group all wave slots
        if (2 * mlds + 1024 <= p->lds_limit && bthr <= 512)
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int) bthr, mlds));
    } else {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int) bthr, mlds));
    }
    if (per_cu < 1)
        per_cu = 1;
    unsigned long long want = ((unsigned long long) p->n_rays + bthr - 1) / bthr;
    unsigned long long cap  = (unsigned long long) p->cu_count * (unsigned) per_cu;
    const unsigned grid     = (unsigned) (want < cap ? want : cap);
    // rays reserved per counter fetch: big enough to amortise the atomic, small enough that the last
    // chunks balance (about 8 chunks per wave), within 64 ... 192 -- swept on the stand-in and on its
    // strong-scaling shards (tools/shard_sweep2.py): a whole 64-ray refill per fetch is the least that
    // pays (798 K rays: 0.88 ms at 16, 0.55 at 32, 0.44 at 64, 0.60 at 96), 64 ... 192 is flat at 6.4 M
    // rays (2.02 ms; 2.66 at 32, 2.05 at 256)
    unsigned long long ch = grid ? p->n_rays / ((unsigned long long) grid * (bthr / 64) * 8) : 64;
    ch                    = ch < 64 ? 64 : (ch > 192 ? 192 : ch);
    p->P.chunk            = (unsigned) ((ch + 15) / 16 * 16);
    p->P.chunk = env_unsigned("RT_HIP_MARCH_CHUNK", p->P.chunk, 1, 4096); // tuning
    // lanes that must wait for block [A] of the march before it runs (swept 1 ... 40 on the 6.4 M-ray
    // stand-in: 2.36 ms at 1, flat optimum 2.12 ms at 8 ... 24, 2.63 ms at 40)
    p->P.park    = env_unsigned("RT_HIP_MARCH_PARK", 12, 1, 64);
    p->P.express_age  = env_unsigned("RT_HIP_EXPRESS_AGE", 0, 0, 1u << 20);
    p->P.express_hold = env_unsigned("RT_HIP_EXPRESS_HOLD", 0, 0, 1);
    p->P.express_park = env_unsigned("RT_HIP_EXPRESS_PARK", 0, 0, 64);
    p->P.express_tail = env_unsigned("RT_HIP_EXPRESS_TAIL", 0, 0, 2);
    p->P.path_on = p->path_on ? 1u : 0u;
    p->P.spin_limit = env_unsigned("RT_HIP_MARCH_SPIN_LIMIT", 1u << 24, 1024, 0x7fffffffu); // (tests lower it)
    p->P.no_skip = p->gv_has_nan ? 1u : 0u; // the CPU loop multiplies 0 * gv[row 0] for sub-segments a ray never entered
    if (p->path_on) {
        const size_t n2 = (size_t) p->P.L * RT_N_SUB + 1;
        if (p->path_rays != (size_t) p->n_rays || !p->path_dev) {
            (void) hipFree(p->path_dev);
            (void) hipFree(p->path_err);
            p->path_dev = nullptr;
            p->path_err = nullptr;
            HIP_TRY(dev_malloc((void **) &p->path_dev, (size_t) p->n_rays * n2 * 3 * sizeof(float) + 16));
            HIP_TRY(hipMalloc((void **) &p->path_err, (size_t) p->n_rays * sizeof(int32_t) + 16));
            p->path_rays = (size_t) p->n_rays;
        }
        HIP_TRY(hipMemsetAsync(p->path_dev, 0, (size_t) p->n_rays * n2 * 3 * sizeof(float), stream));
        HIP_TRY(hipMemsetAsync(p->path_err, 0, (size_t) p->n_rays * sizeof(int32_t), stream));
        p->P.path     = p->path_dev;
        p->P.path_err = p->path_err;
    }
    HIP_TRY(hipEventRecord(p->ev0, stream));
    p->last_fused = false;
    p->P.late_chunks = 0;
    p->P.late_waves  = 0;
    p->P.late_first  = 0;
    if (fused_cand && grid > 0) {
        const unsigned nw       = bthr / 64;
        // doubles per wave: transposition rows + window totals of the few-runs deposit for 2 pixel runs per tile (a pixel
        // has at least 64 rays) or 3, no row cache
        const bool emis         = p->P.use_emis != 0;
        const int maxq          = emis && p->P.rays.nga * p->P.rays.ngb >= 64 ? 2 : 3;
        // gain-only: rows of the per-wave row cache (a seeded tile holds ~7 pixels; fewer than 4 rows is not worth having)
        int nslot               = emis ? 0 : (int) env_unsigned("RT_HIP_FUSED_ROWS", 7, 4, 16);
        size_t per_wave         = (size_t) rt::fused_wave_doubles(maxq) + (size_t) nslot * (size_t) rt::freq_row_stride(p->P.Kp);
        rt::FusedLay lay;
        lay.off_exp  = (unsigned) align_up(p->P.blob_bytes, 16);
        lay.off_iang = lay.off_exp + 2u * rt::EXP_TAB * (unsigned) sizeof(double);
        lay.off_ctl  = lay.off_iang + (unsigned) (((p->n_iang + 1) & ~(size_t) 1) * sizeof(double));
        lay.off_rem  = lay.off_ctl + 16u;
        lay.off_nodes = lay.off_rem + nw * 32u * (unsigned) sizeof(unsigned);
        // nodes of the work-group's tile list in LDS: room for twice a work-group's share of the entries (a tile is one
        // entry, a split tile four; a work-group that marches faster owes more), within 64 ... 1024; the surplus of a
        // work-group that pushes more takes the global links
        {
            const unsigned long long wgs = (p->n_rays + bthr - 1) / bthr < (unsigned long long) p->cu_count ? (p->n_rays + bthr - 1) / bthr
                                                                                                          : (unsigned long long) p->cu_count;
            unsigned long long cap = wgs ? 2ull * ((unsigned long long) p->P.n_tiles / wgs + 1) + 32 : 64;
            cap           = cap < 64 ? 64 : (cap > 1024 ? 1024 : cap);
            if (!emis) // (LDS is what the row caches are short of; the consumers keep the list short)
                cap = cap > 256 ? 256 : cap;
            lay.node_cap  = env_unsigned("RT_HIP_FUSED_NODES", (unsigned) cap, 0, 4096);
        }
        lay.off_buf  = (unsigned) align_up(lay.off_nodes + lay.node_cap * 2u * (unsigned) sizeof(unsigned), 16);
        size_t room = p->lds_limit > lay.off_buf ? (p->lds_limit - lay.off_buf) / (per_wave * sizeof(double)) : 0;
        if (!emis && room < 5 && nslot > 5) { // one more buffer beside the tables is worth two rows of each cache
            nslot    = 5;
            per_wave = (size_t) rt::fused_wave_doubles(maxq) + (size_t) nslot * (size_t) rt::freq_row_stride(p->P.Kp);
            room     = p->lds_limit > lay.off_buf ? (p->lds_limit - lay.off_buf) / (per_wave * sizeof(double)) : 0;
        }
        lay.per_wave = (unsigned) per_wave;
        lay.n_free   = (unsigned) (room < nw ? room : nw);
        // emission: at least half the buffers beside the tables (the others overlay them once the march is over);
        // gain-only: at least three, and they are the consumers'
        const bool fits = (emis ? 2 * lay.n_free >= nw : (lay.n_free >= 3 && nw >= 8)) &&
                          (size_t) (nw - lay.n_free) * per_wave * sizeof(double) <= p->P.blob_bytes;
        if (fits) {
            const size_t flds     = (size_t) lay.off_buf + (size_t) lay.n_free * per_wave * sizeof(double);
            const size_t n_tiles  = 4 * (size_t) p->P.n_tiles; // one link per (tile, part)
            // the last tiles of a work-group in four parts of the frequency range (whole groups of 4 frequencies; not
            // worth it below 32 frequencies); RT_HIP_FUSED_SPLIT = 2: never, 3: every tile (tests)
            const unsigned split_env = env_unsigned("RT_HIP_FUSED_SPLIT", 1, 1, 3);
            lay.split  = split_env == 2 ? 0u : (split_env == 3 ? 2u : 1u);
            lay.k_part = p->P.K >= 32 && emis ? (unsigned) (((p->P.K + 3) / 4 + 3) / 4 * 4) : 0u;
            // a quarter of the work-group -- its last, youngest waves, one per SIMD -- never marches (rt_fused.hip);
            // gain-only: the same, all of them with a buffer beside the tables (four measured better than five or six)
            lay.n_consumers = env_unsigned("RT_HIP_FUSED_CONSUMERS", emis ? nw / 4 : (lay.n_free < nw / 4 ? lay.n_free : nw / 4), 0, nw > 1 ? nw - 1 : 0);
            if (lay.n_consumers >= nw)
                lay.n_consumers = nw - 1;
            lay.consumers_first = env_unsigned("RT_HIP_FUSED_CONSUMERS_FIRST", 0, 0, 1);
            if (p->tile_next_n < n_tiles || !p->tile_next) {
                plan_quiesce(p);
                pool_free(p->device, p->tile_next);
                p->tile_next = nullptr;
                HIP_TRY(pool_alloc(p->device, (void **) &p->tile_next, n_tiles * sizeof(unsigned) + 16));
                p->tile_next_n = n_tiles;
            }
            p->P.ray_begin = 0;
            p->P.ray_end   = (unsigned) p->n_rays;
            p->P.launch_id = 0;
            p->P.chunk     = (p->P.chunk + 32) / 64 * 64; // whole tiles per reservation (64 ... 192 rays)
            p->P.chunk     = p->P.chunk < 64 ? 64 : p->P.chunk;
            // (gain-only: the waves that run the frequency pass during the march are the youngest of their SIMD already,
            // nothing starves the last marchers: no late zone)
            if (emis)
                late_zone(p, (unsigned) ((p->n_rays + bthr - 1) / bthr < (unsigned long long) p->cu_count ? (p->n_rays + bthr - 1) / bthr : p->cu_count),
                          nw - lay.n_consumers, lay.consumers_first ? lay.n_consumers : 0u);
            p->P.tile_begin = 0;
            p->P.tile_end   = p->P.n_tiles;
            p->P.freq_id    = 0;
            rt::FusedKArg fa;
            fa.P         = p->P;
            fa.F         = freq_args(p, true, nslot, (unsigned long long) grid * nw);
            fa.tile_next = p->tile_next;
            fa.lay       = lay;
            const int S  = p->P.L * RT_N_SUB;
            using fused_fn = void (*)(const rt::FusedKArg);
            const fused_fn fk =
                !emis     ? (S == 6 ? (bounded ? rt::rt_fused_kernel<true, 6, 3, false> : rt::rt_fused_kernel<false, 6, 3, false>)
                                    : (bounded ? rt::rt_fused_kernel<true, 0, 3, false> : rt::rt_fused_kernel<false, 0, 3, false>))
                : maxq == 2 ? (S == 6 ? (bounded ? rt::rt_fused_kernel<true, 6, 2> : rt::rt_fused_kernel<false, 6, 2>)
                                    : (bounded ? rt::rt_fused_kernel<true, 0, 2> : rt::rt_fused_kernel<false, 0, 2>))
                          : (S == 6 ? (bounded ? rt::rt_fused_kernel<true, 6, 3> : rt::rt_fused_kernel<false, 6, 3>)
                                    : (bounded ? rt::rt_fused_kernel<true, 0, 3> : rt::rt_fused_kernel<false, 0, 3>));
            {
                const int rc = allow_lds(reinterpret_cast<const void *>(fk), p->device, flds, p->lds_limit);
                if (rc != RT_OK)
                    return rc;
            }
            unsigned long long fwant = ((unsigned long long) p->n_rays + bthr - 1) / bthr;
            const unsigned fgrid     = (unsigned) (fwant < (unsigned long long) p->cu_count ? fwant : (unsigned long long) p->cu_count);
            hipLaunchKernelGGL(fk, dim3(fgrid), dim3(bthr), flds, stream, fa);
            HIP_TRY(hipGetLastError());
            p->host_rays = nullptr;
            HIP_TRY(hipEventRecord(p->evm, stream));
            HIP_TRY(hipEventRecord(p->ev1, stream));
            p->last_fused = true;
            return RT_OK;
        }
    }
    for (unsigned c = 0; c < n_launch && grid > 0 && !(p->P.debug & 2u); c++) {
        const unsigned long long b = p->n_rays * c / n_launch, e = p->n_rays * (c + 1) / n_launch;
        if (p->host_rays) {
            HIP_TRY(hipMemcpy(p->rays_dev + b, p->host_rays + b, (size_t) (e - b) * sizeof(rt_ray), hipMemcpyHostToDevice));
            // Helper.h:409-410 for every ray of the slice, at full lane occupancy, before its march
            if (tan_mode(p->device) == 2) { // this host's tanf is not the restated one: its own values
                std::vector<float> h((size_t) (e - b) * 2);
                host_tangents(p->host_rays + b, (size_t) (e - b), h.data());
                HIP_TRY(hipMemcpy(p->tan_dev + 2 * b, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
            } else {
                hipLaunchKernelGGL(rt::rt_tan_kernel, dim3((unsigned) ((e - b + 255) / 256)), dim3(256), 0, stream,
                                   p->rays_dev + b, (unsigned long long) (e - b), p->tan_dev + 2 * b);
                HIP_TRY(hipGetLastError());
            }
        }
        if (n_launch > 1) { // rays reserved per counter fetch, for this slice
            unsigned long long cs = (e - b) / ((unsigned long long) grid * (bthr / 64) * 8);
            cs                    = cs < 64 ? 64 : (cs > 192 ? 192 : cs);
            p->P.chunk            = (unsigned) ((cs + 15) / 16 * 16);
        }
        p->P.ray_begin = (unsigned) b;
        p->P.ray_end   = (unsigned) e;
        p->P.launch_id = c;
        // (the march as a kernel of its own: the end of the list for the oldest wave of every SIMD as well -- its tail is the
        // drain of the last rays, and one wave per SIMD runs it at the pace of a wave that has the SIMD to itself)
        if (n_launch == 1 && lds_tab)
            late_zone(p, grid, bthr / 64, 0u, "RT_HIP_LATE2_X10", 60); // (seed_small -0.8 %, stand-in as two kernels -1.5 %, its 8-rank shard -6 %)
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(bthr), mlds, stream, p->P);
        HIP_TRY(hipGetLastError());
    }
    p->host_rays = nullptr; // consumed: the list is on the device now
    HIP_TRY(hipEventRecord(p->evm, stream));
    if (p->path_on) {
        // the tracer replaces the frequency / deposit kernel: no image is produced
        if (p->n_rays) {
            hipLaunchKernelGGL(rt::rt_path_kernel, dim3((unsigned) ((p->n_rays + 255) / 256)), dim3(256), 0, stream, p->P);
            HIP_TRY(hipGetLastError());
        }
    } else if (!(p->P.debug & 1u)) {
        const int rc = launch_freq_any(p, stream);
        if (rc != RT_OK)
            return rc;
    }
    HIP_TRY(hipEventRecord(p->ev1, stream));
    return RT_OK;
}

} // namespace rtr

extern "C" {

#ifdef RT_WAVETIMES
// diagnostic build only: wave start / dry / end times of the LAST march launch (100 MHz ticks), then reset
int rt_hip_debug_wavetimes(unsigned long long *summary8, unsigned long long *end8192, unsigned long long *dry8192)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(summary8, HIP_SYMBOL(rt::g_wt), 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(end8192, HIP_SYMBOL(rt::g_wt_end), 8192 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(dry8192, HIP_SYMBOL(rt::g_wt_dry), 8192 * sizeof(unsigned long long)));
    unsigned long long init[8] = { ~0ull, 0, ~0ull, 0, ~0ull, 0, 0, 0 };
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_wt), init, sizeof(init)));
    return RT_OK;
}
// the march trace of every wave (rt_march.hip: g_wt_trace), then cleared
int rt_hip_debug_wavetrace(unsigned long long *trace, int samples, unsigned *blocks)
{
    HIP_TRY(hipDeviceSynchronize());
    if (samples != rt::WT_TRACE)
        return fail_arg("rt_hip_debug_wavetrace: samples");
    if (blocks)
        HIP_TRY(hipMemcpyFromSymbol(blocks, HIP_SYMBOL(rt::g_wt_blocks), sizeof(unsigned) * 8192 * rt::WT_TRACE * 6));
    HIP_TRY(hipMemcpyFromSymbol(trace, HIP_SYMBOL(rt::g_wt_trace), sizeof(unsigned long long) * 8192 * rt::WT_TRACE));
    void *sym = nullptr;
    HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(rt::g_wt_trace)));
    HIP_TRY(hipMemset(sym, 0, sizeof(unsigned long long) * 8192 * rt::WT_TRACE));
    return RT_OK;
}
// time in tile_publish per wave (rt_march.hip: g_wt_pub, g_wt_vm), then cleared
int rt_hip_debug_publish(unsigned long long *pub4, unsigned long long *vm)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(pub4, HIP_SYMBOL(rt::g_wt_pub), sizeof(unsigned long long) * 8192 * 4));
    HIP_TRY(hipMemcpyFromSymbol(vm, HIP_SYMBOL(rt::g_wt_vm), sizeof(unsigned long long) * 8192));
    void *sym = nullptr;
    HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(rt::g_wt_pub)));
    HIP_TRY(hipMemset(sym, 0, sizeof(unsigned long long) * 8192 * 4));
    HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(rt::g_wt_vm)));
    HIP_TRY(hipMemset(sym, 0, sizeof(unsigned long long) * 8192));
    return RT_OK;
}
// ... and of the LAST frequency launch: times[6][8192] = {start, tables ready, first tile done, last tile done, where, tiles} per wave
int rt_hip_debug_freqtimes(unsigned long long *times, unsigned *n_waves)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(times, HIP_SYMBOL(rt::g_ft), 6 * 8192 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(n_waves, HIP_SYMBOL(rt::g_ft_n), sizeof(unsigned)));
    const unsigned zero = 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_ft_n), &zero, sizeof(zero)));
    return RT_OK;
}
#endif

#ifdef RT_INSTRUMENT
// diagnostic build only: loop iterations per ray of the last march (rays below 2^23)
int rt_hip_debug_ray_iters(unsigned short *out, unsigned long long n)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_ray_iters), (size_t) n * sizeof(unsigned short)));
    return RT_OK;
}
#endif
#if defined(RT_INSTRUMENT) || defined(RT_TIMEBLOCKS)
// diagnostic builds only: read and clear the loop-occupancy / block-clock counters
int rt_hip_debug_counters(unsigned long long *out8)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(rt::g_inst), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = { 0 };
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_inst), z, sizeof(z)));
    return RT_OK;
}
#endif

} // extern "C"
