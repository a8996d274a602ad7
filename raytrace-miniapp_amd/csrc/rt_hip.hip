// rt_hip.hip -- the C ABI of include/rt_hip.h on top of the HIP kernels.
//
// Replaces the host side of RayTraceImageCudaLoop (src/RayTraceImageCuda.cu:145-221)
// and of the copy_device helpers (src/RayTraceImageCuda.cu:224-329): where those
// issue ~30 cudaMalloc/cudaMemcpy calls per create_image, a plan packs every
// table into ONE arena, uploads it with ONE copy, zeroes outputs + control block
// and launches the march kernel and the frequency kernel back to back on one
// stream.  No data is cached across calls (Readme.txt:43); freed device allocations and one
// queues per device are (the pool and lease_queue below).
#include "rt_path.hip" // debug path tracer (before rt_freq.hip: no FMA contraction there)
#include "rt_freq.hip" // kernel B (includes rt_march.hip, kernel A)

#include <rccl/rccl.h> // types and prototypes only: librccl.so is loaded on first use (rccl_api below)

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail_hip(hipError_t e, const char *what, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed at rt_hip.hip:%d: %s", what, line, hipGetErrorString(e));
    g_last_error = buf;
    return RT_ERR_HIP;
}
int fail_arg(const char *msg)
{
    g_last_error = msg;
    return RT_ERR_ARG;
}

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess)                        \
            return fail_hip(e_, #expr, __LINE__);    \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over a host staging image of the device arena.
struct ArenaBuilder {
    std::vector<unsigned char> host;
    size_t put(const void *src, size_t bytes)
    {
        size_t off = align_up(host.size(), 256);
        host.resize(off + bytes);
        if (bytes && src)
            memcpy(host.data() + off, src, bytes);
        return off;
    }
    size_t reserve(size_t bytes)
    {
        size_t off = align_up(host.size(), 256);
        host.resize(off + bytes);
        return off;
    }
};

// Device-memory pool: create_image is called once per iteration of the application and may
// not keep DATA across calls (Readme.txt:43), but nothing forbids keeping ALLOCATIONS: the
// ray list (16 B/ray), the tangents and the march records (96 B/ray) are hundreds of MB per
// call and hipMalloc/hipFree of them costs milliseconds.  Freed blocks are parked per device
// (at most POOL_MAX_BLOCKS, POOL_MAX_BYTES) and handed out again best-fit.
constexpr size_t POOL_MAX_BLOCKS = 32;
constexpr size_t POOL_MIN_BYTES  = 0; // every block is worth parking: hipMalloc + hipFree cost ~0.1 ms a pair
size_t pool_max_bytes()
{
    static const size_t cap = [] {
        size_t mb = 32768;
        if (const char *e = getenv("RT_HIP_POOL_MAX_MB")) {
            const long long v = atoll(e);
            if (v >= 0)
                mb = (size_t) v;
        }
        return mb << 20;
    }();
    return cap;
}
struct PoolBlock {
    int device;
    void *ptr;
    size_t bytes;
};
std::mutex g_pool_mutex;
std::vector<PoolBlock> g_pool;
std::unordered_map<void *, size_t> g_pool_sizes; // live blocks handed out by pool_alloc

hipError_t pool_alloc(int device, void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0)
        bytes = 16;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        int best = -1;
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].device == device && g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes + POOL_MIN_BYTES &&
                (best < 0 || g_pool[i].bytes < g_pool[(size_t) best].bytes))
                best = (int) i;
        if (best >= 0) {
            *out                 = g_pool[(size_t) best].ptr;
            g_pool_sizes[*out]   = g_pool[(size_t) best].bytes;
            g_pool.erase(g_pool.begin() + best);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) { // out of memory: drop the parked blocks and retry once
        (void) hipGetLastError();
        int cur = device;
        (void) hipGetDevice(&cur);
        std::vector<PoolBlock> drop;
        {
            std::lock_guard<std::mutex> lk(g_pool_mutex);
            drop.swap(g_pool);
        }
        for (auto &b : drop) {
            (void) hipSetDevice(b.device);
            (void) hipFree(b.ptr);
        }
        (void) hipSetDevice(cur);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        g_pool_sizes[*out] = bytes;
    }
    return e;
}

void pool_trim_all()
{
    std::vector<PoolBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        drop.swap(g_pool);
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto &b : drop) {
        (void) hipSetDevice(b.device);
        (void) hipFree(b.ptr);
    }
    if (have)
        (void) hipSetDevice(cur);
}

// hipMalloc for the large one-off allocations (arena, own image, path, probe): out of memory
// while the pool still parks blocks -> give them back and retry once
hipError_t dev_malloc(void **out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        pool_trim_all();
        e = hipMalloc(out, bytes ? bytes : 16);
    }
    return e;
}

// tuning overrides from the environment: a missing, non-numeric or non-positive value keeps the default
unsigned env_unsigned(const char *name, unsigned def, unsigned lo, unsigned hi)
{
    const char *e = getenv(name);
    if (!e)
        return def;
    const long v = atol(e);
    if (v <= 0)
        return def;
    return (unsigned) (v < (long) lo ? lo : (v > (long) hi ? hi : v));
}

void pool_free(int device, void *ptr)
{
    if (!ptr)
        return;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        auto it = g_pool_sizes.find(ptr);
        if (it != g_pool_sizes.end()) {
            bytes = it->second;
            g_pool_sizes.erase(it);
        }
        size_t held = 0;
        for (auto &b : g_pool)
            held += b.bytes;
        if (bytes >= POOL_MIN_BYTES && g_pool.size() < POOL_MAX_BLOCKS && held + bytes <= pool_max_bytes()) {
            g_pool.push_back({ device, ptr, bytes });
            return;
        }
    }
    (void) hipFree(ptr);
}

} // namespace

struct rt_hip_plan {
    int device         = 0;
    int cu_count       = 0;
    rt::DevParams P    = {};
    unsigned char *arena = nullptr;
    size_t arena_bytes = 0;
    rt_ray *rays_dev   = nullptr;
    double *grid_dev   = nullptr; // ray grids when rays are generated
    float *tan_dev     = nullptr; // tangents: grid mode [nga + ngb], list mode [2 n_rays]
    double *seedtab_dev = nullptr; // grid mode with a seed: per-axis seed factors + support flags
    std::vector<double> beam_x, beam_y, beam_a, beam_b; // host copies, to recognise ray grid == beam grid
    unsigned char *rec = nullptr; // per-ray march records (two-kernel path)
    bool path_on       = false;   // path tracer instead of the image (rt_hip_plan_enable_path)
    float *path_dev    = nullptr; // [n_rays][3L+1][3]
    int32_t *path_err  = nullptr; // [n_rays]
    size_t path_rays   = 0;
    size_t rec_bytes   = 0;
    hipEvent_t evm     = nullptr; // between march and frequency kernels
    const rt_ray *host_rays = nullptr; // ray list still on the host, uploaded by the next run (rt_hip_image_loop)
    double *image_own  = nullptr;
    double *iang_own   = nullptr;
    rt::DevCtl *ctl    = nullptr;
    size_t n_image = 0, n_iang = 0;
    unsigned long long n_rays = 0;
    // probe
    bool probe_on        = false;
    unsigned char *probe = nullptr;
    size_t probe_rays    = 0;
    // last run
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t last_stream = nullptr;
    double *last_image = nullptr, *last_iang = nullptr;
    bool ran = false;
    // timing ring (rt_hip_plan_set_timing_ring): event triples of the last runs, so that a caller can time
    // many back-to-back runs without waiting for each; ev0 / evm / ev1 above are the current run's triple
    std::vector<hipEvent_t> ring; // 3 per slot
    unsigned long long runs = 0;
    bool repeated = false; // the checking repeat of the frequency pass has run for the last run
    unsigned char *bad_dev = nullptr; // failing-ray marks of the checking repeat (plan_repeat_checked)
    size_t bad_rays        = 0;
    std::chrono::steady_clock::time_point t_created;
    // frequency kernel arguments that are not part of DevParams (rt_device.h: FreqHot)
    std::vector<const float *> gv_dev; // [N] lineshape table of every length on the device, entry 0 unused
    const double *dv2_dev = nullptr;   // [Kp] 2 * beam.dv
    bool gv_has_nan       = false;     // host scan of the lineshape tables (emission mode)
    // the refractive-index tables and the segment length lie in the ranges under which the march's divisions
    // need no scaling (rt_march.hip, template parameter BOUNDED); checked by rt_hip_plan_create
    bool tables_bounded   = false;
};

// frequency kernel variants: SF = compile-time number of sub-segments (6 <=> N = 3,
// the shipped inputs; 0 = any N)
template <int SF, bool EMIS>
static int launch_freq(rt_hip_plan *p, hipStream_t stream, unsigned cap_blocks)
{
    const size_t ang_bytes = p->n_iang * sizeof(double);
    // (one global atomic per ray on na*nb addresses serialises badly: the histogram stays in LDS)
    const int in_lds       = ang_bytes <= 32 * 1024;
    // Work-groups of FREQ_WG_WAVES waves; the register budget allows `waves` per SIMD, i.e. wg_per_cu work-groups.
    // Per-wave row cache for tiles with several pixel runs (seeded): up to 16 rows of Kp doubles, as many as fit
    // into the work-group's share of the 160 KB beside the exponent tables, the I_ang histogram and the per-wave
    // transposition rows (rt_freq.hip: freq_lds_doubles); fewer than 4 rows is not worth having.
    const int waves       = EMIS ? RT_FREQ_WAVES : RT_FREQ_WAVES_SEED;
    const bool excl       = p->P.exclusive != 0;
    // (exclusive mode is bound by its stores: 12 waves per CU run 3.6 % faster than 16 -- tools/config5_ab.py)
    int wg_waves          = (int) env_unsigned("RT_HIP_FREQ_WG_WAVES", excl ? 12u : (unsigned) rt::FREQ_WG_WAVES, 1, (unsigned) rt::FREQ_WG_WAVES);
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
        nslot = rows_that_fit((size_t) (160 * 1024) / (size_t) wg_per_cu);
        if (!EMIS && nslot < 7 && wg_per_cu > 1) { // seeded tiles hold ~7 pixels: rather one work-group less per CU than no row for them
            wg_per_cu--;
            nslot = rows_that_fit((size_t) (160 * 1024) / (size_t) wg_per_cu);
        }
        nslot = nslot < 4 ? 0 : nslot;
    }
    const size_t lds = lds_of(nslot);
    // persistent grid: as many work-groups per CU as LDS (160 KB) and the wave slots allow; the
    // occupancy API under-reports large-LDS kernels, and an over-sized grid is harmless here
    // (surplus work-groups find the tile counter exhausted and leave)
    int per_cu = (int) ((160 * 1024) / (lds + 512));
    per_cu     = per_cu > wg_per_cu ? wg_per_cu : (per_cu < 1 ? 1 : per_cu);
    per_cu = (int) env_unsigned("RT_HIP_FREQ_WGS", (unsigned) per_cu, 1, 16); // tuning override
    unsigned long long want = ((unsigned long long) (p->P.tile_end - p->P.tile_begin) + (unsigned) wg_waves - 1) / (unsigned) wg_waves;
    unsigned long long cap  = (unsigned long long) p->cu_count * (unsigned) per_cu;
    if (cap_blocks && cap > cap_blocks)
        cap = cap_blocks;
    const unsigned grid = (unsigned) (want < cap ? want : cap);
    if (grid > 0) {
        // the kernel's own argument block (rt_device.h): hot = what the frequency loop reads, cold = what the
        // per-ray preamble of a tile reads
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
            while ((1ull << sh) < 2ull * grid * (unsigned long long) wg_waves) // 2 x waves
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
                      (p->gv_has_nan ? rt::FQ_GV_NAN : 0u) | (in_lds ? rt::FQ_IANG_LDS : 0u) |
                      ((P.method != 1 || P.has_seed || P.probe_on) ? rt::FQ_NEED_EXIT : 0u) |
                      (P.own_cells ? rt::FQ_OWN_CELLS : 0u) | ((P.debug & 4u) ? rt::FQ_DBG_NOFLUSH : 0u);
        a.cold.beam  = P.beam;
        a.cold.seed  = P.seed;
        a.cold.rays  = P.rays;
        a.cold.probe = P.probe;
        // (dynamic LDS above 64 KB has to be allowed per kernel and device; once, the limit is the whole 160 KB)
        static std::atomic<unsigned long long> lds_allowed{ 0 }; // bit d: done on device d
        if (lds > 64 * 1024 && p->device < 64 && !(lds_allowed.load() >> p->device & 1ull)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&rt::rt_freq_kernel<SF, EMIS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            lds_allowed.fetch_or(1ull << p->device);
        }
        hipLaunchKernelGGL((rt::rt_freq_kernel<SF, EMIS>), dim3(grid), dim3((unsigned) wg_waves * 64), lds, stream, a);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

// Non-blocking queues for the host-pointer entry points, kept across calls like the memory pool
// (a resource, not data; creating one costs ~2 ms): synchronous copies on the host thread do not
// wait for them, which is what lets the ray upload overlap the march.  Every call LEASES a queue
// of its device for its own use, so concurrent calls on one device (create_image is thread-safe,
// RayTrace.h:90-91) neither share a stream nor see each other's kernels in their event times.
static std::mutex g_queue_mutex;
static std::vector<hipStream_t> g_queue_free[64];
static hipStream_t lease_queue(int device)
{
    if (device < 0 || device >= 64)
        return nullptr;
    {
        std::lock_guard<std::mutex> lock(g_queue_mutex);
        if (!g_queue_free[device].empty()) {
            hipStream_t q = g_queue_free[device].back();
            g_queue_free[device].pop_back();
            return q;
        }
    }
    hipStream_t q = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess)
        return nullptr;
    return q;
}
static void release_queue(int device, hipStream_t q)
{
    if (!q || device < 0 || device >= 64)
        return;
    std::lock_guard<std::mutex> lock(g_queue_mutex);
    g_queue_free[device].push_back(q);
}

// Wait for the work of the plan's last run before any of its buffers is freed or parked in the pool:
// another plan may be handed a parked block at once (pool_alloc) and overwrite it.
static void plan_quiesce(rt_hip_plan *p)
{
    if (p && p->ran) {
        if (hipStreamSynchronize(p->last_stream) != hipSuccess)
            (void) hipGetLastError(); // a caller's stream that is gone: nothing is in flight on it
    }
}

// ---- list-mode launch tangents and the host's libm ------------------------------------------------
// rt_tan_kernel restates the float tanf of glibc 2.35 (rt_march.hip).  Whether THIS host's tanf is that
// routine is probed once per process: 8192 angles from 1e-3 mrad to 1.37 rad, both signs, device against
// host, bit for bit.  If they differ anywhere (another libm), list-mode tangents are computed by the
// host's tanf on host threads and uploaded -- slower (two libm calls per ray), but the march then starts
// every ray exactly as RayTraceImageCPULoop on this host does.  (Grid mode always uses the host's tanf.)
static unsigned host_threads(unsigned cap); // (defined with the ray-grid recognition below)
static std::atomic<int> g_tan_mode(0); // 0 unknown, 1 device restatement == host tanf, 2 host tangents

static void host_tangents(const rt_ray *rays, size_t n, float *sxy)
{
    const unsigned threads = n >= (1u << 16) ? host_threads(16) : 1;
    auto work = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            sxy[2 * i]     = tanf(1e-3f * rays[i].a); // Helper.h:409-410
            sxy[2 * i + 1] = tanf(1e-3f * rays[i].b);
        }
    };
    if (threads == 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back(work, n * t / threads, n * (t + 1) / threads);
    for (auto &t : th)
        t.join();
}

static int tan_mode(int device)
{
    int m = g_tan_mode.load();
    if (m != 0)
        return m;
    if (getenv("RT_HIP_TAN_ON_HOST")) {
        g_tan_mode.store(2);
        return 2;
    }
    const size_t n = 8192;
    std::vector<rt_ray> r(n);
    for (size_t i = 0; i < n; i++) {
        // geometric ladder of magnitudes with a wobble in the low bits, alternating signs
        const double mag = 1e-3 * pow(1.37e6, (double) (i / 2) / (double) (n / 2 - 1)); // mrad: 1e-3 ... 1370
        const float a    = (float) (mag * (1.0 + 1e-4 * (double) ((i * 2654435761u) & 1023u)));
        r[i]             = { 0.0f, 0.0f, (i & 1) ? -a : a, (i & 1) ? a : -a };
    }
    rt_ray *d_r  = nullptr;
    float *d_sxy = nullptr;
    std::vector<float> got(2 * n), want(2 * n);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess)
        e = hipMalloc((void **) &d_r, n * sizeof(rt_ray));
    if (e == hipSuccess)
        e = hipMalloc((void **) &d_sxy, 2 * n * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(d_r, r.data(), n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt::rt_tan_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, nullptr, d_r,
                           (unsigned long long) n, d_sxy);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(got.data(), d_sxy, 2 * n * sizeof(float), hipMemcpyDeviceToHost);
    (void) hipFree(d_r);
    (void) hipFree(d_sxy);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        return 1; // the probe could not run: the caller's own HIP calls will report what is wrong
    }
    host_tangents(r.data(), n, want.data());
    m = memcmp(got.data(), want.data(), 2 * n * sizeof(float)) == 0 ? 1 : 2;
    g_tan_mode.store(m);
    return m;
}

static int launch_freq_any(rt_hip_plan *p, hipStream_t stream, unsigned tile_begin = 0, unsigned tile_end = ~0u,
                           unsigned freq_id = 0)
{
    p->P.tile_begin = tile_begin;
    p->P.tile_end   = tile_end < p->P.n_tiles ? tile_end : p->P.n_tiles;
    p->P.freq_id    = freq_id;
    const int S = p->P.L * RT_N_SUB;
    if (p->P.use_emis)
        return (S == 6) ? launch_freq<6, true>(p, stream, 0) : launch_freq<0, true>(p, stream, 0);
    return (S == 6) ? launch_freq<6, false>(p, stream, 0) : launch_freq<0, false>(p, stream, 0);
}

// A run whose frequency pass reported failing rays (error -2 / -3) has deposited them: repeat the pass
// over the same march records, first integrating without depositing to mark the failing rays, then
// depositing all others (DevParams::safe).  Leaves image / I_ang as the CPU loop leaves them
// (RayTraceImageCPU.cpp:29-36) and the failure report as the first pass of the repeat gives it.
static int plan_repeat_checked(rt_hip_plan *p)
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

// Two-kernel path: march (persistent lanes) -> records in HBM -> frequency pass.
static int plan_run_split(rt_hip_plan *p, hipStream_t stream)
{
    const size_t need = (size_t) p->n_rays * p->P.rec_stride;
    if (need > p->rec_bytes || !p->rec) {
        plan_quiesce(p);
        pool_free(p->device, p->rec);
        p->rec = nullptr;
        HIP_TRY(pool_alloc(p->device, (void **) &p->rec, need ? need : 16));
        p->rec_bytes = need;
    }
    p->P.rec = p->rec;
    // march: persistent 256-thread work-groups
    // LDS variant: the whole march blob in LDS, one 1024-thread work-group per CU;
    // global variant when the blob does not fit (RT_HIP_MARCH=global forces it)
    const char *force   = getenv("RT_HIP_MARCH");
    const bool lds_tab  = p->P.blob_bytes <= 152 * 1024 && !(force && strcmp(force, "global") == 0);
    unsigned bthr = lds_tab ? 1024u : 256u;
    if (lds_tab) {
        // Few rays per lane leave the persistent lanes waiting for the longest ray of a short
        // queue: below about three rays per lane, fewer and busier lanes win (ASE_small, 399 000
        // rays on 256 CUs: 0.65 ms with 1024 threads per CU, 0.44 ms with 512; 8 waves per CU is
        // the least that still hides latency).
        const unsigned long long per_cu_rays = p->cu_count ? p->n_rays / (unsigned long long) p->cu_count : 0;
        // (tools/shard_threads.py on pixel-column shards of the stand-in: 3117 rays per CU 0.461 ms with 768 threads,
        // 0.472 with 1024; 4156 per CU: equal; 1558 per CU: 0.376 ms with 512, 0.432 with 1024)
        bthr = per_cu_rays >= 4ull * 1024 ? 1024u : (per_cu_rays >= 2560ull ? 768u : 512u);
    }
    bthr = env_unsigned("RT_HIP_MARCH_THREADS", bthr, 64, lds_tab ? 1024 : 256) / 64 * 64; // occupancy experiments
    const size_t mlds   = lds_tab ? (size_t) p->P.blob_bytes : 0;
    int per_cu          = 0;
    // the integrator's divisions without range bookkeeping where the tables and the step factor allow it
    // (rt_math.h, fdiv_nr; RT_HIP_MARCH_IEEE=1 forces the full IEEE sequences)
    const bool force_ieee = getenv("RT_HIP_MARCH_IEEE") != nullptr;
    const bool bounded = p->tables_bounded && p->P.c_h3 >= 1e-8f && !force_ieee;
    using march_fn = void (*)(const rt::DevParams);
    const march_fn kernel = lds_tab ? (bounded ? rt::rt_march_kernel<true, true> : rt::rt_march_kernel<true, false>)
                                    : (bounded ? rt::rt_march_kernel<false, true> : rt::rt_march_kernel<false, false>);
    if (lds_tab)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) mlds));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int) bthr, mlds));
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
    p->P.path_on = p->path_on ? 1u : 0u;
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
    // A run is one march launch -- or three, when the ray list is still on the host
    // (rt_hip_image_loop): the list crosses PCIe in slices, each with a synchronous copy (the fast
    // pageable path, ~35 GB/s; asynchronous copies of pageable memory reach a third of that), and
    // the march of a slice runs on image_loop's non-blocking queue while the host copies the next
    // one (16 B/ray: 102 MB, ~3 ms for the 6.4 M-ray case; swept: 3 slices 5.8 ms, 1 slice 6.9, 8 slices 7.3).
    unsigned n_launch = (p->host_rays && p->n_rays >= (2ull << 20)) ? 3u : 1u;
    if (p->host_rays)
        n_launch = env_unsigned("RT_HIP_UPLOAD_SLICES", n_launch, 1, 8); // tuning
    n_launch = n_launch < 1 ? 1 : (n_launch > 8 ? 8 : n_launch);
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

extern "C" {

int rt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *rt_hip_last_error(void) { return g_last_error.c_str(); }

int rt_hip_selftest(int device, unsigned long long *n_checked, unsigned long long *n_mismatch)
{
    if (!n_checked || !n_mismatch)
        return fail_arg("rt_hip_selftest: NULL argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr, h[2] = { 0, 0 };
    HIP_TRY(hipMalloc((void **) &d, sizeof(h)));
    hipError_t e = hipMemset(d, 0, sizeof(h));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt::rt_selftest_kernel, dim3(1024), dim3(256), 0, nullptr, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void) hipFree(d);
    HIP_TRY(e);
    *n_checked  = h[0];
    *n_mismatch = h[1];
    return RT_OK;
}

void rt_hip_plan_destroy(rt_hip_plan *p)
{
    if (!p)
        return;
    (void) hipSetDevice(p->device);
    plan_quiesce(p); // kernels of an unfetched (or failed) run may still use the buffers parked below
    if (p->ring.empty()) {
        if (p->ev0)
            (void) hipEventDestroy(p->ev0);
        if (p->ev1)
            (void) hipEventDestroy(p->ev1);
        if (p->evm)
            (void) hipEventDestroy(p->evm);
    }
    for (hipEvent_t e : p->ring) // (with a ring, ev0 / evm / ev1 alias one of its slots)
        (void) hipEventDestroy(e);

    pool_free(p->device, p->tan_dev);
    pool_free(p->device, p->rec);
    (void) hipFree(p->path_dev);
    (void) hipFree(p->path_err);
    pool_free(p->device, p->arena);
    pool_free(p->device, p->rays_dev);
    pool_free(p->device, p->grid_dev);
    (void) hipFree(p->seedtab_dev);
    pool_free(p->device, p->image_own);
    pool_free(p->device, p->iang_own);
    pool_free(p->device, p->ctl);
    (void) hipFree(p->probe);
    (void) hipFree(p->bad_dev);
    delete p;
}

int rt_hip_plan_create(rt_hip_plan **out, int device, int N, const rt_beam *beam, const rt_gain *gain,
                       const rt_seed *seed, int method, double scale)
{
    if (!out || !beam || !gain)
        return fail_arg("rt_hip_plan_create: NULL argument");
    *out = nullptr;
    if (N < 2)
        return fail_arg("rt_hip_plan_create: need at least 2 lengths");
    if (method != 1 && method != 2)
        return fail_arg("rt_hip_plan_create: method must be 1 (backward) or 2 (forward)");
    if (beam->nx < 1 || beam->ny < 1 || beam->na < 1 || beam->nb < 1 || beam->nv < 1)
        return fail_arg("rt_hip_plan_create: empty beam grid");
    const int L = N - 1;
    if (L > 64)
        return fail_arg("rt_hip_plan_create: more than 65 lengths are not supported");
    const int K = beam->nv;
    for (int i = 1; i < N; i++) {
        if (gain[i].Nx < 2 || gain[i].Ny < 2 || !gain[i].x || !gain[i].y || !gain[i].n || !gain[i].g0 ||
            !gain[i].gv)
            return fail_arg("rt_hip_plan_create: incomplete gain table");
        if (gain[i].Nv != K)
            return fail_arg("rt_hip_plan_create: gain.Nv != beam.nv");
    }
    int ndev = rt_hip_device_count();
    if (ndev <= 0) {
        g_last_error = "no HIP device";
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev)
        return fail_arg("rt_hip_plan_create: bad device index");
    HIP_TRY(hipSetDevice(device));

    rt_hip_plan *p = new rt_hip_plan();
    p->t_created   = std::chrono::steady_clock::now();
    p->device      = device;
    {
        // the CU count of a device does not change: asked once (hipGetDeviceProperties costs ~0.3 ms a call)
        static std::mutex mu;
        static int cus[64] = {};
        std::lock_guard<std::mutex> lock(mu);
        if (device < 64 && cus[device] > 0) {
            p->cu_count = cus[device];
        } else {
            int n        = 0;
            hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
            if (e != hipSuccess || n < 1) {
                delete p;
                return fail_hip(e, "hipDeviceGetAttribute(multiprocessor count)", __LINE__);
            }
            p->cu_count = n;
            if (device < 64)
                cus[device] = n;
        }
    }

    static const bool timing = getenv("RT_HIP_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap    = [&](const char *what) {
        if (timing) {
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "    plan_create %-18s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
            t_prev = now;
        }
    };
    lap("device");
    // ---- pack the arena -------------------------------------------------
    ArenaBuilder ab;
    std::vector<rt::DevGain> dg((size_t) N);
    std::vector<size_t> off_gv((size_t) N, 0);
    const bool use_emis = gain[0].E0 != nullptr && seed == nullptr; // Helper.h:402
    const int Kp = (K + 3) & ~3; // rows padded to four frequencies (DevParams::Kp)
    for (int i = 1; i < N; i++) {
        const size_t cells = (size_t) gain[i].Nx * (size_t) gain[i].Ny;
        if (cells * (size_t) Kp * sizeof(float) >= (1ull << 32)) { // the frequency kernel addresses rows with 32 bits
            delete p;
            return fail_arg("rt_hip_plan_create: a lineshape table of 4 GiB or more is not supported");
        }
        if (Kp == K) {
            off_gv[(size_t) i] = ab.put(gain[i].gv, sizeof(float) * cells * (size_t) K);
        } else {
            off_gv[(size_t) i] = ab.reserve(sizeof(float) * cells * (size_t) Kp); // resize() zero-fills
            float *dst         = reinterpret_cast<float *>(ab.host.data() + off_gv[(size_t) i]);
            for (size_t c = 0; c < cells; c++)
                memcpy(dst + c * (size_t) Kp, gain[i].gv + c * (size_t) K, sizeof(float) * (size_t) K);
        }
    }
    const size_t off_bx  = ab.put(beam->x, sizeof(double) * (size_t) beam->nx);
    const size_t off_by  = ab.put(beam->y, sizeof(double) * (size_t) beam->ny);
    const size_t off_ba  = ab.put(beam->a, sizeof(double) * (size_t) beam->na);
    const size_t off_bb  = ab.put(beam->b, sizeof(double) * (size_t) beam->nb);
    const size_t off_bdv = ab.reserve(sizeof(double) * (size_t) Kp);
    memcpy(ab.host.data() + off_bdv, beam->dv, sizeof(double) * (size_t) beam->nv);
    const size_t off_bdv2 = ab.reserve(sizeof(double) * (size_t) Kp); // 2 * dv (exact), RayTraceImageCPU.cpp:66
    for (int k = 0; k < beam->nv; k++) {
        const double d2 = 2.0 * beam->dv[k];
        memcpy(ab.host.data() + off_bdv2 + sizeof(double) * (size_t) k, &d2, sizeof(double));
    }
    size_t off_sx[5] = { 0 }, off_sf[5] = { 0 };
    if (seed) {
        for (int i = 0; i < 5; i++) {
            if (seed->dim[i] < 2 || !seed->x[i] || !seed->f[i]) {
                delete p;
                return fail_arg("rt_hip_plan_create: incomplete seed table");
            }
            if (i == 4 && seed->dim[4] != K) {
                delete p;
                return fail_arg("rt_hip_plan_create: seed.dim[4] != beam.nv");
            }
            off_sx[i] = ab.put(seed->x[i], sizeof(double) * (size_t) seed->dim[i]);
            if (i == 4) { // the frequency profile, padded like the lineshape rows
                off_sf[i] = ab.reserve(sizeof(double) * (size_t) Kp);
                memcpy(ab.host.data() + off_sf[i], seed->f[i], sizeof(double) * (size_t) seed->dim[i]);
            } else {
                off_sf[i] = ab.put(seed->f[i], sizeof(double) * (size_t) seed->dim[i]);
            }
        }
    }
    const size_t off_gain = ab.reserve(sizeof(rt::DevGain) * (size_t) N);
    // march blob: headers + grids + fused corner nodes of every length, copied to LDS
    // verbatim by rt_march_kernel<true>
    std::vector<unsigned char> blob(align_up(sizeof(rt::BlobGain) * (size_t) N, 16));
    bool tiny_spacing = false, bad_index = false, all_bounded = true;
    for (int i = 1; i < N; i++) {
        const rt_gain &g  = gain[i];
        const size_t npix = (size_t) g.Nx * (size_t) g.Ny;
        rt::BlobGain h;
        memset(&h, 0, sizeof(h));
        h.lo_x     = (float) g.x[0]; // Helper.h:445-448
        h.hi_x     = (float) g.x[g.Nx - 1];
        h.lo_y     = (float) g.y[0];
        h.hi_y     = (float) g.y[g.Ny - 1];
        h.mirror_y = 0;
        if (h.lo_y >= 0) { // Helper.h:449-453
            h.lo_y     = -h.hi_y;
            h.mirror_y = 1;
        }
        h.Nx     = g.Nx;
        h.Ny     = g.Ny;
        h.x0f = (float) g.x[0];
        h.y0f = (float) g.y[0];
        const double ihx = (double) (g.Nx - 1) / (g.x[g.Nx - 1] - g.x[0]);
        const double ihy = (double) (g.Ny - 1) / (g.y[g.Ny - 1] - g.y[0]);
        h.inv_hxf        = std::isfinite(ihx) && fabs(ihx) < 1e30 ? (float) ihx : 0.0f;
        h.inv_hyf        = std::isfinite(ihy) && fabs(ihy) < 1e30 ? (float) ihy : 0.0f;
        // per-interval records of both axes (entry 0 unused)
        auto put_intervals = [&](const double *gp, int n, bool mirrored) {
            const int off = (int) blob.size();
            blob.resize(blob.size() + sizeof(rt::Interval) * (size_t) n);
            rt::Interval *iv = reinterpret_cast<rt::Interval *>(blob.data() + off);
            memset(iv, 0, sizeof(rt::Interval) * (size_t) n);
            for (int k = 1; k < n; k++) {
                const double lo = gp[k - 1], hi = gp[k], hk = hi - lo;
                if (!(hk >= 1e-30)) // see rt_math.h, div_by_recip<TINY_OK>; and the integrator's step limits
                    tiny_spacing = true; // 0.1f * (float) hk must be positive (rt_march.hip, block [C])
                iv[k].lo   = lo;
                iv[k].hi   = hi;
                iv[k].rh   = 1.0 / hk;
                iv[k].rw   = 1.0 / (double) (float) hk;
                iv[k].w    = (float) hk;
                iv[k].b_lo = (float) (lo - 0.1 * hk);
                iv[k].b_hi = (float) (hi + 0.1 * hk);
                if (mirrored && k == 1)
                    iv[k].b_lo = -iv[k].b_hi;
            }
            return off;
        };
        h.off_ix   = put_intervals(g.x, g.Nx, false);
        h.off_iy   = put_intervals(g.y, g.Ny, h.mirror_y != 0);
        h.off_node = (int) blob.size();
        blob.resize(blob.size() + sizeof(rt::Node) * npix);
        rt::Node *nd = reinterpret_cast<rt::Node *>(blob.data() + h.off_node);
        for (size_t c = 0; c < npix; c++) { // the three gathered quantities fused per grid point
            if (!std::isfinite(g.n[c]))
                bad_index = true; // (the reference's integrator loop would never advance: Helper.h:279-280)
            nd[c].n  = g.n[c];
            nd[c].g0 = g.g0[c];
            nd[c].E0 = g.E0 ? g.E0[c] : 0.0f;
        }
        memcpy(blob.data() + sizeof(rt::BlobGain) * (size_t) i, &h, sizeof(h));
        // Ranges for the short division sequences of the integrator (rt_math.h, fdiv_nr): with dn = the largest
        // difference of the index between neighbouring nodes, a step sees n within [min n - dn, max n + dn]
        // (bilinear value on the cell box with its 10 % margin, plus |r| < 0.1 w times a gradient of at most
        // 1.3 dn / w per axis) and index gradients of at most 1.3 dn / min(w).
        {
            double n_lo = g.n[0], n_hi = g.n[0], dn = 0.0, w_min = g.x[1] - g.x[0];
            for (int k = 1; k < g.Nx; k++)
                w_min = std::min(w_min, g.x[k] - g.x[k - 1]);
            for (int k = 1; k < g.Ny; k++)
                w_min = std::min(w_min, g.y[k] - g.y[k - 1]);
            for (int iy = 0; iy < g.Ny; iy++) {
                const double *row = g.n + (size_t) iy * (size_t) g.Nx;
                for (int ix = 0; ix < g.Nx; ix++) {
                    n_lo = std::min(n_lo, row[ix]);
                    n_hi = std::max(n_hi, row[ix]);
                    if (ix > 0)
                        dn = std::max(dn, fabs(row[ix] - row[ix - 1]));
                    if (iy > 0)
                        dn = std::max(dn, fabs(row[ix] - row[ix - g.Nx]));
                }
            }
            if (!(n_lo - dn >= 0.25 && n_hi + dn <= 4.0 && dn / w_min <= 1e12 && w_min >= 1e-12))
                all_bounded = false;
        }
    }
    if (!(beam->dz >= 1e-12 && beam->dz <= 1e12))
        all_bounded = false;
    p->tables_bounded = all_bounded;
    if (tiny_spacing) {
        delete p;
        return fail_arg("rt_hip_plan_create: gain grid not strictly increasing, or spacing below 1e-30");
    }
    if (bad_index) {
        delete p;
        return fail_arg("rt_hip_plan_create: non-finite index of refraction");
    }
    const size_t off_blob = ab.put(blob.data(), blob.size());
    lap("pack");

#define PLAN_TRY(expr)                                   \
    do {                                                 \
        hipError_t e_ = (expr);                          \
        if (e_ != hipSuccess) {                          \
            rt_hip_plan_destroy(p);                      \
            return fail_hip(e_, #expr, __LINE__);        \
        }                                                \
    } while (0)

    p->arena_bytes = align_up(ab.host.size(), 256);
    PLAN_TRY(pool_alloc(device, (void **) &p->arena, p->arena_bytes));
    unsigned char *A = p->arena;
    p->gv_dev.assign((size_t) N, nullptr);
    for (int i = 1; i < N; i++) {
        dg[(size_t) i].gv    = reinterpret_cast<const float *>(A + off_gv[(size_t) i]);
        p->gv_dev[(size_t) i] = dg[(size_t) i].gv;
    }
    p->dv2_dev = reinterpret_cast<const double *>(A + off_bdv2);
    memcpy(ab.host.data() + off_gain, dg.data(), sizeof(rt::DevGain) * (size_t) N);
    PLAN_TRY(hipMemcpy(p->arena, ab.host.data(), ab.host.size(), hipMemcpyHostToDevice));
    lap("alloc + upload");

    rt::DevParams &P = p->P;
    P.N        = N;
    P.L        = L;
    P.K        = K;
    P.Kp       = Kp;
    P.method   = method;
    P.use_emis = use_emis ? 1 : 0;
    P.has_seed = seed ? 1 : 0;
    P.dz0      = (float) beam->dz; // RayTraceImageCPU.cpp:31: double -> float at the call
    P.scale    = scale;
    P.beam.x   = reinterpret_cast<const double *>(A + off_bx);
    P.beam.y   = reinterpret_cast<const double *>(A + off_by);
    P.beam.a   = reinterpret_cast<const double *>(A + off_ba);
    P.beam.b   = reinterpret_cast<const double *>(A + off_bb);
    P.beam.dv  = reinterpret_cast<const double *>(A + off_bdv);
    P.beam.nx  = beam->nx;
    P.beam.ny  = beam->ny;
    P.beam.na  = beam->na;
    P.beam.nb  = beam->nb;
    P.beam.nv  = beam->nv;
    P.beam.dx  = beam->dx;
    P.beam.dy  = beam->dy;
    P.beam.da  = beam->da;
    P.beam.db  = beam->db;
    P.beam.inv_dx = 1.0 / beam->dx;
    P.beam.inv_dy = 1.0 / beam->dy;
    P.beam.inv_da = 1.0 / beam->da;
    P.beam.inv_db = 1.0 / beam->db;
    P.beam.g_first[0] = beam->x[0];
    P.beam.g_first[1] = beam->y[0];
    P.beam.g_first[2] = beam->a[0];
    P.beam.g_first[3] = beam->b[0];
    P.beam.g_last[0]  = beam->x[beam->nx - 1];
    P.beam.g_last[1]  = beam->y[beam->ny - 1];
    P.beam.g_last[2]  = beam->a[beam->na - 1];
    P.beam.g_last[3]  = beam->b[beam->nb - 1];
    if (seed) {
        for (int i = 0; i < 5; i++) {
            P.seed.x[i]   = reinterpret_cast<const double *>(A + off_sx[i]);
            P.seed.f[i]   = reinterpret_cast<const double *>(A + off_sf[i]);
            P.seed.dim[i] = seed->dim[i];
        }
        P.seed.f0 = seed->f0;
    }
    P.gain = reinterpret_cast<const rt::DevGain *>(A + off_gain);
    P.blob       = A + off_blob;
    P.blob_bytes = (unsigned) blob.size();
    if (const char *dbg = getenv("RT_HIP_DEBUG"))
        P.debug = (unsigned) strtoul(dbg, nullptr, 0);
    if (const char *ex = getenv("RT_HIP_EXACT_EMISSION")) // the loop signature has no parameter for it
        P.exact_emis = atoi(ex) ? 1 : 0;

    p->beam_x.assign(beam->x, beam->x + beam->nx);
    p->beam_y.assign(beam->y, beam->y + beam->ny);
    p->beam_a.assign(beam->a, beam->a + beam->na);
    p->beam_b.assign(beam->b, beam->b + beam->nb);
    p->n_image = (size_t) beam->nx * (size_t) beam->ny * (size_t) beam->nv;
    p->n_iang  = (size_t) beam->na * (size_t) beam->nb;
    PLAN_TRY(pool_alloc(device, (void **) &p->ctl, sizeof(rt::DevCtl)));
    PLAN_TRY(hipEventCreate(&p->ev0));
    PLAN_TRY(hipEventCreate(&p->ev1));
    PLAN_TRY(hipEventCreate(&p->evm));
    P.rec_stride = (unsigned) align_up((size_t) L * RT_N_SUB * 12 + sizeof(rt::RecMeta), 16);
    P.c_cap      = 0.5f * 1.00001f; // step safety factor c = 0.5 (Helper.h:381), see rt_hip_plan_set_step_factor
    P.c_h1       = 0.5f * 0.1f;
    P.c_h3       = 0.5f * 0.05f;
    {
        // largest finite |lineshape value| of the planes the emission-mode frequency pass reads (integer
        // maximum of the magnitude bits: non-negative floats order like their bit patterns, and the loop
        // vectorises)
        float wmax = 0.0f;
        if (use_emis) {
            uint32_t umax = 0;
            for (int i = 1; i < N; i++) {
                const size_t n    = (size_t) gain[i].Nx * (size_t) gain[i].Ny * (size_t) K;
                const uint32_t *u = reinterpret_cast<const uint32_t *>(gain[i].gv);
                uint32_t m = 0, mall = 0;
                for (size_t c = 0; c < n; c++) {
                    uint32_t a = u[c] & 0x7fffffffu;
                    mall       = a > mall ? a : mall;
                    a          = a < 0x7f800000u ? a : 0u; // inf and NaN do not count
                    m          = a > m ? a : m;
                }
                umax = m > umax ? m : umax;
                if (mall > 0x7f800000u) // a NaN: the frequency kernel then tests every value it reads
                    p->gv_has_nan = true;
            }
            memcpy(&wmax, &umax, sizeof(wmax));
        }
        P.gs_cap = wmax > 0.0f ? 708.0f / wmax : FLT_MAX;
        if (!(P.gs_cap <= FLT_MAX))
            P.gs_cap = FLT_MAX;
    }
    P.ctl = p->ctl;
    *out  = p;
    lap("rest");
    return RT_OK;
}

// rt_hip_image_loop only: the list stays on the host until the run, which uploads it in slices
// beside the march (the caller's buffer outlives the call, the plan does not)
// the kernels index rays with 32 bits, and the march's ray counter overshoots the end: every wave of the
// persistent grid adds one more reservation after the rays have run out (at most 8192 waves x 4096 rays, the
// cap of RT_HIP_MARCH_CHUNK) -- the counter must not wrap
constexpr size_t MAX_LIST_RAYS = 0xffffffffull - (1ull << 26);

static int plan_set_rays_deferred(rt_hip_plan *p, const rt_ray *rays, size_t n_rays)
{
    if (n_rays > MAX_LIST_RAYS)
        return fail_arg("ray list of 2^32 - 512 rays or more: split the call");
    HIP_TRY(hipSetDevice(p->device));
    plan_quiesce(p);
    pool_free(p->device, p->rays_dev);
    pool_free(p->device, p->tan_dev);
    p->rays_dev = nullptr;
    p->tan_dev  = nullptr;
    if (n_rays) {
        HIP_TRY(pool_alloc(p->device, (void **) &p->rays_dev, n_rays * sizeof(rt_ray)));
        HIP_TRY(pool_alloc(p->device, (void **) &p->tan_dev, n_rays * 2 * sizeof(float)));
    }
    p->P.exclusive  = 0;
    p->P.own_cells  = 0;
    p->P.rays       = {};
    p->P.rays.list  = p->rays_dev;
    p->P.rays.sxy   = p->tan_dev;
    p->P.rays.count = n_rays;
    p->n_rays       = n_rays;
    p->host_rays    = n_rays ? rays : nullptr;
    return RT_OK;
}

int rt_hip_plan_set_rays(rt_hip_plan *p, const rt_ray *rays, size_t n_rays)
{
    if (!p || (n_rays && !rays))
        return fail_arg("rt_hip_plan_set_rays: NULL argument");
    if (n_rays > MAX_LIST_RAYS)
        return fail_arg("rt_hip_plan_set_rays: 2^32 - 512 rays or more: split the call");
    HIP_TRY(hipSetDevice(p->device));
    plan_quiesce(p);
    pool_free(p->device, p->rays_dev);
    p->rays_dev = nullptr;
    if (n_rays) {
        HIP_TRY(pool_alloc(p->device, (void **) &p->rays_dev, n_rays * sizeof(rt_ray)));
        HIP_TRY(hipMemcpy(p->rays_dev, rays, n_rays * sizeof(rt_ray), hipMemcpyHostToDevice));
    }
    pool_free(p->device, p->tan_dev);
    p->tan_dev = nullptr;
    if (n_rays) {
        // Helper.h:409-410 for every ray, at full lane occupancy, before the march
        HIP_TRY(pool_alloc(p->device, (void **) &p->tan_dev, n_rays * 2 * sizeof(float)));
        if (tan_mode(p->device) == 2) {
            std::vector<float> h(n_rays * 2);
            host_tangents(rays, n_rays, h.data());
            HIP_TRY(hipMemcpy(p->tan_dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            const unsigned blocks = (unsigned) ((n_rays + 255) / 256);
            hipLaunchKernelGGL(rt::rt_tan_kernel, dim3(blocks), dim3(256), 0, nullptr, p->rays_dev,
                               (unsigned long long) n_rays, p->tan_dev);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    p->P.exclusive  = 0;
    p->P.own_cells  = 0;
    p->P.rays       = {};
    p->P.rays.list  = p->rays_dev;
    p->P.rays.sxy   = p->tan_dev;
    p->P.rays.count = n_rays;
    p->n_rays       = n_rays;
    p->host_rays    = nullptr;
    return RT_OK;
}

// x / d for every x < 2^31 as mulhi(x, mul) >> sh (DevRays::div_mul): with s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1
// satisfies mul d = 2^(31+s) + e, 0 < e <= d <= 2^s, hence x mul / 2^(31+s) = x / d + x e / (d 2^(31+s)) with the
// second term below 1 / d: the floor is that of x / d.  mul < 2^32 for d >= 2; d = 1 is flagged by mul = 0.
static void magic_u31(unsigned d, unsigned &mul, unsigned &sh)
{
    if (d <= 1) {
        mul = 0;
        sh  = 0;
        return;
    }
    unsigned s = 0;
    while ((1ull << s) < d)
        s++;
    mul = (unsigned) ((1ull << (31 + s)) / d + 1);
    sh  = s - 1;
}

// RayTraceImageCPU.cpp:11-16 on the host: grid point i (rounded to float, as the ray carries it) must fall in
// deposit cell i of the grid g with spacing d -- what makes "ray column i deposits into pixel column i" true
static bool grid_points_in_own_cells(const double *g, int n, double d)
{
    for (int i = 0; i < n; i++) {
        const double v = (double) (float) g[i];
        if (v < g[0] - 0.5 * d || v > g[n - 1] + 0.5 * d)
            return false;
        const double t = v - 0.5 * d;
        int idx        = 0;
        if (t < g[0])
            idx = 0;
        else if (t > g[n - 1])
            idx = n;
        else {
            int lo = 0, hi = n - 1;
            if (n == 1)
                hi = 1;
            while (n > 1 && hi - lo != 1) {
                int mid = (hi + lo) / 2;
                if (g[mid] >= t)
                    hi = mid;
                else
                    lo = mid;
            }
            idx = hi;
        }
        if (idx != i)
            return false;
    }
    return true;
}

int rt_hip_plan_set_ray_grid(rt_hip_plan *p, const double *gx, int ngx, const double *gy, int ngy,
                             const double *ga, int nga, const double *gb, int ngb, int64_t first,
                             int64_t stride, int64_t count)
{
    if (!p || !gx || !gy || !ga || !gb || ngx < 1 || ngy < 1 || nga < 1 || ngb < 1)
        return fail_arg("rt_hip_plan_set_ray_grid: bad grid");
    const int64_t total = (int64_t) ngx * ngy * nga * ngb;
    if (total > 0x7fffffffLL) // the reference indexes rays with int (RayTraceImage.cpp:302)
        return fail_arg("rt_hip_plan_set_ray_grid: more than 2^31 rays");
    if (first < 0 || stride < 1 || count < 0 || (count > 0 && first + (count - 1) * stride >= total))
        return fail_arg("rt_hip_plan_set_ray_grid: ray range outside the grid");
    HIP_TRY(hipSetDevice(p->device));
    plan_quiesce(p);
    pool_free(p->device, p->grid_dev);
    p->grid_dev     = nullptr;
    (void) hipFree(p->seedtab_dev);
    p->seedtab_dev  = nullptr;
    const size_t nn = (size_t) ngx + (size_t) ngy + (size_t) nga + (size_t) ngb;
    std::vector<double> h(nn);
    memcpy(h.data(), gx, sizeof(double) * (size_t) ngx);
    memcpy(h.data() + ngx, gy, sizeof(double) * (size_t) ngy);
    memcpy(h.data() + ngx + ngy, ga, sizeof(double) * (size_t) nga);
    memcpy(h.data() + ngx + ngy + nga, gb, sizeof(double) * (size_t) ngb);
    HIP_TRY(pool_alloc(p->device, (void **) &p->grid_dev, nn * sizeof(double)));
    HIP_TRY(hipMemcpy(p->grid_dev, h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    // Helper.h:409-410: tanf(1e-3f * ray.a) depends only on the grid value: nga + ngb
    // evaluations on the host, with the same libm the CPU loop uses
    std::vector<float> ht((size_t) nga + (size_t) ngb);
    for (int k = 0; k < nga; k++)
        ht[(size_t) k] = tanf(1e-3f * (float) ga[k]);
    for (int m = 0; m < ngb; m++)
        ht[(size_t) nga + (size_t) m] = tanf(1e-3f * (float) gb[m]);
    pool_free(p->device, p->tan_dev);
    p->tan_dev = nullptr;
    HIP_TRY(pool_alloc(p->device, (void **) &p->tan_dev, ht.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(p->tan_dev, ht.data(), ht.size() * sizeof(float), hipMemcpyHostToDevice));
    p->host_rays   = nullptr;
    rt::DevRays &R = p->P.rays;
    R              = {};
    R.list         = nullptr;
    R.tan_a        = p->tan_dev;
    R.tan_b        = p->tan_dev + nga;
    R.gx           = p->grid_dev;
    R.gy           = p->grid_dev + ngx;
    R.ga           = p->grid_dev + ngx + ngy;
    R.gb           = p->grid_dev + ngx + ngy + nga;
    R.ngx          = ngx;
    R.ngy          = ngy;
    R.nga          = nga;
    R.ngb          = ngb;
    R.first        = first;
    R.stride       = stride;
    R.count        = (unsigned long long) count;
    p->n_rays      = (unsigned long long) count;
    {
        const int divisors[3] = { ngb, nga, ngy };
        for (int t = 0; t < 3; t++)
            magic_u31((unsigned) divisors[t], R.div_mul[t], R.div_sh[t]);
    }
    if (p->P.has_seed && p->P.method != 1) {
        HIP_TRY(dev_malloc((void **) &p->seedtab_dev, nn * sizeof(double) + nn));
        unsigned char *flags = reinterpret_cast<unsigned char *>(p->seedtab_dev + nn);
        hipLaunchKernelGGL(rt::rt_seed_tab_kernel, dim3((unsigned) ((nn + 255) / 256)), dim3(256), 0, nullptr, p->P.seed,
                           R, p->seedtab_dev, flags);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        R.sf  = p->seedtab_dev;
        R.sin = flags;
    }
    // One ray per pixel, every pixel covered: in ASE mode ray ijkm lands in pixel (i, j)
    // (SURVEY.md 8(c) i) -- the deposit index of the ray is verified per ray by the kernel,
    // which falls back to atomics for any ray that does not land in its own pixel.
    // (as the floats a ray carries: a list recognised as a grid, rt_hip_image_loop, arrives as floats)
    auto same = [](const std::vector<double> &v, const double *g, int n) {
        if ((int) v.size() != n)
            return false;
        for (int i = 0; i < n; i++) {
            const float a = (float) v[(size_t) i], b = (float) g[i];
            if (memcmp(&a, &b, sizeof(float)) != 0)
                return false;
        }
        return true;
    };
    auto own_cell = [](const std::vector<double> &g, double d) { return grid_points_in_own_cells(g.data(), (int) g.size(), d); };
    p->P.own_cells = (p->P.method == 1 && same(p->beam_x, gx, ngx) && same(p->beam_y, gy, ngy) && same(p->beam_a, ga, nga) &&
                      same(p->beam_b, gb, ngb) && own_cell(p->beam_x, p->P.beam.dx) && own_cell(p->beam_y, p->P.beam.dy) &&
                      own_cell(p->beam_a, p->P.beam.da) && own_cell(p->beam_b, p->P.beam.db) &&
                      !getenv("RT_HIP_NO_OWN_CELLS"))
                         ? 1u
                         : 0u;
    p->P.exclusive = (p->P.own_cells && nga == 1 && ngb == 1 && first == 0 && stride == 1 && count == total) ? 1u : 0u;
    return RT_OK;
}

int rt_hip_plan_set_exact_emission(rt_hip_plan *p, int on)
{
    if (!p)
        return fail_arg("rt_hip_plan_set_exact_emission: NULL plan");
    p->P.exact_emis = on ? 1 : 0;
    return RT_OK;
}

int rt_hip_plan_set_step_factor(rt_hip_plan *p, double c)
{
    if (!p || !(c > 0.0) || !(c < 1.0))
        return fail_arg("rt_hip_plan_set_step_factor: c must be in (0, 1)");
    const float cf = (float) c; // RayTraceImage.cpp:462: (float) c at the call
    p->P.c_cap     = cf * 1.00001f;
    p->P.c_h1      = cf * 0.1f;
    p->P.c_h3      = cf * 0.05f;
    return RT_OK;
}

int rt_hip_plan_set_debug(rt_hip_plan *p, unsigned bits)
{
    if (!p)
        return fail_arg("rt_hip_plan_set_debug: NULL plan");
    p->P.debug = bits;
    return RT_OK;
}

int rt_hip_plan_enable_path(rt_hip_plan *p, int on)
{
    if (!p)
        return fail_arg("rt_hip_plan_enable_path: NULL plan");
    p->path_on = on != 0;
    return RT_OK;
}

int rt_hip_plan_fetch_path(rt_hip_plan *p, float *path, int32_t *err)
{
    if (!p || !p->ran || !p->path_on || !p->path_dev)
        return fail_arg("rt_hip_plan_fetch_path: the path tracer was not enabled for the last run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->last_stream));
    const size_t n2 = (size_t) p->P.L * RT_N_SUB + 1;
    if (path)
        HIP_TRY(hipMemcpy(path, p->path_dev, (size_t) p->n_rays * n2 * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (err)
        HIP_TRY(hipMemcpy(err, p->path_err, (size_t) p->n_rays * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_hip_plan_enable_probe(rt_hip_plan *p, int on)
{
    if (!p)
        return fail_arg("rt_hip_plan_enable_probe: NULL plan");
    p->probe_on = on != 0;
    return RT_OK;
}

static int plan_prepare_probe(rt_hip_plan *p)
{
    const size_t n = (size_t) p->n_rays;
    if (!p->probe_on) {
        p->P.probe_on = 0;
        return RT_OK;
    }
    if (p->probe_rays != n || !p->probe) {
        (void) hipFree(p->probe);
        p->probe = nullptr;
        size_t bytes = n * (sizeof(rt_ray) + 8) + 1024;
        HIP_TRY(dev_malloc((void **) &p->probe, bytes));
        p->probe_rays = n;
    }
    unsigned char *b = p->probe;
    p->P.probe.ray2  = reinterpret_cast<rt_ray *>(b);
    b += n * sizeof(rt_ray);
    p->P.probe.flags = reinterpret_cast<uint32_t *>(b);
    b += n * 4;
    p->P.probe.steps = reinterpret_cast<uint32_t *>(b);
    p->P.probe_on    = 1;
    return RT_OK;
}

int rt_hip_plan_run(rt_hip_plan *p, void *stream_v, double *image_dev, double *iang_dev)
{
    if (!p)
        return fail_arg("rt_hip_plan_run: NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (!image_dev) {
        if (!p->image_own)
            HIP_TRY(pool_alloc(p->device, (void **) &p->image_own, p->n_image * sizeof(double)));
        image_dev = p->image_own;
    }
    if (!iang_dev) {
        if (!p->iang_own)
            HIP_TRY(pool_alloc(p->device, (void **) &p->iang_own, p->n_iang * sizeof(double)));
        iang_dev = p->iang_own;
    }
    int rc = plan_prepare_probe(p);
    if (rc != RT_OK)
        return rc;
    if (p->probe_on && p->n_rays)
        HIP_TRY(hipMemsetAsync(p->probe, 0, (size_t) p->n_rays * (sizeof(rt_ray) + 8), stream));
    if (!p->P.exclusive) // exclusive mode writes every image row exactly once
        HIP_TRY(hipMemsetAsync(image_dev, 0, p->n_image * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(iang_dev, 0, p->n_iang * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(p->ctl, 0, sizeof(rt::DevCtl), stream));
    p->P.image   = image_dev;
    p->P.iang    = iang_dev;
    p->P.n_tiles = (unsigned) ((p->n_rays + rt::WAVE - 1) / rt::WAVE);
    if (!p->ring.empty()) {
        const size_t slot = (size_t) (p->runs % (p->ring.size() / 3)) * 3;
        p->ev0            = p->ring[slot];
        p->evm            = p->ring[slot + 1];
        p->ev1            = p->ring[slot + 2];
    }
    p->runs++;

    rc = plan_run_split(p, stream);
    if (rc != RT_OK)
        return rc;
    p->last_stream = stream;
    p->last_image  = image_dev;
    p->last_iang   = iang_dev;
    p->ran         = true;
    p->repeated    = false;
    return RT_OK;
}

int rt_hip_plan_fetch(rt_hip_plan *p, double *image, double *I_ang, unsigned int *failure_code,
                      rt_ray *failed_rays, int max_failed, int *n_failed, rt_stats *stats)
{
    if (!p || !p->ran)
        return fail_arg("rt_hip_plan_fetch: plan has not run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->last_stream));
    {
        // rays that failed in the frequency pass have been deposited: repeat the pass without them
        unsigned code = 0;
        HIP_TRY(hipMemcpy(&code, &p->ctl->failure_code, sizeof(code), hipMemcpyDeviceToHost));
        if ((code & ((1u << 2) | (1u << 3))) && !p->path_on && !(p->P.debug & 1u) && !p->repeated) {
            const int rc = plan_repeat_checked(p);
            if (rc != RT_OK)
                return rc;
            p->repeated = true; // this run's outputs are final; a second fetch must not repeat again
        }
    }
    if (image)
        HIP_TRY(hipMemcpy(image, p->last_image, p->n_image * sizeof(double), hipMemcpyDeviceToHost));
    if (I_ang)
        HIP_TRY(hipMemcpy(I_ang, p->last_iang, p->n_iang * sizeof(double), hipMemcpyDeviceToHost));
    rt::DevCtl c;
    HIP_TRY(hipMemcpy(&c, p->ctl, sizeof(c), hipMemcpyDeviceToHost));
    if (failure_code)
        *failure_code = c.failure_code;
    int nf = (int) (c.n_failed < RT_N_FAILED_MAX ? c.n_failed : RT_N_FAILED_MAX);
    if (nf > max_failed)
        nf = max_failed;
    if (failed_rays)
        for (int i = 0; i < nf; i++)
            failed_rays[i] = c.failed[i];
    if (n_failed)
        *n_failed = failed_rays ? nf : 0;
    if (stats) {
        stats->n_rays     = c.n_rays;
        stats->cell_steps = c.cell_steps;
        stats->n_escaped  = c.n_escaped;
        stats->n_skipped  = c.n_skipped;
        float ms          = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        stats->kernel_ms = ms;
        HIP_TRY(hipEventElapsedTime(&stats->march_ms, p->ev0, p->evm));
        HIP_TRY(hipEventElapsedTime(&stats->freq_ms, p->evm, p->ev1));
        stats->total_ms  = (float) std::chrono::duration<double, std::milli>(
                              std::chrono::steady_clock::now() - p->t_created).count();
    }
    return RT_OK;
}

int rt_hip_plan_kernel_ms(rt_hip_plan *p, float *ms)
{
    if (!p || !p->ran || !ms)
        return fail_arg("rt_hip_plan_kernel_ms: plan has not run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(ms, p->ev0, p->ev1));
    return RT_OK;
}

int rt_hip_plan_set_timing_ring(rt_hip_plan *p, int n_runs)
{
    if (!p || n_runs < 1 || n_runs > 4096)
        return fail_arg("rt_hip_plan_set_timing_ring: 1 .. 4096 runs");
    HIP_TRY(hipSetDevice(p->device));
    plan_quiesce(p);
    if (p->ring.empty()) { // the plan's own triple becomes slot 0
        p->ring = { p->ev0, p->evm, p->ev1 };
    }
    while (p->ring.size() < (size_t) n_runs * 3) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        p->ring.push_back(e);
    }
    while (p->ring.size() > (size_t) n_runs * 3) {
        (void) hipEventDestroy(p->ring.back());
        p->ring.pop_back();
    }
    p->ev0  = p->ring[0];
    p->evm  = p->ring[1];
    p->ev1  = p->ring[2];
    p->runs = 0;
    p->ran  = false;
    return RT_OK;
}

int rt_hip_plan_ring_times(rt_hip_plan *p, float *march_ms, float *freq_ms, int max_runs, int *n_runs)
{
    if (!p || !march_ms || !freq_ms || !n_runs || max_runs < 0)
        return fail_arg("rt_hip_plan_ring_times: bad argument");
    *n_runs = 0;
    if (!p->ran || p->ring.empty())
        return RT_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->last_stream));
    const unsigned long long slots = p->ring.size() / 3;
    const unsigned long long have  = p->runs < slots ? p->runs : slots;
    const unsigned long long take  = have < (unsigned long long) max_runs ? have : (unsigned long long) max_runs;
    for (unsigned long long i = 0; i < take; i++) { // oldest first
        const size_t slot = (size_t) ((p->runs - take + i) % slots) * 3;
        HIP_TRY(hipEventElapsedTime(&march_ms[i], p->ring[slot], p->ring[slot + 1]));
        HIP_TRY(hipEventElapsedTime(&freq_ms[i], p->ring[slot + 1], p->ring[slot + 2]));
    }
    *n_runs = (int) take;
    return RT_OK;
}

int rt_hip_plan_kernel_times(rt_hip_plan *p, float *march_ms, float *freq_ms)
{
    if (!p || !p->ran || !march_ms || !freq_ms)
        return fail_arg("rt_hip_plan_kernel_times: plan has not run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(march_ms, p->ev0, p->evm));
    HIP_TRY(hipEventElapsedTime(freq_ms, p->evm, p->ev1));
    return RT_OK;
}

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

double *rt_hip_plan_image_ptr(rt_hip_plan *p) { return p ? p->image_own : nullptr; }
double *rt_hip_plan_iang_ptr(rt_hip_plan *p) { return p ? p->iang_own : nullptr; }

int rt_hip_plan_fetch_probe(rt_hip_plan *p, float *gvl, float *evl, int32_t *ivl, rt_ray *ray2,
                            uint32_t *flags, uint32_t *steps)
{
    if (!p || !p->ran || !p->probe_on || !p->probe)
        return fail_arg("rt_hip_plan_fetch_probe: probe was not enabled for the last run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->last_stream));
    const size_t n = (size_t) p->n_rays, S = (size_t) p->P.L * RT_N_SUB;
    {
        // the march records themselves are the probe: de-interleave them
        std::vector<unsigned char> h(n * p->P.rec_stride);
        if (n)
            HIP_TRY(hipMemcpy(h.data(), p->rec, h.size(), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < n; r++) {
            const unsigned char *rec = h.data() + r * p->P.rec_stride;
            const rt::RecMeta *mt = reinterpret_cast<const rt::RecMeta *>(rec + 12 * S);
            for (size_t q = 0; q < S; q++) {
                const rt::RecSlot sl = rt::rec_slot(rec, (int) q, (int) S, mt->flags_steps, p->P.method == 1);
                if (gvl)
                    gvl[r * S + q] = sl.g;
                if (evl)
                    evl[r * S + q] = sl.e;
                if (ivl)
                    ivl[r * S + q] = sl.c;
            }
        }
    }
    if (ray2)
        HIP_TRY(hipMemcpy(ray2, p->P.probe.ray2, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    if (flags)
        HIP_TRY(hipMemcpy(flags, p->P.probe.flags, n * 4, hipMemcpyDeviceToHost));
    if (steps)
        HIP_TRY(hipMemcpy(steps, p->P.probe.steps, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- a ray list that is really a tensor grid ---------------------------------------------------
// RayTrace::create_image builds its list from four 1-D grids, b fastest, then a, y, x
// (src/RayTraceImage.cpp:300-328), and hands the back-end loop only the list.  The grids are read back
// from it in O(nx + ny + na + nb) -- the period of each coordinate -- and the whole list is then
// compared with the grid ray by ray, bit for bit, on host threads.
struct GridGuess {
    std::vector<double> g[4]; // x, y, a, b as the doubles of the floats the rays carry
};
static inline bool same_bits(float a, float b) { return memcmp(&a, &b, sizeof(float)) == 0; }

static bool guess_ray_grid(const rt_ray *rays, size_t n, GridGuess &G)
{
    if (n == 0)
        return false;
    // period of b: the first later ray whose b equals that of ray 0 (grid values are distinct)
    size_t nb = 1;
    while (nb < n && !same_bits(rays[nb].b, rays[0].b))
        nb++;
    size_t na = 1;
    while (na * nb < n && !same_bits(rays[na * nb].a, rays[0].a))
        na++;
    size_t ny = 1;
    while (ny * na * nb < n && !same_bits(rays[ny * na * nb].y, rays[0].y))
        ny++;
    const size_t block = nb * na * ny;
    if (block == 0 || n % block != 0)
        return false;
    const size_t nx = n / block;
    if (nx > 0x7fffffffull || ny > 0x7fffffffull || na > 0x7fffffffull || nb > 0x7fffffffull || n > 0x7fffffffull)
        return false;
    G.g[0].resize(nx);
    G.g[1].resize(ny);
    G.g[2].resize(na);
    G.g[3].resize(nb);
    for (size_t i = 0; i < nx; i++)
        G.g[0][i] = (double) rays[i * block].x;
    for (size_t j = 0; j < ny; j++)
        G.g[1][j] = (double) rays[j * na * nb].y;
    for (size_t k = 0; k < na; k++)
        G.g[2][k] = (double) rays[k * nb].a;
    for (size_t m = 0; m < nb; m++)
        G.g[3][m] = (double) rays[m].b;
    return true;
}

// every ray of the list against the grid, on up to `threads` host threads.  A ray is two 64-bit words,
// (x, y) and (a, b); a run of nb rays shares the first word and the a half of the second, so the
// inner loop is two integer compares per ray over a stream the memory system prefetches: ~1 ms for the
// 102 MB of a 6.4 M-ray list on 16 threads, hidden behind the kernels it runs beside.
static bool verify_ray_grid(const rt_ray *rays, size_t n, const GridGuess &G, unsigned threads)
{
    const size_t nb = G.g[3].size(), na = G.g[2].size(), ny = G.g[1].size();
    auto bits = [](double v) {
        const float f = (float) v;
        uint32_t u;
        memcpy(&u, &f, sizeof(u));
        return (uint64_t) u;
    };
    std::vector<uint64_t> bx(G.g[0].size()), by(ny), ba(na), bb(nb);
    for (size_t i = 0; i < bx.size(); i++)
        bx[i] = bits(G.g[0][i]);
    for (size_t i = 0; i < ny; i++)
        by[i] = bits(G.g[1][i]) << 32;
    for (size_t i = 0; i < na; i++)
        ba[i] = bits(G.g[2][i]);
    for (size_t i = 0; i < nb; i++)
        bb[i] = bits(G.g[3][i]) << 32;
    const size_t rows = n / nb; // runs of nb rays that differ only in b
    threads           = threads < 1 ? 1 : threads;
    if (n < (size_t) 1 << 18)
        threads = 1;
    std::atomic<bool> ok(true);
    auto work = [&](size_t r0, size_t r1) {
        uint64_t diff = 0;
        for (size_t r = r0; r < r1; r++) {
            const size_t k = r % na, j = (r / na) % ny, i = r / (na * ny);
            const uint64_t xy = bx[i] | by[j], a = ba[k];
            uint64_t w[2];
            const unsigned char *row = reinterpret_cast<const unsigned char *>(rays + r * nb);
            for (size_t m = 0; m < nb; m++) {
                memcpy(w, row + 16 * m, 16);
                diff |= (w[0] ^ xy) | (w[1] ^ (a | bb[m]));
            }
            if ((r & 1023) == 1023 && (diff != 0 || !ok.load(std::memory_order_relaxed)))
                break;
        }
        if (diff != 0)
            ok.store(false, std::memory_order_relaxed);
    };
    if (threads == 1) {
        work(0, rows);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < threads; t++)
            th.emplace_back(work, rows * t / threads, rows * (t + 1) / threads);
        for (auto &t : th)
            t.join();
    }
    return ok.load();
}

static unsigned host_threads(unsigned cap)
{
    unsigned h = std::thread::hardware_concurrency();
    h          = h ? h : 1;
    return env_unsigned("RT_HIP_HOST_THREADS", h < cap ? h : cap, 1, 64);
}

static int plan_set_guessed_grid(rt_hip_plan *p, const GridGuess &G, int64_t first, int64_t count)
{
    return rt_hip_plan_set_ray_grid(p, G.g[0].data(), (int) G.g[0].size(), G.g[1].data(), (int) G.g[1].size(),
                                    G.g[2].data(), (int) G.g[2].size(), G.g[3].data(), (int) G.g[3].size(), first, 1, count);
}

int rt_hip_image_loop(int device, int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                      int method, const rt_ray *rays, size_t n_rays, double scale, double *image,
                      double *I_ang, unsigned int *failure_code, rt_ray *failed_rays, int max_failed,
                      int *n_failed, rt_stats *stats)
{
    if (!image || !I_ang)
        return fail_arg("rt_hip_image_loop: NULL output");
    if (!rays && n_rays)
        return fail_arg("rt_hip_image_loop: NULL ray list");
    if (n_rays > MAX_LIST_RAYS)
        return fail_arg("rt_hip_image_loop: 2^32 - 512 rays or more: split the call");
    // RT_HIP_TIMING=1: wall-clock split of this call on stderr (diagnostic)
    static const bool timing = getenv("RT_HIP_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap    = [&](const char *what) {
        if (timing) {
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "  rt_hip_image_loop %-22s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
            t_prev = now;
        }
    };
    rt_hip_plan *p = nullptr;
    int rc         = rt_hip_plan_create(&p, device, N, beam, gain, seed, method, scale);
    if (rc != RT_OK)
        return rc;
    lap("plan_create");
    hipStream_t q = lease_queue(device);
    // A list that is a whole tensor grid (what create_image builds) is not uploaded: the device
    // generates the rays while host threads check the list against the grid, ray by ray.
    GridGuess G;
    bool as_grid = n_rays >= (1u << 16) && !getenv("RT_HIP_NO_GRID_DETECT") && guess_ray_grid(rays, n_rays, G);
    lap("guess grid");
    if (as_grid) {
        rc = plan_set_guessed_grid(p, G, 0, (int64_t) n_rays);
        lap("set_ray_grid");
        if (rc == RT_OK)
            rc = rt_hip_plan_run(p, q, nullptr, nullptr); // asynchronous
        lap("run (launch)");
        if (rc == RT_OK && !verify_ray_grid(rays, n_rays, G, host_threads(16))) {
            as_grid = false; // not that grid after all: the speculative result is discarded below
            plan_quiesce(p);
        }
        lap("verify list");
    }
    if (rc == RT_OK && !as_grid) {
        rc = plan_set_rays_deferred(p, rays, n_rays);
        if (rc == RT_OK)
            rc = rt_hip_plan_run(p, q, nullptr, nullptr);
    }
    if (rc == RT_OK)
        rc = rt_hip_plan_fetch(p, image, I_ang, failure_code, failed_rays, max_failed, n_failed, stats);
    lap("fetch (wait + D2H)");
    rt_hip_plan_destroy(p); // waits for whatever is still in flight
    release_queue(device, q);
    lap("destroy");
    return rc;
}

// ---- all devices of the node --------------------------------------------------------------------
namespace {

// librccl.so is half a gigabyte: it is loaded when the multi-device entry is first used, not with
// this library
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort     = nullptr;
    decltype(&ncclGroupStart) GroupStart   = nullptr;
    decltype(&ncclGroupEnd) GroupEnd       = nullptr;
    decltype(&ncclSend) Send               = nullptr;
    decltype(&ncclRecv) Recv               = nullptr;
    decltype(&ncclReduce) Reduce           = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle)
                break;
        }
        if (!api.handle) {
            api.error = std::string("librccl.so not found: ") + dlerror();
            return;
        }
#define RCCL_SYM(field, name)                                                      \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));    \
    if (!api.field)                                                                \
        api.error = std::string("librccl.so lacks ") + name;
        RCCL_SYM(CommInitAll, "ncclCommInitAll")
        RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(CommAbort, "ncclCommAbort")
        RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RCCL_SYM(Send, "ncclSend")
        RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(Reduce, "ncclReduce")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    });
    return &api;
}

// one communicator over devices 0 .. ndev-1, kept across calls (creating it costs ~0.1-1 s)
std::mutex g_multi_mutex; // one multi-device call at a time per process
std::vector<ncclComm_t> g_comms;
std::atomic<bool> g_comms_aborted(false); // multi_abort_all() has released them
thread_local int g_multi_mode = 0;

int multi_comms(int ndev, std::string &err)
{
    RcclApi *R = rccl_api();
    if (!R->error.empty()) {
        err = R->error;
        return RT_ERR_NO_DEVICE;
    }
    if (g_comms_aborted.load()) { // the last call aborted them (multi_abort_all): already released
        g_comms.clear();
        g_comms_aborted.store(false);
    }
    if ((int) g_comms.size() == ndev)
        return RT_OK;
    for (auto c : g_comms)
        (void) R->CommDestroy(c);
    g_comms.assign((size_t) ndev, nullptr);
    std::vector<int> devs((size_t) ndev);
    for (int d = 0; d < ndev; d++)
        devs[(size_t) d] = d;
    ncclResult_t r = R->CommInitAll(g_comms.data(), ndev, devs.data());
    if (r != ncclSuccess) {
        err = std::string("ncclCommInitAll: ") + R->GetErrorString(r);
        g_comms.clear();
        return RT_ERR_HIP;
    }
    return RT_OK;
}

// A worker whose part of the collective failed after the rendezvous (an RCCL error, a faulted queue, a peer that
// never showed up within the time limit) aborts EVERY communicator: ncclCommAbort ends the kernels of a pending
// collective, so the peers' queues drain and their workers come back instead of waiting for a partner that will
// never arrive.  The communicators are gone afterwards; the next call builds new ones.
void multi_abort_all()
{
    bool expected = false;
    if (!g_comms_aborted.compare_exchange_strong(expected, true))
        return; // once
    RcclApi *R = rccl_api();
    for (auto c : g_comms)
        if (c)
            (void) R->CommAbort(c);
}

// all workers arrive, or nobody passes: keeps a failed worker from leaving the others in a collective
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int n, arrived = 0, phase = 0;
    bool failed = false;
    explicit Rendezvous(int n_) : n(n_) {}
    bool arrive(bool ok) // returns true if every worker of this phase was ok
    {
        std::unique_lock<std::mutex> lk(mu);
        failed = failed || !ok;
        const int my = phase;
        if (++arrived == n) {
            arrived = 0;
            phase++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return phase != my; });
        }
        return !failed;
    }
};

inline int tile_cols(int nx, int d, int ndev) { return d < nx ? (nx - d + ndev - 1) / ndev : 0; }

} // namespace

namespace rt {
// device 0 after the gather: recv = [ndev][stride] with stride = n_tile_max + n_ang doubles, part r =
// tile [ny][cols(r)][K] of image columns r, r + ndev, ... followed (at n_tile_max) by its I_ang sums
extern "C" __global__ void __launch_bounds__(256) rt_interleave_kernel(const double *recv, unsigned long long stride, int ndev,
                                                                      int nx, int ny, int K, unsigned long long n_tile_max,
                                                                      int n_ang, double *image, double *iang)
{
    const unsigned long long n_img = (unsigned long long) nx * (unsigned long long) ny * (unsigned long long) K;
    const unsigned long long step  = (unsigned long long) gridDim.x * blockDim.x;
    for (unsigned long long t = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; t < n_img; t += step) {
        const unsigned k   = (unsigned) (t % (unsigned) K);
        const unsigned long long pix = t / (unsigned) K;
        const unsigned i = (unsigned) (pix % (unsigned) nx), j = (unsigned) (pix / (unsigned) nx);
        const unsigned r = i % (unsigned) ndev, c = i / (unsigned) ndev;
        const unsigned cols = r < (unsigned) nx ? ((unsigned) nx - r + (unsigned) ndev - 1) / (unsigned) ndev : 0;
        image[t] = recv[r * stride + ((unsigned long long) j * cols + c) * (unsigned) K + k];
    }
    for (unsigned long long t = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; t < (unsigned long long) n_ang; t += step) {
        double v = 0.0;
        for (int r = 0; r < ndev; r++)
            v += recv[(unsigned long long) r * stride + n_tile_max + t];
        iang[t] = v;
    }
}
} // namespace rt

namespace rt {
// loopback rehearsal of the sum-reduce (rt_hip_multi_image_loop): out = sum over parts of recv[part][.]
extern "C" __global__ void __launch_bounds__(256) rt_sum_parts_kernel(const double *recv, unsigned long long stride, int ndev,
                                                                     unsigned long long n, double *out)
{
    const unsigned long long step = (unsigned long long) gridDim.x * blockDim.x;
    for (unsigned long long t = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; t < n; t += step) {
        double v = 0.0;
        for (int r = 0; r < ndev; r++)
            v += recv[(unsigned long long) r * stride + t];
        out[t] = v;
    }
}
} // namespace rt

int rt_hip_multi_last_mode(void) { return g_multi_mode; }

int rt_hip_host_libm_mode(int device) { return tan_mode(device); }

int rt_hip_ray_list_grid_dims(const rt_ray *rays, size_t n_rays, int dims[4])
{
    GridGuess G;
    if (!rays || !dims || !guess_ray_grid(rays, n_rays, G) || !verify_ray_grid(rays, n_rays, G, host_threads(16)))
        return 0;
    for (int i = 0; i < 4; i++)
        dims[i] = (int) G.g[i].size();
    return 1;
}

void rt_hip_pool_trim(void) { pool_trim_all(); }

int rt_hip_multi_image_loop(int ndev, int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                            int method, const rt_ray *rays, size_t n_rays, double scale, double *image,
                            double *I_ang, unsigned int *failure_code, rt_ray *failed_rays, int max_failed,
                            int *n_failed, rt_stats *stats)
{
    if (!beam || !gain || !image || !I_ang)
        return fail_arg("rt_hip_multi_image_loop: NULL argument");
    if (!rays && n_rays)
        return fail_arg("rt_hip_multi_image_loop: NULL ray list");
    const int have = rt_hip_device_count();
    if (have < 1) {
        g_last_error = "no HIP device";
        return RT_ERR_NO_DEVICE;
    }
    if (ndev <= 0 || ndev > have)
        ndev = have;
    // RT_HIP_MULTI_LOOPBACK=n: rehearsal of an n-device run on ONE device (tests): n workers, n plans, the
    // same partition, buffers and assembly kernels, all on device 0, with the RCCL collective replaced by
    // device-to-device copies into the same receive layout.  Exercises everything of the N > 1 path except
    // the RCCL calls themselves.
    const int loopback = (int) env_unsigned("RT_HIP_MULTI_LOOPBACK", 0, 1, 16);
    if (loopback > 0)
        ndev = loopback;
    auto dev_of = [&](int d) { return loopback > 0 ? 0 : d; };
    const int inject_fail      = getenv("RT_HIP_MULTI_INJECT_FAIL") ? atoi(getenv("RT_HIP_MULTI_INJECT_FAIL")) : -1;
    const unsigned timeout_ms  = env_unsigned("RT_HIP_MULTI_TIMEOUT_MS", 120000, 1, 3600000);
    const auto t_begin = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> serial(g_multi_mutex);
    RcclApi *R = nullptr;
    if (loopback == 0) {
        std::string err;
        const int rc = multi_comms(ndev, err);
        if (rc != RT_OK) {
            g_last_error = "rt_hip_multi_image_loop: " + err;
            return rc;
        }
        R = rccl_api();
    }

    // ---- how to partition ----------------------------------------------------------------------
    GridGuess G;
    const bool is_grid = n_rays >= 1 && !getenv("RT_HIP_NO_GRID_DETECT") && guess_ray_grid(rays, n_rays, G) &&
                         verify_ray_grid(rays, n_rays, G, host_threads(16));
    auto axis_is = [](const std::vector<double> &g, const double *b, int n) {
        if ((int) g.size() != n)
            return false;
        for (int i = 0; i < n; i++)
            if (!same_bits((float) g[(size_t) i], (float) b[i]))
                return false;
        return true;
    };
    // pixel tiles: ASE, and the rays are the beam's own grid -- every ray then deposits into the pixel
    // column it starts in (SURVEY.md 8(c) i; the frequency kernel computes the deposit cell per ray anyway)
    const bool tiles = method == 1 && !seed && is_grid && axis_is(G.g[0], beam->x, beam->nx) &&
                       axis_is(G.g[1], beam->y, beam->ny) && axis_is(G.g[2], beam->a, beam->na) &&
                       axis_is(G.g[3], beam->b, beam->nb) && !getenv("RT_HIP_MULTI_NO_TILES") &&
                       // ... and column i of the rays deposits into pixel column i of the FULL grid (a tile plan
                       // runs the deposit index on its own sub-grid with the original dx): the same host check
                       // that allows the exclusive mode; a beam that fails it takes the chunk mode
                       grid_points_in_own_cells(beam->x, beam->nx, beam->dx) && grid_points_in_own_cells(beam->y, beam->ny, beam->dy);
    g_multi_mode = tiles ? 1 : 2;

    const int nx = beam->nx, ny = beam->ny, K = beam->nv;
    const size_t n_ang = (size_t) beam->na * (size_t) beam->nb;
    const size_t n_img = (size_t) nx * (size_t) ny * (size_t) K;
    const size_t n_tile_max = tiles ? (size_t) ny * (size_t) tile_cols(nx, 0, ndev) * (size_t) K : n_img;
    const size_t stride     = n_tile_max + n_ang; // doubles every device contributes

    struct Worker {
        int rc = RT_OK;
        std::string error;
        unsigned code = 0;
        rt_ray failed[RT_N_FAILED_MAX];
        int n_failed = 0;
        rt_stats st  = {};
    };
    std::vector<Worker> W((size_t) ndev);
    Rendezvous meet(ndev);
    double *recv0 = nullptr, *out0 = nullptr; // device 0: gathered parts / assembled (image | I_ang)

    auto work = [&](int d) {
        Worker &w       = W[(size_t) d];
        rt_hip_plan *p  = nullptr;
        double *buf     = nullptr;
        hipStream_t q   = nullptr;
        auto fail       = [&](int rc, const std::string &what) {
            if (w.rc == RT_OK) {
                w.rc    = rc;
                w.error = what;
            }
        };
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e != hipSuccess)
                fail(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
            return e == hipSuccess;
        };
        // -- plan on this device (the device is bound here, inside the worker)
        const int pd = dev_of(d); // the physical device of this worker
        if (hip_ok(hipSetDevice(pd), "hipSetDevice")) {
            q = lease_queue(pd);
            if (!q)
                fail(RT_ERR_HIP, "no queue");
        }
        std::vector<double> xd;
        if (w.rc == RT_OK) {
            int rc;
            if (tiles) {
                rt_beam bd = *beam;
                for (int i = d; i < nx; i += ndev)
                    xd.push_back(beam->x[i]);
                bd.nx = (int) xd.size();
                bd.x  = xd.data();
                if (bd.nx == 0) { // more devices than columns: an empty tile
                    xd.push_back(beam->x[0]);
                    bd.nx = 1;
                    bd.x  = xd.data();
                }
                rc = rt_hip_plan_create(&p, pd, N, &bd, gain, seed, method, scale);
                if (rc == RT_OK) {
                    const int cols      = tile_cols(nx, d, ndev);
                    const int64_t count = (int64_t) cols * ny * beam->na * beam->nb;
                    rc = rt_hip_plan_set_ray_grid(p, xd.data(), (int) xd.size(), beam->y, ny, beam->a, beam->na, beam->b,
                                                  beam->nb, 0, 1, count);
                }
            } else {
                rc = rt_hip_plan_create(&p, pd, N, beam, gain, seed, method, scale);
                // contiguous ray chunks, as RayTraceImageThreadLoop splits them (RayTraceImage.cpp:107)
                const size_t chunk = n_rays / (size_t) ndev + 1;
                const size_t begin = std::min((size_t) d * chunk, n_rays);
                const size_t count = std::min(chunk, n_rays - begin);
                if (rc == RT_OK)
                    rc = is_grid ? plan_set_guessed_grid(p, G, (int64_t) begin, (int64_t) count)
                                 : rt_hip_plan_set_rays(p, count ? rays + begin : nullptr, count);
            }
            if (rc != RT_OK)
                fail(rc, rt_hip_last_error());
        }
        if (w.rc == RT_OK) {
            hip_ok(pool_alloc(pd, (void **) &buf, stride * sizeof(double)), "device buffer");
            if (w.rc == RT_OK && d == 0) {
                if (tiles || loopback > 0)
                    hip_ok(pool_alloc(0, (void **) &recv0, (size_t) ndev * stride * sizeof(double)), "gather buffer");
                hip_ok(pool_alloc(0, (void **) &out0, (n_img + n_ang) * sizeof(double)), "image buffer");
            }
        }
        if (w.rc == RT_OK && tiles && stride > (size_t) p->n_image + n_ang) // padding of a narrower tile travels too
            hip_ok(hipMemsetAsync(buf, 0, stride * sizeof(double), q), "hipMemsetAsync");
        if (w.rc == RT_OK) {
            const int rc = rt_hip_plan_run(p, q, buf, buf + n_tile_max);
            if (rc != RT_OK)
                fail(rc, rt_hip_last_error());
        }
        // counters and failure report of this device; a run with failing rays repeats its frequency pass
        // here (rt_hip_plan_fetch), before its result travels
        if (w.rc == RT_OK) {
            const int rc = rt_hip_plan_fetch(p, nullptr, nullptr, &w.code, w.failed, RT_N_FAILED_MAX, &w.n_failed, &w.st);
            if (rc != RT_OK)
                fail(rc, rt_hip_last_error());
        }
        // -- the one collective of the image, on the queue the kernels ran on
        if (loopback > 0) {
            // rehearsal: every worker copies its part into the receive layout, worker 0 assembles
            const bool all_ok = meet.arrive(w.rc == RT_OK);
            if (all_ok && inject_fail == d)
                fail(RT_ERR_HIP, "injected failure after the rendezvous (RT_HIP_MULTI_INJECT_FAIL)");
            if (all_ok && w.rc == RT_OK) {
                hip_ok(hipMemcpyAsync(recv0 + (size_t) d * stride, buf, stride * sizeof(double), hipMemcpyDeviceToDevice, q),
                       "loopback copy");
                hip_ok(hipStreamSynchronize(q), "hipStreamSynchronize");
            }
            if (meet.arrive(w.rc == RT_OK) && d == 0) {
                const unsigned long long n_out = (unsigned long long) (tiles ? n_img : stride);
                unsigned blocks = (unsigned) std::min<unsigned long long>((n_out + 255) / 256, 256ull * 64ull);
                blocks          = blocks ? blocks : 1;
                if (tiles)
                    hipLaunchKernelGGL(rt::rt_interleave_kernel, dim3(blocks), dim3(256), 0, q, recv0, (unsigned long long) stride,
                                       ndev, nx, ny, K, (unsigned long long) n_tile_max, (int) n_ang, out0, out0 + n_img);
                else
                    hipLaunchKernelGGL(rt::rt_sum_parts_kernel, dim3(blocks), dim3(256), 0, q, recv0, (unsigned long long) stride,
                                       ndev, (unsigned long long) stride, out0);
                hip_ok(hipGetLastError(), "assembly kernel");
                hip_ok(hipStreamSynchronize(q), "hipStreamSynchronize");
                if (w.rc == RT_OK) {
                    hip_ok(hipMemcpy(image, out0, n_img * sizeof(double), hipMemcpyDeviceToHost), "download image");
                    hip_ok(hipMemcpy(I_ang, out0 + n_img, n_ang * sizeof(double), hipMemcpyDeviceToHost), "download I_ang");
                }
            }
        } else if (meet.arrive(w.rc == RT_OK)) {
            ncclResult_t r = ncclSuccess;
            // (test hook: RT_HIP_MULTI_INJECT_FAIL=d makes worker d fail here, after the rendezvous, without
            // entering the collective -- what a faulted queue or a failed enqueue looks like to its peers)
            const bool injected = inject_fail == d;
            if (injected) {
                fail(RT_ERR_HIP, "injected failure after the rendezvous (RT_HIP_MULTI_INJECT_FAIL)");
            } else if (tiles) {
                r = R->GroupStart();
                if (r == ncclSuccess)
                    r = R->Send(buf, stride, ncclDouble, 0, g_comms[(size_t) d], q);
                for (int src = 0; d == 0 && src < ndev && r == ncclSuccess; src++)
                    r = R->Recv(recv0 + (size_t) src * stride, stride, ncclDouble, src, g_comms[0], q);
                const ncclResult_t e = R->GroupEnd();
                r                    = r == ncclSuccess ? e : r;
                if (r == ncclSuccess && d == 0) {
                    const unsigned long long n_out = (unsigned long long) n_img;
                    unsigned blocks = (unsigned) std::min<unsigned long long>((n_out + 255) / 256, 256ull * 64ull);
                    blocks          = blocks ? blocks : 1;
                    hipLaunchKernelGGL(rt::rt_interleave_kernel, dim3(blocks), dim3(256), 0, q, recv0, (unsigned long long) stride,
                                       ndev, nx, ny, K, (unsigned long long) n_tile_max, (int) n_ang, out0, out0 + n_img);
                    hip_ok(hipGetLastError(), "rt_interleave_kernel");
                }
            } else {
                r = R->Reduce(buf, d == 0 ? out0 : nullptr, stride, ncclDouble, ncclSum, 0, g_comms[(size_t) d], q);
            }
            if (r != ncclSuccess)
                fail(RT_ERR_HIP, std::string("RCCL: ") + R->GetErrorString(r));
            if (w.rc != RT_OK) {
                multi_abort_all(); // the peers must not wait for this worker's part
            } else {
                // wait for the collective -- but not for ever: a peer that failed aborts the communicators
                // (its own abort ends this queue's kernels), and a peer that never arrives is given
                // RT_HIP_MULTI_TIMEOUT_MS (default 120 s) before this worker aborts them itself
                const auto t_wait = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t e = hipStreamQuery(q);
                    if (e == hipSuccess)
                        break;
                    if (e != hipErrorNotReady) {
                        hip_ok(e, "collective");
                        multi_abort_all();
                        break;
                    }
                    if (g_comms_aborted.load()) {
                        fail(RT_ERR_HIP, "collective aborted: another device failed");
                        break;
                    }
                    const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
                    if (waited > (double) timeout_ms) {
                        fail(RT_ERR_HIP, "collective timed out after " + std::to_string(timeout_ms) + " ms");
                        multi_abort_all();
                        break;
                    }
                    if (waited > 0.2) // spin for the first 200 us (the stand-in's gather takes ~50), then yield
                        std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                if (w.rc == RT_OK && g_comms_aborted.load())
                    fail(RT_ERR_HIP, "collective aborted: another device failed");
            }
            if (w.rc == RT_OK && d == 0) {
                hip_ok(hipMemcpy(image, out0, n_img * sizeof(double), hipMemcpyDeviceToHost), "download image");
                hip_ok(hipMemcpy(I_ang, out0 + n_img, n_ang * sizeof(double), hipMemcpyDeviceToHost), "download I_ang");
            }
        }
        // every worker is past the collective before any buffer of it goes back to the pool
        meet.arrive(true);
        rt_hip_plan_destroy(p);
        (void) hipSetDevice(pd);
        if (q)
            (void) hipStreamSynchronize(q);
        pool_free(pd, buf);
        if (d == 0) {
            pool_free(0, recv0);
            pool_free(0, out0);
        }
        release_queue(pd, q);
    };
    std::vector<std::thread> th;
    for (int d = 0; d < ndev; d++)
        th.emplace_back(work, d);
    for (auto &t : th) // join EVERY worker, then report the first error
        t.join();
    // (a worker that was pulled out of the collective by another one's abort is not the one to quote)
    for (int pass = 0; pass < 2; pass++)
        for (int d = 0; d < ndev; d++)
            if (W[(size_t) d].rc != RT_OK && (pass == 1 || W[(size_t) d].error.rfind("collective aborted", 0) != 0)) {
                g_last_error = "device " + std::to_string(d) + ": " + W[(size_t) d].error;
                return W[(size_t) d].rc;
            }
    unsigned code = 0;
    int nf        = 0;
    rt_stats tot  = {};
    for (int d = 0; d < ndev; d++) {
        const Worker &w = W[(size_t) d];
        code |= w.code;
        for (int i = 0; i < w.n_failed && failed_rays && nf < max_failed && nf < RT_N_FAILED_MAX; i++)
            failed_rays[nf++] = w.failed[i];
        tot.n_rays += w.st.n_rays;
        tot.cell_steps += w.st.cell_steps;
        tot.n_escaped += w.st.n_escaped;
        tot.n_skipped += w.st.n_skipped;
        tot.kernel_ms = std::max(tot.kernel_ms, w.st.kernel_ms);
        tot.march_ms  = std::max(tot.march_ms, w.st.march_ms);
        tot.freq_ms   = std::max(tot.freq_ms, w.st.freq_ms);
    }
    tot.total_ms = (float) std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (failure_code)
        *failure_code = code;
    if (n_failed)
        *n_failed = nf;
    if (stats)
        *stats = tot;
    return RT_OK;
}

} // extern "C"
