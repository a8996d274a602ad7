// rt_runtime.h -- what the host-side translation units of librt_hip.so share (internal, not part of the C ABI).
//
// The C ABI of include/rt_hip.h is implemented by five translation units:
//   rt_pool.hip     error text, device-memory pool, leased queues, environment overrides
//   rt_plan.hip     the plan: arena packing, ray list / ray grid, run, fetch, probe and path outputs,
//                   rt_hip_image_loop (the host-pointer entry the C++ adapter calls)
//   rt_raygrid.hip  a ray list that is really a tensor grid: recognition + bit-wise verification,
//                   list-mode launch tangents and the probe of the host's libm
//   rt_launch.hip   the kernels (rt_march.hip, rt_freq.hip, rt_path.hip) and how a run puts them on a queue
//   rt_multi.hip    all devices of the node: RCCL loader, communicator, rt_hip_multi_image_loop
// Only rt_launch.hip and rt_multi.hip contain device code.
#pragma once

#include "rt_device.h"

#include <chrono>
#include <string>
#include <vector>

// One prepared problem on one device (opaque in include/rt_hip.h).
struct rt_hip_plan {
    int device         = 0;
    int cu_count       = 0;
    rt::DevParams P    = {};
    unsigned char *arena = nullptr;
    size_t arena_bytes = 0;
    void *staging        = nullptr; // page-locked source of an arena upload still in flight on upload_q (plan_create_on)
    hipStream_t upload_q = nullptr;
    unsigned char *out_staging = nullptr; // page-locked copy of the last run's outputs, queued behind its kernels (plan_stage_outputs)
    bool out_staged            = false;
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
    // the queue a run has put work on, set BEFORE its first enqueue: a run that fails midway (a launch error, an
    // allocation) leaves `ran` as it was, and plan_quiesce still has to wait for what was already queued
    hipStream_t queued_stream = nullptr;
    bool queued = false;
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
    bool gv_has_nan       = false;     // host scan of the lineshape tables (emission mode): a NaN or an infinity
    // the refractive-index tables and the segment length lie in the ranges under which the march's divisions
    // need no scaling (rt_march.hip, template parameter BOUNDED); checked by rt_hip_plan_create
    bool tables_bounded   = false;
    // links of the fused kernel's work-group tile lists (rt_fused.hip), one word per 64-ray tile; last run fused?
    unsigned *tile_next   = nullptr;
    size_t tile_next_n    = 0;
    bool last_fused       = false;
    // LDS a work-group may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KB on gfx950)
    size_t lds_limit      = 0;
};

namespace rtr {

// ---- rt_pool.hip -------------------------------------------------------------------------------------
// text of the last failure on this thread (rt_hip_last_error)
std::string &last_error();
int fail_hip(hipError_t e, const char *what, const char *file, int line);
int fail_arg(const char *msg);

#define HIP_TRY(expr)                                                \
    do {                                                             \
        hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess)                                        \
            return rtr::fail_hip(e_, #expr, __FILE__, __LINE__);     \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Device-memory pool: allocations (never data) kept across calls, per device.
hipError_t pool_alloc(int device, void **out, size_t bytes);
void pool_free(int device, void *ptr);
void pool_trim_all();
// hipMalloc for the large one-off allocations: out of memory while the pool still parks blocks -> trim and retry
hipError_t dev_malloc(void **out, size_t bytes);
// page-locked host staging for uploads that run beside host work (parked like device blocks; rt_pool.hip)
hipError_t pinned_alloc(void **out, size_t bytes);
void pinned_free(void *ptr);

// tuning overrides from the environment: a missing, non-numeric or non-positive value keeps the default
unsigned env_unsigned(const char *name, unsigned def, unsigned lo, unsigned hi);
// host threads for the list verification and the host tangents (RT_HIP_HOST_THREADS), at most `cap`
unsigned host_threads(unsigned cap);

// non-blocking queues of a device, leased per call and kept across calls
hipStream_t lease_queue(int device);
void release_queue(int device, hipStream_t q);

// ---- rt_plan.hip -------------------------------------------------------------------------------------
// wait for the plan's last run before any of its buffers is freed or parked
void plan_quiesce(rt_hip_plan *p);
// the list stays on the host until the run, which uploads it in slices beside the march
int plan_set_rays_deferred(rt_hip_plan *p, const rt_ray *rays, size_t n_rays);
// after rt_hip_plan_run: queues the download of control block, I_ang and image (if they are small) behind the kernels, into
// page-locked staging, so that rt_hip_plan_fetch finds them on the host when the queue has drained
void plan_stage_outputs(rt_hip_plan *p);
// rt_hip_plan_create with the table upload queued on `upload_q` from page-locked staging instead of waited for (nullptr:
// what rt_hip_plan_create does).  Everything that reads the tables must then run on that queue, or after a wait for it.
int plan_create_on(rt_hip_plan **out, hipStream_t upload_q, int device, int N, const rt_beam *beam, const rt_gain *gain,
                   const rt_seed *seed, int method, double scale);
// most rays a list may hold (the kernels index rays with 32 bits)
extern const size_t MAX_LIST_RAYS;

// ---- rt_raygrid.hip ----------------------------------------------------------------------------------
struct GridGuess {
    std::vector<double> g[4]; // x, y, a, b as the doubles of the floats the rays carry
};
inline bool same_bits(float a, float b) { return __builtin_memcmp(&a, &b, sizeof(float)) == 0; }
bool guess_ray_grid(const rt_ray *rays, size_t n, GridGuess &G);
bool verify_ray_grid(const rt_ray *rays, size_t n, const GridGuess &G, unsigned threads);
int plan_set_guessed_grid(rt_hip_plan *p, const GridGuess &G, int64_t first, int64_t count);
// RayTraceImageCPU.cpp:11-16 on the host: grid point i (as the float a ray carries) falls into deposit cell i
bool grid_points_in_own_cells(const double *g, int n, double d);
// x / d for every x < 2^31 as mulhi(x, mul) >> sh (DevRays::div_mul)
void magic_u31(unsigned d, unsigned &mul, unsigned &sh);
// Helper.h:409-410 on host threads, with the host's tanf
void host_tangents(const rt_ray *rays, size_t n, float *sxy);
// 1: the device restatement of tanf equals this host's tanf; 2: list-mode tangents come from the host
int tan_mode(int device);

// ---- rt_launch.hip -----------------------------------------------------------------------------------
// march -> records -> frequency pass (or the path tracer) on `stream`; records ev0 / evm / ev1 of the plan
int plan_launch_run(rt_hip_plan *p, hipStream_t stream);
// a run that reported failing rays: repeat the frequency pass without them
int plan_repeat_checked(rt_hip_plan *p);
int launch_tan(const rt_ray *rays_dev, unsigned long long n, float *sxy_dev, hipStream_t stream);
int launch_seed_tab(const rt::DevSeed &sd, const rt::DevRays &R, size_t n_points, double *sf, unsigned char *sin);
int launch_selftest(unsigned long long *counts_dev);
// zero up to three device ranges (8-byte multiples; NULL = none) with one launch
int launch_zero3(hipStream_t stream, void *a, size_t a_bytes, void *b, size_t b_bytes, void *c, size_t c_bytes);

} // namespace rtr
