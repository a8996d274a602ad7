// rt_plan.hip -- the plan: one prepared problem on one device, and the host-pointer entry built on it.
//
// Replaces the host side of RayTraceImageCudaLoop (src/RayTraceImageCuda.cu:145-221) and of the
// copy_device helpers (src/RayTraceImageCuda.cu:224-329): where those issue ~30 cudaMalloc/cudaMemcpy
// calls per create_image, a plan packs every table into ONE arena, uploads it with ONE copy, zeroes
// outputs + control block and has rt_launch.hip put the march kernel and the frequency kernel back to back
// on one queue.  No data is cached across calls (Readme.txt:43); freed device allocations and the queues
// of a device are (rt_pool.hip).  Host code only.
#include "rt_runtime.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

using namespace rtr;

namespace {

// Bump allocator over the host image of a plan's arena: one buffer of a capacity computed up front (an upper bound, so nothing moves while the
// tables are copied in), page-locked when the upload is to run beside host work (rt_hip_image_loop), plain otherwise.
struct ArenaBuilder {
    unsigned char *base = nullptr;
    size_t cap = 0, used = 0;
    bool pinned = false, overflow = false;
    std::unique_ptr<unsigned char[]> own;
    ArenaBuilder() = default;
    ArenaBuilder(const ArenaBuilder &) = delete;
    ArenaBuilder &operator=(const ArenaBuilder &) = delete;
    ~ArenaBuilder()
    {
        if (pinned)
            rtr::pinned_free(base);
    }
    void start(size_t capacity, bool want_pinned)
    {
        cap = capacity;
        void *h = nullptr;
        if (want_pinned && rtr::pinned_alloc(&h, capacity) == hipSuccess) {
            base   = static_cast<unsigned char *>(h);
            pinned = true;
        } else {
            own.reset(new unsigned char[capacity]);
            base = own.get();
        }
    }
    // hands the page-locked buffer to the caller (who frees it with pinned_free once the upload has completed)
    void *release()
    {
        pinned = false;
        return base;
    }
    size_t reserve(size_t bytes) // zero-filled
    {
        const size_t off = rtr::align_up(used, 256);
        if (off + bytes > cap) {
            overflow = true;
            return 0;
        }
        memset(base + off, 0, bytes);
        used = off + bytes;
        return off;
    }
    size_t put(const void *src, size_t bytes)
    {
        const size_t off = rtr::align_up(used, 256);
        if (off + bytes > cap) {
            overflow = true;
            return 0;
        }
        if (bytes && src)
            memcpy(base + off, src, bytes);
        used = off + bytes;
        return off;
    }
};

// largest magnitude of n floats as bit patterns (non-negative floats order like their bit patterns): `finite` over the
// values below infinity, `all` over everything.  One pass that the compiler vectorises; the AVX2 instance is
// taken when the CPU has it (a plan of the shipped size scans 286 000 values on the critical path of every call).
template <int>
inline __attribute__((always_inline)) void magnitude_scan(const uint32_t *u, size_t n, uint32_t &finite, uint32_t &all)
{
    uint32_t m = 0, mall = 0;
    for (size_t c = 0; c < n; c++) {
        uint32_t a = u[c] & 0x7fffffffu;
        mall       = a > mall ? a : mall;
        a          = a < 0x7f800000u ? a : 0u; // inf and NaN do not count
        m          = a > m ? a : m;
    }
    finite = m;
    all    = mall;
}
void magnitude_scan_plain(const uint32_t *u, size_t n, uint32_t &finite, uint32_t &all) { magnitude_scan<0>(u, n, finite, all); }
__attribute__((target("avx2"))) void magnitude_scan_avx2(const uint32_t *u, size_t n, uint32_t &finite, uint32_t &all)
{
    magnitude_scan<1>(u, n, finite, all);
}

} // namespace

// The kernels index rays with 32 bits; the march hands rays out in chunks of at most 4096 (the cap of
// RT_HIP_MARCH_CHUNK) and computes (rays of the launch + chunk - 1) / chunk in 32 bits.
const size_t rtr::MAX_LIST_RAYS = 0xffffffffull - 4096;

// Wait for the work of the plan's last run before any of its buffers is freed or parked in the pool:
// another plan may be handed a parked block at once (pool_alloc) and overwrite it.
void rtr::plan_quiesce(rt_hip_plan *p)
{
    if (p && p->ran) {
        if (hipStreamSynchronize(p->last_stream) != hipSuccess)
            (void) hipGetLastError(); // a caller's stream that is gone: nothing is in flight on it
    }
    // a run that failed after its first enqueue (rt_hip_plan_run): zeroing kernel, memsets or march slices may still be
    // writing the plan's blocks on that queue
    if (p && p->queued && !(p->ran && p->queued_stream == p->last_stream)) {
        if (hipStreamSynchronize(p->queued_stream) != hipSuccess)
            (void) hipGetLastError();
    }
}

extern "C" {

int rt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *rt_hip_last_error(void) { return last_error().c_str(); }

int rt_hip_selftest(int device, unsigned long long *n_checked, unsigned long long *n_mismatch)
{
    if (!n_checked || !n_mismatch)
        return fail_arg("rt_hip_selftest: NULL argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr, h[2] = { 0, 0 };
    HIP_TRY(hipMalloc((void **) &d, sizeof(h)));
    hipError_t e = hipMemset(d, 0, sizeof(h));
    if (e == hipSuccess && launch_selftest(d) != RT_OK)
        e = hipErrorLaunchFailure;
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
    if (p->staging) {
        if (hipStreamSynchronize(p->upload_q) != hipSuccess)
            (void) hipGetLastError();
        pinned_free(p->staging);
        p->staging = nullptr;
    }
    pinned_free(p->out_staging);
    p->out_staging = nullptr;
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
    pool_free(p->device, p->tile_next);
    (void) hipFree(p->probe);
    (void) hipFree(p->bad_dev);
    delete p;
}

int rt_hip_plan_create(rt_hip_plan **out, int device, int N, const rt_beam *beam, const rt_gain *gain,
                       const rt_seed *seed, int method, double scale)
{
    return rtr::plan_create_on(out, nullptr, device, N, beam, gain, seed, method, scale);
}

} // extern "C"

int rtr::plan_create_on(rt_hip_plan **out, hipStream_t upload_q, int device, int N, const rt_beam *beam, const rt_gain *gain,
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
        last_error() = "no HIP device";
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev)
        return fail_arg("rt_hip_plan_create: bad device index");
    HIP_TRY(hipSetDevice(device));

    rt_hip_plan *p = new rt_hip_plan();
    p->t_created   = std::chrono::steady_clock::now();
    p->device      = device;
    {
        // the CU count and the LDS a work-group may ask for do not change: asked once per device
        // (hipGetDeviceProperties costs ~0.3 ms a call)
        static std::mutex mu;
        static std::vector<std::pair<int, int>> known; // per device: {CUs, LDS bytes per work-group}
        std::lock_guard<std::mutex> lock(mu);
        if ((size_t) device >= known.size())
            known.resize((size_t) device + 1, { 0, 0 });
        if (known[(size_t) device].first == 0) {
            int n = 0, lds = 0;
            hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
            if (e == hipSuccess)
                e = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
            if (e != hipSuccess || n < 1 || lds < 1) {
                delete p;
                return fail_hip(e, "hipDeviceGetAttribute(multiprocessor count, LDS per work-group)", __FILE__, __LINE__);
            }
            known[(size_t) device] = { n, lds };
        }
        p->cu_count  = known[(size_t) device].first;
        p->lds_limit = (size_t) known[(size_t) device].second;
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
    {
        // capacity: every piece below, each rounded up to the 256-byte alignment of its offset
        auto piece      = [](size_t bytes) { return align_up(bytes, 256) + 256; };
        const size_t kp = (size_t) ((K + 3) & ~3);
        size_t capacity = piece(sizeof(rt::DevGain) * (size_t) N), blob_bytes = align_up(sizeof(rt::BlobGain) * (size_t) N, 16);
        for (int i = 1; i < N; i++) {
            const size_t cells = (size_t) gain[i].Nx * (size_t) gain[i].Ny;
            capacity += piece(cells * kp * sizeof(float));
            blob_bytes += sizeof(rt::Interval) * ((size_t) gain[i].Nx + (size_t) gain[i].Ny) + sizeof(rt::Node) * cells;
        }
        capacity += piece(blob_bytes);
        capacity += piece(sizeof(double) * (size_t) beam->nx) + piece(sizeof(double) * (size_t) beam->ny) +
                    piece(sizeof(double) * (size_t) beam->na) + piece(sizeof(double) * (size_t) beam->nb) + 2 * piece(sizeof(double) * kp);
        if (seed)
            for (int i = 0; i < 5; i++)
                capacity += 2 * piece(sizeof(double) * ((size_t) (seed->dim[i] > 0 ? seed->dim[i] : 0) + kp));
        ab.start(capacity, upload_q != nullptr);
    }
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
            off_gv[(size_t) i] = ab.reserve(sizeof(float) * cells * (size_t) Kp); // zero-filled
            float *dst         = reinterpret_cast<float *>(ab.base + off_gv[(size_t) i]);
            for (size_t c = 0; c < cells; c++)
                memcpy(dst + c * (size_t) Kp, gain[i].gv + c * (size_t) K, sizeof(float) * (size_t) K);
        }
    }
    const size_t off_bx  = ab.put(beam->x, sizeof(double) * (size_t) beam->nx);
    const size_t off_by  = ab.put(beam->y, sizeof(double) * (size_t) beam->ny);
    const size_t off_ba  = ab.put(beam->a, sizeof(double) * (size_t) beam->na);
    const size_t off_bb  = ab.put(beam->b, sizeof(double) * (size_t) beam->nb);
    const size_t off_bdv = ab.reserve(sizeof(double) * (size_t) Kp);
    memcpy(ab.base + off_bdv, beam->dv, sizeof(double) * (size_t) beam->nv);
    const size_t off_bdv2 = ab.reserve(sizeof(double) * (size_t) Kp); // 2 * dv (exact), RayTraceImageCPU.cpp:66
    for (int k = 0; k < beam->nv; k++) {
        const double d2 = 2.0 * beam->dv[k];
        memcpy(ab.base + off_bdv2 + sizeof(double) * (size_t) k, &d2, sizeof(double));
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
                memcpy(ab.base + off_sf[i], seed->f[i], sizeof(double) * (size_t) seed->dim[i]);
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
    // (dz: a straight ray in a medium without refraction advances by up to 1250 cm per integrator step, Helper.h:288-297;
    // 1e6 cm keeps a sub-segment within a few hundred steps -- beyond it the instance with the watchdog marches)
    if (!(beam->dz >= 1e-12 && beam->dz <= 1e6))
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
            return fail_hip(e_, #expr, __FILE__, __LINE__);        \
        }                                                \
    } while (0)

    if (ab.overflow) {
        delete p;
        return fail_arg("rt_hip_plan_create: internal error, the arena outgrew its computed capacity");
    }
    p->arena_bytes = align_up(ab.used, 256);
    PLAN_TRY(pool_alloc(device, (void **) &p->arena, p->arena_bytes));
    unsigned char *A = p->arena;
    p->gv_dev.assign((size_t) N, nullptr);
    for (int i = 1; i < N; i++) {
        dg[(size_t) i].gv    = reinterpret_cast<const float *>(A + off_gv[(size_t) i]);
        p->gv_dev[(size_t) i] = dg[(size_t) i].gv;
    }
    p->dv2_dev = reinterpret_cast<const double *>(A + off_bdv2);
    memcpy(ab.base + off_gain, dg.data(), sizeof(rt::DevGain) * (size_t) N);
    if (upload_q && ab.pinned) {
        // queued, not waited for: the staging buffer stays with the plan until the queue has been waited for
        // (rt_hip_plan_destroy), and every reader of the tables runs on this queue (rt_hip_image_loop)
        const size_t bytes = ab.used;
        p->staging         = ab.release();
        p->upload_q        = upload_q;
        PLAN_TRY(hipMemcpyAsync(p->arena, p->staging, bytes, hipMemcpyHostToDevice, upload_q));
    } else {
        PLAN_TRY(hipMemcpy(p->arena, ab.base, ab.used, hipMemcpyHostToDevice));
    }
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
            uint32_t umax          = 0;
            static const bool avx2 = __builtin_cpu_supports("avx2");
            for (int i = 1; i < N; i++) {
                const size_t n = (size_t) gain[i].Nx * (size_t) gain[i].Ny * (size_t) K;
                uint32_t m = 0, mall = 0;
                (avx2 ? magnitude_scan_avx2 : magnitude_scan_plain)(reinterpret_cast<const uint32_t *>(gain[i].gv), n, m, mall);
                umax = m > umax ? m : umax;
                if (mall >= 0x7f800000u) // a NaN or an infinity: the frequency kernel then tests every value it reads
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

// rt_hip_image_loop only: the outputs of a run that are a few megabytes at most follow its kernels down the queue into
// page-locked staging -- one wait in rt_hip_plan_fetch instead of a wait and three blocking copies (larger images are
// fetched directly: copying them once more on the host would cost what the queueing saves).
void rtr::plan_stage_outputs(rt_hip_plan *p)
{
    constexpr size_t ctl_tail = offsetof(rt::DevCtl, failure_code), ctl_bytes = sizeof(rt::DevCtl) - ctl_tail;
    if (!p || !p->ran || !p->last_stream)
        return;
    const size_t off_ang = align_up(ctl_bytes, 256), off_img = off_ang + align_up(p->n_iang * sizeof(double), 256);
    const size_t total   = off_img + p->n_image * sizeof(double);
    if (total > ((size_t) 4 << 20))
        return;
    if (!p->out_staging) {
        void *h = nullptr;
        if (pinned_alloc(&h, total) != hipSuccess)
            return;
        p->out_staging = static_cast<unsigned char *>(h);
    }
    hipError_t e = hipMemcpyAsync(p->out_staging, reinterpret_cast<const unsigned char *>(p->ctl) + ctl_tail, ctl_bytes,
                                  hipMemcpyDeviceToHost, p->last_stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(p->out_staging + off_ang, p->last_iang, p->n_iang * sizeof(double), hipMemcpyDeviceToHost, p->last_stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(p->out_staging + off_img, p->last_image, p->n_image * sizeof(double), hipMemcpyDeviceToHost, p->last_stream);
    if (e != hipSuccess) {
        (void) hipGetLastError(); // fetched the ordinary way
        return;
    }
    p->out_staged = true;
}

// rt_hip_image_loop only: the list stays on the host until the run, which uploads it in slices
// beside the march (the caller's buffer outlives the call, the plan does not)
int rtr::plan_set_rays_deferred(rt_hip_plan *p, const rt_ray *rays, size_t n_rays)
{
    if (n_rays > MAX_LIST_RAYS)
        return fail_arg("ray list of 2^32 - 4096 rays or more: split the call");
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

extern "C" {

int rt_hip_plan_set_rays(rt_hip_plan *p, const rt_ray *rays, size_t n_rays)
{
    if (!p || (n_rays && !rays))
        return fail_arg("rt_hip_plan_set_rays: NULL argument");
    if (n_rays > MAX_LIST_RAYS)
        return fail_arg("rt_hip_plan_set_rays: 2^32 - 4096 rays or more: split the call");
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
            const int rc = launch_tan(p->rays_dev, (unsigned long long) n_rays, p->tan_dev, nullptr);
            if (rc != RT_OK)
                return rc;
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
    // one upload: the four grids, then the launch tangents of the a and b values -- Helper.h:409-410,
    // tanf(1e-3f * ray.a) depends only on the grid value: nga + ngb evaluations on the host, with the same libm
    // the CPU loop uses
    const size_t nt = (size_t) nga + (size_t) ngb;
    std::vector<double> h(nn + (nt + 1) / 2);
    memcpy(h.data(), gx, sizeof(double) * (size_t) ngx);
    memcpy(h.data() + ngx, gy, sizeof(double) * (size_t) ngy);
    memcpy(h.data() + ngx + ngy, ga, sizeof(double) * (size_t) nga);
    memcpy(h.data() + ngx + ngy + nga, gb, sizeof(double) * (size_t) ngb);
    {
        std::vector<float> ht(nt);
        for (int k = 0; k < nga; k++)
            ht[(size_t) k] = tanf(1e-3f * (float) ga[k]);
        for (int m = 0; m < ngb; m++)
            ht[(size_t) nga + (size_t) m] = tanf(1e-3f * (float) gb[m]);
        memcpy(h.data() + nn, ht.data(), nt * sizeof(float));
    }
    HIP_TRY(pool_alloc(p->device, (void **) &p->grid_dev, h.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(p->grid_dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    pool_free(p->device, p->tan_dev); // (the per-ray tangents of a list this plan may have held before)
    p->tan_dev            = nullptr;
    const float *tan_grid = reinterpret_cast<const float *>(p->grid_dev + nn);
    p->host_rays   = nullptr;
    rt::DevRays &R = p->P.rays;
    R              = {};
    R.list         = nullptr;
    R.tan_a        = tan_grid;
    R.tan_b        = tan_grid + nga;
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
        const int rc = launch_seed_tab(p->P.seed, R, nn, p->seedtab_dev, flags);
        if (rc != RT_OK)
            return rc;
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
    // (with emission only: the gain-only instance of the frequency kernel carries no exclusive deposit)
    p->P.exclusive = (p->P.own_cells && p->P.use_emis && nga == 1 && ngb == 1 && first == 0 && stride == 1 && count == total) ? 1u : 0u;
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

} // extern "C"

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

extern "C" {

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
    // from here on work is queued: whatever happens below, destroy / quiesce must wait for this queue
    p->queued_stream = stream;
    p->queued        = true;
    if (p->probe_on && p->n_rays)
        HIP_TRY(hipMemsetAsync(p->probe, 0, (size_t) p->n_rays * (sizeof(rt_ray) + 8), stream));
    static_assert(sizeof(rt::DevCtl) % 8 == 0 && alignof(rt::DevCtl) >= 8, "zeroed in 8-byte words");
    // (exclusive mode writes every image row exactly once: its image is not zeroed)
    rc = launch_zero3(stream, p->P.exclusive ? nullptr : image_dev, p->n_image * sizeof(double), iang_dev, p->n_iang * sizeof(double),
                      p->ctl, sizeof(rt::DevCtl));
    if (rc != RT_OK)
        return rc;
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

    rc = plan_launch_run(p, stream);
    if (rc != RT_OK)
        return rc;
    p->last_stream = stream;
    p->last_image  = image_dev;
    p->last_iang   = iang_dev;
    p->ran         = true;
    p->queued      = false; // (`ran` + last_stream cover it from here)
    p->repeated    = false;
    p->out_staged  = false;
    return RT_OK;
}

int rt_hip_plan_fetch(rt_hip_plan *p, double *image, double *I_ang, unsigned int *failure_code,
                      rt_ray *failed_rays, int max_failed, int *n_failed, rt_stats *stats)
{
    if (!p || !p->ran)
        return fail_arg("rt_hip_plan_fetch: plan has not run");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->last_stream));
    // the control block behind the chunk counters: failure code, failed rays, statistics
    rt::DevCtl c;
    constexpr size_t ctl_tail = offsetof(rt::DevCtl, failure_code), ctl_bytes = sizeof(rt::DevCtl) - ctl_tail;
    auto read_ctl = [&]() {
        return hipMemcpy(reinterpret_cast<unsigned char *>(&c) + ctl_tail, reinterpret_cast<const unsigned char *>(p->ctl) + ctl_tail,
                         ctl_bytes, hipMemcpyDeviceToHost);
    };
    // (outputs that travelled behind the kernels already: plan_stage_outputs)
    bool staged = p->out_staged && p->out_staging;
    if (staged)
        memcpy(reinterpret_cast<unsigned char *>(&c) + ctl_tail, p->out_staging, ctl_bytes);
    else
        HIP_TRY(read_ctl());
    // rays that failed in the frequency pass have been deposited: repeat the pass without them
    if ((c.failure_code & ((1u << 2) | (1u << 3))) && !p->path_on && !(p->P.debug & 1u) && !p->repeated) {
        const int rc = plan_repeat_checked(p);
        if (rc != RT_OK)
            return rc;
        p->repeated   = true; // this run's outputs are final; a second fetch must not repeat again
        p->out_staged = staged = false;
        HIP_TRY(read_ctl());
    }
    if (staged) {
        const size_t off_ang = align_up(ctl_bytes, 256), off_img = off_ang + align_up(p->n_iang * sizeof(double), 256);
        if (image)
            memcpy(image, p->out_staging + off_img, p->n_image * sizeof(double));
        if (I_ang)
            memcpy(I_ang, p->out_staging + off_ang, p->n_iang * sizeof(double));
    } else {
        if (image)
            HIP_TRY(hipMemcpy(image, p->last_image, p->n_image * sizeof(double), hipMemcpyDeviceToHost));
        if (I_ang)
            HIP_TRY(hipMemcpy(I_ang, p->last_iang, p->n_iang * sizeof(double), hipMemcpyDeviceToHost));
    }
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

int rt_hip_plan_last_fused(rt_hip_plan *p) { return p && p->ran && p->last_fused ? 1 : 0; }

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
        std::vector<unsigned char> h(rt::rec_bytes(n, p->P.rec_stride));
        if (n)
            HIP_TRY(hipMemcpy(h.data(), p->rec, h.size(), hipMemcpyDeviceToHost));
        for (size_t r = 0; r < n; r++) {
            const unsigned char *rec = h.data();
            const rt::RecMeta *mt = reinterpret_cast<const rt::RecMeta *>(rec + rt::rec_meta_off((unsigned) r, (int) S, p->P.rec_stride));
            for (size_t q = 0; q < S; q++) {
                const rt::RecSlot sl = rt::rec_slot(rec, (unsigned) r, p->P.rec_stride, (int) q, (int) S, mt->flags_steps, p->P.method == 1);
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
        return fail_arg("rt_hip_image_loop: 2^32 - 4096 rays or more: split the call");
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
    // The call's own queue first: the tables travel on it from page-locked staging while the host goes on (grid
    // recognition, the small uploads, the launch), and everything that reads them is launched on it.
    hipStream_t q  = lease_queue(device);
    rt_hip_plan *p = nullptr;
    int rc         = plan_create_on(&p, q, device, N, beam, gain, seed, method, scale);
    if (rc != RT_OK) {
        release_queue(device, q);
        return rc;
    }
    // (the seed-factor tables of a seeded plan are filled by a kernel outside this queue: rt_hip_plan_set_ray_grid)
    if (seed && q && hipStreamSynchronize(q) != hipSuccess)
        (void) hipGetLastError();
    lap("plan_create");
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
        if (rc == RT_OK)
            plan_stage_outputs(p);
        lap(p->last_fused ? "run (one launch)" : "run (two launches)");
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
        if (rc == RT_OK)
            plan_stage_outputs(p);
    }
    if (rc == RT_OK)
        rc = rt_hip_plan_fetch(p, image, I_ang, failure_code, failed_rays, max_failed, n_failed, stats);
    lap("fetch (wait + D2H)");
    rt_hip_plan_destroy(p); // waits for whatever is still in flight
    release_queue(device, q);
    lap("destroy");
    return rc;
}

void rt_hip_pool_trim(void) { pool_trim_all(); }

} // extern "C"
