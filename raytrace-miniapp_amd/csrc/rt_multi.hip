// rt_multi.hip -- all devices of the node in one call (rt_hip_multi_image_loop).
//
// Replaces the "cuda-multigpu" arm of the dispatcher (src/RayTraceImage.cpp:389-405 ->
// RayTraceImageThreadLoop :89-134 + setGPU :82-88) and, for the assembly of the image, stands where a
// multi-rank run of the application uses MPI (src/MPI_helpers.h:29-38): one host thread per device with the
// device bound inside the worker, one RCCL communicator over the devices, ONE collective per image -- a
// grouped send/recv gather of pixel-column tiles over xGMI (ASE) or a sum-reduce of whole images (seeded
// mode, arbitrary lists).  librccl.so is loaded on first use.
#include "rt_runtime.h"

#include <rccl/rccl.h> // types and prototypes only: librccl.so is loaded on first use (rccl_api below)

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>

using namespace rtr;

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

extern "C" {

int rt_hip_multi_last_mode(void) { return g_multi_mode; }

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
        last_error() = "no HIP device";
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
            last_error() = "rt_hip_multi_image_loop: " + err;
            return rc;
        }
        R = rccl_api();
    }

    // One attempt at the image.  speculate: a list whose periods say "tensor grid" (O(nx + ny + na + nb)) is taken for
    // that grid at once -- the devices generate their rays and start tracing -- while host threads compare the whole list
    // with the grid, ray by ray (0.7 ms for the 102 MB of the 6.4 M-ray stand-in: as long as the kernels of an 8-device
    // run); the workers wait for the verdict before the collective.  A list that only looked like a grid ends the
    // attempt (RT_RETRY) and the second attempt, which verifies first, traces the list itself.
    constexpr int RT_RETRY = -1;
    auto attempt = [&](const bool speculate) -> int {
    // ---- how to partition ----------------------------------------------------------------------
    GridGuess G;
    const bool guessed = n_rays >= 1 && !getenv("RT_HIP_NO_GRID_DETECT") && guess_ray_grid(rays, n_rays, G);
    std::atomic<int> verdict(guessed && speculate ? 0 : 1); // 0 pending, 1 nothing left to confirm, 2 not that grid
    const bool is_grid = guessed && (speculate || verify_ray_grid(rays, n_rays, G, host_threads(16)));
    std::thread verifier;
    if (guessed && speculate)
        verifier = std::thread([&] { verdict.store(verify_ray_grid(rays, n_rays, G, host_threads(16)) ? 1 : 2); });
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
        // the tables travel on the worker's queue from page-locked staging while the worker goes on to its grid and its
        // buffers (the run is launched on the same queue); a seeded plan fills its seed-factor tables with a kernel
        // outside the queue and therefore waits for its upload as rt_hip_plan_create does
        const hipStream_t up_q = seed ? nullptr : q;
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
                rc = plan_create_on(&p, up_q, pd, N, &bd, gain, seed, method, scale);
                if (rc == RT_OK) {
                    const int cols      = tile_cols(nx, d, ndev);
                    const int64_t count = (int64_t) cols * ny * beam->na * beam->nb;
                    rc = rt_hip_plan_set_ray_grid(p, xd.data(), (int) xd.size(), beam->y, ny, beam->a, beam->na, beam->b,
                                                  beam->nb, 0, 1, count);
                }
            } else {
                rc = plan_create_on(&p, up_q, pd, N, beam, gain, seed, method, scale);
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
        // -- the verdict on a list taken for a grid: nothing travels before the list is known to be that grid
        while (verdict.load() == 0)
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        const bool grid_ok = verdict.load() == 1;
        // -- the one collective of the image, on the queue the kernels ran on
        if (loopback > 0) {
            // rehearsal: every worker copies its part into the receive layout, worker 0 assembles
            const bool all_ok = meet.arrive(w.rc == RT_OK && grid_ok);
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
        } else if (meet.arrive(w.rc == RT_OK && grid_ok)) {
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
    if (verifier.joinable())
        verifier.join();
    // (a worker that was pulled out of the collective by another one's abort is not the one to quote)
    for (int pass = 0; pass < 2; pass++)
        for (int d = 0; d < ndev; d++)
            if (W[(size_t) d].rc != RT_OK && (pass == 1 || W[(size_t) d].error.rfind("collective aborted", 0) != 0)) {
                last_error() = "device " + std::to_string(d) + ": " + W[(size_t) d].error;
                return W[(size_t) d].rc;
            }
    if (verdict.load() == 2)
        return RT_RETRY; // the list only looked like a grid: nothing was assembled
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
    }; // attempt
    int rc = attempt(true);
    if (rc == RT_RETRY)
        rc = attempt(false);
    return rc;
}

} // extern "C"
