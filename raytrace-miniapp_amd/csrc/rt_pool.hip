// rt_pool.hip -- resources the C ABI keeps across calls, and the error text: freed device allocations
// (never data: Readme.txt:43 forbids caching data across create_image calls, not allocations), one set of
// non-blocking queues per device, the environment overrides.  Host code only.
#include "rt_runtime.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace rtr {

std::string &last_error()
{
    thread_local std::string text;
    return text;
}

int fail_hip(hipError_t e, const char *what, const char *file, int line)
{
    const char *base = file;
    for (const char *c = file; *c; c++)
        if (*c == '/')
            base = c + 1;
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, base, line, hipGetErrorString(e));
    last_error() = buf;
    return RT_ERR_HIP;
}
int fail_arg(const char *msg)
{
    last_error() = msg;
    return RT_ERR_ARG;
}

// Device-memory pool: create_image is called once per iteration of the application and may
// not keep DATA across calls (Readme.txt:43), but nothing forbids keeping ALLOCATIONS: the
// ray list (16 B/ray), the tangents and the march records (96 B/ray) are hundreds of MB per
// call and hipMalloc/hipFree of them costs milliseconds.  Freed blocks are parked per device
// (at most POOL_MAX_BLOCKS, POOL_MAX_BYTES) and handed out again best-fit.
namespace {
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
// page-locked host staging blocks (pinned_alloc below), under the same mutex
struct PinnedBlock {
    void *ptr;
    size_t bytes;
};
std::vector<PinnedBlock> g_pinned;                 // parked
std::unordered_map<void *, size_t> g_pinned_sizes; // handed out
constexpr size_t PINNED_MAX_BLOCKS = 8;
} // namespace

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
    std::vector<PinnedBlock> drop_pinned;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        drop.swap(g_pool);
        drop_pinned.swap(g_pinned);
    }
    for (auto &b : drop_pinned)
        (void) hipHostFree(b.ptr);
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

// Page-locked host staging (the arena of a plan created by rt_hip_image_loop travels from it while the host goes on):
// hipHostMalloc costs ~0.1 ms and more, so freed blocks are parked like device blocks -- at most eight, best fit.
hipError_t pinned_alloc(void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0)
        bytes = 16;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        int best = -1;
        for (size_t i = 0; i < g_pinned.size(); i++)
            if (g_pinned[i].bytes >= bytes && g_pinned[i].bytes <= 2 * bytes && (best < 0 || g_pinned[i].bytes < g_pinned[(size_t) best].bytes))
                best = (int) i;
        if (best >= 0) {
            *out                 = g_pinned[(size_t) best].ptr;
            g_pinned_sizes[*out] = g_pinned[(size_t) best].bytes;
            g_pinned.erase(g_pinned.begin() + best);
            return hipSuccess;
        }
    }
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        g_pinned_sizes[*out] = bytes;
    } else {
        (void) hipGetLastError();
        *out = nullptr;
    }
    return e;
}

void pinned_free(void *ptr)
{
    if (!ptr)
        return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        size_t bytes = 0;
        auto it      = g_pinned_sizes.find(ptr);
        if (it != g_pinned_sizes.end()) {
            bytes = it->second;
            g_pinned_sizes.erase(it);
        }
        if (bytes && g_pinned.size() < PINNED_MAX_BLOCKS && bytes <= ((size_t) 256 << 20)) {
            g_pinned.push_back({ ptr, bytes });
            return;
        }
    }
    (void) hipHostFree(ptr);
}

// tuning overrides from the environment: a missing, non-numeric or non-positive value keeps the default
unsigned env_unsigned(const char *name, unsigned def, unsigned lo, unsigned hi)
{
    const char *e = getenv(name);
    if (!e)
        return def;
    const long v = atol(e);
    if (v < 0 || (v == 0 && (lo > 0 || e[0] != '0'))) // (an explicit 0 counts where the range admits it; junk is the default)
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


unsigned host_threads(unsigned cap)
{
    unsigned h = std::thread::hardware_concurrency();
    h          = h ? h : 1;
    return env_unsigned("RT_HIP_HOST_THREADS", h < cap ? h : cap, 1, 64);
}

// Non-blocking queues for the host-pointer entry points, kept across calls like the memory pool
// (a resource, not data; creating one costs ~2 ms): synchronous copies on the host thread do not
// wait for them, which is what lets the ray upload overlap the march.  Every call LEASES a queue
// of its device for its own use, so concurrent calls on one device (create_image is thread-safe,
// RayTrace.h:90-91) neither share a stream nor see each other's kernels in their event times.
namespace {
std::mutex g_queue_mutex;
std::vector<hipStream_t> g_queue_free[64];
} // namespace
hipStream_t lease_queue(int device)
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
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) {
        (void) hipGetLastError(); // the caller goes on without a queue of its own (and reports a bad device itself)
        return nullptr;
    }
    return q;
}
void release_queue(int device, hipStream_t q)
{
    if (!q || device < 0 || device >= 64)
        return;
    std::lock_guard<std::mutex> lock(g_queue_mutex);
    g_queue_free[device].push_back(q);
}

} // namespace rtr
