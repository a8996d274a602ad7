// rt_raygrid.hip -- the ray list of a call, seen from the host: a list that is really a tensor grid
// (RayTrace::create_image builds every list that way, src/RayTraceImage.cpp:300-328) is recognised and
// verified so that the device can generate the rays instead of receiving them; the launch tangents of a
// genuine list (Helper.h:409-410) and the probe that decides who computes them.  Host code only.
#include "rt_runtime.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rtr {

// ---- a ray list that is really a tensor grid ---------------------------------------------------
// RayTrace::create_image builds its list from four 1-D grids, b fastest, then a, y, x
// (src/RayTraceImage.cpp:300-328), and hands the back-end loop only the list.  The grids are read back
// from it in O(nx + ny + na + nb) -- the period of each coordinate -- and the whole list is then
// compared with the grid ray by ray, bit for bit, on host threads.
bool guess_ray_grid(const rt_ray *rays, size_t n, GridGuess &G)
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

// every ray of the list against the grid.  A ray is two 64-bit words, (x, y) and (a, b): the rays of one pixel
// share the first word, and their second words are the same na * nb patterns for every pixel -- one table, built
// once -- so a pixel is one straight loop of two integer compares per ray that the compiler vectorises (the AVX2
// instance is taken when the CPU has it).  Grids with few rays per pixel (na * nb < 64; one ray per pixel in
// BASELINE config 5) are compared a piece of a pixel column at a time instead, against tables that span the column
// (the y half of the first word varies along it).  One thread checks a list of up to 16 MB faster than threads can be
// started (6.4 MB of ASE_small: 0.1 - 0.2 ms, beside a 0.4 ms launch); longer lists go to up to `threads` host
// threads: ~1 ms for the 102 MB of a 6.4 M-ray list on 16 threads, hidden behind the kernels it runs beside.
namespace {
// rays[0 .. n) against first word = xw | y_part[q] (y_part == nullptr: xw alone) and second word = ab[q]
template <bool WITH_Y>
inline __attribute__((always_inline)) uint64_t run_diff(const unsigned char *rays, const uint64_t *y_part, const uint64_t *ab, size_t n,
                                                        uint64_t xw)
{
    uint64_t d0 = 0, d1 = 0;
    for (size_t q = 0; q < n; q++) {
        uint64_t w[2]; // (a list of four-float rays is 4-byte aligned only)
        memcpy(w, rays + 16 * q, 16);
        d0 |= w[0] ^ (WITH_Y ? xw | y_part[q] : xw);
        d1 |= w[1] ^ ab[q];
    }
    return d0 | d1;
}
uint64_t run_diff_plain(const unsigned char *rays, const uint64_t *y_part, const uint64_t *ab, size_t n, uint64_t xw)
{
    return y_part ? run_diff<true>(rays, y_part, ab, n, xw) : run_diff<false>(rays, y_part, ab, n, xw);
}
__attribute__((target("avx2"))) uint64_t run_diff_avx2(const unsigned char *rays, const uint64_t *y_part, const uint64_t *ab, size_t n,
                                                       uint64_t xw)
{
    return y_part ? run_diff<true>(rays, y_part, ab, n, xw) : run_diff<false>(rays, y_part, ab, n, xw);
}
} // namespace

bool verify_ray_grid(const rt_ray *rays, size_t n, const GridGuess &G, unsigned threads)
{
    const size_t nb = G.g[3].size(), na = G.g[2].size(), ny = G.g[1].size(), nab = na * nb;
    auto bits = [](double v) {
        const float f = (float) v;
        uint32_t u;
        memcpy(&u, &f, sizeof(u));
        return (uint64_t) u;
    };
    std::vector<uint64_t> bx(G.g[0].size()), by(ny), ab(nab);
    for (size_t i = 0; i < bx.size(); i++)
        bx[i] = bits(G.g[0][i]);
    for (size_t i = 0; i < ny; i++)
        by[i] = bits(G.g[1][i]) << 32;
    for (size_t k = 0; k < na; k++)
        for (size_t m = 0; m < nb; m++)
            ab[k * nb + m] = bits(G.g[2][k]) | bits(G.g[3][m]) << 32;
    // few rays per pixel: tables over a whole pixel column (unless that would be a table of more than 2^20 entries)
    const bool by_column = nab < 64 && ny * nab <= ((size_t) 1 << 20);
    std::vector<uint64_t> col_y, col_ab;
    if (by_column) {
        col_y.resize(ny * nab);
        col_ab.resize(ny * nab);
        for (size_t j = 0; j < ny; j++)
            for (size_t q = 0; q < nab; q++) {
                col_y[j * nab + q]  = by[j];
                col_ab[j * nab + q] = ab[q];
            }
    }
    const size_t pixels = n / nab;
    threads             = threads < 1 ? 1 : threads;
    if (n * sizeof(rt_ray) <= (size_t) 16 << 20)
        threads = 1;
    static const bool avx2 = __builtin_cpu_supports("avx2");
    const auto diff        = avx2 ? run_diff_avx2 : run_diff_plain;
    std::atomic<bool> ok(true);
    auto work = [&](size_t p0, size_t p1) { // pixels p0 .. p1 of the list (pixel = i * ny + j)
        uint64_t d     = 0;
        size_t checked = 0;
        for (size_t px = p0; px < p1 && d == 0;) {
            const unsigned char *at = reinterpret_cast<const unsigned char *>(rays + px * nab);
            const size_t i = px / ny, j = px % ny;
            size_t step = 1;
            if (by_column) {
                // up to the end of this pixel column, of this thread's range, and of a piece of ~4096 rays
                step = std::min(std::min(ny - j, p1 - px), std::max<size_t>(1, 4096 / nab));
                d |= diff(at, col_y.data() + j * nab, col_ab.data() + j * nab, step * nab, bx[i]);
            } else {
                d |= diff(at, nullptr, ab.data(), nab, bx[i] | by[j]);
            }
            px += step;
            checked += step * nab;
            if (checked >= 65536) { // has another thread found a difference?
                checked = 0;
                if (!ok.load(std::memory_order_relaxed))
                    break;
            }
        }
        if (d != 0)
            ok.store(false, std::memory_order_relaxed);
    };
    if (threads == 1) {
        work(0, pixels);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < threads; t++)
            th.emplace_back(work, pixels * t / threads, pixels * (t + 1) / threads);
        for (auto &t : th)
            t.join();
    }
    return ok.load();
}

int plan_set_guessed_grid(rt_hip_plan *p, const GridGuess &G, int64_t first, int64_t count)
{
    return rt_hip_plan_set_ray_grid(p, G.g[0].data(), (int) G.g[0].size(), G.g[1].data(), (int) G.g[1].size(),
                                    G.g[2].data(), (int) G.g[2].size(), G.g[3].data(), (int) G.g[3].size(), first, 1, count);
}

// x / d for every x < 2^31 as mulhi(x, mul) >> sh (DevRays::div_mul): with s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1
// satisfies mul d = 2^(31+s) + e, 0 < e <= d <= 2^s, hence x mul / 2^(31+s) = x / d + x e / (d 2^(31+s)) with the
// second term below 1 / d: the floor is that of x / d.  mul < 2^32 for d >= 2; d = 1 is flagged by mul = 0.
void magic_u31(unsigned d, unsigned &mul, unsigned &sh)
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
bool grid_points_in_own_cells(const double *g, int n, double d)
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

// ---- list-mode launch tangents and the host's libm ------------------------------------------------
// rt_tan_kernel restates the float tanf of glibc 2.35 (rt_march.hip).  Whether THIS host's tanf is that
// routine is probed once per process: 8192 angles from 1e-3 mrad to 1.37 rad, both signs, device against
// host, bit for bit.  If they differ anywhere (another libm), list-mode tangents are computed by the
// host's tanf on host threads and uploaded -- slower (two libm calls per ray), but the march then starts
// every ray exactly as RayTraceImageCPULoop on this host does.  (Grid mode always uses the host's tanf.)
namespace {
std::atomic<int> g_tan_mode(0); // 0 unknown, 1 device restatement == host tanf, 2 host tangents
} // namespace

void host_tangents(const rt_ray *rays, size_t n, float *sxy)
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

int tan_mode(int device)
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
    if (e == hipSuccess && launch_tan(d_r, (unsigned long long) n, d_sxy, nullptr) != RT_OK)
        e = hipErrorLaunchFailure;
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

} // namespace rtr

extern "C" {

int rt_hip_host_libm_mode(int device) { return rtr::tan_mode(device); }

int rt_hip_ray_list_grid_dims(const rt_ray *rays, size_t n_rays, int dims[4])
{
    rtr::GridGuess G;
    if (!rays || !dims || !rtr::guess_ray_grid(rays, n_rays, G) || !rtr::verify_ray_grid(rays, n_rays, G, rtr::host_threads(16)))
        return 0;
    for (int i = 0; i < 4; i++)
        dims[i] = (int) G.g[i].size();
    return 1;
}

} // extern "C"
