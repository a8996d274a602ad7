#pragma once
// rt_march.hip -- kernel A: the float32 ray march (Helper.h:404-513) as a
// persistent-lane state machine over LDS-resident plasma tables.
//
// Why not "one lane marches one ray of a 64-ray tile": the reference's three
// nested data-dependent loops (cell loop :463, cross-cell loop :326, integrator
// loop :279) leave a wave64 running the innermost loop at 26 % lane occupancy
// (measured, tools/loop_occupancy.py) -- lanes wait for the slowest ray at every
// nesting level and for the longest ray of the tile.  Here a lane never waits:
//   * a lane whose ray is finished immediately takes the next ray of the wave's
//     reserved chunk (one global atomic per `chunk` rays, not per ray);
//   * the nest is flattened: every wave iteration runs three predicated blocks
//     [A] cell-loop bookkeeping + cell setup, [B] cross-cell setup, [C] one
//     integrator step, with the loop-exit tests evaluated eagerly at the end of
//     [C] so that every live lane performs exactly one integrator step per
//     iteration (92 % occupancy of the integrator step, measured).  The arithmetic
//     and its order are exactly those of the nested loops: the march record is
//     bit-identical to the CPU loop.
// Tables: what a cell-step gathers -- per-axis interval records and the fused
// {n, g0, E0} corner nodes of every length -- is one "march blob" (rt_device.h)
// that each work-group copies into LDS once (LDS variant, one work-group per CU)
// so that the dependent index -> coordinate -> corner read chain of block [A] costs
// LDS latency, not three L2 round trips; a blob too large for LDS is read in
// place (global variant).  The per-ray record (gvl/evl/ivl + exit state) leaves
// as one 96-byte line per ray; rt_freq.hip consumes it with lanes = rays.
#include "rt_math.h"

#include <cfloat>

namespace rt {

enum : int { ST_IDLE = 0, ST_CELL = 1, ST_XSETUP = 2, ST_STEP = 3, ST_DONE = 4 };

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// grid mode: indices of ray ridx on the four ray grids
// (RayTraceImage.cpp:300-328: b fastest, then a, y, x)
__device__ __forceinline__ unsigned div_magic(unsigned x, unsigned mul, unsigned sh)
{
    return mul ? __umulhi(x, mul) >> sh : x; // (wave-uniform select)
}
__device__ __forceinline__ void grid_index(const DevRays &R, unsigned ridx, unsigned &i, unsigned &j, unsigned &k,
                                           unsigned &m)
{
    const unsigned ijkm = (unsigned) (R.first + (long long) ridx * R.stride); // < 2^31 (rt_hip_plan_set_ray_grid)
    const unsigned q1   = div_magic(ijkm, R.div_mul[0], R.div_sh[0]);
    m                   = ijkm - q1 * (unsigned) R.ngb;
    const unsigned q2   = div_magic(q1, R.div_mul[1], R.div_sh[1]);
    k                   = q1 - q2 * (unsigned) R.nga;
    i                   = div_magic(q2, R.div_mul[2], R.div_sh[2]);
    j                   = q2 - i * (unsigned) R.ngy;
}

// start ray of flat index ridx: position, and tangent of the launch angles
__device__ __forceinline__ void load_ray(const DevRays &R, unsigned ridx, rt_ray &ray, float &ta, float &tb,
                                         bool want_tan)
{
    if (R.list) {
        ray = R.list[ridx];
        if (want_tan) {
            ta = R.sxy[2 * (size_t) ridx];
            tb = R.sxy[2 * (size_t) ridx + 1];
        }
    } else {
        unsigned i, j, k, m;
        grid_index(R, ridx, i, j, k, m);
        ray.x      = (float) R.gx[i];
        ray.y      = (float) R.gy[j];
        ray.a      = (float) R.ga[k];
        ray.b      = (float) R.gb[m];
        if (want_tan) {
            ta = R.tan_a[k];
            tb = R.tan_b[m];
        }
    }
}

// Wide-argument fallbacks of the two float kernels below.  Kept out of line: inlined, the
// double-precision libm bodies park ~40 VGPRs of polynomial coefficients across the whole
// calling kernel for a branch that the shipped inputs never take.
// Neither hands a NaN or an infinity to the device library's float64 routines: its f64 tangent faulted on an infinity
// (the argument reduction indexes a table with the exponent; found by tests/test_gpu_edges.py in round 4), so the
// non-finite cases get libm's answers here -- tan(+-inf) = tan(NaN) = NaN, atan(+-inf) = +-pi/2, atan(NaN) = NaN.
__device__ __attribute__((noinline)) float tan_wide(float x)
{
    if (!(fabsf(x) <= FLT_MAX))
        return x - x;
    return (float) tan((double) x);
}
__device__ __attribute__((noinline)) float atan_wide(float x)
{
    if (x != x)
        return x + x;
    if (!(fabsf(x) <= FLT_MAX))
        return copysignf(1.5707963705062866f, x); // (float) (pi / 2), what atanf(+-inf) returns
    return (float) atan((double) x);
}

// The three float kernels below (ktanf_flt32 / tanf_flt32_wide / tanf_flt32_kernel, atanf_flt32_kernel) restate
// routines of fdlibm as shipped in GNU libc 2.35 (sysdeps/ieee754/flt-32/k_tanf.c, e_rem_pio2f.c, s_atanf.c;
// float conversions by Ian Lance Taylor, Cygnus Support) -- same coefficients, same evaluation order, because
// bit-exactness with the reference platform's libm requires it.  Their notice:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
//
// tanf as the reference's libm computes it.  Helper.h:409-410 calls tanf(1e-3f * a); on
// the reference platform that is GNU libc 2.35, sysdeps/ieee754/flt-32/{s,k}_tanf.c --
// the fdlibm float kernel: for |x| < 0.6744 a degree-13 odd polynomial evaluated in float
// in a fixed order.  Restated here (same coefficients, same order, no FMA contraction);
// tests/test_float_identities.py checks the restatement against the host tanf for every
// float in [4.6e-10, 0.2].  Launch angles beyond 200 mrad: tanf_flt32_wide below.
// The other branches of the same libm routine, for launch angles of 200 mrad ... 1.375 rad: k_tanf.c
// with its |x| >= 0.6744 transformation and its accurate -1/(x + r), behind the first case (|x| < 3 pi/4,
// n = +-1) of the float argument reduction e_rem_pio2f.c.  Checked against the host tanf for EVERY float of
// [0.2, 1.375] (tests/test_float_identities.py): 0 mismatches; closer to pi/2 the host differs in the last
// bit for 5e-5 of the floats, so beyond 1.375 rad (where a ray is 79 degrees off the axis and fails the
// s.z^2 >= 0.01 test of Helper.h:515 anyway) the float-rounded f64 tangent stands in.  Out of line: only
// rt_tan_kernel calls it.
__device__ __attribute__((noinline)) float ktanf_flt32(float x, float y, int iy)
{
    const float T[13] = { 3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f,
                          3.5920790397e-03f, 1.4562094584e-03f, 5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f,
                          7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f };
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    float z, r, v, w, s;
    const int hx = (int) __float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {
        if ((int) x == 0) {
            if ((ix | (iy + 1)) == 0)
                return 1.0f / fabsf(x);
            else if (iy == 1)
                return x;
            else
                return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) { // |x| >= 0.6744
        if (hx < 0) {
            x = -x;
            y = -y;
        }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w;
        y = 0.0f;
        if (fabsf(x) < 0x1p-13f)
            return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    z = x * x;
    w = z * z;
    r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float) iy;
        return (float) (1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1)
        return w;
    // -1 / (x + r), accurately
    float a, t;
    z = __uint_as_float(__float_as_uint(w) & 0xfffff000u);
    v = r - (z - x);
    t = a = -1.0f / w;
    t = __uint_as_float(__float_as_uint(t) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
__device__ __attribute__((noinline)) float tanf_flt32_wide(float x)
{
    const float pio2_1 = 1.5707855225e+00f, pio2_1t = 1.0804334124e-05f, pio2_2 = 1.0804273188e-05f,
                pio2_2t = 6.0770999344e-11f;
    const int hx = (int) __float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) // |x| <= pi/4
        return ktanf_flt32(x, 0.0f, 1);
    float y0, y1, z;
    if (hx > 0) {
        z = x - pio2_1;
        if ((ix & 0xfffffff0) != 0x3fc90fd0) {
            y0 = z - pio2_1t;
            y1 = (z - y0) - pio2_1t;
        } else {
            z -= pio2_2;
            y0 = z - pio2_2t;
            y1 = (z - y0) - pio2_2t;
        }
    } else {
        z = x + pio2_1;
        if ((ix & 0xfffffff0) != 0x3fc90fd0) {
            y0 = z + pio2_1t;
            y1 = (z - y0) + pio2_1t;
        } else {
            z += pio2_2;
            y0 = z + pio2_2t;
            y1 = (z - y0) + pio2_2t;
        }
    }
    return ktanf_flt32(y0, y1, -1);
}

__device__ __forceinline__ float tanf_flt32_kernel(float x)
{
    const float ax = fabsf(x);
    if (ax < 0x1p-13f) // s_tanf.c/k_tanf.c: (int) x == 0 -> return x
        return x;
    if (ax > 0.2f) {
        // (tanf(+-inf) = NaN, as every libm has it; the device library's f64 tangent must not see an infinity: its
        // argument reduction indexes a table with the exponent and faulted on it -- found by tests/test_gpu_edges.py)
        if (!(ax <= FLT_MAX))
            return x - x;
        return ax <= 1.375f ? tanf_flt32_wide(x) : tan_wide(x);
    }
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
                T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
                T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
                T12 = 2.5907305826e-05f;
    const float z = x * x;
    const float w = z * z;
    float r       = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    const float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    const float s = z * x;
    r             = 0.0f + z * (s * (r + v) + 0.0f);
    r += T0 * s;
    return x + r;
}

// atanf, likewise (Helper.h:520-521: atan(s.x / s.z) * 1e3f): glibc 2.35 flt-32 s_atanf.c,
// the |x| < 7/16 branch (exit directions are far inside it); checked against the host atanf
// for every float of that range by tests/test_float_identities.py.
__device__ __forceinline__ float atanf_flt32_kernel(float x)
{
    const float ax = fabsf(x);
    if (ax < 0x1p-29f)
        return x;
    if (!(ax < 0.4375f))
        return atan_wide(x);
    const float A0 = 3.3333334327e-01f, A1 = -2.0000000298e-01f, A2 = 1.4285714924e-01f, A3 = -1.1111110449e-01f,
                A4 = 9.0908870101e-02f, A5 = -7.6918758452e-02f, A6 = 6.6610731184e-02f, A7 = -5.8335702866e-02f,
                A8 = 4.9768779427e-02f, A9 = -3.6531571299e-02f, A10 = 1.6285819933e-02f;
    const float z  = x * x;
    const float w  = z * z;
    const float s1 = z * (A0 + w * (A2 + w * (A4 + w * (A6 + w * (A8 + w * A10)))));
    const float s2 = w * (A1 + w * (A3 + w * (A5 + w * (A7 + w * A9))));
    return x - x * (s1 + s2);
}

// grid mode with a seed, forward method: the per-axis factors of the seed profile at every
// point of the four ray grids (the grid value rounded to float, as the ray carries it)
extern "C" __global__ void __launch_bounds__(256) rt_seed_tab_kernel(const DevSeed sd, const DevRays R, double *sf,
                                                                     unsigned char *sin)
{
    const int n0 = R.ngx, n1 = R.ngy, n2 = R.nga, n3 = R.ngb;
    for (int t = (int) (blockIdx.x * blockDim.x + threadIdx.x); t < n0 + n1 + n2 + n3; t += (int) (gridDim.x * blockDim.x)) {
        int d           = t < n0 ? 0 : (t < n0 + n1 ? 1 : (t < n0 + n1 + n2 ? 2 : 3));
        const int i     = t - (d > 0 ? n0 : 0) - (d > 1 ? n1 : 0) - (d > 2 ? n2 : 0);
        const double *g = d == 0 ? R.gx : (d == 1 ? R.gy : (d == 2 ? R.ga : R.gb));
        const double v  = (double) (float) g[i];
        const bool in   = v >= sd.x[d][0] && v <= sd.x[d][sd.dim[d] - 1]; // Helper.h:233-236
        sin[t]          = in ? 1 : 0;
        sf[t]           = in ? pchip_eval(sd.dim[d], sd.x[d], sd.f[d], v) : 0.0;
    }
}

// rt_hip_selftest: the two exact shortcuts of the march against the IEEE sequences, bit for bit, on the device
// itself: inv_norm for every float of its shortcut range and every 256th bit pattern elsewhere; fdiv_nr /
// fdiv_one_nr against `/` for every divisor of [0.25, 4) (1/n) and for 2^28 pseudo-random operand pairs drawn
// log-uniformly from the ranges the integrator's step candidates can take under DevParams-bounded tables
// (dividend 2^-84 .. 1, divisor 2^-78 .. 2^44, quotient normal, exponents less than 96 apart).
__device__ __forceinline__ unsigned selftest_hash(unsigned x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
extern "C" __global__ void __launch_bounds__(256) rt_selftest_kernel(unsigned long long *counts)
{
    const unsigned lo = 0x3f700000u, hi = 0x3f880000u; // [0.9375, 1.0625)
    const unsigned n_fast = hi - lo, n_other = 1u << 24;
    unsigned long long checked = 0, bad = 0;
    const unsigned long long tid = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nth = (unsigned long long) gridDim.x * blockDim.x;
    for (unsigned long long t = tid; t < (unsigned long long) n_fast + n_other; t += nth) {
        const unsigned u = t < n_fast ? lo + (unsigned) t : (unsigned) (t - n_fast) << 8;
        float q          = __uint_as_float(u);
        const float a    = inv_norm(q);
        asm volatile("" : "+v"(q));
        const float b = 1.0f / sqrtf(q);
        checked++;
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b))
            bad++;
    }
    // 1 / n for every float n of [0.25, 4)
    for (unsigned long long t = tid; t < (4ull << 23); t += nth) {
        float n       = __uint_as_float(0x3e800000u + (unsigned) t);
        const float a = fdiv_one_nr(n);
        asm volatile("" : "+v"(n));
        const float b = 1.0f / n;
        checked++;
        bad += __float_as_uint(a) != __float_as_uint(b) ? 1 : 0;
    }
    // the step candidates: dividend 2^(ea) * [1, 2), divisor 2^(eb) * [1, 2), ea in [-84, -1], eb in [-78, 43],
    // pairs whose exponents are 96 or more apart or whose quotient could be subnormal are skipped (never the minimum)
    for (unsigned long long t = tid; t < (1ull << 28); t += nth) {
        const unsigned h0 = selftest_hash((unsigned) t * 2u + 1u), h1 = selftest_hash((unsigned) t * 2u + 0x9e3779b9u);
        const int ea = -84 + (int) (h0 % 84u), eb = -78 + (int) (h1 % 122u);
        if (ea - eb >= 96 || eb - ea >= 96 || ea - eb < -120)
            continue;
        float x = __uint_as_float(((unsigned) (ea + 127) << 23) | (selftest_hash(h0) & 0x7fffffu));
        float y = __uint_as_float(((unsigned) (eb + 127) << 23) | (selftest_hash(h1) & 0x7fffffu));
        const float a = fdiv_nr(x, y);
        asm volatile("" : "+v"(x), "+v"(y));
        const float b = x / y;
        checked++;
        bad += __float_as_uint(a) != __float_as_uint(b) ? 1 : 0;
    }
    // the wide-argument fall-backs of the two float kernels on what the device library must never see (tan_wide /
    // atan_wide above): non-finite arguments give libm's answers, the largest finite ones do not fault
    if (tid == 0) {
        float inf = __uint_as_float(0x7f800000u), qnan = __uint_as_float(0x7fc00000u), big = 3.0e38f;
        asm volatile("" : "+v"(inf), "+v"(qnan), "+v"(big));
        const unsigned pio2 = 0x3fc90fdbu;
        const float t0 = tanf_flt32_kernel(inf), t1 = tanf_flt32_kernel(-inf), t2 = tanf_flt32_kernel(qnan), t3 = tan_wide(inf),
                    t4 = tan_wide(qnan), t5 = tanf_flt32_kernel(big);
        const float a0 = atanf_flt32_kernel(inf), a1 = atanf_flt32_kernel(-inf), a2 = atanf_flt32_kernel(qnan),
                    a3 = atanf_flt32_kernel(big), a4 = atanf_flt32_kernel(-big);
        checked += 11;
        bad += (t0 == t0) + (t1 == t1) + (t2 == t2) + (t3 == t3) + (t4 == t4) + (__float_as_uint(t5) == 0x7fc12345u); // (t5: must only not fault)
        bad += (__float_as_uint(a0) != pio2) + (__float_as_uint(a1) != (pio2 | 0x80000000u)) + (a2 == a2) +
               (__float_as_uint(a3) != pio2) + (__float_as_uint(a4) != (pio2 | 0x80000000u));
    }
    atomicAdd(&counts[0], checked);
    atomicAdd(&counts[1], bad);
}

// list mode: tangents of the launch angles (Helper.h:409-410) for every ray, at full lane
// occupancy, before the march
extern "C" __global__ void __launch_bounds__(256) rt_tan_kernel(const rt_ray *rays, unsigned long long n, float *sxy)
{
    unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        rt_ray r       = rays[i];
        sxy[2 * i]     = tanf_flt32_kernel(1e-3f * r.a);
        sxy[2 * i + 1] = tanf_flt32_kernel(1e-3f * r.b);
    }
}

// Helper.h:131-143 on the interval records of one axis: the unique u in [1, n-1] with
// (u == 1 || g[u-1] < v) && (u == n-1 || g[u] >= v) on a non-decreasing grid.  The cell
// setup guesses u arithmetically (exact on the uniform grids the files hold) and verifies
// the guess with the two coordinates it gathers anyway; this bisection runs when the guess
// fails (non-uniform grid, rounding at an interval edge).  g[mid] = iv[mid].hi for the
// interior points the bisection probes.
__device__ __forceinline__ int bisect_interval(const Interval *iv, int n, double v)
{
    int lo = 0, hi = n - 1;
    while (hi - lo != 1) {
        int mid = (hi + lo) / 2;
        if (iv[mid].hi >= v)
            hi = mid;
        else
            lo = mid;
    }
    return hi;
}
__device__ __forceinline__ int guess_interval(int n, float g0, float inv_h, float v)
{
    // a guess only (the caller verifies it against the interval's own coordinates): float is enough
    // (clamped as a float, before the conversion: the overflow of the conversion itself would be undefined)
    return (int) fminf(fmaxf((v - g0) * inv_h, 0.0f), (float) (n - 2)) + 1;
}

// BOUNDED: rt_hip_plan_create has verified the table and step-size ranges under which the integrator's five
// divisions per step need none of the scaling / fix-up instructions of an IEEE division (rt_math.h, fdiv_nr);
// otherwise (exotic tables) every division is the full sequence.  Both give the reference's floats.
#ifdef RT_INSTRUMENT
__device__ unsigned short g_ray_iters[1u << 23]; // diagnostic: loop iterations each ray occupied its lane for (ray number < 2^23)
#endif
#ifdef RT_WAVETIMES // diagnostic: start / counters-dry / end time of every wave of the last launch (100 MHz clock)
__device__ unsigned long long g_wt[8];
__device__ unsigned long long g_wt_end[8192], g_wt_dry[8192];
// ... and a trace of every wave's march: every 16th loop iteration {100 MHz time | live lanes << 40 | runs of [A] in
// the last 16 iterations << 48 | runs of [B] << 56}, up to 64 samples; [wave][0] = samples written
constexpr int WT_TRACE = 64;
__device__ unsigned long long g_wt_trace[8192][WT_TRACE];
// shader-clock cycles of the last 16 iterations spent in {refill + loop head, [A1], [A2], retire + publish, [B], [C]}
__device__ unsigned g_wt_blocks[8192][WT_TRACE][6];
#endif
// Work-group list of finished tiles of the fused kernel (rt_fused.hip): a wave that has marched all 64 rays of a
// chunk -- one tile of the frequency pass -- pushes the tile number; waves whose rays have run out pop tiles and run
// their frequency pass.  A lock-free stack: the head is an LDS word, the links are one word per tile in global
// memory (`next`); tiles are pushed once and never pushed again, so a pop cannot meet a recycled node.
struct TileList {
    unsigned *head; // LDS: reference of the top node, TILE_NONE when empty
    unsigned *next; // global, [4 n_tiles]: link of list entry (tile, part) at 4 tile + part (overflow nodes only)
    // The nodes live in LDS: node i = {entry, link} at lnode[2 i], i handed out once by the counter *nalloc and never
    // again (so a pop cannot meet a recycled node); LDS operations of a wave execute in order, so a push is
    // write-entry, write-link, compare-exchange with no waiting in between.  (Round 4 kept the links in global memory:
    // every attempt of a push then waited for a store to complete -- microseconds under load -- and failed whenever
    // another wave had popped meanwhile: tools/wave_trace.py showed single pushes of up to 100 us, with the 64 lanes of a
    // marching wave standing still behind them.)  A work-group that pushes more than `cap` entries puts the surplus on
    // the same stack with global links, as before: a reference with bit 31 set is such an entry.
    unsigned *lnode;  // LDS, [cap][2]
    unsigned *nalloc; // LDS
    unsigned cap;
    unsigned *rem;  // LDS, [32] per wave: rays of the wave's tiles in flight that have not retired yet
    unsigned *marching; // LDS: waves of the work-group that have not left the march yet
    unsigned n_waves;   // of the work-group
    unsigned split;     // 0: tiles are pushed whole; 1: split where idle waves would wait for them; 2: always (tests)
    unsigned k_part;    // frequencies per part of a split tile (a multiple of 4), 0: never split
};
constexpr unsigned TILE_NONE = 0xffffffffu;
// A list entry is a whole tile, or one of four parts of its frequency range [part k_part, (part + 1) k_part):
// the frequency pass of a tile that is finished when the rest of the work-group has nothing left to do -- the last
// tiles of a launch, behind the longest rays -- is shared by four waves instead of keeping one busy and the launch open.
constexpr unsigned TILE_PART_FLAG = 1u << 30, TILE_PART_SHIFT = 28, TILE_ID_MASK = (1u << TILE_PART_SHIFT) - 1u;
__device__ __forceinline__ unsigned tile_node(unsigned entry)
{
    return (entry & TILE_ID_MASK) * 4u + ((entry & TILE_PART_FLAG) ? (entry >> TILE_PART_SHIFT) & 3u : 0u);
}
#ifdef RT_WAVETIMES
__device__ unsigned long long g_wt_pub[8192][4]; // per wave: {ticks in tile_publish, pushes, failed compare-exchanges, longest publish}
__device__ unsigned long long g_wt_vm[8192]; // ticks waiting for the record stores before a publish
#endif
// one lane of the calling wave executes these
constexpr unsigned TILE_REF_GLOBAL = 1u << 31;
__device__ __forceinline__ void tile_push(const TileList &T, unsigned tile)
{
    const unsigned id = __hip_atomic_fetch_add(T.nalloc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned old      = __hip_atomic_load(T.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (id < T.cap) {
        __hip_atomic_store(&T.lnode[2u * id], tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (;;) {
            __hip_atomic_store(&T.lnode[2u * id + 1u], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (entry and link are in LDS before the head can name the node)
            if (__hip_atomic_compare_exchange_strong(T.head, &old, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                return;
#ifdef RT_WAVETIMES
            g_wt_pub[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u][2]++;
#endif
        }
    }
    tile |= TILE_REF_GLOBAL;
    for (;;) {
        __hip_atomic_store(&T.next[tile_node(tile & ~TILE_REF_GLOBAL)], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (pushed and popped by waves of one work-group)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the link is in memory before the head can name the tile
        if (__hip_atomic_compare_exchange_strong(T.head, &old, tile, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
            return;
#ifdef RT_WAVETIMES
        g_wt_pub[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u][2]++;
#endif
    }
}
// a finished tile: whole, or in four parts when the work-group is down to its last marching waves (an eighth of them)
// and the list is empty, i.e. nearly all its waves are waiting for work.  (Splitting whenever the list was empty and
// SOME wave had left the march cost 2 % on the 6.4 M-ray stand-in: every part repeats the tile's preamble.)
__device__ __forceinline__ void tile_publish(const TileList &T, unsigned tile)
{
    bool parts = T.split == 2u;
    if (T.split == 1u)
        parts = __hip_atomic_load(T.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == TILE_NONE &&
                __hip_atomic_load(T.marching, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 8u <= T.n_waves;
    if (parts && T.k_part != 0u) {
        for (unsigned q = 0; q < 4u; q++)
            tile_push(T, tile | (q << TILE_PART_SHIFT) | TILE_PART_FLAG);
    } else {
        tile_push(T, tile);
    }
}
__device__ __forceinline__ unsigned tile_pop(const TileList &T)
{
    unsigned old = __hip_atomic_load(T.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (old != TILE_NONE) {
        const unsigned nxt = (old & TILE_REF_GLOBAL)
                                 ? __hip_atomic_load(&T.next[tile_node(old & ~TILE_REF_GLOBAL)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                 : __hip_atomic_load(&T.lnode[2u * old + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_compare_exchange_strong(T.head, &old, nxt, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
            break;
    }
    if (old == TILE_NONE)
        return TILE_NONE;
    // the entry the node carries (an LDS node's entry word never changes once the node is on the list)
    return (old & TILE_REF_GLOBAL) ? (old & ~TILE_REF_GLOBAL) : __hip_atomic_load(&T.lnode[2u * old], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The march of one wave over the rays of a launch.  FUSED (rt_fused.hip): the reserved chunks are whole 64-ray tiles
// (P.ray_begin = 0, P.chunk a multiple of 64) and are handed to the lanes tile by tile; every tile in flight has a
// counter of its rays that have not retired (done.rem, one of 32 slots of this wave), and the wave pushes the tile
// onto `done` when the counter reaches zero and all its record stores have landed.
// ---- tables: copy the march blob to LDS once per work-group (every wave of the work-group takes part; ends in a barrier) ----
template <bool LDS_TAB> __device__ __forceinline__ void march_load_tables(const DevParams &P, unsigned char *lds_raw)
{
    if (LDS_TAB) {
        // (eight 16-byte loads in flight per lane: the copy of a ~100 KB blob costs one L2 round trip,
        // not one per 16 KB pass of the work-group)
        const uint4 *src = reinterpret_cast<const uint4 *>(P.blob);
        uint4 *dst       = reinterpret_cast<uint4 *>(lds_raw);
        const unsigned n16 = P.blob_bytes / 16, step = blockDim.x;
        for (unsigned i0 = threadIdx.x; i0 < n16; i0 += 8 * step) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned i = i0 + (unsigned) u * step;
                v[u]             = src[i < n16 ? i : i0];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned i = i0 + (unsigned) u * step;
                if (i < n16)
                    dst[i] = v[u];
            }
        }
        __syncthreads();
    }
}

// MODE fixes the two run-time switches of the march at compile time where the host knows them (1: emission, backward
// method -- every ASE run; 2: gain only, forward method -- every seeded run of create_image; 0: as DevParams says): as
// run-time booleans they live in SGPR pairs that the register allocator spills into VGPR lanes and reads back
// (v_readlane + wait states) in the cell set-up and the retirement block, i.e. in almost every iteration.
template <bool LDS_TAB, bool BOUNDED, bool FUSED, int MODE = 0>
__device__ __forceinline__ void march_wave(const DevParams &P, unsigned char *lds_raw, const TileList done)
{
    const int lane        = lane_id();
    const int L           = P.L;
    const int S           = L * RT_N_SUB;
    const unsigned n_rays = P.ray_end; // this launch marches rays [P.ray_begin, P.ray_end)
    // (MODE 3 / 4: only one of the two fixed, for the gain-only / forward pair -- experiments, see rt_launch.hip)
    const bool backward   = MODE == 1 ? true : ((MODE == 2 || MODE == 3) ? false : P.method == 1);
    const bool use_emis   = MODE == 1 ? true : ((MODE == 2 || MODE == 4) ? false : P.use_emis != 0);
    // (the path tracer is a debugging run: it takes the generic instance, the others carry none of its code)
    const bool path_on    = MODE == 0 ? P.path_on != 0 : false;
    const unsigned CH     = P.chunk;
#ifndef RT_REFILL
#define RT_REFILL 8
#endif
    const int REFILL      = RT_REFILL; // refill when this many lanes are idle (or the wave is empty)

    // ---- tables: the march blob, in LDS (copied by march_load_tables, which the caller has run) or in place ----
    const unsigned char *tab = LDS_TAB ? lds_raw : P.blob;
    const BlobGain *hdr = reinterpret_cast<const BlobGain *>(tab);
    // LDS byte address of the blob, for the hand-written reads of block [A2]
    const unsigned lds_base = LDS_TAB ? (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) lds_raw : 0u;

    // ends of the three sub-segments of a segment, Helper.h:456: dz * (iz + 1) / N_sub
    static_assert(RT_N_SUB == 3, "sub-segment ends are tabulated for three sub-segments");
    // (made opaque: otherwise the optimiser turns the later select of quotients back into a
    // division of selected dividends)
    float zs0 = (P.dz0 * (0.0f + 1.0f) / RT_N_SUB), zs1 = (P.dz0 * (1.0f + 1.0f) / RT_N_SUB),
          zs2 = (P.dz0 * (2.0f + 1.0f) / RT_N_SUB);
    asm volatile("" : "+v"(zs0), "+v"(zs1), "+v"(zs2));

    unsigned chunk_next = 0, chunk_end = 0; // wave-uniform window of reserved ray indices
    // FUSED: the window is the part of ONE tile not handed out yet; the reservation it was cut from ends at fetched_end
    unsigned fetched_end = 0, slot_busy = 0, cur_slot = 0; // (wave-uniform) slots of done.rem in use; slot of the window's tile
    unsigned cslot       = 0;                               // (per lane) slot of the tile of the lane's ray
    bool more           = true;             // wave-uniform: the ray counters are not exhausted
    // The rays of a launch are handed out in chunks of CH by eight counters (shard sh owns the chunks sh, sh + 8,
    // ...; a wave starts on shard blockIdx.x % 8 and moves on when its own is empty): one counter serves ~88
    // returning atomics per microsecond chip-wide, which a launch of a few hundred thousand rays comes close to
    // (rt_freq.hip has the measurements)
    const unsigned n_launch_rays = n_rays - P.ray_begin;
    const unsigned n_chunks      = (n_launch_rays + CH - 1) / CH;
    unsigned shard = blockIdx.x & 7u, shards_seen_empty = 0;
    // The end of a launch.  The last rays handed out set when the march ends -- their length times the time of an
    // iteration -- and the instruction arbiter of a SIMD serves its oldest waves first (tools/wave_trace.py: the
    // marching waves that stall for 100 us while the rest of their SIMD runs the frequency pass are the highest wave
    // numbers of their work-group, whatever s_setprio says).  So the last P.late_chunks chunks of the list are kept for
    // the first P.late_waves waves of each work-group, the oldest wave of every SIMD: the other waves see the counters
    // dry that much earlier, drain and turn to the frequency pass while the old waves are still busy, and the final
    // drain is run by one wave per SIMD that nothing on its SIMD outranks.  Zone 0: chunks [0, n_main), zone 1: the rest,
    // with counters of its own (next_tile[..][shard][8]).
    const unsigned n_main = n_chunks - (P.late_chunks < n_chunks ? P.late_chunks : 0u);
    // (unsigned: waves late_first .. late_first + late_waves - 1; through readfirstlane, so that the compiler knows it for
    // wave-uniform -- derived from threadIdx.x it counts as divergent, and with it `zone`, `more` and every branch of the
    // loop head that tests them, which then run as exec-mask code instead of scalar branches)
    const bool late_wave  = (unsigned) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) - P.late_first < P.late_waves;
    unsigned zone = 0;

    // ---- per-lane state ----
    int st        = ST_IDLE;
    unsigned ridx = 0;
    int seg = 0, iz = 0, ii = 0;
    float z = 0.0f, z_stop = 0.0f;
    float px = 0, py = 0, pz = 0, sx = 0, sy = 0, sz = 1;
    float gacc = 0, eacc = 0;
    int cell_last = 0, sub = 0; // sub: sub-segments committed so far (slots written), 0 .. S
    unsigned char *recp = P.rec; // slot of the lane's current sub-segment inside its ray's record
    unsigned lane12     = 0;     // 12 x (the ray's place in its tile): what the meta block lies beyond the tile's slot rows
    BlobGain G    = hdr[1]; // header of the lane's current length ii, re-read only when ii changes
    unsigned steps = 0;
    bool escaped = false, mirror = false;
    unsigned any_bits = 0; // OR of the magnitude bits of every committed gain / emission sum: 0 <=> all were zero
    // cell
    int c00 = 0;                              // index of the lower-left corner node of the current cell
    // refractive index at the cell's four corners (gathered in [A2] with g0, E0), as block [B] uses it:
    // rounded to float (Helper.h:332) and as the four f64 edge differences of Helper.h:333-334
    float f00 = 1, f10 = 1, f01 = 1, f11 = 1;
    double dnx0 = 0, dnx1 = 0, dny0 = 0, dny1 = 0; // n10 - n00, n11 - n01, n01 - n00, n11 - n10
    // the two interval records of the current cell as block [A2] reads them, 16 bytes at a time (rt_device.h,
    // Interval): {lo = lower-left corner coordinate, rw = 1 / (double) w} and {w, b_lo, b_hi, -} per axis
    f64x2 ix0 = { 0.0, 1.0 }, iy0 = { 0.0, 1.0 };
    f32x4 ix1 = { 1.0f, 0.0f, 0.0f, 0.0f }, iy1 = { 1.0f, 0.0f, 0.0f, 0.0f };
    float g0 = 0, E0 = 0;
    float dzrem = 0, zc = 0, path = 0;
    // integrator
    float rx = 0, ry = 0, rz = 0, n = 0, n0 = 0, gxn = 0, gyn = 0, lim2 = 0, dzcap = 0, hsum = 0;
    // totals of this lane over the whole launch
    unsigned tot_steps = 0, tot_esc = 0, tot_rays = 0, tot_skip = 0;
#ifdef RT_INSTRUMENT
    Inst inst;
#endif

#if defined(RT_TIMEBLOCKS) || defined(RT_WAVEBLOCKS)
    unsigned long long tb_acc[6] = { 0, 0, 0, 0, 0, 0 }, tb_last = __builtin_readcyclecounter(), tb_iters = 0;
#endif
#ifdef RT_WAVETIMES
    const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_dry = 0;
    unsigned wt_a_runs = 0, wt_b_runs = 0;
#endif
#ifdef RT_INSTRUMENT
    unsigned ray_iters = 0;
#endif
    unsigned spin = 0; // (BOUNDED = false only)
    // Express waves.  A launch cannot end before its longest ray has taken its last step, and a ray's steps are
    // sequential: what the launch can do is run the waves that hold the old rays FAST.  `wave_iter` counts this wave's
    // loop iterations, `born` is its value when the lane took its ray; a wave that holds a ray older than P.express_age
    // iterations raises its priority (the instruction arbiter of a SIMD serves priority before age, MI355X_MICROARCH.md)
    // and, with P.express_hold, stops fetching rays from the counters while it does: its short rays retire, fewer lanes
    // mean fewer of the three blocks per iteration, and the long rays it holds advance at the pace of a wave that has
    // its SIMD to itself.  Looked at every fourth iteration; everything here is wave-uniform.
    // (Measured in round 5, profiles/r05_express_ab.txt: +-1 ... 5 % -- what ends a launch is not the old rays, rt_fused.hip.
    // The code is compiled only into the diagnostic build librt_hip_express.so, -DRT_EXPRESS: the loop head is sensitive to
    // every instruction.)
#if defined(RT_EXPRESS) || defined(RT_WAVETIMES)
    unsigned wave_iter = 0;
#endif
#ifdef RT_EXPRESS
    unsigned born = 0;
    bool old_wave = false;
    if (P.express_tail == 1u)
        __builtin_amdgcn_s_setprio(2);
#else
    constexpr bool old_wave = false;
#endif
    for (;;) {
        RT_MARK(5); // [C] of the previous iteration
#ifdef RT_INSTRUMENT
        ray_iters += st != ST_IDLE ? 1u : 0u;
#endif
#ifdef RT_TIMEBLOCKS
        tb_iters++;
#endif
        // ------------------------------------------------------------ refill
        // (one ballot per iteration: the lane masks of the loop head come from `idle2`, which is taken again only when
        // a refill has changed the states; a wave all of whose lanes are idle with nothing left to fetch leaves below)
        unsigned long long idle2      = __ballot(st == ST_IDLE);
        const unsigned long long idle = idle2;
        const int n_idle              = (int) __popcll(idle);
#if defined(RT_EXPRESS) || defined(RT_WAVETIMES)
        wave_iter++;
#endif
#ifdef RT_WAVETIMES
        if ((wave_iter & 15u) == 0u) {
            const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), smp = wave_iter >> 4;
            if (lane == 0 && wid < 8192u && smp < (unsigned) WT_TRACE) {
                g_wt_trace[wid][smp] = (__builtin_amdgcn_s_memrealtime() & 0xffffffffffull) | ((unsigned long long) (WAVE - n_idle) << 40) |
                                       ((unsigned long long) wt_a_runs << 48) | ((unsigned long long) wt_b_runs << 56);
                g_wt_trace[wid][0] = smp;
#ifdef RT_WAVEBLOCKS
                for (int b = 0; b < 6; b++)
                    g_wt_blocks[wid][smp][b] = (unsigned) tb_acc[b];
#endif
            }
#ifdef RT_WAVEBLOCKS
            for (int b = 0; b < 6; b++)
                tb_acc[b] = 0;
#endif
            wt_a_runs = wt_b_runs = 0;
        }
#endif
#ifdef RT_EXPRESS
        if (P.express_age != 0u && (wave_iter & 3u) == 0u) {
            const bool o = __ballot(st != ST_IDLE && wave_iter - born > P.express_age) != 0ull;
            if (o != old_wave) {
                old_wave = o;
                if (o)
                    __builtin_amdgcn_s_setprio(3);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
        }
#endif
        // Watchdog of the instance that takes tables and step sizes as they come (BOUNDED = false: something is outside
        // the ranges rt_hip_plan_create verifies -- an index far from 1, a gradient beyond 1e12, a dz beyond 1e6 cm).
        // The reference's loops have no iteration limit, and with such inputs a ray's steps can stop advancing (an
        // infinite dz in a medium without refraction: 1250 cm per step towards z = inf); on the CPU that is a busy
        // process, here it would be a hung device.  `spin` counts the wave's iterations since it last took rays (while the
        // counters have rays that is every few dozen iterations; afterwards a wave lives as long as its longest ray); a
        // wave that reaches P.spin_limit -- looked at every 4096 iterations -- gives its rays up: escaped, direction
        // zeroed, so that they commit what they have, retire through [A1] and are reported as invalid rays (error -1)
        // by the frequency pass, as rays with a NaN start are.  The BOUNDED instance carries no such counter (it costs
        // 1 - 2 % of the march): inside the verified ranges every step advances by a bounded fraction of a bounded box.
        if (!BOUNDED) {
            if (((++spin) & 0xfffu) == 0u && spin >= P.spin_limit) {
                spin = 0;
                if (st != ST_IDLE) {
                    escaped = true;
                    sx = sy = sz = 0.0f;
                    st           = ST_CELL;
                }
            }
        }
        if (more && (n_idle >= REFILL)) {
            const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (idle >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned) idle, 0u));
            int need = n_idle, off = 0;
            bool got = false;
            while (need > 0) {
                if (chunk_next == chunk_end && !(FUSED && chunk_next != fetched_end)) {
#ifdef RT_EXPRESS
                    if (old_wave && P.express_hold != 0u)
                        break; // an express wave takes no new reservation (what it has reserved it still hands out)
#endif
                    unsigned c = 0;
#ifdef RT_MARCH_ONE_COUNTER // experiment: one counter for all waves
                    if (lane == 0)
                        c = atomicAdd(&P.ctl->next_tile[P.launch_id][0][0], 1u);
                    c = (unsigned) __builtin_amdgcn_readfirstlane((int) c);
                    shards_seen_empty = 7;
#else
                    if (lane == 0)
                        c = atomicAdd(&P.ctl->next_tile[P.launch_id][shard][zone * 8u], 1u);
                    c = (unsigned) __builtin_amdgcn_readfirstlane((int) c) * 8u + shard + zone * n_main; // chunk number
#endif
                    if (c >= (zone ? n_chunks : n_main)) { // this shard is empty: the next one, until all eight have been seen empty
                        if (++shards_seen_empty == 8) {
                            if (zone == 0u && late_wave && n_main < n_chunks) { // on to the chunks kept for the old waves
                                zone              = 1u;
                                shards_seen_empty = 0;
                                continue;
                            }
                            more = false;
#ifdef RT_EXPRESS
                            if (P.express_tail == 2u)
                                __builtin_amdgcn_s_setprio(3);
#endif
#ifdef RT_WAVETIMES
                            wt_dry = __builtin_amdgcn_s_memrealtime();
#endif
                            break;
                        }
                        shard = (shard + 1) & 7u;
                        continue;
                    }
                    chunk_next = P.ray_begin + c * CH;
                    chunk_end  = (n_rays - chunk_next < CH) ? n_rays : chunk_next + CH;
                    if (FUSED) {
                        fetched_end = chunk_end;
                        chunk_end   = chunk_next; // (no tile of the reservation is open yet)
                    }
                }
                if (FUSED && chunk_next == chunk_end) {
                    // open the next tile of the reservation: a free counter slot, loaded with the tile's ray count
#ifndef RT_FUSED_PUBLISH_WAVE
                    // (a slot is free when its counter is back at zero: the lane that took it there published the tile in
                    // the same breath, below; looked up here, once per tile, instead of being tracked in every iteration)
                    const unsigned in_use = lane < 32 ? __hip_atomic_load(&done.rem[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 1u;
                    slot_busy             = ~(unsigned) __ballot(lane < 32 && in_use == 0u);
#endif
                    const int fs = __builtin_ffs((int) ~slot_busy) - 1;
                    if (fs < 0)
                        break; // 32 tiles of this wave in flight (never observed): no new rays until one completes
                    chunk_end = fetched_end - chunk_next < (unsigned) WAVE ? fetched_end : chunk_next + (unsigned) WAVE;
                    cur_slot  = (unsigned) fs;
                    slot_busy |= 1u << fs;
                    if (lane == 0)
                        __hip_atomic_store(&done.rem[fs], chunk_end - chunk_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                int avail = (int) (chunk_end - chunk_next);
                int take  = avail < need ? avail : need;
                if (st == ST_IDLE && !got && rank >= off && rank < off + take) {
                    ridx = chunk_next + (unsigned) (rank - off);
                    got  = true;
                    if (FUSED)
                        cslot = cur_slot;
                }
                chunk_next += (unsigned) take;
                need -= take;
                off += take;
            }
            // (the watchdog restarts only when the wave really took rays: a refill that hands out nothing -- the ray
            // counters dry for this wave, or, FUSED, all 32 tile slots held open by tiles with a ray that does not
            // advance -- must not keep resetting it, or a wave with eight idle lanes would never reach the limit)
            if (!BOUNDED && __ballot(got) != 0ull)
                spin = 0;
            if (got) {
                // Helper.h:404-418
                rt_ray ray;
                float ta = 0, tb = 0;
                load_ray(P.rays, ridx, ray, ta, tb, true);
                px = ray.x;
                py = ray.y;
                pz = 0.0f;
                sx = ta;
                sy = tb;
                sz = 1.0f;
                if (backward) {
                    sx = -sx;
                    sy = -sy;
                    sz = -sz;
                }
                renormalise(sx, sy, sz);
                // A NaN in the start of a ray (a NaN or infinite launch angle -- tanf gives NaN for both --, a NaN
                // position) that does not escape at once: every comparison of the escape test and of the cell box
                // fails on it, z never advances, and the reference's cell loop (Helper.h:463-504) spins for ever from
                // the second segment on.  Such a ray is reported as an invalid ray instead (error -1, what
                // Helper.h:515 reports for a direction it cannot use): no march, direction zeroed so that the
                // s.z^2 < 0.01 test of the frequency kernel sees it.  (A NaN ray that starts outside the plasma
                // escapes in the reference as well and is left alone.)
                const BlobGain G0 = hdr[backward ? P.N - 1 : 1];
                const float dsum  = sx + sy + sz;
                const bool wild   = ((px != px) | (py != py) | (dsum != dsum)) &
                                  !((px < G0.lo_x) | (px > G0.hi_x) | (py < G0.lo_y) | (py > G0.hi_y));
                if (path_on) { // Helper.h:419-426
                    float *pp = P.path + (size_t) ridx * 3 * (size_t) (S + 1) + 3 * (size_t) (backward ? S : 0);
                    pp[0]     = px;
                    pp[1]     = py;
                }
                // the record's slots are visited in marching order: up from slot 0, or down from slot S-1
                recp      = P.rec + rec_slot_off(ridx, backward ? S - 1 : 0, P.rec_stride);
                lane12    = (ridx & 63u) * 12u; // (kept for the meta store at retirement: that block runs in almost every iteration)
                seg       = 0;
                iz        = 0;
                ii        = backward ? P.N - 1 : 1;
                G         = G0;
                z         = 0.0f;
                z_stop    = zs0;
                gacc      = 0.0f;
                eacc      = 0.0f;
                cell_last = 0;
                steps     = 0;
                escaped   = wild; // (an escaped ray commits one empty slot and retires, block [A1])
                sx        = wild ? 0.0f : sx;
                sy        = wild ? 0.0f : sy;
                sz        = wild ? 0.0f : sz;
                any_bits  = P.no_skip; // (non-zero: this ray is never marked F_SKIP -- a lineshape table holds a NaN, see DevParams)
                sub       = 0;
                st        = ST_CELL;
#ifdef RT_EXPRESS
                born      = wave_iter;
#endif
            }
            idle2 = __ballot(st == ST_IDLE);
        }
        // (at this point every lane is idle, waits for [A], or is in [B] / [C]: the three lane masks the loop
        // head needs follow from two ballots)
        if (idle2 == ~0ull) {
            if (!more)
                break;
        } else {

        RT_MARK(0); // refill
        // ------------------------------------------------------------ [A] cell loop (Helper.h:430-504)
        // Straight-line: at most one sub-segment end [A1] and one cell setup [A2] per wave
        // iteration (a lane that needs more -- several empty sub-segments in a row, or the
        // commit after an escape -- simply comes back next iteration).
        // Lane parking: block [A] costs as much as [C] but only a third of the lanes need it in any one
        // iteration; with P.park > 1 it runs only when at least that many lanes wait for it (or nobody
        // has anything else to do), the waiting lanes sit the iteration out.
        // (the threshold follows the number of lanes that still hold a ray: a fifth of them, at most
        // P.park -- in the tail of a launch, when a wave is down to a few rays, a fixed threshold would make
        // each of them wait for all the others: 1024 rays took 0.34 ms with it, as long as 400 000)
        // (while the ray counter still has rays the wave holds more than 64 - REFILL of them, a fifth of which is 12:
        // for P.park <= 12 the threshold is P.park itself, and the adaptive form runs only in the tail of the launch)
        const unsigned long long want_a = __ballot(st == ST_CELL);
        int park_at = (int) P.park;
        if (!more || old_wave || (int) P.park > 12 || REFILL > 8) {
            const int n_live = WAVE - (int) __popcll(idle2);
            const int fifth  = (n_live * 13 + 63) >> 6; // ~ n_live / 5, at least 1 for a live lane
            park_at          = fifth < (int) P.park ? fifth : (int) P.park;
            // an express wave holds long rays only: most of their iterations are integrator steps inside one cell, so
            // [A] is worth deferring much longer than in a wave of average rays (P.express_park 64ths of the live lanes)
#ifdef RT_EXPRESS
            if (old_wave && P.express_park != 0u)
                park_at = (n_live * (int) P.express_park + 63) >> 6;
#endif
        }
        const bool do_a = (int) __popcll(want_a) >= park_at || (~(idle2 | want_a)) == 0ull;
#ifdef RT_FUSED_PUBLISH_WAVE
        bool last_of_tile = false; // FUSED: the lane's ray ends in this iteration and was the last one of its tile
#endif
#ifdef RT_WAVETIMES
        wt_a_runs += (do_a && want_a != 0ull) ? 1u : 0u;
        wt_b_runs += __ballot(st == ST_XSETUP) != 0ull ? 1u : 0u; // (lanes that arrive from [A2] in this iteration not counted)
#endif
        if (do_a && st == ST_CELL) {
            bool in_seg        = !escaped & (z < 0.995f * z_stop);
            if (!in_seg) {
                // [A1] end of this sub-segment: commit its slot (Helper.h:501-503 accumulate from 0),
                // slot (ii - 1) * 3 + (backward ? 2 - iz : iz); straight-line, selects instead of branches
#ifndef RT_ABL_NOSTORE
                *reinterpret_cast<RecSlot *>(recp) = RecSlot{ gacc, eacc, cell_last };
#endif
                recp += backward ? -(int) REC_SLOT_ROW : (int) REC_SLOT_ROW; // the same ray's next slot: one slot row on
                any_bits |= (__float_as_uint(gacc) | __float_as_uint(eacc)) & 0x7fffffffu;
                if (path_on) { // Helper.h:505-511: every remaining sub-segment of an escaped ray's
                                 // segment records the same position, later segments stay zero
                    float *pp = P.path + (size_t) ridx * 3 * (size_t) (S + 1);
                    for (int zz = iz; zz < (escaped ? RT_N_SUB : iz + 1); zz++) {
                        const int idx = RT_N_SUB * (ii - 1) + (backward ? RT_N_SUB - zz - 1 : zz + 1);
                        pp[3 * idx]     = px;
                        pp[3 * idx + 1] = py;
                    }
                }
                gacc      = 0.0f;
                eacc      = 0.0f;
                cell_last = 0;
                sub++;
                // an escaped ray never enters its remaining sub-segments: `sub` tells the readers
                // (RecMeta n_done); otherwise on to the next sub-segment, or the next segment
                const bool wrap = iz == RT_N_SUB - 1;
                const bool fin  = escaped | (sub == S);
                iz              = wrap ? 0 : iz + 1;
                z               = wrap ? 0.0f : z;
                st              = fin ? ST_DONE : ST_CELL;
                if (wrap & !fin) { // twice per ray
                    seg++;
                    ii = backward ? P.N - seg - 1 : seg + 1;
                    G  = hdr[ii];
                }
                z_stop = iz == 0 ? zs0 : (iz == 1 ? zs1 : zs2);
                in_seg = !fin & (z < 0.995f * z_stop);
            }
            RT_MARK(1); // [A1]
            if ((st == ST_CELL) & in_seg) {
                // [A2] escape test + cell setup (Helper.h:465-497)
                // (double)(sz*sz) < 0.01 (Helper.h:466) <=> sz*sz <= 0.01f: 0.01f is the largest float below 0.01
                if ((px < G.lo_x) | (px > G.hi_x) | (py < G.lo_y) | (py > G.hi_y) | (sz * sz <= 0.01f)) {
                    escaped = true; // its slot is committed by [A1] next iteration
                } else {
                    mirror              = G.mirror_y != 0;
                    const Interval *ivx = reinterpret_cast<const Interval *>(tab + G.off_ix);
                    const Interval *ivy = reinterpret_cast<const Interval *>(tab + G.off_iy);
                    // (records are addressed by 32-bit byte offsets from the start of the blob: no
                    // 64-bit multiplies for what is an LDS address)
                    // One round of gathers per cell: two 48-byte interval records and four 16-byte corner nodes (the
                    // two nodes of a grid row are neighbours), ten 16-byte reads issued together.  In the LDS variant
                    // they are written as ds_read_b128 instructions whose results ARE the lane-state registers
                    // (ix0, ix1, iy0, iy1) or are consumed where they land: left to itself the compiler reads the tail
                    // of a record as ds_read_b96 and a node as b64 + b32 + b32 (issue cost per LDS instruction: b32 /
                    // b64 5, b128 10, b96 20 VALU-equivalent ticks -- tools/ubench), and held to whole 16-byte reads
                    // by empty asm statements (round 3) it paid for them with 34 register copies per cell set-up.
                    f64x2 tx, ty;            // {hi, rh} of the two intervals
                    u32x4 q00, q10, q01, q11; // corner nodes {n (f64) | g0 | E0}
                    auto gather = [&](int k1, int k2, int c) {
                        const unsigned ox = (unsigned) G.off_ix + (unsigned) k1 * (unsigned) sizeof(Interval);
                        const unsigned oy = (unsigned) G.off_iy + (unsigned) k2 * (unsigned) sizeof(Interval);
                        const unsigned oa = (unsigned) G.off_node + (unsigned) c * (unsigned) sizeof(Node);
                        const unsigned ob = oa + (unsigned) G.Nx * (unsigned) sizeof(Node);
                        if (LDS_TAB) {
                            // (lds_raw starts at LDS address 0: the kernel has no static LDS; the reads complete
                            // inside the statement, so the compiler's own wait counters stay right)
                            asm volatile("ds_read_b128 %0, %10\n\t"
                                         "ds_read_b128 %1, %10 offset:16\n\t"
                                         "ds_read_b128 %2, %10 offset:32\n\t"
                                         "ds_read_b128 %3, %11\n\t"
                                         "ds_read_b128 %4, %11 offset:16\n\t"
                                         "ds_read_b128 %5, %11 offset:32\n\t"
                                         "ds_read_b128 %6, %12\n\t"
                                         "ds_read_b128 %7, %12 offset:16\n\t"
                                         "ds_read_b128 %8, %13\n\t"
                                         "ds_read_b128 %9, %13 offset:16\n\t"
                                         "s_waitcnt lgkmcnt(0)"
                                         : "=&v"(ix0), "=&v"(ix1), "=&v"(tx), "=&v"(iy0), "=&v"(iy1), "=&v"(ty), "=&v"(q00), "=&v"(q10),
                                           "=&v"(q01), "=&v"(q11)
                                         : "v"(lds_base + ox), "v"(lds_base + oy), "v"(lds_base + oa), "v"(lds_base + ob));
                        } else {
                            const unsigned char *t = tab;
                            ix0 = *reinterpret_cast<const f64x2 *>(t + ox);
                            ix1 = *reinterpret_cast<const f32x4 *>(t + ox + 16);
                            tx  = *reinterpret_cast<const f64x2 *>(t + ox + 32);
                            iy0 = *reinterpret_cast<const f64x2 *>(t + oy);
                            iy1 = *reinterpret_cast<const f32x4 *>(t + oy + 16);
                            ty  = *reinterpret_cast<const f64x2 *>(t + oy + 32);
                            q00 = *reinterpret_cast<const u32x4 *>(t + oa);
                            q10 = *reinterpret_cast<const u32x4 *>(t + oa + 16);
                            q01 = *reinterpret_cast<const u32x4 *>(t + ob);
                            q11 = *reinterpret_cast<const u32x4 *>(t + ob + 16);
                        }
                    };
                    auto node_n = [](const u32x4 &q) { return __hiloint2double((int) q.y, (int) q.x); };
                    const float ya      = mirror ? fabsf(py) : py;
                    const double pxd = (double) px, yad = (double) ya;
                    // one round of gathers on the guessed cell: two interval records, four nodes
                    int k1     = guess_interval(G.Nx, G.x0f, G.inv_hxf, px);
                    int k2     = guess_interval(G.Ny, G.y0f, G.inv_hyf, ya);
                    c00        = (k1 - 1) + (k2 - 1) * G.Nx;
                    gather(k1, k2, c00);
                    const bool ok = ((k1 == 1) | (ix0.x < pxd)) & ((k1 == G.Nx - 1) | (tx.x >= pxd)) &
                                    ((k2 == 1) | (iy0.x < yad)) & ((k2 == G.Ny - 1) | (ty.x >= yad));
                    if (!ok) {
                        k1  = bisect_interval(ivx, G.Nx, pxd);
                        k2  = bisect_interval(ivy, G.Ny, yad);
                        c00 = (k1 - 1) + (k2 - 1) * G.Nx;
                        gather(k1, k2, c00);
                    }
                    const double n00 = node_n(q00), n10 = node_n(q10), n01 = node_n(q01), n11 = node_n(q11);
                    f00       = (float) n00;
                    f10       = (float) n10;
                    f01       = (float) n01;
                    f11       = (float) n11;
                    dnx0      = n10 - n00;
                    dnx1      = n11 - n01;
                    dny0      = n01 - n00;
                    dny1      = n11 - n10;
                    const float u = (float) div_by_recip<true>(pxd - ix0.x, tx.x - ix0.x, tx.y);
                    const float v = (float) div_by_recip<true>(yad - iy0.x, ty.x - iy0.x, ty.y);
                    g0            = lerp2(u, v, __uint_as_float(q00.z), __uint_as_float(q10.z), __uint_as_float(q01.z),
                                          __uint_as_float(q11.z));
                    E0            = 0.0f;
                    if (use_emis) {
                        E0 = lerp2(u, v, __uint_as_float(q00.w), __uint_as_float(q10.w), __uint_as_float(q01.w),
                                   __uint_as_float(q11.w));
                        E0 = E0 >= 0 ? E0 : 0.0f;
                    }
                    pz    = 0.0f;
                    zc    = 0.0f;
                    path  = 0.0f;
                    dzrem = z_stop - z;
                    // Helper.h:327 with zc = 0: 0.0 < 0.999 * (double) dzrem <=> dzrem > 0 (no underflow in double)
                    if ((px > ix1.y) & (px < ix1.z) & (ya > iy1.y) & (ya < iy1.z) & (dzrem > 0.0f)) {
                        st = ST_XSETUP;
                    } else {
                        // no cross-cell iteration at all: the cell step still counts (Helper.h:498-503)
                        z += fabsf(pz);
                        gacc += g0 * path;
                        eacc += E0 * path;
                        cell_last = c00;
                        steps++;
                        RT_TICK(2);
                    }
                }
            }
            RT_MARK(2); // [A2]
            // ---------------------------------------------------------- ray finished
            if (st == ST_DONE) {
                unsigned fl = F_VALID;
                if (escaped)
                    fl |= F_ESCAPED;
                if (use_emis && any_bits == 0u)
                    fl |= F_SKIP; // every frequency update is the identity: contributes exactly +0 (finite lineshape)
                RecMeta m;
                m.px          = px;
                m.py          = py;
                m.sx          = sx;
                m.sy          = sy;
                m.sz          = sz;
                // (the per-ray step count of the record saturates at 2^20 - 1; the launch total below is exact)
                m.flags_steps = fl | ((unsigned) sub << REC_NDONE_SHIFT) |
                                ((steps < 0xfffffu ? steps : 0xfffffu) << REC_STEPS_SHIFT);
                // `sub` slots were committed: recp stands that many slot rows above slot 0 (forward) or below
                // slot S-1 (backward); the meta blocks of the tile follow slot row S-1, 24 bytes per ray where a slot has 12
                unsigned char *metap = recp + (int) REC_SLOT_ROW * (backward ? sub + 1 : S - sub) + lane12;
#ifndef RT_ABL_NOMETA
                *reinterpret_cast<RecMeta *>(metap) = m;
#else
                if (m.px == 1234.5f) // profiling only: the store stays reachable, but never happens
                    *reinterpret_cast<RecMeta *>(metap) = m;
#endif
#ifdef RT_INSTRUMENT
                if (ridx < (1u << 23))
                    g_ray_iters[ridx] = (unsigned short) (ray_iters < 65535u ? ray_iters : 65535u);
                ray_iters = 0;
#endif
                tot_steps += steps;
                tot_esc += escaped ? 1u : 0u;
                tot_skip += (fl & F_SKIP) ? 1u : 0u;
                tot_rays++;
                st = ST_IDLE;
#ifdef RT_FUSED_PUBLISH_WAVE
                if (FUSED) // one ray of the lane's tile less; the lane that takes the counter to zero has finished the tile
                    last_of_tile = __hip_atomic_fetch_add(&done.rem[cslot], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 1u;
#else
                // One ray of the lane's tile less; the lane that takes the counter to zero has finished the tile and
                // publishes it on the spot: every record of the tile was stored by THIS wave (the other rays retired in
                // earlier iterations or in this very block), so once the wave's stores have landed the tile may be read.
                // (Round 4 collected those lanes with a ballot after the block -- in every iteration, for an event that
                // happens once in ~35: the loop head pays for every instruction, profiles/r05_express_cost.txt.)
                if (FUSED) {
                    if (__hip_atomic_fetch_add(&done.rem[cslot], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 1u) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        tile_publish(done, ridx >> 6);
                    }
                }
#endif
            }
        }
#ifdef RT_FUSED_PUBLISH_WAVE
        if (FUSED) {
            unsigned long long lm = __ballot(last_of_tile);
            if (lm != 0ull) {
#ifdef RT_WAVETIMES
                const unsigned long long vm0 = __builtin_amdgcn_s_memrealtime();
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's record stores have landed
#ifdef RT_WAVETIMES
                if (lane == 0)
                    g_wt_vm[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u] += __builtin_amdgcn_s_memrealtime() - vm0;
#endif
                do {
                    const int l0 = (int) __ffsll((long long) lm) - 1;
                    lm &= lm - 1ull;
#ifdef RT_WAVETIMES
                    const unsigned long long pub0 = __builtin_amdgcn_s_memrealtime();
#endif
                    if (lane == l0)
                        tile_publish(done, ridx >> 6);
#ifdef RT_WAVETIMES
                    if (lane == l0) {
                        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - pub0;
                        unsigned long long *q = g_wt_pub[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u];
                        q[0] += dt;
                        q[1] += 1;
                        q[3] = dt > q[3] ? dt : q[3];
                    }
#endif
                    slot_busy &= ~(1u << (unsigned) __builtin_amdgcn_readlane((int) cslot, l0));
                } while (lm != 0ull);
            }
        }
#endif
        RT_MARK(3); // DONE
        // ------------------------------------------------------------ [B] cross-cell setup (Helper.h:328-342)
        if (st == ST_XSETUP) {
            const float ya   = mirror ? fabsf(py) : py;
            const double dwx = (double) ix1.x, dwy = (double) iy1.x; // Helper.h:323-324
            const double xc0 = ix0.x, yc0 = iy0.x, rwx = ix0.y, rwy = iy0.y;
            const float u    = (float) div_by_recip<true>((double) px - xc0, dwx, rwx);
            const float v    = (float) div_by_recip<true>((double) ya - yc0, dwy, rwy);
            n0  = lerp2(u, v, f00, f10, f01, f11);
            gxn = (float) (div_by_recip<true>((1.0 - (double) v) * dnx0, dwx, rwx) +
                           div_by_recip<true>((double) v * dnx1, dwx, rwx));
            gyn = (float) (div_by_recip<true>((1.0 - (double) u) * dny0, dwy, rwy) +
                           div_by_recip<true>((double) u * dny1, dwy, rwy));
            if (mirror && py < 0)
                gyn = -gyn;
            lim2  = dzrem - zc;
            dzcap = P.c_cap * lim2; // Helper.h:274: c * 1.00001f * dx[2]
            rx    = 0.0f;
            ry    = 0.0f;
            rz    = 0.0f;
            n     = n0;
            hsum  = 0.0f;
            st    = ST_STEP;
            RT_TICK(1);
        }

        RT_MARK(4); // [B]
        // ------------------------------------------------------------ [C] one integrator step (Helper.h:279-311)
        if (st == ST_STEP) {
            const float lim0 = 0.1f * ix1.x, lim1 = 0.1f * iy1.x;
            // The loop condition of Helper.h:279-280 holds for every lane that arrives here: a lane from [B]
            // has r = 0, n = n0 and limits that are positive (cell widths are, rt_hip_plan_create checks it;
            // lim2 = dzrem - zc > 0 is the entry condition of [B]); a lane that stays in this state was
            // tested at the end of its last step, below.  So the step runs unconditionally and the
            // condition is evaluated once, after it.
            bool run;
            {
                n        = n0 + rx * gxn + ry * gyn;
#ifdef RT_ABL_FASTDIV
#define RT_FDIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#else
#define RT_FDIV(a, b) ((a) / (b))
#endif
                // one IEEE division, three exact quotients
                const float rn = BOUNDED ? fdiv_one_nr(n) : RT_FDIV(1.0f, n);
                float a0 = sx * gxn + sy * gyn + 1e-12f;
                // (BOUNDED: a0 is never -0 -- x + 1e-12f is +0 or non-zero -- and rn > 0, so the corrected
                // quotient has the sign of a0 without the sign copy)
                float t  = BOUNDED ? fmaf(fmaf(-n, a0 * rn, a0), rn, a0 * rn) : div_by_recip_signed(a0, n, rn);
                float qx = div_by_recip_signed(gxn, n, rn);
                float qy = div_by_recip_signed(gyn, n, rn);
#ifndef RT_ABL_NOGUARD
                // div_by_recip needs a true division for non-zero dividends below 1e-29 (rt_math.h).  One
                // wave-uniform test covers the three quotients: the smallest magnitude of the dividends is
                // below the bound only where a lane holds an exact zero (index gradient of a uniform
                // region; the quotient 0 is right as it is) or, once in a blue moon, such a dividend.
                if (__ballot(fminf(fminf(fabsf(a0), fabsf(gxn)), fabsf(gyn)) < 1e-29f) != 0ull) {
#ifdef RT_INSTRUMENT
                    if (lane == (int) __ffsll((long long) __ballot(1)) - 1)
                        atomicAdd(&g_inst[6], 1ull); // wave-level entries of the tiny-dividend block
#endif
                    if (fabsf(a0) < 1e-29f && a0 != 0.0f) {
                        asm volatile("" : "+v"(a0));
                        t = a0 / n;
                    }
                    if (fabsf(gxn) < 1e-29f && gxn != 0.0f) {
                        float g = gxn;
                        asm volatile("" : "+v"(g));
                        qx = g / n;
                    }
                    if (fabsf(gyn) < 1e-29f && gyn != 0.0f) {
                        float g = gyn;
                        asm volatile("" : "+v"(g));
                        qy = g / n;
                    }
                }
#endif
                float fx = qx - sx * t;
                float fy = qy - sy * t;
                float fz = -sz * t;
                float h;
                if (BOUNDED) {
                    // The step candidates (Helper.h:288-297) without the range bookkeeping of an IEEE division
                    // (fdiv_nr, rt_math.h): exact wherever a candidate can be the minimum; a candidate that the
                    // short sequence gets wrong is huge, infinite or NaN on both paths (its true value is above
                    // dzcap), and the NaN-dropping minimum below passes over it as the `<` chain passes over +inf.
                    const float h1 = fdiv_nr(P.c_h1, fabsf(t)); // c * 0.1f / |t|
                    const float h2 = fdiv_nr(1.0001f * (lim2 - fabsf(rz)), fabsf(sz));
                    const float h3 = fdiv_nr(P.c_h3 * (fabsf(sx) + 5e-4f), (fabsf(fx) + 1e-8f)); // c * 0.05f * ...
                    const float h4 = fdiv_nr(P.c_h3 * (fabsf(sy) + 5e-4f), (fabsf(fy) + 1e-8f));
                    h = fmin_nan_drop(fmin_nan_drop(h1, dzcap), fmin_nan_drop(h2, fmin_nan_drop(h3, h4)));
                } else {
                    h        = RT_FDIV(P.c_h1, fabsf(t)); // c * 0.1f / |t|
                    h        = h < dzcap ? h : dzcap;
                    float h2 = RT_FDIV(1.0001f * (lim2 - fabsf(rz)), fabsf(sz));
                    float h3 = RT_FDIV(P.c_h3 * (fabsf(sx) + 5e-4f), (fabsf(fx) + 1e-8f)); // c * 0.05f * ...
                    float h4 = RT_FDIV(P.c_h3 * (fabsf(sy) + 5e-4f), (fabsf(fy) + 1e-8f));
                    h        = h < h2 ? h : h2;
                    h        = h < h3 ? h : h3;
                    h        = h < h4 ? h : h4;
                }
                float ht = h * t;
                const float R3 = 1.0f / 3.0f, R6 = 1.0f / 6.0f, R12 = 1.0f / 12.0f; // RN(1/b), folded
                float c1 = 0.5f * h * h * (1.0f - div_by_recip<true>(ht, 3.0f, R3) + div_by_recip<true>(ht * ht, 12.0f, R12));
                rx += sx * h + c1 * fx;
                ry += sy * h + c1 * fy;
                rz += sz * h + c1 * fz;
                float c2 = h * (1.0f - 0.5f * ht + div_by_recip<true>(ht * ht, 6.0f, R6));
                sx += c2 * fx;
                sy += c2 * fy;
                sz += c2 * fz;
                renormalise(sx, sy, sz);
                hsum += h;
                RT_TICK(0);
                // (double)|n - n0| < 0.05 (Helper.h:280) <=> |n - n0| < 0.05f: 0.05f is the smallest float above 0.05
                run = (fabsf(rx) < lim0) & (fabsf(ry) < lim1) & (fabsf(rz) < lim2) & (fabsf(n - n0) < 0.05f);
            }
            if (!run) {
                // integrator loop over: close this cross-cell iteration (Helper.h:343-348)
                path += hsum;
                px += rx;
                py += ry;
                pz += rz;
                zc += fabsf(rz);
                const float ya = mirror ? fabsf(py) : py;
                if ((px > ix1.y) & (px < ix1.z) & (ya > iy1.y) & (ya < iy1.z) & ((double) zc < 0.999 * (double) dzrem)) {
                    st = ST_XSETUP;
                } else {
                    // cross-cell loop over: close the cell step (Helper.h:499-503)
                    z += fabsf(pz);
                    gacc += g0 * path;
                    eacc += E0 * path;
                    cell_last = c00;
                    steps++;
                    RT_TICK(2);
                    st = ST_CELL;
                }
            }
        }
        } // some lane is marching
    }
#ifdef RT_EXPRESS
    if (P.express_age != 0u || P.express_tail != 0u)
        __builtin_amdgcn_s_setprio(0);
#endif

#ifdef RT_WAVETIMES
    if (lane == 0) {
        const unsigned long long wt_end = __builtin_amdgcn_s_memrealtime();
        atomicMin(&g_wt[0], wt_start);
        atomicMax(&g_wt[1], wt_start);
        const unsigned w = atomicAdd((unsigned *) &g_wt[6], 1u);
        if (w < 8192)
            g_wt_end[w] = wt_end | ((unsigned long long) (threadIdx.x >> 6) << 56), g_wt_dry[w] = wt_dry;
    }
#endif
    // ---- launch totals ----
    {
        unsigned s = wave_sum_u32(tot_steps), e = wave_sum_u32(tot_esc);
        unsigned k = wave_sum_u32(tot_skip), r = wave_sum_u32(tot_rays);
#ifndef RT_ABL_NOTOTALS
        if (lane == 0) {
            atomicAdd(&P.ctl->cell_steps, (unsigned long long) s);
            atomicAdd(&P.ctl->n_escaped, (unsigned long long) e);
            atomicAdd(&P.ctl->n_skipped, (unsigned long long) k);
            atomicAdd(&P.ctl->n_rays, (unsigned long long) r);
        }
#else
        if (lane == 0 && s + e + k + r == 0xffffffffu)
            atomicAdd(&P.ctl->cell_steps, 1ull);
#endif
#ifdef RT_TIMEBLOCKS
        if (lane == 0) {
            for (int i = 0; i < 6; i++)
                atomicAdd(&g_inst[i], tb_acc[i]);
            atomicAdd(&g_inst[7], tb_iters);
        }
#endif
#ifdef RT_INSTRUMENT
        for (int i = 0; i < 3; i++) {
            unsigned tw = wave_sum_u32(inst.w[i]), ta = wave_sum_u32(inst.a[i]);
            if (lane == 0) {
                atomicAdd(&g_inst[2 * i], (unsigned long long) tw);
                atomicAdd(&g_inst[2 * i + 1], (unsigned long long) ta);
            }
        }
#endif
    }
}

template <bool LDS_TAB, bool BOUNDED, int MODE = 0>
__global__ void __launch_bounds__(LDS_TAB ? 1024 : 256) rt_march_kernel(const DevParams P)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    march_load_tables<LDS_TAB>(P, lds_raw);
    march_wave<LDS_TAB, BOUNDED, false, MODE>(P, lds_raw, TileList{ nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, 0u, 0u, 0u });
}

} // namespace rt
