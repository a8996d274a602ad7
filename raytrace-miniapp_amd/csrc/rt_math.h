// rt_math.h -- device helpers shared by the kernels: wave utilities and the
// float32/float64 building blocks shared by the march and frequency kernels, each
// restating one helper of src/common/RayTraceImageHelper.h with its promotions
// spelled out.  Compiled with -ffp-contract=off: the march must take the
// reference's steps bit for bit.
#pragma once

#include "rt_device.h"

namespace rt {

#ifdef RT_INSTRUMENT
// Diagnostic build only (make instrument): lane-occupancy of the three nested
// march loops.  g_inst[2i] = wave-level iterations, g_inst[2i+1] = active-lane
// iterations, i = 0 inner (Helper.h:279), 1 cross (:326), 2 cell (:463).
__device__ unsigned long long g_inst[8];
struct Inst {
    unsigned w[3] = { 0, 0, 0 }, a[3] = { 0, 0, 0 };
    __device__ __forceinline__ void tick(int i)
    {
        a[i]++;
        unsigned long long m = __ballot(1);
        if ((int) (threadIdx.x & 63) == __ffsll((long long) m) - 1)
            w[i]++;
    }
};
#define RT_TICK(i) inst.tick(i)
#else
#define RT_TICK(i)
#endif
// Second diagnostic build (-DRT_TIMEBLOCKS): wave clock spent in each block of the march
// loop, g_inst[i] = cycles between mark i and mark i+1 summed over waves, g_inst[7] = iterations.
#if defined(RT_TIMEBLOCKS) || defined(RT_WAVEBLOCKS)
#ifndef RT_INSTRUMENT
__device__ unsigned long long g_inst[8];
#endif
#define RT_MARK(i)                                                      \
    {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        tb_acc[i] += now_ - tb_last;                                    \
        tb_last = now_;                                                 \
    }
#else
#define RT_MARK(i)
#endif

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & (WAVE - 1)); }

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (u & 0xffffffffu), l);
    unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}

// ------------------------------------------------------------------ march math
// Helper.h:73-89.  The reference computes (float)(1.0 / (double)sqrtf(q)).  A
// correctly rounded float division 1.0f / y gives the same float for every
// float y (the double-rounded quotient can differ only if 1/y lies within
// 2^-54 of a 25-bit midpoint, which a 24-bit y cannot produce unless it is a
// power of two, where both are exact) -- checked exhaustively for y in
// [0.25, 4) by tests/test_float_identities.py.
//
// inv_norm computes that float without the two IEEE sequences (17 + 11 VALU) when q is near
// 1, where a direction vector's squared norm always is: y = v_rsq_f32(q) (1 ulp), one
// Markstein step for the root, s1 = RN(s0 + (q - s0^2) y/2) with s0 = RN(q y), one Newton
// step for its reciprocal from the same y, RN(y + y (1 - s1 y)); a root whose significand is
// all ones -- the one case the Newton step can miss, and a common one (q just below 1) -- has
// the closed form RN(1/s1) = bits(0x7f000000 - bits(s1)).  tests/test_float_identities.py
// checks the sequence against 1.0f / sqrtf(q) for every q in [0.9375, 1.0625) and every y
// within 1 ulp of 1/sqrt(q); rt_hip_selftest checks it on the device itself for every q.
__device__ __forceinline__ float inv_norm(float q)
{
    float inv;
#ifdef RT_ABL_FASTDIV
    inv = __builtin_amdgcn_rsqf(q);
#else
    {
        // the shortcut for every lane; the IEEE sequence replaces it behind ONE wave-uniform branch in
        // the (never observed) case that some lane's q lies outside the shortcut's range
        const float y    = __builtin_amdgcn_rsqf(q);
        const float s0   = q * y;
        const float e    = fmaf(-s0, s0, q);
        const float s1   = fmaf(e, 0.5f * y, s0); // = sqrtf(q)
        const float r    = fmaf(-s1, y, 1.0f);
        inv              = fmaf(r, y, y);
        const unsigned u = __float_as_uint(s1);
        inv              = ((u & 0x7fffffu) == 0x7fffffu) ? __uint_as_float(0x7f000000u - u) : inv;
        if (__ballot(!(fabsf(q - 1.0f) < 0.0625f)) != 0ull) {
            asm volatile("" : "+v"(q)); // keep the general case behind its branch
            if (!(fabsf(q - 1.0f) < 0.0625f))
                inv = 1.0f / sqrtf(q);
        }
    }
#endif
    return inv;
}
__device__ __forceinline__ void renormalise(float &sx, float &sy, float &sz)
{
    float q   = sx * sx + sy * sy + sz * sz;
    float inv = inv_norm(q);
    sx *= inv;
    sy *= inv;
    sz *= inv;
}

// Division by a divisor whose correctly rounded reciprocal y = RN(1/b) is known:
// q = RN(a*y), r = a - b*q (exact with FMA), result RN(q + r*y) = RN(a/b)
// (Markstein's correction step).  Checked against IEEE division on the CPU by
// tests/test_float_identities.py (1e9 random pairs, and every float for the
// constant divisors).  The correction term is at most half an ulp of q, so the result has
// the sign of q; copying it over preserves the sign of a zero quotient (-0 + +0 would give
// +0), the one thing the correction gets wrong.  Dividends so small that the residual
// leaves the normal range (|a| < 1e-29f, resp. 1e-280) take a true division -- unless the
// caller states
// that such a quotient cannot matter (TINY_OK):
//   * float, TINY_OK: the quotient is added to 1.0f (ht/3, ht^2/12, ht^2/6 in the
//     integrator step): below 2^-26 it is absorbed whatever its last bits are;
//   * double, TINY_OK: the quotient (or its sum with another such quotient) is narrowed
//     to float: with divisors above 1e-200 (plan_create checks the grid spacings) a
//     quotient of a dividend below 1e-280 lies below 1e-80 and narrows to a zero of
//     the right sign either way.
template <bool TINY_OK = false> __device__ __forceinline__ float div_by_recip(float a, float b, float y)
{
    const float q = a * y;
    const float r = fmaf(-b, q, a);
    // (TINY_OK callers add the quotient to 1.0f: the sign of a zero quotient cannot show)
    float c       = TINY_OK ? fmaf(r, y, q) : copysignf(fmaf(r, y, q), q);
#ifndef RT_ABL_NOGUARD
    if (!TINY_OK && fabsf(a) < 1e-29f && a != 0.0f) { // the residual must stay a normal float: |a| > 2^-102 (CPU test: none above 3e-32)
        asm volatile("" : "+v"(a)); // keep the rare true division behind its branch
        c = a / b;
    }
#endif
    return c;
}
// the unguarded float form with the sign of a zero quotient kept: for callers that test the dividends
// for the tiny range themselves (the integrator step's merged guard, rt_march.hip)
__device__ __forceinline__ float div_by_recip_signed(float a, float b, float y)
{
    const float q = a * y;
    const float r = fmaf(-b, q, a);
    return copysignf(fmaf(r, y, q), q);
}
template <bool TINY_OK = false> __device__ __forceinline__ double div_by_recip(double a, double b, double y)
{
    const double q = a * y;
    const double r = fma(-b, q, a);
    double c       = copysign(fma(r, y, q), q);
#ifndef RT_ABL_NOGUARD
    if (!TINY_OK && fabs(a) < 1e-280 && a != 0.0) { // residual normal: |a| > 2^-969
        asm volatile("" : "+v"(a)); // keep the rare true division behind its branch
        c = a / b;
    }
#endif
    return c;
}

// a / b as the hardware's IEEE sequence computes it (v_div_scale x2, v_rcp, five fma/mul, v_div_fmas,
// v_div_fixup) minus the scaling and fix-up instructions: the same reciprocal, Newton step, quotient and two
// residual corrections, hence the same float, whenever v_div_scale would leave both operands unscaled and
// v_div_fixup would pass the result through -- normal operands whose exponents differ by less than 96, a
// dividend of at least 2^-102, a divisor below 2^126, a normal quotient.  rt_hip_plan_create establishes those
// ranges for the integrator's divisions from the tables (DevParams::bounded); rt_hip_selftest compares the two
// sequences on the device over the ranges it relies on.
__device__ __forceinline__ float fdiv_nr(float a, float b)
{
    float r       = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r             = fmaf(e, r, r);
    float q       = a * r;
    q             = fmaf(fmaf(-b, q, a), r, q);
    return fmaf(fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ float fdiv_one_nr(float b) // 1.0f / b, likewise (the quotient a * r is r itself)
{
    float r       = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r             = fmaf(e, r, r);
    const float q = fmaf(fmaf(-b, r, 1.0f), r, r);
    return fmaf(fmaf(-b, q, 1.0f), r, q);
}
// min(a, b) that drops a NaN operand (v_min_f32, IEEE minNum): one instruction instead of compare + select
__device__ __forceinline__ float fmin_nan_drop(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Helper.h:153-158
__device__ __forceinline__ float lerp2(float u, float v, float f00, float f10, float f01, float f11)
{
    float u1 = 1.0f - u;
    float v1 = 1.0f - v;
    return (u * f10 + u1 * f00) * v1 + (u * f11 + u1 * f01) * v;
}

// Helper.h:101-117
__device__ __forceinline__ int first_not_below(const double *g, int n, double v)
{
    if (v < g[0])
        return 0;
    if (v > g[n - 1])
        return n;
    int lo = 0, hi = n - 1;
    while (hi - lo != 1) {
        int mid = (hi + lo) / 2;
        if (g[mid] >= v)
            hi = mid;
        else
            lo = mid;
    }
    return hi;
}

// Helper.h:168-220
__device__ double pchip_eval(int n, const double *xs, const double *ys, double x)
{
    if (x <= xs[0] || n <= 2) {
        double t = (x - xs[0]) / (xs[1] - xs[0]);
        return (1.0 - t) * ys[0] + t * ys[1];
    } else if (x >= xs[n - 1]) {
        double t = (x - xs[n - 2]) / (xs[n - 1] - xs[n - 2]);
        return (1.0 - t) * ys[n - 2] + t * ys[n - 1];
    }
    int i     = first_not_below(xs, n, x);
    double fl = ys[i - 1];
    double fr = ys[i];
    double t  = (x - xs[i - 1]) / (xs[i] - xs[i - 1]);
    double gl = 0, gr = 0;
    if (i <= 1) {
        gl = fr - fl;
    } else if ((fl < fr && fl > ys[i - 2]) || (fl > fr && fl < ys[i - 2])) {
        double fp   = ys[i - 2];
        double h1   = xs[i - 1] - xs[i - 2];
        double h2   = xs[i] - xs[i - 1];
        double w1   = (h2 - h1) / h1;
        double w2   = h1 / (h1 + h2);
        gl          = w1 * (fl - fp) + w2 * (fr - fp);
        double s1   = fabs(fl - fp) / h1;
        double s2   = fabs(fr - fl) / h2;
        double gmax = 2 * h2 * (s1 < s2 ? s1 : s2);
        gl          = ((gl >= 0) ? 1 : -1) * (fabs(gl) < gmax ? fabs(gl) : gmax);
    }
    if (i >= n - 1) {
        gr = fr - fl;
    } else if ((fr < fl && fr > ys[i + 1]) || (fr > fl && fr < ys[i + 1])) {
        double fn   = ys[i + 1];
        double h1   = xs[i] - xs[i - 1];
        double h2   = xs[i + 1] - xs[i];
        double w1   = -h2 / (h1 + h2);
        double w2   = (h2 - h1) / h2;
        gr          = w1 * (fl - fn) + w2 * (fr - fn);
        double s1   = fabs(fr - fl) / h1;
        double s2   = fabs(fn - fr) / h2;
        double gmax = 2 * h1 * (s1 < s2 ? s1 : s2);
        gr          = ((gr >= 0) ? 1 : -1) * (fabs(gr) < gmax ? fabs(gr) : gmax);
    }
    double t2 = t * t;
    return fl + t2 * (2 * t - 3) * (fl - fr) + t * gl - t2 * (gl + (1 - t) * (gl + gr));
}

// Helper.h:230-244: the frequency-independent factor f of the seed profile.
__device__ double seed_factor(const DevSeed &sd, double x, double y, double a, double b)
{
    double f = 0.0;
    if (x >= sd.x[0][0] && x <= sd.x[0][sd.dim[0] - 1] && y >= sd.x[1][0] && y <= sd.x[1][sd.dim[1] - 1] &&
        a >= sd.x[2][0] && a <= sd.x[2][sd.dim[2] - 1] && b >= sd.x[3][0] && b <= sd.x[3][sd.dim[3] - 1]) {
        double fx = pchip_eval(sd.dim[0], sd.x[0], sd.f[0], x);
        double fy = pchip_eval(sd.dim[1], sd.x[1], sd.f[1], y);
        double fa = pchip_eval(sd.dim[2], sd.x[2], sd.f[2], a);
        double fb = pchip_eval(sd.dim[3], sd.x[3], sd.f[3], b);
        f         = sd.f0 * fx * fy * fa * fb;
        f         = f < 0.0 ? 0.0 : f;
    }
    return f;
}


} // namespace rt
