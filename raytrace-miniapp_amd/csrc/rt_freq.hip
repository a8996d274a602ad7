#pragma once
// rt_freq.hip -- kernel B: frequency integration + deposit (Helper.h:515-594,
// RayTraceImageCPU.cpp:37-68) with lanes = rays.
//
// One wavefront owns a tile of 64 consecutive rays; lane r integrates ray r over
// all K frequencies, VEC = 4 frequencies at a time (four independent dependency chains
// per lane; the lineshape row gv[cell][k..k+4) is one 16-byte load per
// sub-segment).  Every lane is busy for the whole tile -- the regular half of the
// path sees none of the march's divergence -- and the two reductions of the
// deposit become cheap:
//   I_ang : sum over k of 2 dv_k Iv_k is a sequential sum inside the lane, in the
//           CPU's own order (RayTraceImageCPU.cpp:63-68); one LDS atomic per ray
//           into the work-group's private I_ang, flushed once per work-group with
//           coalesced native f64 atomics.
//   image : consecutive rays hit the same pixel (ASE: all na*nb rays of a pixel;
//           seeded: a handful of pixels per tile), so most of the per-pixel sum over
//           rays happens inside the wave -- one of four deposit modes per tile (few
//           runs: LDS-transposed wave sums; several pixels: per-wave LDS row cache;
//           one ray per pixel: staged plain stores; otherwise a segmented scan), each
//           ending in coalesced f64 atomics or stores.  See freq_tile.
#include "rt_march.hip"

#include <cfloat>

namespace rt {

#ifndef RT_FREQ_WAVES_SEED
#define RT_FREQ_WAVES_SEED 4 // the gain-only instance keeps far less per-lane state
#endif
#ifndef RT_FREQ_WAVES
#define RT_FREQ_WAVES 4 // waves per SIMD of the emission instance (127 VGPRs, no scratch; 168 at 3 before the argument block was split)
#endif

constexpr int VEC = 4; // frequencies per lane and pass; rows are padded to a multiple (DevParams::Kp)
// tiles a wave reserves per fetch of the tile counter, at most (see rt_freq_kernel).  With eight counters a fetch per
// tile is affordable, and for the emission instance it is what measures best (tools/ablate_shard.py with SHARDS=...:
// stand-in halves 0.47 against 0.52 ms, quarters 0.278 / 0.282, whole 0.865 / 0.868; config 5 at 2048^2 6.64 / 6.89 ms);
// the gain-only instance keeps the guided chunks (seed_small 1.212 against 1.228 ms)
#ifdef RT_FREQ_TILES_PER_FETCH
constexpr unsigned FREQ_TILES_PER_FETCH_EMIS = RT_FREQ_TILES_PER_FETCH, FREQ_TILES_PER_FETCH_GAIN = RT_FREQ_TILES_PER_FETCH;
#else
constexpr unsigned FREQ_TILES_PER_FETCH_EMIS = 1, FREQ_TILES_PER_FETCH_GAIN = 8;
#endif
struct alignas(16) FVec { float v[VEC]; };

// ---- float64 building blocks of the frequency pass ---------------------------------
// The frequency pass is the float64 half of the path; its results are compared with
// the CPU loop under the 1e-5 rel-L2 gate, not bit for bit (different libm exp,
// different summation order of the deposit, the source-function form of the update
// below).  Inside that contract exp and division are 1-2 ulp table / Newton kernels and
// FMA contraction is allowed.
#pragma clang fp contract(fast)

// exp(x), <= 2 ulp: x = (256 m + j) ln2/256 + r, |r| <= ln2/512;
// exp(x) = 2^m * 2^(j/256) * (1 + r + ... + r^5/120)  (remainder r^6/720 < 1e-20).
// tab[j] = 2^(j/256) lives in LDS.  Overflow -> inf, underflow -> 0, NaN -> NaN.
constexpr int EXP_TAB = 256;
__device__ __forceinline__ double exp_tab(double x, const double *tab)
{
    const double L2E   = 369.3299304675746;        // 256 / ln 2
    const double C_HI  = 0x1.62e42fef00000p-9;     // ln2/256, low 20 bits clear: t*C_HI exact
    const double C_LO  = 0x1.473de6af278edp-42;
    const double xc    = fmin(fmax(x, -1100.0), 1100.0);
    const double t     = rint(xc * L2E);
    const int n        = (int) t;
    double r           = fma(-t, C_HI, xc);
    r                  = fma(-t, C_LO, r);
    double p           = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p                  = fma(r, p, 1.0 / 6.0);
    p                  = fma(r, p, 0.5);
    p                  = fma(r, p, 1.0);
    p                  = fma(r, p, 1.0);
    return ldexp(tab[n & (EXP_TAB - 1)] * p, n >> 8); // a NaN argument is clamped away: callers that need it re-test
}

// The same kernel for VEC arguments at once: the four chains advance in lock step (the table
// reads are issued together), rint comes from the magic-constant add.  e^x of a NaN is NaN.
__device__ __forceinline__ void exp_tab_vec(const double (&x)[VEC], const double *tab, double (&e)[VEC])
{
    const double L2E   = 369.3299304675746;
    const double C_HI  = 0x1.62e42fef00000p-9;
    const double C_LO  = 0x1.473de6af278edp-42;
    const double MAGIC = 0x1.8p52; // adding it leaves rint(.) in the low mantissa bits
    double r[VEC], T[VEC];
    int m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        const double xc = fmin(fmax(x[j], -1100.0), 1100.0);
        double t        = fma(xc, L2E, MAGIC);
        const int n     = __double2loint(t);
        t -= MAGIC;
        r[j] = fma(-t, C_HI, xc);
        r[j] = fma(-t, C_LO, r[j]);
        T[j] = tab[n & (EXP_TAB - 1)];
        m[j] = n >> 8;
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        double p = fma(r[j], 1.0 / 24.0, 1.0 / 6.0); // |r| <= ln2/512: the r^5/120 term is below 4e-17
        p        = fma(r[j], p, 0.5);
        p        = fma(r[j], p, 1.0);
        p        = fma(r[j], p, 1.0);
        const double v = ldexp(T[j] * p, m[j]);
        e[j]           = x[j] != x[j] ? x[j] : v;
    }
}

// a / b to ~1 ulp: hardware reciprocal, one Newton step, one residual correction
__device__ __forceinline__ double div_fast(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y        = fma(fma(-b, y, 1.0), y, y);
    double q = a * y;
    return fma(fma(-b, q, a), y, q);
}

// v + (v moved by one DPP control): the building block of the in-register reductions (row_shr
// trees inside a row of 16 lanes, quad_perm exchanges) -- VALU latency instead of ds_bpermute
// round trips through the LDS crossbar.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_step(double v)
{
    const int lo  = __double2loint(v), hi = __double2hiint(v);
    const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(thi, tlo);
}
// Helper.h:549-557, one (sub-segment, frequency) update with emission, as the CPU writes it
// (kept for the sub-segments whose gain sum is zero or denormal-small, see ase_step)
__device__ __forceinline__ double ase_update(double Iv, float gs, float es, float w, const double *tab)
{
    const double gl = (double) (gs * w); // f32 product, then widened (Helper.h:549-550)
    const double el = (double) (es * w);
    if (fabs(gl) < 1e-3)
        return el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) + Iv * (1.0 + gl * (1.0 + 0.5 * gl));
    const double eg = exp_tab(gl, tab);
    return div_fast(el, gl) * (eg - 1.0) + Iv * eg;
}

// The same update in the form the kernel runs for every regular sub-segment:
//     Iv' = (el/gl) (e^gl - 1) + Iv e^gl  =  Iv + (e^gl - 1) (Iv + rs),   rs = es/gs,
// with gl = (double)(gs*w) exactly as the CPU rounds it (the exponent is where rounding of gl
// matters) and rs taken once per sub-segment: el/gl = fl(es w)/fl(gs w) = rs (1 + d), |d| <
// 1.2e-7, a float rounding of the two products that the 1e-5 gate does not resolve.
// e^gl - 1 is built without cancellation from gl = (256 m + j) ln2/256 + r:
//     S = 2^m 2^(j/256),  e^r - 1 = r Q(r),  e^gl - 1 = (S - 1) + S r Q(r),
// so one branch-free sequence covers |gl| < 1e-3 (where the CPU switches to a cubic whose
// own truncation, gl^3/24, is 4e-11) as well as large gains.  |r| <= ln2/512, so the
// quadratic Q = 1 + r/2 + r^2/6 is e^r - 1 to 1e-10 -- two orders below the rounding of rs.
// The caller keeps |gs * w| <= 708 (DevParams::gs_cap): e^gl stays a normal double, S is
// assembled by an integer add into the exponent field, and the reduction needs one constant
// (|256 m + j| < 2^18, so the rounding of ln2/256 moves r by < 6e-14).  A NaN lineshape value
// gives garbage here; the caller tests for it.  VEC independent chains, table reads issued
// together.
__device__ __forceinline__ void ase_step(double (&Iv)[VEC], const float gs, const double rs, const float (&w)[VEC],
                                         const double *tab)
{
    const double L2E   = 369.3299304675746;     // 256 / ln 2
    const double LN2_N = 0.0027076061740622863; // ln 2 / 256
    const double MAGIC = 0x1.8p52;              // adding it leaves rint(.) in the low mantissa bits
    double rq[VEC], T[VEC];
    int m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        const double x = (double) (gs * w[j]);
        double t       = fma(x, L2E, MAGIC);
        const int n    = __double2loint(t);
        t -= MAGIC;
        rq[j] = fma(-t, LN2_N, x);
        T[j]  = tab[n & (EXP_TAB - 1)];
        m[j]  = n >> 8;
    }
    // the four polynomials first: twelve instructions that need only r, while the table reads travel
    // (scheduled freely, the compiler waits for the first read four instructions after issuing it; measured:
    // no difference at four waves per SIMD -- the kernel is bound by instruction issue, not by this latency)
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        double q = fma(rq[j], 1.0 / 6.0, 0.5);
        q        = fma(rq[j], q, 1.0);
        rq[j] *= q;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        int hi; // exponent field += m in one v_lshl_add_u32 (the compiler's own choice is shift, mask, add)
        asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(hi) : "v"(m[j]), "v"(__double2hiint(T[j])));
        const double S   = __hiloint2double(hi, __double2loint(T[j]));
        const double em1 = fma(S, rq[j], S - 1.0);
        Iv[j]            = fma(em1, Iv[j] + rs, Iv[j]);
    }
}

// The same update with the range reduction and the polynomial in float32 -- for sub-segments with |gs w| <= 80, the
// rule (DevParams::gs_cap scaled by 80/708 says so per tile).  The argument x = gs * w IS a float (Helper.h:549:
// the product is rounded to float before it is widened), so nothing is lost by reducing it in float:
//     n = rint(x 256/ln2) by the magic-constant add (|n| < 2^22), r = x - n ln2/256 with the constant split in two
//     floats (two fma, |error| < 3e-10), rq = r (1 + r/2 + r^2/6) in float (relative 1e-7 of a term below 1.4e-3);
// S = 2^m 2^(j/256) and everything that carries magnitude stay float64: e^x - 1 = (S - 1) + S rq to 3e-10 of S,
// against 1e-10 for the all-float64 form and 1.2e-7 for the rounding of rs that both share.  Five float64-rate
// instructions per update instead of eleven (the kernel is bound by VALU issue, and a float64 instruction
// costs two to three float32 ones).
__device__ __forceinline__ void ase_step_f32(double (&Iv)[VEC], const float gs, const double rs, const float (&w)[VEC],
                                             const double *tab2)
{
    const float L2E   = 369.32993f;        // 256 / ln 2
    const float C_HI  = 2.7076062e-3f;     // ln 2 / 256 rounded to float ...
    const float C_LO  = (float) (0.0027076061740622863 - (double) 2.7076062e-3f); // ... and the rest
    const float MAGIC = 12582912.0f;       // 1.5 * 2^23: adding it leaves rint(.) in the low mantissa bits
    // (the float32 part as v_pk_*_f32 on pairs of frequencies -- two operations per instruction at the issue cost of
    // one float64 instruction in isolation -- was built and measured: 0.935 against 0.865 ms.  Packed float32 shares
    // the float64 pipe, which is what this loop is short of; plain float32 instructions of one wave overlap with
    // float64 instructions of another.)
    float rq[VEC];
    double T[VEC];
    unsigned nb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        const float x = gs * w[j];
        const float t = fmaf(x, L2E, MAGIC);
        nb[j]         = __float_as_uint(t); // 0x4B400000 + n: low byte = table index, bits 8.. = m
        const float n = t - MAGIC;
        float r       = fmaf(-n, C_HI, x);
        // (C_LO is only -7.4e-12, and leaving this correction out saves 2.5 % of the kernel -- but the error it leaves,
        // 2.7e-9 |x| relative on e^x, is systematic: every gain a little too large.  Measured: the image moved by 1e-8
        // against the CPU loop.  Kept.)
        r             = fmaf(-n, C_LO, r);
        T[j]          = tab2[nb[j] & (EXP_TAB - 1)];
        float q       = fmaf(r, 1.0f / 6.0f, 0.5f);
        q             = fmaf(r, q, 1.0f);
        rq[j]         = r * q;
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        // exponent field += m: the high word of table entry j is stored less j << 12 (exp2_tab2), so that adding
        // (0x4B400000 + 256 m + j) << 12 = (m << 20) + (j << 12) modulo 2^32 leaves exactly m << 20 on top of it
        int hi;
        asm("v_lshl_add_u32 %0, %1, 12, %2" : "=v"(hi) : "v"(nb[j]), "v"(__double2hiint(T[j])));
        const double S   = __hiloint2double(hi, __double2loint(T[j]));
        const double em1 = fma(S, (double) rq[j], S - 1.0);
        Iv[j]            = fma(em1, Iv[j] + rs, Iv[j]);
    }
}

// gain sums below this magnitude (or NaN) take the CPU's own formula: es/gs would overflow
// or the float product gs*w would underflow where es*w does not
#define RT_RS_MIN 1e-30f

// uniform-grid shortcut of deposit_index (RayTraceImageCPU.cpp:11-16): the grids
// of the beam are uniform (create_image checks it); guess the cell arithmetically,
// verify the two defining inequalities of findfirstsingle, else bisect.
__device__ __forceinline__ int deposit_index_fast(int n, const double *g, double d, double inv_d, double v)
{
    const double g0 = g[0], gl = g[n - 1];
    if (v < g0 - 0.5 * d || v > gl + 0.5 * d)
        return -1;
    const double t = v - 0.5 * d;
    if (t < g0)
        return 0;
    if (t > gl)
        return n;
    if (n < 2)
        return n;
    // a guess: the two inequalities below decide (clamped as a double: an infinite or huge v must not reach the
    // conversion, whose overflow is undefined -- the compiler may drop the integer clamp behind it)
    int u = (int) fmin(fmax((t - g0) * inv_d, 0.0), (double) (n - 2)) + 1;
    // first_not_below on [g0, gl]: unique u in [1, n-1] with g[u-1] < t <= g[u]
    // (u = 1 also when t == g0, where the bisection never tests g[0])
    if ((u == 1 || g[u - 1] < t) && g[u] >= t)
        return u;
    return first_not_below(g, n, t);
}

// The four deposit cells of a ray at once (RayTraceImageCPU.cpp:11-16 per axis, as deposit_index_fast): the
// guesses first, then the eight grid values they need in ONE round of loads, then the verification -- a wave
// that runs the axes one after the other, each behind its own early returns, pays four dependent memory round
// trips per tile.  g0 / gl = g[0] / g[n-1] come from the argument block (DevBeam::g_first / g_last).
struct AxisIn {
    int n;
    const double *g;
    double d, inv_d, g0, gl, v;
};
__device__ __forceinline__ void deposit_index4(const AxisIn (&A)[4], int (&idx)[4])
{
    int u[4];
    double t[4], lo[4], hi[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        t[a]   = A[a].v - 0.5 * A[a].d;
        // a guess (NaN or out of range: clamped, decided below).  Clamped as a double: an infinite or huge coordinate
        // must not reach the conversion -- its overflow is undefined and the compiler may then drop an integer clamp
        // behind it (a ray launched at a = +inf read the grid two billion entries off; tests/test_gpu_edges.py)
        const int last = A[a].n - 1;
        const int uu   = (int) fmin(fmax((t[a] - A[a].g0) * A[a].inv_d, 0.0), (double) (last > 0 ? last - 1 : 0)) + 1;
        u[a] = A[a].n >= 2 ? uu : 0;
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        lo[a] = A[a].g[u[a] > 0 ? u[a] - 1 : 0];
        hi[a] = A[a].g[u[a]];
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const double v = A[a].v, d = A[a].d, g0 = A[a].g0, gl = A[a].gl;
        int r;
        if (v < g0 - 0.5 * d || v > gl + 0.5 * d)
            r = -1;
        else if (t[a] < g0)
            r = 0;
        else if (t[a] > gl || A[a].n < 2)
            r = A[a].n;
        else if ((u[a] == 1 || lo[a] < t[a]) && hi[a] >= t[a])
            r = u[a];
        else
            r = -2; // the guess was wrong: bisect (below, behind one branch for the four axes)
        idx[a] = r;
    }
    if (idx[0] == -2 || idx[1] == -2 || idx[2] == -2 || idx[3] == -2) {
#pragma unroll
        for (int a = 0; a < 4; a++)
            if (idx[a] == -2)
                idx[a] = first_not_below(A[a].g, A[a].n, t[a]);
    }
}

// per-wave LDS scratch of the few-runs deposit: [4][XP_ROW] transposition rows (row stride
// 66 doubles: 16-byte aligned, rows 4 banks apart) + [FREQ_MAXQ][64] window totals
constexpr int XP_ROW          = 66;
constexpr int XS_ROW          = 18; // exclusive mode: staging row of 16 frequencies + 2 (bank spread; rows stay 16-byte aligned)
constexpr int FREQ_MAXQ       = 3;
constexpr int FREQ_WAVE_XPOSE = 4 * XP_ROW + FREQ_MAXQ * WAVE; // doubles per wave
// row stride of the per-wave row cache in doubles: Kp is a multiple of 4, so rows Kp apart start 8 Kp mod 128 bytes apart
// -- for the 84 frequencies of the seeded input every fourth row on the same LDS banks; one double more makes the
// stride odd and spreads sixteen rows over all 32 banks (lanes of different pixels add to the same column of their rows)
#ifdef RT_FREQ_ROW_NOPAD
__host__ __device__ constexpr int freq_row_stride(int Kp) { return Kp; }
#else
__host__ __device__ constexpr int freq_row_stride(int Kp) { return Kp + 1; }
#endif
// doubles of dynamic LDS of a work-group (layout: rt_freq_kernel)
inline size_t freq_lds_doubles(bool iang_in_lds, int n_ang, bool exclusive, int nslot, int Kp, int wg_waves)
{
    const size_t per_wave = exclusive ? (size_t) WAVE * XS_ROW : (size_t) FREQ_WAVE_XPOSE + (size_t) nslot * (size_t) freq_row_stride(Kp);
    return (size_t) 2 * EXP_TAB + (iang_in_lds ? (size_t) ((n_ang + 1) & ~1) : 0) + (size_t) wg_waves * per_wave;
}

// constant-address-space views: loads through them are scalar (s_load), whatever else the kernel stores
#define RT_CONST_AS __attribute__((address_space(4)))
typedef const RT_CONST_AS FreqCold *ColdPtr;
typedef const RT_CONST_AS double *ConstF64;
// a copy of one member struct of the cold block (dword loads through the constant address space)
template <typename T> __device__ __forceinline__ T load_cold(const RT_CONST_AS T *src)
{
    static_assert(sizeof(T) % 4 == 0, "dword copy");
    T r;
    unsigned *d                   = reinterpret_cast<unsigned *>(&r);
    const RT_CONST_AS unsigned *q = reinterpret_cast<const RT_CONST_AS unsigned *>(src);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++)
        d[i] = q[i];
    return r;
}

// Where a ray deposits and what it starts with -- the per-ray part of the frequency pass that does not depend on
// the frequency (Helper.h:518-533, RayTraceImageCPU.cpp:37-54): exit angles, seed factor, the four deposit cells.
// (Round 4 ran it for every ray of a launch in a pass of its own before the frequency kernel, result left in the
// record: seed_small 1.446 against 1.222 ms -- the pass re-reads and re-writes every line of the 750 MB of records,
// which costs more than the chain of loads it takes out of the tile preamble, where four waves per SIMD cover it.)
struct Placed {
    double f0; // seed factor (0 without a seed, for escaped rays and outside the profile)
    int pix;   // pixel i + j nx, or -1
    int ang;   // angle cell k + m na, or -1
};
__device__ __forceinline__ Placed place_ray(const unsigned hflags, ColdPtr C, const DevRays &R, const int nx, const bool backward,
                                            const unsigned ridx, const RecMeta &m, const unsigned fl, const rt_ray &ray)
{
    Placed P   = { 0.0, -1, -1 };
    rt_ray out = ray;
    rt_ray r2  = { m.px, m.py, 0.0f, 0.0f };
    if (hflags & FQ_NEED_EXIT) {
        // Helper.h:518-521: atanf(s.x / s.z) * 1e3f
        r2.a = atanf_flt32_kernel(m.sx / m.sz) * 1e3f;
        r2.b = atanf_flt32_kernel(m.sy / m.sz) * 1e3f;
    }
    if ((hflags & FQ_HAS_SEED) && !(fl & F_ESCAPED)) { // Helper.h:523-533
        if (backward || !R.sf) {
            const DevSeed SD = load_cold(&C->seed);
            P.f0 = backward ? seed_factor(SD, (double) m.px, (double) m.py, (double) r2.a, (double) r2.b)
                            : seed_factor(SD, (double) ray.x, (double) ray.y, (double) ray.a, (double) ray.b);
        } else {
            // the launch ray is a grid point: product of the tabulated factors, in seed_factor's order
            unsigned gi, gj, gk, gm;
            grid_index(R, ridx, gi, gj, gk, gm);
            const unsigned oj = (unsigned) R.ngx, ok = oj + (unsigned) R.ngy, om = ok + (unsigned) R.nga;
            if (R.sin[gi] & R.sin[oj + gj] & R.sin[ok + gk] & R.sin[om + gm]) {
                P.f0 = C->seed.f0 * R.sf[gi] * R.sf[oj + gj] * R.sf[ok + gk] * R.sf[om + gm];
                P.f0 = P.f0 < 0.0 ? 0.0 : P.f0;
            }
        }
    }
    if (hflags & FQ_PROBE)
        C->probe.ray2[ridx] = r2;
    if (hflags & FQ_OWN_CELLS) {
        unsigned gi, gj, gk, gm;
        grid_index(R, ridx, gi, gj, gk, gm);
        P.pix = (int) (gi + gj * (unsigned) nx);
        P.ang = (int) (gk + gm * (unsigned) R.nga);
    } else {
        if (!backward) { // RayTraceImageCPU.cpp:37-49
            out   = r2;
            out.a = -out.a;
            out.b = -out.b;
            if ((double) out.y < 0.0 && C->beam.g_first[1] >= 0.0)
                out.y = -out.y;
        }
        const AxisIn A[4] = {
            { C->beam.nx, C->beam.x, C->beam.dx, C->beam.inv_dx, C->beam.g_first[0], C->beam.g_last[0], (double) out.x },
            { C->beam.ny, C->beam.y, C->beam.dy, C->beam.inv_dy, C->beam.g_first[1], C->beam.g_last[1], (double) out.y },
            { C->beam.na, C->beam.a, C->beam.da, C->beam.inv_da, C->beam.g_first[2], C->beam.g_last[2], (double) out.a },
            { C->beam.nb, C->beam.b, C->beam.db, C->beam.inv_db, C->beam.g_first[3], C->beam.g_last[3], (double) out.b } };
        int ix[4];
        deposit_index4(A, ix);
        if (ix[0] >= 0 && ix[1] >= 0)
            P.pix = ix[0] + ix[1] * nx;
        if (ix[2] >= 0 && ix[3] >= 0)
            P.ang = ix[2] + ix[3] * C->beam.na;
    }
    return P;
}

// MAXQ: pixel runs of a tile that the few-runs deposit takes (its window totals are [MAXQ][64] doubles of the wave's
// LDS scratch): 3 in the stand-alone kernel, 2 in the one-launch run where LDS is short and a tile spans two pixels
// at most (rt_fused.hip)
// EXCL: the exclusive deposit (one ray per pixel, plain stores: DevParams::exclusive) -- a launch either is exclusive
// or is not, and as a compile-time parameter the mode's code and registers stay out of the instances that never use it
// (with a run-time flag the one-launch kernel carried it, and 32 bytes of scratch with it)
template <int SF, bool EMIS, int MAXQ = FREQ_MAXQ, bool EXCL = false>
__device__ __forceinline__ void freq_tile(const FreqHot &H, const unsigned hflags, ColdPtr C, double *lds_iang, const double *tab,
                                          double *xpose, double *cache, const unsigned tile, const int lane, const int k0 = 0,
                                          const int k1 = 0x7fffffff)
{
    // [k0, k1): the frequencies this call integrates and deposits (multiples of VEC; default: all K).  The one-launch
    // kernel splits the last tiles of a work-group over several waves that way (rt_fused.hip); every part repeats the
    // per-ray preamble, adds its share of the I_ang sums, and only the part that starts at 0 reports error -1.
    double *win = xpose + 4 * XP_ROW; // [MAXQ][64] totals of the current window of 64 frequencies
    const int S           = SF ? SF : H.L * RT_N_SUB;
    const int K           = H.K;
    const int k_end       = k1 < K ? k1 : K; // this call's frequencies are [k0, k_end)
    const int Kp          = H.Kp; // row stride of the lineshape tables and of the row cache
    const int nslot       = H.nslot;
    const unsigned n_rays = H.n_rays;
    const unsigned ridx   = tile * WAVE + (unsigned) lane;
    const bool have       = ridx < n_rays;
    const bool backward   = H.method == 1;
    // (records are tile-wise, rt_device.h: slot s of the 64 lanes is one contiguous run)
    const unsigned rrec      = have ? ridx : 0u;
    const unsigned char *rec = H.rec;
    constexpr bool use_emis = EMIS; // Helper.h:402, fixed per kernel instance (see launch_freq)
    const bool safe_check = (hflags & FQ_SAFE_CHECK) != 0, safe_skip = (hflags & FQ_SAFE_SKIP) != 0;
    const bool probe_on   = (hflags & FQ_PROBE) != 0;

    // ---- per-ray preamble: exit ray, seed factor, deposit cells ------------------
    // (everything read through C is loaded here and dead before the frequency loop)
    // The preamble is a chain of memory round trips, and with ~25 tiles per wave its latency is what the kernel
    // is made of once the frequency loop is fast (measured: 0.50 of 1.00 ms with the loop compiled out): so every
    // load that does not depend on another is issued up front -- the whole record (meta + slots) here -- and the
    // dependent rounds are as few as the mode allows (own-cell rays: none; otherwise one for the deposit cells).
    unsigned fl = 0, steps = 0;
    rt_ray ray  = { 0, 0, 0, 0 };
    RecMeta m   = { 0, 0, 0, 0, 1, 0 };
    RecSlot raw[SF ? SF : 1]; // slots as stored; those the ray never entered are masked with n_done below
#pragma unroll
    for (int s = 0; s < (SF ? SF : 1); s++)
        raw[s] = RecSlot{ 0.0f, 0.0f, 0 };
    const DevRays R = load_cold(&C->rays);
    // own-cell rays (FQ_OWN_CELLS): backward method, the rays are the beam's own grid points and the host has
    // verified that grid point i lands in deposit cell i on all four axes (grid_points_in_own_cells, the CPU's
    // getIndex on the very float the ray carries): pixel and angle cell ARE the grid indices of the ray
    const bool own = (hflags & FQ_OWN_CELLS) != 0;
    const bool need_ray = !own || probe_on; // (a failing own-cell ray loads its start ray when it is reported)
    if (have) {
        m = *reinterpret_cast<const RecMeta *>(rec + rec_meta_off(rrec, S, H.rec_stride));
        if (SF) {
            const unsigned char *slot0 = rec + rec_slot_off(rrec, 0, H.rec_stride);
#pragma unroll
            for (int s = 0; s < SF; s++)
                raw[s] = *reinterpret_cast<const RecSlot *>(slot0 + (size_t) s * REC_SLOT_ROW);
        }
        if (need_ray) {
            float ta, tb;
            load_ray(R, ridx, ray, ta, tb, false);
        }
        fl    = m.flags_steps & REC_FLAG_MASK;
        steps = m.flags_steps >> REC_STEPS_SHIFT;
    }
    auto start_ray = [&]() { // the launch ray, for the failure reports
        rt_ray r = ray;
        if (!need_ray) {
            float ta, tb;
            load_ray(R, ridx, r, ta, tb, false);
        }
        return r;
    };
    bool err1   = have && (double) (m.sz * m.sz) < 0.01; // Helper.h:515
    double f0   = 0.0;
    int pix = -1, ang = -1;
    if (have && !err1) {
        const Placed P = place_ray(hflags, C, R, H.nx, backward, ridx, m, fl, ray);
        f0  = P.f0;
        pix = P.pix;
        ang = P.ang;
    }
    if (have && probe_on) {
        C->probe.flags[ridx] = fl | (err1 ? F_ERR1 : 0u);
        C->probe.steps[ridx] = steps;
    }
    if (err1 && !safe_skip && k0 == 0) { // error -1: the ray is reported (once) and deposits nothing
        atomicOr(&H.ctl->failure_code, 1u << 1);
        unsigned slot_f = atomicAdd(&H.ctl->n_failed, 1u);
        if (slot_f < RT_N_FAILED_MAX)
            H.ctl->failed[slot_f] = start_ray();
    }
    const bool live = have && !err1 && !(fl & F_SKIP) && !(safe_skip && H.bad[ridx]);
    // exclusive mode: this ray is the only contributor of pixel own_pix and must write its
    // whole row (zeros if it contributes nothing); a ray that deposits elsewhere (never the
    // case for a consistent grid) keeps the atomic path for the foreign pixel.
    constexpr bool excl_all = EXCL;
    int own_pix = -1;
    if (excl_all && have) {
        const unsigned j = ridx % (unsigned) H.ny, i = ridx / (unsigned) H.ny;
        own_pix          = (int) (i + j * (unsigned) H.nx);
    }
    if (__ballot(live) == 0ull && !excl_all)
        return;
    if (!live) {
        pix = -1;
        ang = -1;
    }

    // ---- pixels of the tile (built once per tile) ----------------------------------
    // Few runs of equal pixel (ASE: a tile lies inside one pixel, or straddles two): one wave
    // sum per run and frequency, totals flushed as ONE coalesced atomic per run and 64
    // frequencies.  Otherwise (seeded mode: 4..8 pixels per tile): every distinct pixel owns one
    // row of the wave's LDS row cache [nslot][Kp]; lanes add into their pixel's row with LDS f64
    // atomics, rows are flushed per tile with coalesced atomics.  More distinct pixels than
    // rows: a segmented shuffle scan over the runs (adding the lanes of the surplus pixels
    // to the image one by one was tried: twice as slow on the 124.8 M-ray seeded case).
    const int pix_before             = __shfl_up(pix, 1, WAVE);
    const unsigned long long head_m  = __ballot(lane == 0 || pix_before != pix);
    const int n_runs                 = (int) __popcll(head_m);
    const bool few                   = n_runs <= MAXQ;
    const unsigned long long le_mask = (lane == WAVE - 1) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    const int run_id                 = (int) __popcll(head_m & le_mask) - 1;
    // distinct pixels -> cache rows: the first unassigned lane names a pixel, every lane with
    // that pixel takes the row; lane q keeps the pixel of row q for the flush
    int slot = -1, slot_pix = -1, n_slots = 0;
    bool cached = false;
    if (!few && nslot > 0) {
        unsigned long long rem = __ballot(pix >= 0);
        while (rem != 0ull && n_slots < nslot) {
            const int p = __builtin_amdgcn_readlane(pix, (int) __ffsll((long long) rem) - 1);
            if (pix == p)
                slot = n_slots;
            if (lane == n_slots)
                slot_pix = p;
            rem &= ~__ballot(pix == p);
            n_slots++;
        }
        cached = rem == 0ull;
    }

    // ---- the march record of this lane's ray ---------------------------------------
    // off[s]: byte offset of the lineshape row of sub-segment s inside its length's table (32 bits:
    // rt_hip_plan_create refuses tables of 4 GiB), so that a row load is SGPR base + VGPR offset
    float gs[SF ? SF : 1];
    double rs[SF ? SF : 1]; // es/gs, the source function of the sub-segment (see ase_step)
    unsigned off[SF ? SF : 1];
    const bool exact_emis = (hflags & FQ_EXACT_EMIS) != 0;
    bool irregular = false;
    if (SF) {
        const int n_done = (int) ((m.flags_steps >> REC_NDONE_SHIFT) & REC_NDONE_MASK);
#pragma unroll
        for (int s = 0; s < SF; s++) {
            // (rec_slot's rule on the slots loaded up front: only the first n_done in marching order were written)
            const bool written = backward ? s >= SF - n_done : s < n_done;
            const RecSlot sl   = written ? raw[s] : RecSlot{ 0.0f, 0.0f, 0 };
            gs[s]              = sl.g;
            const float e1   = sl.e;
            off[s]           = (unsigned) sl.c * (unsigned) Kp * 4u;
            // regular: the source-function form (ase_step) takes this sub-segment; not when the gain
            // sum is tiny or NaN, and never in the exact mode (rt_hip_plan_set_exact_emission), which
            // runs the CPU's own formula with its per-frequency division throughout
            // (|gs| <= gs_cap keeps |gs * gv| <= 708 for every lineshape value; NaN fails both tests)
            const bool regular = fabsf(gs[s]) >= RT_RS_MIN && fabsf(gs[s]) <= H.gs_cap && !exact_emis;
            rs[s]              = regular ? div_fast((double) e1, (double) gs[s]) : 0.0;
            // (a sub-segment with both sums zero is the identity either way: x = 0, e^x - 1 = 0)
            irregular = irregular || (!regular && (gs[s] != 0.0f || e1 != 0.0f));
        }
    }
    // no such sub-segment in the whole tile (the rule): the six updates of a frequency batch run
    // as one straight-line block, so the table reads of one overlap the arithmetic of another
    const bool all_regular = __ballot(irregular) == 0ull;
    // ... and every |gs w| of the tile stays below 80 (the rule as well): the float32 range reduction (ase_step_f32)
    bool big = false;
    if (SF) {
#pragma unroll
        for (int s = 0; s < SF; s++)
            big = big || !(fabsf(gs[s]) <= H.gs_cap * (80.0f / 708.0f));
    }
#ifdef RT_FREQ_NO_F32
    const bool all_small = false;
#else
    const bool all_small = all_regular && __ballot(big) == 0ull;
#endif
    // a NaN or an infinity among the lineshape values (the CPU's 0 * NaN, 0 * inf and inf / inf: every one of them
    // leaves Iv = NaN, Helper.h:549-557) is tested per frequency only when the host scan of the tables found one
    const bool gv_nan = (hflags & FQ_GV_NAN) != 0;

    double angsum = 0.0; // RayTraceImageCPU.cpp:63-68, sequential in k like the CPU
    double iv_min = 0.0; // min over k of Iv, NaNs ignored: negative <=> error -2 (Helper.h:582-594)
    double *img_row = H.image + (size_t) (pix >= 0 ? pix : 0) * (size_t) K;
    const ConstF64 dv2 = (ConstF64) (unsigned long long) H.dv2;      // wave-uniform reads: scalar loads
    const ConstF64 sfk = (ConstF64) (unsigned long long) H.seed_fk;

    // row of sub-segment s, frequencies kb .. kb+3 (SF: the tables of lengths 1 and 2 are kernel arguments)
    auto load_rows = [&](FVec (&w)[SF ? SF : 1], const int kb) {
#pragma unroll
        for (int s = 0; s < (SF ? SF : 1); s++) {
            const float *base = (s < RT_N_SUB ? H.gv0 : H.gv1) + kb;
            // (opaque here, so that the zero-extension of the offset stays beside the load and the
            // instruction selector finds the SGPR-base + 32-bit-VGPR-offset form)
            unsigned o = off[s];
            asm volatile("" : "+v"(o));
#ifdef RT_ABL_NOLOAD
            o &= 15u;
#endif
            w[s] = *reinterpret_cast<const FVec *>(reinterpret_cast<const char *>(base) + o);
        }
    };

    // The frequency loop, instantiated once per deposit mode (exclusive / few runs / row
    // cache / segmented scan) so that each instance keeps only its own deposit state in
    // registers: `deposit(kb, v)` consumes the lane's values of frequencies kb .. kb+VEC-1.
    // scale_late: the deposit mode multiplies by `scale` itself, once per summed value instead of once per ray
    // and frequency (the wave sums of the few-runs mode: sum_r (Iv_r scale) and (sum_r Iv_r) scale differ by
    // summation-order rounding, which the deposit does not preserve anyway)
    auto frequency_loop = [&](auto deposit, const bool scale_late = false) {
        // (Requesting the rows of batch kb + 1 early was measured and dropped: a second set of row registers costs
        // the fourth wave per SIMD, 1.34 against 1.30 ms; a request into the same registers right after batch kb
        // has consumed its own, so that the rows travel during the deposit, 1.31 against 1.30 ms and 1.65 against
        // 1.60 ms seeded -- with four waves per SIMD the latency is covered by the other waves.)
#ifdef RT_ABL_NOFREQ // profiling only: the tile preamble alone
        const int K_loop = K > 1000000 ? K : 0;
        const int kb_first = 0;
#elif defined(RT_ABL_ONEBATCH) // profiling only: one batch of four frequencies per tile
        const int K_loop = K < VEC ? K : VEC;
        const int kb_first = 0;
#else
        const int K_loop = k_end;
        const int kb_first = k0;
#endif
        for (int kb = kb_first; kb < K_loop; kb += VEC) {
            double Iv[VEC];
            if (use_emis) {
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    Iv[j] = 0.0;
                if (SF) {
                    FVec w[SF ? SF : 1];
                    load_rows(w, kb);
                    if (all_small) {
#pragma unroll
                        for (int s = 0; s < SF; s++)
                            ase_step_f32(Iv, gs[s], rs[s], w[s].v, tab + EXP_TAB);
                    } else if (all_regular) {
#pragma unroll
                        for (int s = 0; s < SF; s++)
                            ase_step(Iv, gs[s], rs[s], w[s].v, tab);
                    } else
#pragma unroll
                    for (int s = 0; s < SF; s++) {
                        if (fabsf(gs[s]) >= RT_RS_MIN && fabsf(gs[s]) <= H.gs_cap && !exact_emis) {
                            ase_step(Iv, gs[s], rs[s], w[s].v, tab);
                        } else {
                            const float e1 = rec_slot(rec, rrec, H.rec_stride, s, SF, m.flags_steps, backward).e;
                            if (gs[s] != 0.0f || e1 != 0.0f) { // else the update is the identity
#pragma unroll
                                for (int j = 0; j < VEC; j++)
                                    Iv[j] = ase_update(Iv[j], gs[s], e1, w[s].v[j], tab);
                            }
                        }
                    }
                    if (gv_nan) {
#pragma unroll
                        for (int j = 0; j < VEC; j++) {
                            bool wn = false;
#pragma unroll
                            for (int s = 0; s < SF; s++)
                                wn = wn || !(fabsf(w[s].v[j]) <= FLT_MAX);
                            Iv[j] = wn ? __builtin_nan("") : Iv[j];
                        }
                    }
                } else {
                    bool wnan[VEC]; // a NaN or infinity anywhere in this frequency's lineshape values (0 * NaN on the CPU)
#pragma unroll
                    for (int j = 0; j < VEC; j++)
                        wnan[j] = false;
                    for (int s = 0; s < S; s++) {
                        const RecSlot sl = rec_slot(rec, rrec, H.rec_stride, s, S, m.flags_steps, backward);
                        const float g1 = sl.g, e1 = sl.e;
                        const int c1   = sl.c;
                        const float *row  = H.gain[s / RT_N_SUB + 1].gv + (size_t) c1 * (size_t) Kp + kb;
                        const FVec w = *reinterpret_cast<const FVec *>(row);
#pragma unroll
                        for (int j = 0; j < VEC; j++)
                            wnan[j] = wnan[j] || !(fabsf(w.v[j]) <= FLT_MAX);
                        if (fabsf(g1) >= RT_RS_MIN && fabsf(g1) <= H.gs_cap && !exact_emis) {
                            const double r1 = div_fast((double) e1, (double) g1);
                            ase_step(Iv, g1, r1, w.v, tab);
                        } else if (g1 != 0.0f || e1 != 0.0f) {
#pragma unroll
                            for (int j = 0; j < VEC; j++)
                                Iv[j] = ase_update(Iv[j], g1, e1, w.v[j], tab);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < VEC; j++)
                        Iv[j] = wnan[j] ? __builtin_nan("") : Iv[j];
                }
            } else {
                // gain only, Helper.h:569-580: f64 products summed in sub-segment order
                double gl[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    gl[j] = 0.0;
                if (SF) {
                    FVec w[SF ? SF : 1];
                    load_rows(w, kb);
#pragma unroll
                    for (int s = 0; s < SF; s++) {
#pragma unroll
                        for (int j = 0; j < VEC; j++)
                            gl[j] += (double) gs[s] * (double) w[s].v[j];
                    }
                } else {
                    for (int s = 0; s < S; s++) {
                        const RecSlot sl = rec_slot(rec, rrec, H.rec_stride, s, S, m.flags_steps, backward);
                        const float *row = H.gain[s / RT_N_SUB + 1].gv + (size_t) sl.c * (size_t) Kp + kb;
                        const FVec w     = *reinterpret_cast<const FVec *>(row);
#pragma unroll
                        for (int j = 0; j < VEC; j++)
                            gl[j] += (double) sl.g * (double) w.v[j];
                    }
                }
                // Iv = f0 f[4][k] exp(gl); for f0 = 0 that is exactly 0 unless exp overflows (0 * inf):
                // a wave none of whose lanes needs the exponential skips it
                bool need = f0 != 0.0;
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    need = need || gl[j] > 700.0 || gl[j] != gl[j];
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    Iv[j] = f0 * sfk[kb + j];
                if (__ballot(need) != 0ull) {
                    double eg[VEC];
                    exp_tab_vec(gl, tab, eg);
#pragma unroll
                    for (int j = 0; j < VEC; j++)
                        Iv[j] *= eg[j];
                }
            }
            // No masking here: lanes without a live ray sit in runs of pixel -1, which no deposit
            // mode flushes, and their error flags and I_ang sum are dropped below; the padding
            // columns K .. Kp-1 carry w = dv = 0, hence Iv = 0 (deposits test k < K themselves).
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                iv_min = fmin(iv_min, Iv[j]);
                angsum += dv2[kb + j] * Iv[j]; // RayTraceImageCPU.cpp:66: (2.0 * dv) * Iv
                if (!scale_late)
                    Iv[j] = Iv[j] * H.scale;   // RayTraceImageCPU.cpp:59
            }
#ifndef RT_ABL_NODEPOSIT
            if (!safe_check) // the checking pass of a failing run integrates without depositing
                deposit(kb, Iv);
#endif
        }
    };

    if (excl_all) {
        // One ray per pixel: plain stores of the row, no reduction, no atomics.  A lane owns a
        // whole image row (K doubles), so lane-wise stores would put 32 bytes into each of 64 rows
        // per instruction; instead the wave stages 16 frequencies of all its rows in LDS
        // (cache = [64][XS_ROW] here) and stores them as 4 rows x 128 contiguous bytes per
        // instruction.
        // The rule (checked per tile, wave-uniform): a full tile whose 64 rays deposit into their own pixels, the pixels
        // nx apart (consecutive rays are consecutive in y: no wrap to the next x inside the tile), K a multiple of the
        // 16 staged frequencies.  The flush is then 16 x (LDS read at a constant offset, one 64-bit add, a 16-byte store) with no
        // per-row pixel look-up and no range tests -- the general form below costs ~12 instructions per store, a sixth
        // of the deposit's instructions on the 4096^2 x 512 image.
        const int own0        = __builtin_amdgcn_readlane(own_pix, 0);
        const bool regular_tile = __ballot(have && pix == own_pix) == ~0ull && (K & 15) == 0 &&
                                  __builtin_amdgcn_readlane(own_pix, WAVE - 1) == own0 + (WAVE - 1) * H.nx;
        frequency_loop([&](int kb, double (&v)[VEC]) {
            double *mine = cache + lane * XS_ROW + (kb & 12);
            if (regular_tile) {
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    mine[j] = v[j];
                if ((kb & 12) == 12) {
                    __builtin_amdgcn_wave_barrier();
                    // 16 bytes per lane: eight lanes cover the 128 staged bytes of a row, a store instruction eight rows
                    // (half as many store instructions as with 8 bytes per lane: the deposit is bound by store issue)
                    typedef double f64x2s __attribute__((ext_vector_type(2)));
                    double *dst         = H.image + ((size_t) (own0 + (lane >> 3) * H.nx) * (size_t) K + (size_t) ((kb & ~15) + 2 * (lane & 7)));
                    const size_t gstep  = (size_t) 8 * (size_t) H.nx * (size_t) K; // eight image rows on
                    const double *src   = cache + (lane >> 3) * XS_ROW + 2 * (lane & 7);
#pragma unroll 4
                    for (int g = 0; g < WAVE / 8; g++)
                        __builtin_nontemporal_store(*reinterpret_cast<const f64x2s *>(src + g * 8 * XS_ROW), reinterpret_cast<f64x2s *>(dst + (size_t) g * gstep));
                    __builtin_amdgcn_wave_barrier();
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                mine[j] = (pix == own_pix) ? v[j] : 0.0;
                if (pix >= 0 && pix != own_pix && kb + j < K)
                    unsafeAtomicAdd(&img_row[kb + j], v[j]);
            }
            if ((kb & 12) == 12 || kb + VEC >= K) {
                __builtin_amdgcn_wave_barrier();
                const int k = (kb & ~15) + (lane & 15);
#pragma unroll 4
                for (int g = 0; g < WAVE / 4; g++) {
                    const int row  = 4 * g + (lane >> 4);
                    const int opix = __shfl(own_pix, row, WAVE);
                    if (opix >= 0 && k < K)
                        H.image[(size_t) opix * (size_t) K + (size_t) k] = cache[row * XS_ROW + (lane & 15)];
                }
                __builtin_amdgcn_wave_barrier();
            }
        });
    } else if (few) {
        // Per run and frequency one sum over the wave: the lanes park their four
        // values in the wave's LDS scratch [4][XP_ROW], lane (j, p) = (lane / 16, lane % 16)
        // adds four neighbours of frequency j, a row_shr tree inside the row of 16 lanes
        // finishes the sum (15 VALU operations for 4 frequencies instead of 4 x 18 for four
        // full-wave DPP trees).  Totals wait in the LDS window win[run][k mod 64] and leave
        // as one coalesced atomic per run and 64 frequencies.
        int pixq[MAXQ];
        unsigned long long mm = head_m;
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            const int l = mm ? (int) __ffsll((long long) mm) - 1 : 0;
            pixq[q]     = mm ? __builtin_amdgcn_readlane(pix, l) : -1;
            mm &= mm - 1;
        }
        const bool single = n_runs == 1;
        auto wave_sums    = [&](const int q, const int kb, double (&v)[VEC], const bool masked) {
#pragma unroll
            for (int j = 0; j < VEC; j++)
                xpose[j * XP_ROW + lane] = (!masked || run_id == q) ? v[j] : 0.0;
            __builtin_amdgcn_wave_barrier();
            const double *src = xpose + (lane >> 4) * XP_ROW + 4 * (lane & 15);
            double t          = (src[0] + src[1]) + (src[2] + src[3]);
            __builtin_amdgcn_wave_barrier();
            t = dpp_step<0x111, 0xf>(t);
            t = dpp_step<0x112, 0xf>(t);
            t = dpp_step<0x114, 0xf>(t);
            t = dpp_step<0x118, 0xf>(t);
            if ((lane & 15) == 15)
                win[q * WAVE + ((kb + (lane >> 4)) & (WAVE - 1))] = t;
        };
        frequency_loop([&](int kb, double (&v)[VEC]) {
            if (single) {
                wave_sums(0, kb, v, false);
            } else {
#pragma unroll
                for (int q = 0; q < MAXQ; q++) {
                    if (pixq[q] >= 0)
                        wave_sums(q, kb, v, true);
                }
            }
            if ((((kb + VEC) & (WAVE - 1)) == 0) || kb + VEC >= k_end) {
                // flush the window of 64 frequencies that ends here (the part of it this call has integrated)
                __builtin_amdgcn_wave_barrier();
                const int k = ((kb + VEC - 1) & ~(WAVE - 1)) + lane;
#pragma unroll
                for (int q = 0; q < MAXQ; q++) {
                    if (pixq[q] >= 0 && k < k_end && k >= k0)
                        unsafeAtomicAdd(&H.image[(size_t) pixq[q] * (size_t) K + (size_t) k], win[q * WAVE + lane] * H.scale);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }, true);
    } else if (cached) {
        // LDS atomics serialise on equal addresses (the lanes of one pixel): quads whose four
        // lanes share a pixel add their values with two quad_perm DPP steps and send one atomic
        double *my_row       = cache + (size_t) (slot >= 0 ? slot : 0) * (size_t) freq_row_stride(Kp);
        const int slot_first = __builtin_amdgcn_update_dpp(0, slot, 0x00, 0xf, 0xf, true); // quad_perm:[0,0,0,0]
        const unsigned long long same = __ballot(slot == slot_first && slot >= 0);
        const bool quad_one  = ((same >> (lane & ~3)) & 0xfull) == 0xfull;
        const bool sender    = slot >= 0 && (!quad_one || (lane & 3) == 0);
        frequency_loop([&](int kb, double (&v)[VEC]) {
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                double q4 = dpp_step<0xb1, 0xf>(v[j]); // quad_perm:[1,0,3,2]
                q4        = dpp_step<0x4e, 0xf>(q4);   // quad_perm:[2,3,0,1]
                if (sender)
                    unsafeAtomicAdd(&my_row[kb + j], quad_one ? q4 : v[j]);
            }
        });
    } else {
        // segmented shuffle scan over the runs, one atomic per run and frequency
        const int pix_prev = __shfl_up(pix, 1, WAVE);
        const bool head    = lane == 0 || pix_prev != pix;
        int run_start      = head ? lane : -1;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t = __shfl_up(run_start, o, WAVE);
            if (lane >= o && t > run_start)
                run_start = t;
        }
        const int head_next = __shfl_down(head ? 1 : 0, 1, WAVE);
        const bool tail     = (lane == WAVE - 1 || head_next != 0) && pix >= 0;
        frequency_loop([&](int kb, double (&v)[VEC]) {
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                double a = v[j];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double t = __shfl_up(a, 1 << i, WAVE);
                    if ((lane - (1 << i)) >= run_start)
                        a += t;
                }
                if (tail && kb + j < k_end)
                    unsafeAtomicAdd(&img_row[kb + j], a);
            }
        });
    }
    if (cached) {
        // flush this tile's rows: one coalesced run of atomics per pixel, rows re-zeroed
        for (int q = 0; q < n_slots; q++) {
            const int pq = __builtin_amdgcn_readlane(slot_pix, q);
            for (int k = lane; k < K; k += WAVE) {
                const double v = cache[q * freq_row_stride(Kp) + k];
                cache[q * freq_row_stride(Kp) + k] = 0.0;
#ifdef RT_ABL_NOROWFLUSH // profiling only
                if (v == 1234.5)
#endif
                unsafeAtomicAdd(&H.image[(size_t) pq * (size_t) K + (size_t) k], v);
            }
        }
    }
    // a NaN intensity makes the I_ang sum NaN (Helper.h:590-593: error -3, after the sign test)
    const bool bad_neg = iv_min < 0.0, bad_nan = angsum != angsum;
    if (live && (bad_neg || bad_nan) && !safe_skip) {
        atomicOr(&H.ctl->failure_code, bad_neg ? (1u << 2) : (1u << 3));
        unsigned slot_f = atomicAdd(&H.ctl->n_failed, 1u);
        if (slot_f < RT_N_FAILED_MAX)
            H.ctl->failed[slot_f] = start_ray();
        if (safe_check)
            H.bad[ridx] = 1;
    }
    // a failing ray adds nothing to I_ang (RayTraceImageCPU.cpp:29-36: `continue` before the deposit)
    if (ang >= 0 && !safe_check && !(bad_neg || bad_nan)) {
        if (lds_iang)
            unsafeAtomicAdd(&lds_iang[ang], angsum);
        else
            unsafeAtomicAdd(&H.iang[ang], angsum);
    }
}

#ifdef RT_WAVETIMES
// diagnostic build: per wave of the last frequency launch {start, tables ready, first tile done, end} (100 MHz ticks)
__device__ unsigned long long g_ft[6][8192]; // [4]: blockIdx | wave << 16 | XCC_ID << 24 | CU/SE id << 32, [5]: tiles done
__device__ unsigned g_ft_n;
#endif
template <int SF, bool EMIS, bool EXCL>
__global__ void __launch_bounds__(FREQ_WG_WAVES * 64, EMIS ? RT_FREQ_WAVES : RT_FREQ_WAVES_SEED) rt_freq_kernel(const FreqKArg A)
{
#ifdef RT_WAVETIMES
    const unsigned long long ft_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long ft_first = 0, ft_tiles = 0;
#endif
    // LDS of a work-group, all dynamic (launch_freq sizes it with freq_lds_doubles):
    //   [2^(j/256), j = 0..255, twice: as it is, and with the high word less j << 12 (ase_step_f32)]
    //   [I_ang histogram, na*nb doubles rounded up to even (if it fits)]
    //   per wave: [transposition rows + window totals of the few-runs deposit, FREQ_WAVE_XPOSE doubles][row cache [nslot][Kp]],
    //             or in exclusive mode the store staging rows [64][XS_ROW] alone
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const FreqHot &H       = A.hot;
    const bool iang_in_lds = (H.flags & FQ_IANG_LDS) != 0;
    constexpr bool excl    = EXCL;
    const int nslot        = H.nslot;
    const int n_ang        = H.n_ang;
    double *exp2_tab       = reinterpret_cast<double *>(lds_raw);
    double *lds_iang       = iang_in_lds ? exp2_tab + 2 * EXP_TAB : nullptr;
    double *waves_base     = exp2_tab + 2 * EXP_TAB + (iang_in_lds ? ((n_ang + 1) & ~1) : 0);
    const size_t per_wave  = excl ? (size_t) WAVE * XS_ROW : (size_t) FREQ_WAVE_XPOSE + (size_t) nslot * (size_t) freq_row_stride(H.Kp);
    double *mine           = waves_base + (size_t) (unsigned) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) * per_wave;
    double *xpose          = mine; // (not used in exclusive mode)
    double *cache          = excl ? mine : mine + FREQ_WAVE_XPOSE;
    for (size_t c = threadIdx.x; c < (size_t) (blockDim.x >> 6) * per_wave; c += blockDim.x)
        waves_base[c] = 0.0;
    for (int c = (int) threadIdx.x; c < EXP_TAB; c += (int) blockDim.x) {
        const double e        = exp2((double) c * (1.0 / EXP_TAB));
        exp2_tab[c]           = e;
        exp2_tab[EXP_TAB + c] = __hiloint2double(__double2hiint(e) - (c << 12), __double2loint(e));
    }
    if (lds_iang) {
        for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x)
            lds_iang[c] = 0.0;
    }
    __syncthreads();
    const int lane = lane_id();
#ifdef RT_WAVETIMES
    const unsigned long long ft_ready = __builtin_amdgcn_s_memrealtime();
#endif
    // Tiles are handed out dynamically.  A returning atomic on ONE word is served at ~88 per microsecond chip-wide
    // (MI355X_MICROARCH.md, "dequeue"): one fetch per 64-ray tile from one counter bounded the whole kernel at
    // 1.13 ms for the stand-in's 99 750 tiles (measured: 1.22 of 1.26 ms with the frequency loop compiled out) and
    // made a 12 000-tile launch a pure counter benchmark.  So (i) the tile range is cut into eight shards with a
    // counter each, on separate cache lines; a wave starts on shard blockIdx.x % 8 (work-groups b and b + 8 share
    // an XCD: a speed matter only) and moves on to the next shard when its own is empty; (ii) a wave reserves
    // up to FREQ_TILES_PER_FETCH tiles per fetch -- guided self-scheduling: (tiles left in the shard) / (2 x waves per
    // shard), down to one at the end, where balance matters.
    const unsigned n_tiles_run = H.tile_end - H.tile_begin;
    // shard sh owns the tiles sh, sh + 8, sh + 16, ... (interleaved: every shard sees the same mix of cheap and
    // expensive regions of the image, so the eight run dry together and stealing is an end-game matter)
    unsigned shard = blockIdx.x & 7u, tried = 0;
    auto shard_size = [&](unsigned sh) { return (n_tiles_run + 7u - sh) / 8u; };
    unsigned s_n    = shard_size(shard);
    unsigned t_next = 0, t_end = 0; // wave-uniform window of reserved tiles (indices inside the shard)
    const unsigned sh_shift = H.fetch_shift > 3 ? H.fetch_shift - 3 : 0; // log2(2 x waves per shard)
    constexpr unsigned FREQ_TILES_PER_FETCH = EMIS ? FREQ_TILES_PER_FETCH_EMIS : FREQ_TILES_PER_FETCH_GAIN;
    auto chunk_of = [&](unsigned left) {
        const unsigned c = left >> sh_shift;
        return c < 1u ? 1u : (c > FREQ_TILES_PER_FETCH ? FREQ_TILES_PER_FETCH : c);
    };
    unsigned tch = chunk_of(s_n);
    // (The instruction arbiter of a SIMD favours its lower wave slots: measured, tools/freq_wave_times.py, 25 / 31 / 41 /
    // 57 us per tile for the waves in slots 0 / 1 / 2 / 3, and the launch ends when the slowest wave has finished the
    // tiles it holds -- the last 10 % of the launch run with fewer than four waves per SIMD.  Tried against it:
    // s_setprio rotated from tile to tile, evens the slots out and costs 3 % of the throughput; the slow slots stop
    // fetching early, worse; reversed priorities in the end game only, no effect.  Left to the hardware.)
    for (;;) {
        if (t_next == t_end) {
            unsigned base = 0;
            if (lane == 0)
                base = atomicAdd(&H.ctl->next_tile_f[H.freq_id][shard][0], tch);
            base = (unsigned) __builtin_amdgcn_readfirstlane((int) base);
            if (base >= s_n) { // this shard is empty: on to the next one, until all eight have been seen empty
                if (++tried == 8)
                    break;
                shard = (shard + 1) & 7u;
                s_n   = shard_size(shard);
                tch   = 1; // a guest takes single tiles
                continue;
            }
            t_next = base;
            t_end  = s_n - base < tch ? s_n : base + tch;
            tch    = chunk_of(s_n - t_end);
        }
        const unsigned tile = H.tile_begin + (t_next++) * 8u + shard;
        // the cold half of the argument block, addressed inside the kernarg segment; made opaque per tile so
        // that its loads stay in the tile's preamble instead of being hoisted (and kept live) above this loop
        ColdPtr C = (ColdPtr) ((const RT_CONST_AS char *) __builtin_amdgcn_kernarg_segment_ptr() + offsetof(FreqKArg, cold));
        asm volatile("" : "+s"(C));
        // likewise the flag word and the lane number: the dozens of wave-uniform predicates and lane masks derived
        // from them are recomputed per tile (one instruction each) rather than parked in SGPRs across all tiles
        unsigned hflags = H.flags;
        int lane_t      = lane;
        asm volatile("" : "+s"(hflags), "+v"(lane_t));
        freq_tile<SF, EMIS, FREQ_MAXQ, EXCL>(H, hflags, C, lds_iang, exp2_tab, xpose, cache, tile, lane_t);
#ifdef RT_WAVETIMES
        if (!ft_first)
            ft_first = __builtin_amdgcn_s_memrealtime();
        ft_tiles++;
#endif
    }
#ifdef RT_WAVETIMES
    const unsigned long long ft_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef RT_ABL_NOIANGFLUSH
    if (lds_iang && !(H.flags & FQ_DBG_NOFLUSH)) {
        __syncthreads();
        for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x) {
            const double v = lds_iang[c];
            if (v != 0.0)
                unsafeAtomicAdd(&H.iang[c], v);
        }
    }
#endif
#ifdef RT_WAVETIMES
    if (lane == 0) {
        const unsigned w = atomicAdd(&g_ft_n, 1u);
        if (w < 8192) {
            g_ft[0][w] = ft_start;
            g_ft[1][w] = ft_ready;
            g_ft[2][w] = ft_first ? ft_first : ft_loop_end;
            g_ft[3][w] = ft_loop_end;
            unsigned hwid = 0, xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_ft[4][w] = (unsigned long long) blockIdx.x | ((unsigned long long) (threadIdx.x >> 6) << 16) | ((unsigned long long) (xcc & 0xf) << 24) | ((unsigned long long) hwid << 32);
            g_ft[5][w] = ft_tiles;
        }
    }
#endif
}

} // namespace rt
