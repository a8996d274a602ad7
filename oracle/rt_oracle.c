/*
 * rt_oracle.c -- CPU restatement of the XRayTrace ray-trace imaging path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and
 * only as the checker / reported baseline -- never as the product path.
 *
 * Pinned (tests/test_oracle_pin.py): bit-identical image and I_ang to the
 * reference's own RayTraceImageCPULoop compiled from /root/reference (recipe
 * oracle/Makefile -> oracle/_ref/librt_ref.so) on ASE_small.dat and
 * seed_small.dat, and within 5.2e-7 rel-L2 of the golden image embedded in
 * those files.
 *
 * It restates, in plain C with every float/double promotion spelled out:
 *   march + frequency integration : src/common/RayTraceImageHelper.h:379-595
 *   inner integrators             : Helper.h:270-313 (propagate), :318-351 (propagate2)
 *   helpers                       : Helper.h:73-89, :101-117, :131-143, :153-158, :168-247
 *   deposit loop                  : src/RayTraceImageCPU.cpp:11-70
 *   thread split                  : src/RayTraceImage.cpp:89-134
 * Build with -ffp-contract=off (the reference is bit-stable across -O2/-O3
 * only without FMA contraction).
 *
 * Differences from the reference by design: K (nv) and N are runtime sizes
 * (no K_MAX/N_MAX stack arrays), the march and the frequency pass are separate
 * functions so that the per-ray march record can be probed, and counters
 * (cell-steps, escaped rays) are measured.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float x, y, z;
} vec3f;

/* Helper.h:73-89 -- f32 dot, sqrtf, f64 reciprocal, rounded back to f32. */
static inline void renormalise(vec3f *s)
{
    float q   = s->x * s->x + s->y * s->y + s->z * s->z;
    float inv = (float) (1.0 / (double) sqrtf(q));
    s->x *= inv;
    s->y *= inv;
    s->z *= inv;
}

/* Helper.h:101-117 -- index of the first grid point >= v, 0..n. */
static inline size_t first_not_below(const double *g, size_t n, double v)
{
    if (v < g[0])
        return 0;
    if (v > g[n - 1])
        return n;
    size_t lo = 0, hi = n - 1;
    while (hi - lo != 1) {
        size_t mid = (hi + lo) / 2;
        if (g[mid] >= v)
            hi = mid;
        else
            lo = mid;
    }
    return hi;
}

/* Helper.h:131-143 -- interpolation interval, always 1..n-1. */
static inline uint32_t interval_index(const double *g, uint32_t n, double v)
{
    uint32_t lo = 0, hi = n - 1;
    while (hi - lo != 1) {
        uint32_t mid = (hi + lo) / 2;
        if (g[mid] >= v)
            hi = mid;
        else
            lo = mid;
    }
    return hi;
}

/* Helper.h:153-158 */
static inline float lerp2(float u, float v, float f00, float f10, float f01, float f11)
{
    float u1 = 1.0f - u;
    float v1 = 1.0f - v;
    return (u * f10 + u1 * f00) * v1 + (u * f11 + u1 * f01) * v;
}

/* Helper.h:168-220 -- monotone cubic Hermite with limited 3-point slopes. */
static double pchip_eval(size_t n, const double *xs, const double *ys, double x)
{
    if (x <= xs[0] || n <= 2) {
        double t = (x - xs[0]) / (xs[1] - xs[0]);
        return (1.0 - t) * ys[0] + t * ys[1];
    } else if (x >= xs[n - 1]) {
        double t = (x - xs[n - 2]) / (xs[n - 1] - xs[n - 2]);
        return (1.0 - t) * ys[n - 2] + t * ys[n - 1];
    }
    size_t i   = first_not_below(xs, n, x);
    double fl  = ys[i - 1];
    double fr  = ys[i];
    double t   = (x - xs[i - 1]) / (xs[i] - xs[i - 1]);
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

/* Helper.h:230-247 -- separable seed intensity, clamped at 0, zero off-grid. */
static void seed_intensity(const rt_seed *sd, double x, double y, double a, double b, double *Iv)
{
    double f = 0.0;
    if (x >= sd->x[0][0] && x <= sd->x[0][sd->dim[0] - 1] && y >= sd->x[1][0] &&
        y <= sd->x[1][sd->dim[1] - 1] && a >= sd->x[2][0] && a <= sd->x[2][sd->dim[2] - 1] &&
        b >= sd->x[3][0] && b <= sd->x[3][sd->dim[3] - 1]) {
        double px = pchip_eval((size_t) sd->dim[0], sd->x[0], sd->f[0], x);
        double py = pchip_eval((size_t) sd->dim[1], sd->x[1], sd->f[1], y);
        double pa = pchip_eval((size_t) sd->dim[2], sd->x[2], sd->f[2], a);
        double pb = pchip_eval((size_t) sd->dim[3], sd->x[3], sd->f[3], b);
        f         = sd->f0 * px * py * pa * pb;
        f         = f < 0.0 ? 0.0 : f;
    }
    for (int k = 0; k < sd->dim[4]; k++)
        Iv[k] = f * sd->f[4][k];
}

/* Helper.h:270-313 -- adaptive 2nd-order Taylor steps in a linear-index
 * medium n = n0 + gx*x + gy*y; returns the path length, displacement in *r. */
static float step_linear_medium(vec3f *r, vec3f *s, float n0, float gx, float gy,
                                const float lim[3], float c, uint64_t *iters)
{
    float path  = 0.0f;
    float dzcap = c * 1.00001f * lim[2];
    r->x = 0;
    r->y = 0;
    r->z = 0;
    float n = n0;
    while (fabsf(r->x) < lim[0] && fabsf(r->y) < lim[1] && fabsf(r->z) < lim[2] &&
           (double) fabsf(n - n0) < 0.05) {
        n        = n0 + r->x * gx + r->y * gy;
        float t  = (s->x * gx + s->y * gy + 1e-12f) / n;
        float fx = gx / n - s->x * t;
        float fy = gy / n - s->y * t;
        float fz = -s->z * t;
        float h  = c * 0.1f / fabsf(t);
        h        = h < dzcap ? h : dzcap;
        float h2 = 1.0001f * (lim[2] - fabsf(r->z)) / fabsf(s->z);
        float h3 = c * 0.05f * (fabsf(s->x) + 5e-4f) / (fabsf(fx) + 1e-8f);
        float h4 = c * 0.05f * (fabsf(s->y) + 5e-4f) / (fabsf(fy) + 1e-8f);
        h        = h < h2 ? h : h2;
        h        = h < h3 ? h : h3;
        h        = h < h4 ? h : h4;
        float ht = h * t;
        float c1 = 0.5f * h * h * (1.0f - ht / 3.0f + ht * ht / 12.0f);
        r->x += s->x * h + c1 * fx;
        r->y += s->y * h + c1 * fy;
        r->z += s->z * h + c1 * fz;
        float c2 = h * (1.0f - 0.5f * ht + ht * ht / 6.0f);
        s->x += c2 * fx;
        s->y += c2 * fy;
        s->z += c2 * fz;
        renormalise(s);
        path += h;
        if (iters)
            ++*iters;
    }
    return path;
}

/* Helper.h:318-351 -- advance inside one cell box until the ray leaves it or
 * reaches dzrem; index gradient recomputed from the 4 corners each pass. */
static float cross_cell(vec3f *pos, vec3f *s, float dzrem, const double xc[2], const double yc[2],
                        const float box[4], const double nc[4], int mirror_y, float c,
                        uint64_t *iters2, uint64_t *iters1)
{
    float z        = 0.0f;
    float path     = 0.0f;
    const float wx = (float) (xc[1] - xc[0]);
    const float wy = (float) (yc[1] - yc[0]);
    float ya       = mirror_y ? fabsf(pos->y) : pos->y;
    while (pos->x > box[0] && pos->x < box[1] && ya > box[2] && ya < box[3] &&
           (double) z < 0.999 * (double) dzrem) {
        ya       = mirror_y ? fabsf(pos->y) : pos->y;
        float u  = (float) (((double) pos->x - xc[0]) / (double) wx);
        float v  = (float) (((double) ya - yc[0]) / (double) wy);
        float n0 = lerp2(u, v, (float) nc[0], (float) nc[1], (float) nc[2], (float) nc[3]);
        float gx = (float) ((1.0 - (double) v) * (nc[1] - nc[0]) / (double) wx +
                            (double) v * (nc[3] - nc[2]) / (double) wx);
        float gy = (float) ((1.0 - (double) u) * (nc[2] - nc[0]) / (double) wy +
                            (double) u * (nc[3] - nc[1]) / (double) wy);
        if (mirror_y && pos->y < 0)
            gy = -gy;
        vec3f r;
        float lim[3] = { 0.1f * wx, 0.1f * wy, dzrem - z };
        path += step_linear_medium(&r, s, n0, gx, gy, lim, c, iters1);
        pos->x += r.x;
        pos->y += r.y;
        pos->z += r.z;
        z += fabsf(r.z);
        ya = mirror_y ? fabsf(pos->y) : pos->y;
        if (iters2)
            ++*iters2;
    }
    return path;
}

/*
 * March one ray (Helper.h:404-533 minus the seed lookup).  Fills
 * gvl/evl/ivl [L][3] (L = N-1), the exit ray and flags.
 * Returns 0, or -1 if the ray ends ~perpendicular to z (Helper.h:515).
 */
static int march_impl(const rt_ray *ray, int N, float dz0, const rt_gain *gain, int use_emis,
                      int method, float c, float *gvl, float *evl, int32_t *ivl, rt_ray *ray_out,
                      int *escaped_out, rt_oracle_counters *cnt, float *path /* [3*(3L+1)] or NULL */)
{
    const int L = N - 1;
    for (int i = 0; i < L * RT_N_SUB; i++) {
        gvl[i] = 0.0f;
        evl[i] = 0.0f;
        ivl[i] = 0;
    }
    vec3f s, pos;
    pos.x = ray->x;
    pos.y = ray->y;
    pos.z = 0.0f;
    s.x   = tanf(1e-3f * ray->a);
    s.y   = tanf(1e-3f * ray->b);
    s.z   = 1.0f;
    if (method == 1) {
        s.x = -s.x;
        s.y = -s.y;
        s.z = -s.z;
    }
    renormalise(&s);
    if (path) { /* Helper.h:419-426: start point of the recorded path */
        const int i0 = (method == 1) ? (N - 1) * RT_N_SUB : 0;
        memset(path, 0, sizeof(float) * 3 * (size_t) (RT_N_SUB * (N - 1) + 1));
        path[3 * i0 + 0] = pos.x;
        path[3 * i0 + 1] = pos.y;
    }

    int escaped    = 0;
    uint64_t steps = 0, it2 = 0, it1 = 0;
    for (int seg = 0; seg < L && !escaped; seg++) {
        const int ii      = (method == 1) ? N - seg - 1 : seg + 1;
        const rt_gain *g  = &gain[ii];
        const uint32_t Nx = (uint32_t) g->Nx, Ny = (uint32_t) g->Ny;
        float lo_x = (float) g->x[0], hi_x = (float) g->x[Nx - 1];
        float lo_y = (float) g->y[0], hi_y = (float) g->y[Ny - 1];
        int mirror_y = 0;
        if (lo_y >= 0) {
            lo_y     = -hi_y;
            mirror_y = 1;
        }
        float z = 0.0f;
        for (int iz = 0; iz < RT_N_SUB; iz++) {
            const int is = (method == 1) ? RT_N_SUB - iz - 1 : iz;
            float z_stop = (dz0 * ((float) iz + 1.0f) / RT_N_SUB);
            while (z < 0.995f * z_stop) {
                if (pos.x < lo_x || pos.x > hi_x || pos.y < lo_y || pos.y > hi_y ||
                    (double) (s.z * s.z) < 0.01) {
                    escaped = 1;
                    break;
                }
                float ya    = mirror_y ? fabsf(pos.y) : pos.y;
                uint32_t k1 = interval_index(g->x, Nx, (double) pos.x);
                uint32_t k2 = interval_index(g->y, Ny, (double) ya);
                uint32_t c00 = (k1 - 1) + (k2 - 1) * Nx;
                uint32_t c10 = k1 + (k2 - 1) * Nx;
                uint32_t c01 = (k1 - 1) + k2 * Nx;
                uint32_t c11 = k1 + k2 * Nx;
                double xc[2] = { g->x[k1 - 1], g->x[k1] };
                double yc[2] = { g->y[k2 - 1], g->y[k2] };
                double nc[4] = { g->n[c00], g->n[c10], g->n[c01], g->n[c11] };
                float u = (float) (((double) pos.x - g->x[k1 - 1]) / (g->x[k1] - g->x[k1 - 1]));
                float v = (float) (((double) ya - g->y[k2 - 1]) / (g->y[k2] - g->y[k2 - 1]));
                float g0 = lerp2(u, v, g->g0[c00], g->g0[c10], g->g0[c01], g->g0[c11]);
                float E0 = 0.0f;
                if (use_emis) {
                    E0 = lerp2(u, v, g->E0[c00], g->E0[c10], g->E0[c01], g->E0[c11]);
                    E0 = E0 >= 0 ? E0 : 0.0f;
                }
                pos.z = 0.0f;
                float box[4] = { (float) (xc[0] - 0.1 * (g->x[k1] - g->x[k1 - 1])),
                                 (float) (xc[1] + 0.1 * (g->x[k1] - g->x[k1 - 1])),
                                 (float) (yc[0] - 0.1 * (g->y[k2] - g->y[k2 - 1])),
                                 (float) (yc[1] + 0.1 * (g->y[k2] - g->y[k2 - 1])) };
                if (mirror_y && k2 <= 1)
                    box[2] = -box[3];
                float path = cross_cell(&pos, &s, z_stop - z, xc, yc, box, nc, mirror_y, c, &it2, &it1);
                z += fabsf(pos.z);
                const int slot = (ii - 1) * RT_N_SUB + is;
                gvl[slot] += g0 * path;
                evl[slot] += E0 * path;
                ivl[slot] = (int32_t) c00;
                steps++;
            }
            if (path) { /* Helper.h:505-511: position at the end of every sub-segment */
                const int idx     = RT_N_SUB * (ii - 1) + is + (method == 1 ? 0 : 1);
                path[3 * idx + 0] = pos.x;
                path[3 * idx + 1] = pos.y;
            }
        }
    }
    if (cnt) {
        cnt->cell_steps += steps;
        cnt->cross_iters += it2;
        cnt->inner_iters += it1;
        cnt->n_escaped += (uint64_t) escaped;
    }
    *escaped_out = escaped;
    if ((double) (s.z * s.z) < 0.01)
        return -1;
    ray_out->x = pos.x;
    ray_out->y = pos.y;
    ray_out->a = atanf(s.x / s.z) * 1e3f;
    ray_out->b = atanf(s.y / s.z) * 1e3f;
    return 0;
}

int rt_oracle_march(const rt_ray *ray, int N, float dz0, const rt_gain *gain, int use_emis,
                    int method, float c, float *gvl, float *evl, int32_t *ivl, rt_ray *ray_out,
                    int *escaped_out, rt_oracle_counters *cnt)
{
    return march_impl(ray, N, dz0, gain, use_emis, method, c, gvl, evl, ivl, ray_out, escaped_out, cnt, NULL);
}

/*
 * Frequency pass (Helper.h:543-594).  Iv[K] holds the start intensity on
 * entry (zero, or the seed profile).  Returns 0, -2 (negative), -3 (NaN).
 */
int rt_oracle_integrate(int N, const rt_gain *gain, int use_emis, int K, const float *gvl,
                        const float *evl, const int32_t *ivl, double *Iv)
{
    const int L = N - 1;
    if (use_emis) {
        for (int i = 0; i < L; i++) {
            for (int is = 0; is < RT_N_SUB; is++) {
                const int slot   = i * RT_N_SUB + is;
                const float *row = &gain[i + 1].gv[(size_t) ivl[slot] * (size_t) K];
                const float gs = gvl[slot], es = evl[slot];
                for (int k = 0; k < K; k++) {
                    double gl = (double) (gs * row[k]); /* f32 product, then widened */
                    double el = (double) (es * row[k]);
                    if (fabs(gl) < 1e-3) {
                        Iv[k] = el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) +
                                Iv[k] * (1.0 + gl * (1.0 + 0.5 * gl));
                    } else {
                        double eg = exp(gl);
                        Iv[k]     = el / gl * (eg - 1.0) + Iv[k] * eg;
                    }
                }
            }
        }
    } else {
        for (int k = 0; k < K; k++) {
            double gl = 0;
            for (int i = 0; i < L; i++) {
                for (int is = 0; is < RT_N_SUB; is++) {
                    const int slot = i * RT_N_SUB + is;
                    double w = (double) gain[i + 1].gv[(size_t) k + (size_t) ivl[slot] * (size_t) K];
                    gl += (double) gvl[slot] * w; /* f64 product here (Helper.h:575-576) */
                }
            }
            Iv[k] *= exp(gl);
        }
    }
    int neg = 0, nan = 0;
    for (int k = 0; k < K; k++) {
        neg = neg || Iv[k] < 0.0;
        nan = nan || Iv[k] != Iv[k];
    }
    return neg ? -2 : (nan ? -3 : 0);
}

/* Whole ray = Helper.h:379-595. */
int rt_oracle_calc_ray(const rt_ray *ray, int N, float dz0, const rt_gain *gain,
                       const rt_seed *seed, int K, int method, double *Iv, rt_ray *ray_out,
                       float *gvl, float *evl, int32_t *ivl, int *escaped_out,
                       rt_oracle_counters *cnt)
{
    const int use_emis = gain[0].E0 != NULL && seed == NULL;
    for (int k = 0; k < K; k++)
        Iv[k] = 0.0;
    int escaped = 0;
    int err = rt_oracle_march(ray, N, dz0, gain, use_emis, method, 0.5f, gvl, evl, ivl, ray_out,
                              &escaped, cnt);
    if (escaped_out)
        *escaped_out = escaped;
    if (err)
        return err;
    if (seed != NULL && !escaped) {
        if (method == 1)
            seed_intensity(seed, ray_out->x, ray_out->y, (double) ray_out->a, (double) ray_out->b, Iv);
        else if (method == 2)
            seed_intensity(seed, ray->x, ray->y, ray->a, ray->b, Iv);
    }
    return rt_oracle_integrate(N, gain, use_emis, K, gvl, evl, ivl, Iv);
}

/* RayTraceImageCPU.cpp:11-16 */
static inline int deposit_index(int n, const double *g, double d, double v)
{
    if (v < g[0] - 0.5 * d || v > g[n - 1] + 0.5 * d)
        return -1;
    return (int) first_not_below(g, (size_t) n, v - 0.5 * d);
}

typedef struct {
    int N;
    const rt_beam *beam;
    const rt_gain *gain;
    const rt_seed *seed;
    int method;
    const rt_ray *rays;
    size_t n_rays;
    double scale;
    double *image;
    double *I_ang;
    unsigned int failure_code;
    rt_ray failed[RT_N_FAILED_MAX];
    int n_failed;
    rt_oracle_counters cnt;
} loop_job;

/* RayTraceImageCPU.cpp:19-70, serial over [rays, rays+n_rays). */
static void run_loop(loop_job *j)
{
    const rt_beam *bm = j->beam;
    const int K       = bm->nv;
    const int L       = j->N - 1;
    double *Iv        = (double *) malloc(sizeof(double) * (size_t) (K > 0 ? K : 1));
    float *gvl        = (float *) malloc(sizeof(float) * (size_t) (L * RT_N_SUB + 1));
    float *evl        = (float *) malloc(sizeof(float) * (size_t) (L * RT_N_SUB + 1));
    int32_t *ivl      = (int32_t *) malloc(sizeof(int32_t) * (size_t) (L * RT_N_SUB + 1));
    const float dz0   = (float) bm->dz; /* RayTraceImageCPU.cpp:31, f64 -> f32 at the call */
    for (size_t it = 0; it < j->n_rays; ++it) {
        const rt_ray ray = j->rays[it];
        rt_ray out;
        int err = rt_oracle_calc_ray(&ray, j->N, dz0, j->gain, j->seed, K, j->method, Iv, &out, gvl,
                                     evl, ivl, NULL, &j->cnt);
        j->cnt.n_rays++;
        if (err != 0) {
            if (j->n_failed < RT_N_FAILED_MAX)
                j->failed[j->n_failed++] = ray;
            j->failure_code |= 1u << (unsigned) (-err);
            continue;
        }
        if (j->method == 1) {
            out = ray;
        } else {
            out.a = -out.a;
            out.b = -out.b;
            if ((double) out.y < 0.0 && bm->y[0] >= 0.0)
                out.y = -out.y;
        }
        int i1 = deposit_index(bm->nx, bm->x, bm->dx, (double) out.x);
        int i2 = deposit_index(bm->ny, bm->y, bm->dy, (double) out.y);
        int i3 = deposit_index(bm->na, bm->a, bm->da, (double) out.a);
        int i4 = deposit_index(bm->nb, bm->b, bm->db, (double) out.b);
        if (i1 >= 0 && i2 >= 0) {
            double *px = &j->image[(size_t) K * ((size_t) i1 + (size_t) i2 * (size_t) bm->nx)];
            for (int k = 0; k < K; k++)
                px[k] += Iv[k] * j->scale;
        }
        if (i3 >= 0 && i4 >= 0) {
            double acc = 0.0;
            for (int k = 0; k < K; k++)
                acc += 2.0 * bm->dv[k] * Iv[k];
            j->I_ang[i3 + i4 * bm->na] += acc;
        }
    }
    free(Iv);
    free(gvl);
    free(evl);
    free(ivl);
}

static void *run_loop_thread(void *p)
{
    run_loop((loop_job *) p);
    return NULL;
}

/*
 * The back-end loop.  n_threads <= 1: the serial loop, bit-identical to
 * RayTraceImageCPULoop.  n_threads > 1: contiguous ray chunks with private
 * images summed in chunk order, as RayTraceImageThreadLoop does
 * (RayTraceImage.cpp:89-134) -- equal to serial up to f64 summation order.
 * image / I_ang are accumulated into (+=): pass them zeroed.
 */
int rt_oracle_image_loop(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                         int method, const rt_ray *rays, size_t n_rays, double scale,
                         double *image, double *I_ang, unsigned int *failure_code,
                         rt_ray *failed_rays, int max_failed, int *n_failed,
                         rt_oracle_counters *counters, int n_threads)
{
    if (N < 2 || !beam || !gain || !image || !I_ang || (n_rays && !rays))
        return RT_ERR_ARG;
    const size_t n_img = (size_t) beam->nx * (size_t) beam->ny * (size_t) beam->nv;
    const size_t n_ang = (size_t) beam->na * (size_t) beam->nb;
    if (n_threads < 1)
        n_threads = 1;
    if ((size_t) n_threads > n_rays)
        n_threads = n_rays ? (int) n_rays : 1;
    loop_job *jobs = (loop_job *) calloc((size_t) n_threads, sizeof(loop_job));
    pthread_t *th  = (pthread_t *) calloc((size_t) n_threads, sizeof(pthread_t));
    if (!jobs || !th)
        return RT_ERR_NOMEM;
    const size_t chunk = n_rays / (size_t) n_threads + 1;
    size_t begin       = 0;
    for (int t = 0; t < n_threads; t++) {
        loop_job *j = &jobs[t];
        j->N        = N;
        j->beam     = beam;
        j->gain     = gain;
        j->seed     = seed;
        j->method   = method;
        j->rays     = rays + begin;
        j->n_rays   = (begin + chunk <= n_rays) ? chunk : n_rays - begin;
        j->scale    = scale;
        begin += j->n_rays;
        if (n_threads == 1) {
            j->image = image;
            j->I_ang = I_ang;
            run_loop(j);
        } else {
            j->image = (double *) calloc(n_img ? n_img : 1, sizeof(double));
            j->I_ang = (double *) calloc(n_ang ? n_ang : 1, sizeof(double));
            if (!j->image || !j->I_ang)
                return RT_ERR_NOMEM;
            pthread_create(&th[t], NULL, run_loop_thread, j);
        }
    }
    unsigned int code = 0;
    int nf            = 0;
    rt_oracle_counters tot;
    memset(&tot, 0, sizeof(tot));
    for (int t = 0; t < n_threads; t++) {
        loop_job *j = &jobs[t];
        if (n_threads > 1) {
            pthread_join(th[t], NULL);
            for (size_t i = 0; i < n_img; i++)
                image[i] += j->image[i];
            for (size_t i = 0; i < n_ang; i++)
                I_ang[i] += j->I_ang[i];
            free(j->image);
            free(j->I_ang);
        }
        code |= j->failure_code;
        for (int f = 0; f < j->n_failed; f++)
            if (failed_rays && nf < max_failed)
                failed_rays[nf++] = j->failed[f];
        tot.n_rays += j->cnt.n_rays;
        tot.cell_steps += j->cnt.cell_steps;
        tot.cross_iters += j->cnt.cross_iters;
        tot.inner_iters += j->cnt.inner_iters;
        tot.n_escaped += j->cnt.n_escaped;
    }
    if (failure_code)
        *failure_code = code;
    if (n_failed)
        *n_failed = nf;
    if (counters)
        *counters = tot;
    free(jobs);
    free(th);
    return RT_OK;
}

/*
 * Per-ray probe for parity tests: march record + Iv of each ray.
 * gvl/evl/ivl: [n][L][3]; ray2: [n]; flags: bit0 escaped, bit1 error -1;
 * steps: cell-steps; Iv: [n][K] (may be NULL); err: [n] return codes.
 */
int rt_oracle_probe(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                    int method, const rt_ray *rays, size_t n_rays, float *gvl, float *evl,
                    int32_t *ivl, rt_ray *ray2, uint32_t *flags, uint32_t *steps, double *Iv,
                    int32_t *err)
{
    const int K     = beam->nv;
    const int L     = N - 1;
    const int S     = L * RT_N_SUB;
    const float dz0 = (float) beam->dz;
    double *iv      = (double *) malloc(sizeof(double) * (size_t) (K > 0 ? K : 1));
    float *g        = (float *) malloc(sizeof(float) * (size_t) (S + 1));
    float *e        = (float *) malloc(sizeof(float) * (size_t) (S + 1));
    int32_t *c      = (int32_t *) malloc(sizeof(int32_t) * (size_t) (S + 1));
    for (size_t r = 0; r < n_rays; r++) {
        rt_oracle_counters cnt;
        memset(&cnt, 0, sizeof(cnt));
        rt_ray out = { 0, 0, 0, 0 };
        int esc    = 0;
        int rc = rt_oracle_calc_ray(&rays[r], N, dz0, gain, seed, K, method, iv, &out, g, e, c, &esc, &cnt);
        if (gvl)
            memcpy(gvl + r * (size_t) S, g, sizeof(float) * (size_t) S);
        if (evl)
            memcpy(evl + r * (size_t) S, e, sizeof(float) * (size_t) S);
        if (ivl)
            memcpy(ivl + r * (size_t) S, c, sizeof(int32_t) * (size_t) S);
        if (ray2)
            ray2[r] = out;
        if (flags)
            flags[r] = (uint32_t) (esc ? 1 : 0) | (uint32_t) (rc == -1 ? 2 : 0);
        if (steps)
            steps[r] = (uint32_t) cnt.cell_steps;
        if (Iv)
            memcpy(Iv + r * (size_t) K, iv, sizeof(double) * (size_t) K);
        if (err)
            err[r] = rc;
    }
    free(iv);
    free(g);
    free(e);
    free(c);
    return RT_OK;
}

/*
 * Path tracer: what RayTrace_calc_ray leaves in its `debug` array (Helper.h:419-426,
 * 505-511, 536-542, 559-566) and RayTrace::calc_ray_path returns
 * (src/RayTraceImage.cpp:440-477): for every ray the (x, y) position at the
 * 3(N-1)+1 sub-segment boundaries and the frequency-integrated intensity
 * sum_k 2 Iv_k dv_k after each sub-segment (float accumulation, k order).  With a
 * debug array the reference always takes the per-sub-segment emission formula, also
 * in seeded mode (Helper.h:543).  path: [n][3*(3L+1)] as {x, y, I} triples; err [n].
 * c is the step safety factor (0.5 in create_image).
 */
int rt_oracle_calc_ray_path(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                            int method, float c, const rt_ray *rays, size_t n_rays, float *path,
                            int32_t *err)
{
    const int K        = beam->nv;
    const int L        = N - 1;
    const int S        = L * RT_N_SUB;
    const int N2       = S + 1;
    const float dz0    = (float) beam->dz;
    const int use_emis = gain[0].E0 != NULL && seed == NULL;
    double *Iv   = (double *) malloc(sizeof(double) * (size_t) (K > 0 ? K : 1));
    float *gvl   = (float *) malloc(sizeof(float) * (size_t) (S + 1));
    float *evl   = (float *) malloc(sizeof(float) * (size_t) (S + 1));
    int32_t *ivl = (int32_t *) malloc(sizeof(int32_t) * (size_t) (S + 1));
    for (size_t r = 0; r < n_rays; r++) {
        float *dbg = path + r * 3 * (size_t) N2;
        rt_ray out = { 0, 0, 0, 0 };
        int esc    = 0;
        for (int k = 0; k < K; k++)
            Iv[k] = 0.0;
        int rc = march_impl(&rays[r], N, dz0, gain, use_emis, method, c, gvl, evl, ivl, &out, &esc, NULL, dbg);
        err[r] = rc;
        if (rc)
            continue;
        if (seed != NULL && !esc) {
            if (method == 1)
                seed_intensity(seed, out.x, out.y, (double) out.a, (double) out.b, Iv);
            else if (method == 2)
                seed_intensity(seed, rays[r].x, rays[r].y, rays[r].a, rays[r].b, Iv);
        }
        dbg[2] = 0.0f;
        for (int k = 0; k < K; k++)
            dbg[2] += (float) (2 * Iv[k] * beam->dv[k]);
        for (int i = 0; i < L; i++) {
            for (int is = 0; is < RT_N_SUB; is++) {
                const int slot   = i * RT_N_SUB + is;
                const float *row = &gain[i + 1].gv[(size_t) ivl[slot] * (size_t) K];
                for (int k = 0; k < K; k++) {
                    double gl = (double) (gvl[slot] * row[k]);
                    double el = (double) (evl[slot] * row[k]);
                    if (fabs(gl) < 1e-3) {
                        Iv[k] = el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) +
                                Iv[k] * (1.0 + gl * (1.0 + 0.5 * gl));
                    } else {
                        double eg = exp(gl);
                        Iv[k]     = el / gl * (eg - 1.0) + Iv[k] * eg;
                    }
                }
                const int idx = 3 * (slot + 1) + 2;
                dbg[idx]      = 0.0f;
                for (int k = 0; k < K; k++)
                    dbg[idx] += (float) (2 * Iv[k] * beam->dv[k]);
            }
        }
        int neg = 0, nan = 0;
        for (int k = 0; k < K; k++) {
            neg = neg || Iv[k] < 0.0;
            nan = nan || Iv[k] != Iv[k];
        }
        err[r] = neg ? -2 : (nan ? -3 : 0);
    }
    free(Iv);
    free(gvl);
    free(evl);
    free(ivl);
    return RT_OK;
}
