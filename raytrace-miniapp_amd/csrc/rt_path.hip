// rt_path.hip -- the debug path tracer of the reference on the GPU:
// RayTrace::calc_ray_path (src/RayTraceImage.cpp:440-477), i.e. RayTrace_calc_ray with a
// `debug` array (Helper.h:419-426, 505-511, 536-542, 559-566).  The march kernel
// records the (x, y) position at every sub-segment boundary; this kernel adds the
// frequency-integrated intensity after every sub-segment.  A diagnostic, not a hot path:
// lanes = rays, libm exp and IEEE division, the reference's float accumulation order
// (sum over k of (float)(2 Iv_k dv_k), k ascending).
#pragma once
#include "rt_march.hip"

namespace rt {

extern "C" __global__ void __launch_bounds__(256) rt_path_kernel(const DevParams P)
{
    const unsigned n_rays = (unsigned) P.rays.count;
    const unsigned ridx   = blockIdx.x * blockDim.x + threadIdx.x;
    if (ridx >= n_rays)
        return;
    const int S  = P.L * RT_N_SUB;
    const int K  = P.K;
    const unsigned char *rec = P.rec; // (tile-wise records, rt_device.h)
    const RecMeta m = *reinterpret_cast<const RecMeta *>(rec + rec_meta_off(ridx, S, P.rec_stride));
    const unsigned fl = m.flags_steps & REC_FLAG_MASK;
    float *dbg        = P.path + (size_t) ridx * 3 * (size_t) (S + 1);
    if ((double) (m.sz * m.sz) < 0.01) { // Helper.h:515: positions only
        P.path_err[ridx] = -1;
        return;
    }
    rt_ray ray;
    float ta, tb;
    load_ray(P.rays, ridx, ray, ta, tb, false);
    double f0 = 0.0;
    if (P.has_seed && !(fl & F_ESCAPED)) { // Helper.h:523-533
        if (P.method == 1) {
            const float a2 = atanf_flt32_kernel(m.sx / m.sz) * 1e3f;
            const float b2 = atanf_flt32_kernel(m.sy / m.sz) * 1e3f;
            f0             = seed_factor(P.seed, (double) m.px, (double) m.py, (double) a2, (double) b2);
        } else {
            f0 = seed_factor(P.seed, (double) ray.x, (double) ray.y, (double) ray.a, (double) ray.b);
        }
    }
    bool neg = false, nan = false;
    for (int k = 0; k < K; k++) {
        double Iv = P.has_seed ? f0 * P.seed.f[4][k] : 0.0;
        const double dvk = P.beam.dv[k];
        dbg[2] += (float) (2 * Iv * dvk); // Helper.h:536-542
        for (int s = 0; s < S; s++) {     // Helper.h:543-566: emission formula per sub-segment
            const RecSlot sl = rec_slot(rec, ridx, P.rec_stride, s, S, m.flags_steps, P.method == 1);
            const float gs = sl.g, es = sl.e;
            const int cell = sl.c;
            const float w   = P.gain[s / RT_N_SUB + 1].gv[(size_t) cell * (size_t) P.Kp + (size_t) k];
            const double gl = (double) (gs * w);
            const double el = (double) (es * w);
            if (fabs(gl) < 1e-3) {
                Iv = el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) + Iv * (1.0 + gl * (1.0 + 0.5 * gl));
            } else {
                const double eg = exp(gl);
                Iv              = el / gl * (eg - 1.0) + Iv * eg;
            }
            dbg[3 * (s + 1) + 2] += (float) (2 * Iv * dvk);
        }
        neg = neg || Iv < 0.0;
        nan = nan || Iv != Iv;
    }
    P.path_err[ridx] = neg ? -2 : (nan ? -3 : 0);
}

} // namespace rt
