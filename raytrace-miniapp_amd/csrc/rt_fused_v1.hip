// rt_kernels.hip -- hand-written HIP (gfx950 / CDNA4) kernel of the ray-trace
// imaging path: per-ray march through the gain medium + frequency integration
// + deposit into image / I_ang.  Wave64 only; no MFMA (no dense contraction on
// this path); compiled with -ffp-contract=off so that the float32 march takes
// bit-for-bit the steps of RayTraceImageCPULoop.
//
// Mapping (one wavefront owns a tile of 64 consecutive rays, i.e. one pixel x
// angle tile in ASE mode, and is independent of every other wavefront):
//   phase A  lanes = rays.   Each lane marches its ray (Helper.h:404-521) and
//            leaves the per-ray record gvl/evl/ivl[L][3] in the wave's LDS
//            slab (layout [slot][lane], conflict-free), plus deposit indices.
//   phase B  lanes = frequencies.  The wave walks its 64 rays in order; ray r's
//            record is an LDS broadcast read; lane k integrates frequency k
//            through the L*3 sub-segments (Helper.h:543-581) reading one
//            coalesced lineshape row gv[cell][0..K) per sub-segment.
//   deposit  image rows accumulate in registers while consecutive rays hit the
//            same pixel and are flushed with one coalesced run of native f64
//            atomics per pixel change (RayTraceImageCPU.cpp:56-61); the
//            per-ray frequency sum for I_ang (RayTraceImageCPU.cpp:63-68) is a
//            wave-level shuffle reduction, parked in lane r and flushed as one
//            64-lane atomic instruction per tile.
// Waves pull tiles from one atomic counter (persistent waves): ray cost varies
// 30x across the image, a static grid would idle on the cheap pixels.
#include "rt_math.h"

namespace rt {

// ------------------------------------------------------------------- the kernel
extern "C" __global__ void __launch_bounds__(256)
rt_trace_kernel(const DevParams P)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = (int) (threadIdx.x >> 6);
    const int S    = P.L * RT_N_SUB;
    // this wave's LDS slab: g[S][64] e[S][64] c[S][64]
    float *lds_g = reinterpret_cast<float *>(lds_raw) + (size_t) wave * (size_t) S * WAVE * 3;
    float *lds_e = lds_g + (size_t) S * WAVE;
    int *lds_c   = reinterpret_cast<int *>(lds_e + (size_t) S * WAVE);
    const int K  = P.K;

    for (;;) {
        unsigned tile = 0;
        if (lane == 0)
            tile = atomicAdd(&P.ctl->next_tile, 1u);
        tile = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
        if (tile >= P.n_tiles)
            break;

        // ============================ phase A: lanes = rays ============================
        const unsigned long long ridx = (unsigned long long) tile * WAVE + (unsigned) lane;
        const bool have               = ridx < P.rays.count;
        rt_ray ray                    = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (have) {
            if (P.rays.list) {
                ray = P.rays.list[ridx];
            } else {
                // RayTraceImage.cpp:300-328: b fastest, then a, y, x; grids rounded to float
                unsigned ijkm = (unsigned) (P.rays.first + (long long) ridx * P.rays.stride);
                unsigned m    = ijkm % (unsigned) P.rays.ngb;
                unsigned q    = ijkm / (unsigned) P.rays.ngb;
                unsigned k    = q % (unsigned) P.rays.nga;
                q /= (unsigned) P.rays.nga;
                unsigned j = q % (unsigned) P.rays.ngy;
                unsigned i = q / (unsigned) P.rays.ngy;
                ray.x      = (float) P.rays.gx[i];
                ray.y      = (float) P.rays.gy[j];
                ray.a      = (float) P.rays.ga[k];
                ray.b      = (float) P.rays.gb[m];
            }
        }
        float px = ray.x, py = ray.y, pz = 0.0f;
        // Helper.h:409-411: tanf of an f32 product.  Computed as the f64 tan
        // rounded to float (equal to glibc tanf on the whole +-20 mrad range).
        float sx = (float) tan((double) (1e-3f * ray.a));
        float sy = (float) tan((double) (1e-3f * ray.b));
        float sz = 1.0f;
        if (P.method == 1) {
            sx = -sx;
            sy = -sy;
            sz = -sz;
        }
        renormalise(sx, sy, sz);

#ifdef RT_INSTRUMENT
        Inst inst;
#endif
        bool escaped   = !have; // lanes without a ray never enter the loops
        unsigned steps = 0;
        for (int seg = 0; seg < P.L; seg++) {
            const int ii         = (P.method == 1) ? P.N - seg - 1 : seg + 1;
            const DevGain &G     = P.gain[ii];
            const double *gx     = G.x;
            const double *gy     = G.y;
            const Node *node     = G.node;
            const int Nx         = G.Nx;
            const int Ny         = G.Ny;
            const float lo_x     = G.lo_x, hi_x = G.hi_x, lo_y = G.lo_y, hi_y = G.hi_y;
            const bool mirror_y  = G.mirror_y != 0;
            const double inv_hx  = G.inv_hx, inv_hy = G.inv_hy;
            float z              = 0.0f;
            for (int iz = 0; iz < RT_N_SUB; iz++) {
                const int is       = (P.method == 1) ? RT_N_SUB - iz - 1 : iz;
                const int slot     = (ii - 1) * RT_N_SUB + is;
                const float z_stop = (P.dz0 * ((float) iz + 1.0f) / RT_N_SUB);
                float gacc = 0.0f, eacc = 0.0f;
                int cell = 0;
                while (!escaped && z < 0.995f * z_stop) {
                    if (px < lo_x || px > hi_x || py < lo_y || py > hi_y || (double) (sz * sz) < 0.01) {
                        escaped = true;
                        break;
                    }
                    float ya    = mirror_y ? fabsf(py) : py;
                    uint32_t k1 = interval_index(gx, (uint32_t) Nx, (double) px, inv_hx);
                    uint32_t k2 = interval_index(gy, (uint32_t) Ny, (double) ya, inv_hy);
                    uint32_t c00 = (k1 - 1) + (k2 - 1) * (uint32_t) Nx;
                    double xc0 = gx[k1 - 1], xc1 = gx[k1];
                    double yc0 = gy[k2 - 1], yc1 = gy[k2];
                    Node a00 = node[c00], a10 = node[c00 + 1];
                    Node a01 = node[c00 + (uint32_t) Nx], a11 = node[c00 + (uint32_t) Nx + 1];
                    double hx = xc1 - xc0, hy = yc1 - yc0;
                    float u   = (float) (((double) px - xc0) / hx);
                    float v   = (float) (((double) ya - yc0) / hy);
                    float g0  = lerp2(u, v, a00.g0, a10.g0, a01.g0, a11.g0);
                    float E0  = 0.0f;
                    if (P.use_emis) {
                        E0 = lerp2(u, v, a00.E0, a10.E0, a01.E0, a11.E0);
                        E0 = E0 >= 0 ? E0 : 0.0f;
                    }
                    pz       = 0.0f;
                    float b0 = (float) (xc0 - 0.1 * hx);
                    float b1 = (float) (xc1 + 0.1 * hx);
                    float b2 = (float) (yc0 - 0.1 * hy);
                    float b3 = (float) (yc1 + 0.1 * hy);
                    if (mirror_y && k2 <= 1)
                        b2 = -b3;
                    float path = cross_cell(px, py, pz, sx, sy, sz, z_stop - z, xc0, xc1, yc0, yc1, b0, b1, b2,
                                            b3, a00.n, a10.n, a01.n, a11.n, mirror_y RT_INST_PASS);
                    RT_TICK(2);
                    z += fabsf(pz);
                    gacc += g0 * path;
                    eacc += E0 * path;
                    cell = (int) c00;
                    steps++;
                }
                lds_g[slot * WAVE + lane] = gacc;
                lds_e[slot * WAVE + lane] = eacc;
                lds_c[slot * WAVE + lane] = cell;
            }
        }
        escaped = escaped && have;

        // Helper.h:515-533 + RayTraceImageCPU.cpp:37-54: exit ray, seed factor, deposit cells
        unsigned flags = have ? F_VALID : 0u;
        if (escaped)
            flags |= F_ESCAPED;
        rt_ray out = ray;
        double fseed = 0.0;
        if (have) {
            if ((double) (sz * sz) < 0.01) {
                flags |= F_ERR1;
            } else {
                rt_ray r2;
                r2.x = px;
                r2.y = py;
                r2.a = (float) atan((double) (sx / sz)) * 1e3f;
                r2.b = (float) atan((double) (sy / sz)) * 1e3f;
                if (P.has_seed && !escaped) {
                    if (P.method == 1)
                        fseed = seed_factor(P.seed, (double) px, (double) py, (double) r2.a, (double) r2.b);
                    else if (P.method == 2)
                        fseed = seed_factor(P.seed, (double) ray.x, (double) ray.y, (double) ray.a, (double) ray.b);
                }
                if (P.method != 1) {
                    out   = r2;
                    out.a = -out.a;
                    out.b = -out.b;
                    if ((double) out.y < 0.0 && P.beam.y[0] >= 0.0)
                        out.y = -out.y;
                }
                if (P.probe_on)
                    P.probe.ray2[ridx] = r2;
            }
        }
        int pix = -1, ang = -1;
        if (have && !(flags & F_ERR1)) {
            int i1 = deposit_index(P.beam.nx, P.beam.x, P.beam.dx, (double) out.x);
            int i2 = deposit_index(P.beam.ny, P.beam.y, P.beam.dy, (double) out.y);
            int i3 = deposit_index(P.beam.na, P.beam.a, P.beam.da, (double) out.a);
            int i4 = deposit_index(P.beam.nb, P.beam.b, P.beam.db, (double) out.b);
            if (i1 >= 0 && i2 >= 0)
                pix = i1 + i2 * P.beam.nx;
            if (i3 >= 0 && i4 >= 0)
                ang = i3 + i4 * P.beam.na;
        }
        // A ray whose frequency pass is identically zero adds exactly +0.0
        // everywhere: with emission, when no sub-segment collected gain or
        // emissivity (every update is the identity);
        // gain-only, when the start intensity is zero (checked in phase B
        // against overflow of exp).
        if (have && !(flags & F_ERR1) && P.use_emis) {
            bool any = false;
            for (int s = 0; s < S; s++)
                any = any || lds_g[s * WAVE + lane] != 0.0f || lds_e[s * WAVE + lane] != 0.0f;
            if (!any)
                flags |= F_SKIP;
        }
        if (have && (flags & F_ERR1)) {
            atomicOr(&P.ctl->failure_code, 1u << 1);
            unsigned slot_f = atomicAdd(&P.ctl->n_failed, 1u);
            if (slot_f < RT_N_FAILED_MAX)
                P.ctl->failed[slot_f] = ray;
        }
        if (P.probe_on && have) {
            for (int s = 0; s < S; s++) {
                P.probe.gvl[ridx * (unsigned) S + (unsigned) s] = lds_g[s * WAVE + lane];
                P.probe.evl[ridx * (unsigned) S + (unsigned) s] = lds_e[s * WAVE + lane];
                P.probe.ivl[ridx * (unsigned) S + (unsigned) s] = lds_c[s * WAVE + lane];
            }
            P.probe.flags[ridx] = flags;
            P.probe.steps[ridx] = steps;
        }
        {
            unsigned tot_steps = wave_sum_u32(steps);
            unsigned tot_esc   = wave_sum_u32(escaped ? 1u : 0u);
            unsigned tot_skip  = wave_sum_u32((flags & F_SKIP) ? 1u : 0u);
            unsigned tot_rays  = wave_sum_u32(have ? 1u : 0u);
            if (lane == 0) {
                atomicAdd(&P.ctl->cell_steps, (unsigned long long) tot_steps);
                atomicAdd(&P.ctl->n_escaped, (unsigned long long) tot_esc);
                atomicAdd(&P.ctl->n_skipped, (unsigned long long) tot_skip);
                atomicAdd(&P.ctl->n_rays, (unsigned long long) tot_rays);
            }
        }
#ifdef RT_INSTRUMENT
        for (int i = 0; i < 3; i++) {
            unsigned tw = wave_sum_u32(inst.w[i]), ta = wave_sum_u32(inst.a[i]);
            if (lane == 0) {
                atomicAdd(&g_inst[2 * i], (unsigned long long) tw);
                atomicAdd(&g_inst[2 * i + 1], (unsigned long long) ta);
            }
        }
#endif
        // LDS writes of this wave are visible to its own later reads in program order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ========================= phase B: lanes = frequencies =========================
        if (P.debug & 1u)
            continue;
        const unsigned long long any_mask = __ballot((flags & F_VALID) != 0);
        const int n_in_tile               = (int) __popcll(any_mask);
        double angsum                     = 0.0;
        bool ray_failed                   = false; // lane r: ray r failed the neg/NaN scan
        for (int kc = 0; kc < K; kc += WAVE) {
            const int k     = kc + lane;
            const bool kact = k < K;
            const double dvk = kact ? 2.0 * P.beam.dv[k] : 0.0;
            const double sf4 = (P.has_seed && kact) ? P.seed.f[4][k] : 0.0;
            double acc  = 0.0;
            int cur_pix = -1;
            for (int r = 0; r < n_in_tile; r++) {
                const unsigned fl = (unsigned) __builtin_amdgcn_readlane((int) flags, r);
                if (fl & (F_ERR1 | F_SKIP))
                    continue;
                double Iv = 0.0;
                if (P.use_emis) {
                    // Helper.h:543-558, sub-segments in physical order
                    for (int s = 0; s < S; s++) {
                        const float gs = sgpr_f(lds_g[s * WAVE + r]);
                        const float es = sgpr_f(lds_e[s * WAVE + r]);
                        if (gs == 0.0f && es == 0.0f)
                            continue; // identity update
                        const int cell   = sgpr_i(lds_c[s * WAVE + r]);
                        const float *row = P.gain[s / RT_N_SUB + 1].gv + (size_t) cell * (size_t) K;
                        const float w    = kact ? row[k] : 0.0f;
                        const double gl  = (double) (gs * w);
                        const double el  = (double) (es * w);
                        if (fabs(gl) < 1e-3) {
                            Iv = el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) + Iv * (1.0 + gl * (1.0 + 0.5 * gl));
                        } else {
                            const double eg = exp(gl);
                            Iv              = el / gl * (eg - 1.0) + Iv * eg;
                        }
                    }
                } else {
                    // Helper.h:569-580, gain only
                    const double f0 = readlane_f64(fseed, r);
                    double gl       = 0.0;
                    for (int s = 0; s < S; s++) {
                        const float gs   = sgpr_f(lds_g[s * WAVE + r]);
                        const int cell   = sgpr_i(lds_c[s * WAVE + r]);
                        const float *row = P.gain[s / RT_N_SUB + 1].gv + (size_t) cell * (size_t) K;
                        const double w   = kact ? (double) row[k] : 0.0;
                        gl += (double) gs * w;
                    }
                    Iv = f0 * sf4;
                    // 0 * exp(gl) is exactly 0 unless exp overflows: skip the exp then
                    if (f0 != 0.0 || __ballot(gl > 700.0) != 0ull)
                        Iv *= exp(gl);
                }
                // Helper.h:582-594
                const bool bad_neg = kact && Iv < 0.0;
                const bool bad_nan = kact && Iv != Iv;
                const unsigned long long m_neg = __ballot(bad_neg);
                const unsigned long long m_nan = __ballot(bad_nan);
                if ((m_neg | m_nan) != 0ull) {
                    if (lane == r && !ray_failed) {
                        ray_failed = true;
                        atomicOr(&P.ctl->failure_code, m_neg ? (1u << 2) : (1u << 3));
                        unsigned slot_f = atomicAdd(&P.ctl->n_failed, 1u);
                        if (slot_f < RT_N_FAILED_MAX)
                            P.ctl->failed[slot_f] = ray;
                    }
                    continue;
                }
                // RayTraceImageCPU.cpp:56-61
                const int rpix = __builtin_amdgcn_readlane(pix, r);
                if (rpix != cur_pix) {
                    if (cur_pix >= 0 && kact)
                        unsafeAtomicAdd(&P.image[(size_t) cur_pix * (size_t) K + (size_t) k], acc);
                    acc     = 0.0;
                    cur_pix = rpix;
                }
                if (rpix >= 0)
                    acc += Iv * P.scale;
                // RayTraceImageCPU.cpp:63-68
                const int rang = __builtin_amdgcn_readlane(ang, r);
                if (rang >= 0) {
                    const double t = wave_sum_f64(dvk * Iv);
                    if (lane == r)
                        angsum += t;
                }
            }
            if (cur_pix >= 0 && kact)
                unsafeAtomicAdd(&P.image[(size_t) cur_pix * (size_t) K + (size_t) k], acc);
        }
        if (ang >= 0 && !ray_failed && !(flags & (F_ERR1 | F_SKIP)))
            unsafeAtomicAdd(&P.iang[ang], angsum);
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace rt
