// rt_freq.hip -- kernel B: frequency integration + deposit (Helper.h:515-594,
// RayTraceImageCPU.cpp:37-68) with lanes = rays.
//
// One wavefront owns a tile of 64 consecutive rays; lane r integrates ray r over
// all K frequencies, VEC frequencies at a time (VEC independent dependency chains
// per lane; the lineshape row gv[cell][k..k+VEC) is one 4*VEC-byte load per
// sub-segment).  Every lane is busy for the whole tile -- the regular half of the
// path sees none of the march's divergence -- and the two reductions of the
// deposit become cheap:
//   I_ang : sum over k of 2 dv_k Iv_k is a sequential sum inside the lane, in the
//           CPU's own order (RayTraceImageCPU.cpp:63-68); one LDS atomic per ray
//           into the work-group's private I_ang, flushed once per work-group with
//           coalesced native f64 atomics.
//   image : consecutive rays hit the same pixel (ASE: all na*nb rays of a pixel;
//           seeded: runs of neighbouring source angles), so the per-pixel sum over
//           rays is a segmented wave scan over runs of equal pixel index (shuffles,
//           masks built once per tile) and one atomic per run and frequency.
#include "rt_march.hip"

namespace rt {

template <int VEC> struct FVec;
template <> struct FVec<1> { float v[1]; };
template <> struct alignas(8) FVec<2> { float v[2]; };
template <> struct alignas(16) FVec<4> { float v[4]; };

// Helper.h:549-557, one (sub-segment, frequency) update with emission
__device__ __forceinline__ double ase_update(double Iv, float gs, float es, float w)
{
    const double gl = (double) (gs * w); // f32 product, then widened (Helper.h:549-550)
    const double el = (double) (es * w);
    if (fabs(gl) < 1e-3)
        return el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) + Iv * (1.0 + gl * (1.0 + 0.5 * gl));
    const double eg = exp(gl);
    return el / gl * (eg - 1.0) + Iv * eg;
}

// uniform-grid shortcut of deposit_index (RayTraceImageCPU.cpp:11-16): the grids
// of the beam are uniform (create_image checks it); guess the cell arithmetically,
// verify the two defining inequalities of findfirstsingle, else bisect.
__device__ __forceinline__ int deposit_index_fast(int n, const double *g, double d, double v)
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
    int u = (int) ((t - g0) / d) + 1;
    u     = u < 1 ? 1 : (u > n - 1 ? n - 1 : u);
    // first_not_below on [g0, gl]: unique u in [1, n-1] with g[u-1] < t <= g[u]
    // (u = 1 also when t == g0, where the bisection never tests g[0])
    if ((u == 1 || g[u - 1] < t) && g[u] >= t)
        return u;
    return first_not_below(g, n, t);
}

template <int SF, int VEC>
__device__ __forceinline__ void freq_tile(const DevParams &P, double *lds_iang, const unsigned tile, const int lane)
{
    const int S           = SF ? SF : P.L * RT_N_SUB;
    const int K           = P.K;
    const unsigned n_rays = (unsigned) P.rays.count;
    const unsigned ridx   = tile * WAVE + (unsigned) lane;
    const bool have       = ridx < n_rays;
    const unsigned char *rec = P.rec + (size_t) (have ? ridx : 0) * P.rec_stride;
    const bool use_emis   = P.use_emis != 0;

    // ---- per-ray preamble: exit ray, seed factor, deposit cells ------------------
    unsigned fl = 0, steps = 0;
    rt_ray ray  = { 0, 0, 0, 0 };
    RecMeta m   = { 0, 0, 0, 0, 1, 0 };
    if (have) {
        m     = *reinterpret_cast<const RecMeta *>(rec + 12 * (size_t) S);
        fl    = m.flags_steps & 0xffu;
        steps = m.flags_steps >> 8;
        float ta, tb;
        load_ray(P.rays, ridx, ray, ta, tb, false);
    }
    bool err1   = have && (double) (m.sz * m.sz) < 0.01; // Helper.h:515
    rt_ray out  = ray;
    double f0   = 0.0;
    int pix = -1, ang = -1;
    if (have && !err1) {
        const bool need_exit = P.method != 1 || P.has_seed || P.probe_on;
        rt_ray r2 = { m.px, m.py, 0.0f, 0.0f };
        if (need_exit) {
            // Helper.h:518-521: atanf(s.x / s.z) * 1e3f
            r2.a = (float) atan((double) (m.sx / m.sz)) * 1e3f;
            r2.b = (float) atan((double) (m.sy / m.sz)) * 1e3f;
        }
        if (P.has_seed && !(fl & F_ESCAPED)) { // Helper.h:523-533
            if (P.method == 1)
                f0 = seed_factor(P.seed, (double) m.px, (double) m.py, (double) r2.a, (double) r2.b);
            else
                f0 = seed_factor(P.seed, (double) ray.x, (double) ray.y, (double) ray.a, (double) ray.b);
        }
        if (P.method != 1) { // RayTraceImageCPU.cpp:37-49
            out   = r2;
            out.a = -out.a;
            out.b = -out.b;
            if ((double) out.y < 0.0 && P.beam.y[0] >= 0.0)
                out.y = -out.y;
        }
        if (P.probe_on)
            P.probe.ray2[ridx] = r2;
        const int i1 = deposit_index_fast(P.beam.nx, P.beam.x, P.beam.dx, (double) out.x);
        const int i2 = deposit_index_fast(P.beam.ny, P.beam.y, P.beam.dy, (double) out.y);
        const int i3 = deposit_index_fast(P.beam.na, P.beam.a, P.beam.da, (double) out.a);
        const int i4 = deposit_index_fast(P.beam.nb, P.beam.b, P.beam.db, (double) out.b);
        if (i1 >= 0 && i2 >= 0)
            pix = i1 + i2 * P.beam.nx;
        if (i3 >= 0 && i4 >= 0)
            ang = i3 + i4 * P.beam.na;
    }
    if (have && P.probe_on) {
        P.probe.flags[ridx] = fl | (err1 ? F_ERR1 : 0u);
        P.probe.steps[ridx] = steps;
    }
    if (err1) { // error -1: the ray is reported and deposits nothing
        atomicOr(&P.ctl->failure_code, 1u << 1);
        unsigned slot_f = atomicAdd(&P.ctl->n_failed, 1u);
        if (slot_f < RT_N_FAILED_MAX)
            P.ctl->failed[slot_f] = ray;
    }
    const bool live = have && !err1 && !(fl & F_SKIP);
    if (__ballot(live) == 0ull)
        return;
    if (!live) {
        pix = -1;
        ang = -1;
    }

    // ---- runs of equal pixel index (built once per tile) --------------------------
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
    bool addm[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
        addm[i] = (lane - (1 << i)) >= run_start;

    // ---- the march record of this lane's ray ---------------------------------------
    float gs[SF ? SF : 1], es[SF ? SF : 1];
    int cs[SF ? SF : 1];
    if (SF) {
#pragma unroll
        for (int s = 0; s < SF; s++) {
            gs[s] = reinterpret_cast<const float *>(rec)[s];
            es[s] = reinterpret_cast<const float *>(rec)[SF + s];
            cs[s] = reinterpret_cast<const int *>(rec)[2 * SF + s];
        }
    }

    double angsum = 0.0;
    bool bad_neg = false, bad_nan = false;
    double *img_row = P.image + (size_t) (pix >= 0 ? pix : 0) * (size_t) K;
    for (int kb = 0; kb < K; kb += VEC) {
        double Iv[VEC];
        if (use_emis) {
#pragma unroll
            for (int j = 0; j < VEC; j++)
                Iv[j] = 0.0;
            if (SF) {
                FVec<VEC> w[SF ? SF : 1];
#pragma unroll
                for (int s = 0; s < SF; s++) {
                    const float *row = P.gain[s / RT_N_SUB + 1].gv + (size_t) cs[s] * (size_t) K + kb;
                    w[s]             = *reinterpret_cast<const FVec<VEC> *>(row);
                }
#pragma unroll
                for (int s = 0; s < SF; s++) {
                    if (gs[s] != 0.0f || es[s] != 0.0f) { // else the update is the identity
#pragma unroll
                        for (int j = 0; j < VEC; j++)
                            Iv[j] = ase_update(Iv[j], gs[s], es[s], w[s].v[j]);
                    }
                }
            } else {
                for (int s = 0; s < S; s++) {
                    const float g1 = reinterpret_cast<const float *>(rec)[s];
                    const float e1 = reinterpret_cast<const float *>(rec)[S + s];
                    const int c1   = reinterpret_cast<const int *>(rec)[2 * S + s];
                    if (g1 != 0.0f || e1 != 0.0f) {
                        const float *row  = P.gain[s / RT_N_SUB + 1].gv + (size_t) c1 * (size_t) K + kb;
                        const FVec<VEC> w = *reinterpret_cast<const FVec<VEC> *>(row);
#pragma unroll
                        for (int j = 0; j < VEC; j++)
                            Iv[j] = ase_update(Iv[j], g1, e1, w.v[j]);
                    }
                }
            }
        } else {
            // gain only, Helper.h:569-580: f64 products summed in sub-segment order
            double gl[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j++)
                gl[j] = 0.0;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const float g1    = SF ? gs[SF ? s : 0] : reinterpret_cast<const float *>(rec)[s];
                const int c1      = SF ? cs[SF ? s : 0] : reinterpret_cast<const int *>(rec)[2 * S + s];
                const float *row  = P.gain[s / RT_N_SUB + 1].gv + (size_t) c1 * (size_t) K + kb;
                const FVec<VEC> w = *reinterpret_cast<const FVec<VEC> *>(row);
#pragma unroll
                for (int j = 0; j < VEC; j++)
                    gl[j] += (double) g1 * (double) w.v[j];
            }
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                Iv[j] = f0 * P.seed.f[4][kb + j];
                // 0 * exp(gl) is exactly 0 unless exp overflows: skip the exp then
                if (f0 != 0.0 || gl[j] > 700.0)
                    Iv[j] *= exp(gl[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const double iv = live ? Iv[j] : 0.0;
            bad_neg         = bad_neg || iv < 0.0; // Helper.h:582-594
            bad_nan         = bad_nan || iv != iv;
            angsum += (2.0 * P.beam.dv[kb + j]) * iv; // RayTraceImageCPU.cpp:66
            // RayTraceImageCPU.cpp:59, summed over the run of rays that share the pixel
            double v = pix >= 0 ? iv * P.scale : 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double t = __shfl_up(v, 1 << i, WAVE);
                if (addm[i])
                    v += t;
            }
            if (tail)
                unsafeAtomicAdd(&img_row[kb + j], v);
        }
    }
    if (live && (bad_neg || bad_nan)) {
        atomicOr(&P.ctl->failure_code, bad_neg ? (1u << 2) : (1u << 3));
        unsigned slot_f = atomicAdd(&P.ctl->n_failed, 1u);
        if (slot_f < RT_N_FAILED_MAX)
            P.ctl->failed[slot_f] = ray;
    }
    if (ang >= 0) {
        if (lds_iang)
            unsafeAtomicAdd(&lds_iang[ang], angsum);
        else
            unsafeAtomicAdd(&P.iang[ang], angsum);
    }
}

template <int SF, int VEC>
__global__ void __launch_bounds__(256) rt_freq_kernel(const DevParams P, const int iang_in_lds)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *lds_iang = iang_in_lds ? reinterpret_cast<double *>(lds_raw) : nullptr;
    const int n_ang  = P.beam.na * P.beam.nb;
    if (lds_iang) {
        for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x)
            lds_iang[c] = 0.0;
        __syncthreads();
    }
    const int lane = lane_id();
    for (;;) {
        unsigned tile = 0;
        if (lane == 0)
            tile = atomicAdd(&P.ctl->next_tile_b, 1u);
        tile = (unsigned) __builtin_amdgcn_readfirstlane((int) tile);
        if (tile >= P.n_tiles)
            break;
        freq_tile<SF, VEC>(P, lds_iang, tile, lane);
    }
    if (lds_iang) {
        __syncthreads();
        for (int c = (int) threadIdx.x; c < n_ang; c += (int) blockDim.x) {
            const double v = lds_iang[c];
            if (v != 0.0)
                unsafeAtomicAdd(&P.iang[c], v);
        }
    }
}

} // namespace rt
