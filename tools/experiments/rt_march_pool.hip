#pragma once
// rt_march_pool.hip -- EXPERIMENT, NOT PART OF THE PRODUCT (round 2; kept for the record).
// Result on the 6.4 M-ray stand-in: every march record bit-identical to the oracle, but 4.7 ms against the
// 2.05 ms of rt_march_kernel (trigger swept 16 ... 64).  One private spare per lane does not decouple [A] from
// [B]/[C]: after a swap the spare waits for [A], and the lane needs a READY spare again at the end of its very
// next cell step -- so between two [A] passes every lane advances exactly one cell step and then idles until
// enough lanes have done the same.  A deeper buffer (a wave-shared pool of >= 128 rays of 112 bytes) does not
// fit 16 waves in the 160 KB of LDS.  To try it: include this file behind rt_march.hip, add `pool_slot_off` to
// DevParams and launch it with 1024 threads and headers + intervals + 7 x 16 bytes per thread of LDS.
//
// The march of rt_march.hip with the cell set-up taken out of the per-iteration path.
//
// In rt_march_kernel every wave iteration runs block [A] (sub-segment bookkeeping + cell set-up, Helper.h:
// 430-504) for the ~30 % of its lanes that need it, at the price of a full pass of the block: [A] is ~40 % of
// the instructions of an iteration.  Here every lane owns TWO rays: the ACTIVE one lives in
// registers and only ever runs the cross-cell set-up [B] and the integrator step [C]; the SPARE one lives in a
// lane-private LDS slot and is either WAITING for [A], READY (its cell is set up) or EMPTY.  A lane whose
// active ray leaves its cell swaps it against its ready spare and goes on integrating; [A] runs as a separate
// pass over the waiting spares of the wave, only when enough of them have gathered -- so it runs at high lane
// occupancy and a fraction of the iterations.  New rays enter, and finished rays leave, in that pass.  The
// arithmetic of every block is that of rt_march.hip, statement for statement: the march record stays
// bit-identical to RayTraceImageCPULoop.
//
// LDS: headers + interval records of every length (the corner nodes stay in L2: measured +3 % on the plain
// kernel) and 7 x 16 bytes of slot per lane, stored as [group][thread] so that every access is a
// conflict-free ds_read/write_b128.
#include "rt_march.hip"

namespace rt {

// state of a lane's spare ray (kept in a register, mirrored by what the slot holds)
enum : int { SP_EMPTY = 0, SP_WAIT = 1, SP_READY = 2 };
// state of the active ray
enum : int { AC_NONE = 0, AC_XSETUP = 2, AC_STEP = 3, AC_NEEDA = 5 };

constexpr int POOL_GROUPS = 7; // 16-byte groups per slot

struct alignas(16) PoolQ {
    unsigned a, b, c, d;
};

__global__ void __launch_bounds__(1024) rt_march_pool_kernel(const DevParams P)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lane        = lane_id();
    const int L           = P.L;
    const int S           = L * RT_N_SUB;
    const unsigned n_rays = P.ray_end;
    const bool backward   = P.method == 1;
    const bool use_emis   = P.use_emis != 0;
    const unsigned CH     = P.chunk;
    const int TRIG        = (int) P.park; // spares that must wait for [A] before the pass runs

    // ---- LDS: [headers | intervals of every length] [slots] ----
    // the blob is [BlobGain[N]] then per length [Interval x[Nx]] [Interval y[Ny]] [Node[Nx*Ny]]: copy the
    // headers and, per length, the interval part, packed; lds_ix / lds_iy are the packed offsets
    const BlobGain *ghdr = reinterpret_cast<const BlobGain *>(P.blob);
    const unsigned hdr_bytes = (unsigned) (sizeof(BlobGain) * (unsigned) P.N + 15u) & ~15u;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(P.blob);
        uint4 *dst       = reinterpret_cast<uint4 *>(lds_raw);
        for (unsigned i = threadIdx.x; i < hdr_bytes / 16; i += blockDim.x)
            dst[i] = src[i];
        unsigned at = hdr_bytes;
        for (int ii = 1; ii < P.N; ii++) {
            const unsigned from = (unsigned) ghdr[ii].off_ix, len = (unsigned) ghdr[ii].off_node - from;
            const uint4 *s2     = reinterpret_cast<const uint4 *>(P.blob + from);
            uint4 *d2           = reinterpret_cast<uint4 *>(lds_raw + at);
            for (unsigned i = threadIdx.x; i < len / 16; i += blockDim.x)
                d2[i] = s2[i];
            at += len;
        }
    }
    __syncthreads();
    const BlobGain *hdr = reinterpret_cast<const BlobGain *>(lds_raw);
    unsigned char *slots = lds_raw + P.pool_slot_off;
    auto slot_q          = [&](int g) { return reinterpret_cast<PoolQ *>(slots + ((unsigned) g * blockDim.x + threadIdx.x) * 16u); };
    // packed LDS offset of length ii's interval records: headers, then the interval parts in order
    auto lds_ivx = [&](int ii) {
        unsigned at = hdr_bytes;
        for (int j = 1; j < ii; j++)
            at += (unsigned) hdr[j].off_node - (unsigned) hdr[j].off_ix;
        return at;
    };

    float zs0 = (P.dz0 * (0.0f + 1.0f) / RT_N_SUB), zs1 = (P.dz0 * (1.0f + 1.0f) / RT_N_SUB),
          zs2 = (P.dz0 * (2.0f + 1.0f) / RT_N_SUB);
    asm volatile("" : "+v"(zs0), "+v"(zs1), "+v"(zs2));

    unsigned chunk_next = 0, chunk_end = 0;
    bool more           = true;

    // ---- active ray (registers) ----
    int st        = AC_NONE;
    unsigned ridx = 0;
    int iz = 0, seg = 0, sub = 0;
    float z = 0.0f, z_stop = 0.0f;
    float px = 0, py = 0, pz = 0, sx = 0, sy = 0, sz = 1;
    float gacc = 0, eacc = 0;
    int cell_last = 0;
    unsigned steps = 0, any_bits = 0;
    bool mirror = false;
    int c00 = 0;
    float f00 = 1, f10 = 1, f01 = 1, f11 = 1;
    double dnx0 = 0, dnx1 = 0, dny0 = 0, dny1 = 0;
    double xc0 = 0, yc0 = 0, rwx = 1, rwy = 1;
    float wx = 1, wy = 1, b0 = 0, b1 = 0, b2 = 0, b3 = 0, g0 = 0, E0 = 0;
    float dzrem = 0, zc = 0, path = 0;
    float rx = 0, ry = 0, rz = 0, n = 0, n0 = 0, gxn = 0, gyn = 0, lim2 = 0, dzcap = 0, hsum = 0;
    // ---- spare ray ----
    int sp = SP_EMPTY;
    unsigned tot_steps = 0, tot_esc = 0, tot_rays = 0, tot_skip = 0;

    // slot groups: 0 {ridx, iz | seg << 2 | sub << 12, z, steps}  1 {gacc, eacc, cell_last, any_bits}
    //              2 {px, py, sx, sy}  3 {sz, c00, k1 | k2 << 16, g0}  4 {E0, -, -, -}  5 {n00, n10}  6 {n01, n11}
    unsigned guard = 0; // every wave must drain whatever happens: a bound far above any real iteration count
    for (;;) {
        if (++guard > (1u << 23))
            break;
        // ------------------------------------------------------------ does the [A] pass run?
        const unsigned long long m_wait  = __ballot(sp == SP_WAIT);
        const unsigned long long m_empty = __ballot(sp == SP_EMPTY);
        const unsigned long long m_busy  = __ballot((st == AC_XSETUP) | (st == AC_STEP));
        const unsigned long long m_needa = __ballot(st == AC_NEEDA);
        const int n_work = (int) __popcll(m_wait) + (more ? (int) __popcll(m_empty) : 0);
        if (m_busy == 0ull && m_needa == 0ull && m_wait == 0ull && __ballot(sp == SP_READY) == 0ull && !more)
            break;
        // lanes that can swap right away (a ready spare) count as busy for the purpose of the trigger
        const bool can_swap = __ballot(((st == AC_NEEDA) | (st == AC_NONE)) & (sp == SP_READY)) != 0ull;
        if (n_work > 0 && (n_work >= TRIG || (m_busy == 0ull && !can_swap))) {
            // ======================================================== [A] pass over the spares
            // -- new rays for the empty slots
            bool pend = sp == SP_WAIT; // this lane's spare is worked on in this pass
            // spare state in registers for the duration of the pass
            unsigned q_ridx = 0, q_steps = 0, q_any = 0;
            int q_iz = 0, q_seg = 0, q_sub = 0, q_cell = 0;
            float q_z = 0, q_gacc = 0, q_eacc = 0, q_px = 0, q_py = 0, q_sx = 0, q_sy = 0, q_sz = 1;
            bool q_esc = false;
            if (pend) {
                const PoolQ a = *slot_q(0), b = *slot_q(1), c = *slot_q(2);
                q_ridx  = a.a;
                q_iz    = (int) (a.b & 3u);
                q_seg   = (int) ((a.b >> 2) & 0x3ffu);
                q_sub   = (int) (a.b >> 12);
                q_z     = __uint_as_float(a.c);
                q_steps = a.d;
                q_gacc  = __uint_as_float(b.a);
                q_eacc  = __uint_as_float(b.b);
                q_cell  = (int) b.c;
                q_any   = b.d;
                q_px    = __uint_as_float(c.a);
                q_py    = __uint_as_float(c.b);
                q_sx    = __uint_as_float(c.c);
                q_sy    = __uint_as_float(c.d);
                q_sz    = __uint_as_float(slot_q(3)->a);
            }
            if (more && m_empty != 0ull) {
                const unsigned long long idle = m_empty;
                const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) idle, 0u));
                int need = (int) __popcll(idle), off = 0;
                bool got = false;
                while (need > 0) {
                    if (chunk_next == chunk_end) {
                        unsigned base = 0;
                        if (lane == 0)
                            base = P.ray_begin + atomicAdd(&P.ctl->next_tile[P.launch_id], CH);
                        base = (unsigned) __builtin_amdgcn_readfirstlane((int) base);
                        if (base >= n_rays) {
                            more = false;
                            break;
                        }
                        chunk_next = base;
                        chunk_end  = (n_rays - base < CH) ? n_rays : base + CH;
                    }
                    const int avail = (int) (chunk_end - chunk_next);
                    const int take  = avail < need ? avail : need;
                    if (sp == SP_EMPTY && !got && rank >= off && rank < off + take) {
                        q_ridx = chunk_next + (unsigned) (rank - off);
                        got    = true;
                    }
                    chunk_next += (unsigned) take;
                    need -= take;
                    off += take;
                }
                if (got) { // Helper.h:404-418
                    rt_ray ray;
                    float ta = 0, tb = 0;
                    load_ray(P.rays, q_ridx, ray, ta, tb, true);
                    q_px = ray.x;
                    q_py = ray.y;
                    q_sx = ta;
                    q_sy = tb;
                    q_sz = 1.0f;
                    if (backward) {
                        q_sx = -q_sx;
                        q_sy = -q_sy;
                        q_sz = -q_sz;
                    }
                    renormalise(q_sx, q_sy, q_sz);
                    q_seg = 0;
                    q_iz = 0;
                    q_sub = 0;
                    q_z = 0.0f;
                    q_gacc = 0.0f;
                    q_eacc = 0.0f;
                    q_cell = 0;
                    q_steps = 0;
                    q_any = 0;
                    q_esc = false;
                    pend = true;
                }
            }
            // -- [A1] / [A2] until every spare of the pass is READY or its ray is finished
            bool ready = false; // this lane's spare has its cell set up
            int q_c00 = 0, q_k = 0;
            float q_g0 = 0, q_E0 = 0;
            double q_n00 = 0, q_n10 = 0, q_n01 = 0, q_n11 = 0;
            for (int round = 0; round < 4 * S + 64 && __ballot(pend) != 0ull; round++) {
                if (pend) {
                    const int ii       = backward ? P.N - q_seg - 1 : q_seg + 1;
                    float q_zstop      = q_iz == 0 ? zs0 : (q_iz == 1 ? zs1 : zs2);
                    bool in_seg        = !q_esc & (q_z < 0.995f * q_zstop);
                    bool fin           = false;
                    if (!in_seg) {
                        // [A1] end of this sub-segment: commit its slot (Helper.h:501-503)
                        unsigned char *recp = P.rec + (size_t) q_ridx * P.rec_stride + 12 * (backward ? S - 1 - q_sub : q_sub);
                        *reinterpret_cast<RecSlot *>(recp) = RecSlot{ q_gacc, q_eacc, q_cell };
                        q_any |= (__float_as_uint(q_gacc) | __float_as_uint(q_eacc)) & 0x7fffffffu;
                        q_gacc = 0.0f;
                        q_eacc = 0.0f;
                        q_cell = 0;
                        q_sub++;
                        const bool wrap = q_iz == RT_N_SUB - 1;
                        fin             = q_esc | (q_sub == S);
                        q_iz            = wrap ? 0 : q_iz + 1;
                        q_z             = wrap ? 0.0f : q_z;
                        q_seg += (wrap & !fin) ? 1 : 0;
                        if (fin) {
                            // ---- ray finished
                            unsigned fl = F_VALID;
                            if (q_esc)
                                fl |= F_ESCAPED;
                            if (use_emis && q_any == 0u)
                                fl |= F_SKIP;
                            RecMeta m;
                            m.px          = q_px;
                            m.py          = q_py;
                            m.sx          = q_sx;
                            m.sy          = q_sy;
                            m.sz          = q_sz;
                            m.flags_steps = fl | ((unsigned) q_sub << REC_NDONE_SHIFT) |
                                            ((q_steps < 0xfffffu ? q_steps : 0xfffffu) << REC_STEPS_SHIFT);
                            *reinterpret_cast<RecMeta *>(P.rec + (size_t) q_ridx * P.rec_stride + 12 * (size_t) S) = m;
                            tot_steps += q_steps;
                            tot_esc += q_esc ? 1u : 0u;
                            tot_skip += (fl & F_SKIP) ? 1u : 0u;
                            tot_rays++;
                            pend = false;
                            sp   = SP_EMPTY;
                        }
                    } else {
                        // [A2] escape test + cell setup (Helper.h:465-497)
                        const BlobGain G = hdr[ii];
                        if ((q_px < G.lo_x) | (q_px > G.hi_x) | (q_py < G.lo_y) | (q_py > G.hi_y) | (q_sz * q_sz <= 0.01f)) {
                            q_esc = true; // its slot is committed by [A1] in the next round of this pass
                        } else {
                            const bool mir      = G.mirror_y != 0;
                            const unsigned lx   = lds_ivx(ii), ly = lx + (unsigned) G.Nx * (unsigned) sizeof(Interval);
                            const Interval *ivx = reinterpret_cast<const Interval *>(lds_raw + lx);
                            const Interval *ivy = reinterpret_cast<const Interval *>(lds_raw + ly);
                            auto interval_at    = [&](unsigned off, int u) {
                                return *reinterpret_cast<const Interval *>(lds_raw + (off + (unsigned) u * (unsigned) sizeof(Interval)));
                            };
                            auto node_at = [&](int c) {
                                return *reinterpret_cast<const Node *>(P.blob + ((unsigned) G.off_node + (unsigned) c * (unsigned) sizeof(Node)));
                            };
                            const float ya   = mir ? fabsf(q_py) : q_py;
                            const double pxd = (double) q_px, yad = (double) ya;
                            int k1     = guess_interval(G.Nx, G.x0f, G.inv_hxf, q_px);
                            int k2     = guess_interval(G.Ny, G.y0f, G.inv_hyf, ya);
                            Interval X = interval_at(lx, k1), Y = interval_at(ly, k2);
                            const bool ok = ((k1 == 1) | (X.lo < pxd)) & ((k1 == G.Nx - 1) | (X.hi >= pxd)) &
                                            ((k2 == 1) | (Y.lo < yad)) & ((k2 == G.Ny - 1) | (Y.hi >= yad));
                            if (!ok) {
                                k1 = bisect_interval(ivx, G.Nx, pxd);
                                k2 = bisect_interval(ivy, G.Ny, yad);
                                X  = interval_at(lx, k1);
                                Y  = interval_at(ly, k2);
                            }
                            q_c00          = (k1 - 1) + (k2 - 1) * G.Nx;
                            const Node a00 = node_at(q_c00), a10 = node_at(q_c00 + 1);
                            const Node a01 = node_at(q_c00 + G.Nx), a11 = node_at(q_c00 + G.Nx + 1);
                            const float u = (float) div_by_recip<true>(pxd - X.lo, X.hi - X.lo, X.rh);
                            const float v = (float) div_by_recip<true>(yad - Y.lo, Y.hi - Y.lo, Y.rh);
                            q_g0          = lerp2(u, v, a00.g0, a10.g0, a01.g0, a11.g0);
                            q_E0          = 0.0f;
                            if (use_emis) {
                                q_E0 = lerp2(u, v, a00.E0, a10.E0, a01.E0, a11.E0);
                                q_E0 = q_E0 >= 0 ? q_E0 : 0.0f;
                            }
                            const float dzr = q_zstop - q_z;
                            if ((q_px > X.b_lo) & (q_px < X.b_hi) & (ya > Y.b_lo) & (ya < Y.b_hi) & (dzr > 0.0f)) {
                                q_k   = k1 | (k2 << 16);
                                q_n00 = a00.n;
                                q_n10 = a10.n;
                                q_n01 = a01.n;
                                q_n11 = a11.n;
                                ready = true;
                                pend  = false;
                            } else {
                                // no cross-cell iteration at all: the cell step still counts (Helper.h:498-503)
                                q_z += fabsf(0.0f);
                                q_gacc += q_g0 * 0.0f;
                                q_eacc += q_E0 * 0.0f;
                                q_cell = q_c00;
                                q_steps++;
                            }
                        }
                    }
                }
            }
            // -- what the pass leaves in the slot
            if (ready) {
                *slot_q(0) = PoolQ{ q_ridx, (unsigned) q_iz | ((unsigned) q_seg << 2) | ((unsigned) q_sub << 12), __float_as_uint(q_z), q_steps };
                *slot_q(1) = PoolQ{ __float_as_uint(q_gacc), __float_as_uint(q_eacc), (unsigned) q_cell, q_any };
                *slot_q(2) = PoolQ{ __float_as_uint(q_px), __float_as_uint(q_py), __float_as_uint(q_sx), __float_as_uint(q_sy) };
                *slot_q(3) = PoolQ{ __float_as_uint(q_sz), (unsigned) q_c00, (unsigned) q_k, __float_as_uint(q_g0) };
                slot_q(4)->a = __float_as_uint(q_E0);
                *reinterpret_cast<double2 *>(slot_q(5)) = make_double2(q_n00, q_n10);
                *reinterpret_cast<double2 *>(slot_q(6)) = make_double2(q_n01, q_n11);
                sp = SP_READY;
            }
        }

        // ------------------------------------------------------------ swap: the active ray left its cell
        if ((st == AC_NEEDA) | (st == AC_NONE)) {
            if (sp == SP_READY) {
                // the spare becomes the active ray; the old active ray (if any) waits in the slot for [A]
                const PoolQ a = *slot_q(0), b = *slot_q(1), c = *slot_q(2), d = *slot_q(3);
                const float e0       = __uint_as_float(slot_q(4)->a);
                const double2 nlo    = *reinterpret_cast<const double2 *>(slot_q(5));
                const double2 nhi    = *reinterpret_cast<const double2 *>(slot_q(6));
                if (st == AC_NEEDA) {
                    *slot_q(0) = PoolQ{ ridx, (unsigned) iz | ((unsigned) seg << 2) | ((unsigned) sub << 12), __float_as_uint(z), steps };
                    *slot_q(1) = PoolQ{ __float_as_uint(gacc), __float_as_uint(eacc), (unsigned) cell_last, any_bits };
                    *slot_q(2) = PoolQ{ __float_as_uint(px), __float_as_uint(py), __float_as_uint(sx), __float_as_uint(sy) };
                    slot_q(3)->a = __float_as_uint(sz);
                    sp = SP_WAIT;
                } else {
                    sp = SP_EMPTY;
                }
                ridx      = a.a;
                iz        = (int) (a.b & 3u);
                seg       = (int) ((a.b >> 2) & 0x3ffu);
                sub       = (int) (a.b >> 12);
                z         = __uint_as_float(a.c);
                steps     = a.d;
                gacc      = __uint_as_float(b.a);
                eacc      = __uint_as_float(b.b);
                cell_last = (int) b.c;
                any_bits  = b.d;
                px        = __uint_as_float(c.a);
                py        = __uint_as_float(c.b);
                sx        = __uint_as_float(c.c);
                sy        = __uint_as_float(c.d);
                sz        = __uint_as_float(d.a);
                c00       = (int) d.b;
                g0        = __uint_as_float(d.d);
                E0        = e0;
                const int k1 = (int) (d.c & 0xffffu), k2 = (int) (d.c >> 16);
                const int ii = backward ? P.N - seg - 1 : seg + 1;
                const BlobGain *G = hdr + ii;
                mirror            = G->mirror_y != 0;
                const unsigned lx = lds_ivx(ii), ly = lx + (unsigned) G->Nx * (unsigned) sizeof(Interval);
                const Interval X  = *reinterpret_cast<const Interval *>(lds_raw + (lx + (unsigned) k1 * (unsigned) sizeof(Interval)));
                const Interval Y  = *reinterpret_cast<const Interval *>(lds_raw + (ly + (unsigned) k2 * (unsigned) sizeof(Interval)));
                f00  = (float) nlo.x;
                f10  = (float) nlo.y;
                f01  = (float) nhi.x;
                f11  = (float) nhi.y;
                dnx0 = nlo.y - nlo.x;
                dnx1 = nhi.y - nhi.x;
                dny0 = nhi.x - nlo.x;
                dny1 = nhi.y - nlo.y;
                xc0  = X.lo;
                yc0  = Y.lo;
                rwx  = X.rw;
                rwy  = Y.rw;
                wx   = X.w;
                wy   = Y.w;
                b0   = X.b_lo;
                b1   = X.b_hi;
                b2   = Y.b_lo;
                b3   = Y.b_hi;
                z_stop = iz == 0 ? zs0 : (iz == 1 ? zs1 : zs2);
                pz     = 0.0f;
                zc     = 0.0f;
                path   = 0.0f;
                dzrem  = z_stop - z;
                st     = AC_XSETUP;
            } else if (st == AC_NEEDA && sp == SP_EMPTY) {
                // nothing to swap with: the active ray goes to wait for [A]
                *slot_q(0) = PoolQ{ ridx, (unsigned) iz | ((unsigned) seg << 2) | ((unsigned) sub << 12), __float_as_uint(z), steps };
                *slot_q(1) = PoolQ{ __float_as_uint(gacc), __float_as_uint(eacc), (unsigned) cell_last, any_bits };
                *slot_q(2) = PoolQ{ __float_as_uint(px), __float_as_uint(py), __float_as_uint(sx), __float_as_uint(sy) };
                slot_q(3)->a = __float_as_uint(sz);
                sp = SP_WAIT;
                st = AC_NONE;
            }
        }

        // ------------------------------------------------------------ [B] cross-cell setup (Helper.h:328-342)
        if (st == AC_XSETUP) {
            const float ya   = mirror ? fabsf(py) : py;
            const double dwx = (double) wx, dwy = (double) wy;
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
            dzcap = P.c_cap * lim2;
            rx    = 0.0f;
            ry    = 0.0f;
            rz    = 0.0f;
            n     = n0;
            hsum  = 0.0f;
            st    = AC_STEP;
        }

        // ------------------------------------------------------------ [C] one integrator step (Helper.h:279-311)
        if (st == AC_STEP) {
            const float lim0 = 0.1f * wx, lim1 = 0.1f * wy;
            bool run;
            {
                n              = n0 + rx * gxn + ry * gyn;
                const float rn = 1.0f / n;
                float a0       = sx * gxn + sy * gyn + 1e-12f;
                float t        = div_by_recip_signed(a0, n, rn);
                float qx       = div_by_recip_signed(gxn, n, rn);
                float qy       = div_by_recip_signed(gyn, n, rn);
                if (__ballot(fminf(fminf(fabsf(a0), fabsf(gxn)), fabsf(gyn)) < 1e-29f) != 0ull) {
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
                float fx = qx - sx * t;
                float fy = qy - sy * t;
                float fz = -sz * t;
                float h  = P.c_h1 / fabsf(t);
                h        = h < dzcap ? h : dzcap;
                float h2 = (1.0001f * (lim2 - fabsf(rz))) / fabsf(sz);
                float h3 = (P.c_h3 * (fabsf(sx) + 5e-4f)) / (fabsf(fx) + 1e-8f);
                float h4 = (P.c_h3 * (fabsf(sy) + 5e-4f)) / (fabsf(fy) + 1e-8f);
                h        = h < h2 ? h : h2;
                h        = h < h3 ? h : h3;
                h        = h < h4 ? h : h4;
                float ht = h * t;
                const float R3 = 1.0f / 3.0f, R6 = 1.0f / 6.0f, R12 = 1.0f / 12.0f;
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
                run = (fabsf(rx) < lim0) & (fabsf(ry) < lim1) & (fabsf(rz) < lim2) & (fabsf(n - n0) < 0.05f);
            }
            if (!run) {
                path += hsum;
                px += rx;
                py += ry;
                pz += rz;
                zc += fabsf(rz);
                const float ya = mirror ? fabsf(py) : py;
                if ((px > b0) & (px < b1) & (ya > b2) & (ya < b3) & ((double) zc < 0.999 * (double) dzrem)) {
                    st = AC_XSETUP;
                } else {
                    z += fabsf(pz);
                    gacc += g0 * path;
                    eacc += E0 * path;
                    cell_last = c00;
                    steps++;
                    st = AC_NEEDA;
                }
            }
        }
    }

    // ---- launch totals ----
    {
        unsigned s = wave_sum_u32(tot_steps), e = wave_sum_u32(tot_esc);
        unsigned k = wave_sum_u32(tot_skip), r = wave_sum_u32(tot_rays);
        if (lane == 0) {
            atomicAdd(&P.ctl->cell_steps, (unsigned long long) s);
            atomicAdd(&P.ctl->n_escaped, (unsigned long long) e);
            atomicAdd(&P.ctl->n_skipped, (unsigned long long) k);
            atomicAdd(&P.ctl->n_rays, (unsigned long long) r);
        }
    }
}

} // namespace rt
