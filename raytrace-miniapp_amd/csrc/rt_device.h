// rt_device.h -- device-side records of the HIP ray-trace backend (gfx950).
//
// Layout in HBM (one arena per plan, packed by rt_hip_plan_create):
//   march blob (BlobGain, Interval below): per length ii = 1..N-1 (gain[0] is never read on the
//     path except for the E0 != NULL test, src/common/RayTraceImageHelper.h:402,435-441)
//     x[Nx], y[Ny] (f64), reciprocal pairs per grid interval, and
//     node[Nx*Ny] = {double n; float g0; float E0}: the three quantities a cell-step
//     gathers at each of the 4 cell corners (Helper.h:474-489) fused into one 16-byte
//     record (corner pairs are adjacent).  Copied into LDS by the march kernel.
//   gv[Nx*Ny][K] float per length: lineshape rows, k fastest (frequency kernel)
//   beam grids x,y,a,b,dv (double), seed tables (double), tangent tables,
//   the ray list (16 B/ray) when rays are given explicitly,
//   march records (96 B/ray for N = 3, tile-wise: 64 rays share a block) between the two kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

namespace rt {

constexpr int WAVE = 64; // lanes of a wavefront; a frequency tile is WAVE consecutive rays

struct alignas(16) Node {
    double n;
    float g0;
    float E0;
};

// Per-length table that stays in HBM/L2: the lineshape rows read by the frequency kernel.
// (Everything the march reads lives in the march blob below.)
struct DevGain {
    const float *gv; // [Nx*Ny][Kp], k fastest, columns K..Kp-1 zero
};

// Header of one length inside the "march blob": everything the march gathers, laid
// out so that a work-group can copy the whole blob into LDS verbatim.
//   blob = BlobGain[N] | per length: Interval x[Nx] | Interval y[Ny] | Node[Nx*Ny]
// off_* are byte offsets from the start of the blob (16-byte aligned).
struct alignas(16) BlobGain {
    float lo_x, hi_x, lo_y, hi_y; // plasma box as floats (Helper.h:445-453), lo_y = -hi_y if mirrored
    int Nx, Ny, mirror_y, off_ix;
    int off_iy, off_node, pad0, pad1;
    float x0f, y0f;               // x[0], y[0]
    float inv_hxf, inv_hyf;       // (Nx-1)/(x[Nx-1]-x[0]): index guess on uniform grids
};

// One grid interval (g[u-1], g[u]] of one axis, u = 1 .. n-1 (entry 0 unused): everything a
// cell setup derives from the two coordinates (Helper.h:465-497, :323-324), computed once on
// the host with the same IEEE operations, so that the setup is one 48-byte gather per axis.
// rh and rw are the correctly rounded reciprocals of the two divisors a cell uses
// (Helper.h:482-483 and :330-334): the kernel divides by multiplication + exact residual
// correction (rt_math.h, div_by_recip).
struct alignas(16) Interval {
    // first 32 bytes: what stays in the lane's registers while the ray is in the cell (two 16-byte reads straight
    // into the lane state, rt_march.hip block [A2])
    double lo;     // g[u-1]
    double rw;     // RN(1 / (double) (float) (hi - lo))
    float w;       // (float) (hi - lo)
    float b_lo;    // (float) (lo - 0.1 (hi - lo)); mirrored y axis, u = 1: -b_hi (Helper.h:494-495)
    float b_hi;    // (float) (hi + 0.1 (hi - lo))
    float pad;
    // last 16 bytes: used by the cell set-up only
    double hi;     // g[u]
    double rh;     // RN(1 / (hi - lo))
};

struct DevSeed {
    const double *x[5];
    const double *f[5];
    int dim[5];
    int pad;
    double f0;
};

struct DevBeam {
    const double *x, *y, *a, *b, *dv;
    int nx, ny, na, nb, nv;
    int pad;
    double dx, dy, da, db;
    double inv_dx, inv_dy, inv_da, inv_db; // 1/d, for the deposit-cell guess (verified against the grid)
    double g_first[4], g_last[4];          // x[0], y[0], a[0], b[0] and the last grid values (no load for them)
};

struct DevRays {
    const rt_ray *list; // explicit list, or NULL -> generated from the grids
    const float *sxy;   // list mode: {tan(1e-3 a), tan(1e-3 b)} per ray (rt_tan_kernel)
    const double *gx, *gy, *ga, *gb;
    const float *tan_a, *tan_b; // grid mode: tanf(1e-3f * (float) ga[k]), host-computed
    // grid mode, forward method with a seed: the seed profile is a product of one factor per
    // grid axis (Helper.h:230-244), tabulated per grid point by rt_seed_tab_kernel with the very
    // routine the per-ray path uses: sf = [ngx | ngy | nga | ngb] factors, sin = inside the
    // profile's support (1) or not (0).  NULL: evaluate per ray.
    const double *sf;
    const unsigned char *sin;
    int ngx, ngy, nga, ngb;
    long long first, stride;
    unsigned long long count;
    // exact division of a ray number below 2^31 by ngb, nga, ngy (in that order) as multiply-high + shift:
    // x / d = mulhi(x, mul) >> sh with mul = floor(2^(31+s) / d) + 1, s = ceil(log2 d), sh = s - 1 (rt_raygrid.hip,
    // magic_u31); d = 1 is marked by mul = 0 (then x / d = x)
    unsigned div_mul[3], div_sh[3];
};

// Per-ray march record: S slots + one meta block,
//   RecSlot slot[S];    {gvl, evl, ivl} of each sub-segment, S = (N-1)*3, Helper.h:386-388
//                       (one 12-byte store when a sub-segment ends)
//   RecMeta meta;       exit position / direction, flags | n_done << 4 | steps << 12
// laid out TILE-WISE since round 5: the 64 rays 64 t .. 64 t + 63 (a tile of the frequency pass) share a block of
// 64 * rec_stride bytes at rec + t * 64 * rec_stride, slot s of ray 64 t + l at s * 768 + l * 12 inside it, the meta block
// of that ray at S * 768 + l * 24 (rec_slot_off / rec_meta_off below).  The frequency pass reads slot s of its 64 lanes
// as ONE contiguous 768-byte run (round 4: 12 bytes every 96), and the 32-byte sector a slot store of the march lands in holds
// the same slot of the neighbouring rays -- rays a launch angle apart, which end that sub-segment within a few iterations
// of each other and merge in L2 -- instead of the ray's own other slots, written tens of microseconds apart.
// Only the first n_done sub-segments in marching order were entered and have their slot written
// (an escaped ray stops early, Helper.h:465-469, 505-511); readers take the others as zero:
// rec_slot() below.  Forward march: slots 0 .. n_done-1, backward: S-n_done .. S-1.
// rec_stride = round16(12 S + 24).
struct RecSlot {
    float g, e; // sum of g0 * path, E0 * path over the cells of the sub-segment
    int c;      // index of its last cell
};
struct RecMeta {
    float px, py, sx, sy, sz;
    unsigned flags_steps;
};

// Zeroed by a memset node before every run.
struct DevCtl {
    // march kernel: next unreserved chunk of rays of each of the 8 shards of a launch's ray range, per launch of a
    // run (launch_id); 64 bytes apart (see next_tile_f below)
    unsigned int next_tile[8][8][16];
    // frequency kernel: next tile of each of the 8 shards of a launch's tile range, per launch of a run (freq_id);
    // 16 words = 64 bytes apart: a returning atomic on one word is served at ~88 per microsecond chip-wide, eight
    // words on eight lines at eight times that (MI355X_MICROARCH.md "dequeue")
    unsigned int next_tile_f[4][8][16];
    unsigned int failure_code;
    unsigned int n_failed;
    unsigned long long cell_steps;
    unsigned long long n_escaped;
    unsigned long long n_skipped;
    unsigned long long n_rays;
    rt_ray failed[RT_N_FAILED_MAX];
};

// probe outputs of the frequency kernel (gvl / evl / ivl are read back from the records)
struct DevProbe {
    rt_ray *ray2;
    uint32_t *flags;
    uint32_t *steps;
};

struct DevParams {
    int N, L, K, method;
    int use_emis, has_seed;
    float dz0;
    int probe_on;
    // Kp = K rounded up to a multiple of 4: row stride of the lineshape tables and length of
    // the two per-frequency vectors (beam.dv, seed.f[4]) on the device, zero padded, so that the
    // frequency kernel always takes four frequencies per pass
    int Kp;
    int exact_emis; // emission mode: CPU formula with el/gl per frequency instead of the source-function form
    double scale;
    DevBeam beam;
    DevSeed seed;
    const DevGain *gain; // [N], entry 0 unused
    DevRays rays;
    double *image;
    double *iang;
    DevCtl *ctl;
    DevProbe probe;
    unsigned int n_tiles;
    // frequency launch: tiles [tile_begin, tile_end) of 64 rays, tile counter next_tile_f[freq_id]
    // (a run is one launch over all tiles; the range exists for experiments that split it)
    unsigned int tile_begin, tile_end, freq_id;
    // march launch: rays [ray_begin, ray_end) of the list / grid, ray counter next_tile[launch_id]
    // (a run is one launch, or several when the ray list is still arriving from the host)
    unsigned int ray_begin, ray_end, launch_id, pad_launch;
    unsigned int debug; // bit0: skip the frequency kernel (profiling only, RT_HIP_DEBUG env)
    // rt_march.hip -> records -> rt_freq.hip
    const unsigned char *blob; // march blob (global copy)
    unsigned int blob_bytes;
    // 1: never mark a ray F_SKIP (rt_march.hip): a lineshape table holds a NaN or an infinity, and on the CPU even a
    // ray with all-zero sums reads row ivl = 0 of every table and fails with 0 * NaN (Helper.h:543-594)
    unsigned int no_skip;
    // watchdog of the march instance for tables / step sizes outside the verified ranges (rt_march.hip): iterations of a
    // wave since it last took rays after which the rays it holds are given up as invalid (error -1).  The reference
    // would loop for ever on a ray whose steps do not advance (an infinite dz, say) -- on a GPU that is a hung device,
    // not a hung process.  2^24 by default, four orders of magnitude above the longest ray of the shipped inputs.
    unsigned int spin_limit;
    unsigned char *rec;
    unsigned int rec_stride;
    unsigned int chunk; // rays a wave reserves per fetch of the global ray counter
    // 1: every pixel receives exactly one ray of this launch (ASE, ray grid == beam grid,
    // na*nb == 1): the frequency kernel stores image rows instead of adding to them
    unsigned int exclusive;
    // 1: backward method, the rays are grid points of the beam's own grids and every grid point lands in its own
    // deposit cell (host check, RayTraceImageCPU.cpp:11-16 on the float the ray carries): the frequency kernel
    // takes pixel and angle cell from the ray's grid indices instead of searching the grids
    unsigned int own_cells;
    // path tracer (RayTrace::calc_ray_path, src/RayTraceImage.cpp:440-477): when path != NULL
    // the march records (x, y) at every sub-segment boundary and rt_path_kernel fills I
    unsigned int path_on;
    float *path;   // [n_rays][3 (N-1) + 1][3] = {x, y, I} triples, the reference's debug layout
    int32_t *path_err; // [n_rays] return code of each ray
    // step safety factor c of Helper.h:270-313 folded into its three uses (c = 0.5 in create_image)
    float c_cap, c_h1, c_h3, gs_cap; // c*1.00001f, c*0.1f, c*0.05f; see below
    // gs_cap = 708 / max |gv|: a sub-segment whose gain sum exceeds it in magnitude could take gs * gv
    // out of the range where e^x is a normal double; it runs the CPU's own formula (overflow to inf,
    // NaN and all) instead of the fast form of the update (rt_freq.hip)
    // Rays that fail (error -2 / -3, Helper.h:582-594) are only known after their frequencies have been
    // integrated -- and, in the normal pass, deposited.  A run that reports such a failure repeats the
    // frequency pass (rt_hip_plan_fetch): safe = 1 integrates without depositing and marks the failing
    // rays in bad[], safe = 2 deposits all others.  The CPU loop skips failing rays
    // (RayTraceImageCPU.cpp:29-36); after the repeat so does this image.  0 = the normal pass.
    unsigned int safe;
    unsigned int park; // march: lanes that must wait for block [A] before it runs (rt_march.hip); 1 = every iteration
    // march, express waves (rt_march.hip): a wave that holds a ray older than express_age loop iterations raises its
    // wave priority (0 = never); with express_hold it also fetches no further rays from the launch's counters while it
    // holds such a ray, so that it thins out and its iterations get shorter
    unsigned int express_age, express_hold, express_park, express_tail;
    // the last late_chunks chunks of a march launch are handed out to the first late_waves waves of each work-group only
    // (rt_march.hip, "The end of a launch"); 0: no such zone
    unsigned int late_chunks, late_waves, late_first; // late_first: the first of those waves (the first MARCHING wave of a work-group)
    unsigned char *bad; // [n_rays], only in the repeat
};

// ---- argument block of the frequency kernel (rt_freq.hip) ------------------------------------
// The kernel argument is FreqKArg = { hot, cold }.  `hot` is what the frequency loop reads: the
// compiler loads it once and keeps it in SGPRs.  `cold` is what the per-ray preamble of a tile
// reads (beam grids, seed tables, ray grids, probe outputs): the kernel takes its address inside
// the kernarg segment and reads the fields it needs where it needs them (scalar loads), so none of
// them is live across the frequency loop.  (With one ~600-byte by-value block every field was
// hoisted to the kernel entry: 207 SGPR spills and a dependent global load per frequency batch.)
struct FreqCold {
    DevBeam beam;
    DevSeed seed;
    DevRays rays;
    DevProbe probe;
};
enum : unsigned {
    FQ_EXCLUSIVE  = 1u,   // DevParams::exclusive
    FQ_SAFE_CHECK = 2u,   // DevParams::safe == 1: integrate without depositing, mark failing rays
    FQ_SAFE_SKIP  = 4u,   // DevParams::safe == 2: deposit all rays but the marked ones
    FQ_EXACT_EMIS = 8u,   // DevParams::exact_emis
    FQ_HAS_SEED   = 16u,
    FQ_PROBE      = 32u,
    FQ_GV_NAN     = 64u,  // some lineshape value is a NaN or an infinity (found by the host scan): test per frequency
    FQ_IANG_LDS   = 128u, // the I_ang histogram of a work-group lives in LDS
    FQ_NEED_EXIT  = 256u, // the exit angles are needed (forward method, seed, or probe)
    FQ_OWN_CELLS  = 512u, // DevParams::own_cells: pixel and angle cell of a ray are its grid indices
    FQ_DBG_NOFLUSH = 1024u // profiling only (debug bit 2): the work-groups do not add their I_ang histograms to the result
};
// Waves per work-group of the frequency kernel.  One 16-wave work-group per CU (four waves per SIMD, what the
// registers allow) instead of four 4-wave ones: one I_ang histogram and one pair of exponent tables per CU in LDS,
// and a quarter of the atomics when the histograms are added to the result at the end of the launch.
#ifndef RT_FREQ_WG_WAVES
#define RT_FREQ_WG_WAVES 16
#endif
constexpr int FREQ_WG_WAVES = RT_FREQ_WG_WAVES;
struct FreqHot {
    const float *gv0, *gv1; // SF == 6 (N = 3): lineshape tables of lengths 1 and 2, rows of Kp floats
    const DevGain *gain;    // any N: [N] lineshape pointers, entry 0 unused
    const unsigned char *rec;
    double *image;
    double *iang;
    DevCtl *ctl;
    const double *dv2;      // [Kp] 2 * beam.dv (RayTraceImageCPU.cpp:66), zero padded
    const double *seed_fk;  // [Kp] seed.f[4], zero padded; NULL without a seed
    unsigned char *bad;     // [n_rays] failing-ray marks of the checking repeat, else NULL
    double scale;
    float gs_cap;
    int K, Kp, L, method;
    unsigned rec_stride, n_rays;
    unsigned tile_begin, tile_end, freq_id;
    unsigned fetch_shift; // ceil(log2(2 x waves of the grid)): tiles per counter fetch = (tiles left) >> fetch_shift, clamped
    unsigned flags; // FQ_*
    int nslot;      // rows of the per-wave LDS row cache
    int nx, ny, n_ang;
};
struct FreqKArg {
    FreqHot hot;
    FreqCold cold;
};

// packing of RecMeta::flags_steps
constexpr unsigned REC_FLAG_MASK = 0xfu, REC_NDONE_SHIFT = 4, REC_NDONE_MASK = 0xffu, REC_STEPS_SHIFT = 12;

// flag bits of the per-ray march record
enum : unsigned {
    F_ESCAPED = 1u, // left the plasma (Helper.h:465-469)
    F_ERR1    = 2u, // error -1, ray ~perpendicular to z (Helper.h:515)
    F_SKIP    = 4u, // frequency pass provably contributes exactly zero
    F_VALID   = 8u  // lane holds a ray of this tile
};

#ifdef __HIPCC__
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif
constexpr unsigned REC_SLOT_ROW = WAVE * 12u; // bytes of one slot of the 64 rays of a tile
// byte offsets of slot s / of the meta block of ray ridx from the start of the record buffer
RT_HD inline size_t rec_slot_off(unsigned ridx, int s, unsigned rec_stride)
{
    return (size_t) (ridx >> 6) * ((size_t) WAVE * rec_stride) + (size_t) s * REC_SLOT_ROW + (size_t) (ridx & 63u) * 12u;
}
RT_HD inline size_t rec_meta_off(unsigned ridx, int S, unsigned rec_stride)
{
    return (size_t) (ridx >> 6) * ((size_t) WAVE * rec_stride) + (size_t) S * REC_SLOT_ROW + (size_t) (ridx & 63u) * sizeof(RecMeta);
}
// bytes of the record buffer of n_rays rays (whole tiles)
RT_HD inline size_t rec_bytes(unsigned long long n_rays, unsigned rec_stride)
{
    return (size_t) ((n_rays + WAVE - 1) / WAVE) * ((size_t) WAVE * rec_stride);
}
// slot s of the record of ray ridx (rec = start of the record buffer), zero if the ray never entered that sub-segment
RT_HD inline RecSlot rec_slot(const unsigned char *rec, unsigned ridx, unsigned rec_stride, int s, int S, unsigned flags_steps,
                              bool backward)
{
    const int n_done   = (int) ((flags_steps >> REC_NDONE_SHIFT) & REC_NDONE_MASK);
    const bool written = backward ? s >= S - n_done : s < n_done;
    RecSlot z          = { 0.0f, 0.0f, 0 };
    return written ? *reinterpret_cast<const RecSlot *>(rec + rec_slot_off(ridx, s, rec_stride)) : z;
}

} // namespace rt
