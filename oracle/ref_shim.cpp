/*
 * ref_shim.cpp -- C entry points around the UNMODIFIED reference CPU path.
 *
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with the
 * reference's own source files where they lie under /root/reference/src into
 * oracle/_ref/librt_ref.so (git-ignored; never copied into the repo).  It is
 * used (a) to pin oracle/rt_oracle.c against the real reference,
 * (b) to generate the fixtures under tests/golden/ (tests/golden/make_golden.py)
 * and (c) as bench.py's cpu_baseline of kind "reference".
 *
 * Nothing of the reference is restated here: the shim only builds the
 * reference's structs around caller-owned arrays and calls
 *   RayTraceImageCPULoop      (src/RayTraceImageCPU.cpp:19)
 *   RayTrace::create_image    (src/RayTraceImage.cpp:227)
 *   create_image_struct::unpack (src/RayTraceStructures.cpp:2224)
 *   scale_problem             (src/CreateImageHelpers.cpp:144)
 */
#include "RayTrace.h"
#include "common/RayTraceImageHelper.h"

#include "../include/rt_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern void RayTraceImageCPULoop(int N, const RayTrace::EUV_beam_struct &euv_beam,
    const RayTrace::ray_gain_struct *gain, const RayTrace::ray_seed_struct *seed, int method,
    const std::vector<ray_struct> &rays, double scale, double *image, double *I_ang,
    unsigned int &failure_code, std::vector<ray_struct> &failed_rays);

void scale_problem(RayTrace::create_image_struct &info, double scale); // src/CreateImageHelpers.h:18

namespace {

// Point a reference EUV_beam_struct at borrowed arrays (detached again before
// destruction so that its destructor frees nothing of ours).
void attach(RayTrace::EUV_beam_struct &b, const rt_beam *p)
{
    b.nx = p->nx; b.ny = p->ny; b.na = p->na; b.nb = p->nb; b.nv = p->nv; b.nz = 1;
    b.dx = p->dx; b.dy = p->dy; b.da = p->da; b.db = p->db; b.dz = p->dz;
    b.x  = const_cast<double *>(p->x);
    b.y  = const_cast<double *>(p->y);
    b.a  = const_cast<double *>(p->a);
    b.b  = const_cast<double *>(p->b);
    b.dv = const_cast<double *>(p->dv);
}
void detach(RayTrace::EUV_beam_struct &b)
{
    b.x = b.y = b.a = b.b = b.z = b.v = b.dv = NULL;
}
void attach(RayTrace::ray_gain_struct &g, const rt_gain *p)
{
    g.Nx = p->Nx; g.Ny = p->Ny; g.Nv = p->Nv;
    g.x  = const_cast<double *>(p->x);
    g.y  = const_cast<double *>(p->y);
    g.n  = const_cast<double *>(p->n);
    g.g0 = const_cast<float *>(p->g0);
    g.E0 = const_cast<float *>(p->E0);
    g.gv = const_cast<float *>(p->gv);
    g.gv0 = NULL;
}
void detach(RayTrace::ray_gain_struct &g)
{
    g.x = g.y = g.n = NULL;
    g.g0 = g.E0 = g.gv = g.gv0 = NULL;
}
void attach(RayTrace::ray_seed_struct &s, const rt_seed *p)
{
    for (int i = 0; i < 5; i++) {
        s.dim[i] = p->dim[i];
        s.x[i]   = const_cast<double *>(p->x[i]);
        s.f[i]   = const_cast<double *>(p->f[i]);
    }
    s.f0 = p->f0;
}
void detach(RayTrace::ray_seed_struct &s)
{
    for (int i = 0; i < 5; i++)
        s.x[i] = s.f[i] = NULL;
}

RayTrace::create_image_struct *load_file(const char *path)
{
    FILE *fid = fopen(path, "rb");
    if (!fid)
        return NULL;
    uint64_t n = 0;
    if (fread(&n, sizeof(n), 1, fid) != 1) {
        fclose(fid);
        return NULL;
    }
    std::vector<char> buf(n);
    size_t got = fread(buf.data(), 1, n, fid);
    fclose(fid);
    if (got != n)
        return NULL;
    RayTrace::create_image_struct *info = new RayTrace::create_image_struct();
    info->unpack(std::pair<const char *, size_t>(buf.data(), (size_t) n));
    return info;
}
void free_info(RayTrace::create_image_struct *info)
{
    if (!info)
        return;
    delete info->euv_beam;
    delete info->seed_beam;
    delete[] info->gain;
    delete info->seed;
    delete info; // frees image / I_ang
}

} // namespace

extern "C" {

/* The reference's serial back-end loop on caller-owned records.
 * image / I_ang are accumulated into: pass them zeroed. */
int ref_cpu_loop(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed, int method,
                 const rt_ray *rays, size_t n_rays, double scale, double *image, double *I_ang,
                 unsigned int *failure_code, int *n_failed)
{
    RayTrace::EUV_beam_struct eb;
    attach(eb, beam);
    RayTrace::ray_gain_struct *g = new RayTrace::ray_gain_struct[N];
    for (int i = 0; i < N; i++)
        attach(g[i], &gain[i]);
    RayTrace::ray_seed_struct sd;
    if (seed)
        attach(sd, seed);
    std::vector<ray_struct> rv(n_rays);
    static_assert(sizeof(ray_struct) == sizeof(rt_ray), "ray layout");
    if (n_rays)
        memcpy(rv.data(), rays, n_rays * sizeof(rt_ray));
    std::vector<ray_struct> failed;
    unsigned int code = 0;
    RayTraceImageCPULoop(N, eb, g, seed ? &sd : NULL, method, rv, scale, image, I_ang, code, failed);
    if (failure_code)
        *failure_code = code;
    if (n_failed)
        *n_failed = (int) failed.size();
    detach(eb);
    for (int i = 0; i < N; i++)
        detach(g[i]);
    delete[] g;
    detach(sd);
    return 0;
}

/* dims[0..11] = N, N_start, N_parallel, nx, ny, na, nb, nv, has_seed,
 * seed_nx*seed_ny (lo 31 bits irrelevant), gain Nx, gain Ny. */
int ref_file_dims(const char *path, int *dims)
{
    RayTrace::create_image_struct *info = load_file(path);
    if (!info)
        return -1;
    dims[0] = info->N;
    dims[1] = info->N_start;
    dims[2] = info->N_parallel;
    dims[3] = info->euv_beam->nx;
    dims[4] = info->euv_beam->ny;
    dims[5] = info->euv_beam->na;
    dims[6] = info->euv_beam->nb;
    dims[7] = info->euv_beam->nv;
    dims[8] = info->seed != NULL;
    dims[9] = info->seed_beam ? info->seed_beam->nx : 0;
    dims[10] = info->gain[0].Nx;
    dims[11] = info->gain[0].Ny;
    free_info(info);
    return 0;
}

/* The whole reference path on a .dat file: unpack + RayTrace::create_image
 * with the given method string ("cpu", "threads").  image_out [nx*ny*nv] and
 * iang_out [na*nb] receive the computed arrays; golden_image / golden_iang
 * (may be NULL) the arrays embedded in the file.  seconds = wall time of
 * create_image alone. */
int ref_create_image_file(const char *path, const char *method, double *image_out,
                          double *iang_out, double *golden_image, double *golden_iang,
                          double *seconds)
{
    RayTrace::create_image_struct *info = load_file(path);
    if (!info)
        return -1;
    const size_t n_img = (size_t) info->euv_beam->nx * info->euv_beam->ny * info->euv_beam->nv;
    const size_t n_ang = (size_t) info->euv_beam->na * info->euv_beam->nb;
    if (golden_image && info->image)
        memcpy(golden_image, info->image, n_img * sizeof(double));
    if (golden_iang && info->I_ang)
        memcpy(golden_iang, info->I_ang, n_ang * sizeof(double));
    free(info->image);
    info->image = NULL;
    free(info->I_ang);
    info->I_ang = NULL;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    RayTrace::create_image(info, std::string(method));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds)
        *seconds = (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
    memcpy(image_out, info->image, n_img * sizeof(double));
    memcpy(iang_out, info->I_ang, n_ang * sizeof(double));
    free_info(info);
    return 0;
}

/* RayTrace::calc_ray on the first n rays the reference itself would build for
 * this file (src/RayTraceImage.cpp:300-328), for per-ray checks: Iv_out
 * [n][nv], ray2_out [n][4] (as double), err_out [n]; stride picks every
 * stride-th ray of the list. */
int ref_calc_rays_file(const char *path, size_t stride, size_t n, double *Iv_out,
                       double *ray2_out, int *err_out, double *ray_in_out)
{
    RayTrace::create_image_struct *info = load_file(path);
    if (!info)
        return -1;
    const RayTrace::EUV_beam_struct *eb = info->euv_beam;
    int N2[4]       = { eb->nx, eb->ny, eb->na, eb->nb };
    const double *g[4] = { eb->x, eb->y, eb->a, eb->b };
    int method = 1;
    if (info->seed != NULL) {
        method = 2;
        N2[0] = info->seed_beam->nx; N2[1] = info->seed_beam->ny;
        N2[2] = info->seed_beam->na; N2[3] = info->seed_beam->nb;
        g[0] = info->seed_beam->x; g[1] = info->seed_beam->y;
        g[2] = info->seed_beam->a; g[3] = info->seed_beam->b;
    }
    const long Nt = (long) N2[0] * N2[1] * N2[2] * N2[3];
    for (size_t r = 0; r < n; r++) {
        long ijkm = (long) (r * stride);
        if (ijkm >= Nt)
            ijkm = Nt - 1;
        int m = (int) (ijkm % N2[3]);
        int k = (int) ((ijkm / N2[3]) % N2[2]);
        int j = (int) ((ijkm / ((long) N2[2] * N2[3])) % N2[1]);
        int i = (int) (ijkm / ((long) N2[1] * N2[2] * N2[3]));
        double ray[4] = { g[0][i], g[1][j], g[2][k], g[3][m] }, out[4] = { 0, 0, 0, 0 };
        err_out[r] = RayTrace::calc_ray(ray, info->N, eb->dz, info->gain, info->seed, eb->nv,
                                        method, Iv_out + r * (size_t) eb->nv, out);
        for (int q = 0; q < 4; q++) {
            ray2_out[4 * r + q] = out[q];
            if (ray_in_out)
                ray_in_out[4 * r + q] = ray[q];
        }
    }
    free_info(info);
    return 0;
}

/* The reference's own enlargement rule, scale_problem (src/CreateImageHelpers.cpp:104-150), applied to a loaded file:
 * the grids it leaves in euv_beam (which = 0) or seed_beam (which = 1).  dims[0..3] = nx, ny, na, nb; d[0..3] = dx, dy,
 * da, db; grids (may be NULL: sizes only) receives x | y | a | b back to back.  Returns 1 if the file has no such beam. */
int ref_scale_file(const char *path, double scale, int which, int *dims, double *d, double *grids)
{
    RayTrace::create_image_struct *info = load_file(path);
    if (!info)
        return -1;
    scale_problem(*info, scale);
    int rc = 0;
    if (which == 0) {
        const RayTrace::EUV_beam_struct *b = info->euv_beam;
        dims[0] = b->nx; dims[1] = b->ny; dims[2] = b->na; dims[3] = b->nb;
        d[0] = b->dx; d[1] = b->dy; d[2] = b->da; d[3] = b->db;
        if (grids) {
            double *q = grids;
            memcpy(q, b->x, sizeof(double) * b->nx); q += b->nx;
            memcpy(q, b->y, sizeof(double) * b->ny); q += b->ny;
            memcpy(q, b->a, sizeof(double) * b->na); q += b->na;
            memcpy(q, b->b, sizeof(double) * b->nb);
        }
    } else if (info->seed_beam) {
        const RayTrace::seed_beam_struct *b = info->seed_beam;
        dims[0] = b->nx; dims[1] = b->ny; dims[2] = b->na; dims[3] = b->nb;
        d[0] = b->dx; d[1] = b->dy; d[2] = b->da; d[3] = b->db;
        if (grids) {
            double *q = grids;
            memcpy(q, b->x, sizeof(double) * b->nx); q += b->nx;
            memcpy(q, b->y, sizeof(double) * b->ny); q += b->ny;
            memcpy(q, b->a, sizeof(double) * b->na); q += b->na;
            memcpy(q, b->b, sizeof(double) * b->nb);
        }
    } else {
        rc = 1;
    }
    free_info(info);
    return rc;
}

/* RayTrace::calc_ray_path (src/RayTraceImage.cpp:440-477) on a sub-grid of the file's own
 * ray grid: n[4] points starting at i0[4] along x, y, a, b.  xr/yr/Ir: the reference's
 * layout, N2 * (i + j*nx + k*nx*ny + m*nx*ny*na) + step.  Returns the error count. */
int ref_calc_ray_path_file(const char *path, const int *i0, const int *n, double c, float *xr, float *yr,
                           float *Ir)
{
    RayTrace::create_image_struct *info = load_file(path);
    if (!info)
        return -1;
    const RayTrace::EUV_beam_struct *eb = info->euv_beam;
    const double *g[4] = { eb->x, eb->y, eb->a, eb->b };
    int method         = 1;
    if (info->seed != NULL) {
        method = 2;
        g[0] = info->seed_beam->x; g[1] = info->seed_beam->y;
        g[2] = info->seed_beam->a; g[3] = info->seed_beam->b;
    }
    std::vector<float> vx, vy, vi;
    int nerr = RayTrace::calc_ray_path(n[0], n[1], n[2], n[3], g[0] + i0[0], g[1] + i0[1], g[2] + i0[2],
                                       g[3] + i0[3], info->N, eb->dz, info->gain, info->seed, eb->nv, eb->dv,
                                       method, c, vx, vy, vi);
    memcpy(xr, vx.data(), vx.size() * sizeof(float));
    memcpy(yr, vy.data(), vy.size() * sizeof(float));
    memcpy(Ir, vi.data(), vi.size() * sizeof(float));
    free_info(info);
    return nerr;
}

} // extern "C"
