// CreateImageHip.cpp -- stand-alone harness for the HIP back-end, built ONLY
// where the reference tree is present (oracle/Makefile `hipharness`), against
// the reference's own structs, file unpacking, scale_problem, check_ans and CPU
// loop.  It exists because the dispatcher arm of INTEGRATION.md cannot be
// applied to the read-only reference here: it drives the same back-end loop
// functions with the ray list built the way RayTrace::create_image builds it
// (src/RayTraceImage.cpp:283-328) and prints the harness table of
// src/CreateImage.cpp:166-173 with the columns SURVEY.md 8(f-3) asks for.
//
//   CreateImageHip [-methods=cpu,hip,hip-multigpu] [-iterations=N] [-scale=f] file.dat
//
// Checks per method: the reference's own one-sided norm gate against the golden
// arrays in the file (check_ans, src/CreateImageHelpers.cpp:66-100) AND a
// two-sided rel-L2 gate (1e-5) against the CPU loop of the same run.
#include "CreateImageHelpers.h"
#include "RayTrace.h"
#include "common/RayTraceImageHelper.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern void RayTraceImageCPULoop(int, const RayTrace::EUV_beam_struct &, const RayTrace::ray_gain_struct *,
    const RayTrace::ray_seed_struct *, int, const std::vector<ray_struct> &, double, double *, double *,
    unsigned int &, std::vector<ray_struct> &);
extern void RayTraceImageHipLoop(int, const RayTrace::EUV_beam_struct &, const RayTrace::ray_gain_struct *,
    const RayTrace::ray_seed_struct *, int, const std::vector<ray_struct> &, double, double *, double *,
    unsigned int &, std::vector<ray_struct> &);
extern void RayTraceImageHipMultiGPULoop(int, const RayTrace::EUV_beam_struct &, const RayTrace::ray_gain_struct *,
    const RayTrace::ray_seed_struct *, int, const std::vector<ray_struct> &, double, double *, double *,
    unsigned int &, std::vector<ray_struct> &);
extern int RayTraceImageHipDeviceCount();

typedef void (*loop_fn)(int, const RayTrace::EUV_beam_struct &, const RayTrace::ray_gain_struct *,
    const RayTrace::ray_seed_struct *, int, const std::vector<ray_struct> &, double, double *, double *,
    unsigned int &, std::vector<ray_struct> &);

static RayTrace::create_image_struct *load(const std::string &file, double scale, std::vector<double> &img0,
                                           std::vector<double> &ang0)
{
    FILE *fid = fopen(file.c_str(), "rb");
    if (!fid)
        return NULL;
    uint64_t n = 0;
    fread2(&n, sizeof(n), 1, fid);
    std::vector<char> buf(n);
    fread2(buf.data(), 1, n, fid);
    fclose(fid);
    RayTrace::create_image_struct *info = new RayTrace::create_image_struct();
    info->unpack(std::pair<const char *, size_t>(buf.data(), (size_t) n));
    const RayTrace::EUV_beam_struct *b = info->euv_beam;
    if (info->image)
        img0.assign(info->image, info->image + (size_t) b->nx * b->ny * b->nv);
    if (info->I_ang)
        ang0.assign(info->I_ang, info->I_ang + (size_t) b->na * b->nb);
    free(info->image);
    free(info->I_ang);
    info->image = NULL;
    info->I_ang = NULL;
    if (scale != 1.0)
        scale_problem(*info, scale);
    return info;
}

static double rel_l2(const std::vector<double> &a, const std::vector<double> &b)
{
    double e = 0, n = 0;
    for (size_t i = 0; i < a.size(); i++) {
        e += (a[i] - b[i]) * (a[i] - b[i]);
        n += b[i] * b[i];
    }
    return n > 0 ? std::sqrt(e / n) : std::sqrt(e);
}

int main(int argc, char **argv)
{
    Options opt;
    std::vector<std::string> files = opt.read_cmd(argc, argv);
    if (files.empty())
        return -2;
    if (opt.methods.empty()) {
        opt.methods.push_back("cpu");
        opt.methods.push_back("Hip");
        if (RayTraceImageHipDeviceCount() > 1)
            opt.methods.push_back("Hip-MultiGPU");
    }
    int n_errors = 0;
    for (size_t fi = 0; fi < files.size(); fi++) {
        std::vector<double> img0, ang0;
        RayTrace::create_image_struct *info = load(files[fi], opt.scale, img0, ang0);
        if (!info) {
            fprintf(stderr, "Error opening file: %s\n", files[fi].c_str());
            return -2;
        }
        printf("\nRunning tests for %s\n\n", files[fi].c_str());
        const RayTrace::EUV_beam_struct &eb = *info->euv_beam;
        // mode select + ray list, as create_image does (RayTraceImage.cpp:283-328)
        int method    = info->seed ? 2 : 1;
        double scale  = 1.0;
        int dims[4]   = { eb.nx, eb.ny, eb.na, eb.nb };
        const double *grid[4] = { eb.x, eb.y, eb.a, eb.b };
        if (info->seed) {
            const RayTrace::seed_beam_struct &sb = *info->seed_beam;
            dims[0] = sb.nx; dims[1] = sb.ny; dims[2] = sb.na; dims[3] = sb.nb;
            grid[0] = sb.x; grid[1] = sb.y; grid[2] = sb.a; grid[3] = sb.b;
            scale   = (sb.dx * sb.dy * sb.da * sb.db) / (eb.dx * eb.dy);
        }
        const long total = (long) dims[0] * dims[1] * dims[2] * dims[3];
        std::vector<ray_struct> rays;
        rays.reserve((size_t) (total / info->N_parallel + 1));
        for (long id = info->N_start; id < total; id += info->N_parallel) {
            ray_struct r;
            r.b = (float) grid[3][id % dims[3]];
            r.a = (float) grid[2][(id / dims[3]) % dims[2]];
            r.y = (float) grid[1][(id / ((long) dims[2] * dims[3])) % dims[1]];
            r.x = (float) grid[0][id / ((long) dims[1] * dims[2] * dims[3])];
            rays.push_back(r);
        }
        const size_t n_img = (size_t) eb.nx * eb.ny * eb.nv, n_ang = (size_t) eb.na * eb.nb;
        std::vector<double> cpu_img, cpu_ang;
        std::vector<std::vector<double>> times(opt.methods.size());
        for (size_t m = 0; m < opt.methods.size(); m++) {
            std::string name = opt.methods[m];
            std::transform(name.begin(), name.end(), name.begin(), ::tolower);
            loop_fn fn = NULL;
            if (name == "cpu")
                fn = RayTraceImageCPULoop;
            else if (name == "hip")
                fn = RayTraceImageHipLoop;
            else if (name == "hip-multigpu")
                fn = RayTraceImageHipMultiGPULoop;
            else {
                fprintf(stderr, "Unknown method: %s\n", name.c_str());
                n_errors++;
                continue;
            }
            printf("Running %s\n", opt.methods[m].c_str());
            std::vector<double> img, ang;
            for (int it = 0; it < opt.iterations; it++) {
                img.assign(n_img, 0.0);
                ang.assign(n_ang, 0.0);
                unsigned int code = 0;
                std::vector<ray_struct> failed;
                double t0 = getTime();
                fn(info->N, eb, info->gain, info->seed, method, rays, scale, img.data(), ang.data(), code, failed);
                times[m].push_back(getTime() - t0);
                if (code != 0) {
                    fprintf(stderr, "  Some rays failed (code %u)\n", code);
                    n_errors++;
                }
            }
            if (name == "cpu") {
                cpu_img = img;
                cpu_ang = ang;
            }
            if (opt.scale == 1.0 && !img0.empty()) {
                info->image = img.data();
                info->I_ang = ang.data();
                if (!check_ans(img0.data(), ang0.data(), *info))
                    n_errors++;
                info->image = NULL;
                info->I_ang = NULL;
            }
            if (!cpu_img.empty() && name != "cpu") {
                double ei = rel_l2(img, cpu_img), ea = rel_l2(ang, cpu_ang);
                printf("   two-sided rel-L2 vs cpu: image %.3e  I_ang %.3e\n", ei, ea);
                if (!(ei <= 1e-5) || !(ea <= 1e-5)) {
                    printf("   Answers do not match the CPU loop (tol 1e-5)\n");
                    n_errors++;
                }
            }
        }
        printf("\n        METHOD    Avg     Min     Max   Std Dev    rays/s\n");
        for (size_t m = 0; m < opt.methods.size(); m++) {
            if (times[m].empty())
                continue;
            printf("%14s %7.3f %7.3f %7.3f %7.3f  %9.3e\n", opt.methods[m].c_str(), getAvg(times[m]),
                   getMin(times[m]), getMax(times[m]), getDev(times[m]), (double) rays.size() / getMin(times[m]));
        }
        delete info->euv_beam;
        delete info->seed_beam;
        delete[] info->gain;
        delete info->seed;
        delete info;
    }
    printf(n_errors == 0 ? "\nAll tests passed\n" : "\nSome tests failed\n");
    return n_errors;
}
