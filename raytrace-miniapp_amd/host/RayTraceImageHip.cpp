// RayTraceImageHip.cpp -- the C++ adapter a reference maintainer adds next to
// RayTraceImageCPU.cpp / RayTraceImageCuda.cu.  It implements the back-end loop
// signature that RayTrace::create_image dispatches to (declared `extern` at
// src/RayTraceImage.cpp:47-75) by flattening the reference structs into the
// POD records of include/rt_hip.h and calling the C ABI of librt_hip.so.
//
// Compiled against the reference's own headers (-I<reference>/src
// -I<reference>/src/include) -- nothing of the reference is restated here.
// See INTEGRATION.md for the dispatcher / harness / CMake lines that go with it.
//
//   RayTraceImageHipLoop          one device ("hip" arm)
//   RayTraceImageHipMultiGPULoop  all devices of the node ("hip-multigpu" arm) through
//        rt_hip_multi_image_loop: stands where the reference runs RayTraceImageThreadLoop
//        (src/RayTraceImage.cpp:89-134) with setGPU called in the spawning thread (:116).
#include "RayTrace.h"
#include "common/RayTraceImageHelper.h"
#include "utilities/RayUtilityMacros.h"

#include "rt_hip.h"

#include <cstring>
#include <string>
#include <vector>

static_assert(sizeof(ray_struct) == sizeof(rt_ray), "ray_struct and rt_ray must share a layout");

namespace {

struct Flat {
    rt_beam beam;
    std::vector<rt_gain> gain;
    rt_seed seed;
    bool has_seed;
};

Flat flatten(int N, const RayTrace::EUV_beam_struct &b, const RayTrace::ray_gain_struct *g,
             const RayTrace::ray_seed_struct *s)
{
    Flat f;
    f.beam.nx = b.nx; f.beam.ny = b.ny; f.beam.na = b.na; f.beam.nb = b.nb; f.beam.nv = b.nv;
    f.beam.dx = b.dx; f.beam.dy = b.dy; f.beam.da = b.da; f.beam.db = b.db; f.beam.dz = b.dz;
    f.beam.x = b.x; f.beam.y = b.y; f.beam.a = b.a; f.beam.b = b.b; f.beam.dv = b.dv;
    f.gain.resize((size_t) N);
    for (int i = 0; i < N; i++) {
        rt_gain &o = f.gain[(size_t) i];
        o.Nx = g[i].Nx; o.Ny = g[i].Ny; o.Nv = g[i].Nv;
        o.x = g[i].x; o.y = g[i].y; o.n = g[i].n;
        o.g0 = g[i].g0; o.E0 = g[i].E0; o.gv = g[i].gv;
    }
    f.has_seed = s != NULL;
    memset(&f.seed, 0, sizeof(f.seed));
    if (s) {
        for (int i = 0; i < 5; i++) {
            f.seed.dim[i] = s->dim[i];
            f.seed.x[i]   = s->x[i];
            f.seed.f[i]   = s->f[i];
        }
        f.seed.f0 = s->f0;
    }
    return f;
}

// counters and device times of the last loop call of this thread (the loop signature has no slot for
// them): read by harnesses that print ray-steps/s next to the reference's timing table
thread_local rt_stats g_last_stats = {};

void run_on_device(int device, int N, const Flat &f, int method, const ray_struct *rays, size_t n_rays,
                   double scale, double *image, double *I_ang, unsigned int &failure_code,
                   std::vector<ray_struct> &failed_rays, std::string &error)
{
    rt_ray failed[RT_N_FAILED_MAX];
    int n_failed      = 0;
    unsigned int code = 0;
    int rc = rt_hip_image_loop(device, N, &f.beam, f.gain.data(), f.has_seed ? &f.seed : NULL, method,
                               reinterpret_cast<const rt_ray *>(rays), n_rays, scale, image, I_ang, &code,
                               failed, RT_N_FAILED_MAX, &n_failed, &g_last_stats);
    if (rc != RT_OK) {
        error = std::string("HIP backend error: ") + rt_hip_last_error();
        return;
    }
    failure_code |= code;
    for (int i = 0; i < n_failed; i++) {
        ray_struct r;
        memcpy(&r, &failed[i], sizeof(r));
        failed_rays.push_back(r);
    }
}

} // namespace

int RayTraceImageHipDeviceCount() { return rt_hip_device_count(); }

// rt_stats of the last RayTraceImageHip*Loop call made by the calling thread
const rt_stats *RayTraceImageHipLastStats() { return &g_last_stats; }

void RayTraceImageHipLoop(int N, const RayTrace::EUV_beam_struct &beam, const RayTrace::ray_gain_struct *gain,
    const RayTrace::ray_seed_struct *seed, int method, const std::vector<ray_struct> &rays, double scale,
    double *image, double *I_ang, unsigned int &failure_code, std::vector<ray_struct> &failed_rays)
{
    failure_code = 0;
    Flat f       = flatten(N, beam, gain, seed);
    std::string error;
    run_on_device(0, N, f, method, rays.empty() ? NULL : &rays[0], rays.size(), scale, image, I_ang,
                  failure_code, failed_rays, error);
    if (!error.empty())
        RAY_ERROR(error); // device/runtime errors end the process, as CUDA_CHECK does (RayTraceImageCuda.cu:8-18)
}

void RayTraceImageHipMultiGPULoop(int N, const RayTrace::EUV_beam_struct &beam,
    const RayTrace::ray_gain_struct *gain, const RayTrace::ray_seed_struct *seed, int method,
    const std::vector<ray_struct> &rays, double scale, double *image, double *I_ang,
    unsigned int &failure_code, std::vector<ray_struct> &failed_rays)
{
    failure_code = 0;
    if (rt_hip_device_count() < 1)
        RAY_ERROR("Hip-MultiGPU is not availible");
    Flat f = flatten(N, beam, gain, seed);
    rt_ray failed[RT_N_FAILED_MAX];
    int n_failed      = 0;
    unsigned int code = 0;
    // all devices of the node behind one call: pixel-column tiles + one RCCL gather (ASE), ray chunks +
    // one RCCL sum-reduce otherwise; device binding, communicator and assembly live behind the C ABI
    int rc = rt_hip_multi_image_loop(0, N, &f.beam, f.gain.data(), f.has_seed ? &f.seed : NULL, method,
                                     rays.empty() ? NULL : reinterpret_cast<const rt_ray *>(&rays[0]), rays.size(), scale,
                                     image, I_ang, &code, failed, RT_N_FAILED_MAX, &n_failed, &g_last_stats);
    if (rc != RT_OK)
        RAY_ERROR(std::string("HIP backend error: ") + rt_hip_last_error());
    failure_code |= code;
    for (int i = 0; i < n_failed; i++) {
        ray_struct r;
        memcpy(&r, &failed[i], sizeof(r));
        failed_rays.push_back(r);
    }
}
