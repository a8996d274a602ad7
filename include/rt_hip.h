/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) ray-trace imaging backend.
 *
 * This is the drop-in boundary for ONE path of the XRayTrace miniapp: the
 * back-end loop that RayTrace::create_image dispatches to
 *
 *     void RayTraceImage<Backend>Loop( int N, const EUV_beam_struct& beam,
 *         const ray_gain_struct* gain, const ray_seed_struct* seed, int method,
 *         const std::vector<ray_struct>& rays, double scale, double* image,
 *         double* I_ang, unsigned int& failure_code,
 *         std::vector<ray_struct>& failed_rays );
 *                                   (reference: src/RayTraceImage.cpp:47-75)
 *
 * Everything here is plain C: PODs, pointers and sizes.  No C++ types, no
 * exceptions, no torch types.  The C++ adapter that a reference maintainer
 * links (raytrace-miniapp_amd/host/RayTraceImageHip.cpp) flattens the
 * reference structs into these records; tests and bench.py bind the same
 * symbols through ctypes.  oracle/rt_oracle.c (test infrastructure only)
 * consumes the same records so that parity tests feed both sides one input.
 *
 * All host arrays are borrowed for the duration of a call and never cached
 * across calls (reference Readme.txt:43).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_N_SUB 3         /* sub-segments per length (RayTraceImageHelper.h:31) */
#define RT_N_FAILED_MAX 32 /* failed rays reported back (RayTraceImageHelper.h:32) */

/* Status codes returned by every entry point. */
enum {
    RT_OK            = 0,
    RT_ERR_ARG       = 1, /* inconsistent sizes / NULL where data is required   */
    RT_ERR_NO_DEVICE = 2, /* no usable gfx950 device                           */
    RT_ERR_HIP       = 3, /* a HIP runtime call failed (see rt_hip_last_error) */
    RT_ERR_NOMEM     = 4
};

/* One ray: position (cm) and angle (mrad).  Replaces ray_struct
 * (src/common/RayTraceImageHelper.h:36-41); same layout, 16 bytes. */
typedef struct rt_ray {
    float x, y, a, b;
} rt_ray;

/* The fields of EUV_beam_struct (src/RayTraceStructures.h:26-52) that the
 * path reads: the output (deposit) grid and the frequency weights. */
typedef struct rt_beam {
    int32_t nx, ny, na, nb, nv;
    double dx, dy, da, db, dz;
    const double *x;  /* [nx] */
    const double *y;  /* [ny] */
    const double *a;  /* [na] */
    const double *b;  /* [nb] */
    const double *dv; /* [nv] */
} rt_beam;

/* Plasma tables of one length.  Replaces ray_gain_struct
 * (src/RayTraceStructures.h:218-228).  n,g0,E0: [ix + iy*Nx];
 * gv: [k + (ix + iy*Nx)*Nv].  gv0 is never read on the path. */
typedef struct rt_gain {
    int32_t Nx, Ny, Nv;
    const double *x; /* [Nx] */
    const double *y; /* [Ny] */
    const double *n; /* [Nx*Ny] index of refraction */
    const float *g0; /* [Nx*Ny] line-centre gain */
    const float *E0; /* [Nx*Ny] line-centre emissivity, may be NULL */
    const float *gv; /* [Nx*Ny*Nv] normalised lineshape */
} rt_gain;

/* Separable seed profile.  Replaces ray_seed_struct
 * (src/RayTraceStructures.h:276-281). */
typedef struct rt_seed {
    int32_t dim[5];
    const double *x[5];
    const double *f[5];
    double f0;
} rt_seed;

/* Counters measured by the run (not assumed): SURVEY.md 8(d). */
typedef struct rt_stats {
    uint64_t n_rays;      /* rays traced                                        */
    uint64_t cell_steps;  /* iterations of the cell loop, Helper.h:463-504      */
    uint64_t n_escaped;   /* rays that left the plasma                          */
    uint64_t n_skipped;   /* rays whose frequency pass was provably all-zero    */
    float kernel_ms;      /* device time of the trace kernel(s), HIP events     */
    float total_ms;       /* H2D + kernels + D2H as seen by the host-pointer API*/
    float march_ms;       /* device time of the march kernel (rt_march_kernel)  */
    float freq_ms;        /* device time of the frequency kernel (rt_freq_kernel)*/
} rt_stats;

/* Number of usable devices; replaces cudaGetDeviceCount in the multi-GPU arm
 * (src/RayTraceImage.cpp:398-399).  Returns 0 when there is none. */
int rt_hip_device_count(void);

/* Text of the last HIP failure on this thread ("" if none). */
const char *rt_hip_last_error(void);

/* Known-answer test of the exact-arithmetic shortcuts of the march on the device itself
 * (no reference counterpart: the CPU code divides and takes square roots with the IEEE
 * operators, src/common/RayTraceImageHelper.h:73-89).  Evaluates the shortcut and the IEEE
 * sequence for every float of the shortcut's range (and a strided sample of all others) and
 * counts the values where they differ in any bit.  n_mismatch must come back 0. */
int rt_hip_selftest(int device, unsigned long long *n_checked, unsigned long long *n_mismatch);

/*
 * Host-pointer entry point: what RayTraceImageHipLoop calls.
 * Replaces RayTraceImageCudaLoop (src/RayTraceImageCuda.cu:145-221):
 * one packed upload, the trace kernel, one download.
 *   image  [nx*ny*nv], I_ang [na*nb]: overwritten with this call's result
 *          (create_image hands them over zeroed, RayTraceImage.cpp:271-274).
 *   failure_code: bit (-error) set for error -1/-2/-3 (Helper.h:47-56,
 *          RayTraceImageCPU.cpp:32-36); failed_rays receives at most
 *          max_failed rays, *n_failed the number stored.
 *   stats may be NULL.
 * A ray list that is the full tensor grid of four 1-D grids in create_image's order
 * (src/RayTraceImage.cpp:300-328) is recognised: the rays are then generated on the device
 * (rt_hip_plan_set_ray_grid) while host threads verify the list ray by ray, bit for bit; should the
 * verification fail the result is discarded and the list itself is uploaded and traced.  Lists of
 * 2^32 - 4096 rays or more are rejected (RT_ERR_ARG): the kernels index rays with 32 bits.
 * If failure_code comes back non-zero, image and I_ang hold exactly what RayTraceImageCPULoop leaves:
 * the failing rays deposit nothing (RayTraceImageCPU.cpp:29-36) -- the frequency pass is repeated in a
 * checking mode for such a run.
 * Two kinds of ray on which the reference's loops never end are reported as invalid rays (error -1) instead of
 * being marched: a ray that starts inside the plasma with a NaN position or direction, and -- for tables or a dz
 * outside the ranges rt_hip_plan_create verifies -- the rays of a wave that has marched 2^24 iterations without
 * taking a new ray (steps that do not advance: an infinite dz in a medium without refraction).
 */
int rt_hip_image_loop(int device, int N, const rt_beam *beam, const rt_gain *gain,
                      const rt_seed *seed, int method, const rt_ray *rays, size_t n_rays,
                      double scale, double *image, double *I_ang, unsigned int *failure_code,
                      rt_ray *failed_rays, int max_failed, int *n_failed, rt_stats *stats);

/*
 * All devices of the node in one call: what RayTraceImageHipMultiGPULoop calls.  Replaces the
 * "cuda-multigpu" arm of the dispatcher (src/RayTraceImage.cpp:389-405), which runs
 * RayTraceImageThreadLoop (:89-134: contiguous ray chunks, one host thread and one private full image
 * per device, images added on the host at join) and, for the assembly, stands where a multi-rank run
 * of the application uses MPI (src/MPI_helpers.h:29-38, intensity_step_struct::sum_reduce).
 *   One process, one host thread per device, the device bound INSIDE the worker (the reference binds it
 *   in the spawning thread, RayTraceImage.cpp:116), one RCCL communicator over the ndev devices
 *   (ncclCommInitAll, kept across calls like the queues; librccl is loaded on first use).
 *   ASE (method 1, no seed) with `rays` = the full tensor grid of the beam (recognised from the list
 *   itself and verified ray by ray): pixel-column tiles -- device d traces image columns d, d+ndev, ...
 *   from a ray grid generated on the device and holds a compact tile [ny][nx_d][nv]; the tiles and the
 *   I_ang partial sums travel to device 0 in ONE grouped ncclSend/ncclRecv gather (every peer over its
 *   own xGMI link), one kernel interleaves the columns and adds the I_ang parts, one download.
 *   Anything else (seeded mode, arbitrary ray lists): contiguous ray chunks, a full image per device,
 *   ncclReduce(sum, f64) to device 0, one download.
 * ndev <= 0: all devices.  ndev = 1 is a degenerate communicator (self send/recv) and gives the image of
 * rt_hip_image_loop.  Same outputs and error convention as rt_hip_image_loop; stats: counters summed,
 * times = maximum over devices, total_ms = wall time of the call.
 */
int rt_hip_multi_image_loop(int ndev, int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                            int method, const rt_ray *rays, size_t n_rays, double scale, double *image,
                            double *I_ang, unsigned int *failure_code, rt_ray *failed_rays, int max_failed,
                            int *n_failed, rt_stats *stats);

/* Host-only helper (no device needed): 1 if the list is exactly the tensor grid of four 1-D grids in
 * create_image's order (b fastest, then a, y, x; src/RayTraceImage.cpp:300-328) -- every ray compared
 * bit for bit -- with dims = {nx, ny, na, nb}; 0 otherwise.  This is the test the two entry points
 * above apply before they generate the rays on the device. */
int rt_hip_ray_list_grid_dims(const rt_ray *rays, size_t n_rays, int dims[4]);

/* How the last rt_hip_multi_image_loop of this thread was partitioned: 1 = pixel-column tiles + gather,
 * 2 = ray chunks + sum-reduce, 0 = none yet.  (Diagnostics and tests.) */
int rt_hip_multi_last_mode(void);

/* List-mode launch tangents (Helper.h:409-410) are computed on the device by a restatement of glibc
 * 2.35's float tanf; once per process that restatement is compared with the host's tanf on 8192 angles.
 * 1: they agree, the device computes the tangents; 2: they do not (another libm; or RT_HIP_TAN_ON_HOST
 * set): the host's tanf computes them on host threads, so that every ray starts as RayTraceImageCPULoop on
 * this host starts it.  Ray grids always use the host's tanf (na + nb values). */
int rt_hip_host_libm_mode(int device);

/* Device allocations -- never data -- are kept across calls (ray lists, tangents, march records;
 * Readme.txt:43 forbids caching data only).  This returns every parked block of every device to the
 * driver; RT_HIP_POOL_MAX_MB in the environment caps what may be parked (default 32768).  The host-pointer entry
 * points also park up to eight page-locked staging buffers (tables on their way up, small outputs on their way
 * down, at most 256 MB each); this call frees those too. */
void rt_hip_pool_trim(void);

/*
 * Device-resident plan: the same path with inputs already in HBM, so that a
 * caller (bench.py, the multi-GPU driver) can time and re-run the kernel and
 * hand the output buffers to RCCL without a host round trip.
 */
typedef struct rt_hip_plan rt_hip_plan;

/* Upload beam grids, gain tables and seed tables in one arena. */
int rt_hip_plan_create(rt_hip_plan **plan, int device, int N, const rt_beam *beam,
                       const rt_gain *gain, const rt_seed *seed, int method, double scale);

/* Explicit ray list (the Loop signature's `rays`). */
int rt_hip_plan_set_rays(rt_hip_plan *plan, const rt_ray *rays, size_t n_rays);

/* Ray list generated on the device from the four 1-D grids exactly as
 * RayTrace::create_image builds it (src/RayTraceImage.cpp:300-328):
 * ray t has ijkm = first + t*stride; m = ijkm % nb fastest, then a, y, x;
 * coordinates are the grids rounded to float.  count rays are generated. */
int rt_hip_plan_set_ray_grid(rt_hip_plan *plan, const double *gx, int ngx, const double *gy,
                             int ngy, const double *ga, int nga, const double *gb, int ngb,
                             int64_t first, int64_t stride, int64_t count);

/* Zero the outputs and run the trace on `stream` (a hipStream_t, may be
 * NULL = the default stream).  image_dev / iang_dev are device pointers owned
 * by the caller (e.g. torch tensors) or NULL to use the plan's own buffers.
 * Asynchronous with respect to the host. */
int rt_hip_plan_run(rt_hip_plan *plan, void *stream, double *image_dev, double *iang_dev);

/* Wait for the last run; copy results of the plan's own buffers to the host
 * (either pointer may be NULL), and report failures and counters. */
int rt_hip_plan_fetch(rt_hip_plan *plan, double *image, double *I_ang,
                      unsigned int *failure_code, rt_ray *failed_rays, int max_failed,
                      int *n_failed, rt_stats *stats);

/* Device time in ms of the trace kernel of the last run (HIP events recorded on
 * the run's stream around the launch).  Waits for that run to finish. */
int rt_hip_plan_kernel_ms(rt_hip_plan *plan, float *ms);

/* The same, split by kernel: the march kernel (rt_march_kernel) and the frequency /
 * deposit kernel (rt_freq_kernel) of the last run. */
int rt_hip_plan_kernel_times(rt_hip_plan *plan, float *march_ms, float *freq_ms);

/* 1 if the plan's last run took the whole path in ONE launch (march and frequency pass as two phases of the same
 * persistent waves, raytrace-miniapp_amd/csrc/rt_fused.hip -- the shape of the reference's own GPU kernel,
 * src/RayTraceImageCuda.cu:66-127, in behaviour only): march_ms is then the time of that launch and freq_ms 0.
 * Taken for the emission mode on the beam's own ray grid when the frequency pass fits into LDS beside the march
 * tables; RT_HIP_FUSED=2 in the environment keeps the two-kernel run. */
int rt_hip_plan_last_fused(rt_hip_plan *plan);

/* Timing many back-to-back runs without waiting for each: keep the event triples of the last n_runs runs
 * (1 <= n_runs <= 4096); rt_hip_plan_ring_times waits for the last run and returns the kernel durations of
 * the most recent runs, oldest first (at most max_runs of them; *n_runs = how many).  Without a ring a plan
 * keeps the events of its last run only (rt_hip_plan_kernel_times). */
int rt_hip_plan_set_timing_ring(rt_hip_plan *plan, int n_runs);
int rt_hip_plan_ring_times(rt_hip_plan *plan, float *march_ms, float *freq_ms, int max_runs, int *n_runs);

/* Device pointers of the plan's own output buffers (for RCCL / torch views). */
double *rt_hip_plan_image_ptr(rt_hip_plan *plan);
double *rt_hip_plan_iang_ptr(rt_hip_plan *plan);

/* Per-ray march record of the last run, for parity tests against the oracle:
 * gvl/evl [n][L][3] float, ivl [n][L][3] int32, ray2 [n] rt_ray,
 * flags [n] (bit0 escaped, bit1 error -1, bit2 frequency pass skipped),
 * steps [n] cell-steps.  Any pointer may be NULL.  Requires
 * rt_hip_plan_enable_probe(plan, 1) before the run. */
int rt_hip_plan_enable_probe(rt_hip_plan *plan, int on);
int rt_hip_plan_fetch_probe(rt_hip_plan *plan, float *gvl, float *evl, int32_t *ivl,
                            rt_ray *ray2, uint32_t *flags, uint32_t *steps);

/* Emission (ASE) mode only.  By default the frequency pass advances
 *   Iv' = Iv + (e^gl - 1)(Iv + evl/gvl)
 * with the ratio taken once per sub-segment; the CPU (Helper.h:549-557) divides the two float32
 * products el/gl per frequency, which differs by one float rounding per term: image and I_ang agree
 * with RayTraceImageCPULoop to ~1e-8 rel-L2.  on = 1 runs the CPU's formula as written (~1e-14,
 * about twice the time of the frequency kernel).  No effect with a seed (gain-only mode). */
int rt_hip_plan_set_exact_emission(rt_hip_plan *plan, int on);

/* Step safety factor c of the integrator (`c` of RayTrace_calc_ray, Helper.h:381;
 * create_image always uses 0.5, RayTrace::calc_ray_path passes its own).  0 < c < 1. */
int rt_hip_plan_set_step_factor(rt_hip_plan *plan, double c);

/* Path tracer, replaces RayTrace::calc_ray_path (src/RayTraceImage.cpp:440-477): with it
 * enabled a run produces, instead of the image, for every ray the {x, y, I} triples at
 * the 3 (N-1) + 1 sub-segment boundaries -- exactly the `debug` array of
 * RayTrace_calc_ray (Helper.h:419-426, 505-511, 536-542, 559-566) -- and the ray's
 * return code (0, -1, -2, -3).  path: [n_rays][3 (N-1) + 1][3] floats, err: [n_rays]. */
int rt_hip_plan_enable_path(rt_hip_plan *plan, int on);
int rt_hip_plan_fetch_path(rt_hip_plan *plan, float *path, int32_t *err);

/* Profiling aid (no reference counterpart): bit 0 = skip the frequency / deposit kernel, bit 1 = skip
 * the march and run the frequency pass over the records of the previous run of this plan, bit 2 = the
 * frequency kernel keeps its per-work-group I_ang sums to itself (I_ang stays zero).  0 = normal. */
int rt_hip_plan_set_debug(rt_hip_plan *plan, unsigned bits);

void rt_hip_plan_destroy(rt_hip_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
