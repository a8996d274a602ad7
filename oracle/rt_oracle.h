/*
 * rt_oracle.h -- interface of the CPU parity oracle (TEST INFRASTRUCTURE ONLY;
 * see the header of rt_oracle.c).  Uses the record types of the product's
 * C ABI (include/rt_hip.h) so that one input feeds both sides.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_oracle_counters {
    uint64_t n_rays;
    uint64_t cell_steps;  /* Helper.h:463 loop iterations  */
    uint64_t cross_iters; /* Helper.h:326 loop iterations  */
    uint64_t inner_iters; /* Helper.h:279 loop iterations  */
    uint64_t n_escaped;
} rt_oracle_counters;

int rt_oracle_march(const rt_ray *ray, int N, float dz0, const rt_gain *gain, int use_emis,
                    int method, float c, float *gvl, float *evl, int32_t *ivl, rt_ray *ray_out,
                    int *escaped_out, rt_oracle_counters *cnt);

int rt_oracle_integrate(int N, const rt_gain *gain, int use_emis, int K, const float *gvl,
                        const float *evl, const int32_t *ivl, double *Iv);

int rt_oracle_calc_ray(const rt_ray *ray, int N, float dz0, const rt_gain *gain,
                       const rt_seed *seed, int K, int method, double *Iv, rt_ray *ray_out,
                       float *gvl, float *evl, int32_t *ivl, int *escaped_out,
                       rt_oracle_counters *cnt);

int rt_oracle_image_loop(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                         int method, const rt_ray *rays, size_t n_rays, double scale,
                         double *image, double *I_ang, unsigned int *failure_code,
                         rt_ray *failed_rays, int max_failed, int *n_failed,
                         rt_oracle_counters *counters, int n_threads);

int rt_oracle_probe(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                    int method, const rt_ray *rays, size_t n_rays, float *gvl, float *evl,
                    int32_t *ivl, rt_ray *ray2, uint32_t *flags, uint32_t *steps, double *Iv,
                    int32_t *err);

int rt_oracle_calc_ray_path(int N, const rt_beam *beam, const rt_gain *gain, const rt_seed *seed,
                            int method, float c, const rt_ray *rays, size_t n_rays, float *path,
                            int32_t *err);

#ifdef __cplusplus
}
#endif
#endif
