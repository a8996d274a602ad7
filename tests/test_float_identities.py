"""Floating-point identities the HIP kernel relies on to reproduce the
reference's mixed-precision expressions with cheaper instructions."""
import numpy as np


def test_f32_reciprocal_equals_double_rounded_reciprocal():
    """Helper.h:83-84 computes (float)(1.0 / (double)sqrtf(q)); the kernel uses
    the correctly rounded float division 1.0f / y.  Exhaustive over every float
    in [0.25, 4): q = |s|^2 stays within a few ulp of 1."""
    lo = np.float32(0.25).view(np.uint32)
    hi = np.float32(4.0).view(np.uint32)
    step = 1 << 22
    for start in range(int(lo), int(hi), step):
        y = np.arange(start, min(start + step, int(hi)), dtype=np.uint32).view(np.float32)
        a = (np.float64(1.0) / y.astype(np.float64)).astype(np.float32)
        b = np.float32(1.0) / y
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


_TANF_C = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float fb(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static inline uint32_t bf(float f){uint32_t u; memcpy(&u,&f,4); return u;}
/* raytrace-miniapp_amd/csrc/rt_march.hip, tanf_flt32_kernel, restated for the host */
static float k(float x){
  const float ax = fabsf(x);
  if (ax < 0x1p-13f) return x;
  const float T0=3.3333334327e-01f,T1=1.3333334029e-01f,T2=5.3968254477e-02f,T3=2.1869488060e-02f,T4=8.8632395491e-03f,
    T5=3.5920790397e-03f,T6=1.4562094584e-03f,T7=5.8804126456e-04f,T8=2.4646313977e-04f,T9=7.8179444245e-05f,
    T10=7.1407252108e-05f,T11=-1.8558637748e-05f,T12=2.5907305826e-05f;
  const float z=x*x, w=z*z;
  float r=T1+w*(T3+w*(T5+w*(T7+w*(T9+w*T11))));
  const float v=z*(T2+w*(T4+w*(T6+w*(T8+w*(T10+w*T12)))));
  const float s=z*x;
  r=0.0f+z*(s*(r+v)+0.0f); r+=T0*s; return x+r; }
/* rt_march.hip, atanf_flt32_kernel (|x| < 7/16 branch of s_atanf.c) */
static float ka(float x){
  const float ax = fabsf(x);
  if (ax < 0x1p-29f) return x;
  const float A0=3.3333334327e-01f,A1=-2.0000000298e-01f,A2=1.4285714924e-01f,A3=-1.1111110449e-01f,A4=9.0908870101e-02f,
    A5=-7.6918758452e-02f,A6=6.6610731184e-02f,A7=-5.8335702866e-02f,A8=4.9768779427e-02f,A9=-3.6531571299e-02f,A10=1.6285819933e-02f;
  const float z=x*x, w=z*z;
  const float s1=z*(A0+w*(A2+w*(A4+w*(A6+w*(A8+w*A10)))));
  const float s2=w*(A1+w*(A3+w*(A5+w*(A7+w*A9))));
  return x-x*(s1+s2); }
long check(uint32_t stride){ long bad=0;
  for (uint64_t u=0x30000000u; u<=0x3e4ccccdu; u+=stride){ float x=fb((uint32_t)u);
    if (bf(tanf(x))!=bf(k(x))) bad++; if (bf(tanf(-x))!=bf(k(-x))) bad++; }
  for (uint32_t u=1; u<0x30000000u; u+=9973){ float x=fb(u); if (bf(tanf(x))!=bf(k(x))) bad++; }
  for (uint64_t u=0x2f000000u; u<0x3ee00000u; u+=stride){ float x=fb((uint32_t)u);
    if (bf(atanf(x))!=bf(ka(x))) bad++; if (bf(atanf(-x))!=bf(ka(-x))) bad++; }
  return bad; }
"""


def test_tanf_atanf_restatements_equal_the_host_libm(tmp_path):
    """Helper.h:409-410: s = tanf(1e-3f * a).  The kernel restates the float tanf kernel of
    the reference platform's libm (glibc 2.35 flt-32 k_tanf.c, fdlibm); this checks the
    restatement against the host tanf on every 7th float of [4.6e-10, 0.2] (exhaustively
    verified once: 0 mismatches of 2.4e8) and on a stride of the tiny range.  A failure here
    means the host libm computes tanf differently: list-mode tangents would then differ
    from the CPU loop in the last bit (grid mode uses the host's own tanf and is unaffected).
    Same for atanf (Helper.h:520-521, exit angles) on its |x| < 7/16 branch."""
    import ctypes
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(_TANF_C)
    so = tmp_path / "libt.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
    lib = ctypes.CDLL(str(so))
    lib.check.restype = ctypes.c_long
    lib.check.argtypes = [ctypes.c_uint32]
    assert lib.check(7) == 0


_TANF_WIDE_C = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float fb(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static inline uint32_t bf(float f){uint32_t u; memcpy(&u,&f,4); return u;}
/* raytrace-miniapp_amd/csrc/rt_march.hip, ktanf_flt32 / tanf_flt32_wide, restated for the host */
static const float T[13]={3.3333334327e-01f,1.3333334029e-01f,5.3968254477e-02f,2.1869488060e-02f,8.8632395491e-03f,
    3.5920790397e-03f,1.4562094584e-03f,5.8804126456e-04f,2.4646313977e-04f,7.8179444245e-05f,
    7.1407252108e-05f,-1.8558637748e-05f,2.5907305826e-05f};
static float ktan(float x, float y, int iy){
  const float pio4=7.8539812565e-01f, pio4lo=3.7748947079e-08f;
  float z,r,v,w,s; int32_t hx=(int32_t)bf(x), ix=hx&0x7fffffff;
  if (ix<0x39000000){ if ((int)x==0){ if ((ix|(iy+1))==0) return 1.0f/fabsf(x); else if (iy==1) return x; else return -1.0f/x; } }
  if (ix>=0x3f2ca140){ if (hx<0){x=-x;y=-y;} z=pio4-x; w=pio4lo-y; x=z+w; y=0.0f;
     if (fabsf(x)<0x1p-13f) return (1-((hx>>30)&2))*iy*(1.0f-2*iy*x); }
  z=x*x; w=z*z;
  r=T[1]+w*(T[3]+w*(T[5]+w*(T[7]+w*(T[9]+w*T[11]))));
  v=z*(T[2]+w*(T[4]+w*(T[6]+w*(T[8]+w*(T[10]+w*T[12])))));
  s=z*x; r=y+z*(s*(r+v)+y); r+=T[0]*s; w=x+r;
  if (ix>=0x3f2ca140){ v=(float)iy; return (float)(1-((hx>>30)&2))*(v-(float)2.0*(x-(w*w/(w+v)-r))); }
  if (iy==1) return w;
  { float a,t; int32_t i; z=w; i=(int32_t)bf(z); z=fb((uint32_t)(i&0xfffff000)); v=r-(z-x); t=a=-(float)1.0/w;
    i=(int32_t)bf(t); t=fb((uint32_t)(i&0xfffff000)); s=(float)1.0+t*z; return t+a*(s+t*v); }
}
static float mytan(float x){
  const float pio2_1=1.5707855225e+00f,pio2_1t=1.0804334124e-05f,pio2_2=1.0804273188e-05f,pio2_2t=6.0770999344e-11f;
  int32_t hx=(int32_t)bf(x), ix=hx&0x7fffffff; float y0,y1,z;
  if (ix<=0x3f490fda) return ktan(x,0.0f,1);
  if (ix<0x4016cbe4){
    if (hx>0){ z=x-pio2_1; if ((ix&0xfffffff0)!=0x3fc90fd0){ y0=z-pio2_1t; y1=(z-y0)-pio2_1t; } else { z-=pio2_2; y0=z-pio2_2t; y1=(z-y0)-pio2_2t; } }
    else { z=x+pio2_1; if ((ix&0xfffffff0)!=0x3fc90fd0){ y0=z+pio2_1t; y1=(z-y0)+pio2_1t; } else { z+=pio2_2; y0=z+pio2_2t; y1=(z-y0)+pio2_2t; } }
    return ktan(y0,y1,-1); }
  return (float)tan((double)x);
}
long check(uint32_t stride){ long bad=0;
  for (uint64_t u=0x3e4ccccdu; u<=0x3fb00000u; u+=stride){ float x=fb((uint32_t)u);   /* [0.2, 1.375] */
    if (bf(tanf(x))!=bf(mytan(x))) bad++; if (bf(tanf(-x))!=bf(mytan(-x))) bad++; }
  return bad; }
"""


def test_wide_angle_tanf_restatement_equals_the_host_libm(tmp_path):
    """Launch angles of 200 mrad ... 1.375 rad in list mode: the k_tanf.c branches for |x| >= 0.6744 and
    for the reduced argument of e_rem_pio2f.c's |x| < 3 pi/4 case, as rt_tan_kernel evaluates them, against
    the host tanf on every 5th float of the range (exhaustively verified once: 0 mismatches of 2 x 2.3e7)."""
    import ctypes
    import subprocess
    src = tmp_path / "tw.c"
    src.write_text(_TANF_WIDE_C)
    so = tmp_path / "libtw.so"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
    lib = ctypes.CDLL(str(so))
    lib.check.restype = ctypes.c_long
    lib.check.argtypes = [ctypes.c_uint32]
    assert lib.check(5) == 0


_MARKSTEIN_C = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float fb(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static inline uint32_t bf(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline uint64_t bd(double f){uint64_t u; memcpy(&u,&f,8); return u;}
/* raytrace-miniapp_amd/csrc/rt_math.h, div_by_recip, restated for the host */
static inline float d32(float a, float b, float y){ float q=a*y, r=fmaf(-b,q,a), c=copysignf(fmaf(r,y,q), q);
    if (fabsf(a) < 1e-29f && a != 0.0f) c = a/b; return c; }
static inline double d64(double a, double b, double y){ double q=a*y, r=fma(-b,q,a), c=copysign(fma(r,y,q), q);
    if (fabs(a) < 1e-280 && a != 0.0) c = a/b; return c; }
static uint64_t s = 88172645463325252ULL;
static inline uint64_t xr(void){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
/* returns the number of mismatches against IEEE division */
long check(long n_random, uint32_t const_stride)
{
    long bad = 0;
    const float cs[3] = {3.0f, 6.0f, 12.0f};
    for (int c = 0; c < 3; c++) {               /* constant divisors: every const_stride-th float, both signs */
        const float b = cs[c], y = 1.0f / b;
        for (uint64_t u = 0; u < 0x7f800000u; u += const_stride) {
            const float a = fb((uint32_t) u);
            if (bf(a / b) != bf(d32(a, b, y))) bad++;
            if (bf(-a / b) != bf(d32(-a, b, y))) bad++;
        }
        for (uint32_t u = 0; u < 0x02000000u; u++) {   /* all of the subnormal / tiny range */
            const float a = fb(u);
            if (bf(a / b) != bf(d32(a, b, y))) bad++;
        }
    }
    for (long i = 0; i < n_random; i++) {        /* divisor = index of refraction ~ 1, any dividend */
        uint64_t r = xr();
        float b = fb(0x3f000000u + (uint32_t)(r & 0x00ffffffu));
        uint32_t ea = (uint32_t)((r >> 24) % 254);
        float a = fb((ea << 23) | ((uint32_t)(r >> 40) & 0x7fffffu));
        if (r >> 63) a = -a;
        if (bf(a / b) != bf(d32(a, b, 1.0f / b))) bad++;
    }
    for (long i = 0; i < n_random; i++) {        /* f64: divisor a cell width (a float or a double), dividend any */
        uint64_t r = xr(), r2 = xr();
        double b = (i & 1) ? (double) fb(((uint32_t)(100 + (r % 60)) << 23) | ((uint32_t)(r >> 8) & 0x7fffffu))
                           : 0;
        if (!(i & 1)) { uint64_t ub = ((uint64_t)(1000 + (r % 40)) << 52) | (r >> 12); memcpy(&b, &ub, 8); }
        uint64_t ua = (((r2 % 400) + 823) << 52) | (r2 >> 12); double a; memcpy(&a, &ua, 8);
        if (r2 & 1) a = -a;
        if (bd(a / b) != bd(d64(a, b, 1.0 / b))) bad++;
    }
    return bad;
}
"""


def test_division_by_correctly_rounded_reciprocal_is_ieee_division(tmp_path):
    """rt_math.h div_by_recip (Markstein's correction with the two guarded cases)
    against IEEE division, compiled for the host with true fused multiply-add."""
    import ctypes
    import subprocess
    src = tmp_path / "mk.c"
    src.write_text(_MARKSTEIN_C)
    so = tmp_path / "libmk.so"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
    lib = ctypes.CDLL(str(so))
    lib.check.restype = ctypes.c_long
    lib.check.argtypes = [ctypes.c_long, ctypes.c_uint32]
    assert lib.check(20_000_000, 97) == 0


_INV_NORM_C = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
static float fb(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* rt_math.h, inv_norm: the shortcut branch, with y standing for the result of v_rsq_f32 */
static float inv_norm_fast(float q, float y)
{
    const float s0 = q * y;
    const float e  = fmaf(-s0, s0, q);
    const float s1 = fmaf(e, 0.5f * y, s0);
    const float r  = fmaf(-s1, y, 1.0f);
    float inv      = fmaf(r, y, y);
    const uint32_t u = bf(s1);
    if ((u & 0x7fffffu) == 0x7fffffu) inv = fb(0x7f000000u - u);
    return inv;
}
/* every q with |q - 1| < 0.0625, every float y within 1 ulp of 1/sqrt(q) (the accuracy the ISA
 * states for v_rsq_f32): mismatches against the reference's 1.0f / sqrtf(q) */
long check(void)
{
    long bad = 0;
    for (uint32_t u = bf(0.9375f); u <= bf(1.0625f); u++) {
        const float q = fb(u);
        if (!(fabsf(q - 1.0f) < 0.0625f)) continue;
        const float want = 1.0f / sqrtf(q);
        const double ex  = 1.0 / sqrt((double) q);
        const float yt   = (float) ex;
        for (int k = -1; k <= 1; k++) {
            const float y   = fb(bf(yt) + k);
            const double ul = (double) fb(bf(yt) + 1) - (double) yt;
            if (fabs((double) y - ex) > ul) continue;
            if (bf(inv_norm_fast(q, y)) != bf(want)) bad++;
        }
    }
    return bad;
}
"""


def test_inv_norm_shortcut_equals_ieee_sqrt_and_division(tmp_path):
    """rt_math.h inv_norm: rsq + one Markstein step + one Newton step (+ the all-ones closed form)
    against 1.0f / sqrtf(q) over the whole shortcut range, for any 1-ulp rsq result."""
    import ctypes
    import subprocess
    src = tmp_path / "inv.c"
    src.write_text(_INV_NORM_C)
    so = tmp_path / "libinv.so"
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"], check=True)
    lib = ctypes.CDLL(str(so))
    lib.check.restype = ctypes.c_long
    assert lib.check() == 0


def test_float_thresholds_equal_the_double_comparisons():
    """rt_march.hip compares in float where the reference compares a widened float with
    a double literal: (double)x < 0.05 <=> x < 0.05f (Helper.h:280) and
    (double)x < 0.01 <=> x <= 0.01f (Helper.h:466, :515).  Comparisons are monotone in x,
    so checking the floats around each literal is exhaustive."""
    for lit, as_float, op in ((0.05, np.float32(0.05), np.less), (0.01, np.float32(0.01), np.less_equal)):
        centre = as_float.view(np.uint32)
        x = np.arange(int(centre) - 1000, int(centre) + 1000, dtype=np.uint32).view(np.float32)
        x = np.concatenate([x, np.float32([0.0, 1e-30, 1.0, np.inf])])
        want = x.astype(np.float64) < lit
        got = op(x, as_float)
        assert np.array_equal(want, got)
    assert float(np.float32(0.05)) > 0.05 and float(np.float32(0.01)) < 0.01


def test_source_function_form_of_the_ase_update_stays_within_its_bound():
    """rt_freq.hip ase_step, restated in numpy, against the CPU formula (Helper.h:549-557) over
    the range of the shipped tables and far beyond: six chained updates per sample, as a ray
    sees them.  Bounds: the expm1 construction alone (same ratio on both sides) 1e-9; with the
    per-sub-segment ratio es/gs, one float rounding per term, 2e-7."""
    import numpy as np
    rng = np.random.default_rng(11)
    n = 200_000
    tab = np.exp2(np.arange(256) / 256.0)
    L2E, LN2_N, MAGIC = 369.3299304675746, 0.0027076061740622863, float.fromhex("0x1.8p52")

    def em1_kernel(x):                      # e^x - 1 as the kernel builds it
        t = x * L2E + MAGIC
        nn = (t - MAGIC).astype(np.int64)
        r = x - (t - MAGIC) * LN2_N
        q = 1.0 + r * (0.5 + r / 6.0)
        S = np.ldexp(tab[nn & 255], (nn >> 8).astype(np.int32))
        return S * (r * q) + (S - 1.0)

    def cpu_update(Iv, gl, el):             # Helper.h:551-557
        small = np.abs(gl) < 1e-3
        with np.errstate(divide="ignore", invalid="ignore"):
            big = el / gl * (np.exp(gl) - 1.0) + Iv * np.exp(gl)
        return np.where(small, el * (1.0 + 0.5 * gl * (1.0 + 0.3333333333 * gl)) + Iv * (1.0 + gl * (1.0 + 0.5 * gl)), big)

    Iv_cpu = np.zeros(n)
    Iv_same = np.zeros(n)                   # kernel expm1, CPU ratio el/gl
    Iv_gpu = np.zeros(n)                    # kernel expm1, ratio es/gs
    for s in range(6):
        gs = (rng.uniform(-3.0, 3.0, n) * 10.0 ** rng.uniform(-6, 0, n)).astype(np.float32)
        es = (np.abs(gs) * 10.0 ** rng.uniform(-6, -2, n)).astype(np.float32)
        w = rng.uniform(0.002, 1.0, n).astype(np.float32)
        gl = (gs * w).astype(np.float64)    # float32 product, then widened
        el = (es * w).astype(np.float64)
        Iv_cpu = cpu_update(Iv_cpu, gl, el)
        em1 = em1_kernel(gl)
        Iv_same = Iv_same + em1 * (Iv_same + el / gl)
        Iv_gpu = Iv_gpu + em1 * (Iv_gpu + es.astype(np.float64) / gs.astype(np.float64))
    scale = np.abs(Iv_cpu) + 1e-300
    assert np.max(np.abs(Iv_same - Iv_cpu) / scale) < 1e-9
    assert np.max(np.abs(Iv_gpu - Iv_cpu) / scale) < 2e-7
    assert np.linalg.norm(Iv_gpu - Iv_cpu) / np.linalg.norm(Iv_cpu) < 5e-8
