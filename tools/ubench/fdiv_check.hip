// Diagnostic (not product): where does the short division sequence of rt_math.h (fdiv_nr) differ from the
// hardware's IEEE division?  Prints mismatch counts per (dividend exponent, divisor exponent) band and examples.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o fdiv_check fdiv_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ __forceinline__ float fdiv_nr(float a, float b)
{
    float r       = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r             = fmaf(e, r, r);
    float q       = a * r;
    q             = fmaf(fmaf(-b, q, a), r, q);
    return fmaf(fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// out: [0] checked, [1] bad, then per (ea+128, eb+128) bad counts [256*256], then examples
__global__ void k(unsigned long long *out, unsigned *ex, int ea_lo, int ea_n, int eb_lo, int eb_n, unsigned long long n)
{
    const unsigned long long tid = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nth = (unsigned long long) gridDim.x * blockDim.x;
    for (unsigned long long t = tid; t < n; t += nth) {
        const unsigned h0 = hash((unsigned) t * 2u + 1u), h1 = hash((unsigned) t * 2u + 0x9e3779b9u);
        const int ea = ea_lo + (int) (h0 % (unsigned) ea_n), eb = eb_lo + (int) (h1 % (unsigned) eb_n);
        float x = __uint_as_float(((unsigned) (ea + 127) << 23) | (hash(h0) & 0x7fffffu));
        float y = __uint_as_float(((unsigned) (eb + 127) << 23) | (hash(h1) & 0x7fffffu));
        const float a = fdiv_nr(x, y);
        asm volatile("" : "+v"(x), "+v"(y));
        const float b = x / y;
        atomicAdd(&out[0], 1ull);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            unsigned long long i = atomicAdd(&out[1], 1ull);
            atomicAdd(&out[2 + (ea + 128) * 256 + (eb + 128)], 1ull);
            if (i < 16) { ex[4*i] = __float_as_uint(x); ex[4*i+1] = __float_as_uint(y); ex[4*i+2] = __float_as_uint(a); ex[4*i+3] = __float_as_uint(b); }
        }
    }
}
int main()
{
    unsigned long long *d; unsigned *e;
    const size_t nb = (2 + 256 * 256) * 8;
    hipMalloc(&d, nb); hipMalloc(&e, 64 * 4);
    hipMemset(d, 0, nb); hipMemset(e, 0, 256);
    k<<<1024, 256>>>(d, e, -84, 84, -78, 122, 1ull << 28);
    hipDeviceSynchronize();
    static unsigned long long h[2 + 256 * 256]; unsigned ex[64];
    hipMemcpy(h, d, nb, hipMemcpyDeviceToHost); hipMemcpy(ex, e, 256, hipMemcpyDeviceToHost);
    printf("checked %llu bad %llu\n", h[0], h[1]);
    for (int ea = -128; ea < 128; ea++) for (int eb = -128; eb < 128; eb++) {
        unsigned long long c = h[2 + (ea + 128) * 256 + (eb + 128)];
        if (c) printf("  ea %4d eb %4d (ea-eb %4d): %llu\n", ea, eb, ea - eb, c);
    }
    for (int i = 0; i < 16 && i < (int) h[1]; i++) {
        float x, y, a, b; memcpy(&x, &ex[4*i], 4); memcpy(&y, &ex[4*i+1], 4); memcpy(&a, &ex[4*i+2], 4); memcpy(&b, &ex[4*i+3], 4);
        printf("  x %a y %a nr %a (%08x) ieee %a (%08x)\n", x, y, a, ex[4*i+2], b, ex[4*i+3]);
    }
    return 0;
}
