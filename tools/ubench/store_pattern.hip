// Diagnostic (not product): what HBM write bandwidth does the store pattern of the exclusive-mode deposit reach
// (rt_freq.hip: a wave owns 64 image rows of K doubles and writes them in segments of SEG contiguous bytes per row),
// and what would longer segments or neighbouring rows give?  Writes an image of `rows` x 4096 B (K = 512 doubles).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// one wave per 64 rows; row r of wave w is image row (w * 64 + r) * row_stride_rows (mod rows): stride 1 = neighbouring
// rows (4 KB apart), stride 4096 = the kernel's pattern (consecutive rays are nx image rows apart)
template <int SEG> // bytes per row and flush: 128, 256, 512, 1024
__global__ void __launch_bounds__(768) store_kernel(double *img, unsigned long long rows, unsigned long long stride_rows, unsigned *next)
{
    const int lane = threadIdx.x & 63;
    constexpr int LPR = SEG / 8;      // lanes per row segment
    constexpr int RPI = 64 / LPR;     // rows per store instruction
    for (;;) {
        unsigned t = 0;
        if (lane == 0)
            t = atomicAdd(next, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if ((unsigned long long) t * 64 >= rows)
            break;
        for (int kb = 0; kb < 4096; kb += SEG) {                 // all segments of the 64 rows of this tile
            for (int g = 0; g < 64 / RPI; g++) {
                const unsigned long long r  = (unsigned long long) t * 64 + (unsigned) (g * RPI + lane / LPR);
                const unsigned long long rr = (r % 4096) * stride_rows % rows + (r / 4096) * (stride_rows == 1 ? 4096 : 1);
                double *p = img + (stride_rows == 1 ? r : rr) * 512 + kb / 8 + lane % LPR;
                *p = (double) kb;
            }
        }
    }
}

int main()
{
    const unsigned long long rows = 4096ull * 4096ull; // 68.7 GB
    double *img; unsigned *next;
    if (hipMalloc(&img, rows * 4096) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMalloc(&next, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, unsigned long long stride) {
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            hipMemset(next, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(768), 0, 0, img, rows, stride, next);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-28s row stride %5llu: %7.2f ms  %6.2f TB/s\n", name, stride, best, rows * 4096.0 / best / 1e9);
    };
    for (unsigned long long stride : { 1ull, 4096ull }) {
        run(store_kernel<128>, "128 B per row and flush", stride);
        run(store_kernel<256>, "256 B per row and flush", stride);
        run(store_kernel<512>, "512 B per row and flush", stride);
        run(store_kernel<1024>, "1024 B per row and flush", stride);
    }
    return 0;
}
