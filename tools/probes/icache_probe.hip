// Probe: what a COLD instruction cache costs a lone workgroup — a straight line of 4096 eight-byte vector instructions (32 KiB of
// code, about the length of k_sector's path) executed twice in a row by one wave per workgroup: first pass (cold) against second
// (warm), with 1 workgroup on the chip and with 512 (every CU fetching at once).
// Build: hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define I1 asm volatile("v_mad_u32_u24 %0, %0, 3, 1" : "+v"(x));
#define I8 I1 I1 I1 I1 I1 I1 I1 I1
#define I64 I8 I8 I8 I8 I8 I8 I8 I8
#define I512 I64 I64 I64 I64 I64 I64 I64 I64
#define I4096 I512 I512 I512 I512 I512 I512 I512 I512

__global__ void k_line(long long *out, unsigned int *sink) {
    unsigned int x = threadIdx.x;
    long long t[3];
    t[0] = clock64();
    for (int pass = 0; pass < 2; ++pass) {
        I4096
        t[pass + 1] = clock64();
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t[1] - t[0];
        out[2 * blockIdx.x + 1] = t[2] - t[1];
    }
    if (x == 0xdeadbeefu) *sink = x;
}
__global__ void k_other(unsigned int *p) { p[threadIdx.x] = threadIdx.x; } // (evicts nothing: just a different kernel in between)

int main() {
    long long *d;
    unsigned int *s;
    hipMalloc(&d, 2 * 512 * sizeof(long long));
    hipMalloc(&s, 4096);
    static long long h[2 * 512];
    for (int grid : {1, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_other, dim3(1), dim3(64), 0, 0, s);
            hipLaunchKernelGGL(k_line, dim3(grid), dim3(64), 0, 0, d, s);
            hipMemcpy(h, d, 2 * grid * sizeof(long long), hipMemcpyDeviceToHost);
            double c = 0, w = 0, cmax = 0;
            for (int b = 0; b < grid; ++b) {
                c += (double)h[2 * b];
                w += (double)h[2 * b + 1];
                if ((double)h[2 * b] > cmax) cmax = (double)h[2 * b];
            }
            printf("grid %3d rep %d: first pass %7.2f us (max %7.2f), second pass %6.2f us  (4096 instructions, 32 KiB)\n", grid, rep, c / grid / 2400.0, cmax / 2400.0,
                   w / grid / 2400.0);
        }
    }
    return 0;
}
