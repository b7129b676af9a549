// Probe: what a DEPENDENT trip to memory costs a lone workgroup right after another kernel wrote the data (the single-frame
// kernels are chains of such trips) — as a function of where the hops land: distinct allocations (as a frame slot's ~30 arrays),
// one allocation at 2 MiB / 64 KiB / 4 KiB strides, and whether the translation is warm (second pass over the same addresses after
// the writer ran again: the L2 is cold again, the TLBs are not).
// Build: hipcc --offload-arch=gfx950 -O3 -o trip_probe trip_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define HOPS 24

// writer: many workgroups (all XCDs) — element k of the chain gets the ADDRESS of element k + 1
__global__ void k_write(unsigned long long **slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *slots[i] = (unsigned long long)slots[(i + 1) % n];
}
// chaser: one wave walks the chain; per-hop shader clocks
__global__ void k_chase(unsigned long long *start, long long *out) {
    if (threadIdx.x != 0) return;
    unsigned long long p = (unsigned long long)start;
    long long t0 = clock64();
    for (int k = 0; k < HOPS; ++k) {
        p = *(volatile unsigned long long *)p;
        const long long t1 = clock64();
        out[k] = t1 - t0;
        t0 = t1;
    }
    out[HOPS] = (long long)p;
}

static void run(const char *name, std::vector<unsigned long long *> &addr) {
    const int n = (int)addr.size();
    unsigned long long **d_slots;
    long long *d_out;
    hipMalloc(&d_slots, n * sizeof(void *));
    hipMalloc(&d_out, (HOPS + 1) * sizeof(long long));
    hipMemcpy(d_slots, addr.data(), n * sizeof(void *), hipMemcpyHostToDevice);
    long long h[HOPS + 1];
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(k_write, dim3((n + 63) / 64 * 8), dim3(64), 0, 0, d_slots, n);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, addr[0], d_out);
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int k = 1; k < HOPS; ++k) s += (double)h[k];
        printf("%-44s pass %d: first hop %6.0f ns, later hops %6.0f ns each\n", name, pass, h[0] / 2.4, s / (HOPS - 1) / 2.4);
    }
    hipFree(d_slots);
    hipFree(d_out);
}

int main() {
    std::vector<unsigned long long *> a;
    // (a) distinct allocations of 8 MiB each
    std::vector<void *> bufs;
    for (int k = 0; k < HOPS; ++k) {
        void *b;
        hipMalloc(&b, 8u << 20);
        bufs.push_back(b);
        a.push_back((unsigned long long *)b);
    }
    run("24 separate 8 MiB allocations", a);
    // (b) one allocation, strides
    char *big;
    hipMalloc((void **)&big, (size_t)HOPS << 21);
    for (size_t stride : {(size_t)2 << 20, (size_t)64 << 10, (size_t)4 << 10, (size_t)256}) {
        a.clear();
        for (int k = 0; k < HOPS; ++k) a.push_back((unsigned long long *)(big + k * stride));
        char nm[64];
        snprintf(nm, sizeof nm, "one allocation, stride %zu KiB", stride >> 10);
        run(nm, a);
    }
    // (c) a 1 GiB allocation, hops 40 MiB apart (as arrays inside a large frame slot)
    char *huge;
    if (hipMalloc((void **)&huge, (size_t)1 << 30) == hipSuccess) {
        a.clear();
        for (int k = 0; k < HOPS; ++k) a.push_back((unsigned long long *)(huge + (size_t)k * (40u << 20)));
        run("one 1 GiB allocation, stride 40 MiB", a);
    }
    return 0;
}
