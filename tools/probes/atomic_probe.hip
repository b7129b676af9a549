// Probe: cost of global atomics on MI355X as a function of which XCDs touch a region and of the memory scope.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned int xcc_id() {
    unsigned int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

template <int SCOPE, int AFFINE, int RET>
__global__ void k_probe(uint32_t *base, size_t words_per_frame, int n_frames, int K, uint32_t *xcc_hist, uint32_t *sink) {
    const unsigned int xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xcc], 1u);
    int f;
    if (AFFINE) f = (int)xcc + 8 * (int)((blockIdx.x >> 3) % (unsigned)(n_frames / 8));
    else f = (int)((blockIdx.x >> 3) % (unsigned)n_frames);
    uint32_t *A = base + (size_t)f * words_per_frame;
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int k = 0; k < K; ++k) {
        h = h * 1664525u + 1013904223u;
        // addresses clustered like awareness cells: a block works inside a 4 KiB window picked per block, 16-byte records
        const size_t win = ((size_t)(blockIdx.x * 40503u) % (words_per_frame / 1024)) * 1024;
        const size_t idx = win + ((h >> 8) & 1023u);
        if (RET) acc += __hip_atomic_fetch_add(A + idx, 1u, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(A + idx, 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int SCOPE, int AFFINE, int RET> void run(const char *name, uint32_t *d, size_t wpf, int nf, uint32_t *hist, uint32_t *sink) {
    const int blocks = 19200, K = 8;
    hipMemset(d, 0, wpf * nf * 4);
    hipMemset(hist, 0, 128 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k_probe<SCOPE, AFFINE, RET><<<blocks, 256>>>(d, wpf, nf, K, hist, sink); // warm
    hipMemset(d, 0, wpf * nf * 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_probe<SCOPE, AFFINE, RET><<<blocks, 256>>>(d, wpf, nf, K, hist, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(wpf * nf);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    for (auto v : h) sum += v;
    const unsigned long long expect = (unsigned long long)blocks * 256 * K;
    printf("%-44s %8.1f us  %6.2f G atomics/s  sum %llu / %llu %s\n", name, ms * 1e3, expect / (ms * 1e6), sum, expect,
           sum == expect ? "OK" : "LOST UPDATES");
}

int main() {
    const size_t wpf = 1u << 20; // 4 MiB per frame
    const int nf = 16;
    uint32_t *d, *hist, *sink;
    hipMalloc(&d, wpf * nf * 4);
    hipMalloc(&hist, 128 * 4);
    hipMalloc(&sink, 4);
    run<__HIP_MEMORY_SCOPE_AGENT, 0, 0>("agent scope, frames spread over XCDs, noret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_AGENT, 0, 1>("agent scope, frames spread over XCDs, ret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_AGENT, 1, 0>("agent scope, frame = XCD affine, noret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_AGENT, 1, 1>("agent scope, frame = XCD affine, ret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 0, 0>("workgroup scope, spread, noret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 1, 0>("workgroup scope, XCD affine, noret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 1, 1>("workgroup scope, XCD affine, ret", d, wpf, nf, hist, sink);
    run<__HIP_MEMORY_SCOPE_SYSTEM, 0, 0>("system scope, spread, noret", d, wpf, nf, hist, sink);
    std::vector<uint32_t> h(128);
    hipMemcpy(h.data(), hist, 512, hipMemcpyDeviceToHost);
    printf("blockIdx%%8 -> XCC_ID histogram (rows blockIdx&7):\n");
    for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) printf("%6u", h[r * 16 + c]);
        printf("\n");
    }
    return 0;
}
