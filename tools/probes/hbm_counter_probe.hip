// Probe: what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for kernels whose HBM-side byte counts are known, by access width —
// the calibration of bench.py's roofline.traffic.  Each kernel moves BYTES = 1 GiB (far beyond the 256 MiB
// Infinity Cache) once.  Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./hbm_counter_probe
//                              rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- ./hbm_counter_probe
// then tools/hbm_counter_table.py <fetch csv> <write csv> prints counter / true bytes per kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_counter_probe hbm_counter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define BYTES (1ull << 30)

template <class T> __global__ void k_store(T *dst, size_t n, T v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
template <class T> __global__ void k_load(const T *src, size_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        acc += *(const unsigned char *)&v;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
// one 4-byte store per 64-byte line (a scattered list: partial-line writes)
__global__ void k_store_sparse(uint32_t *dst, size_t n_lines) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) dst[i * 16] = (uint32_t)i;
}
// one 4-byte load per 64-byte line
__global__ void k_load_sparse(const uint32_t *src, size_t n_lines, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) acc += src[i * 16];
    if (acc == 0x123456789ull) out[0] = acc;
}
// fire-and-forget atomics, one per 64-byte line
__global__ void k_atomic_sparse(uint32_t *dst, size_t n_lines) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_fetch_or(&dst[i * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    void *buf;
    unsigned long long *out;
    hipMalloc(&buf, BYTES);
    hipMalloc(&out, 8);
    hipMemset(buf, 0, BYTES);
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_store<uint4>, g, b, 0, 0, (uint4 *)buf, BYTES / 16, make_uint4(1, 2, 3, 4));
        hipLaunchKernelGGL(k_store<uint2>, g, b, 0, 0, (uint2 *)buf, BYTES / 8, make_uint2(1, 2));
        hipLaunchKernelGGL(k_store<uint32_t>, g, b, 0, 0, (uint32_t *)buf, BYTES / 4, 7u);
        hipLaunchKernelGGL(k_store<uint16_t>, g, b, 0, 0, (uint16_t *)buf, BYTES / 2, (uint16_t)7);
        hipLaunchKernelGGL(k_store<uint8_t>, g, b, 0, 0, (uint8_t *)buf, BYTES / 4, (uint8_t)7); // (a quarter of the buffer)
        hipLaunchKernelGGL(k_store_sparse, g, b, 0, 0, (uint32_t *)buf, BYTES / 64);
        hipLaunchKernelGGL(k_atomic_sparse, g, b, 0, 0, (uint32_t *)buf, BYTES / 64);
        hipLaunchKernelGGL(k_load<uint4>, g, b, 0, 0, (const uint4 *)buf, BYTES / 16, out);
        hipLaunchKernelGGL(k_load<uint2>, g, b, 0, 0, (const uint2 *)buf, BYTES / 8, out);
        hipLaunchKernelGGL(k_load<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, BYTES / 4, out);
        hipLaunchKernelGGL(k_load<uint16_t>, g, b, 0, 0, (const uint16_t *)buf, BYTES / 2, out);
        hipLaunchKernelGGL(k_load_sparse, g, b, 0, 0, (const uint32_t *)buf, BYTES / 64, out);
    }
    hipDeviceSynchronize();
    printf("done: every kernel moved %llu bytes (u8 store: a quarter; sparse: one 4-byte access per 64-byte line of the buffer)\n", (unsigned long long)BYTES);
    return 0;
}
