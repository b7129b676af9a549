// Probe: is v_cndmask_b32 as slow as tools/probes/valu_issue_probe measured it (0.10 G wave-instructions/s per SIMD, a tenth of v_mul_f32)?
// Variants: VOP2 with VCC set once outside the loop, VOP3 with an SGPR pair, the compiler's own selects, v_cmp + v_cndmask pairs, and
// v_mul_f32 as the yardstick.  Same harness: 256 CUs x W workgroups of 256 threads, ITER x 64 instructions per wave, wall clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o cndmask_probe cndmask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define ITER 2048

template <int K> __global__ __launch_bounds__(256) void k_probe(unsigned long long *out, uint32_t seed) {
    extern __shared__ unsigned char pad[];
    uint32_t r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = seed * 2654435761u + threadIdx.x * 8 + c;
    if (seed == 0xFFFFFFFFu) pad[threadIdx.x] = 1;
    unsigned long long sm = ((unsigned long long)seed << 32) | 0x5555AAAA3333CCCCull;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(r[0]), "v"(r[1]) : "vcc");
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = j & 7;
            if (K == 0) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[c]));
            if (K == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[c]) : "v"(r[(c + 1) & 7]), "s"(sm));
            if (K == 3) r[c] = (r[(c + 3) & 7] & (1u << (j & 31))) ? r[c] + 1u : r[(c + 1) & 7]; // the compiler's select (v_cmp / v_and + v_cndmask)
            if (K == 4) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(r[(c + 1) & 7]) : "vcc");
            if (K == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 7) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 8) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 9) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[c]));
            if (K == 10) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 12) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
            if (K == 13) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r[c]));
            if (K == 14) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(r[c]));
            if (K == 15) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[c]));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc ^= r[c];
    if (acc == 0x12345678u) out[0] = 1;
}
static const char *names[] = {"v_mul_f32 (yardstick)", "v_cndmask_b32 vcc (VOP2)", "v_cndmask_b32_e64 sgpr pair", "compiler select (C)", "v_cmp + v_cndmask pair (2 inst)",
                              "v_mov_b32", "v_and_b32", "v_add3_u32", "v_and_or_b32", "v_lshrrev_b32", "v_lshl_add_u32", "v_xor_b32", "v_bcnt_u32_b32", "v_ffbl_b32", "v_sub_f32", "v_cvt_f32_u32"};
template <int K> static void run(int n_cu, unsigned long long *d_out) {
    for (int W : {1, 4, 8}) {
        const size_t lds = (size_t)(160 * 1024 / W) - 1024;
        (void)hipFuncSetAttribute((const void *)k_probe<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k_probe<K>), dim3(n_cu * W), dim3(256), lds, 0, d_out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_probe<K>), dim3(n_cu * W), dim3(256), lds, 0, d_out, 1u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (K == 4 ? 2.0 : 1.0) * ITER * 64.0;
        printf("%-34s waves/SIMD %d : %7.3f ms -> %6.3f G wave-inst/s/SIMD\n", names[K], W, ms, (double)W * n / (ms * 1e6));
    }
}
int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 64);
    const int n = p.multiProcessorCount;
    run<0>(n, d_out); run<1>(n, d_out); run<2>(n, d_out); run<3>(n, d_out); run<4>(n, d_out); run<5>(n, d_out); run<6>(n, d_out); run<7>(n, d_out);
    run<8>(n, d_out); run<9>(n, d_out); run<10>(n, d_out); run<11>(n, d_out); run<12>(n, d_out); run<13>(n, d_out); run<14>(n, d_out); run<15>(n, d_out);
    return 0;
}
