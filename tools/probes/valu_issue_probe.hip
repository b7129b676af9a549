// Probe: how many vector wave-instructions per cycle one SIMD of this part issues, by instruction kind and by the number of waves
// resident on the SIMD — the calibration of bench.py's "issue" roofline.  Every CU gets W workgroups of 256 threads
// (= W waves per SIMD; the LDS request keeps a (W+1)-th away), each wave runs ITER x 64 instructions of one kind in CHAINS independent
// dependency chains and reads the shader clock (s_memtime) around them; the wall clock of the launch gives the sustained frequency.
//   rate = W * instructions per wave / cycles of the slowest wave   [wave-instructions per cycle per SIMD]
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define ITER 2048

enum Kind { ADD_U32, FMA_F32, MUL_F32, FMA_F64, ADD_F64, MUL_LO_U32, MAD_U64, LSHL_OR, CNDMASK, PK_FMA_F32, RCP_F32, BFE_U32, POPC_B64, SALU_ADD, AND_B32, LSHR_B32, ADD3_U32, AND_OR, MOV_B32, CMP_U32, N_KINDS };
static const char *kNames[N_KINDS] = {"v_add_u32", "v_fma_f32", "v_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_lo_u32", "v_mad_u64_u32", "v_lshl_or_b32",
                                      "v_cndmask_b32", "v_pk_fma_f32", "v_rcp_f32", "v_bfe_u32", "v_bcnt_u32_b32", "s_add_u32 (scalar)", "v_and_b32", "v_lshrrev_b32", "v_add3_u32", "v_and_or_b32", "v_mov_b32", "v_cmp_lt_u32"};

// 64 instructions of kind K over CH chains (registers r0..r7 or d0..d7)
template <int K, int CH> __device__ __forceinline__ void body(uint32_t (&r)[8], double (&d)[8], uint32_t &s) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int c = j % CH;
        if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == FMA_F32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[c]));
        if (K == MUL_F32) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[c]));
        if (K == FMA_F64) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[c]));
        if (K == ADD_F64) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[c]));
        if (K == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(r[c]));
        if (K == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d[c]) : "v"(r[c]) : "vcc");
        if (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == LSHR_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[c]));
        if (K == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == CMP_U32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[c]), "v"(r[(c + 1) & 7]) : "vcc");
        if (K == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %0" : "+v"(r[c]));
        if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(r[(c + 1) & 7]));
        if (K == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[c]));
        if (K == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[c]));
        if (K == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r[c]));
        if (K == POPC_B64) asm volatile("v_bcnt_u32_b32 %0, %0, %0" : "+v"(r[c]));
        if (K == SALU_ADD) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
    }
}

template <int K, int CH> __global__ __launch_bounds__(256) void k_probe(unsigned long long *out, uint32_t seed) {
    extern __shared__ unsigned char pad[];
    uint32_t r[8];
    double d[8];
    uint32_t s = seed;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        r[c] = seed + threadIdx.x * 8 + c;
        d[c] = 1.0 + 1e-9 * (double)(threadIdx.x + c);
    }
    if (seed == 0xFFFFFFFFu) pad[threadIdx.x] = 1; // (keeps the LDS request alive)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(); // s_memtime
    for (int it = 0; it < ITER; ++it) body<K, CH>(r, d, s);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = s;
    double dacc = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        acc ^= r[c];
        dacc += d[c];
    }
    if (acc == 0x12345678u && dacc == 3.25) out[0] = 1; // (keeps the results alive)
    if ((threadIdx.x & 63) == 0) atomicMax(&out[1], t1 - t0);
    if ((threadIdx.x & 63) == 0) atomicMin(&out[2], t1 - t0);
}

template <int K, int CH> static void run(int W, int n_cu, unsigned long long *d_out, bool header) {
    unsigned long long init[3] = {0, 0, ~0ull};
    hipMemcpy(d_out, init, sizeof init, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(160 * 1024 / W) - 1024; // W workgroups fill a CU's LDS: no (W+1)-th
    hipFuncSetAttribute((const void *)k_probe<K, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<K, CH>), dim3(n_cu * W), dim3(256), lds, 0, d_out, 1u); // warm-up (clocks, code)
    hipDeviceSynchronize();
    hipMemcpy(d_out, init, sizeof init, hipMemcpyHostToDevice);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_probe<K, CH>), dim3(n_cu * W), dim3(256), lds, 0, d_out, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    const double n_inst = (double)ITER * 64.0;
    // s_memtime counts at a constant 100 MHz on this part? -> report both the counter's own unit and the wall-clock view
    const double cyc_max = (double)h[1], cyc_min = (double)h[2];
    const double wall_rate_ghz = (double)W * n_inst / (ms * 1e6); // wave-instructions per ns per SIMD
    printf("%-28s chains %d  waves/SIMD %d : %7.3f ms  counter ticks/wave %9.0f..%9.0f  -> %6.3f inst/tick/SIMD, %6.3f G wave-inst/s/SIMD (wall)\n", kNames[K], CH, W, ms, cyc_min,
           cyc_max, (double)W * n_inst / cyc_max, wall_rate_ghz);
}

template <int K> static void sweep(int n_cu, unsigned long long *d_out) {
    for (int W : {1, 2, 4, 8}) run<K, 8>(W, n_cu, d_out, false);
    run<K, 1>(1, n_cu, d_out, false); // one dependent chain, one wave: the dependent-issue interval
    run<K, 1>(8, n_cu, d_out, false);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    unsigned long long *d_out;
    hipMalloc(&d_out, 3 * sizeof(unsigned long long));
    const int n_cu = p.multiProcessorCount;
    sweep<ADD_U32>(n_cu, d_out);
    sweep<FMA_F32>(n_cu, d_out);
    sweep<MUL_F32>(n_cu, d_out);
    sweep<FMA_F64>(n_cu, d_out);
    sweep<ADD_F64>(n_cu, d_out);
    sweep<MUL_LO_U32>(n_cu, d_out);
    sweep<MAD_U64>(n_cu, d_out);
    sweep<LSHL_OR>(n_cu, d_out);
    sweep<CNDMASK>(n_cu, d_out);
    sweep<PK_FMA_F32>(n_cu, d_out);
    sweep<RCP_F32>(n_cu, d_out);
    sweep<BFE_U32>(n_cu, d_out);
    sweep<POPC_B64>(n_cu, d_out);
    sweep<SALU_ADD>(n_cu, d_out);
    sweep<AND_B32>(n_cu, d_out);
    sweep<LSHR_B32>(n_cu, d_out);
    sweep<ADD3_U32>(n_cu, d_out);
    sweep<AND_OR>(n_cu, d_out);
    sweep<MOV_B32>(n_cu, d_out);
    sweep<CMP_U32>(n_cu, d_out);
    return 0;
}
