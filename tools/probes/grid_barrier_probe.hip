// Probe: what a grid-wide barrier costs inside one kernel on MI355X (all workgroups resident, cooperative launch): a completion
// counter in memory, agent-scope release before the arrival and acquire after the wait — the hand-over a fused single-frame
// kernel would use between its phases instead of kernel boundaries (4.4 us in a HIP graph).
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip ; run on the GPU box under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_barriers(unsigned int *cnt, unsigned int *data, int n_rounds, long long *out) {
    const long long t0 = clock64();
    unsigned int target = 0;
    for (int r = 0; r < n_rounds; ++r) {
        // some traffic to hand over: every workgroup writes a word, reads its neighbour's after the barrier
        if (threadIdx.x == 0) data[blockIdx.x] = (unsigned int)r;
        __syncthreads();
        target += gridDim.x;
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(cnt, 1u);
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
        if (threadIdx.x == 0 && data[(blockIdx.x + 1) % gridDim.x] != (unsigned int)r) out[1] = -1; // (stale read)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = clock64() - t0;
}

int main() {
    unsigned int *cnt, *data;
    long long *out, h[2];
    hipMalloc(&cnt, 4);
    hipMalloc(&data, 4096 * 4);
    hipMalloc(&out, 16);
    for (int grid : {64, 256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            int n_rounds = 200;
            hipMemset(cnt, 0, 4);
            hipMemset(out, 0, 16);
            void *args[] = {&cnt, &data, &n_rounds, &out};
            hipError_t e = hipLaunchCooperativeKernel((const void *)k_barriers, dim3(grid), dim3(256), args, 0, 0);
            if (e != hipSuccess) {
                printf("grid %d: cooperative launch refused: %s\n", grid, hipGetErrorString(e));
                break;
            }
            hipDeviceSynchronize();
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("grid %3d x 256 threads: %.2f us per grid barrier (%d rounds)%s\n", grid, h[0] / 2400.0 / n_rounds, n_rounds, h[1] ? "  STALE READ SEEN" : "");
        }
    }
    return 0;
}
