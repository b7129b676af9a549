// mlm_kernels.h — HIP kernels of the per-frame map update (gfx950, wave64): the CELL-TABLE path of Stage A and its
// two-kernel Stage B+C (frontier mode, images wider than 2040 pixels, and the fall-back of frames in which an azimuth
// sector overflows its LDS tables), the rehash-replay kernels, k_chain, queries, inflation, exports.  The default Stage A
// (by azimuth sector) and its one-launch-per-batch apply kernel are in mlm_kernels_sector.h.
//
// Stage A  (awareness_map_cylindrical::input_pc_pose, map_awareness.cpp:173-282)
//   k_bin_points      point -> (rho,phi,z) bin, noise-spread hit contributions as wave groups, de-duplicated ray walk
//   k_assign_nodes    per group: first-touch time / kind mask / count of its cell, position inside the cell
//   k_collect_hits    dense sweep of the hit scratch -> unique-hit list + per-cell contribution segments
//   k_expand_nodes / k_sort_contribs / k_chain   point-order replay of the float noisy-OR chain -> odd, logit
// Stage B  (iteration order of hit_idx_odds_hashmap, i.e. libstdc++ _Hashtable list order)
//   k_bucket_min / k_make_keys (+ rank kernels on rehash frames)
// Stage C  (local_map_cartesian::input_pc_pose_direct, map_local.cpp:143-237)
//   k_prepare_voxels (Stage A, map independent) / k_voxelize / k_apply
// Queries  (mlmap.h:142-295, mlmap.cpp:388-407)
#pragma once
#include "mlm_device.h"

#define MLM_BLOCK 256

__device__ __forceinline__ void mlm_cell_rpz(const MlmDev &P, uint32_t cell, int &rho, int &phi, int &z) {
    z = (int)(cell / (uint32_t)P.nRhoPhi);
    const int rem = (int)(cell % (uint32_t)P.nRhoPhi);
    phi = rem / P.nRho;
    rho = rem - phi * P.nRho;
}

// ---------------------------------------------------------------------------------------------------------------
// Stage A
// ---------------------------------------------------------------------------------------------------------------
// A hit contribution is identified by its insertion time t = point*21 + sub, sub = 0 (centre), 2d-1 (+d
// neighbour), 2d (-d neighbour): the order update_hits inserts them (map_awareness.cpp:146-168).  Its odd is
// get_odds_table[row(sub)][rho of the point], so (cell, sub) fixes the value.

// Split the wave into groups of lanes holding the same key and run f(key, lane mask) on each group's lowest
// lane.  Must be called by all 64 lanes.  Neighbouring pixels share awareness cells, so this turns ~64 global
// atomics on a handful of addresses into a handful of atomics.
template <class F> __device__ __forceinline__ void mlm_wave_groups(int key, bool valid, F &&f) {
    unsigned long long todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = mlm_readlane(key, leader);
        const unsigned long long m = __ballot(valid && key == k);
        if (lane == leader) f(k, m);
        todo &= ~m;
    }
}

// wave64 inclusive prefix sum with DPP moves (no LDS): Hillis-Steele inside the rows of 16 lanes, then the row totals
// are carried over with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).  Must be called by all 64 lanes.
__device__ __forceinline__ uint32_t mlm_wave_incl_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// In dense mode a wave owns an 8x8 pixel tile (lanes row-major inside the tile, so a lower lane always has the smaller
// pixel index = earlier insertion time) and the block (MlmDev::bin_block threads) a 32 x (bin_block/32) tile: the larger
// the tile, the fewer (block, cell) pairs a frame produces, and each pair costs three device-scope atomics.
struct MlmTile {
    int i;      // work item (pixel index / list position / point index), -1 = none
    bool valid;
};
template <int MODE> __device__ __forceinline__ MlmTile mlm_tile_item(const MlmFrame &F) {
    MlmTile t;
    if (MODE == 0) {
        // block = 32 pixels wide, 8 rows per 4 waves (blockDim 256: 32x8 strip, 1024: 32x32 tile)
        const int tile_h = (int)(blockDim.x >> 8) * 8;
        const int tiles_x = (F.width + 31) >> 5;
        const int by = blockIdx.x / tiles_x;
        const int bx = blockIdx.x - by * tiles_x;
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        const int px = bx * 32 + (w & 3) * 8 + (l & 7);
        const int py = by * tile_h + (w >> 2) * 8 + (l >> 3);
        t.valid = px < F.width && py < F.height;
        t.i = py * F.width + px;
    } else {
        t.i = blockIdx.x * blockDim.x + threadIdx.x;
        t.valid = t.i < F.n;
    }
    return t;
}

// ray walk, map_awareness.cpp:243-274, from the binned start (rho,phi,z) of a point (in range or not):
// slope = (z - zc)/rho, clamp to the outer border, then r = rho-1 .. 1, z' = round(z - (rho-r)*slope).
// Every lane may bring one ray start (has, rho, phi, z).  The per-ray setup (slope: an FP64 division; the clamp to the
// outer border) is done by all lanes at once; then the WAVE walks the rays one after the other: lane l takes the steps
// r = l+1, l+65, ...; lanes whose cells fall into the same 32-bit word of the miss mask (consecutive r with equal z')
// are merged so that one atomicOr per word run is issued.  Must be called by all 64 lanes.
__device__ __forceinline__ void mlm_walk_rays_wave(const MlmDev &P, bool has, int rho, int phi, int z,
                                                   uint32_t *s_wkey = nullptr, uint32_t *s_wbits = nullptr) {
    MLM_GLOBAL uint32_t *miss = mlm_gp(P.miss_bits) + (size_t)(blockIdx.x & (MLM_MISS_COPIES - 1)) * P.nMissWords;
    const int lane = threadIdx.x & 63;
    double slope = 0.0;
    int row_base = 0;
    if (has) {
        slope = (rho > 0) ? (z - P.zc) / (rho * 1.0) : 0.0;
        if (rho >= P.nRho) {
            z = mlm_cvt_int(round(z - ((rho - P.nRho + 1) * slope)));
            rho = P.nRho - 1;
        }
        row_base = phi * P.RW;
    }
    const int row_pitch = P.nPhi * P.RW;
    unsigned long long todo = __ballot(has && rho > 1);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int rho_s = mlm_readlane(rho, src), z_s = mlm_readlane(z, src), rb_s = mlm_readlane(row_base, src);
        const double slope_s = __hiloint2double(mlm_readlane(__double2hiint(slope), src), mlm_readlane(__double2loint(slope), src));
        for (int r0 = 1; r0 < rho_s; r0 += 64) {
            const int r = r0 + lane;
            int w = -1;
            if (r < rho_s) {
                const int diff_r = rho_s - r;
                const int zr = mlm_cvt_int(round(z_s - (diff_r * slope_s)));
                if (0 <= zr && zr < P.nZ) w = zr * row_pitch + rb_s + (r >> 5);
            }
            // run heads: lanes whose word differs from the previous lane's (wave_shr:1; lane 0 keeps -2)
            const int w_prev = __builtin_amdgcn_update_dpp(-2, w, 0x138, 0xf, 0xf, false);
            const bool head = w >= 0 && w_prev != w;
            const unsigned long long valid = __ballot(w >= 0);
            const unsigned long long heads = __ballot(head);
            if (head) {
                // the run ends before the next head or the next invalid lane
                const unsigned long long above = ~((2ull << lane) - 1ull); // lanes > lane
                const unsigned long long stop = (heads | ~valid) & above;
                const int end = stop ? __ffsll((long long)stop) - 1 : 64; // exclusive
                const int len = end - lane;                               // <= 32: a run stays inside one word
                const uint32_t bits = (len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u)) << (r & 31);
                bool staged = false;
                if (s_wkey) { // OR into the workgroup's LDS table (s_wkey: word index or MLM_NIL); flushed by the caller
                    uint32_t s = ((uint32_t)w * 2654435761u) >> 22; // 10 bits: MLM_BOOK_WORDS entries
                    for (int probe = 0; probe < 16 && !staged; ++probe) {
                        const uint32_t prev = atomicCAS(&s_wkey[s], MLM_NIL, (uint32_t)w);
                        if (prev == MLM_NIL || prev == (uint32_t)w) {
                            atomicOr(&s_wbits[s], bits);
                            staged = true;
                        }
                        s = (s + 1) & 1023u;
                    }
                }
                if (!staged) g_atomic_or(&miss[w], bits);
            }
        }
    }
}

// the same walk by ONE lane (rare paths: LDS buffers of k_bin_points overflowed)
__device__ __forceinline__ void mlm_walk_ray_lane(const MlmDev &P, int rho, int phi, int z) {
    MLM_GLOBAL uint32_t *miss = mlm_gp(P.miss_bits) + (size_t)(blockIdx.x & (MLM_MISS_COPIES - 1)) * P.nMissWords;
    const double slope = (rho > 0) ? (z - P.zc) / (rho * 1.0) : 0.0;
    if (rho >= P.nRho) {
        z = mlm_cvt_int(round(z - ((rho - P.nRho + 1) * slope)));
        rho = P.nRho - 1;
    }
    for (int r = 1; r < rho; ++r) {
        const int zr = mlm_cvt_int(round(z - ((rho - r) * slope)));
        if (0 <= zr && zr < P.nZ) g_atomic_or(&miss[(zr * P.nPhi + phi) * P.RW + (r >> 5)], 1u << (r & 31));
    }
}

// neighbour cells of the noise spread (update_hits, map_awareness.cpp:149-168) for step d; -1 = none
// slope = raycasting_z_over_rho of the point's cell (map_awareness.cpp:64-71): (zi - zc) / rho, 0 for rho == 0
__device__ __forceinline__ void mlm_spread_cells(const MlmDev &P, int rho, int phi, int zi, int d, double slope, int &c_plus,
                                                 int &c_minus) {
    c_plus = -1;
    c_minus = -1;
    int rz = mlm_cvt_int(round(zi + (d * slope)));
    if (0 <= rz && rz < P.nZ) c_plus = rz * P.nRhoPhi + phi * P.nRho + rho + d;
    rz = mlm_cvt_int(round(zi - (d * slope)));
    if (0 <= rz && rz < P.nZ && rho - d >= 0) c_minus = rz * P.nRhoPhi + phi * P.nRho + rho - d;
}
__device__ __forceinline__ bool mlm_spread_active(const MlmDev &P, int rho, int d, float s3) {
    return (float)d < s3 && (rho + d < P.nRho) && d <= MLM_DIFF_RANGE;
}


// MlmDev::node_lds contribution nodes are buffered per k_bin_points block; MlmDev::agg_lds (a power of two) entries of
// a block-local table merge the block's groups per awareness cell
struct MlmCellAgg {
    uint32_t cell;      // MLM_NIL = empty
    uint32_t tmin;      // earliest insertion time of the block's contributions to the cell
    uint32_t kmask;     // kinds
    uint32_t cnt;       // contributions
    uint32_t idx;       // dense index of the entry inside the block = its slot in the block's slice of MlmDev::pairs
    uint32_t start_min; // explore mode: first point whose hit centre is the cell
};

// MODE 0: dense depth image, 1: indexed depth pixels, 2: explicit sensor-frame points
// Stage A kernels are launched once per BATCH: blockIdx.z selects the frame slot (its MlmDev and MlmFrame live in
// device memory), so one launch covers all frames in flight.
#define MLM_SLOT_ARGS const MlmDev *__restrict__ slot_tab, const MlmFrame *__restrict__ frame_tab, int slot_base
#define MLM_SLOT_SETUP                                                                                                \
    const MlmDev &P = slot_tab[slot_base + blockIdx.z];                                                               \
    const MlmFrame &F = frame_tab[slot_base + blockIdx.z];                                                            \
    (void)F;

// a 24-byte node record as one 16-byte and one 8-byte global store
__device__ __forceinline__ void mlm_store_node(MLM_GLOBAL MlmNode *dst, const MlmNode &nd) {
    MLM_GLOBAL uint32_t *d = (MLM_GLOBAL uint32_t *)dst;
    d[0] = nd.cell;
    d[1] = nd.pos;
    d[2] = nd.i00_sub;
    d[3] = nd.pad;
    *(MLM_GLOBAL unsigned long long *)(d + 4) = nd.mask;
}

#define MLM_RAY_LDS 128 // rays buffered per k_bin_points block (overflow: walked on the spot)
#ifdef MLM_PHASE_PROF // diagnostic build only (tools/phase_prof.py): shader-clock cycles per phase of k_bin_points
#define MLM_PHASE_BLOCKS 32768
__device__ unsigned long long g_mlm_phase[MLM_PHASE_BLOCKS * 16]; // per block: no contended atomics in the measurement
#define MLM_PHASE(n)                                                                                                   \
    do {                                                                                                               \
        const long long now_ = clock64();                                                                              \
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_phase[n], (unsigned long long)(now_ - ph_t_));                       \
        ph_t_ = now_;                                                                                                  \
    } while (0)
#define MLM_PHASE_BEGIN                                                                                                \
    __shared__ unsigned long long s_phase[16];                                                                         \
    if (threadIdx.x < 16) s_phase[threadIdx.x] = 0;                                                                    \
    __syncthreads();                                                                                                   \
    long long ph_t_ = clock64();
#define MLM_PHASE_END                                                                                                  \
    __syncthreads();                                                                                                   \
    {                                                                                                                  \
        const unsigned int b_ = blockIdx.x + gridDim.x * blockIdx.z;                                                   \
        if (threadIdx.x < 16 && b_ < MLM_PHASE_BLOCKS) g_mlm_phase[b_ * 16 + threadIdx.x] = s_phase[threadIdx.x];      \
    }
#else
#define MLM_PHASE(n)
#define MLM_PHASE_BEGIN
#define MLM_PHASE_END
#endif
// When a kernel's workgroups start and end (tools/kernel_spans.py): first start, last end and the longest workgroup, on the
// constant 100 MHz clock — what of a lone frame's kernel time is the workgroups' own and what is launch and drain
#ifdef MLM_PHASE_PROF
__device__ unsigned long long g_mlm_span[8 * 4];
__device__ unsigned long long g_mlm_wg[2 * 2048 * 2]; // per kernel (2) and workgroup (blockIdx.x < 2048): start, end
#define MLM_SPAN_BEGIN(k)                                                                                              \
    const unsigned long long span_t0_ = wall_clock64();                                                               \
    const long long span_c0_ = clock64();                                                                              \
    if (threadIdx.x == 0) atomicMin(&g_mlm_span[(k) * 4 + 0], span_t0_);
#define MLM_SPAN_END(k)                                                                                                \
    __syncthreads();                                                                                                   \
    if (threadIdx.x == 0) {                                                                                            \
        const unsigned long long t1_ = wall_clock64();                                                                 \
        atomicMax(&g_mlm_span[(k) * 4 + 1], t1_);                                                                      \
        atomicMax(&g_mlm_span[(k) * 4 + 2], t1_ - span_t0_);                                                           \
        atomicAdd(&g_mlm_span[(k) * 4 + 3], 1ull);                                                                     \
        if (blockIdx.x == 70) g_mlm_span[28 + (k)] = (unsigned long long)(clock64() - span_c0_) * 1000ull / (t1_ - span_t0_ + 1ull); \
        if (blockIdx.x < 2048) {                                                                                       \
            g_mlm_wg[((k) * 2048 + blockIdx.x) * 2] = span_t0_;                                                        \
            g_mlm_wg[((k) * 2048 + blockIdx.x) * 2 + 1] = t1_;                                                         \
        }                                                                                                              \
    }
#else
#define MLM_SPAN_BEGIN(k)
#define MLM_SPAN_END(k)
#endif
// ... and of k_tile (tools/tile_phase.py), in an array of its own, accumulated over the launches
#ifdef MLM_PHASE_PROF
__device__ unsigned long long g_mlm_tphase[4096 * 8];
#define MLM_TPHASE(n)                                                                                                  \
    do {                                                                                                               \
        const long long now_ = clock64();                                                                              \
        if (threadIdx.x == 0) s_tphase[n] += (unsigned long long)(now_ - tph_t_);                                      \
        tph_t_ = now_;                                                                                                 \
    } while (0)
#define MLM_TPHASE_BEGIN                                                                                               \
    __shared__ unsigned long long s_tphase[8];                                                                         \
    if (threadIdx.x < 8) s_tphase[threadIdx.x] = 0;                                                                    \
    __syncthreads();                                                                                                   \
    long long tph_t_ = clock64();
#define MLM_TPHASE_END                                                                                                 \
    if (threadIdx.x == 0)                                                                                              \
        for (int i_ = 0; i_ < 8; ++i_) g_mlm_tphase[(tile & 4095u) * 8 + i_] += s_tphase[i_];
#else
#define MLM_TPHASE(n)
#define MLM_TPHASE_BEGIN
#define MLM_TPHASE_END
#endif
template <int MODE>
__global__ __launch_bounds__(1024) void k_bin_points(MLM_SLOT_ARGS) {
    MLM_SLOT_SETUP
    MLM_PHASE_BEGIN
    __shared__ unsigned int s_cnt[16];
    __shared__ unsigned int s_nray;
    __shared__ unsigned int s_nnode, s_na, s_nbase2;
    // dynamic LDS, sized by the host from the configuration (how many kinds a point can spread into):
    // [node_lds nodes][agg_lds cell aggregates][MLM_RAY_LDS rays]
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const unsigned int MLM_NODE_LDS = P.node_lds, MLM_AGG_LDS = P.agg_lds;
    MlmNode *s_node = (MlmNode *)s_dyn;
    MlmCellAgg *s_agg = (MlmCellAgg *)(s_node + MLM_NODE_LDS);
    int(*s_ray)[4] = (int(*)[4])(s_agg + MLM_AGG_LDS);
    if (threadIdx.x == 0) {
        s_nnode = 0;
        s_nray = 0;
        s_na = 0;
    }
    for (unsigned int e = threadIdx.x; e < MLM_AGG_LDS; e += blockDim.x) {
        s_agg[e].cell = MLM_NIL;
        s_agg[e].tmin = MLM_EMPTY_T;
        s_agg[e].kmask = 0;
        s_agg[e].cnt = 0;
        s_agg[e].start_min = MLM_EMPTY_T;
    }
    __syncthreads();
    MLM_PHASE(0);
    const MlmTile T = mlm_tile_item<MODE>(F);
    const int i = T.i;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bool have = T.valid;
    double xs = 0, ys = 0, zs = 0;
    if (have) {
        if (MODE == 2) {
            xs = mlm_gp(F.pts)[3 * (size_t)i + 0];
            ys = mlm_gp(F.pts)[3 * (size_t)i + 1];
            zs = mlm_gp(F.pts)[3 * (size_t)i + 2];
        } else {
            const int pix = (MODE == 1) ? mlm_gp(F.pix)[i] : i;
            const int v = pix / F.width;
            const int u = pix - v * F.width;
            const uint16_t raw = (MODE == 1 && F.raw) ? (uint16_t)mlm_gp(F.raw)[i] : mlm_gp(F.img)[(size_t)v * F.row_stride + u];
            if (raw == 0) { // mlmap.cpp:338-341
                have = false;
            } else {
                // mlmap.cpp:329,344-346: (size_t u - float cx_) is a float subtraction, the rest is double
                const double depth = raw * P.inv_factor;
                xs = ((float)u - P.cx) * depth / P.fx;
                ys = ((float)v - P.cy) * depth / P.fy;
                zs = depth;
            }
        }
    }
    int rho = 0, phi = 0, zi = 0, c0 = -1;
    bool can_do_cast = false, inside = false;
    if (have) {
        // p_l = T_ls * p_s (map_awareness.cpp:222; se3.cpp:91-95)
        double x, y, z;
        mlm_quat_rot(F.q_ls, xs, ys, zs, x, y, z);
        x = x + F.t_ls[0];
        y = y + F.t_ls[1];
        z = z + F.t_ls[2];
        inside = mlm_bin_point(P, x, y, z, rho, phi, zi, can_do_cast);
        if (inside) c0 = zi * P.nRhoPhi + phi * P.nRho + rho;
    }
    MLM_PHASE(1);
    // work item of lane 0 of this wave (see MlmNode)
    const uint32_t i00 = (uint32_t)mlm_readlane(i, 0);

    // queue a ray for phase D; if the LDS queue is full (rare) walk it on the spot / hand it to k_ex_walk_rays directly
    auto queue_ray = [&](int rh, int ph, int z, int pt) {
        const unsigned int k = atomicAdd(&s_nray, 1u);
        if (k < MLM_RAY_LDS) {
            s_ray[k][0] = rh;
            s_ray[k][1] = ph;
            s_ray[k][2] = z;
            s_ray[k][3] = pt;
        } else if (!P.explore) {
            mlm_walk_ray_lane(P, rh, ph, z);
            g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[blockIdx.x & 7][0], 1u);
        } else {
            MLM_GLOBAL int32_t *q = mlm_gp(P.ex_rays) + 4 * (size_t)g_atomic_add(&mlm_gp(P.ctr)->n_ex_rays, 1u);
            q[0] = rh;
            q[1] = ph;
            q[2] = z;
            q[3] = pt;
            g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[blockIdx.x & 7][0], 1u);
        }
    };
    // ---- phase A: one group = lanes of this wave that contribute kind `sub` to `cell`.  Its lowest lane (= earliest
    //      insertion time) records the group as a node in LDS; no global memory is touched here.
    //      The lanes are split ONCE, by centre cell (ballot/readlane peel): the +-d neighbour cells are functions of the
    //      centre cell (map_awareness.cpp:149-168), so every centre group is also a group of each spread kind and only
    //      its leader computes the neighbour cells.  (Two centre groups can land in the same neighbour cell: they stay
    //      two groups, the LDS table merges them.)
    auto post_node = [&](int cell, unsigned long long mask, int sub) { // lanes with cell >= 0 record a group
        if (cell >= 0) {
            const unsigned int k = atomicAdd(&s_nnode, 1u);
            MlmNode nd;
            nd.cell = (uint32_t)cell;
            nd.pos = 0;
            nd.i00_sub = i00 | ((uint32_t)sub << 27);
            nd.pad = 0;
            nd.mask = mask;
            if (k < MLM_NODE_LDS) {
                s_node[k] = nd;
            } else { // LDS buffer full: store the node directly; k_assign_nodes books it (and walks its ray)
                const unsigned int reg = blockIdx.x & 7;
                const unsigned int g = g_atomic_add(&mlm_gp(P.ctr)->node_cnt[reg][0], 1u);
                nd.pad = MLM_NIL;
                if (g < P.node_cap) mlm_store_node(mlm_gp(P.nodes) + ((size_t)reg * P.node_cap + g), nd);
                g_atomic_add(&mlm_gp(P.ctr)->n_unassigned, 1u);
            }
        }
    };
    unsigned long long my_mask = 0;
    bool leader = false;
    mlm_wave_groups(c0, inside, [&](int, unsigned long long m) {
        leader = true;
        my_mask = m;
    });
    post_node(leader ? c0 : -1, my_mask, 0);
    const float s3 = leader ? mlm_gp(P.sigma3)[rho] : 0.0f;
    const double slope = (leader && rho > 0) ? (zi - P.zc) / (rho * 1.0) : 0.0;
    for (int d = 1; __any(leader && mlm_spread_active(P, rho, d, s3)); ++d) {
        int cp = -1, cm = -1;
        if (leader && mlm_spread_active(P, rho, d, s3)) mlm_spread_cells(P, rho, phi, zi, d, slope, cp, cm);
        post_node(cp, my_mask, 2 * d - 1);
        post_node(cm, my_mask, 2 * d);
    }
    MLM_PHASE(2);
    // ---- points outside the map that can still cast (map_awareness.cpp:241,249-265): identical starts inside
    //      the wave are merged, across waves they are simply walked again (idempotent bit sets).  Rays of in-range
    //      points start at their hit cell; those are de-duplicated per FRAME in phase C.
    const bool outer = have && !inside && can_do_cast && P.visibility;
    bool emit_ray = false;
    {
        unsigned long long todo = __ballot(outer);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int kr = mlm_readlane(rho, leader), kp = mlm_readlane(phi, leader), kz = mlm_readlane(zi, leader);
            const unsigned long long m = __ballot(outer && rho == kr && phi == kp && zi == kz);
            if (lane == leader) emit_ray = true;
            todo &= ~m;
        }
    }
    // ---- statistics: per-block partial sums, no atomics
    const unsigned int n_pts = (unsigned int)__popcll(__ballot(have));
    const unsigned int n_oor = (unsigned int)__popcll(__ballot(have && !(can_do_cast && P.visibility)));
    if (emit_ray) queue_ray(rho, phi, zi, i);
    if (lane == 0) s_cnt[wid] = n_pts | (n_oor << 10);
    MLM_PHASE(3);
    __syncthreads();
    MLM_PHASE(4);
    // ---- phase B: merge the block's groups per cell in an LDS table, so that a cell costs three device-scope atomics
    //      per BLOCK (first-touch time min, kind mask or, count add -> position) instead of three per group.
    const unsigned int reg = blockIdx.x & 7;
    const unsigned int nn = min(s_nnode, (unsigned int)MLM_NODE_LDS);
    const int tile_w = (MODE == 0) ? F.width : 0;
    auto late_ray = [&](uint32_t cell) { // a ray found outside phase C (rare path)
        int z, ph, rh;
        mlm_cell_rpz(P, cell, rh, ph, z);
        queue_ray(rh, ph, z, -1);
    };
    for (unsigned int k = threadIdx.x; k < nn; k += blockDim.x) {
        const MlmNode nd = s_node[k];
        const int sub = (int)(nd.i00_sub >> 27);
        const int l0 = __ffsll((long long)nd.mask) - 1; // lowest lane = earliest insertion time of the group
        const uint32_t i_first = (nd.i00_sub & 0x07FFFFFFu) + (tile_w > 0 ? (uint32_t)((l0 >> 3) * tile_w + (l0 & 7)) : (uint32_t)l0);
        const uint32_t t = i_first * MLM_TIME_SLOTS + (uint32_t)sub;
        const uint32_t cnt = (uint32_t)__popcll(nd.mask);
        uint32_t e = (nd.cell * 2654435761u) >> P.agg_shift; // log2(agg_lds) bits
        bool placed = false;
        for (unsigned int probe = 0; probe < MLM_AGG_LDS; ++probe) {
            const uint32_t prev = atomicCAS(&s_agg[e].cell, MLM_NIL, nd.cell);
            if (prev == MLM_NIL) s_agg[e].idx = atomicAdd(&s_na, 1u); // claimed a new entry
            if (prev == MLM_NIL || prev == nd.cell) {
                placed = true;
                break;
            }
            e = (e + 1) & (MLM_AGG_LDS - 1);
        }
        if (placed) {
            atomicMin(&s_agg[e].tmin, t);
            atomicOr(&s_agg[e].kmask, 1u << sub);
            s_node[k].pos = atomicAdd(&s_agg[e].cnt, cnt);
            if (sub == 0) atomicMin(&s_agg[e].start_min, i_first);
            s_node[k].pad = e;
        } else { // table full (never seen): book the group on its own
            g_atomic_min(&mlm_gp(P.cs)[nd.cell].t, t);
            const uint32_t old = g_atomic_or(&mlm_gp(P.cs)[nd.cell].mask, 1u << sub);
            const uint32_t pos = g_atomic_add(&mlm_gp(P.cs)[nd.cell].cnt, cnt);
            if (P.explore && sub == 0) g_atomic_min(&mlm_gp(P.start_t)[nd.cell], i_first);
            s_node[k].pos = pos;
            s_node[k].pad = MLM_NIL;
            if (pos == 0) {
                const unsigned int g = g_atomic_add(&mlm_gp(P.ctr)->touch_cnt[reg][0], 1u);
                if (g < P.touch_cap) mlm_gp(P.touched)[(size_t)reg * P.touch_cap + g] = nd.cell;
            }
            if (sub == 0 && P.visibility && !(old & 1u)) late_ray(nd.cell);
        }
    }
    MLM_PHASE(5);
    __syncthreads();
    MLM_PHASE(6);
    // ---- phase C: write-out into the block's own slices of `bnodes` and `pairs`: plain stores, no atomics, no round
    //      trip to memory.  The (block, cell) pairs are booked on their cells by k_book_cells, where the wait for the
    //      returned counts does not hold this block's LDS and wave slots (measured: 18.8 -> see DESIGN.md).
    const unsigned int na = s_na;
    const unsigned int nr = min(s_nray, (unsigned int)MLM_RAY_LDS);
    if (threadIdx.x == 0) {
        unsigned int pts = 0, oor = 0;
        for (unsigned int w = 0; w < (blockDim.x >> 6); ++w) {
            pts += s_cnt[w] & 1023u;
            oor += s_cnt[w] >> 10;
        }
        *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.blk_stats) + 4 * (size_t)blockIdx.x) = mlm_u32x4{pts, oor, nn, na};
        if (nr) {
            g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[reg][0], nr); // statistic only
            if (P.explore) s_nbase2 = g_atomic_add(&mlm_gp(P.ctr)->n_ex_rays, nr);
        }
    }
    {
        MLM_GLOBAL MlmNode *out = mlm_gp(P.bnodes) + (size_t)blockIdx.x * MLM_NODE_LDS;
        const uint32_t pair0 = blockIdx.x * MLM_AGG_LDS;
        for (unsigned int k = threadIdx.x; k < nn; k += blockDim.x) {
            MlmNode nd = s_node[k];
            if (nd.pad != MLM_NIL) nd.pad = pair0 + s_agg[nd.pad].idx;
            mlm_store_node(out + k, nd);
        }
        MLM_GLOBAL uint32_t *pr = (MLM_GLOBAL uint32_t *)(mlm_gp(P.pairs) + (size_t)pair0);
        for (unsigned int e = threadIdx.x; e < MLM_AGG_LDS; e += blockDim.x) {
            const uint32_t cell = s_agg[e].cell;
            if (cell == MLM_NIL) continue;
            MLM_GLOBAL uint32_t *d = pr + 6 * (size_t)s_agg[e].idx;
            *(MLM_GLOBAL mlm_u32x2 *)(d + 0) = mlm_u32x2{cell, s_agg[e].tmin};
            *(MLM_GLOBAL mlm_u32x2 *)(d + 2) = mlm_u32x2{s_agg[e].kmask, s_agg[e].cnt};
            *(MLM_GLOBAL mlm_u32x2 *)(d + 4) = mlm_u32x2{s_agg[e].start_min, 0u};
        }
    }
    MLM_PHASE(9);
    // ---- the queued rays (starts outside the map, rare late rays): one ray per wave at a time
    if (nr) {
        if (!P.explore) {
            for (unsigned int r0 = wid * 64; r0 < nr; r0 += blockDim.x) {
                const unsigned int r = min(r0 + lane, nr - 1);
                mlm_walk_rays_wave(P, r0 + lane < nr, s_ray[r][0], s_ray[r][1], s_ray[r][2]);
            }
        } else {
            // frontier mode needs each miss cell's insertion time, which depends on the FIRST point of a start cell;
            // that is only known after the whole frame was binned: k_ex_walk_rays walks the queued rays
            __syncthreads(); // s_nbase2 (uniform branch)
            for (unsigned int r = threadIdx.x; r < nr; r += blockDim.x) {
                MLM_GLOBAL int32_t *q = mlm_gp(P.ex_rays) + 4 * (size_t)(s_nbase2 + r);
                q[0] = s_ray[r][0];
                q[1] = s_ray[r][1];
                q[2] = s_ray[r][2];
                q[3] = s_ray[r][3];
            }
        }
    }
    MLM_PHASE(10);
    MLM_PHASE(11);
    MLM_PHASE_END
}

// odd of the contribution kind `sub` into a cell at rho_c (see the top of this section): its place in the odds table ...
__device__ __forceinline__ int mlm_contribution_index(const MlmDev &P, int rho_c, int sub) {
    int row = MLM_DIFF_RANGE, rho_s = rho_c;
    if (sub > 0) {
        const int d = (sub + 1) >> 1;
        if (sub & 1) {
            row = MLM_DIFF_RANGE + d;
            rho_s = rho_c - d;
        } else {
            row = MLM_DIFF_RANGE - d;
            rho_s = rho_c + d;
        }
    }
    return row * P.nRho + rho_s;
}
// ... and the value
__device__ __forceinline__ float mlm_contribution_odd(const MlmDev &P, const float *table, int rho_c, int sub) {
    return table[mlm_contribution_index(P, rho_c, sub)];
}
// logit macro, map_local.h:8, on a float: log10f(x / (1 - x)) — the HOST libm's log10f in the reference.  P.logit_exact:
// mlm_create found that this host's log10f is glibc's table-driven one, restated bit for bit in mlm_glibc_log10f
// (mlm_host.h): the increment then has the reference's float bits.  Otherwise (an unknown libm): FP64 log10 rounded once,
// which differs from a float log10f in the last place on some inputs.
__device__ __forceinline__ float mlm_logit(const MlmDev &P, float p) {
    const float ratio = p / (1 - p);
    if (P.logit_exact) return mlm_glibc_log10f(ratio);
    return (float)log10((double)ratio);
}

// Book the (block, cell) pairs of k_bin_points on their cells — second level of the aggregation.  One workgroup takes
// the pairs of MLM_BOOK_GROUP neighbouring k_bin_points blocks (dense mode: a 4x4 arrangement of 32-pixel-wide tiles),
// merges them per cell in an LDS table and books each distinct cell with three device-scope atomics: first-touch time
// (min), kind mask (or), contribution count (add).  Device-scope atomics are executed at the memory side (~35 G/s for
// the whole chip, tools/probes/atomic_probe.hip) and are what bounds this path, so every level of merging pays.
//  - The returned count is the position of the group's contributions inside the cell's segment (0 = the cell's first
//    contributions of the frame: queue it for k_collect_hits); every pair gets its share through MlmPair::base.
//  - The returned mask tells whether this workgroup is the first of the frame to put a hit CENTRE into the cell — every
//    point of one (rho,phi,z) cell casts the identical ray (map_awareness.cpp:243-274), so exactly this workgroup walks
//    it.  The rays of neighbouring cells run through the same words of the miss mask (all of them converge on the
//    sensor), so they are OR-ed into an LDS table first and each distinct word costs one global atomic.
#define MLM_BOOK_GROUP 16
#define MLM_BOOK_THREADS 1024 // one wave per k_bin_points block of the group
#define MLM_BOOK_CELLS 2048 // LDS cell table (power of two)
#define MLM_BOOK_WORDS 1024 // LDS miss-word table (power of two)
struct MlmBookCell {
    uint32_t cell, tmin, kmask, cnt, start_min, base;
};
// bin block handled at position j (0..MLM_BOOK_GROUP-1) of workgroup g; tiles_x > 0: 4 x (MLM_BOOK_GROUP/4) tile arrangement
__device__ __forceinline__ int mlm_book_block(int g, int j, int tiles_x, int tiles_y, int n_bin_blocks) {
    if (tiles_x <= 0) {
        const int b = g * MLM_BOOK_GROUP + j;
        return b < n_bin_blocks ? b : -1;
    }
    const int groups_x = (tiles_x + 3) >> 2;
    const int gy = g / groups_x, gx = g - gy * groups_x;
    const int tx = gx * 4 + (j & 3), ty = gy * (MLM_BOOK_GROUP / 4) + (j >> 2);
    return (tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : -1;
}
__global__ __launch_bounds__(MLM_BOOK_THREADS) void k_book_cells(MLM_SLOT_ARGS, int tiles_x, int tiles_y, int n_bin_blocks) {
    MLM_SLOT_SETUP
    __shared__ MlmBookCell s_cell[MLM_BOOK_CELLS];
    __shared__ uint32_t s_wkey[MLM_BOOK_WORDS], s_wbits[MLM_BOOK_WORDS];
    __shared__ uint32_t s_touch[MLM_BOOK_THREADS];
    __shared__ unsigned int s_ntouch, s_tbase, s_nray;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned int reg = blockIdx.x & 7;
    for (unsigned int e = threadIdx.x; e < MLM_BOOK_CELLS; e += blockDim.x) {
        s_cell[e].cell = MLM_NIL;
        s_cell[e].tmin = MLM_EMPTY_T;
        s_cell[e].kmask = 0;
        s_cell[e].cnt = 0;
        s_cell[e].start_min = MLM_EMPTY_T;
    }
    for (unsigned int e = threadIdx.x; e < MLM_BOOK_WORDS; e += blockDim.x) {
        s_wkey[e] = MLM_NIL;
        s_wbits[e] = 0;
    }
    if (threadIdx.x == 0) {
        s_ntouch = 0;
        s_nray = 0;
    }
    __syncthreads();
    // ---- pass A: merge the pairs per cell.  A pair that finds the table full books itself (never seen).
    for (int j = wid; j < MLM_BOOK_GROUP; j += MLM_BOOK_THREADS / 64) {
        const int b = mlm_book_block((int)blockIdx.x, j, tiles_x, tiles_y, n_bin_blocks);
        if (b < 0) continue;
        const unsigned int na = min(mlm_gp(P.blk_stats)[4 * (size_t)b + 3], P.agg_lds);
        for (unsigned int e = lane; e < na; e += 64) {
            MLM_GLOBAL uint32_t *d = (MLM_GLOBAL uint32_t *)(mlm_gp(P.pairs) + ((size_t)b * P.agg_lds + e));
            const mlm_u32x2 a = *(MLM_GLOBAL mlm_u32x2 *)(d + 0), m = *(MLM_GLOBAL mlm_u32x2 *)(d + 2);
            const uint32_t cell = a.x, start_min = d[4];
            uint32_t s = (cell * 2654435761u) >> (32 - 11);
            bool placed = false;
            for (int probe = 0; probe < 64; ++probe) {
                const uint32_t prev = atomicCAS(&s_cell[s].cell, MLM_NIL, cell);
                if (prev == MLM_NIL || prev == cell) {
                    placed = true;
                    break;
                }
                s = (s + 1) & (MLM_BOOK_CELLS - 1);
            }
            if (placed) {
                atomicMin(&s_cell[s].tmin, a.y);
                atomicOr(&s_cell[s].kmask, m.x);
                d[5] = atomicAdd(&s_cell[s].cnt, m.y); // offset inside the group's share; pass C adds the group's base
                if (start_min != MLM_EMPTY_T) atomicMin(&s_cell[s].start_min, start_min);
            } else {
                g_atomic_min(&mlm_gp(P.cs)[cell].t, a.y);
                const uint32_t old = g_atomic_or(&mlm_gp(P.cs)[cell].mask, m.x);
                const uint32_t base = g_atomic_add(&mlm_gp(P.cs)[cell].cnt, m.y);
                if (P.explore && start_min != MLM_EMPTY_T) g_atomic_min(&mlm_gp(P.start_t)[cell], start_min);
                d[5] = base;
                d[0] = MLM_NIL; // pass C: final
                if (base == 0) {
                    const unsigned int g = g_atomic_add(&mlm_gp(P.ctr)->touch_cnt[reg][0], 1u);
                    if (g < P.touch_cap) mlm_gp(P.touched)[(size_t)reg * P.touch_cap + g] = cell;
                }
                if (P.visibility && (m.x & 1u) && !(old & 1u)) {
                    int z, ph, rh;
                    mlm_cell_rpz(P, cell, rh, ph, z);
                    g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[reg][0], 1u);
                    if (!P.explore) {
                        mlm_walk_ray_lane(P, rh, ph, z);
                    } else {
                        MLM_GLOBAL int32_t *q = mlm_gp(P.ex_rays) + 4 * (size_t)g_atomic_add(&mlm_gp(P.ctr)->n_ex_rays, 1u);
                        q[0] = rh;
                        q[1] = ph;
                        q[2] = z;
                        q[3] = -1;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- pass B: the round trip to memory.  A lane owns MLM_BOOK_CELLS / MLM_BOOK_THREADS table entries; the atomics of all of
    //      them are issued before the first returned value is used (one round trip, not one per entry).  Then the rays of
    //      the cells this workgroup touched first with a hit centre are OR-ed into the LDS word table.
    const unsigned long long below = (1ull << lane) - 1ull;
    constexpr int PER = MLM_BOOK_CELLS / MLM_BOOK_THREADS;
    uint32_t b_cell[PER], b_old[PER], b_base[PER], b_kmask[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned int e = i * MLM_BOOK_THREADS + threadIdx.x;
        b_cell[i] = s_cell[e].cell;
        b_kmask[i] = s_cell[e].kmask;
        b_old[i] = 0;
        b_base[i] = 1;
        if (b_cell[i] != MLM_NIL) {
            g_atomic_min(&mlm_gp(P.cs)[b_cell[i]].t, s_cell[e].tmin);
            b_old[i] = g_atomic_or(&mlm_gp(P.cs)[b_cell[i]].mask, b_kmask[i]);
            b_base[i] = g_atomic_add(&mlm_gp(P.cs)[b_cell[i]].cnt, s_cell[e].cnt);
            if (P.explore && s_cell[e].start_min != MLM_EMPTY_T)
                g_atomic_min(&mlm_gp(P.start_t)[b_cell[i]], s_cell[e].start_min);
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned int e = i * MLM_BOOK_THREADS + threadIdx.x;
        const uint32_t cell = b_cell[i];
        const bool valid = cell != MLM_NIL;
        if (valid) {
            s_cell[e].base = b_base[i];
            if (b_base[i] == 0) { // the cell's first contributions of the frame
                const unsigned int k = atomicAdd(&s_ntouch, 1u);
                if (k < MLM_BOOK_THREADS) {
                    s_touch[k] = cell;
                } else { // more than the staging buffer holds: append directly
                    const unsigned int g = g_atomic_add(&mlm_gp(P.ctr)->touch_cnt[reg][0], 1u);
                    if (g < P.touch_cap) mlm_gp(P.touched)[(size_t)reg * P.touch_cap + g] = cell;
                }
            }
        }
        const bool cast = valid && P.visibility && (b_kmask[i] & 1u) && !(b_old[i] & 1u);
        const unsigned long long bc = __ballot(cast);
        if (!bc) continue;
        if (lane == 0) atomicAdd(&s_nray, (unsigned int)__popcll(bc));
        int z = 0, ph = 0, rh = 0;
        if (cast) mlm_cell_rpz(P, cell, rh, ph, z);
        if (!P.explore) {
            mlm_walk_rays_wave(P, cast, rh, ph, z, s_wkey, s_wbits);
        } else {
            // frontier mode: k_ex_walk_rays walks the queued rays once every cell's first point is known
            unsigned int ray_base = 0;
            if (lane == 0) ray_base = g_atomic_add(&mlm_gp(P.ctr)->n_ex_rays, (unsigned int)__popcll(bc));
            ray_base = mlm_readlane(ray_base, 0);
            if (cast) {
                MLM_GLOBAL int32_t *q = mlm_gp(P.ex_rays) + 4 * (size_t)(ray_base + (unsigned int)__popcll(bc & below));
                q[0] = rh;
                q[1] = ph;
                q[2] = z;
                q[3] = -1; // in-range start: take the cell's first point
            }
        }
    }
    __syncthreads();
    // ---- pass C: reserve the first-touch slots (second round trip, hidden behind the fix-up of the pairs and the
    //      flush of the miss words), hand every pair its final base
    const unsigned int nt = min(s_ntouch, (unsigned int)MLM_BOOK_THREADS);
    unsigned int touch_base = 0;
    if (threadIdx.x == 0) {
        if (nt) touch_base = g_atomic_add(&mlm_gp(P.ctr)->touch_cnt[reg][0], nt);
        if (s_nray) g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[reg][0], s_nray); // statistic only
    }
    for (int j = wid; j < MLM_BOOK_GROUP; j += MLM_BOOK_THREADS / 64) {
        const int b = mlm_book_block((int)blockIdx.x, j, tiles_x, tiles_y, n_bin_blocks);
        if (b < 0) continue;
        const unsigned int na = min(mlm_gp(P.blk_stats)[4 * (size_t)b + 3], P.agg_lds);
        for (unsigned int e = lane; e < na; e += 64) {
            MLM_GLOBAL uint32_t *d = (MLM_GLOBAL uint32_t *)(mlm_gp(P.pairs) + ((size_t)b * P.agg_lds + e));
            const uint32_t cell = d[0];
            if (cell == MLM_NIL) continue; // booked itself in pass A
            uint32_t s = (cell * 2654435761u) >> (32 - 11);
            while (s_cell[s].cell != cell) s = (s + 1) & (MLM_BOOK_CELLS - 1);
            d[5] += s_cell[s].base;
        }
    }
    for (unsigned int e = threadIdx.x; e < MLM_BOOK_WORDS; e += blockDim.x) {
        const uint32_t w = s_wkey[e];
        if (w != MLM_NIL) g_atomic_or(&mlm_gp(P.miss_bits)[(size_t)(blockIdx.x & (MLM_MISS_COPIES - 1)) * P.nMissWords + w], s_wbits[e]);
    }
    if (nt) {
        if (threadIdx.x == 0) s_tbase = touch_base;
        __syncthreads();
        for (unsigned int k = threadIdx.x; k < nt; k += blockDim.x)
            if (s_tbase + k < P.touch_cap) mlm_gp(P.touched)[(size_t)reg * P.touch_cap + s_tbase + k] = s_touch[k];
    }
}

// Groups that overflowed a k_bin_points block's LDS buffer (flag pad == 1; normally none): book each on its cell —
// first-touch time (min), kind mask (or), count (add -> position); the one that finds the count at 0 queues the cell
// for k_collect_hits.  gridDim.y = node region, blockIdx.z = slot.
__global__ __launch_bounds__(MLM_BLOCK) void k_assign_nodes(MLM_SLOT_ARGS, int tile_w) {
    MLM_SLOT_SETUP
    __shared__ unsigned int s_cnt[MLM_BLOCK / 64];
    __shared__ unsigned int s_base;
    const unsigned int reg = blockIdx.y;
    if (P.ctr->n_unassigned == 0) return; // the common case: every group was booked by its block
    const unsigned int n = min(P.ctr->node_cnt[reg][0], P.node_cap);
    for (unsigned int k0 = blockIdx.x * blockDim.x; k0 < n; k0 += gridDim.x * blockDim.x) { // uniform per block
        const unsigned int k = k0 + threadIdx.x;
        bool first = false;
        uint32_t cell = 0;
        if (k < n) {
            MlmNode *nd = &P.nodes[(size_t)reg * P.node_cap + k];
            nd->pad = MLM_NIL; // its position is absolute
            cell = nd->cell;
            const unsigned long long m = nd->mask;
            const int l0 = __ffsll((long long)m) - 1; // lowest lane = earliest insertion time of the group
            const uint32_t off = tile_w > 0 ? (uint32_t)((l0 >> 3) * tile_w + (l0 & 7)) : (uint32_t)l0;
            const uint32_t is = nd->i00_sub;
            const uint32_t t = ((is & 0x07FFFFFFu) + off) * MLM_TIME_SLOTS + (is >> 27);
            atomicMin(&P.cs[cell].t, t);
            const uint32_t old = atomicOr(&P.cs[cell].mask, 1u << (is >> 27));
            if (P.explore && (is >> 27) == 0) atomicMin(&P.start_t[cell], (is & 0x07FFFFFFu) + off);
            const uint32_t pos = atomicAdd(&P.cs[cell].cnt, (unsigned int)__popcll(m));
            nd->pos = pos;
            first = pos == 0;
            if ((is >> 27) == 0 && P.visibility && !(old & 1u)) { // first hit centre of the frame in this cell: its ray
                int z, ph, rh;
                mlm_cell_rpz(P, cell, rh, ph, z);
                atomicAdd(&P.ctr->ray_cnt[reg][0], 1u);
                if (!P.explore) {
                    mlm_walk_ray_lane(P, rh, ph, z);
                } else {
                    int32_t *q = P.ex_rays + 4 * (size_t)atomicAdd(&P.ctr->n_ex_rays, 1u);
                    q[0] = rh;
                    q[1] = ph;
                    q[2] = z;
                    q[3] = -1;
                }
            }
        }
        const unsigned int at = mlm_block_append(P.ctr->touch_cnt, first, s_cnt, &s_base);
        if (first && at < P.touch_cap) P.touched[(size_t)(blockIdx.x & 7) * P.touch_cap + at] = cell;
        __syncthreads();
    }
}

// One thread per first-touched cell (queued by k_assign_nodes): build the compact unique-hit list
//  - cell, first-touch time;
//  - cells with a single kind of contribution: odd (n applications of one value commute) and its logit;
//  - cells with several kinds: a segment of `contrib` for the point-order replay (k_sort_contribs / k_chain);
// and reset hit_t / hit_mask / hit_cnt for the next frame.  gridDim.y = sub-list.
__global__ __launch_bounds__(MLM_BLOCK) void k_collect_hits(MLM_SLOT_ARGS, int n_stat_blocks) {
    MLM_SLOT_SETUP
    __shared__ uint32_t s_w[3][4];
    __shared__ uint32_t s_base[3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (blockIdx.x == 0 && blockIdx.y == 0 && wid == 0) { // fold the per-block statistics of k_bin_points
        unsigned int a = 0, b = 0, g = 0;
        for (int j = lane; j < n_stat_blocks; j += 64) {
            a += P.blk_stats[4 * j];
            b += P.blk_stats[4 * j + 1];
            g += P.blk_stats[4 * j + 2];
        }
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            b += __shfl_xor(b, off, 64);
            g += __shfl_xor(g, off, 64);
        }
        if (lane == 0) {
            P.ctr->n_points = a;
            P.ctr->n_oor = b;
            P.ctr->n_groups = g;
        }
    }
    const unsigned int k = blockIdx.y;
    const unsigned int n = min(P.ctr->touch_cnt[k][0], P.touch_cap);
    for (unsigned int r0 = blockIdx.x * blockDim.x; r0 < n; r0 += gridDim.x * blockDim.x) { // uniform per block
        const unsigned int r = r0 + threadIdx.x;
        const bool has = r < n;
        uint32_t c = 0, t = 0, mask = 0, cnt = 0;
        if (has) {
            c = P.touched[(size_t)k * P.touch_cap + r];
            const uint4 st = *(const uint4 *)&P.cs[c]; // t, cnt, mask, seg in one 16-byte load
            t = st.x;
            cnt = st.y;
            mask = st.z;
        }
        const bool multi = has && __popc(mask) > 1;
        const unsigned long long bh = __ballot(has), bm = __ballot(multi);
        const uint32_t cpad = multi ? ((cnt + 15u) & ~15u) : 0u; // segments start 16-byte aligned in `subs`
        const uint32_t cincl = mlm_wave_incl_scan(cpad);
        if (lane == 63) {
            s_w[0][wid] = (uint32_t)__popcll(bh);
            s_w[1][wid] = (uint32_t)__popcll(bm);
            s_w[2][wid] = cincl;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            uint32_t tot = 0;
            for (int w = 0; w < 4; ++w) tot += s_w[threadIdx.x][w];
            unsigned int *ctr = threadIdx.x == 0 ? &P.ctr->u_hit : (threadIdx.x == 1 ? &P.ctr->n_multi : &P.ctr->n_contrib);
            s_base[threadIdx.x] = tot ? atomicAdd(ctr, tot) : 0u;
        }
        __syncthreads();
        uint32_t off_h = s_base[0], off_m = s_base[1], off_c = s_base[2];
        for (int w = 0; w < wid; ++w) {
            off_h += s_w[0][w];
            off_m += s_w[1][w];
            off_c += s_w[2][w];
        }
        if (has) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t pos = off_h + (uint32_t)__popcll(bh & below);
            P.cs[c].t = MLM_EMPTY_T;
            P.cs[c].mask = 0;
            P.cs[c].cnt = 0;
            if ((mask & 1u) && P.explore) P.start_t[c] = MLM_EMPTY_T; // (k_ex_walk_rays runs before this kernel)
            P.hl_cell[pos] = c;
            P.hl_t[pos] = t;
            P.hl_vt[pos] = t;
            if (multi) {
                const uint32_t base = off_c + cincl - cpad;
                P.cs[c].seg = base;
                P.hl_base[pos] = base;
                P.hl_cnt[pos] = cnt;
                P.mt_list[off_m + (uint32_t)__popcll(bm & below)] = pos;
                P.mt_rec[off_m + (uint32_t)__popcll(bm & below)] = make_uint4(pos, base, cnt, t);
                if (cnt > 1024u) P.mt_big[atomicAdd(&P.ctr->n_big, 1u)] = pos; // few: second k_sort_contribs launch
            } else {
                // cnt applications of one value (update_odds_hashmap, map_awareness.h:147-154); 1.0f is absorbing
                P.cs[c].seg = MLM_NIL;
                const int rho_c = (int)(c % (uint32_t)P.nRho);
                const float a = mlm_contribution_odd(P, P.odds_table, rho_c, __ffs((int)mask) - 1);
                float p = a;
                for (uint32_t j = 1; j < cnt && p != 1.0f; ++j) p = 1 - (1 - p) * (1 - a);
                P.hl_odd[pos] = p;
                P.hl_inc[pos] = mlm_logit(P, p);
                P.hl_cnt[pos] = 0;
            }
        }
        __syncthreads();
    }
}

// Write the insertion times of every group that feeds a multi-kind cell into the cell's segment of `contrib`.
// A wave fetches 64 groups at once (one per lane, coalesced) and then expands them one after the other (lane = lane of
// the group's mask).  tile_w > 0: dense 8x8 tiles of an image of that width; 0: linear work items.
// blockIdx.x < n_bin_blocks: the groups k_bin_points block blockIdx.x left in its slice of `bnodes` (their positions are
// relative to the pair's base); the blocks beyond take the overflow list (normally empty), region by region.
__global__ __launch_bounds__(MLM_BLOCK) void k_expand_nodes(MLM_SLOT_ARGS, int tile_w, int n_bin_blocks) {
    MLM_SLOT_SETUP
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const MlmNode *list;
    unsigned int n, k_first, k_step;
    if ((int)blockIdx.x < n_bin_blocks) {
        list = P.bnodes + (size_t)blockIdx.x * P.node_lds;
        n = min(P.blk_stats[4 * (size_t)blockIdx.x + 2], P.node_lds);
        k_first = wid * 64;
        k_step = blockDim.x;
    } else {
        if (P.ctr->n_unassigned == 0) return;
        const unsigned int ob = blockIdx.x - n_bin_blocks, reg = ob & 7, chunk = ob >> 3, n_chunks = (gridDim.x - n_bin_blocks) >> 3;
        list = P.nodes + (size_t)reg * P.node_cap;
        n = min(P.ctr->node_cnt[reg][0], P.node_cap);
        k_first = (chunk * (blockDim.x >> 6) + wid) * 64;
        k_step = n_chunks * blockDim.x;
    }
    // eight lanes per group, one per row of its 8x8 lane mask: a lane writes the (at most eight) insertion times of its row
    const int sub_lane = lane & 7, grp = lane >> 3;
    const uint32_t row_off = tile_w > 0 ? (uint32_t)(sub_lane * tile_w) : (uint32_t)(sub_lane * 8);
    for (unsigned int k0 = k_first; k0 < n; k0 += k_step) {
        MlmNode nd{};
        uint32_t base = MLM_NIL;
        if (k0 + lane < n) {
            nd = list[k0 + lane];
            base = P.cs[nd.cell].seg;
            if (base != MLM_NIL) {
                base += nd.pos;
                if (nd.pad != MLM_NIL) base += P.pairs[nd.pad].base;
            }
        }
        if (!__any(base != MLM_NIL)) continue;
#pragma unroll 1
        for (int pass = 0; pass < 8; ++pass) {
            const int src = pass * 8 + grp; // the group this lane helps with
            const uint32_t b = __shfl(base, src, 64);
            if (!__any(b != MLM_NIL)) continue;
            const uint32_t is = __shfl(nd.i00_sub, src, 64);
            const uint32_t mlo = __shfl((uint32_t)nd.mask, src, 64), mhi = __shfl((uint32_t)(nd.mask >> 32), src, 64);
            if (b == MLM_NIL) continue;
            const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
            uint32_t bits = (uint32_t)(m >> (8 * sub_lane)) & 0xFFu;          // this lane's row
            uint32_t at = b + (uint32_t)__popcll(m & ((1ull << (8 * sub_lane)) - 1ull)); // contributions of the rows before
            const uint32_t key0 = ((is & 0x07FFFFFFu) + row_off) * MLM_TIME_SLOTS + (is >> 27);
            while (bits) {
                const int c = __ffs((int)bits) - 1;
                bits &= bits - 1;
                if (at < P.contrib_cap) P.contrib[at] = key0 + (uint32_t)c * MLM_TIME_SLOTS;
                ++at;
            }
        }
    }
}

// One wave per multi-kind hit cell: order the cell's contributions by insertion time and store their kinds (`sub`) in
// that order.  A work item (pixel) contributes to one cell at most once (centre and +-d neighbours differ in rho), so
// the order is the order of the pixels.
//  - Bitmap path (the bulk): the pixels of one cell lie in a small image window.  The wave marks them in a
//    128-column x 128-row bitmap in LDS anchored at the cell's first pixel (rows of the image = rows of the bitmap; for
//    the list modes the work-item index is folded into rows of `row_w` = 64); the rank of a contribution is the number
//    of marked pixels before its own — a prefix sum over the bitmap words plus one popcount.  O(n) per cell.
//  - Fallbacks when a contribution falls outside the window: rank by counting (n <= 320), bitonic sort in LDS, or
//    counting straight from memory beyond the LDS window.
// Launched twice per batch: CAP = 1024 (4 KB of LDS per wave, full occupancy) takes the cells with n <= 1024, CAP = 4096
// the few larger ones (n_lo = 1024).  div_m / div_s: exact division of a work-item index (< 2^27) by row_w as
// (i * div_m) >> div_s.
#define MLM_BMP_ROWS 128
template <int MLM_SORT_CAP>
__global__ __launch_bounds__(MLM_BLOCK) void k_sort_contribs(MLM_SLOT_ARGS, unsigned int n_lo, int row_w,
                                                             unsigned long long div_m, int div_s) {
    MLM_SLOT_SETUP
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[MLM_BLOCK / 64][MLM_SORT_CAP];
    const bool big = MLM_SORT_CAP > 1024;
    const unsigned int n_cells = big ? P.ctr->n_big : P.ctr->n_multi;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    volatile uint32_t *K = s_keys[wid];
    // One cell ahead: the cell's record {hit-list position, segment base, contribution count, first-touch time} (written
    // by k_collect_hits) and its first 256 keys are loaded while the previous cell is processed — a cell is otherwise a
    // chain of four dependent memory round trips.
    for (int j = lane; j < 2 * MLM_BMP_ROWS; j += 64) ((unsigned long long *)s_keys[wid])[j] = 0ull; // ranking bitmap
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto load_rec = [&](unsigned int w) -> uint4 {
        if (!big) return P.mt_rec[w];
        const uint32_t pos = P.mt_big[w];
        return make_uint4(pos, P.hl_base[pos], P.hl_cnt[pos], P.hl_t[pos]);
    };
    auto load_keys = [&](const uint4 &r, uint32_t (&k)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = (uint32_t)lane + 64u * q;
            k[q] = (j < r.z) ? P.contrib[r.y + j] : 0u;
        }
    };
    auto process = [&](const uint4 &rec, const uint32_t (&kreg)[4]) {
        const uint32_t pos = rec.x, base = rec.y, n = rec.z;
        (void)pos;
        if (n <= n_lo || (n > MLM_SORT_CAP && !big)) return; // the other launch's cells
        if (n <= MLM_SORT_CAP) {
            // ---- bitmap path.  LDS window of the wave: [0,2K) 256 x u64 bitmap words (row-major, 2 per row),
            //      [2K,3K) exclusive prefix of their popcounts, [3K,..) the ordered kinds
            unsigned long long *rows = (unsigned long long *)s_keys[wid];
            volatile uint32_t *pre = (volatile uint32_t *)(s_keys[wid] + 512);
            volatile uint8_t *S = (volatile uint8_t *)(s_keys[wid] + 768);
            const uint32_t pix0 = rec.w / MLM_TIME_SLOTS; // the cell's first pixel: smallest row of the window
            const uint32_t y0 = (uint32_t)(((unsigned long long)pix0 * div_m) >> div_s);
            const int xlo = (int)(pix0 - y0 * (uint32_t)row_w) - 64;
            // (the bitmap is all zero here: cleared at kernel start and after every use)
            bool bad = false;
            // the keys of the first four rounds are in registers (loaded one cell ahead; covers n <= 256: the bulk)
            auto mark = [&](uint32_t key) {
                const uint32_t pix = key / MLM_TIME_SLOTS;
                const uint32_t y = (uint32_t)(((unsigned long long)pix * div_m) >> div_s);
                const int dx = (int)(pix - y * (uint32_t)row_w) - xlo;
                const uint32_t dy = y - y0;
                if (dx < 0 || dx >= 128 || dy >= MLM_BMP_ROWS) {
                    bad = true;
                } else {
                    atomicOr(&rows[2 * dy + ((uint32_t)dx >> 6)], 1ull << (dx & 63));
                }
            };
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = (uint32_t)lane + 64u * q;
                if (j < n) mark(kreg[q]);
            }
            for (uint32_t j = lane + 256u; j < n; j += 64) mark(P.contrib[base + j]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t carry = 0;
            int used = 2 * MLM_BMP_ROWS; // bitmap words to clear afterwards
            if (!__any(bad)) {
                // the window is anchored at the cell's first row, so the marked words come first: stop once all n
                // contributions are accounted for (typically after the first 64 words)
                int j0 = 0;
                for (; j0 < 2 * MLM_BMP_ROWS && carry < n; j0 += 64) {
                    const uint32_t c = (uint32_t)__popcll(((volatile unsigned long long *)rows)[j0 + lane]);
                    const uint32_t incl = mlm_wave_incl_scan(c);
                    pre[j0 + lane] = carry + incl - c;
                    carry += mlm_readlane(incl, 63);
                }
                if (carry == n) used = j0;
            }
            if (carry == n) { // every contribution marked its own pixel
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                auto place = [&](uint32_t key) {
                    const uint32_t pix = key / MLM_TIME_SLOTS;
                    const uint32_t y = (uint32_t)(((unsigned long long)pix * div_m) >> div_s);
                    const uint32_t dx = (uint32_t)((int)(pix - y * (uint32_t)row_w) - xlo);
                    const uint32_t wi = 2 * (y - y0) + (dx >> 6);
                    const uint32_t r = pre[wi] + (uint32_t)__popcll(((volatile unsigned long long *)rows)[wi] & ((1ull << (dx & 63)) - 1ull));
                    const uint8_t sub = (uint8_t)(key - pix * MLM_TIME_SLOTS);
                    if (MLM_SORT_CAP <= 1024) S[r] = sub;
                    else P.subs[base + r] = sub; // large cells: straight to memory
                };
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((uint32_t)lane + 64u * q < n) place(kreg[q]);
                for (uint32_t j = lane + 256u; j < n; j += 64) place(P.contrib[base + j]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int j = lane; j < used; j += 64) rows[j] = 0ull; // clean for the next cell
                if (MLM_SORT_CAP <= 1024) {
                    const uint32_t n4 = (n + 3u) & ~3u;
                    for (uint32_t j = lane; j < n4 / 4; j += 64)
                        ((uint32_t *)(P.subs + base))[j] = ((volatile uint32_t *)(s_keys[wid] + 768))[j];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                return;
            }
        }
        if (n <= 320) {
            // small cells (the bulk): rank by counting — every lane counts the keys below its own with 16-byte
            // broadcast reads; the ordered kinds are staged in the unused upper part of the wave's LDS window
            const uint32_t n4 = (n + 3u) & ~3u;
            for (uint32_t j = lane; j < n4; j += 64) K[j] = j < n ? P.contrib[base + j] : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // LDS ops of one wave execute in order
            __builtin_amdgcn_wave_barrier();
            volatile uint8_t *S = (volatile uint8_t *)(s_keys[wid] + 640);
            for (uint32_t j = lane; j < n; j += 64) {
                const uint32_t my = K[j];
                uint32_t r = 0;
                for (uint32_t q = 0; q < n4; q += 4) {
                    const uint4 v = *(const uint4 *)(const_cast<uint32_t *>(&s_keys[wid][q]));
                    r += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
                }
                S[r] = (uint8_t)(my % MLM_TIME_SLOTS);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < n4 / 4; j += 64)
                ((uint32_t *)(P.subs + base))[j] = ((volatile uint32_t *)(s_keys[wid] + 640))[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if (n <= MLM_SORT_CAP) {
            // bitonic sort of the keys in LDS by one wave (padding keys 0xFFFFFFFF sink to the end)
            uint32_t N = 64;
            while (N < n) N <<= 1;
            for (uint32_t j = lane; j < N; j += 64) K[j] = j < n ? P.contrib[base + j] : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // LDS ops of one wave execute in order
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = 2; k <= N; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = lane; t < (N >> 1); t += 64) {
                        const uint32_t i = 2u * t - (t & (j - 1u));
                        const uint32_t l = i + j;
                        const uint32_t a = K[i], b = K[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) {
                            K[i] = b;
                            K[l] = a;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // kinds in insertion-time order, 4 per lane per step, written as whole dwords (the segment is 16-byte
            // aligned and padded to 16)
            const uint32_t n4 = (n + 3u) & ~3u;
            for (uint32_t j = lane; j < n4 / 4; j += 64) {
                uint32_t w4 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t key = K[4 * j + q]; // beyond n: padding, never read back as a kind
                    w4 |= (key % MLM_TIME_SLOTS) << (8 * q);
                }
                ((uint32_t *)(P.subs + base))[j] = w4;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            // larger than the LDS window: same rank-by-counting straight from memory (rare, slow, exact)
            for (uint32_t j = lane; j < n; j += 64) {
                const uint32_t my = P.contrib[base + j];
                uint32_t r = 0;
                for (uint32_t q = 0; q < n; ++q) r += P.contrib[base + q] < my;
                P.subs[base + r] = (uint8_t)(my % MLM_TIME_SLOTS);
            }
        }
        // only the fallbacks get here, and they used the wave's LDS window for keys: clear the ranking bitmap again
        for (int j = lane; j < 2 * MLM_BMP_ROWS; j += 64) ((unsigned long long *)s_keys[wid])[j] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    uint4 rec_cur = make_uint4(0, 0, 0, 0), rec_nxt = make_uint4(0, 0, 0, 0);
    uint32_t k_cur[4] = {0, 0, 0, 0}, k_nxt[4] = {0, 0, 0, 0};
    if (wave < n_cells) {
        rec_cur = load_rec(wave);
        load_keys(rec_cur, k_cur);
    }
    if (wave + n_waves < n_cells) rec_nxt = load_rec(wave + n_waves);
    for (unsigned int w = wave; w < n_cells; w += n_waves) {
        uint4 rec_nn = make_uint4(0, 0, 0, 0);
        if (w + n_waves < n_cells) load_keys(rec_nxt, k_nxt);
        if (w + 2 * n_waves < n_cells) rec_nn = load_rec(w + 2 * n_waves);
        process(rec_cur, k_cur);
        rec_cur = rec_nxt;
        rec_nxt = rec_nn;
#pragma unroll
        for (int q = 0; q < 4; ++q) k_cur[q] = k_nxt[q];
    }
}

// One lane per multi-kind hit cell: replay update_odds_hashmap (map_awareness.h:147-154) over the ordered kinds —
// the float noisy-OR chain is not associative, so the order is part of the result.  p == 1.0f is absorbing
// (1-(1-1)(1-a) == 1), which ends long chains early.
__global__ __launch_bounds__(MLM_BLOCK) void k_chain(MLM_SLOT_ARGS, unsigned int rehash_threshold) {
    MLM_SLOT_SETUP
    const int frame_idx = F.seq;
    extern __shared__ float s_table[]; // get_odds_table, 21*nRho floats
    for (int j = threadIdx.x; j < (2 * MLM_DIFF_RANGE + 1) * P.nRho; j += blockDim.x) s_table[j] = P.odds_table[j];
    __syncthreads();
    // Stage B is launched assuming that this frame's unique hit cells fit the emulated container without a rehash
    // (element count <= _M_next_resize).  If they do not, flag the frame: its Stage B/C kernels (and those of later
    // frames) turn into no-ops and the host replays them with the exact rehash plan.
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.ctr->u_hit > rehash_threshold) atomicMin(&P.g->fail_frame, frame_idx);
    const unsigned int n_cells = P.ctr->n_multi;
    for (unsigned int w = blockIdx.x * blockDim.x + threadIdx.x; w < n_cells; w += gridDim.x * blockDim.x) {
        const uint32_t pos = P.mt_list[w];
        const uint32_t base = P.hl_base[pos]; // multiple of 16
        const uint32_t n = P.hl_cnt[pos];
        const int rho_c = (int)(P.hl_cell[pos] % (uint32_t)P.nRho);
        float p = 0.0f;
        bool first = true;
        for (uint32_t j0 = 0; j0 < n && p != 1.0f; j0 += 16) {
            const uint4 v = *(const uint4 *)(P.subs + base + j0); // 16 kinds
            const uint32_t word[4] = {v.x, v.y, v.z, v.w};
            float a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int sub = (int)((word[q >> 2] >> ((q & 3) * 8)) & 0xFFu);
                if (j0 + q >= n) sub = 0; // padding bytes are not kinds
                a[q] = mlm_contribution_odd(P, s_table, rho_c, sub);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (j0 + q < n) {
                    if (first) {
                        p = a[q];
                        first = false;
                    } else {
                        p = 1 - (1 - p) * (1 - a[q]);
                    }
                }
            }
        }
        P.hl_odd[pos] = p;
        P.hl_inc[pos] = mlm_logit(P, p);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B — iteration order of std::unordered_map<Vec3I,float,VectorHasher> (libstdc++ _Hashtable):
// the node list is a sequence of bucket chains; a bucket that becomes non-empty later sits nearer the head, and
// inside a chain later insertions sit nearer the head.  So "x is visited before y" <=> (first-insert time of x's
// bucket, insert time of x) > (… y) lexicographically.
// ---------------------------------------------------------------------------------------------------------------
#define MLM_SKIP_IF_FAILED(P, frame_idx)                                                                              \
    if (__hip_atomic_load(&(P).g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (frame_idx)) return;
__global__ __launch_bounds__(MLM_BLOCK) void k_bucket_min(const MlmDev P, int frame_idx, unsigned long long n_bkt,
                                                          unsigned int arr_limit, int use_arr) {
    MLM_SKIP_IF_FAILED(P, frame_idx)
    const unsigned int n = P.ctr->u_hit;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (use_arr && P.hl_arr[i] >= arr_limit) continue;
        int rho, phi, z;
        mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
        const unsigned long long b = mlm_hash_rpz(rho, phi, z) % n_bkt;
        atomicMin(&P.bkt_first[b], P.hl_vt[i]);
    }
}
// final = 1: hl_key = (bucket_first<<32)|vt for every element.
// final = 0: sort key for the re-densify pass: ~key for members (ascending sort = list order), all-ones otherwise.
__global__ __launch_bounds__(MLM_BLOCK) void k_make_keys(const MlmDev P, int frame_idx, unsigned long long n_bkt,
                                                         unsigned int arr_limit, int use_arr, int final_pass,
                                                         unsigned long long *sort_keys, uint32_t *sort_vals) {
    MLM_SKIP_IF_FAILED(P, frame_idx)
    const unsigned int n = P.ctr->u_hit;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool member = !(use_arr && P.hl_arr[i] >= arr_limit);
    unsigned long long key = 0;
    if (member) {
        int rho, phi, z;
        mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
        const unsigned long long b = mlm_hash_rpz(rho, phi, z) % n_bkt;
        key = ((unsigned long long)(P.bkt_first[b] + 1u) << 32) | (unsigned long long)P.hl_vt[i]; // +1: never 0
    }
    if (final_pass) {
        P.hl_key[i] = key;
    } else {
        sort_keys[i] = member ? ~key : ~0ull;
        sort_vals[i] = i;
    }
    }
}
// after sorting (t, idx): arrival index of each element
__global__ __launch_bounds__(MLM_BLOCK) void k_assign_rank(const MlmDev P, unsigned int n, const uint32_t *sorted_idx,
                                                           unsigned int limit, int to_arr) {
    for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < n && r < limit; r += gridDim.x * blockDim.x) {
        const uint32_t i = sorted_idx[r];
        P.hl_vt[i] = r;
        if (to_arr) P.hl_arr[i] = r;
    }
}
__global__ __launch_bounds__(MLM_BLOCK) void k_time_keys(const MlmDev P, unsigned int n, unsigned long long *sort_keys,
                                                         uint32_t *sort_vals) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        sort_keys[i] = P.hl_t[i];
        sort_vals[i] = i;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage C
// ---------------------------------------------------------------------------------------------------------------
// Last Stage A kernel (batched): everything about Stage C that does not depend on the map — the world voxel
// (packed block key + cell id) of every unique hit cell and of every unique miss cell, and a speculative lookup of the
// block's pool slot.  Hits: one per lane.  Misses: a wave takes 64 words of the miss mask (one per lane), reserves room
// for all their set bits with one atomic, then expands the non-zero words two at a time (lane = bit); the mask is
// cleared on the way.  No block barriers.
__global__ __launch_bounds__(MLM_BLOCK) void k_prepare_voxels(MLM_SLOT_ARGS) {
    MLM_SLOT_SETUP
    {
        const unsigned int n = P.ctr->u_hit;
        for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            int rho, phi, z;
            mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
            double wx, wy, wz;
            mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz);
            int gx, gy, gz, cid;
            mlm_voxel_of(P, wx, wy, wz, gx, gy, gz, cid);
            const unsigned long long bkey = mlm_pack_key(gx, gy, gz);
            P.hl_bkey[i] = bkey;
            P.hl_cid[i] = (uint32_t)cid;
            // blocks are only ever added and keep their slot, so a block found now stays valid; a block that does not
            // exist yet (or is being inserted by an earlier frame's Stage C right now) is resolved by k_voxelize
            P.hl_slot[i] = mlm_block_find_k(P, bkey);
        }
    }
    if (P.explore) return; // frontier mode keeps insertion times instead of a bit mask (k_ex_collect_misses)
    const int lane = threadIdx.x & 63;
    const unsigned int sl = blockIdx.x & 7;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    const int half = lane >> 5, b = lane & 31;
    for (unsigned int w0 = wave * 64; w0 < (unsigned int)P.nMissWords; w0 += n_waves * 64) {
        const unsigned int w = w0 + lane;
        uint32_t bits = 0;
        if (w < (unsigned int)P.nMissWords) {
#pragma unroll
            for (int c = 0; c < MLM_MISS_COPIES; ++c) {
                const uint32_t v = P.miss_bits[(size_t)c * P.nMissWords + w];
                if (v) P.miss_bits[(size_t)c * P.nMissWords + w] = 0;
                bits |= v;
            }
        }
        unsigned long long nz = __ballot(bits != 0);
        if (!nz) continue;
        const uint32_t cnt = (uint32_t)__popc(bits);
        const uint32_t incl = mlm_wave_incl_scan(cnt);
        const uint32_t total = mlm_readlane(incl, 63);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&P.ctr->mc_cnt[sl][0], total);
        base = mlm_readlane(base, 0);
        const uint32_t excl = incl - cnt;
        while (nz) {
            const int sa = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
            int sb = sa;
            uint32_t bits_b = 0;
            if (nz) {
                sb = __ffsll((long long)nz) - 1;
                nz &= nz - 1;
                bits_b = mlm_readlane(bits, sb);
            }
            const uint32_t my_bits = half ? bits_b : mlm_readlane(bits, sa);
            const uint32_t my_off = half ? mlm_readlane(excl, sb) : mlm_readlane(excl, sa);
            const int wi_all = (int)w0 + (half ? sb : sa);
            if ((my_bits >> b) & 1u) {
                const int row = wi_all / P.RW;
                const int wi = wi_all - row * P.RW;
                const int z = row / P.nPhi;
                const int phi = row - z * P.nPhi;
                const int rho = wi * 32 + b;
                double wx, wy, wz;
                mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz);
                int gx, gy, gz, c;
                mlm_voxel_of(P, wx, wy, wz, gx, gy, gz, c);
                const unsigned long long bkey = mlm_pack_key(gx, gy, gz);
                if (P.record_awareness) {
                    const unsigned int p2 = atomicAdd(&P.ctr->n_miss_list, 1u);
                    P.ml_cell[p2] = (uint32_t)(z * P.nRhoPhi + phi * P.nRho + rho);
                }
                const uint32_t pos = base + my_off + (uint32_t)__popc(my_bits & ((1u << b) - 1u));
                if (pos < P.mc_cap) {
                    P.mc_bkey[(size_t)sl * P.mc_cap + pos] = bkey;
                    P.mc_cid[(size_t)sl * P.mc_cap + pos] = (uint32_t)c;
                    P.mc_slot[(size_t)sl * P.mc_cap + pos] = mlm_block_find_k(P, bkey);
                }
            }
        }
    }
}

#define MLM_APPLY_REGS 12 // hit contributions of one voxel ordered in registers by k_apply
#define MLM_HAS_HITS 0x80000000u // flag in vox_miss: the voxel has pending hit contributions this frame

// bucket-first table of the speculative path: entry = (~seq << 32) | first insertion time.  atomicMin keeps the
// newest frame's smallest time, so the table never needs clearing between frames.
__device__ __forceinline__ unsigned long long mlm_bkt_entry(int seq, uint32_t t) {
    return ((unsigned long long)(0xFFFFFFFFu - (uint32_t)seq) << 32) | (unsigned long long)t;
}

// Stage B+C, kernel 1 of 2 (the part that needs the map, in frame order).  blockIdx.y == 0: one unique hit per lane —
// bucket-first time of the emulated container (speculative single epoch; n_bkt == 0 on the exact path where hl_key
// is already final), block lookup/creation (allocate_ram, map_local.h:215-231), push on the voxel's pending list.
// blockIdx.y == 1 + k: miss-cell sub-list k — count the frame's misses per voxel (their order is irrelevant: every
// miss adds the same constant, map_local.cpp:188-192).
// Both Stage B+C kernels are short dependent chains of memory round trips (a frame is 17 k hits + 79 k miss cells), so
// every lane issues the loads of its first item together with the loads that decide whether there is an item at all
// (frame not failed, list length): two round trips per kernel instead of four.
__device__ __forceinline__ void mlm_voxelize_body(const MlmDev &P, const MlmFrame &F, unsigned long long n_bkt, unsigned int by) {
    const int frame_idx = F.seq;
    // per-voxel scratch and bucket table of this frame's parity (see MlmDev::vox_head)
    int *vox_head = P.vox_head + (size_t)(frame_idx & 1) * P.vox_stride;
    uint32_t *vox_miss = P.vox_miss + (size_t)(frame_idx & 1) * P.vox_stride;
    unsigned long long *bkt64 = P.bkt64 + (size_t)(frame_idx & 1) * P.bkt_stride;
    const unsigned int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (by == 0) {
        // speculative loads of item i0 (in bounds of the arrays, whatever u_hit turns out to be)
        const bool inb = i0 < (unsigned int)P.nCells;
        uint32_t p_cell = 0, p_vt = 0, p_cid = 0;
        int p_slot = -1;
        unsigned long long p_bkey = 0;
        if (inb) {
            p_cell = P.hl_cell[i0];
            p_vt = P.hl_vt[i0];
            p_slot = P.hl_slot[i0];
            p_cid = P.hl_cid[i0];
            p_bkey = P.hl_bkey[i0];
        }
        const int ff = __hip_atomic_load(&P.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int n = P.ctr->u_hit;
        if (ff <= frame_idx) return;
        if (n > F.rehash_thr) { // the frame does not fit the emulated container without a rehash: speculation miss
            if (i0 == 0) atomicMin(&P.g->fail_frame, frame_idx);
            return;
        }
        for (unsigned int i = i0; i < n; i += stride) {
            if (i != i0) {
                p_cell = P.hl_cell[i];
                p_vt = P.hl_vt[i];
                p_slot = P.hl_slot[i];
                p_cid = P.hl_cid[i];
                p_bkey = P.hl_bkey[i];
            }
            if (n_bkt) {
                int rho, phi, z;
                mlm_cell_rpz(P, p_cell, rho, phi, z);
                const unsigned long long b = mlm_hash_rpz(rho, phi, z) % n_bkt;
                atomicMin(&bkt64[b], mlm_bkt_entry(frame_idx, p_vt));
                P.hl_bkt[i] = (uint32_t)b;
            }
            int slot = p_slot;
            if (slot < 0) slot = mlm_block_slot(P, p_bkey);
            if (P.explore && slot >= 0 && P.blk_collapsed[slot]) slot = -3; // released block: allocate_ram() is false
            if (slot < 0) {
                P.hl_vox[i] = -1;
                P.hl_next[i] = -2;
                continue;
            }
            const int v = slot * P.cells + (int)p_cid;
            P.hl_vox[i] = v;
            P.hl_next[i] = atomicExch(&vox_head[v], (int)i);
            if (!P.explore) atomicOr(&vox_miss[v], MLM_HAS_HITS); // tells k_apply's miss side that a hit owner exists
        }
        return;
    }
    const unsigned int sl = by - 1;
    const bool inb = i0 < P.mc_cap;
    int p_slot = -1;
    uint32_t p_cid = 0;
    if (inb) {
        p_slot = P.mc_slot[(size_t)sl * P.mc_cap + i0];
        p_cid = P.mc_cid[(size_t)sl * P.mc_cap + i0];
    }
    const int ff = __hip_atomic_load(&P.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int n = min(P.ctr->mc_cnt[sl][0], P.mc_cap);
    const unsigned int n_hits = P.ctr->u_hit;
    if (ff <= frame_idx || n_hits > F.rehash_thr) return; // (the hit side flags the speculation miss)
    unsigned int n_here = 0;
    for (unsigned int i = i0; i < n; i += stride) {
        ++n_here;
        const size_t at = (size_t)sl * P.mc_cap + i;
        if (i != i0) {
            p_slot = P.mc_slot[at];
            p_cid = P.mc_cid[at];
        }
        int slot = p_slot;
        if (slot < 0) slot = mlm_block_slot(P, P.mc_bkey[at]);
        int v = -1;
        if (slot >= 0) {
            v = slot * P.cells + (int)p_cid;
            // the first miss of a voxel this frame owns it in k_apply; the others only count
            if ((atomicAdd(&vox_miss[v], 1u) & ~MLM_HAS_HITS) != 0) v = -1;
        }
        P.mc_vox[at] = v;
    }
    // statistics: unique miss cells
    for (int off = 32; off > 0; off >>= 1) n_here += __shfl_xor(n_here, off, 64);
    if ((threadIdx.x & 63) == 0 && n_here) atomicAdd(&P.ctr->umiss_part[blockIdx.x & 7][0], n_here);
}
__global__ __launch_bounds__(MLM_BLOCK) void k_voxelize(const MlmDev P, const MlmFrame F, unsigned long long n_bkt) {
    mlm_voxelize_body(P, F, n_bkt, blockIdx.y);
}

// iteration-order key of unique hit j: larger = visited earlier by the reference's container walk
__device__ __forceinline__ unsigned long long mlm_order_key(const MlmDev &P, const unsigned long long *bkt64, int j, int explicit_keys) {
    if (explicit_keys) return P.hl_key[j];
    const unsigned long long first = bkt64[P.hl_bkt[j]] & 0xFFFFFFFFull;
    return ((first + 1ull) << 32) | (unsigned long long)P.hl_vt[j];
}

// map_local.cpp:188-203: k misses on one voxel
__device__ __forceinline__ void mlm_apply_misses(const MlmDev &P, float &L, uint8_t &o, uint32_t k) {
    for (uint32_t j = 0; j < k; ++j) {
        if (L >= P.lo_min) {
            L = L + P.lo_miss;
            L = L < P.lo_min ? P.lo_min : L;
        }
        if (L < P.lo_sh && o != 'f') o = 'f';
    }
}

// Kernel 2 of 2.  blockIdx.y == 0: the first-pushed entry of each voxel list (next == -1) owns the voxel: it replays
// the voxel's hit contributions in the reference's iteration order (descending key, map_local.cpp:157-171), then the
// voxel's misses of this frame (the reference runs all hits before all misses, map_local.cpp:147,176).
// blockIdx.y == 1 + k: miss cells of sub-list k that were their voxel's first miss — voxels that also have hits are left to the hit owner
// (the MLM_HAS_HITS flag in the miss counter tells the two sides apart, so each voxel's misses are applied once).
__device__ __forceinline__ void mlm_apply_body(const MlmDev &P, int frame_idx, int explicit_keys, unsigned int by) {
    int *vox_head = P.vox_head + (size_t)(frame_idx & 1) * P.vox_stride;
    uint32_t *vox_miss = P.vox_miss + (size_t)(frame_idx & 1) * P.vox_stride;
    const unsigned long long *bkt64 = P.bkt64 + (size_t)(frame_idx & 1) * P.bkt_stride;
    const unsigned int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (by == 0) {
        // speculative loads of item i0 (see k_voxelize)
        int p_next = 0, p_vox = -1;
        float p_inc = 0.0f;
        if (i0 < (unsigned int)P.nCells) {
            p_next = P.hl_next[i0];
            p_vox = P.hl_vox[i0];
            p_inc = P.hl_inc[i0];
        }
        const int ff = __hip_atomic_load(&P.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int n = P.ctr->u_hit;
        if (ff <= frame_idx) return;
        for (unsigned int i = i0; i < n; i += stride) {
            if (i != i0) {
                p_next = P.hl_next[i];
                p_vox = P.hl_vox[i];
                p_inc = P.hl_inc[i];
            }
            if (p_next != -1) continue;
            const int v = p_vox;
            // the voxel's miss count is fetched (and reset) in the same round trip as its state
            uint32_t km = 0;
            if (!P.explore) km = atomicExch(&vox_miss[v], 0u) & ~MLM_HAS_HITS;
            const int head = vox_head[v];
            float L = P.log_odds[v];
            uint8_t o = P.occ[v];
            const uint8_t o0 = o;
            if (head == (int)i) { // the common case: a single contribution
                if (L < P.lo_max) {
                    L = L + p_inc;
                    L = L > P.lo_max ? P.lo_max : L;
                }
                if (L > P.lo_sh && o != 'o') o = 'o';
            } else {
                // one walk over the list collects (key, increment) into a register-resident array kept in descending
                // key order (static indexing: fully unrolled compare-exchange insertion); lists longer than the array
                // fall back to repeated selection straight from memory
                unsigned long long ks[MLM_APPLY_REGS];
                float vs[MLM_APPLY_REGS];
#pragma unroll
                for (int q = 0; q < MLM_APPLY_REGS; ++q) {
                    ks[q] = 0; // real keys are never 0
                    vs[q] = 0.0f;
                }
                int cnt = 0;
                for (int j = head; j >= 0; j = P.hl_next[j]) {
                    unsigned long long k = mlm_order_key(P, bkt64, j, explicit_keys);
                    float inc = P.hl_inc[j];
                    ++cnt;
#pragma unroll
                    for (int q = 0; q < MLM_APPLY_REGS; ++q) {
                        if (k > ks[q]) {
                            const unsigned long long tk = ks[q];
                            const float tv = vs[q];
                            ks[q] = k;
                            vs[q] = inc;
                            k = tk;
                            inc = tv;
                        }
                    }
                }
                if (cnt <= MLM_APPLY_REGS) {
#pragma unroll
                    for (int q = 0; q < MLM_APPLY_REGS; ++q) {
                        if (q < cnt) {
                            if (L < P.lo_max) {
                                L = L + vs[q];
                                L = L > P.lo_max ? P.lo_max : L;
                            }
                            if (L > P.lo_sh && o != 'o') o = 'o';
                        }
                    }
                } else {
                    unsigned long long last = ~0ull;
                    for (;;) {
                        int best = -1;
                        unsigned long long bestkey = 0;
                        for (int j = head; j >= 0; j = P.hl_next[j]) {
                            const unsigned long long k = mlm_order_key(P, bkt64, j, explicit_keys);
                            if (k < last && (best < 0 || k > bestkey)) {
                                best = j;
                                bestkey = k;
                            }
                        }
                        if (best < 0) break;
                        if (L < P.lo_max) {
                            L = L + P.hl_inc[best];
                            L = L > P.lo_max ? P.lo_max : L;
                        }
                        if (L > P.lo_sh && o != 'o') o = 'o';
                        last = bestkey;
                    }
                }
            }
            if (!P.explore) {
                mlm_apply_misses(P, L, o, km);
            } else if (o == 'o' && o0 != 'o') {
                P.frnt[v] = 0; // frontier.erase(subbox_id), map_local.cpp:167-168
            }
            P.log_odds[v] = L;
            P.occ[v] = o;
            vox_head[v] = -1;
        }
        return;
    }
    const unsigned int sl = by - 1;
    int p_v = -1;
    if (i0 < P.mc_cap) p_v = P.mc_vox[(size_t)sl * P.mc_cap + i0];
    const int ff = __hip_atomic_load(&P.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int n = min(P.ctr->mc_cnt[sl][0], P.mc_cap);
    if (ff <= frame_idx) return;
    for (unsigned int i = i0; i < n; i += stride) {
        const int v = (i == i0) ? p_v : P.mc_vox[(size_t)sl * P.mc_cap + i];
        if (v < 0) continue; // not the voxel's first miss (or no block)
        // vox_miss[v]: 0 = a hit owner already applied the misses; MLM_HAS_HITS set = a hit owner will; else the
        // voxel has misses only and this lane is the only one that touches it
        const uint32_t k = __hip_atomic_load(&vox_miss[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float L = P.log_odds[v]; // same round trip as the counter; unused if a hit owner handles the voxel
        uint8_t o = P.occ[v];
        if (k == 0 || (k & MLM_HAS_HITS)) continue;
        vox_miss[v] = 0;
        mlm_apply_misses(P, L, o, k);
        P.log_odds[v] = L;
        P.occ[v] = o;
    }
}
// see MlmDev::spec_on
__device__ __forceinline__ bool mlm_ex_spec_skip(const MlmDev &P) {
    if (!P.spec_on) return false;
    const MlmCounters *c = P.ctr;
    return c->sector_overflow != 0u || c->u_hit > P.spec_hit_thr || c->n_ex_miss > P.spec_miss_thr;
}
__global__ __launch_bounds__(MLM_BLOCK) void k_apply(const MlmDev P, int frame_idx, int explicit_keys) {
    if (mlm_ex_spec_skip(P)) return;
    mlm_apply_body(P, frame_idx, explicit_keys, blockIdx.y);
}

// k_apply of frame f and k_voxelize of frame f+1 in ONE launch (speculative path of a batch): the two touch different
// copies of the per-voxel scratch (sequence parity) and different frame slots, and k_voxelize(f+1) needs nothing of
// k_apply(f) — so the serial chain of a batch is n+1 launches instead of 2n.  blockIdx.y < 1 + MLM_RAY_LISTS: the apply
// side; the rest: the voxelize side.  has_a / has_v: the first launch of a batch has no apply side, the last no
// voxelize side.
__global__ __launch_bounds__(MLM_BLOCK) void k_apply_voxelize(const MlmDev Pa, int frame_a, int has_a, const MlmDev Pv,
                                                              const MlmFrame Fv, unsigned long long n_bkt, int has_v) {
    __builtin_amdgcn_s_setprio(3); // the serial chain of the pipeline: its few waves issue ahead of Stage A's
    if (blockIdx.y < 1 + MLM_RAY_LISTS) {
        if (has_a) mlm_apply_body(Pa, frame_a, 0, blockIdx.y);
    } else {
        if (has_v) mlm_voxelize_body(Pv, Fv, n_bkt, blockIdx.y - (1 + MLM_RAY_LISTS));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Queries
// ---------------------------------------------------------------------------------------------------------------
// getOccupancy, mlmap.h:170-193
__device__ __forceinline__ int mlm_get_occupancy(const MlmDev &P, double x, double y, double z) {
    int gx, gy, gz, cid;
    mlm_voxel_of(P, x, y, z, gx, gy, gz, cid);
    const int slot = mlm_block_find(P, gx, gy, gz);
    if (slot < 0) return -1;
    if (P.explore && P.blk_collapsed[slot]) cid = 0; // occupancy.size() == 1 -> occupancy[0], mlmap.h:183-184
    const uint8_t r = P.occ[(size_t)slot * P.cells + cid];
    return r == 'o' ? 0 : (r == 'f' ? 1 : -1);
}
// logit_inv macro, mlmap.h:40, evaluated in double then narrowed (getOdd returns float)
__device__ __forceinline__ float mlm_logit_inv(float L) {
    const double p = pow(10.0, (double)L);
    return (float)(p / (1 + p));
}
// getOdd(glb_id, subbox_id), mlmap.h:227-235
__device__ __forceinline__ float mlm_get_odd_at(const MlmDev &P, int gx, int gy, int gz, int cid) {
    const int slot = mlm_block_find(P, gx, gy, gz);
    if (slot < 0) return 0.5f;
    if (P.explore && P.blk_collapsed[slot]) cid = 0; // log_odds.size() == 1 -> log_odds[0], mlmap.h:221-222
    return mlm_logit_inv(P.log_odds[(size_t)slot * P.cells + cid]);
}
// 6-neighbour step of subbox_neighbors (map_local.cpp:77-120): order +z,-z,+y,-y,+x,-x
__device__ __forceinline__ void mlm_neighbor(const MlmDev &P, int dir, int &gx, int &gy, int &gz, int &cid) {
    int cz = cid / (P.n * P.n);
    int cy = (cid - cz * P.n * P.n) / P.n;
    int cx = cid - cz * P.n * P.n - cy * P.n;
    int *c, *g;
    int step;
    switch (dir) {
    case 0: c = &cz; g = &gz; step = 1; break;
    case 1: c = &cz; g = &gz; step = -1; break;
    case 2: c = &cy; g = &gy; step = 1; break;
    case 3: c = &cy; g = &gy; step = -1; break;
    case 4: c = &cx; g = &gx; step = 1; break;
    default: c = &cx; g = &gx; step = -1; break;
    }
    *c += step;
    if (*c >= P.n) {
        *g += 1;
        *c = 0;
    } else if (*c < 0) {
        *g -= 1;
        *c = P.n - 1;
    }
    cid = cz * P.n * P.n + cy * P.n + cx;
}

// mode 0: getOccupancy  1: getOccupancy(pos, inflate)  2: getInflateOccupancy  3: getOdd  4: getOddGrad
__global__ __launch_bounds__(MLM_BLOCK) void k_query(const MlmDev P, int mode, const double *pos, int n, float inflate,
                                                     int max_iter, int8_t *out_i8, float *out_f, double *out_d3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
    if (mode == 0) {
        out_i8[i] = (int8_t)mlm_get_occupancy(P, x, y, z);
    } else if (mode == 1) {
        // 19-point stencil, mlmap.h:142-169; Vec3(±inflate) promotes the float to double
        const double f = inflate;
        // (the signs of the nineteen offsets as a constant table: a per-thread array of doubles indexed by the loop would live in
        // scratch memory)
        static const int8_t o[19][3] = {{0, 0, 0},   {0, 0, 1},   {0, 0, -1}, {0, 1, 0},   {0, -1, 0}, {1, 0, 0},  {-1, 0, 0},
                                        {-1, 1, 0},  {-1, -1, 0}, {1, 1, 0},  {1, -1, 0},  {0, -1, 1}, {0, -1, -1}, {0, 1, 1},
                                        {0, 1, -1},  {-1, 0, 1},  {-1, 0, -1}, {1, 0, 1},  {1, 0, -1}};
        int res = 1;
        for (int k = 0; k < 19; ++k)
            if (mlm_get_occupancy(P, x + (o[k][0] ? (o[k][0] > 0 ? f : -f) : 0.0), y + (o[k][1] ? (o[k][1] > 0 ? f : -f) : 0.0),
                                  z + (o[k][2] ? (o[k][2] > 0 ? f : -f) : 0.0)) == 0) {
                res = 0;
                break;
            }
        out_i8[i] = (int8_t)res;
    } else if (mode == 2) {
        // getInflateOccupancy, mlmap.h:195-211
        int gx, gy, gz, cid;
        mlm_voxel_of(P, x, y, z, gx, gy, gz, cid);
        const int slot = mlm_block_find(P, gx, gy, gz);
        int res = -1;
        if (slot >= 0 && !(P.explore && P.blk_collapsed[slot]) && P.infl[(size_t)slot * P.cells + cid] == 'o') res = 0;
        out_i8[i] = (int8_t)res;
    } else if (mode == 3) {
        int gx, gy, gz, cid;
        mlm_voxel_of(P, x, y, z, gx, gy, gz, cid);
        out_f[i] = mlm_get_odd_at(P, gx, gy, gz, cid);
    } else {
        // getOddGrad, mlmap.h:237-295
        int gx, gy, gz, cid;
        mlm_voxel_of(P, x, y, z, gx, gy, gz, cid);
        float min_odd = mlm_get_odd_at(P, gx, gy, gz, cid);
        const float ori_odd = min_odd;
        int ngx[6], ngy[6], ngz[6], ncid[6];
        int mgx = 0, mgy = 0, mgz = 0, mcid = 0;
        bool flag = false;
        for (int iter = 0; iter < max_iter && !flag; ++iter) {
            for (int d = 0; d < 6; ++d) {
                if (iter == 0) {
                    ngx[d] = gx;
                    ngy[d] = gy;
                    ngz[d] = gz;
                    ncid[d] = cid;
                }
                mlm_neighbor(P, d, ngx[d], ngy[d], ngz[d], ncid[d]);
                const float tmp = mlm_get_odd_at(P, ngx[d], ngy[d], ngz[d], ncid[d]);
                if (tmp < min_odd) {
                    min_odd = tmp;
                    mgx = ngx[d];
                    mgy = ngy[d];
                    mgz = ngz[d];
                    mcid = ncid[d];
                    flag = true;
                }
            }
        }
        double rx = 0.0, ry = 0.0, rz = 0.0;
        if (flag) {
            // subbox_id2xyz_glb_vec, map_local.h:208-213
            const int cz = mcid / (P.n * P.n);
            const int cy = (mcid - cz * P.n * P.n) / P.n;
            const int cx = mcid - cz * P.n * P.n - cy * P.n;
            const double s = (double)(ori_odd - min_odd);
            rx = ((mgx * P.d_glb + cx * P.d_sub + P.d_sub_half) - x) * s;
            ry = ((mgy * P.d_glb + cy * P.d_sub + P.d_sub_half) - y) * s;
            rz = ((mgz * P.d_glb + cz * P.d_sub + P.d_sub_half) - z) * s;
        }
        out_d3[3 * (size_t)i] = rx;
        out_d3[3 * (size_t)i + 1] = ry;
        out_d3[3 * (size_t)i + 2] = rz;
    }
}

// getOdd(const Vec3I &glb_id, size_t subbox_id), mlmap.h:227-235
__global__ __launch_bounds__(MLM_BLOCK) void k_query_odds_at(const MlmDev P, const int32_t *glb, const int32_t *sub, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = mlm_get_odd_at(P, glb[3 * (size_t)i], glb[3 * (size_t)i + 1], glb[3 * (size_t)i + 2], sub[i]);
}

// ---------------------------------------------------------------------------------------------------------------
// Host mirror of the map (mlm_mirror.h): the reference answers getOccupancy / getOdd / getOddGrad with ONE hash lookup on the
// host (mlmap.h:170-295, ~30 ns); a planner calls them position by position.  Small query batches are therefore answered from a
// pinned host copy of the block planes that mirrors the pool slot by slot.  This kernel brings the copy up to date: a workgroup
// per block copies the planes of the blocks that are new (slot >= n_known) or whose key lies inside one of the boxes of block
// indices the host recorded for the map-changing calls since the last refresh, straight into the host planes over the link.
// ---------------------------------------------------------------------------------------------------------------
#define MLM_MIRROR_BOXES 16
struct MlmMirrorBoxes {
    int n, all;
    int lo[MLM_MIRROR_BOXES][3], hi[MLM_MIRROR_BOXES][3]; // inclusive bounds in block indices
};
__global__ __launch_bounds__(MLM_BLOCK) void k_mirror_refresh(const MlmDev P, unsigned int n_known, const MlmMirrorBoxes B, float *m_lo, uint8_t *m_occ,
                                                              uint8_t *m_infl, uint8_t *m_col, int *m_keys, unsigned int m_cap, unsigned int *m_stat) {
    const unsigned int nb = min(mlm_gp(P.g)->n_blocks, (unsigned int)P.max_blocks);
    __shared__ unsigned int s_copied;
    if (threadIdx.x == 0) s_copied = 0u;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) m_stat[0] = nb; // (the host learns the block count here: more than m_cap -> it enlarges the planes and asks again)
    const unsigned int lim = min(nb, m_cap);
    for (unsigned int b = blockIdx.x; b < lim; b += gridDim.x) {
        const int gx = P.block_keys[3 * (size_t)b], gy = P.block_keys[3 * (size_t)b + 1], gz = P.block_keys[3 * (size_t)b + 2];
        bool dirty = B.all || b >= n_known;
        for (int k = 0; k < B.n && !dirty; ++k)
            dirty = gx >= B.lo[k][0] && gx <= B.hi[k][0] && gy >= B.lo[k][1] && gy <= B.hi[k][1] && gz >= B.lo[k][2] && gz <= B.hi[k][2];
        if (!dirty) continue; // (uniform over the workgroup)
        const size_t v0 = (size_t)b * P.cells;
        for (int c = threadIdx.x; c < P.cells; c += blockDim.x) {
            m_lo[v0 + c] = P.log_odds[v0 + c];
            m_occ[v0 + c] = P.occ[v0 + c];
            m_infl[v0 + c] = P.infl[v0 + c];
        }
        if (threadIdx.x == 0) {
            m_keys[3 * (size_t)b] = gx;
            m_keys[3 * (size_t)b + 1] = gy;
            m_keys[3 * (size_t)b + 2] = gz;
            m_col[b] = P.explore ? P.blk_collapsed[b] : (uint8_t)0;
            s_copied++;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) m_stat[2 + blockIdx.x] = s_copied; // (statistic: blocks this workgroup copied — plain stores, no atomics across the link)
}

// mlm_import_blocks: find or create the block of every imported key (allocate_ram, map_local.h:215-231) ...
// grow_pool: re-insert the blocks' keys into the new, larger table (slot = the block's index in the pool)
__global__ __launch_bounds__(MLM_BLOCK) void k_rehash_blocks(const MlmDev P, unsigned int n) {
    const unsigned int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const unsigned long long key = mlm_pack_key(P.block_keys[3 * (size_t)b], P.block_keys[3 * (size_t)b + 1], P.block_keys[3 * (size_t)b + 2]);
    uint32_t h = mlm_mix(key) & P.ht_mask;
    for (uint32_t probe = 0; probe <= P.ht_mask; ++probe) {
        if (atomicCAS(&P.ht_keys[h], MLM_HT_EMPTY, key) == MLM_HT_EMPTY) {
            P.ht_slot[h] = (int)b;
            return;
        }
        h = (h + 1) & P.ht_mask;
    }
}
__global__ __launch_bounds__(MLM_BLOCK) void k_import_slots(const MlmDev P, const int32_t *keys, int n, int *slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slots[i] = mlm_block_slot(P, mlm_pack_key(keys[3 * (size_t)i], keys[3 * (size_t)i + 1], keys[3 * (size_t)i + 2]));
}
// ... and overwrite its cells (any source may be null: that plane is left as it is)
__global__ __launch_bounds__(MLM_BLOCK) void k_import_cells(const MlmDev P, const int *slots, int n, const float *lo, const uint8_t *occ,
                                                            const uint8_t *infl, const uint8_t *collapsed) {
    const long long total = (long long)n * P.cells;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / P.cells), c = (int)(i - (long long)b * P.cells);
        const int slot = slots[b];
        if (slot < 0) continue;
        const size_t v = (size_t)slot * P.cells + c;
        if (lo) P.log_odds[v] = lo[i];
        if (occ) P.occ[v] = occ[i];
        if (infl) P.infl[v] = infl[i];
        if (c == 0 && collapsed && P.explore) P.blk_collapsed[slot] = collapsed[b];
    }
}

// Global-map merge (no reference counterpart), packing side: row b of the dense exchange buffers = the
// cells of block keys[b] (zeros / "not observed" where this map does not hold the block).  A released block answers with
// its element 0 (map_local.cpp:221-226).
__global__ __launch_bounds__(MLM_BLOCK) void k_merge_pack(const MlmDev P, const int32_t *keys, int n, float *lo, uint8_t *seen) {
    const long long total = (long long)n * P.cells;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / P.cells);
        int c = (int)(i - (long long)b * P.cells);
        const int slot = mlm_block_find(P, keys[3 * (size_t)b], keys[3 * (size_t)b + 1], keys[3 * (size_t)b + 2]);
        float L = 0.0f;
        uint8_t s = 0;
        if (slot >= 0) {
            if (P.explore && P.blk_collapsed[slot]) c = 0;
            L = P.log_odds[(size_t)slot * P.cells + c];
            s = P.occ[(size_t)slot * P.cells + c] != 'u';
        }
        lo[i] = L;
        seen[i] = s;
    }
}
// ... and the finishing side: summed log-odds clamped to [min, max]; class 'o' above occupied_sh, else 'f' where any map
// had observed the voxel, else 'u'
__global__ __launch_bounds__(MLM_BLOCK) void k_merge_finish(const MlmDev P, float *lo, const uint8_t *seen, size_t n, uint8_t *occ) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float L = lo[i];
        L = L < P.lo_min ? P.lo_min : (L > P.lo_max ? P.lo_max : L);
        lo[i] = L;
        occ[i] = L > P.lo_sh ? 'o' : (seen[i] ? 'f' : 'u');
    }
}

// setFree_map_in_bound, mlmap.cpp:388-407: the lattice coordinates are produced on the host by the same
// accumulating additions (x += d) and handed over as three axis arrays.
__global__ __launch_bounds__(MLM_BLOCK) void k_set_free(const MlmDev P, const double *xs, int nx, const double *ys, int ny,
                                                        const double *zs, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * ny * nz) return;
    const int iz = (int)(i % nz);
    const int iy = (int)((i / nz) % ny);
    const int ix = (int)(i / ((long long)nz * ny));
    int gx, gy, gz, cid;
    mlm_voxel_of(P, xs[ix], ys[iy], zs[iz], gx, gy, gz, cid);
    const int slot = mlm_block_find(P, gx, gy, gz);
    if (slot < 0 || (P.explore && P.blk_collapsed[slot])) return; // only blocks with occupancy.size() > 1, mlmap.cpp:399
    P.occ[(size_t)slot * P.cells + cid] = 'f';
    P.log_odds[(size_t)slot * P.cells + cid] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// Inflation (mlmap::inflate_map, mlmap.cpp:286-309; local_map_cartesian::inflate_atpos, map_local.h:233-264)
// ---------------------------------------------------------------------------------------------------------------
// The reference visits the (2G+1)^3 blocks around the vehicle in x-outer/z-inner order; visiting a block first wipes
// its inflate_occupancy and then dilates its own 'o' cells (L1 ball of radius inflate_n) — also into neighbouring
// blocks.  A neighbour that is visited LATER is wiped after that write, so a write from block Bs into block Bt inside
// the cube survives iff rank(Bs) >= rank(Bt); blocks outside the cube are never wiped and only accumulate.  That rule
// is order free, which is what the two kernels below implement.
__device__ __forceinline__ int mlm_cube_rank(int ox, int oy, int oz, int G) {
    const int w = 2 * G + 1;
    return ((ox + G) * w + (oy + G)) * w + (oz + G);
}
// one thread per (cube block, cell): wipe
__global__ __launch_bounds__(MLM_BLOCK) void k_inflate_reset(const MlmDev P, int cgx, int cgy, int cgz, int G) {
    const int w = 2 * G + 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)w * w * w * P.cells) return;
    const int c = (int)(i % P.cells);
    const int b = (int)(i / P.cells);
    const int oz = b % w - G, oy = (b / w) % w - G, ox = b / (w * w) - G;
    const int slot = mlm_block_find(P, cgx + ox, cgy + oy, cgz + oz);
    if (slot >= 0 && !(P.explore && P.blk_collapsed[slot])) P.infl[(size_t)slot * P.cells + c] = 'u';
}
// one thread per (cube block, cell): dilate the 'o' cells above flate_height
__global__ __launch_bounds__(MLM_BLOCK) void k_inflate_spread(const MlmDev P, int cgx, int cgy, int cgz, int G, int R,
                                                              double flate_height) {
    const int w = 2 * G + 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)w * w * w * P.cells) return;
    const int c = (int)(i % P.cells);
    const int b = (int)(i / P.cells);
    const int oz = b % w - G, oy = (b / w) % w - G, ox = b / (w * w) - G;
    const int gx = cgx + ox, gy = cgy + oy, gz = cgz + oz;
    const int slot = mlm_block_find(P, gx, gy, gz);
    if (slot < 0 || (P.explore && P.blk_collapsed[slot])) return;
    if (P.occ[(size_t)slot * P.cells + c] != 'o') return;
    const int cz = c / (P.n * P.n), cy = (c - cz * P.n * P.n) / P.n, cx = c - cz * P.n * P.n - cy * P.n;
    // subbox_id2xyz_glb_vec(temp_glb, it)(2) > flate_height, mlmap.cpp:303 / map_local.h:208-213
    if (!(gz * P.d_glb + cz * P.d_sub + P.d_sub_half > flate_height)) return;
    const int my_rank = mlm_cube_rank(ox, oy, oz, G);
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -R; dz <= R; ++dz) {
                if (abs(dx) + abs(dy) + abs(dz) > R) continue;
                int tx = cx + dx, ty = cy + dy, tz = cz + dz;
                int bx = 0, by = 0, bz = 0; // block displacement (one wrap: inflate_n < subbox_n)
                if (tx >= P.n) { bx = 1; tx -= P.n; } else if (tx < 0) { bx = -1; tx += P.n; }
                if (ty >= P.n) { by = 1; ty -= P.n; } else if (ty < 0) { by = -1; ty += P.n; }
                if (tz >= P.n) { bz = 1; tz -= P.n; } else if (tz < 0) { bz = -1; tz += P.n; }
                int tslot = slot;
                if (bx | by | bz) {
                    const int tox = ox + bx, toy = oy + by, toz = oz + bz;
                    const bool in_cube = abs(tox) <= G && abs(toy) <= G && abs(toz) <= G;
                    tslot = mlm_block_slot(P, mlm_pack_key(gx + bx, gy + by, gz + bz)); // allocate_ram happens anyway
                    if (tslot < 0 || (P.explore && P.blk_collapsed[tslot])) continue;
                    if (in_cube && mlm_cube_rank(tox, toy, toz, G) > my_rank) continue; // wiped later by the reference
                }
                P.infl[(size_t)tslot * P.cells + (tz * P.n * P.n + ty * P.n + tx)] = 'o';
            }
}
// /global_map payload (rviz_vis.cpp:296-327): float centres (PointXYZ) of the cells whose inflate_occupancy is 'o'
// which = 1: the /frontier payload instead (rviz_vis.cpp:267-293): centres of the cells in the blocks' frontier sets
__global__ __launch_bounds__(MLM_BLOCK) void k_export_global(const MlmDev P, unsigned int n_blocks, float *xyz,
                                                             unsigned int cap, unsigned int *counter, int which) {
    __shared__ unsigned int s_cnt[MLM_BLOCK / 64];
    __shared__ unsigned int s_base;
    const long long total = (long long)n_blocks * P.cells;
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < total; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = i0 + threadIdx.x;
        const bool on = i < total && (which ? P.frnt[i] != 0 : P.infl[i] == 'o');
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const unsigned long long m = __ballot(on);
        if (lane == 0) s_cnt[wid] = (unsigned int)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int k = 0; k < MLM_BLOCK / 64; ++k) tot += s_cnt[k];
            s_base = tot ? atomicAdd(counter, tot) : 0u;
        }
        __syncthreads();
        if (on) {
            unsigned int pos = s_base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < wid; ++k) pos += s_cnt[k];
            if (pos < cap) {
                const int slot = (int)(i / P.cells), c = (int)(i % P.cells);
                const int cz = c / (P.n * P.n), cy = (c - cz * P.n * P.n) / P.n, cx = c - cz * P.n * P.n - cy * P.n;
                // subbox_id2xyz_glb (map_local.h:201-206): doubles narrowed by PointXYZ(float,float,float)
                xyz[3 * (size_t)pos + 0] = (float)(P.block_keys[3 * slot + 0] * P.d_glb + cx * P.d_sub + P.d_sub_half);
                xyz[3 * (size_t)pos + 1] = (float)(P.block_keys[3 * slot + 1] * P.d_glb + cy * P.d_sub + P.d_sub_half);
                xyz[3 * (size_t)pos + 2] = (float)(P.block_keys[3 * slot + 2] * P.d_glb + cz * P.d_sub + P.d_sub_half);
            }
        }
        __syncthreads();
    }
}

// cv::Mat::convertTo(CV_16UC1, 1000) of the 32FC1 depth image (mlmap.cpp:482), see mlm_cv_f32_to_u16
__global__ __launch_bounds__(MLM_BLOCK) void k_convert_f32_u16(const float *src, uint16_t *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        dst[i] = (uint16_t)mlm_cv_f32_to_u16(src[i]);
    }
}

