// mlm_kernels_explore.h — kernels used only with use_exploration_frontiers: true (config2.yaml:36):
// frontier bookkeeping (local_map_cartesian::update_observation, map_local.cpp:7-33) and block release
// (map_local.cpp:208-232).
//
// What makes this mode different: when an unknown voxel A is first seen free, the FIRST of its six neighbours
// (+z,-z,+y,-y,+x,-x) that is still unknown *at that moment* joins the frontier and the search stops (`break`).  "At that
// moment" refers to the iteration order of miss_idx_set (an std::unordered_set<size_t>), so that order has to be
// reproduced too.  It is enough to know, per voxel, the order key of its first miss (tau): a neighbour N is still
// unknown when A is processed iff N was unknown after the hit phase and N's own first miss comes later (or never).
// With tau known, every newly freed voxel decides independently; the frontier set itself is a per-voxel flag whose final
// value is (old flag or added this frame) and the cell is still unknown — erasures happen when a cell turns 'f' or 'o'.
#pragma once
#include "mlm_kernels.h"

// one ray per wave, explore flavour: besides the mask bit, record the cell's first insertion time
// (miss_idx_set.emplace order: points in order, each ray from rho-1 down to 1, map_awareness.cpp:266-274)
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_walk_rays(MLM_SLOT_ARGS) {
    MLM_SLOT_SETUP
    const unsigned int n = P.ctr->n_ex_rays;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned int k = wave; k < n; k += n_waves) {
        const int32_t *q = P.ex_rays + 4 * (size_t)k;
        int rho = q[0];
        const int phi = q[1];
        int z = q[2];
        uint32_t p0 = (uint32_t)q[3];
        if (q[3] < 0) p0 = P.start_t[z * P.nRhoPhi + phi * P.nRho + rho]; // in-range start: first point of that cell
        const double slope = (rho > 0) ? (z - P.zc) / (rho * 1.0) : 0.0;
        if (rho >= P.nRho) {
            z = mlm_cvt_int(round(z - ((rho - P.nRho + 1) * slope)));
            rho = P.nRho - 1;
        }
        for (int r0 = 1; r0 < rho; r0 += 64) {
            const int r = r0 + lane;
            if (r < rho) {
                const int diff_r = rho - r;
                const int zr = mlm_cvt_int(round(z - (diff_r * slope)));
                if (0 <= zr && zr < P.nZ)
                    atomicMin(&P.miss_t[zr * P.nRhoPhi + phi * P.nRho + r], p0 * 256u + (uint32_t)(diff_r - 1));
            }
        }
    }
}

// dense sweep of miss_t -> unique miss list (ex_cell, ex_t); resets miss_t
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_collect_misses(MLM_SLOT_ARGS) {
    MLM_SLOT_SETUP
    __shared__ unsigned int s_cnt[MLM_BLOCK / 64];
    __shared__ unsigned int s_base;
    for (int c0 = blockIdx.x * blockDim.x; c0 < P.nCells; c0 += gridDim.x * blockDim.x) { // uniform per block
        const int c = c0 + (int)threadIdx.x;
        uint32_t t = MLM_EMPTY_T;
        if (c < P.nCells) t = P.miss_t[c];
        const bool has = t != MLM_EMPTY_T;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const unsigned long long m = __ballot(has);
        if (lane == 0) s_cnt[wid] = (unsigned int)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int k = 0; k < MLM_BLOCK / 64; ++k) tot += s_cnt[k];
            s_base = tot ? atomicAdd(&P.ctr->n_ex_miss, tot) : 0u;
        }
        __syncthreads();
        if (has) {
            unsigned int pos = s_base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < wid; ++k) pos += s_cnt[k];
            P.miss_t[c] = MLM_EMPTY_T;
            P.ex_cell[pos] = (uint32_t)c;
            P.ex_t[pos] = t;
            P.ex_vt[pos] = t;
            // its world voxel (pure geometry: here in the batched stage rather than in the per-frame chain)
            int rho, phi, z;
            mlm_cell_rpz(P, (uint32_t)c, rho, phi, z);
            double wx, wy, wz;
            mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz);
            int gx, gy, gz, cid;
            mlm_voxel_of(P, wx, wy, wz, gx, gy, gz, cid);
            P.ex_bkey[pos] = mlm_pack_key(gx, gy, gz);
            P.ex_cid[pos] = (uint32_t)cid;
        }
        __syncthreads();
    }
}

// ---- iteration order of miss_idx_set: std::unordered_set<size_t>, std::hash<size_t> is the identity, so
//      bucket = cell idx % bucket_count; same list rules as the hit container (see Stage B in mlm_kernels.h)
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_time_keys(const MlmDev P, unsigned int n, unsigned long long *sort_keys,
                                                            uint32_t *sort_vals) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        sort_keys[i] = P.ex_t[i];
        sort_vals[i] = i;
    }
}
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_assign_rank(const MlmDev P, unsigned int n, const uint32_t *sorted_idx,
                                                              unsigned int limit, int to_arr) {
    for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < n && r < limit; r += gridDim.x * blockDim.x) {
        const uint32_t i = sorted_idx[r];
        P.ex_vt[i] = r;
        if (to_arr) P.ex_arr[i] = r;
    }
}
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_bucket_min(const MlmDev P, unsigned int n, unsigned long long n_bkt,
                                                             unsigned int arr_limit, int use_arr) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (use_arr && P.ex_arr[i] >= arr_limit) continue;
        atomicMin(&P.bktm_first[(unsigned long long)P.ex_cell[i] % n_bkt], P.ex_vt[i]);
    }
}
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_make_keys(const MlmDev P, unsigned int n, unsigned long long n_bkt,
                                                            unsigned int arr_limit, int use_arr, int final_pass,
                                                            unsigned long long *sort_keys, uint32_t *sort_vals) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool member = !(use_arr && P.ex_arr[i] >= arr_limit);
        unsigned long long key = 0;
        if (member)
            key = ((unsigned long long)(P.bktm_first[(unsigned long long)P.ex_cell[i] % n_bkt] + 1u) << 32) |
                  (unsigned long long)P.ex_vt[i];
        if (final_pass) {
            P.ex_key[i] = key;
        } else {
            sort_keys[i] = member ? ~key : ~0ull;
            sort_vals[i] = i;
        }
    }
}

__device__ __forceinline__ void mlm_ex_apply_misses_body(const MlmDev &P);
__device__ __forceinline__ void mlm_ex_release_body(const MlmDev &P);
// Both containers in one launch when neither rehashes this frame (the usual case): blockIdx.y = 0 the hit container
// (hashed (rho,phi,z) keys), 1 the miss container (identity hash).  The bucket-first tables are the two copies of bkt64,
// tagged with a per-frame counter (newest frame's smallest time wins the min): no clearing between frames.
// blockIdx.y = 2 (when launched with three rows): the miss phase of the PREVIOUS frame (k_ex_apply_misses), which nothing in
// this frame's ordering depends on — the per-frame chain is two launches shorter (see also k_ex_order_keys).
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_order_min(const MlmDev P, unsigned int n_hit, unsigned int n_miss, unsigned long long nb_hit,
                                                            unsigned long long nb_miss, int tag, const MlmDev Pprev) {
    __builtin_amdgcn_s_setprio(3); // (the serial chain of this mode: ahead of the next batch's Stage A)
    if (mlm_ex_spec_skip(P)) return;
    if (P.spec_on) { // (the host has not seen the counts yet)
        n_hit = P.ctr->u_hit;
        n_miss = P.ctr->n_ex_miss;
    }
    const unsigned int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (blockIdx.y == 2) {
        mlm_ex_apply_misses_body(Pprev);
    } else if (blockIdx.y == 0) {
        for (unsigned int i = i0; i < n_hit; i += stride) {
            int rho, phi, z;
            mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
            atomicMin(&P.bkt64[mlm_hash_rpz(rho, phi, z) % nb_hit], mlm_bkt_entry(tag, P.hl_vt[i]));
        }
    } else {
        for (unsigned int i = i0; i < n_miss; i += stride)
            atomicMin(&P.bkt64[P.bkt_stride + (unsigned long long)P.ex_cell[i] % nb_miss], mlm_bkt_entry(tag, P.ex_vt[i]));
    }
}
// blockIdx.y = 2: the release scan of the PREVIOUS frame (after its miss phase, which ran with k_ex_order_min)
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_order_keys(const MlmDev P, unsigned int n_hit, unsigned int n_miss, unsigned long long nb_hit,
                                                             unsigned long long nb_miss, const MlmDev Pprev) {
    __builtin_amdgcn_s_setprio(3); // (the serial chain of this mode: ahead of the next batch's Stage A)
    if (mlm_ex_spec_skip(P)) return;
    if (P.spec_on) {
        n_hit = P.ctr->u_hit;
        n_miss = P.ctr->n_ex_miss;
    }
    const unsigned int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (blockIdx.y == 2) {
        mlm_ex_release_body(Pprev);
    } else if (blockIdx.y == 0) {
        for (unsigned int i = i0; i < n_hit; i += stride) {
            int rho, phi, z;
            mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
            const unsigned long long first = P.bkt64[mlm_hash_rpz(rho, phi, z) % nb_hit] & 0xFFFFFFFFull;
            P.hl_key[i] = ((first + 1ull) << 32) | (unsigned long long)P.hl_vt[i];
        }
    } else {
        for (unsigned int i = i0; i < n_miss; i += stride) {
            const unsigned long long first = P.bkt64[P.bkt_stride + (unsigned long long)P.ex_cell[i] % nb_miss] & 0xFFFFFFFFull;
            P.ex_key[i] = ((first + 1ull) << 32) | (unsigned long long)P.ex_vt[i];
        }
    }
}

// ---- Stage C, miss side --------------------------------------------------------------------------------------------
// allocate_ram as the update loops use it (map_local.h:215-231): false for a released block
__device__ __forceinline__ int mlm_ex_block_slot(const MlmDev &P, unsigned long long key) {
    const int slot = mlm_block_slot(P, key);
    if (slot >= 0 && P.blk_collapsed[slot]) return -3;
    return slot;
}

// one unique miss cell per lane: its voxel (computed in Stage A), the voxel's miss count and tau = key of its first miss in
// iteration order.  ex_vox[i] = voxel address, or -1 - address when the cell is not its voxel's first miss of the frame
// (k_ex_apply_misses takes the first ones), or INT_MIN when the block is unavailable.
__device__ __forceinline__ void mlm_ex_miss_tau_body(const MlmDev &P) {
    const unsigned int n = P.ctr->n_ex_miss;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int slot = mlm_ex_block_slot(P, P.ex_bkey[i]);
        int v = (int)0x80000000;
        if (slot >= 0) {
            v = slot * P.cells + (int)P.ex_cid[i];
            atomicMax(&P.vox_tau[v], P.ex_key[i]);
            if (atomicAdd(&P.vox_miss[v], 1u) != 0) v = -1 - v;
        }
        P.ex_vox[i] = v;
    }
}
// the hit push (k_voxelize's hit side, explicit keys) and the miss registration of a frame in ONE launch: they touch
// different per-voxel words (blockIdx.y = 0 hits, 1 misses)
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_register(const MlmDev P, const MlmFrame F) {
    __builtin_amdgcn_s_setprio(3);
    if (mlm_ex_spec_skip(P)) return;
    if (blockIdx.y == 0) mlm_voxelize_body(P, F, 0ull, 0u);
    else mlm_ex_miss_tau_body(P);
}

__device__ __forceinline__ bool mlm_inside_exp_bd(double x, double y, double z) { // map_local.h:160-165, map_local.cpp:124
    return x >= -30 && x < 30 && y >= -30 && y < 30 && z >= 0 && z < 5;
}

// The miss cell that is first on its voxel (its key == tau) plays update_observation for that voxel if the voxel
// is still unknown (it turns 'f' at this very miss: L <= occupied_sh for an unknown cell and the miss lowers it).
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_observe(const MlmDev P, const MlmFrame F) {
    __builtin_amdgcn_s_setprio(3);
    if (mlm_ex_spec_skip(P)) return;
    const unsigned int n = P.ctr->n_ex_miss;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int v = P.ex_vox[i];
        if (v == (int)0x80000000) continue;
        if (v < 0) v = -1 - v;
        const unsigned long long tau = P.vox_tau[v];
        if (P.ex_key[i] != tau || P.occ[v] != 'u') continue;
        int rho, phi, z;
        mlm_cell_rpz(P, P.ex_cell[i], rho, phi, z);
        double wx, wy, wz;
        mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz); // p_w of THIS awareness cell (map_local.cpp:180,198)
        if (!mlm_inside_exp_bd(wx, wy, wz)) continue;
        const int slot = v / P.cells, cid = v - slot * P.cells;
        P.blk_observed[slot] = 1; // observed_subboxes.emplace(glb_idx)
        const int gx0 = P.block_keys[3 * slot], gy0 = P.block_keys[3 * slot + 1], gz0 = P.block_keys[3 * slot + 2];
        // nbr_disp_real (map_local.cpp:84-89): +z,-z,+y,-y,+x,-x.  The reference stops at the first neighbour that is
        // still unknown; here the state of all six is fetched at once (one round trip instead of up to six) and the first
        // qualifying one is taken.
        int nvs[6], ncids[6];
        unsigned long long nbk[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            nvs[d] = -1;
            nbk[d] = 0;
            ncids[d] = 0;
            const double nx = wx + (d == 4 ? P.d_sub : (d == 5 ? -P.d_sub : 0.0));
            const double ny = wy + (d == 2 ? P.d_sub : (d == 3 ? -P.d_sub : 0.0));
            const double nz = wz + (d == 0 ? P.d_sub : (d == 1 ? -P.d_sub : 0.0));
            if (!mlm_inside_exp_bd(nx, ny, nz)) continue;
            int gx = gx0, gy = gy0, gz = gz0, ncid = cid;
            mlm_neighbor(P, d, gx, gy, gz, ncid);
            // a neighbour inside the voxel's own block needs no table probe; for the others a read-only probe: the
            // reference allocates a neighbour's block (allocate_ram) only if it gets that far, so blocks that do not
            // exist yet are created in the ordered pass below
            nbk[d] = mlm_pack_key(gx, gy, gz);
            ncids[d] = ncid;
            int ns = slot;
            if (!(gx == gx0 && gy == gy0 && gz == gz0)) {
                ns = mlm_block_find_k(P, nbk[d]);
                if (ns >= 0 && P.blk_collapsed[ns]) ns = -3; // released block: allocate_ram() is false
                if (ns == -1) ns = -4;                       // not there (yet): allocate in order
            }
            nvs[d] = ns >= 0 ? ns * P.cells + ncid : ns;
        }
        uint8_t n_occ[6];
        uint32_t n_miss[6];
        unsigned long long n_tau[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            n_occ[d] = 0;
            n_miss[d] = 0;
            n_tau[d] = 0;
            if (nvs[d] >= 0) {
                n_occ[d] = P.occ[nvs[d]];
                n_miss[d] = __hip_atomic_load(&P.vox_miss[nvs[d]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                n_tau[d] = P.vox_tau[nvs[d]];
            }
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            int nv = nvs[d];
            uint8_t o = n_occ[d];
            uint32_t km = n_miss[d];
            unsigned long long nt = n_tau[d];
            if (nv == -4) { // the block did not exist when probed: allocate_ram now (rare; its state is read afresh)
                const int ns = mlm_ex_block_slot(P, nbk[d]);
                if (ns < 0) continue;
                nv = ns * P.cells + ncids[d];
                o = P.occ[nv];
                km = __hip_atomic_load(&P.vox_miss[nv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nt = P.vox_tau[nv];
            }
            if (nv < 0) continue;
            // unknown at this moment: unknown after the hit phase, and not already turned free by an earlier miss
            const bool turns_earlier = (km != 0) && nt > tau;
            if (o == 'u' && !turns_earlier) {
                P.frnt[nv] = 1;
                break;
            }
        }
    }
}

// per touched voxel (the miss cell that was its first this frame): k misses (map_local.cpp:188-203); a cell that turns
// 'f' leaves the frontier
__device__ __forceinline__ void mlm_ex_apply_misses_body(const MlmDev &P) {
    const unsigned int n = P.ctr->n_ex_miss;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int v = P.ex_vox[i];
        if (v < 0) continue;
        const uint32_t k = P.vox_miss[v];
        P.vox_miss[v] = 0;
        P.vox_tau[v] = 0;
        float L = P.log_odds[v];
        uint8_t o = P.occ[v];
        const uint8_t o0 = o;
        mlm_apply_misses(P, L, o, k);
        P.log_odds[v] = L;
        P.occ[v] = o;
        if (o == 'f' && o0 != 'f') P.frnt[v] = 0;
    }
}
__device__ __forceinline__ void mlm_hand_back(const MlmDev &P, MlmCounters *host_ctr, MlmGlobal *host_g, unsigned int ticket, bool cleared); // (mlm_kernels_sector.h)
// host_ctr (a synchronous call's lone frame, explore_stage_bc_spec): the last workgroup hands the slot's counters, the map-wide flags and
// the ticket to the host (instead of two copies on the stream, 4 us each and a gap in front) — from HERE, one launch before the frame's
// last: the release scan that follows changes nothing the caller is handed (it marks uniform, frontier-free blocks as collapsed for the
// launches behind it in the stream), so it runs while the calling thread is already sampling its next frame.  A frame that ran leaves
// the slot's counters clear (launch_stage_a_sector); k_ex_release then finds them clear and runs, as it must.
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_apply_misses(const MlmDev P, MlmCounters *host_ctr, MlmGlobal *host_g, unsigned int ticket) {
    const bool ran = !mlm_ex_spec_skip(P);
    if (ran) mlm_ex_apply_misses_body(P);
    if (host_ctr) mlm_hand_back(P, host_ctr, host_g, ticket, ran);
}

// release scan (map_local.cpp:208-232): one workgroup per allocated block; blocks observed this frame whose frontier is
// empty and whose occupancy is uniform are collapsed (they stop accepting updates; element 0 answers queries)
__device__ __forceinline__ void mlm_ex_release_body(const MlmDev &P) {
    // A workgroup's blocks are blockIdx.x, blockIdx.x + gridDim.x, ...; few of them were observed this frame: the flags of MLM_BLOCK
    // candidates are read in ONE trip (a thread each) and the observed ones queued in LDS, so that a workgroup's time is one trip
    // for the flags plus the scans of its observed blocks — not a dependent load per candidate.
    __shared__ int s_bad;
    __shared__ unsigned int s_nq, s_q[MLM_BLOCK];
    const unsigned int n_blocks = min(P.g->n_blocks, (unsigned int)P.max_blocks); // read on the device: no host sync
    for (unsigned int b0 = blockIdx.x; b0 < n_blocks; b0 += gridDim.x * MLM_BLOCK) { // (uniform)
        __syncthreads();
        if (threadIdx.x == 0) s_nq = 0;
        __syncthreads();
        const unsigned long long bb = (unsigned long long)b0 + (unsigned long long)threadIdx.x * gridDim.x;
        if (bb < n_blocks && P.blk_observed[bb]) s_q[atomicAdd(&s_nq, 1u)] = (unsigned int)bb;
        __syncthreads();
        const unsigned int nq = s_nq;
        for (unsigned int k = 0; k < nq; ++k) {
            const unsigned int b = s_q[k];
            __syncthreads();
            if (threadIdx.x == 0) s_bad = 0;
            __syncthreads();
            if (!P.blk_collapsed[b]) {
                const uint8_t first = P.occ[(size_t)b * P.cells];
                int bad = 0;
                for (int c = threadIdx.x; c < P.cells; c += blockDim.x)
                    bad |= (P.frnt[(size_t)b * P.cells + c] != 0) | (P.occ[(size_t)b * P.cells + c] != first);
                if (bad) s_bad = 1;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                if (!P.blk_collapsed[b] && !s_bad) P.blk_collapsed[b] = 1;
                P.blk_observed[b] = 0;
            }
        }
    }
}
// (a lone frame's release scan runs behind the frame's hand-back: see k_ex_apply_misses.  A frame whose launches were held back —
// mlm_ex_spec_skip — kept its counters, and is skipped here as well.)
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_release(const MlmDev P) {
    if (mlm_ex_spec_skip(P)) return;
    mlm_ex_release_body(P);
}

// frontier read-out: (gx,gy,gz,cell) of every frontier cell
__global__ __launch_bounds__(MLM_BLOCK) void k_ex_export_frontier(const MlmDev P, unsigned int n_blocks, int32_t *out,
                                                                  unsigned int cap, unsigned int *counter) {
    const long long total = (long long)n_blocks * P.cells;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (!P.frnt[i]) continue;
        const unsigned int pos = atomicAdd(counter, 1u);
        if (pos < cap) {
            const int slot = (int)(i / P.cells);
            out[4 * (size_t)pos + 0] = P.block_keys[3 * slot];
            out[4 * (size_t)pos + 1] = P.block_keys[3 * slot + 1];
            out[4 * (size_t)pos + 2] = P.block_keys[3 * slot + 2];
            out[4 * (size_t)pos + 3] = (int)(i % P.cells);
        }
    }
}
