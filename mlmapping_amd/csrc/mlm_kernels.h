// mlm_kernels.h — HIP kernels of the per-frame map update (gfx950, wave64).
//
// Stage A  (awareness_map_cylindrical::input_pc_pose, map_awareness.cpp:173-282)
//   k_bin_points      point -> (rho,phi,z) bin, noise-spread hit contributions, de-duplicated ray walk
//   k_collect_hits    dense sweep of the hit scratch -> unique-hit list + per-cell contribution segments
//   k_scatter_contribs / k_noisy_or   point-order replay of the float noisy-OR chain -> odd, logit
// Stage B  (iteration order of hit_idx_odds_hashmap, i.e. libstdc++ _Hashtable list order)
//   k_bucket_min / k_make_keys (+ rank kernels on rehash frames)
// Stage C  (local_map_cartesian::input_pc_pose_direct, map_local.cpp:143-237)
//   k_hits_to_voxels / k_apply_hits / k_misses_to_voxels / k_apply_misses
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
// neighbour), 2d (-d neighbour): the order update_hits inserts them (map_awareness.cpp:146-168).
__device__ __forceinline__ void mlm_count_contribution(const MlmDev &P, int cell, uint32_t t) {
    atomicMin(&P.hit_t[cell], t);
    atomicAdd(&P.hit_cnt[cell], 1u);
}

// Enumerate the contributions of one in-range point (update_hits, map_awareness.cpp:135-171).
template <class F> __device__ __forceinline__ void mlm_for_each_contribution(const MlmDev &P, int rho, int phi, int zi,
                                                                            uint32_t t0, F &&f) {
    const int c0 = zi * P.nRhoPhi + phi * P.nRho + rho;
    f(c0, t0);
    const float s3 = P.sigma3[rho];
    if (!(1.0f < s3)) return;
    const double slope = (rho > 0) ? (zi - P.zc) / (rho * 1.0) : 0.0; // raycasting_z_over_rho, map_awareness.cpp:64-71
    for (int d = 1; (float)d < s3 && (rho + d < P.nRho) && d <= MLM_DIFF_RANGE; ++d) {
        int rz = mlm_cvt_int(round(zi + (d * slope)));
        if (0 <= rz && rz < P.nZ) f(rz * P.nRhoPhi + phi * P.nRho + rho + d, t0 + 2 * d - 1);
        rz = mlm_cvt_int(round(zi - (d * slope)));
        if (0 <= rz && rz < P.nZ && rho - d >= 0) f(rz * P.nRhoPhi + phi * P.nRho + rho - d, t0 + 2 * d);
    }
}

// ray walk, map_awareness.cpp:266-274: r = rho-1 .. 1, z' = round(z - (rho-r)*slope).  Bits of one (phi,z') row
// that fall into the same 32-bit word are OR-ed in registers and flushed with one atomic.
__device__ __forceinline__ void mlm_walk_ray(const MlmDev &P, int rho, int phi, int z, double slope) {
    int cur_w = -1;
    uint32_t cur_m = 0;
    for (int r = rho - 1; r > 0; --r) {
        const int diff_r = rho - r;
        const int zr = mlm_cvt_int(round(z - (diff_r * slope)));
        if (0 <= zr && zr < P.nZ) {
            const int w = (zr * P.nPhi + phi) * P.RW + (r >> 5);
            if (w != cur_w) {
                if (cur_m) atomicOr(&P.miss_bits[cur_w], cur_m);
                cur_w = w;
                cur_m = 0;
            }
            cur_m |= 1u << (r & 31);
        }
    }
    if (cur_m) atomicOr(&P.miss_bits[cur_w], cur_m);
}

// MODE 0: dense depth image, 1: indexed depth pixels, 2: explicit sensor-frame points
template <int MODE>
__global__ __launch_bounds__(MLM_BLOCK) void k_bin_points(const MlmDev P, const MlmFrame F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    double xs, ys, zs;
    if (MODE == 2) {
        xs = F.pts[3 * (size_t)i + 0];
        ys = F.pts[3 * (size_t)i + 1];
        zs = F.pts[3 * (size_t)i + 2];
    } else {
        const int pix = (MODE == 1) ? F.pix[i] : i;
        const int v = pix / F.width;
        const int u = pix - v * F.width;
        const uint16_t raw = F.img[(size_t)v * F.row_stride + u];
        if (raw == 0) { // mlmap.cpp:338-341
            P.pt_cell[i] = -1;
            return;
        }
        // mlmap.cpp:329,344-346: (size_t u - float cx_) is a float subtraction, the rest is double
        const double depth = raw * P.inv_factor;
        xs = ((float)u - P.cx) * depth / P.fx;
        ys = ((float)v - P.cy) * depth / P.fy;
        zs = depth;
    }
    atomicAdd(&P.ctr->n_points, 1u);

    // p_l = T_ls * p_s (map_awareness.cpp:222; se3.cpp:91-95)
    double x, y, z;
    mlm_quat_rot(F.q_ls, xs, ys, zs, x, y, z);
    x = x + F.t_ls[0];
    y = y + F.t_ls[1];
    z = z + F.t_ls[2];

    int rho, phi, zi;
    bool can_do_cast;
    const bool inside = mlm_bin_point(P, x, y, z, rho, phi, zi, can_do_cast);
    const uint32_t t0 = (uint32_t)i * MLM_TIME_SLOTS;
    bool walk = false;
    double slope = 0;
    int c0 = -1;

    if (inside) {
        c0 = zi * P.nRhoPhi + phi * P.nRho + rho;
        slope = (rho > 0) ? (zi - P.zc) / (rho * 1.0) : 0.0;
        mlm_for_each_contribution(P, rho, phi, zi, t0, [&](int cell, uint32_t t) { mlm_count_contribution(P, cell, t); });
        if (P.visibility) {
            // every point of one (rho,phi,z) cell casts the identical ray: only the first one walks it
            const uint32_t bit = 1u << (c0 & 31);
            const uint32_t old = atomicOr(&P.start_bits[c0 >> 5], bit);
            walk = (old & bit) == 0;
        }
    } else if (can_do_cast && P.visibility) {
        // map_awareness.cpp:249-265: slope from the point's own (out-of-range) indices, clamp to the border
        slope = (rho > 0) ? (zi - P.zc) / (rho * 1.0) : 0.0;
        if (rho >= P.nRho) {
            zi = mlm_cvt_int(round(zi - ((rho - P.nRho + 1) * slope)));
            rho = P.nRho - 1;
        }
        walk = true;
    }
    P.pt_cell[i] = c0;
    if (!(can_do_cast && P.visibility)) atomicAdd(&P.ctr->n_oor, 1u); // map_awareness.cpp:277-278
    if (walk) mlm_walk_ray(P, rho, phi, zi, slope);
}

// One thread per awareness cell: emit the unique hit cells, carve a segment of `contrib` for each, and reset
// hit_t for the next frame (hit_cnt is consumed — and thereby zeroed — by k_scatter_contribs).
__global__ __launch_bounds__(MLM_BLOCK) void k_collect_hits(const MlmDev P) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t = MLM_EMPTY_T;
    if (c < P.nCells) t = P.hit_t[c];
    const bool has = (t != MLM_EMPTY_T);
    const unsigned int pos = mlm_wave_append(&P.ctr->u_hit, has);
    const uint32_t cnt = has ? P.hit_cnt[c] : 0u;
    // wave-inclusive prefix sum of cnt, one atomic per wave for the segment space
    const int lane = threadIdx.x & 63;
    uint32_t incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t wave_base = 0;
    if (lane == 63 && total) wave_base = atomicAdd(&P.ctr->n_contrib, total);
    wave_base = __shfl(wave_base, 63, 64);
    if (!has) return;
    const uint32_t base = wave_base + incl - cnt;
    P.hit_t[c] = MLM_EMPTY_T;
    P.seg_base[c] = base;
    P.hl_cell[pos] = (uint32_t)c;
    P.hl_t[pos] = t;
    P.hl_vt[pos] = t;
    P.hl_base[pos] = base;
    P.hl_cnt[pos] = cnt;
}

// Second sweep over the points: write each contribution's insertion time into its cell's segment.
__global__ __launch_bounds__(MLM_BLOCK) void k_scatter_contribs(const MlmDev P, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = P.pt_cell[i];
    if (c0 < 0) return;
    int rho, phi, zi;
    mlm_cell_rpz(P, (uint32_t)c0, rho, phi, zi);
    mlm_for_each_contribution(P, rho, phi, zi, (uint32_t)i * MLM_TIME_SLOTS, [&](int cell, uint32_t t) {
        const uint32_t k = atomicSub(&P.hit_cnt[cell], 1u) - 1u;
        const uint32_t slot = P.seg_base[cell] + k;
        if (slot < P.contrib_cap) P.contrib[slot] = t;
    });
}

__device__ __forceinline__ uint32_t mlm_wave_min_u32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// One wave per unique hit cell: replay update_odds_hashmap (map_awareness.h:147-154) over the cell's contributions
// in insertion-time order — the float noisy-OR chain is not associative, so the order is part of the result.
// p == 1.0f is absorbing (1-(1-1)(1-a) == 1), which ends long chains after a few steps.
__global__ __launch_bounds__(MLM_BLOCK) void k_noisy_or(const MlmDev P) {
    const unsigned int n_cells = P.ctr->u_hit;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned int w = wave; w < n_cells; w += n_waves) {
        const uint32_t cell = P.hl_cell[w];
        const uint32_t base = P.hl_base[w];
        const uint32_t n = P.hl_cnt[w];
        const int rho_c = (int)(cell % (uint32_t)P.nRho);
        // keys of the first 64 contributions stay in registers
        const uint32_t k0 = ((uint32_t)lane < n) ? P.contrib[base + lane] : 0xFFFFFFFFu;
        float p = 0.0f;
        bool first = true;
        long long last = -1;
        for (;;) {
            uint32_t m = ((long long)k0 > last) ? k0 : 0xFFFFFFFFu;
            for (uint32_t j = 64 + lane; j < n; j += 64) {
                const uint32_t k = P.contrib[base + j];
                if ((long long)k > last && k < m) m = k;
            }
            m = mlm_wave_min_u32(m);
            if (m == 0xFFFFFFFFu) break;
            // decode: sub 0 -> centre; 2d-1 -> "+d" neighbour of a point at rho_c-d; 2d -> "-d" neighbour of rho_c+d
            const int sub = (int)(m % MLM_TIME_SLOTS);
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
            const float a = P.odds_table[row * P.nRho + rho_s];
            if (first) {
                p = a;
                first = false;
            } else {
                p = 1 - (1 - p) * (1 - a);
            }
            last = (long long)m;
            if (p == 1.0f) break;
        }
        if (lane == 0) {
            // logit macro, map_local.h:8, on a float: log10f(x / (1 - x))
            const float ratio = p / (1 - p);
            P.hl_odd[w] = p;
            P.hl_inc[w] = (float)log10((double)ratio);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B — iteration order of std::unordered_map<Vec3I,float,VectorHasher> (libstdc++ _Hashtable):
// the node list is a sequence of bucket chains; a bucket that becomes non-empty later sits nearer the head, and
// inside a chain later insertions sit nearer the head.  So "x is visited before y" <=> (first-insert time of x's
// bucket, insert time of x) > (… y) lexicographically.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MLM_BLOCK) void k_bucket_min(const MlmDev P, unsigned int n, unsigned long long n_bkt,
                                                          unsigned int arr_limit, int use_arr) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (use_arr && P.hl_arr[i] >= arr_limit) return;
    int rho, phi, z;
    mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
    const unsigned long long b = mlm_hash_rpz(rho, phi, z) % n_bkt;
    atomicMin(&P.bkt_first[b], P.hl_vt[i]);
}
// final = 1: hl_key = (bucket_first<<32)|vt for every element.
// final = 0: sort key for the re-densify pass: ~key for members (ascending sort = list order), all-ones otherwise.
__global__ __launch_bounds__(MLM_BLOCK) void k_make_keys(const MlmDev P, unsigned int n, unsigned long long n_bkt,
                                                         unsigned int arr_limit, int use_arr, int final_pass,
                                                         unsigned long long *sort_keys, uint32_t *sort_vals) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
// after sorting (t, idx): arrival index of each element
__global__ __launch_bounds__(MLM_BLOCK) void k_assign_rank(const MlmDev P, unsigned int n, const uint32_t *sorted_idx,
                                                           unsigned int limit, int to_arr) {
    const unsigned int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || r >= limit) return;
    const uint32_t i = sorted_idx[r];
    P.hl_vt[i] = r;
    if (to_arr) P.hl_arr[i] = r;
}
__global__ __launch_bounds__(MLM_BLOCK) void k_time_keys(const MlmDev P, unsigned int n, unsigned long long *sort_keys,
                                                         uint32_t *sort_vals) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sort_keys[i] = P.hl_t[i];
    sort_vals[i] = i;
}

// ---------------------------------------------------------------------------------------------------------------
// Stage C
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MLM_BLOCK) void k_hits_to_voxels(const MlmDev P, const MlmFrame F, unsigned int n) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int rho, phi, z;
    mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
    double wx, wy, wz;
    mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz);
    int gx, gy, gz, cid;
    mlm_voxel_of(P, wx, wy, wz, gx, gy, gz, cid);
    const int slot = mlm_block_find_or_insert(P, gx, gy, gz);
    if (slot < 0) {
        P.hl_vox[i] = -1;
        P.hl_next[i] = -2;
        return;
    }
    const int v = slot * P.cells + cid;
    P.hl_vox[i] = v;
    P.hl_next[i] = atomicExch(&P.vox_head[v], (int)i);
}

// The first-inserted node of each voxel list (next == -1) owns the voxel: it replays the voxel's hit
// contributions in the reference's iteration order (descending hl_key) — map_local.cpp:157-171.
__global__ __launch_bounds__(MLM_BLOCK) void k_apply_hits(const MlmDev P, unsigned int n) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (P.hl_next[i] != -1) return;
    const int v = P.hl_vox[i];
    const int head = P.vox_head[v];
    float L = P.log_odds[v];
    uint8_t o = P.occ[v];
    unsigned long long last = ~0ull;
    for (;;) {
        int best = -1;
        unsigned long long bestkey = 0;
        for (int j = head; j >= 0; j = P.hl_next[j]) {
            const unsigned long long k = P.hl_key[j];
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
    P.log_odds[v] = L;
    P.occ[v] = o;
    P.vox_head[v] = -1;
}

// One thread per word of the miss bit mask: count the frame's misses per voxel (their order is irrelevant:
// every miss adds the same constant, map_local.cpp:188-192) and clear the word.
__global__ __launch_bounds__(MLM_BLOCK) void k_misses_to_voxels(const MlmDev P, const MlmFrame F) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = 0;
    if (w < P.nMissWords) bits = P.miss_bits[w];
    if (bits == 0) return;
    P.miss_bits[w] = 0;
    atomicAdd(&P.ctr->u_miss, (unsigned int)__popc(bits));
    const int row = w / P.RW;
    const int wi = w - row * P.RW;
    const int z = row / P.nPhi;
    const int phi = row - z * P.nPhi;
    while (bits) {
        const int b = __ffs((int)bits) - 1;
        bits &= bits - 1;
        const int rho = wi * 32 + b;
        if (P.record_awareness) {
            const unsigned int pos = atomicAdd(&P.ctr->n_miss_list, 1u);
            P.ml_cell[pos] = (uint32_t)(z * P.nRhoPhi + phi * P.nRho + rho);
        }
        double wx, wy, wz;
        mlm_cell_center_w(P, F.t_wa, rho, phi, z, wx, wy, wz);
        int gx, gy, gz, cid;
        mlm_voxel_of(P, wx, wy, wz, gx, gy, gz, cid);
        const int slot = mlm_block_find_or_insert(P, gx, gy, gz);
        if (slot < 0) continue;
        const int v = slot * P.cells + cid;
        const uint32_t old = atomicAdd(&P.vox_miss[v], 1u);
        if (old == 0) {
            const unsigned int pos = atomicAdd(&P.ctr->n_miss_vox, 1u);
            P.miss_vox[pos] = v;
        }
    }
}
// map_local.cpp:188-203, k times
__global__ __launch_bounds__(MLM_BLOCK) void k_apply_misses(const MlmDev P) {
    const unsigned int n = P.ctr->n_miss_vox;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int v = P.miss_vox[i];
        const uint32_t k = P.vox_miss[v];
        P.vox_miss[v] = 0;
        float L = P.log_odds[v];
        uint8_t o = P.occ[v];
        for (uint32_t j = 0; j < k; ++j) {
            if (L >= P.lo_min) {
                L = L + P.lo_miss;
                L = L < P.lo_min ? P.lo_min : L;
            }
            if (L < P.lo_sh && o != 'f') o = 'f';
        }
        P.log_odds[v] = L;
        P.occ[v] = o;
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
        const double o[19][3] = {{0, 0, 0},   {0, 0, f},   {0, 0, -f}, {0, f, 0},   {0, -f, 0}, {f, 0, 0},  {-f, 0, 0},
                                 {-f, f, 0},  {-f, -f, 0}, {f, f, 0},  {f, -f, 0},  {0, -f, f}, {0, -f, -f}, {0, f, f},
                                 {0, f, -f},  {-f, 0, f},  {-f, 0, -f}, {f, 0, f},  {f, 0, -f}};
        int res = 1;
        for (int k = 0; k < 19; ++k)
            if (mlm_get_occupancy(P, x + o[k][0], y + o[k][1], z + o[k][2]) == 0) {
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
        if (slot >= 0 && P.infl[(size_t)slot * P.cells + cid] == 'o') res = 0;
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
    if (slot < 0) return;
    P.occ[(size_t)slot * P.cells + cid] = 'f';
    P.log_odds[(size_t)slot * P.cells + cid] = 0.0f;
}

__global__ __launch_bounds__(MLM_BLOCK) void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(MLM_BLOCK) void k_fill_u8(uint8_t *p, uint8_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
