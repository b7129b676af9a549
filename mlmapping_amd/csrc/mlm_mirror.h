// mlm_mirror.h — host-resident mirror of the map for SMALL query batches.  Part of mlmap_hip.hip.
//
// The reference's queries are inline hash lookups on the host (include/mlmap.h:170-295: one observed_group_map.find per position,
// ~30 ns); planners call them position by position (README.md:4-27).  A kernel launch + two copies + a stream synchronisation per
// call (run_query's device path, ~20 us) is the wrong cost model for that pattern.  So the handle keeps a pinned host copy of the
// block planes (log-odds, occupancy, inflated occupancy, the "released" flag) laid out exactly like the device pool — block slot s
// of the pool is slot s of the mirror — plus a host table block key -> slot, and answers batches of up to MlmMirror::max_clean
// positions from it with the reference's own arithmetic (true FP64 divisions, the host libm's pow).
//
// Coherence: every entry point that changes the map records WHERE it may have changed it — a box of block indices (an integrate
// call: the awareness cylinder around t_wa, map_awareness.cpp:184 + map_local.cpp:151,180; setFree_map_in_bound: its box;
// inflate_map: its cube) or "anywhere" (imports) — and marks the mirror dirty.  The first small query after that drains the
// handle (a query observes the map of the last integrate call that returned) and runs k_mirror_refresh ONCE: the new blocks and
// the blocks inside the recorded boxes are copied by the device straight into the host planes; later queries cost a lock, a hash
// probe and a load.  Large batches keep the kernel (k_query); both paths read the same map state.
#pragma once
namespace {

inline uint32_t mir_mix(unsigned long long k) { // (the device's mlm_mix)
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}
inline bool mir_key_ok(int gx, int gy, int gz) {
    return !(((unsigned int)(gx + (1 << 20)) >> 21) || ((unsigned int)(gy + (1 << 20)) >> 21) || ((unsigned int)(gz + (1 << 20)) >> 21));
}
inline unsigned long long mir_pack(int gx, int gy, int gz) {
    return ((unsigned long long)((uint32_t)(gx + (1 << 20)) & 0x1FFFFFu) << 42) | ((unsigned long long)((uint32_t)(gy + (1 << 20)) & 0x1FFFFFu) << 21) |
           (unsigned long long)((uint32_t)(gz + (1 << 20)) & 0x1FFFFFu);
}

void mirror_mark_all(mlm_handle *h) {
    h->mir.dirty = true;
    h->mir.all = true;
    h->mir.n_box = 0;
}
void mirror_mark_box(mlm_handle *h, const int lo[3], const int hi[3]) {
    MlmMirror &M = h->mir;
    M.dirty = true;
    if (M.all) return;
    for (int k = 0; k < M.n_box; ++k) { // already covered?
        bool in = true;
        for (int a = 0; a < 3; ++a) in = in && lo[a] >= M.box_lo[k][a] && hi[a] <= M.box_hi[k][a];
        if (in) return;
    }
    if (M.n_box == MLM_MIRROR_BOXES) { // the list is full: one box around everything recorded so far
        for (int k = 1; k < M.n_box; ++k)
            for (int a = 0; a < 3; ++a) {
                M.box_lo[0][a] = std::min(M.box_lo[0][a], M.box_lo[k][a]);
                M.box_hi[0][a] = std::max(M.box_hi[0][a], M.box_hi[k][a]);
            }
        M.n_box = 1;
    }
    for (int a = 0; a < 3; ++a) {
        M.box_lo[M.n_box][a] = lo[a];
        M.box_hi[M.n_box][a] = hi[a];
    }
    M.n_box++;
}
// a box of world coordinates (+ one block of margin each side); anything not finite or beyond the key range: "anywhere"
void mirror_mark_world(mlm_handle *h, const double wlo[3], const double whi[3]) {
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        const double l = std::floor(wlo[a] / h->P.d_glb) - 1.0, u = std::floor(whi[a] / h->P.d_glb) + 1.0;
        if (!(l > -1048000.0 && u < 1048000.0 && l <= u)) { // (NaN fails)
            mirror_mark_all(h);
            return;
        }
        lo[a] = (int)l;
        hi[a] = (int)u;
    }
    mirror_mark_box(h, lo, hi);
}
// The frames described in the current slot set are about to be integrated: every hit and miss cell of a frame is an awareness cell,
// whose world position is its centre + t_wa (map_local.cpp:151,180) — inside the cylinder of radius nRho * dRho around t_wa between
// z_border_min and z_border_min + nZ * dZ.  (Frontier mode also flags neighbours of freed voxels and releases observed blocks:
// both within a voxel of that box, which the margin of one block covers.)
void mirror_mark_frames(mlm_handle *h, int n) {
    const MlmDev &P = h->P;
    const double R = P.nRho * P.dRho + P.d_sub;
    for (int j = 0; j < n && !h->mir.all; ++j) {
        const MlmFrame &F = h->slots[(size_t)(h->cur_set * h->lim.max_batch + j)].F;
        const double wlo[3] = {F.t_wa[0] - R, F.t_wa[1] - R, F.t_wa[2] + P.z_border_min - P.d_sub};
        const double whi[3] = {F.t_wa[0] + R, F.t_wa[1] + R, F.t_wa[2] + P.z_border_min + P.nZ * P.dZ + P.d_sub};
        mirror_mark_world(h, wlo, whi);
    }
    h->mir.dirty = true;
}

void mirror_free(mlm_handle *h) {
    MlmMirror &M = h->mir;
    if (M.lo) hipHostFree(M.lo);
    if (M.occ) hipHostFree(M.occ);
    if (M.infl) hipHostFree(M.infl);
    if (M.col) hipHostFree(M.col);
    if (M.keys) hipHostFree(M.keys);
    M.lo = nullptr;
    M.occ = M.infl = M.col = nullptr;
    M.keys = nullptr;
    M.cap = 0;
    M.n_known = 0;
    M.tk.clear();
    M.ts.clear();
    M.tmask = 0;
    mirror_mark_all(h);
}
constexpr unsigned int kMirrorGrid = 1024;
int mirror_reserve(mlm_handle *h, size_t blocks) {
    MlmMirror &M = h->mir;
    if (!M.stat) {
        HIPCHK(h, hipHostMalloc((void **)&M.stat, (2 + kMirrorGrid) * sizeof(unsigned int), hipHostMallocDefault));
        std::memset(M.stat, 0, (2 + kMirrorGrid) * sizeof(unsigned int));
    }
    if (blocks <= M.cap) return MLM_OK;
    mirror_free(h); // (the planes are refilled from the device: everything is dirty)
    const size_t cap = std::max<size_t>(256, blocks), C = (size_t)h->P.cells;
    hipError_t e = hipHostMalloc((void **)&M.lo, cap * C * sizeof(float), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&M.occ, cap * C, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&M.infl, cap * C, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&M.col, cap, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&M.keys, cap * 3 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        mirror_free(h);
        h->err = std::string("host mirror: ") + hipGetErrorString(e);
        M.alloc_failed = true; // (the queries fall back to the kernel path for good: run_query)
        M.enabled = false;
        return MLM_ERR_HIP;
    }
    M.cap = cap;
    size_t ht = 1024;
    while (ht < cap * 4) ht <<= 1;
    M.tk.assign(ht, MLM_HT_EMPTY);
    M.ts.assign(ht, -1);
    M.tmask = (uint32_t)(ht - 1);
    return MLM_OK;
}
// Bring the mirror up to date.  The caller holds the lock and has drained the handle.
int mirror_refresh(mlm_handle *h) {
    MlmMirror &M = h->mir;
    // (after a drain the host copy of the map-wide state usually knows the block count already; the kernel reports the exact one)
    size_t guess = std::min<size_t>(h->h_g ? h->h_g->n_blocks : 0u, (size_t)h->P.max_blocks);
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = mirror_reserve(h, M.cap >= guess ? M.cap : 2 * guess);
        if (rc) return rc;
        MlmMirrorBoxes B{};
        B.all = M.all ? 1 : 0;
        B.n = M.all ? 0 : M.n_box;
        for (int k = 0; k < B.n; ++k)
            for (int a = 0; a < 3; ++a) {
                B.lo[k][a] = M.box_lo[k][a];
                B.hi[k][a] = M.box_hi[k][a];
            }
        hipLaunchKernelGGL(k_mirror_refresh, dim3(kMirrorGrid), dim3(MLM_BLOCK), 0, h->stream, h->P, M.n_known, B, M.lo, M.occ, M.infl, M.col, M.keys,
                           (unsigned int)M.cap, M.stat);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const unsigned int nb = M.stat[0];
        if (nb > M.cap) { // more blocks than the planes hold: enlarge (everything is copied again) and repeat
            guess = nb;
            continue;
        }
        for (unsigned int i = 0; i < kMirrorGrid; ++i) M.n_copied += M.stat[2 + i];
        for (unsigned int b = M.n_known; b < nb; ++b) { // the new blocks' keys
            const unsigned long long key = mir_pack(M.keys[3 * (size_t)b], M.keys[3 * (size_t)b + 1], M.keys[3 * (size_t)b + 2]);
            uint32_t p = mir_mix(key) & M.tmask;
            while (M.tk[p] != MLM_HT_EMPTY) p = (p + 1) & M.tmask;
            M.tk[p] = key;
            M.ts[p] = (int)b;
        }
        M.n_known = nb;
        M.dirty = false;
        M.all = false;
        M.n_box = 0;
        M.n_refresh++;
        return MLM_OK;
    }
    h->err = "host mirror: the block count kept growing while nothing was in flight";
    return MLM_ERR_HIP;
}

// ---- the reference's query arithmetic on the host ------------------------------------------------------------------------
inline int mir_cvt_int(double v) { // x86 cvttsd2si (mlm_cvt_int)
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}
inline int mir_mul(int a, int b) { return (int)((unsigned int)a * (unsigned int)b); } // (wraps like the hardware; INT_MIN * n is UB in C++)
// get_global_idx / get_subbox_id, map_local.h:148-152,167-173 — two independent divisions per axis; a cell coordinate outside
// [0, n) maps to id 0 (operator[] default-inserts in the reference)
inline void mir_voxel_of(const MlmDev &P, double x, double y, double z, int g[3], int &cid) {
    g[0] = mir_cvt_int(std::floor(x / P.d_glb));
    g[1] = mir_cvt_int(std::floor(y / P.d_glb));
    g[2] = mir_cvt_int(std::floor(z / P.d_glb));
    const int cx = mir_cvt_int(std::floor(x / P.d_sub) - mir_mul(g[0], P.n));
    const int cy = mir_cvt_int(std::floor(y / P.d_sub) - mir_mul(g[1], P.n));
    const int cz = mir_cvt_int(std::floor(z / P.d_sub) - mir_mul(g[2], P.n));
    if (cx < 0 || cy < 0 || cz < 0 || cx >= P.n || cy >= P.n || cz >= P.n) cid = 0;
    else cid = cz * P.n * P.n + cy * P.n + cx;
}
inline int mir_find(const MlmMirror &M, int gx, int gy, int gz) {
    if (!mir_key_ok(gx, gy, gz) || !M.tmask) return -1;
    const unsigned long long key = mir_pack(gx, gy, gz);
    for (uint32_t p = mir_mix(key) & M.tmask;; p = (p + 1) & M.tmask) {
        const unsigned long long k = M.tk[p];
        if (k == key) return M.ts[p];
        if (k == MLM_HT_EMPTY) return -1;
    }
}
// getOccupancy, mlmap.h:170-193
inline int mir_occupancy(const mlm_handle *h, double x, double y, double z) {
    const MlmMirror &M = h->mir;
    int g[3], cid;
    mir_voxel_of(h->P, x, y, z, g, cid);
    const int slot = mir_find(M, g[0], g[1], g[2]);
    if (slot < 0) return MLM_UNKNOWN;
    if (M.col[slot]) cid = 0; // occupancy.size() == 1 -> occupancy[0], mlmap.h:183-184
    const uint8_t r = M.occ[(size_t)slot * h->P.cells + cid];
    return r == 'o' ? MLM_OCCUPIED : (r == 'f' ? MLM_FREE : MLM_UNKNOWN);
}
// logit_inv, mlmap.h:40: pow(10, x) / (1 + pow(10, x)) in double, narrowed by getOdd's float return
inline float mir_logit_inv(float L) {
    const double p = std::pow(10.0, (double)L);
    return (float)(p / (1 + p));
}
// getOdd(glb_id, subbox_id), mlmap.h:227-235
inline float mir_odd_at(const mlm_handle *h, int gx, int gy, int gz, int cid) {
    const MlmMirror &M = h->mir;
    const int slot = mir_find(M, gx, gy, gz);
    if (slot < 0) return 0.5f;
    if (M.col[slot]) cid = 0; // log_odds.size() == 1 -> log_odds[0], mlmap.h:221-222
    return mir_logit_inv(M.lo[(size_t)slot * h->P.cells + cid]);
}
// one step along direction dir of subbox_neighbors (map_local.cpp:77-120): order +z,-z,+y,-y,+x,-x
inline void mir_neighbor(const MlmDev &P, int dir, int g[3], int &cid) {
    int c[3];
    c[2] = cid / (P.n * P.n);
    c[1] = (cid - c[2] * P.n * P.n) / P.n;
    c[0] = cid - c[2] * P.n * P.n - c[1] * P.n;
    const int axis = 2 - dir / 2, step = (dir & 1) ? -1 : 1;
    c[axis] += step;
    if (c[axis] >= P.n) {
        g[axis] += 1;
        c[axis] = 0;
    } else if (c[axis] < 0) {
        g[axis] -= 1;
        c[axis] = P.n - 1;
    }
    cid = c[2] * P.n * P.n + c[1] * P.n + c[0];
}
// mode 0: getOccupancy  1: getOccupancy(pos, inflate)  2: getInflateOccupancy  3: getOdd  4: getOddGrad (k_query's modes)
void mirror_answer(const mlm_handle *h, int mode, const double *pos, int n, float inflate, int max_iter, void *out) {
    const MlmDev &P = h->P;
    const MlmMirror &M = h->mir;
    for (int i = 0; i < n; ++i) {
        const double x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
        if (mode == 0) {
            ((int8_t *)out)[i] = (int8_t)mir_occupancy(h, x, y, z);
        } else if (mode == 1) {
            // the 19-point stencil in the reference's order, mlmap.h:142-169; Vec3(+-inflate) promotes the float to double
            const double f = inflate;
            static const int8_t o[19][3] = {{0, 0, 0},  {0, 0, 1},   {0, 0, -1}, {0, 1, 0},   {0, -1, 0}, {1, 0, 0},  {-1, 0, 0},
                                            {-1, 1, 0}, {-1, -1, 0}, {1, 1, 0},  {1, -1, 0},  {0, -1, 1}, {0, -1, -1}, {0, 1, 1},
                                            {0, 1, -1}, {-1, 0, 1},  {-1, 0, -1}, {1, 0, 1},  {1, 0, -1}};
            int res = MLM_FREE;
            for (int k = 0; k < 19 && res == MLM_FREE; ++k)
                if (mir_occupancy(h, x + (o[k][0] ? (o[k][0] > 0 ? f : -f) : 0.0), y + (o[k][1] ? (o[k][1] > 0 ? f : -f) : 0.0),
                                  z + (o[k][2] ? (o[k][2] > 0 ? f : -f) : 0.0)) == MLM_OCCUPIED)
                    res = MLM_OCCUPIED;
            ((int8_t *)out)[i] = (int8_t)res;
        } else if (mode == 2) { // getInflateOccupancy, mlmap.h:195-211
            int g[3], cid;
            mir_voxel_of(P, x, y, z, g, cid);
            const int slot = mir_find(M, g[0], g[1], g[2]);
            int res = MLM_UNKNOWN;
            if (slot >= 0 && !M.col[slot] && M.infl[(size_t)slot * P.cells + cid] == 'o') res = MLM_OCCUPIED;
            ((int8_t *)out)[i] = (int8_t)res;
        } else if (mode == 3) {
            int g[3], cid;
            mir_voxel_of(P, x, y, z, g, cid);
            ((float *)out)[i] = mir_odd_at(h, g[0], g[1], g[2], cid);
        } else { // getOddGrad, mlmap.h:237-295
            int g[3], cid;
            mir_voxel_of(P, x, y, z, g, cid);
            float min_odd = mir_odd_at(h, g[0], g[1], g[2], cid);
            const float ori_odd = min_odd;
            int ng[6][3], ncid[6], mg[3] = {0, 0, 0}, mcid = 0;
            bool flag = false;
            for (int iter = 0; iter < max_iter && !flag; ++iter)
                for (int d = 0; d < 6; ++d) {
                    if (iter == 0) {
                        ng[d][0] = g[0], ng[d][1] = g[1], ng[d][2] = g[2];
                        ncid[d] = cid;
                    }
                    mir_neighbor(P, d, ng[d], ncid[d]); // (keeps searching along the original direction, mlmap.h:267-268)
                    const float tmp = mir_odd_at(h, ng[d][0], ng[d][1], ng[d][2], ncid[d]);
                    if (tmp < min_odd) {
                        min_odd = tmp;
                        mg[0] = ng[d][0], mg[1] = ng[d][1], mg[2] = ng[d][2];
                        mcid = ncid[d];
                        flag = true;
                    }
                }
            double r[3] = {0.0, 0.0, 0.0};
            if (flag) { // subbox_id2xyz_glb_vec, map_local.h:208-213
                const int cz = mcid / (P.n * P.n), cy = (mcid - cz * P.n * P.n) / P.n, cx = mcid - cz * P.n * P.n - cy * P.n;
                const double s = (double)(ori_odd - min_odd);
                r[0] = ((mg[0] * P.d_glb + cx * P.d_sub + P.d_sub_half) - x) * s;
                r[1] = ((mg[1] * P.d_glb + cy * P.d_sub + P.d_sub_half) - y) * s;
                r[2] = ((mg[2] * P.d_glb + cz * P.d_sub + P.d_sub_half) - z) * s;
            }
            ((double *)out)[3 * (size_t)i] = r[0];
            ((double *)out)[3 * (size_t)i + 1] = r[1];
            ((double *)out)[3 * (size_t)i + 2] = r[2];
        }
    }
}

// Is this batch answered on the host?  Small batches are a planner sampling positions one by one; a large batch after the map
// changed is cheaper as one kernel than a refresh plus a host loop.
bool mirror_wanted(const mlm_handle *h, int mode, int n, int max_iter) {
    const MlmMirror &M = h->mir;
    if (!M.enabled) return false;
    long long work = n; // lookups, roughly
    if (mode == 1) work *= 19;
    if (mode == 4) work *= 1 + 6 * (long long)std::min(max_iter, 64);
    return work <= (M.dirty ? M.max_dirty : M.max_clean);
}
// drain + refresh if the map changed since the mirror was filled; the caller holds the lock
int mirror_sync(mlm_handle *h) {
    if (!h->mir.dirty) return MLM_OK;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc) return rc;
    return mirror_refresh(h);
}

} // namespace
