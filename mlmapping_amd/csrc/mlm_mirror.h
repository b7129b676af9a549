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

void mirror_mark_all(mlm_handle *h) {
    h->mir.dirty = true;
    h->mir.boxes.mark_all();
}
void mirror_mark_box(mlm_handle *h, const int lo[3], const int hi[3]) {
    h->mir.dirty = true;
    h->mir.boxes.mark(lo, hi);
}
// a box of world coordinates (+ one block of margin each side); anything not finite or beyond the key range: "anywhere"
void mirror_mark_world(mlm_handle *h, const double wlo[3], const double whi[3]) {
    h->mir.dirty = true;
    h->mir.boxes.mark_world(wlo, whi, h->P.d_glb);
}
// The frames described in the current slot set are about to be integrated: every hit and miss cell of a frame is an awareness cell,
// whose world position is its centre + t_wa (map_local.cpp:151,180) — inside the cylinder of radius nRho * dRho around t_wa between
// z_border_min and z_border_min + nZ * dZ.  (Frontier mode also flags neighbours of freed voxels and releases observed blocks:
// both within a voxel of that box, which the margin of one block covers.)
void mirror_mark_frames(mlm_handle *h, int n) {
    const MlmDev &P = h->P;
    const double R = P.nRho * P.dRho + P.d_sub;
    for (int j = 0; j < n && !h->mir.boxes.all; ++j) {
        const MlmFrame &F = h->slots[(size_t)(h->cur_set * h->lim.max_batch + j)].F;
        const double wlo[3] = {F.t_wa[0] - R, F.t_wa[1] - R, F.t_wa[2] + P.z_border_min - P.d_sub};
        const double whi[3] = {F.t_wa[0] + R, F.t_wa[1] + R, F.t_wa[2] + P.z_border_min + P.nZ * P.dZ + P.d_sub};
        mirror_mark_world(h, wlo, whi);
    }
    h->mir.dirty = true;
}

void mirror_free(mlm_handle *h) {
    MlmMirror &M = h->mir;
    if (M.eager_pending) { // (a refresh kernel may still be writing the planes)
        hipEventSynchronize(M.eager_ev);
        M.eager_pending = false;
    }
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
    M.view.table_clear();
    M.view.lo = nullptr;
    M.view.occ = M.view.infl = M.view.col = nullptr;
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
    const size_t cap = std::max<size_t>(256, blocks), C = (size_t)h->P.cells;
    if (cap * (C * 6 + 13) > M.max_bytes) { // (the map has outgrown what the caller lets the mirror pin: small queries run as kernels from now on)
        mirror_free(h);
        M.over_limit = true;
        M.enabled = false;
        return MLM_OK;
    }
    // New planes; what the old ones hold of blocks [0, n_known) is still valid wherever no box is pending, so it is copied on the
    // host and only new or changed blocks cross the link (the pending boxes stay as they are).
    float *lo = nullptr;
    uint8_t *occ = nullptr, *infl = nullptr, *col = nullptr;
    int *keys = nullptr;
    hipError_t e = hipHostMalloc((void **)&lo, cap * C * sizeof(float), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&occ, cap * C, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&infl, cap * C, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&col, cap, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&keys, cap * 3 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (lo) hipHostFree(lo);
        if (occ) hipHostFree(occ);
        if (infl) hipHostFree(infl);
        if (col) hipHostFree(col);
        if (keys) hipHostFree(keys);
        mirror_free(h);
        h->err = std::string("host mirror: ") + hipGetErrorString(e);
        M.alloc_failed = true; // (the queries fall back to the kernel path for good: run_query)
        M.enabled = false;
        return MLM_ERR_HIP;
    }
    const bool carry = M.cap > 0 && M.n_known > 0 && !M.boxes.all && !M.eager_pending;
    const unsigned int known = carry ? M.n_known : 0u;
    if (carry) {
        std::memcpy(lo, M.lo, (size_t)known * C * sizeof(float));
        std::memcpy(occ, M.occ, (size_t)known * C);
        std::memcpy(infl, M.infl, (size_t)known * C);
        std::memcpy(col, M.col, known);
        std::memcpy(keys, M.keys, (size_t)known * 3 * sizeof(int));
    }
    const mlm_host::DirtyBoxes boxes = M.boxes;
    mirror_free(h); // (marks everything stale: undone below where the old contents were carried over)
    M.lo = lo, M.occ = occ, M.infl = infl, M.col = col, M.keys = keys;
    M.cap = cap;
    M.view.table_reset(cap);
    if (carry) {
        for (unsigned int b = 0; b < known; ++b) M.view.table_insert(keys[3 * (size_t)b], keys[3 * (size_t)b + 1], keys[3 * (size_t)b + 2], (int)b);
        M.n_known = known;
        M.boxes = boxes;
    }
    M.view.d_sub = h->P.d_sub, M.view.d_glb = h->P.d_glb, M.view.d_sub_half = h->P.d_sub_half;
    M.view.n = h->P.n, M.view.cells = h->P.cells;
    M.view.lo = M.lo, M.view.occ = M.occ, M.view.infl = M.infl, M.view.col = M.col;
    return MLM_OK;
}
// The refresh kernel for the boxes recorded so far, on the main stream (the planes hold M.cap blocks).
int mirror_launch(mlm_handle *h) {
    MlmMirror &M = h->mir;
    MlmMirrorBoxes B{};
    B.all = M.boxes.all ? 1 : 0;
    B.n = M.boxes.all ? 0 : M.boxes.n;
    for (int k = 0; k < B.n; ++k)
        for (int a = 0; a < 3; ++a) {
            B.lo[k][a] = M.boxes.lo[k][a];
            B.hi[k][a] = M.boxes.hi[k][a];
        }
    hipLaunchKernelGGL(k_mirror_refresh, dim3(kMirrorGrid), dim3(MLM_BLOCK), 0, h->stream, h->P, M.n_known, B, M.lo, M.occ, M.infl, M.col, M.keys,
                       (unsigned int)M.cap, M.stat);
    HIPCHK(h, hipGetLastError());
    return MLM_OK;
}
// ... and what it left behind, once it has finished: the new blocks' keys go into the host table.  false: there are more blocks than
// the planes hold — nothing is taken in, the planes must be enlarged and everything copied again.
bool mirror_collect(mlm_handle *h, unsigned int *n_blocks_seen = nullptr) {
    MlmMirror &M = h->mir;
    const unsigned int nb = M.stat[0];
    if (n_blocks_seen) *n_blocks_seen = nb;
    if (nb > M.cap) return false;
    for (unsigned int i = 0; i < kMirrorGrid; ++i) M.n_copied += M.stat[2 + i];
    for (unsigned int b = M.n_known; b < nb; ++b) // the new blocks' keys
        M.view.table_insert(M.keys[3 * (size_t)b], M.keys[3 * (size_t)b + 1], M.keys[3 * (size_t)b + 2], (int)b);
    M.n_known = nb;
    M.n_refresh++;
    return true;
}
// Bring the mirror up to date.  The caller holds the lock and has drained the handle; no eager refresh is pending.
int mirror_refresh(mlm_handle *h) {
    MlmMirror &M = h->mir;
    // (the geometry first: an empty map has no planes yet, but getOddGrad walks neighbours of absent blocks all the same)
    M.view.d_sub = h->P.d_sub, M.view.d_glb = h->P.d_glb, M.view.d_sub_half = h->P.d_sub_half;
    M.view.n = h->P.n, M.view.cells = h->P.cells;
    // (after a drain the host copy of the map-wide state usually knows the block count already; the kernel reports the exact one)
    size_t guess = std::min<size_t>(h->h_g ? h->h_g->n_blocks : 0u, (size_t)h->P.max_blocks);
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = mirror_reserve(h, M.cap >= guess ? M.cap : 2 * guess);
        if (rc) return rc;
        if ((rc = mirror_launch(h))) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        unsigned int nb = 0;
        if (!mirror_collect(h, &nb)) { // more blocks than the planes hold: enlarge (everything is copied again) and repeat
            guess = nb;
            continue;
        }
        M.dirty = false;
        M.boxes.clear();
        return MLM_OK;
    }
    h->err = "host mirror: the block count kept growing while nothing was in flight";
    return MLM_ERR_HIP;
}
// A synchronous integrate call is about to return (nothing in flight, the map-wide state on the host): launch the refresh now if
// queries have been following the integrate calls (MlmMirror::eager_on).  Never fails the call: whatever goes wrong here is left to
// the first query's own refresh.
void mirror_eager(mlm_handle *h) {
    MlmMirror &M = h->mir;
    if (!M.enabled || M.alloc_failed || !M.dirty || !M.eager_on) return;
    if (h->async_mode || !h->pending.empty() || !h->ex_q.empty() || h->wait_ticket || h->timing) return;
    if (M.eager_pending || M.n_host_queries == M.q_at_eager) { // the previous eager refresh found no taker: stop until a query asks again
        M.eager_on = false;
        return;
    }
    const size_t nb = std::min<size_t>(h->h_g ? h->h_g->n_blocks : 0u, (size_t)h->P.max_blocks);
    if (M.cap == 0 || nb > M.cap || M.boxes.all) return; // (the planes must grow, or everything is stale: the query's refresh does that)
    if (!M.eager_ev && hipEventCreateWithFlags(&M.eager_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        M.eager_ev = nullptr;
        return;
    }
    if (mirror_launch(h) != MLM_OK || hipEventRecord(M.eager_ev, h->stream) != hipSuccess) {
        (void)hipGetLastError();
        hipStreamSynchronize(h->stream); // (whatever did get launched has finished before anybody touches the planes again)
        mirror_mark_all(h);
        return;
    }
    M.eager_pending = true;
    M.q_at_eager = M.n_host_queries;
    M.n_eager++;
    M.dirty = false; // (what the boxes named is on its way; a later change marks the mirror again)
    M.boxes.clear();
}
// take in a pending eager refresh (wait for it if it is still running)
int mirror_finish_eager(mlm_handle *h) {
    MlmMirror &M = h->mir;
    if (!M.eager_pending) return MLM_OK;
    M.eager_pending = false;
    hipError_t e = hipErrorNotReady;
    for (int spins = 0; spins < 4096 && e == hipErrorNotReady; ++spins) e = hipEventQuery(M.eager_ev); // (usually fired long ago)
    if (e == hipErrorNotReady) e = hipEventSynchronize(M.eager_ev);
    if (e != hipSuccess || !mirror_collect(h)) { // (a failed wait, or more blocks than the planes hold: everything again, by the regular refresh)
        (void)hipGetLastError();
        mirror_mark_all(h);
    }
    return MLM_OK;
}

// (the query arithmetic itself — get_global_idx, getOccupancy, getOdd, getOddGrad on the mirrored planes — is pure host code:
// mlm_mapview.h, tested on the CPU against the oracle)
void mirror_answer(const mlm_handle *h, int mode, const double *pos, int n, float inflate, int max_iter, void *out) {
    h->mir.view.answer(mode, pos, n, inflate, max_iter, out);
}
inline float mir_odd_at(const mlm_handle *h, int gx, int gy, int gz, int cid) { return h->mir.view.odd_at(gx, gy, gz, cid); }

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
    if (h->mir.eager_pending) {
        HIPCHK(h, hipSetDevice(h->device));
        mirror_finish_eager(h);
    }
    if (!h->mir.dirty) return MLM_OK;
    h->mir.eager_on = true; // (a query found the mirror stale: queries do follow the map's changes)
    HIPCHK(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc) return rc;
    return mirror_refresh(h);
}

} // namespace
