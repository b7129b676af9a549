// mlm_resources.h — device memory of a handle: staging buffers, the block table + pool and its growth, the column table's widening,
// one frame slot's scratch; the query launcher.  Part of mlmap_hip.hip.
#pragma once
namespace {

int ensure_img(mlm_handle *h, MlmSlot &S, size_t n_px) {
    if (n_px <= S.img_cap) return MLM_OK;
    if (S.d_img) hipFree(S.d_img);
    S.d_img = nullptr;
    HIPCHK(h, hipMalloc((void **)&S.d_img, n_px * sizeof(uint16_t)));
    S.img_cap = n_px;
    return MLM_OK;
}

// device staging of a slot's pixel list / point list, allocated at first use (mlm_limits.max_points entries)
int ensure_pix(mlm_handle *h, MlmSlot &S) {
    if (S.d_pix) return MLM_OK;
    HIPCHK(h, hipMalloc((void **)&S.d_pix, std::max<size_t>((size_t)h->lim.max_points, 1) * sizeof(int32_t)));
    return MLM_OK;
}
int ensure_pts(mlm_handle *h, MlmSlot &S) {
    if (S.d_pts) return MLM_OK;
    HIPCHK(h, hipMalloc((void **)&S.d_pts, std::max<size_t>((size_t)h->lim.max_points, 1) * 3 * sizeof(double)));
    return MLM_OK;
}

int ensure_query(mlm_handle *h, size_t n) {
    if (n <= h->q_cap) return MLM_OK;
    if (h->d_qpos) hipFree(h->d_qpos);
    if (h->d_qout) hipFree(h->d_qout);
    h->d_qpos = nullptr;
    h->d_qout = nullptr;
    h->q_cap = 0;
    const size_t cap = std::max<size_t>(n, 4096);
    HIPCHK(h, hipMalloc((void **)&h->d_qpos, cap * 3 * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->d_qout, cap * 3 * sizeof(double)));
    h->q_cap = cap;
    return MLM_OK;
}

bool mirror_wanted(const mlm_handle *h, int mode, int n, int max_iter);
int mirror_sync(mlm_handle *h);
void mirror_answer(const mlm_handle *h, int mode, const double *pos, int n, float inflate, int max_iter, void *out);
int run_query(mlm_handle *h, int mode, const double *pos, int n, float inflate, int max_iter, void *out,
              size_t out_elem) {
    if (!h || !pos || !out || n < 0) return MLM_ERR_INVALID;
    if (n == 0) return MLM_OK;
    MLM_LOCK(h);
    if (mirror_wanted(h, mode, n, max_iter)) { // a planner's position-by-position calls: answered on the host (mlm_mirror.h)
        const int rc = mirror_sync(h);
        if (rc == MLM_OK) {
            mirror_answer(h, mode, pos, n, inflate, max_iter, out);
            h->mir.n_host_queries += n;
            return MLM_OK;
        }
        if (!h->mir.alloc_failed) return rc; // (an error of the frames in flight, reported by the drain)
        // (no pinned host memory for the mirror: this and all later queries run as kernels)
    }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc) return rc;
    rc = ensure_query(h, (size_t)n);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_qpos, pos, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_query, dim3(grid_for((size_t)n)), dim3(MLM_BLOCK), 0, h->stream, h->P, mode, h->d_qpos, n,
                       inflate, max_iter, (int8_t *)h->d_qout, (float *)h->d_qout, (double *)h->d_qout);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_qout, (size_t)n * out_elem, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int read_global(mlm_handle *h) {
    const int rc = drain(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

// The pool fields of MlmDev (everything sized by max_blocks): copied into every slot's parameter block when the pool grows
void copy_pool_fields(MlmDev &d, const MlmDev &s) {
    d.ht_keys = s.ht_keys;
    d.ht_slot = s.ht_slot;
    d.ht_mask = s.ht_mask;
    d.max_blocks = s.max_blocks;
    d.block_keys = s.block_keys;
    d.log_odds = s.log_odds;
    d.occ = s.occ;
    d.infl = s.infl;
    d.vox_head = s.vox_head;
    d.vox_miss = s.vox_miss;
    d.vox_stride = s.vox_stride;
    d.frnt = s.frnt;
    d.blk_collapsed = s.blk_collapsed;
    d.blk_observed = s.blk_observed;
    d.vox_tau = s.vox_tau;
}
void dev_free(mlm_handle *h, void *p, size_t bytes) {
    if (!p) return;
    auto it = std::find(h->allocs.begin(), h->allocs.end(), p);
    if (it != h->allocs.end()) *it = nullptr; // (slot indices into `allocs` stay valid: MlmSlot::alloc_end)
    hipFree(p);
    h->alloc_bytes -= std::min(h->alloc_bytes, bytes);
}
// Allocate and initialise a block table + pool for `max_blocks` blocks into the pool fields of P (allocate_ram leaves a new
// block at log_odds 0, occupancy 'u', inflate_occupancy 'u': map_local.h:215-231 — pre-initialised, so creating a block is
// one CAS).
int alloc_pool(mlm_handle *h, MlmDev &P, int max_blocks) {
    int rc;
    if (max_blocks <= 0 || (long long)max_blocks * P.cells > 0x7FFFFFFFll) { // (voxel addresses are 32-bit on the cell-table path)
        h->err = "block pool beyond 2^31 voxels";
        return MLM_ERR_CAPACITY;
    }
    P.max_blocks = max_blocks;
    size_t ht = 1;
    while (ht < (size_t)P.max_blocks * 4) ht <<= 1;
    P.ht_mask = (uint32_t)(ht - 1);
    const size_t NV = (size_t)P.max_blocks * P.cells;
    // (only the cell-table path and frontier mode use the per-map-voxel scratch: the sector path groups by voxel in the
    // frame-local grid)
    if ((rc = dev_alloc(h, &P.ht_keys, ht))) return rc;
    if ((rc = dev_alloc(h, &P.ht_slot, ht))) return rc;
    if ((rc = dev_alloc(h, &P.block_keys, (size_t)P.max_blocks * 3))) return rc;
    if ((rc = dev_alloc(h, &P.log_odds, NV))) return rc;
    if ((rc = dev_alloc(h, &P.occ, NV))) return rc;
    if ((rc = dev_alloc(h, &P.infl, NV))) return rc;
    P.vox_stride = NV;
    if ((rc = dev_alloc(h, &P.vox_head, 2 * NV))) return rc;
    if ((rc = dev_alloc(h, &P.vox_miss, 2 * NV))) return rc;
    if (P.explore) {
        if ((rc = dev_alloc(h, &P.frnt, NV))) return rc;
        if ((rc = dev_alloc(h, &P.vox_tau, NV))) return rc;
        if ((rc = dev_alloc(h, &P.blk_collapsed, (size_t)P.max_blocks))) return rc;
        if ((rc = dev_alloc(h, &P.blk_observed, (size_t)P.max_blocks))) return rc;
        HIPCHK(h, hipMemset(P.frnt, 0, NV));
        HIPCHK(h, hipMemset(P.vox_tau, 0, NV * sizeof(unsigned long long)));
        HIPCHK(h, hipMemset(P.blk_collapsed, 0, (size_t)P.max_blocks));
        HIPCHK(h, hipMemset(P.blk_observed, 0, (size_t)P.max_blocks));
    }
    HIPCHK(h, hipMemset(P.ht_keys, 0xFF, ht * sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(P.ht_slot, 0xFF, ht * sizeof(int)));
    HIPCHK(h, hipMemset(P.log_odds, 0, NV * sizeof(float)));            // allocate_ram: log_odds 0
    HIPCHK(h, hipMemset(P.occ, 'u', NV));                               //               occupancy 'u'
    HIPCHK(h, hipMemset(P.infl, 'u', NV));                              //               inflate_occupancy 'u'
    HIPCHK(h, hipMemset(P.vox_head, 0xFF, 2 * NV * sizeof(int)));
    HIPCHK(h, hipMemset(P.vox_miss, 0, 2 * NV * sizeof(uint32_t)));
    return MLM_OK;
}
void free_pool(mlm_handle *h, const MlmDev &P) {
    const size_t NV = (size_t)P.max_blocks * P.cells, ht = (size_t)P.ht_mask + 1;
    dev_free(h, P.ht_keys, ht * 8);
    dev_free(h, P.ht_slot, ht * 4);
    dev_free(h, P.block_keys, (size_t)P.max_blocks * 12);
    dev_free(h, P.log_odds, NV * 4);
    dev_free(h, P.occ, NV);
    dev_free(h, P.infl, NV);
    dev_free(h, P.vox_head, 2 * NV * 4);
    dev_free(h, P.vox_miss, 2 * NV * 4);
    if (P.explore) {
        dev_free(h, P.frnt, NV);
        dev_free(h, P.vox_tau, NV * 8);
        dev_free(h, P.blk_collapsed, (size_t)P.max_blocks);
        dev_free(h, P.blk_observed, (size_t)P.max_blocks);
    }
}
// The reference's observed_group_map grows without bound (allocate_ram, map_local.h:215-231).  Here: a new table + pool of at
// least `want` blocks, the blocks copied over, the table rebuilt on the device, every parameter block re-pointed.  Nothing
// may be in flight (callers drain first).  MLM_ERR_CAPACITY only if the device cannot hold the larger pool.
// the slots' parameter blocks as the kernels see them (device-resident tables), after the host copies changed
// MlmSlot::Pfb from MlmSlot::P: the buffers the cell-table path shares across lean slots, and — slots sized by need — the handle's
// full-size set of the lists that path fills (null until ensure_ct_full has run: no cell-table frame before that)
void make_fb_params(mlm_handle *h, MlmSlot &S) {
    S.Pfb = S.P;
    if (!h->lean) return;
    S.Pfb.bnodes = h->fb_bnodes;
    S.Pfb.pairs = h->fb_pairs;
    S.Pfb.nodes = h->fb_nodes;
    if (h->need_sized && h->ct_full.ready) {
        const auto &c = h->ct_full;
        S.Pfb.mt_list = c.mt_list, S.Pfb.mt_rec = c.mt_rec, S.Pfb.contrib = c.contrib, S.Pfb.subs = c.subs;
        S.Pfb.hl_cell = c.hl_cell, S.Pfb.hl_t = c.hl_t, S.Pfb.hl_odd = c.hl_odd, S.Pfb.hl_inc = c.hl_inc, S.Pfb.hl_base = c.hl_base;
        S.Pfb.hl_cnt = c.hl_cnt, S.Pfb.hl_vt = c.hl_vt, S.Pfb.hl_key = c.hl_key, S.Pfb.hl_bkt = c.hl_bkt;
        S.Pfb.hl_cap = S.Pfb.mt_cap = (unsigned int)h->caps_worst.hl;
        S.Pfb.contrib_cap = (unsigned int)h->caps_worst.sub;
    }
}
int upload_slot_tab(mlm_handle *h) {
    std::vector<MlmDev> tab(h->slots.size());
    for (size_t i = 0; i < h->slots.size(); ++i) tab[i] = h->slots[i].P;
    HIPCHK(h, hipMemcpy(h->d_slot_tab, tab.data(), tab.size() * sizeof(MlmDev), hipMemcpyHostToDevice));
    for (auto &S : h->slots) make_fb_params(h, S);
    if (h->lean && h->d_slot_tab_fb) {
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = h->slots[i].Pfb;
        HIPCHK(h, hipMemcpy(h->d_slot_tab_fb, tab.data(), tab.size() * sizeof(MlmDev), hipMemcpyHostToDevice));
    }
    return MLM_OK;
}

// The scene keeps overflowing the columns' cell table (fill_stats counts the overflowed columns of each batch's last frame; every
// one of them is redone by the pass that has a CU to itself): double the table.  The table only exists in LDS, so this is a
// change of parameters — at a point where nothing is in flight.  The smaller table is the default because its footprint is
// worth 5 % of throughput on scenes that fit it.
int widen_sec_tab(mlm_handle *h) {
    MlmDev &P = h->P;
    const unsigned int tab = P.sec_tab * 2u, n_miss = (unsigned int)(P.nZ * (P.explore ? P.nRho : P.RW));
    const unsigned int lds = mlm_sec_lds(tab, n_miss, (unsigned int)P.nRho, (unsigned int)P.nZ, P.explore).total;
    const int nt = h->sec_threads == 256 && tab <= 1024u ? 256 : 512;
    if (!h->use_sectors || tab > 2048u || tab > 4u * (unsigned int)nt || lds > 159u * 1024u || (P.sec_tab_big && tab >= P.sec_tab_big)) return MLM_OK;
    {
        const int rc = drain(h);
        if (rc) return rc;
    }
    HIPCHK(h, hipDeviceSynchronize());
    P.sec_tab = tab;
    P.sec_lds_bytes = lds;
    h->sec_threads = nt;
    for (auto &S : h->slots) {
        S.P.sec_tab = tab;
        S.P.sec_lds_bytes = lds;
    }
    if (P.explore) {
        HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec); // (the single-frame graphs hold the old launch geometry)
    h->graphs.clear();
    if (getenv("MLM_DEBUG_CREATE")) fprintf(stderr, "[sector] cell table widened to %u entries (%u bytes of LDS per column, %d threads)\n", tab, lds, nt);
    return upload_slot_tab(h);
}

int grow_pool(mlm_handle *h, size_t want) {
    MlmGlobal g{};
    HIPCHK(h, hipMemcpy(&g, h->P.g, sizeof(g), hipMemcpyDeviceToHost));
    const unsigned int nb = std::min<unsigned int>(g.n_blocks, (unsigned int)h->P.max_blocks);
    const size_t cap = (size_t)(0x7FFFFFFFll / h->P.cells);
    size_t target = std::max<size_t>(want, 2 * (size_t)h->P.max_blocks);
    target = std::min(target, cap);
    if (target <= (size_t)h->P.max_blocks) {
        h->err = "block pool cannot grow further (2^31 voxels)";
        return MLM_ERR_CAPACITY;
    }
    if (h->grow_failed_at && target >= h->grow_failed_at) { // (the device could not hold this much before: do not allocate-and-fail every frame)
        h->err = "device memory exhausted while growing the block pool (a pool of " + std::to_string(h->grow_failed_at) + " blocks did not fit)";
        return MLM_ERR_CAPACITY;
    }
    MlmDev N = h->P;
    N.ht_keys = nullptr, N.ht_slot = nullptr, N.block_keys = nullptr, N.log_odds = nullptr, N.occ = nullptr, N.infl = nullptr, N.vox_head = nullptr,
    N.vox_miss = nullptr, N.frnt = nullptr, N.vox_tau = nullptr, N.blk_collapsed = nullptr, N.blk_observed = nullptr;
    int rc = alloc_pool(h, N, (int)target);
    if (rc) {
        (void)hipGetLastError();
        free_pool(h, N); // (what was allocated before the failure; dev_free skips the null fields)
        h->grow_failed_at = target;
        h->err = "device memory exhausted while growing the block pool: " + h->err;
        return MLM_ERR_CAPACITY;
    }
    // (alloc_pool initialises the new arrays with hipMemset on the null stream, which the handle's non-blocking streams do not
    // wait for: the copies below must not overtake it)
    HIPCHK(h, hipDeviceSynchronize());
    const size_t C = (size_t)h->P.cells;
    if (nb) {
        hipStream_t st = h->stream;
        HIPCHK(h, hipMemcpyAsync(N.block_keys, h->P.block_keys, (size_t)nb * 3 * sizeof(int), hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(N.log_odds, h->P.log_odds, nb * C * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(N.occ, h->P.occ, nb * C, hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(N.infl, h->P.infl, nb * C, hipMemcpyDeviceToDevice, st));
        if (N.explore) {
            HIPCHK(h, hipMemcpyAsync(N.frnt, h->P.frnt, nb * C, hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(N.blk_collapsed, h->P.blk_collapsed, nb, hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(N.blk_observed, h->P.blk_observed, nb, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(k_rehash_blocks, dim3(grid_for(nb)), dim3(MLM_BLOCK), 0, st, N, nb);
        HIPCHK(h, hipGetLastError());
    }
    g.n_blocks = nb; // (allocations that failed had pushed the counter past the old capacity)
    g.err &= ~1u;
    HIPCHK(h, hipMemcpyAsync(h->P.g, &g, sizeof(g), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_pool(h, h->P);
    copy_pool_fields(h->P, N);
    *h->h_g = g;
    for (int k = 0; k < MLM_SETS; ++k) *h->h_gb[k] = g;
    for (size_t i = 0; i < h->slots.size(); ++i) copy_pool_fields(h->slots[i].P, N);
    {
        const int rc = upload_slot_tab(h);
        if (rc) return rc;
    }
    h->n_pool_grows++;
    if (getenv("MLM_DEBUG_CREATE")) fprintf(stderr, "[pool] grown to %d blocks (%u in use)\n", h->P.max_blocks, nb);
    return MLM_OK;
}
// Paths that cannot replay a frame after the fact (the cell-table path's two map-dependent kernels, frontier mode, inflation,
// imports) make sure beforehand that the pool can take what they may create at most.
int ensure_free_blocks_idle(mlm_handle *h, size_t need) { // (nothing in flight on any stream)
    if (!h->pool_grow) return MLM_OK;
    HIPCHK(h, hipMemcpy(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost));
    const size_t nb = std::min<size_t>(h->h_g->n_blocks, (size_t)h->P.max_blocks);
    if (nb + need <= (size_t)h->P.max_blocks) return MLM_OK;
    // (a bound beyond what a pool can ever hold — tiny voxels over a long range — is clamped: the pool then grows as far as it
    // can and a frame that really needs more is reported as MLM_ERR_CAPACITY)
    const size_t cap = (size_t)(0x7FFFFFFFll / h->P.cells);
    const size_t want = std::min(cap, nb + 2 * need);
    if (want <= (size_t)h->P.max_blocks) return MLM_OK;
    return grow_pool(h, want);
}
int ensure_free_blocks(mlm_handle *h, size_t need) {
    if (!h->pool_grow) return MLM_OK;
    const size_t known = std::min<size_t>(h->h_g->n_blocks, (size_t)h->P.max_blocks);
    if (known + need <= (size_t)h->P.max_blocks) return MLM_OK;
    const int rc = drain(h);
    if (rc) return rc;
    return ensure_free_blocks_idle(h, need);
}

int alloc_slot(mlm_handle *h, MlmSlot &S, size_t index, const std::vector<float> &sigma3) {
    if (!h->lean) // (test hook: the full slots "do not fit" from slot k on, so that the lean retry of mlm_create runs)
        if (long long kv; knob("debug_fail_slot", kv))
            if ((long long)index >= kv) {
                h->err = "simulated allocation failure (MLM_DEBUG_FAIL_SLOT)";
                return MLM_ERR_HIP;
            }
    S.P = h->P;
    MlmDev &P = S.P;
    int rc;
    const size_t NC = (size_t)P.nCells;
    const mlm_handle::SlotCaps &caps = h->caps_now; // (== caps_worst unless the slots are sized by need)
    S.h_ctr = h->h_ctr_all + index;
    P.ctr = h->d_ctr_all + index;
    // Lean slots of a sector-path handle (not frontier mode, whose own map-dependent part reads them per frame): the per-frame
    // state only the cell-table path keeps — per-cell records, miss-mask copies, queues, the voxel addresses of its two
    // map-dependent kernels — exists ONCE, in slot 0's name; a frame that takes that path (a fall-back, a batch submitted while
    // the sector path backs off, a frame too wide for it) runs alone from its Stage A to the end of its apply kernel.
    const bool share = h->lean && !P.explore, own = !share || index == 0;
#define MLM_CT_ALLOC(field, count)                                                                                    \
    do {                                                                                                              \
        if (!own) P.field = h->slots[0].P.field;                                                                      \
        else if ((rc = dev_alloc(h, &P.field, (count)))) return rc;                                                   \
    } while (0)
    MLM_CT_ALLOC(cs, NC);
    MLM_CT_ALLOC(miss_bits, (size_t)MLM_MISS_COPIES * P.nMissWords);
    P.hl_cap = (unsigned int)caps.hl;
    P.mt_cap = (unsigned int)caps.mt;
    P.vh_cap = (unsigned int)caps.vh;
    if ((rc = dev_alloc(h, &P.mt_list, caps.mt))) return rc;
    if ((rc = dev_alloc(h, &P.mt_rec, caps.mt))) return rc;
    MLM_CT_ALLOC(mt_big, NC);
    P.touch_cap = (unsigned int)NC;
    MLM_CT_ALLOC(touched, (size_t)MLM_RAY_LISTS * P.touch_cap);
    size_t max_contrib = 0; // most contributions one frame can make
    // (256 work items per bin block; image edges and short lists add blocks: twice the quotient + 256)
    P.nb_cap = (unsigned int)((size_t)h->lim.max_points / 128 + 256);
    if ((rc = dev_alloc(h, &P.blk_stats, 4 * (size_t)P.nb_cap))) return rc;
    // (lean: k_bin_sectors writes at most 256 records per block; the cell-table path's buffers are shared, mlm_create)
    if ((rc = dev_alloc(h, &P.bnodes, (size_t)P.nb_cap * (h->lean ? 256u : P.node_lds)))) return rc;
    if (!h->lean && (rc = dev_alloc(h, &P.pairs, (size_t)P.nb_cap * P.agg_lds))) return rc;
    {
        // most contributions one point can make: centre + (+d,-d) while d < 3*sigma(rho) (map_awareness.cpp:149)
        int dmax = 0;
        for (int r = 0; r < P.nRho; ++r) {
            int d = 1;
            while ((float)d < sigma3[r] && r + d < P.nRho && d <= MLM_DIFF_RANGE) ++d;
            dmax = std::max(dmax, d - 1);
        }
        const size_t cap = (size_t)h->lim.max_points * (size_t)(1 + 2 * dmax);
        max_contrib = cap;
        if (cap > 0xFFFFFFF0ull) {
            h->err = "contribution buffer too large";
            return MLM_ERR_UNSUPPORTED;
        }
        // segments are padded to 16 entries; a multi-kind cell has >= 2 contributions
        const size_t cap_pad = cap + 15 * std::min<size_t>(NC, cap / 2) + 64;
        if (cap_pad > 0xFFFFFFF0ull) {
            h->err = "contribution buffer too large";
            return MLM_ERR_UNSUPPORTED;
        }
        P.contrib_cap = (unsigned int)caps.sub; // (<= cap_pad, the worst case: mlm_create)
        if ((rc = dev_alloc(h, &P.contrib, caps.sub))) return rc;
        if ((rc = dev_alloc(h, &P.subs, caps.sub))) return rc;
        P.node_cap = (unsigned int)(cap / MLM_RAY_LISTS + 4096);
        if (!h->lean && (rc = dev_alloc(h, &P.nodes, (size_t)MLM_RAY_LISTS * P.node_cap))) return rc;
    }
    if ((rc = dev_alloc(h, &P.ov_list, (size_t)P.nPhi))) return rc;
    P.chunk_cap = P.nb_cap; // a column can at most get one run from every bin block
    if ((rc = dev_alloc(h, &P.col_cnt, (size_t)P.nPhi))) return rc;
    if ((rc = dev_alloc(h, &P.col_chunks, h->use_sectors ? 2 * (size_t)P.nPhi * P.chunk_cap : 2))) return rc;
    if (h->use_sectors && !P.explore) { // (frontier mode's own Stage B+C takes over after k_sector: no tiles)
        P.mc_list_cap = (unsigned int)NC; // unique miss cells of a frame
        if ((rc = dev_alloc(h, &P.mc_list, (size_t)P.mc_list_cap + 8))) return rc;
        if ((rc = dev_alloc(h, &P.hl_vt16, caps.hl + 8))) return rc;
        if ((rc = dev_alloc(h, &P.tile_cols, (size_t)P.n_tiles * P.tile_words))) return rc;
        HIPCHK(h, hipMemset(P.tile_cols, 0, (size_t)P.n_tiles * P.tile_words * sizeof(uint32_t)));
        if ((rc = dev_alloc(h, &P.tile_desc, 4 * (size_t)P.n_tiles * (size_t)P.nPhi))) return rc;
        // a frame touches at most one voxel per awareness cell, and no more voxels than its grid has
        P.rec_cap = (unsigned int)caps.rec;
        if ((rc = dev_alloc(h, &P.vr_rec, (size_t)P.rec_cap))) return rc;
        if ((rc = dev_alloc(h, &P.vr_hit, caps.vh))) return rc;
        if ((rc = dev_alloc(h, &P.tile_dir, 4 * (size_t)P.n_tiles))) return rc;
        HIPCHK(h, hipMemset(P.tile_dir, 0xFF, 4 * (size_t)P.n_tiles * sizeof(uint32_t))); // (no frame carries that sequence number)
        // bucket-first table of this slot: room for the emulated container of a frame with up to 2 * max_points unique
        // hit cells (more: the handle continues on the cell-table path)
        P.sbkt_cap = (unsigned int)caps.sbkt;
        if ((rc = dev_alloc(h, &P.sbkt, (size_t)P.sbkt_cap))) return rc;
        HIPCHK(h, hipMemset(P.sbkt, 0xFF, (size_t)P.sbkt_cap * sizeof(unsigned long long)));
    }
    // (a reference — one row of a group's lane mask — stands for at least one contribution; a cell's references start at a multiple
    // of MLM_SEC_REF_ALIGN, and a cell that needs references has at least two contributions)
    (void)max_contrib;
    P.refs_cap = (unsigned int)caps.refs;
    if ((rc = dev_alloc(h, &P.refs, h->use_sectors ? (size_t)P.refs_cap : 4))) return rc;
    if ((rc = dev_alloc(h, &P.mt_ref, h->use_sectors ? 2 * caps.mt : 2))) return rc;
    HIPCHK(h, hipMemset(P.col_cnt, 0, (size_t)P.nPhi * sizeof(unsigned int)));
    if ((rc = dev_alloc(h, &P.hl_cell, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_t, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_odd, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_inc, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_base, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_cnt, caps.hl))) return rc;
    if ((rc = dev_alloc(h, &P.hl_vt, caps.hl))) return rc;
    MLM_CT_ALLOC(hl_arr, NC);
    if ((rc = dev_alloc(h, &P.hl_key, caps.hl))) return rc;
    MLM_CT_ALLOC(hl_next, NC);
    MLM_CT_ALLOC(hl_vox, NC);
    if ((rc = dev_alloc(h, &P.hl_bkt, caps.hl))) return rc;
    MLM_CT_ALLOC(hl_bkey, NC);
    MLM_CT_ALLOC(hl_cid, NC);
    MLM_CT_ALLOC(hl_slot, NC);
    P.mc_cap = (unsigned int)((size_t)P.nMissWords * 32 / MLM_RAY_LISTS + 4096);
    MLM_CT_ALLOC(mc_bkey, (size_t)MLM_RAY_LISTS * P.mc_cap);
    MLM_CT_ALLOC(mc_cid, (size_t)MLM_RAY_LISTS * P.mc_cap);
    MLM_CT_ALLOC(mc_slot, (size_t)MLM_RAY_LISTS * P.mc_cap);
    MLM_CT_ALLOC(mc_vox, (size_t)MLM_RAY_LISTS * P.mc_cap);
    if ((rc = dev_alloc(h, &P.ml_cell, P.record_awareness ? NC : 1))) return rc;
    if (P.explore) {
        if ((rc = dev_alloc(h, &P.start_t, NC))) return rc;
        if ((rc = dev_alloc(h, &P.miss_t, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_rays, (size_t)h->lim.max_points * 4 + 4096))) return rc;
        if ((rc = dev_alloc(h, &P.ex_cell, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_t, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_vt, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_arr, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_key, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_vox, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_bkey, NC))) return rc;
        if ((rc = dev_alloc(h, &P.ex_cid, NC))) return rc;
        HIPCHK(h, hipMemset(P.start_t, 0xFF, NC * sizeof(uint32_t)));
        HIPCHK(h, hipMemset(P.miss_t, 0xFF, NC * sizeof(uint32_t)));
    }
    P.mvox_cap = (unsigned int)((size_t)P.nMissWords * 32 / MLM_RAY_LISTS + 512);
    MLM_CT_ALLOC(miss_vox, (size_t)MLM_RAY_LISTS * P.mvox_cap);
#undef MLM_CT_ALLOC
    if (own) {
        std::vector<MlmCell> init(NC, MlmCell{MLM_EMPTY_T, 0u, 0u, MLM_NIL});
        HIPCHK(h, hipMemcpy(P.cs, init.data(), NC * sizeof(MlmCell), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemset(P.miss_bits, 0, (size_t)MLM_MISS_COPIES * P.nMissWords * sizeof(uint32_t)));
    }
    // (the staging of host images, pixel lists and point lists is allocated by the calls that use it: ensure_img / ensure_list)
    S.alloc_end = h->allocs.size();
    return MLM_OK;
}

// The worst-case and the initial capacities of a frame slot's lists (mlm_create, before the slots are allocated).
// Worst case: every awareness cell a multi-kind hit.  By need: what camera frames of max_points pixels produce, with room —
// config 2 (307 k pixels): 20 k hits, 8 k ranked cells, 260 k references, 730 k ordered kinds, 33 k voxel records per frame;
// config 3 (922 k pixels): 137 k hits, all ranked, 4.1 M references, 7.7 M kinds, 350 k records.
void slot_capacities(mlm_handle *h, const std::vector<float> &sigma3) {
    const MlmDev &P = h->P;
    const size_t NC = (size_t)P.nCells, Pn = (size_t)std::max(1, h->lim.max_points);
    int dmax = 0;
    for (int r = 0; r < P.nRho; ++r) {
        int d = 1;
        while ((float)d < sigma3[r] && r + d < P.nRho && d <= MLM_DIFF_RANGE) ++d;
        dmax = std::max(dmax, d - 1);
    }
    const size_t max_contrib = Pn * (size_t)(1 + 2 * dmax);
    auto &w = h->caps_worst;
    w.hl = w.mt = w.vh = NC;
    w.rec = std::min<size_t>(NC, (size_t)P.lv_nx * P.lv_ny * P.lv_nz);
    w.refs = std::min<size_t>(0xFFFFFFF0ull, max_contrib + (MLM_SEC_REF_ALIGN - 1) * std::min<size_t>(NC, max_contrib / 2) + 64);
    w.sub = std::min<size_t>(0xFFFFFFF0ull, max_contrib + 15 * std::min<size_t>(NC, max_contrib / 2) + 64); // (checked against 2^32 by alloc_slot's caller)
    w.sbkt = std::min<size_t>(h->max_buckets, std::__detail::_Prime_rehash_policy()._M_next_bkt(4 * Pn + 2));
    h->caps_now = w;
    if (!h->need_sized) return;
    auto &c = h->caps_now;
    c.hl = std::min(w.hl, std::max<size_t>(32768, Pn / 4));
    c.mt = c.hl;
    c.vh = c.hl;
    c.rec = std::min(w.rec, 2 * c.hl);
    c.refs = std::min(w.refs, std::max<size_t>(262144, 6 * Pn));
    c.sub = std::min(w.sub, std::max<size_t>(1u << 20, 10 * Pn));
    c.sbkt = std::min(w.sbkt, std::__detail::_Prime_rehash_policy()._M_next_bkt(2 * c.hl + 2));
}

// One list of one slot re-allocated with room for `n_new` elements; `keep`: the slot holds a frame whose Stage A output is still
// needed (it is pending behind the frame that ran out of room), so the old contents move over.
template <class T> int regrow(mlm_handle *h, T **field, size_t n_old, size_t n_new, bool keep, int fill = -1) {
    if (n_new <= n_old) return MLM_OK;
    T *fresh = nullptr;
    const int rc = dev_alloc(h, &fresh, n_new);
    if (rc) return rc;
    if (fill >= 0) HIPCHK(h, hipMemsetAsync(fresh, fill, n_new * sizeof(T), h->stream));
    if (keep && *field) HIPCHK(h, hipMemcpyAsync(fresh, *field, n_old * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    h->regrow_trash.emplace_back((void *)*field, n_old * sizeof(T)); // (freed by resize_slots once the copies are through)
    *field = fresh;
    return MLM_OK;
}
// Apply h->caps_now to every slot (lists only ever grow).  Nothing may be in flight.
int resize_slots(mlm_handle *h) {
    const auto &c = h->caps_now;
    std::vector<char> keep(h->slots.size(), 0);
    for (const MlmSlot *R : h->pending) keep[(size_t)(R - h->slots.data())] = 1;
    for (size_t i = 0; i < h->slots.size(); ++i) {
        MlmDev &P = h->slots[i].P;
        const bool k = keep[i] != 0;
        int rc = MLM_OK;
        const size_t hl0 = P.hl_cap, mt0 = P.mt_cap;
#define MLM_RG(field, n0, n1, ...)                                                                                    \
    if (rc == MLM_OK) rc = regrow(h, &P.field, (size_t)(n0), (size_t)(n1), k, ##__VA_ARGS__)
        MLM_RG(hl_cell, hl0, c.hl);
        MLM_RG(hl_t, hl0, c.hl);
        MLM_RG(hl_odd, hl0, c.hl);
        MLM_RG(hl_inc, hl0, c.hl);
        MLM_RG(hl_base, hl0, c.hl);
        MLM_RG(hl_cnt, hl0, c.hl);
        MLM_RG(hl_vt, hl0, c.hl);
        MLM_RG(hl_key, hl0, c.hl);
        MLM_RG(hl_bkt, hl0, c.hl);
        MLM_RG(hl_vt16, hl0 + 8, c.hl + 8);
        MLM_RG(mt_list, mt0, c.mt);
        MLM_RG(mt_rec, mt0, c.mt);
        MLM_RG(mt_ref, 2 * mt0, 2 * c.mt);
        MLM_RG(vr_hit, P.vh_cap, c.vh);
        MLM_RG(vr_rec, P.rec_cap, c.rec);
        MLM_RG(refs, P.refs_cap, c.refs);
        MLM_RG(contrib, P.contrib_cap, c.sub);
        MLM_RG(subs, P.contrib_cap, c.sub);
        MLM_RG(sbkt, P.sbkt_cap, c.sbkt, 0xFF); // (no frame carries that sequence number; entries of pending frames move over)
#undef MLM_RG
        if (rc) { // (the device ran out of memory half-way: the lists replaced so far stay — every slot's capacities describe what it holds)
            (void)hipStreamSynchronize(h->stream);
            for (auto &t : h->regrow_trash) dev_free(h, t.first, t.second);
            h->regrow_trash.clear();
            (void)upload_slot_tab(h); // (the kernels' copies of the slots must not keep pointing at what was just freed)
            h->err = "device memory exhausted while enlarging the frame slots: " + h->err;
            return rc;
        }
        P.hl_cap = (unsigned int)std::max<size_t>(hl0, c.hl);
        P.mt_cap = (unsigned int)std::max<size_t>(mt0, c.mt);
        P.vh_cap = (unsigned int)std::max<size_t>(P.vh_cap, c.vh);
        P.rec_cap = (unsigned int)std::max<size_t>(P.rec_cap, c.rec);
        P.refs_cap = (unsigned int)std::max<size_t>(P.refs_cap, c.refs);
        P.contrib_cap = (unsigned int)std::max<size_t>(P.contrib_cap, c.sub);
        P.sbkt_cap = (unsigned int)std::max<size_t>(P.sbkt_cap, c.sbkt);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto &t : h->regrow_trash) dev_free(h, t.first, t.second);
    h->regrow_trash.clear();
    h->n_slot_grows++;
    if (getenv("MLM_DEBUG_CREATE"))
        fprintf(stderr, "[slots] lists enlarged: %zu hits, %zu ranked cells, %zu references, %zu kinds, %zu voxel records, %zu buckets per frame; device memory %.2f GB\n",
                c.hl, c.mt, c.refs, c.sub, c.rec, c.sbkt, h->alloc_bytes / 1e9);
    return upload_slot_tab(h);
}
// A frame's Stage A ran out of room in a list sized by need (sector_overflow 3): `demand` are that frame's counters — what its
// columns and tiles asked for.  Every list that was too short grows to at least twice its size and 5/4 of the demand.
int grow_slots(mlm_handle *h, const MlmCounters &demand) {
    auto &c = h->caps_now;
    const auto &w = h->caps_worst;
    auto want = [](size_t now, size_t need, size_t worst) { return need > now ? std::min(worst, std::max(2 * now, need + need / 4)) : now; };
    const mlm_handle::SlotCaps before = c;
    c.hl = want(c.hl, demand.u_hit, w.hl);
    c.mt = want(c.mt, demand.n_multi, w.mt);
    c.refs = want(c.refs, demand.n_refs, w.refs);
    c.sub = want(c.sub, demand.n_contrib, w.sub);
    c.rec = want(c.rec, demand.mvox_cnt[0][0], w.rec);
    c.vh = want(c.vh, demand.mvox_cnt[1][0], w.vh);
    if (c.hl == before.hl && c.mt == before.mt && c.refs == before.refs && c.sub == before.sub && c.rec == before.rec && c.vh == before.vh) {
        // (the counters of a frame that gave up may be short of its real demand: a column returns at its first full list) — double them all
        c.hl = std::min(w.hl, 2 * c.hl), c.mt = std::min(w.mt, 2 * c.mt), c.refs = std::min(w.refs, 2 * c.refs);
        c.sub = std::min(w.sub, 2 * c.sub), c.rec = std::min(w.rec, 2 * c.rec), c.vh = std::min(w.vh, 2 * c.vh);
    }
    const int rc = resize_slots(h);
    if (rc) c = before; // (some slots may hold more than this now — resize_slots only ever grows a list —, none holds less)
    return rc;
}
// the emulated hit container has more buckets than the slots' bucket-first tables hold (they are sized by need too)
int grow_sbkt(mlm_handle *h, size_t buckets) {
    auto &c = h->caps_now;
    if (buckets <= c.sbkt) return MLM_OK;
    const size_t before = c.sbkt;
    c.sbkt = std::min(h->caps_worst.sbkt, std::max(2 * c.sbkt, buckets));
    const int rc = resize_slots(h);
    if (rc) c.sbkt = before; // (the next submission asks again: the slots' own sbkt_cap is what the kernels go by)
    return rc;
}
// The cell-table path's own full-size lists, once per handle, at the first frame that takes that path (slots sized by need)
int ensure_ct_full(mlm_handle *h) {
    if (!h->need_sized || h->ct_full.ready) return MLM_OK;
    auto &c = h->ct_full;
    const size_t NC = h->caps_worst.hl, SUB = h->caps_worst.sub;
    int rc = MLM_OK;
    // (a list obtained by an earlier, partly failed attempt is kept: only the members that are still missing are asked for again)
    auto need = [&](auto **p, size_t n) { return *p == nullptr && (rc = dev_alloc(h, p, n)) != MLM_OK; };
    if (need(&c.mt_list, NC) || need(&c.mt_rec, NC) || need(&c.contrib, SUB) || need(&c.subs, SUB) || need(&c.hl_cell, NC) || need(&c.hl_t, NC) ||
        need(&c.hl_odd, NC) || need(&c.hl_inc, NC) || need(&c.hl_base, NC) || need(&c.hl_cnt, NC) || need(&c.hl_vt, NC) || need(&c.hl_key, NC) ||
        need(&c.hl_bkt, NC))
        return rc;
    c.ready = true;
    if (getenv("MLM_DEBUG_CREATE")) fprintf(stderr, "[slots] full-size lists of the cell-table path allocated; device memory %.2f GB\n", h->alloc_bytes / 1e9);
    return upload_slot_tab(h);
}

} // namespace
