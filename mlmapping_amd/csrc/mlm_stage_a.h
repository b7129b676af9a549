// mlm_stage_a.h — host side of Stage A (the map-independent part of a frame: binning, columns, ranking, tiles) for both paths, the
// exact ordering of the emulated hit container on rehash frames, and the per-frame statistics.  Part of mlmap_hip.hip.
#pragma once
namespace {

// T_ls and t_wa of one frame (map_awareness.cpp:184-186)
void frame_setup(const mlm_handle *h, const double q_wb_in[4], const double t_wb_in[3], MlmFrame &F) {
    frame_pose(h->q_bs, h->t_bs, q_wb_in, t_wb_in, F.q_ls, F.t_ls, F.t_wa);
    F.rehash_thr = 0xFFFFFFFFu; // only the speculative Stage B arms the check (submit_batch)
    {
        const double w = F.q_ls[0], x = F.q_ls[1], y = F.q_ls[2], z = F.q_ls[3];
        // v + w (2 q x v) + q x (2 q x v)  =  (I + 2 w [q]x + 2 [q]x^2) v,   [q]x^2 = q q^T - |q_v|^2 I
        const double n2 = x * x + y * y + z * z;
        const double m[9] = {1 + 2 * (x * x - n2), 2 * (x * y - w * z),   2 * (x * z + w * y),
                             2 * (x * y + w * z),   1 + 2 * (y * y - n2), 2 * (y * z - w * x),
                             2 * (x * z - w * y),   2 * (y * z + w * x),   1 + 2 * (z * z - n2)};
        for (int i = 0; i < 9; ++i) F.m_ls[i] = m[i];
        const double q1 = std::fabs(x) + std::fabs(y) + std::fabs(z);
        F.m_gain = 1 + 2 * std::fabs(w) * q1 + 2 * q1 * q1;
        F.t_l1 = std::fabs(F.t_ls[0]) + std::fabs(F.t_ls[1]) + std::fabs(F.t_ls[2]) + 1.0;
    }
    // origin of the frame-local voxel grid: the awareness cylinder around t_wa with a margin of four voxels
    const MlmDev &P = h->P;
    const double R = P.nRho * P.dRho;
    // (x, y snapped down to a tile boundary: a frame-local tile is then a WORLD tile, which is what lets one workgroup own a
    // tile's voxels across the frames of a batch — k_apply_tiles)
    const int edge_mask = (1 << P.tile_sh) - 1;
    F.lv_o[0] = ((int)std::floor((F.t_wa[0] - R) / P.d_sub) - 4) & ~edge_mask;
    F.lv_o[1] = ((int)std::floor((F.t_wa[1] - R) / P.d_sub) - 4) & ~edge_mask;
    F.lv_o[2] = (int)std::floor((F.t_wa[2] + P.z_border_min) / P.d_sub) - 4;
}

std::vector<std::pair<size_t, size_t>> plan_epochs(mlm_handle *h, size_t U) {
    return plan_epochs_for(h->hit_pol, h->hit_n_bkt, U);
}

// Stage B for one frame whose unique-hit count U is known on the host: exact, with rehash epochs.
int order_hits_exact(mlm_handle *h, MlmSlot &S, unsigned int U, int frame_idx) {
    const MlmDev &P = eff_params(h, S);
    const auto ep = plan_epochs(h, U);
    h->stats.n_rehash_epochs = (int64_t)ep.size();
    if (h->hit_n_bkt > h->max_buckets) {
        h->err = "emulated bucket count exceeds capacity";
        return MLM_ERR_CAPACITY;
    }
    if (U == 0) return MLM_OK;
    const bool multi = ep.size() > 1;
    if (multi) {
        // arrival index = rank of the first-touch time
        tlaunch(h, "k_time_keys", k_time_keys, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sk_in, h->sv_in);
        if (mlm_sort_pairs_u64_u32(h->sort_tmp, h->sort_tmp_bytes, h->sk_in, h->sk_out, h->sv_in, h->sv_out, U,
                                   h->stream) != 0) {
            h->err = "radix sort failed";
            return MLM_ERR_HIP;
        }
        tlaunch(h, "k_assign_rank", k_assign_rank, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sv_out, U, 1);
    }
    for (size_t e = 0; e < ep.size(); ++e) {
        const unsigned int m = (unsigned int)ep[e].first;
        const unsigned long long nb = ep[e].second;
        const bool final_pass = (e + 1 == ep.size());
        HIPCHK(h, hipMemsetAsync(P.bkt_first, 0xFF, nb * sizeof(uint32_t), h->stream));
        tlaunch(h, "k_bucket_min", k_bucket_min, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, frame_idx, nb, m,
                           multi ? 1 : 0);
        tlaunch(h, "k_make_keys", k_make_keys, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, frame_idx, nb, m,
                           multi ? 1 : 0, final_pass ? 1 : 0, h->sk_in, h->sv_in);
        if (!final_pass) {
            // list order of the epoch = descending key; the rehash re-inserts the nodes in that order
            if (mlm_sort_pairs_u64_u32(h->sort_tmp, h->sort_tmp_bytes, h->sk_in, h->sk_out, h->sv_in, h->sv_out, U,
                                       h->stream) != 0) {
                h->err = "radix sort failed";
                return MLM_ERR_HIP;
            }
            tlaunch(h, "k_assign_rank", k_assign_rank, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sv_out, m, 0);
        }
    }
    return MLM_OK;
}

// workgroups of k_book_cells (MLM_BOOK_GROUP k_bin_points blocks each; dense mode: 4x4 tiles) and the tile geometry
inline unsigned int book_grid(const MlmDev &P, const MlmFrame &F, int mode, int nb, int &tiles_x, int &tiles_y) {
    if (mode != 0) {
        tiles_x = tiles_y = 0;
        return (unsigned int)((nb + MLM_BOOK_GROUP - 1) / MLM_BOOK_GROUP);
    }
    const int tile_h = (int)(P.bin_block / 256) * 8;
    tiles_x = (F.width + 31) / 32;
    tiles_y = (F.height + tile_h - 1) / tile_h;
    return (unsigned int)(((tiles_x + 3) / 4) * ((tiles_y + MLM_BOOK_GROUP / 4 - 1) / (MLM_BOOK_GROUP / 4)));
}

// Stage A of a whole batch (slots base..base+n, same mode and image geometry) on stream_a: awareness raycast ->
// unique hit lists (+odds) and miss masks.  One launch per kernel covers all n frames (blockIdx.z = slot).
// on_main: on the main stream instead of the slot set's (lean slots whose cell-table state is shared: the frame runs alone,
// its map-dependent kernels follow on the same stream).
int launch_stage_a_batch(mlm_handle *h, int base, int n, bool on_main = false) {
    if (h->lean && n > 1) { // the cell-table path's large buffers exist once per handle: one frame at a time
        for (int j = 0; j < n; ++j) {
            const int rc = launch_stage_a_batch(h, base + j, 1, on_main);
            if (rc) return rc;
        }
        return MLM_OK;
    }
    {
        const int rc = ensure_ct_full(h); // (slots sized by need: this path's own full-size lists, at its first use)
        if (rc) return rc;
    }
    const MlmDev *slot_tab = h->lean ? h->d_slot_tab_fb : h->d_slot_tab;
    const MlmSlot &S0 = h->slots[(size_t)base];
    const MlmDev &P = S0.P;
    const MlmFrame &F = S0.F;
    const int mode = S0.mode;
    const int set = base / (h->lim.max_batch);
    hipStream_t st = on_main ? h->stream : h->stream_as[set];
    // the previous user of this slot set must have been consumed by the main stream
    if (!on_main) HIPCHK(h, hipStreamWaitEvent(st, h->set_free[set], 0));
    if (!h->own_stream && !on_main) {
        // mlm_set_stream: device inputs (the *_dev entry points) may still be being produced by work the caller enqueued
        // on that stream; Stage A reads them on its own stream, so order it behind everything enqueued there so far.
        // (Costs the overlap of this batch's Stage A with the previous batch's Stage B+C; the handle's own stream,
        // which nobody else can enqueue on, needs no such edge.)
        HIPCHK(h, hipEventRecord(h->inputs_ready, h->stream));
        HIPCHK(h, hipStreamWaitEvent(st, h->inputs_ready, 0));
    }
    for (int j = 0; j < n; ++j) {
        h->h_frame_tab[base + j] = h->slots[(size_t)(base + j)].F;
        h->slots[(size_t)(base + j)].h_ctr->apply_done = 0u; // (the slot's counters are in use: not "left clear by a single-frame graph", submit_single_graph)
    }
    HIPCHK(h, hipMemcpyAsync(h->d_frame_tab + base, h->h_frame_tab + base, (size_t)n * sizeof(MlmFrame),
                             hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemsetAsync(h->d_ctr_all + base, 0, (size_t)n * sizeof(MlmCounters), st));
    if (h->lean && !on_main) HIPCHK(h, hipStreamWaitEvent(st, h->fb_done, 0)); // (the previous user of the shared buffers, on whatever stream)
    unsigned int nb = 0;
    if (F.n > 0) {
        nb = bin_grid(P, F, mode);
        if (nb > (unsigned int)h->lim.max_points / 64 + 1024) {
            h->err = "frame geometry exceeds the queues sized from mlm_limits.max_points";
            return MLM_ERR_CAPACITY;
        }
        if (mode == 0)
            tlaunch(h, "k_bin_points", k_bin_points<0>, dim3(nb, 1, n), dim3(P.bin_block), P.bin_lds_bytes, st, slot_tab, h->d_frame_tab, base);
        else if (mode == 1)
            tlaunch(h, "k_bin_points", k_bin_points<1>, dim3(nb, 1, n), dim3(P.bin_block), P.bin_lds_bytes, st, slot_tab, h->d_frame_tab, base);
        else
            tlaunch(h, "k_bin_points", k_bin_points<2>, dim3(nb, 1, n), dim3(P.bin_block), P.bin_lds_bytes, st, slot_tab, h->d_frame_tab, base);
    }
    if (nb) {
        int tx, ty;
        const unsigned int ng = book_grid(P, F, mode, (int)nb, tx, ty);
        tlaunch(h, "k_book_cells", k_book_cells, dim3(ng, 1, n), dim3(MLM_BOOK_THREADS), 0, st, slot_tab, h->d_frame_tab, base, tx, ty, (int)nb);
    }
    {
        tlaunch(h, "k_assign_nodes", k_assign_nodes, dim3(8, MLM_RAY_LISTS, n), dim3(MLM_BLOCK), 0, st, slot_tab,
                           h->d_frame_tab, base, mode == 0 ? F.width : 0);
    }
    if (P.explore) // frontier mode: the queued rays are walked once every start cell's first point is known
        tlaunch(h, "k_ex_walk_rays", k_ex_walk_rays, dim3(1024, 1, n), dim3(MLM_BLOCK), 0, st, slot_tab, h->d_frame_tab, base);
    {
        tlaunch(h, "k_collect_hits", k_collect_hits, dim3(h->collect_grid, MLM_RAY_LISTS, n), dim3(MLM_BLOCK), 0, st, slot_tab,
                           h->d_frame_tab, base, (int)nb);
    }
    {
        tlaunch(h, "k_expand_nodes", k_expand_nodes, dim3(nb + 8 * MLM_RAY_LISTS, 1, n), dim3(h->expand_block), 0, st, slot_tab,
                           h->d_frame_tab, base, mode == 0 ? F.width : 0, (int)nb);
    }
    {
        const int row_w = mode == 0 ? F.width : 64; // rows of the ranking bitmap (see k_sort_contribs)
        unsigned long long dm;
        int ds;
        div_magic((unsigned int)row_w, dm, ds);
        tlaunch(h, "k_sort_contribs", k_sort_contribs<1024>, dim3((n > 4 ? h->sort_grid : 1024) * (MLM_BLOCK / h->sort_block), 1, n), dim3(h->sort_block), 0, st, slot_tab,
                           h->d_frame_tab, base, 0u, row_w, dm, ds);
        tlaunch(h, "k_sort_contribs", k_sort_contribs<4096>, dim3(n > 4 ? 128 : 512, 1, n), dim3(MLM_BLOCK), 0, st, slot_tab,
                           h->d_frame_tab, base, 1024u, row_w, dm, ds);
    }
    {
        tlaunch(h, "k_chain", k_chain, dim3(n > 4 ? 64 : 256, 1, n), dim3(MLM_BLOCK), (size_t)21 * P.nRho * sizeof(float), st,
                           slot_tab, h->d_frame_tab, base,
                           P.explore ? 0xFFFFFFFFu : (unsigned int)std::min<size_t>(h->hit_pol._M_next_resize, 0xFFFFFFFFu));
    }
    {
        // one 256-word slice of the miss mask per block (the unique hits, far fewer, are strided over the same blocks)
        const unsigned int pb = std::max(64u, grid_for((size_t)P.nMissWords));
        tlaunch(h, "k_prepare_voxels", k_prepare_voxels, dim3(pb, 1, n), dim3(MLM_BLOCK), 0, st, slot_tab, h->d_frame_tab, base);
    }
    if (P.explore)
        tlaunch(h, "k_ex_collect_misses", k_ex_collect_misses, dim3(1024, 1, n), dim3(MLM_BLOCK), 0, st, slot_tab, h->d_frame_tab, base);
    if (h->lean) HIPCHK(h, hipEventRecord(h->fb_done, st));
    if (!on_main) HIPCHK(h, hipEventRecord(h->stage_a_done[set], st));
    return MLM_OK;
}

// the callback's sampled pixels lie in the handle's pinned staging buffer, indices first, depths `n` entries behind: a small frame's first
// kernel fetches its entries together with the parameters (only that buffer: its extent is known).  n = 0: some other list, or none.
inline unsigned int staged_list_len(const mlm_handle *h, const MlmSlot &S) {
    if (S.mode == 1 && h->h_stage && S.F.pix == h->h_stage && S.F.raw > S.F.pix && (size_t)(S.F.raw - h->h_stage) + (size_t)(S.F.raw - S.F.pix) <= h->stage_cap)
        return (unsigned int)(S.F.raw - S.F.pix);
    return 0u;
}
// Stage A by azimuth sector (mlm_kernels_sector.h): two launches per batch.
// on_main: on the main stream, in front of the frames' map-dependent launches (frontier mode's synchronous calls: nothing to overlap
// with, and a dependency between two streams costs such a call up to 60 us — measured: the first stream a process creates after the
// main one shares a hardware queue with it, tools/frontier_latency.py)
int launch_stage_a_sector(mlm_handle *h, int base, int n, bool on_main = false) {
    const MlmSlot &S0 = h->slots[(size_t)base];
    const MlmDev &P = S0.P;
    const MlmFrame &F = S0.F;
    const int mode = S0.mode;
    const int set = base / (h->lim.max_batch);
    hipStream_t st = on_main ? h->stream : h->stream_as[set];
    if (!on_main) HIPCHK(h, hipStreamWaitEvent(st, h->set_free[set], 0));
    if (!h->own_stream && !on_main) { // see launch_stage_a_batch
        HIPCHK(h, hipEventRecord(h->inputs_ready, h->stream));
        HIPCHK(h, hipStreamWaitEvent(st, h->inputs_ready, 0));
    }
    // a lone small frame in a slot whose last frame left the counters clear (mlm_hand_back, MLM_CTR_CLEARED): no prologue launch, the
    // first kernel takes the parameters from pinned memory itself (k_bin_sectors_hostf; submit_single_graph does the same for its graph)
    const unsigned int nb_all = F.n <= 0 ? 0u : mode == 0 ? (unsigned int)(((F.width + 31) / 32) * ((F.height + 7) / 8)) : (unsigned int)(((size_t)F.n + 255) / 256);
    const bool no_prologue = n == 1 && nb_all > 0 && nb_all <= kHostFrameStrips && nb_all <= P.nb_cap && S0.h_ctr->apply_done == MLM_CTR_CLEARED;
    for (int j = 0; j < n; ++j) {
        h->h_frame_tab[base + j] = h->slots[(size_t)(base + j)].F;
        h->slots[(size_t)(base + j)].h_ctr->apply_done = 0u; // (the slot's counters are in use from here on)
    }
    // (the frames' parameters into the device table, their counters cleared: one kernel that reads the pinned table itself)
    if (!no_prologue)
        tlaunch(h, "k_frame_prologue", k_frame_prologue, dim3((unsigned int)n), dim3(128), 0, st, (const MlmFrame *)(h->h_frame_tab + base), h->d_frame_tab + base, h->d_ctr_all + base);
    unsigned int nb = 0;
    if (no_prologue) {
        nb = nb_all;
        const unsigned int ln = staged_list_len(h, S0);
        const int32_t *lp = ln ? F.pix : nullptr, *lr = ln ? F.raw : nullptr;
        const MlmFrame *hf = h->h_frame_tab + base;
        if (mode == 0) tlaunch(h, "k_bin_sectors", k_bin_sectors_hostf<0>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
        else if (mode == 1) tlaunch(h, "k_bin_sectors", k_bin_sectors_hostf<1>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
        else tlaunch(h, "k_bin_sectors", k_bin_sectors_hostf<2>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
    } else if (F.n > 0) {
        nb = nb_all;
        if (nb > P.nb_cap) {
            h->err = "frame geometry exceeds the queues sized from mlm_limits.max_points";
            return MLM_ERR_CAPACITY;
        }
        // (one strip per workgroup: two or four strips worked on together are 5 % quicker with nothing else on the GPU — 3.22 -> 3.06
        // us per frame — and cost the pipeline 6 %: 91.0 -> 85.3 k frames/s, profiles/r4b; MLM_BIN_STRIPS selects them for experiments)
        if (mode == 0)
            tlaunch(h, "k_bin_sectors", k_bin_sectors<0, 1>, dim3(nb, 1, n), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else if (mode == 1)
            tlaunch(h, "k_bin_sectors", k_bin_sectors<1, 1>, dim3(nb, 1, n), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else
            tlaunch(h, "k_bin_sectors", k_bin_sectors<2, 1>, dim3(nb, 1, n), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
    }
    {
        const int row_w = mode == 0 ? F.width : 64; // rows of the ranking bitmap
        unsigned long long dm;
        int ds;
        div_magic((unsigned int)row_w, dm, ds);
        unsigned long long rm;
        int rs;
        div_magic((unsigned int)P.nRho, rm, rs);
        const int big = P.sec_tab_big && h->big_armed > 0 ? 1 : 0; // the pass with the large cell table follows (see k_sector_big)
        if (h->big_armed > 0) --h->big_armed;
        // (a frame on its own is alone on the GPU: the 512-thread workgroup finishes a column sooner; the table is the same)
        const int nt = (n == 1 && P.sec_tab >= 512u) ? 512 : h->sec_threads;
        if (P.explore && nt == 256)
            tlaunch(h, "k_sector", k_sector<true, 256>, dim3((unsigned int)P.nPhi, 1, n), dim3(256), P.sec_lds_bytes, st, h->d_slot_tab,
                    h->d_frame_tab, base, mode == 0 ? F.width : 0, (int)nb, rm, rs, 1ull, big, dm, ds);
        else if (P.explore)
            tlaunch(h, "k_sector", k_sector<true, 512>, dim3((unsigned int)P.nPhi, 1, n), dim3(512), P.sec_lds_bytes, st, h->d_slot_tab,
                    h->d_frame_tab, base, mode == 0 ? F.width : 0, (int)nb, rm, rs, 1ull, big, dm, ds);
        else if (nt == 256)
            tlaunch(h, "k_sector", k_sector<false, 256>, dim3((unsigned int)P.nPhi, 1, n), dim3(256), P.sec_lds_bytes, st, h->d_slot_tab,
                    h->d_frame_tab, base, mode == 0 ? F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big, dm, ds);
        else
            tlaunch(h, "k_sector", k_sector<false, 512>, dim3((unsigned int)P.nPhi, 1, n), dim3(512), P.sec_lds_bytes, st, h->d_slot_tab,
                    h->d_frame_tab, base, mode == 0 ? F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big, dm, ds);
        if (big) { // the columns whose cell table overflowed, with the large table (a few workgroups per frame walk the list)
            if (P.explore)
                tlaunch(h, "k_sector_big", k_sector_big<true>, dim3(h->big_grid), dim3(MLM_SEC_THREADS), P.sec_big_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base, n,
                        mode == 0 ? F.width : 0, (int)nb, rm, rs, 1ull, dm, ds);
            else
                tlaunch(h, "k_sector_big", k_sector_big<false>, dim3(h->big_grid), dim3(MLM_SEC_THREADS), P.sec_big_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base, n,
                        mode == 0 ? F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, dm, ds);
        }
        const bool fused_chain = n == 1 && nb <= kFusedChainStrips; // (a small frame on its own: k_rank<true> runs the chains of the cells it ranks)
        // (frontier mode, a synchronous call's lone frame: the bucket-first pass of its containers rides along, explore_spec_begin)
        const MlmExOrder om = (on_main && n == 1 && P.explore) ? h->ex_om : MlmExOrder{};
        h->ex_om_launched = om.on != 0;
        if (fused_chain)
            tlaunch(h, "k_rank", k_rank<true>, dim3(1024, 1, n), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base, mode == 0 ? F.width : 0, row_w, dm, ds, om);
        else
            tlaunch(h, "k_rank", k_rank<false>, dim3(n > 4 ? h->rank_grid : 1024, 1, n), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base,
                    mode == 0 ? F.width : 0, row_w, dm, ds, om);
        // blocks per frame: about 400 ranked cells per block in a batch (a lane that finishes a chain draws the next cell; each
        // block builds the transposed odds table in LDS), as many as the last confirmed frame had; single frames spread wider
        unsigned int cg = h->chain_grid;
        if (!cg) {
            const long long cells = std::max<long long>(1, h->stats.n_multi_cells);
            cg = n > 4 ? (unsigned int)std::min<long long>(64, std::max<long long>(8, cells / 400)) : (unsigned int)std::min<long long>(128, std::max<long long>(16, cells / 128));
        }
        if (!h->no_spread && !fused_chain) {
            if (n > 4)
                tlaunch(h, "k_chain_lanes", k_chain_lanes<1>, dim3(cg, 1, n), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab, base, 128u);
            else // (a frame or a few on their own: the kernel lasts as long as the longest chain has rounds)
                tlaunch(h, "k_chain_lanes", k_chain_lanes<4>, dim3(cg, 1, n), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab, base, 64u);
        }
        // the frame's hits and misses grouped by voxel, tile by tile (needs the increments and keys of the kernels above)
        if (!P.explore)
            tlaunch(h, "k_tile", k_tile, dim3(n > 1 ? h->tile_grid : (unsigned int)(P.n_tiles <= 4096 ? P.n_tiles : 1024), 1, n), dim3(MLM_TILE_THREADS), h->tile_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base);
    }
    if (!on_main) HIPCHK(h, hipEventRecord(h->stage_a_done[set], st)); // (on the main stream the order is the stream's: an event between two launches is a 5 us gap)
    return MLM_OK;
}

// Stage B+C of one frame on the main stream.  n_bkt != 0: speculative single-epoch ordering inside k_voxelize;
// n_bkt == 0: hl_key was produced by order_hits_exact.
void launch_stage_bc(mlm_handle *h, MlmSlot &S, unsigned long long n_bkt) {
    const MlmDev &P = eff_params(h, S); // (a cell-table frame: see MlmSlot::Pfb)
    // exact keys: nothing to check; speculative relaunch: against the policy state the host holds NOW
    S.F.rehash_thr = n_bkt ? (unsigned int)std::min<size_t>(h->hit_pol._M_next_resize, 0xFFFFFFFFu) : 0xFFFFFFFFu;
    {
        tlaunch(h, "k_voxelize", k_voxelize, dim3(160, 1 + MLM_RAY_LISTS), dim3(MLM_BLOCK), 0, h->stream, P, S.F, n_bkt);
    }
    {
        tlaunch(h, "k_apply", k_apply, dim3(160, 1 + MLM_RAY_LISTS), dim3(MLM_BLOCK), 0, h->stream, P, S.F.seq, n_bkt ? 0 : 1);
    }
}

void fill_stats(mlm_handle *h, const MlmSlot &S) {
    const MlmCounters &c = *S.h_ctr;
    h->stats.n_points = c.n_points;
    h->stats.n_hit_cells = c.u_hit;
    unsigned int um = 0;
    for (int k = 0; k < MLM_RAY_LISTS; ++k) um += c.umiss_part[k][0];
    h->stats.n_miss_cells = um;
    h->stats.n_out_of_range = c.n_oor;
    h->stats.n_blocks = std::min<unsigned int>(h->h_g->n_blocks, (unsigned int)h->P.max_blocks);
    h->stats.hit_bucket_count = (int64_t)h->hit_n_bkt;
    h->stats.n_multi_cells = c.n_multi;
    h->stats.n_contrib_slots = c.n_contrib;
    int64_t ng = c.n_groups, nr = 0, na = 0;
    for (int k = 0; k < MLM_RAY_LISTS; ++k) {
        ng += c.node_cnt[k][0];
        nr += c.ray_cnt[k][0];
        na += c.ray_cnt[k][1];
    }
    h->stats.n_device_atomics = na;
    if (S.sector && !h->P.explore) {
        h->stats.n_miss_cells = c.mvox_cnt[3][0]; // (the reservation counter of the frame's miss list)
    }
    h->stats.n_groups = ng;
    h->stats.n_rays = nr;
    h->stats.n_spec_replays = h->n_spec_miss;
    h->stats.n_sector_fallbacks = h->n_sector_fallbacks;
    if (c.n_ov > 0) h->big_armed = h->big_arm_len; // the scene still overflows the small cell table: keep the second pass scheduled
    // (eight or more overflowed columns in the last frame of two confirmed batches in a row: the table is too small for the scene)
    h->ov_heavy = c.n_ov >= 8u ? h->ov_heavy + 1 : 0;
    if (h->ov_heavy >= 2) h->want_widen = true;
    h->stats.logit_bit_exact = h->P.logit_exact;
    h->stats.n_pool_grows = h->n_pool_grows;
    h->stats.n_graph_launches = h->n_graph_launches;
    h->stats.n_bin_exact_waves = c.bin_exact;
    h->stats.block_capacity = h->P.max_blocks;
}

int check_queues(mlm_handle *h, const MlmSlot &S) {
    const MlmCounters &c = *S.h_ctr;
    const MlmDev &P = eff_params(h, S);
    bool over = c.n_contrib > P.contrib_cap;
    for (int k = 0; k < MLM_RAY_LISTS; ++k)
        over = over || c.touch_cnt[k][0] > P.touch_cap || c.node_cnt[k][0] > P.node_cap || c.mc_cnt[k][0] > P.mc_cap;
    if (over) {
        h->err = "a per-frame device queue overflowed (raise mlm_limits.max_points)";
        return MLM_ERR_CAPACITY;
    }
    if (h->h_g->err) {
        h->err = "block pool or block hash table full (raise mlm_limits.max_blocks)";
        return MLM_ERR_CAPACITY;
    }
    return MLM_OK;
}

} // namespace
