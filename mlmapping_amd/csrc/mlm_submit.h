// mlm_submit.h — submission and confirmation of frames: the batched apply launch, speculative submission, drain with the replay of
// frames whose speculation failed, the single-frame HIP graph, run_slots.  Part of mlmap_hip.hip.
#pragma once
namespace {

// ---- submission / confirmation ---------------------------------------------------------------------------------
// Frames carry a monotonically increasing sequence number.  Stage B/C of a frame is submitted speculatively; the
// device flag g->fail_frame holds the first sequence number whose speculation did not hold (sticky), and every
// Stage B/C kernel of a frame >= it is a no-op.  `pending` lists submitted-but-unconfirmed frames in order.

// The map-dependent part of the frames in slots base..base+n (sector path): one launch, a workgroup per world tile of the box the
// frames' grids span (k_apply_tiles).  Frames whose poses lie far apart are applied in several launches so that the box stays small.
int launch_apply_tiles(mlm_handle *h, int base, int n, int f_begin = 0) {
    const MlmDev &P = h->slots[(size_t)base].P;
    const int sh = P.tile_sh, n_ty = P.n_tiles / P.n_tx;
    int j0 = std::max(0, f_begin);
    while (j0 < n) {
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0, j1 = j0;
        for (; j1 < n; ++j1) {
            const MlmFrame &F = h->slots[(size_t)(base + j1)].F;
            const int wx = F.lv_o[0] >> sh, wy = F.lv_o[1] >> sh, wz = F.lv_o[2];
            const int nx0 = j1 == j0 ? wx : std::min(x0, wx), nx1 = j1 == j0 ? wx : std::max(x1, wx);
            const int ny0 = j1 == j0 ? wy : std::min(y0, wy), ny1 = j1 == j0 ? wy : std::max(y1, wy);
            const int nz0 = j1 == j0 ? wz : std::min(z0, wz), nz1 = j1 == j0 ? wz : std::max(z1, wz);
            // (a launch's box of world tiles stays small, and its frames' z origins within one grid height: k_apply_tiles keeps
            // two grid heights of a tile's layers in LDS)
            if (j1 > j0 && ((long long)(nx1 - nx0 + P.n_tx) * (ny1 - ny0 + n_ty) > (1ll << 20) || nz1 - nz0 > P.lv_nz)) break;
            x0 = nx0;
            x1 = nx1;
            y0 = ny0;
            y1 = ny1;
            z0 = nz0;
            z1 = nz1;
        }
        const long long grid = (long long)(x1 - x0 + P.n_tx) * (y1 - y0 + n_ty);
        if (grid > 0x7FFFFFFFll) {
            h->err = "frame-local grid too large for one launch";
            return MLM_ERR_UNSUPPORTED;
        }
        // (the kernel derives the box from the frames [base + j0, base + j1) itself: it gets that range as ITS slot range)
        if (j1 - j0 == 1) // (one frame: nothing to keep in LDS between frames)
            tlaunch(h, "k_apply_single", k_apply_single, dim3(h->single_apply_grid, 1, 1), dim3(MLM_BLOCK), 0, h->stream, h->d_slot_tab, h->d_frame_tab, base + j0,
                    (MlmCounters *)nullptr, (MlmGlobal *)nullptr);
        else
            tlaunch(h, "k_apply_tiles", k_apply_tiles, dim3((unsigned int)grid), dim3(MLM_BLOCK),
                    (size_t)(P.lv_nz + (z1 - z0)) * 9u * (1u << (2 * sh)) + 16u, h->stream, h->d_slot_tab, h->d_frame_tab, base + j0, j1 - j0, 0, z1 - z0);
        j0 = j1;
    }
    return MLM_OK;
}
// A frame whose blocks k_tile could not create (the pool was full): create them now, growing the pool as often as it takes.
// Nothing may be in flight.  On return the frame's records carry their slots and the device's error flag is clear.
int fix_pool_short(mlm_handle *h, MlmSlot &R) {
    while (R.h_ctr->pool_short) {
        if (!h->pool_grow) {
            h->err = "block pool or block hash table full (raise mlm_limits.max_blocks)";
            return MLM_ERR_CAPACITY;
        }
        HIPCHK(h, hipMemcpy(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost));
        const size_t nb = std::min<size_t>(h->h_g->n_blocks, (size_t)h->P.max_blocks);
        if ((h->h_g->err & 1u) || nb + h->frame_block_bound > (size_t)h->P.max_blocks) { // (else: grown since, on account of an earlier frame)
            const int rc = grow_pool(h, (size_t)h->P.max_blocks + 2 * h->frame_block_bound);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_alloc_retry, dim3(256), dim3(MLM_BLOCK), 0, h->stream, R.P, R.F);
        hipLaunchKernelGGL(k_alloc_retry_done, dim3(1), dim3(64), 0, h->stream, R.P);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(R.h_ctr, R.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return MLM_OK;
}

int submit_batch(mlm_handle *h, int base, int n) {
    if (h->hit_n_bkt > h->max_buckets) {
        h->err = "emulated bucket count exceeds capacity";
        return MLM_ERR_CAPACITY;
    }
    const int set = base / (h->lim.max_batch);
    int rc;
    if (h->need_sized && h->hit_n_bkt > h->slots[(size_t)base].P.sbkt_cap && h->hit_n_bkt <= h->caps_worst.sbkt) {
        // (the slots' bucket-first tables are sized by need as well: the emulated container has outgrown them — a dozen times in a stream's life)
        rc = drain(h);
        if (rc == MLM_OK) rc = grow_sbkt(h, h->hit_n_bkt);
        if (rc) return rc;
    }
    // (a column record holds its tile's column in 13 bits: images up to 65 528 pixels wide; the slots' bucket-first tables hold sbkt_cap buckets)
    const bool sectors = h->use_sectors && h->sector_backoff == 0 && h->slots[(size_t)base].F.width <= MLM_SEC_MAX_WIDTH &&
                         h->hit_n_bkt <= h->slots[(size_t)base].P.sbkt_cap;
    if (h->sector_backoff > 0) --h->sector_backoff;
    if (!sectors) { // the cell-table path cannot replay a frame that ran out of blocks: room for everything in flight + this batch
        rc = ensure_free_blocks(h, (h->pending.size() + (size_t)n) * h->frame_block_bound);
        if (rc) return rc;
    }
    for (int j = 0; j < n; ++j) {
        MlmSlot &S = h->slots[(size_t)(base + j)];
        S.seq = h->next_seq++;
        S.F.seq = S.seq;
        S.F.rehash_thr = (unsigned int)std::min<size_t>(h->hit_pol._M_next_resize, 0xFFFFFFFFu);
    }
    for (int j = 0; j < n; ++j) {
        h->slots[(size_t)(base + j)].sector = sectors;
        h->slots[(size_t)(base + j)].keys_exact = false;
    }
    if (!sectors && share_ct(h)) {
        // the cell-table path's per-frame state exists once: every frame runs alone, Stage A and the two map-dependent kernels
        // back to back on the main stream (behind whatever the frames before it left there)
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->set_free[set], 0));
        if (h->async_mode) {
            // (an asynchronous call's inputs — image, pixel list, points — went up on the set's Stage A stream, where its Stage A was
            // expected to run: run_slots_inner; this one runs on the main stream, which must wait for them.  Found by the recovery fuzzer's
            // scenario in tests/test_gpu_small_frames.py: a list read before it had arrived, one run in twenty-five.)
            if (!h->upload_ev) HIPCHK(h, hipEventCreateWithFlags(&h->upload_ev, hipEventDisableTiming));
            HIPCHK(h, hipEventRecord(h->upload_ev, h->stream_as[set]));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->upload_ev, 0));
        }
        for (int j = 0; j < n; ++j) {
            MlmSlot &S = h->slots[(size_t)(base + j)];
            rc = launch_stage_a_batch(h, base + j, 1, true);
            if (rc) return rc;
            launch_stage_bc(h, S, h->hit_n_bkt);
            HIPCHK(h, hipMemcpyAsync(S.h_ctr, S.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream)); // (before the next frame's Stage A
            h->pending.push_back(&S);                                                                              // reuses nothing of it, but for symmetry)
        }
        HIPCHK(h, hipMemcpyAsync(h->h_gb[set], h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipEventRecord(h->batch_done[set], h->stream));
        HIPCHK(h, hipEventRecord(h->set_free[set], h->stream));
        h->set_pending[set] = n;
        return MLM_OK;
    }
    {
        // (a synchronous call has nothing to overlap its Stage A with: on the main stream, no dependency between two streams — which
        // costs a call up to 60 us on the first slot set, see launch_stage_a_sector)
        const bool on_main = !h->async_mode;
        Timed t(h, on_main ? h->stream : h->stream_as[set], "stage_a_batch");
        rc = sectors ? launch_stage_a_sector(h, base, n, on_main) : launch_stage_a_batch(h, base, n, on_main);
    }
    if (rc) return rc;
    if (h->async_mode) HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[set], 0)); // (a synchronous call's Stage A ran on this stream)
    if (sectors) {
        // ONE launch for the batch: Stage A has grouped every frame's hits and misses by voxel, tile by tile (k_apply_tiles)
        Timed t(h, h->stream, "stage_bc_batch");
        rc = launch_apply_tiles(h, base, n);
        if (rc) return rc;
        for (int j = 0; j < n; ++j) h->pending.push_back(&h->slots[(size_t)(base + j)]);
    } else {
        // launch j = k_apply of frame j-1 + k_voxelize of frame j (see k_apply_voxelize): n+1 launches for n frames
        Timed t(h, h->stream, "stage_bc_batch");
        // blocks per list: one item per thread for a frame like the last confirmed one (grid-stride loops take the rest)
        unsigned int scg = h->sc_grid;
        if (!h->sc_grid_fixed) {
            const long long items = std::max<long long>(h->stats.n_hit_cells, h->stats.n_miss_cells / MLM_RAY_LISTS);
            scg = (unsigned int)std::min<long long>(1024, std::max<long long>(h->sc_grid, (items * 5 / 4 + MLM_BLOCK - 1) / MLM_BLOCK));
        }
        for (int j = 0; j <= n; ++j) {
            MlmSlot &Sa = h->slots[(size_t)(base + (j > 0 ? j - 1 : 0))];
            MlmSlot &Sv = h->slots[(size_t)(base + (j < n ? j : n - 1))];
            tlaunch(h, "k_apply_voxelize", k_apply_voxelize, dim3(scg * (MLM_BLOCK / h->sc_block), 2 * (1 + MLM_RAY_LISTS)), dim3(h->sc_block), 0, h->stream, eff_params(h, Sa),
                    Sa.F.seq, j > 0 ? 1 : 0, eff_params(h, Sv), Sv.F, h->hit_n_bkt, j < n ? 1 : 0);
            if (j < n) h->pending.push_back(&Sv);
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, (size_t)n * sizeof(MlmCounters),
                             hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_gb[set], h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventRecord(h->batch_done[set], h->stream));
    HIPCHK(h, hipEventRecord(h->set_free[set], h->stream));
    h->set_pending[set] = n;
    return MLM_OK;
}

// A frame of the sector path gave up (sector_overflow >= 2) and is redone from its Stage A on.  The columns that finished, and the
// ones that failed, have cleared their chunk counts — but a crowded column that was WAITING for the large-table pass when another
// column gave the frame up (sector_overflow 1 raised to 2 or 3) still holds its chunks: binned again on top of them, its points would
// count twice (in the rerun, or — after a fall-back to the cell-table path — in the slot's next frame).  Nothing is in flight.
int forget_columns(mlm_handle *h, const MlmSlot &R) {
    HIPCHK(h, hipMemsetAsync(R.P.col_cnt, 0, (size_t)R.P.nPhi * sizeof(unsigned int), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

// A frame whose crowded columns overflowed the small cell table while the large-table pass was not scheduled (sector_overflow == 1:
// the columns kept their records and sit on the frame's overflow list; nothing of the frame has been grouped by tile or applied).
// Nothing is in flight: the large-table pass and the rest of Stage A run for this frame alone on the main stream, and the pass is
// scheduled for the submissions to come.  On return the frame's counters are on the host (sector_overflow 0, or 2 if even the
// large table was too small: then the cell-table path takes it).
int redo_overflow_columns(mlm_handle *h, MlmSlot &R) {
    const MlmDev &P = R.P;
    const int si = (int)(&R - h->slots.data());
    hipStream_t st = h->stream;
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipMemsetAsync(&P.ctr->sector_overflow, 0, sizeof(unsigned int), st));
    HIPCHK(h, hipMemsetAsync(&P.ctr->chain_next, 0, sizeof(unsigned int), st));
    HIPCHK(h, hipMemsetAsync(&P.ctr->big_next, 0, sizeof(unsigned int), st)); // (k_sector_big's task counter: this frame is the launch's first)
    const int tile_w = R.mode == 0 ? R.F.width : 0, row_w = R.mode == 0 ? R.F.width : 64;
    unsigned long long dm, rm;
    int ds, rs;
    div_magic((unsigned int)row_w, dm, ds);
    div_magic((unsigned int)P.nRho, rm, rs);
    tlaunch(h, "k_sector_big", k_sector_big<false>, dim3(h->big_grid), dim3(MLM_SEC_THREADS), P.sec_big_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, si, 1, tile_w, 0, rm,
            rs, (unsigned long long)h->hit_n_bkt, dm, ds);
    tlaunch(h, "k_rank", k_rank<false>, dim3(1024, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, si, tile_w, row_w, dm, ds, MlmExOrder{});
    tlaunch(h, "k_chain_lanes", k_chain_lanes<4>, dim3(64, 1, 1), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab, si, 64u);
    tlaunch(h, "k_tile", k_tile, dim3((unsigned int)(P.n_tiles <= 4096 ? P.n_tiles : 1024), 1, 1), dim3(MLM_TILE_THREADS), h->tile_lds_bytes, st, h->d_slot_tab,
            h->d_frame_tab, si);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(R.h_ctr, P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (h->big_armed <= 0) h->big_armed_from = h->next_seq; // (frames submitted from now on have the pass behind them)
    h->big_armed = h->big_arm_len;
    h->n_big_redos++;
    return MLM_OK;
}

int confirm_front(mlm_handle *h, int count) {
    for (int j = 0; j < count; ++j) {
        const int rc = check_queues(h, *h->pending[(size_t)j]);
        if (rc) return rc;
    }
    if (count > 0) {
        MlmSlot *last = h->pending[(size_t)count - 1];
        h->last_slot = (int)(last - h->slots.data());
        fill_stats(h, *last);
    }
    h->pending.erase(h->pending.begin(), h->pending.begin() + count);
    return MLM_OK;
}

// Wait for everything submitted, replay frames whose speculation failed, leave nothing pending.
int drain(mlm_handle *h, bool g_copied) {
    if (h->P.explore) return drain_explore(h);
    for (;;) {
        if (!g_copied) HIPCHK(h, hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
        bool seen = false;
        if (g_copied && h->wait_ticket && !h->timing) {
            // a single frame's graph: its last store is the ticket — poll it for a while (a frame takes 0.1-0.3 ms) instead of
            // sleeping in hipStreamSynchronize, whose wake-up would be a tenth of the call
            const volatile unsigned int *ticket = &h->h_g->pad;
            const auto t0 = std::chrono::steady_clock::now();
            clk_mark(h, 3);
            for (unsigned int spins = 0; !(seen = *ticket == h->wait_ticket); ++spins)
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        h->wait_ticket = 0u;
        g_copied = false;
        if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipGetLastError());
        const int f = h->h_g->fail_frame;
        // (a full pool is no error of the frames confirmed here: k_tile flags the frame whose blocks did not fit, the batch stops in
        // front of it and fix_pool_short below grows the pool)
        const unsigned int err_bits = h->h_g->err;
        if (h->pool_grow) h->h_g->err &= ~1u;
        size_t ok = 0;
        while (ok < h->pending.size() && h->pending[ok]->seq < f) ++ok;
        int rc = confirm_front(h, (int)ok);
        if (rc) return rc;
        if (h->pending.empty() && (err_bits & 1u) && h->pool_grow) { // (nothing left to fix it for: a path without replay overflowed)
            h->err = "block pool overflowed on a path that cannot be replayed";
            return MLM_ERR_CAPACITY;
        }
        if (h->pending.empty()) break;
        // pending.front() does not fit the emulated container without a rehash (replay its Stage B exactly), or one of
        // its azimuth sectors overflowed its LDS tables (redo its Stage A on the cell-table path first)
        MlmSlot &S = *h->pending.front();
        if (getenv("MLM_DEBUG_DRAIN"))
            fprintf(stderr, "[drain] fail at seq %d: u_hit %u thr %zu n_bkt %zu pending %zu overflow %u\n", S.seq, S.h_ctr->u_hit,
                    (size_t)h->hit_pol._M_next_resize, h->hit_n_bkt, h->pending.size(), S.h_ctr->sector_overflow);
        h->h_g->fail_frame = 0x7FFFFFFF;
        HIPCHK(h, hipMemcpyAsync(&h->P.g->fail_frame, &h->h_g->fail_frame, sizeof(int), hipMemcpyHostToDevice, h->stream));
        bool any_sector = share_ct(h); // (async mode holds up to three batches: a cell-table batch may be followed by sector batches)
        for (const MlmSlot *R : h->pending) any_sector = any_sector || R->sector;
        if (any_sector) {
            // Sector path: the frames in flight were binned into buckets with the bucket count of their submission, which
            // the rehash changes — every pending frame is finished with exact keys, in order (no further speculation).
            // Cell-table frames among them are finished the same way (k_voxelize would read hl_slot / hl_cid / hl_bkey,
            // which k_sector never writes for a frame of the sector path).
            for (size_t j = 0; j < h->pending.size(); ++j) {
                MlmSlot &R = *h->pending[j];
                // a sector frame whose Stage A gave up, or (shared cell-table state) a cell-table frame: the frames behind it
                // have run their Stage A over the same buffers since — it takes the cell-table path from its Stage A on, alone
                if (j == 0 && !R.h_ctr->sector_overflow) h->n_spec_miss++;
                if (R.sector && R.h_ctr->sector_overflow == 1u) { // columns wait for the large-table pass (it was not scheduled)
                    rc = redo_overflow_columns(h, R);
                    if (rc) return rc;
                }
                // a list of the frame's slot, sized by need, was too short: enlarge the slots (every pending frame's Stage A output moves
                // over) and run this frame's Stage A again on the sector path — nothing of the attempt is left (k_tile consumed what the
                // columns that did finish handed out).  Still short after a few rounds, or the lists at their worst case: the cell-table path.
                for (int round = 0; R.sector && R.h_ctr->sector_overflow == 3u && h->need_sized && round < 6; ++round) {
                    const int si = (int)(&R - h->slots.data());
                    const int set = si / (h->lim.max_batch);
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    rc = grow_slots(h, *R.h_ctr);
                    if (rc) return rc;
                    rc = forget_columns(h, R);
                    if (rc) return rc;
                    rc = launch_stage_a_sector(h, si, 1);
                    if (rc) return rc;
                    HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[set], 0));
                    HIPCHK(h, hipMemcpyAsync(R.h_ctr, R.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    HIPCHK(h, hipMemcpyAsync(&h->P.g->fail_frame, &h->h_g->fail_frame, sizeof(int), hipMemcpyHostToDevice, h->stream)); // (the rerun may have set it again)
                    if (R.h_ctr->sector_overflow == 1u) { // (its crowded columns now wait for the large table: as above)
                        rc = redo_overflow_columns(h, R);
                        if (rc) return rc;
                    }
                }
                if (R.h_ctr->sector_overflow || (!R.sector && share_ct(h))) {
                    if (R.sector) {
                        h->n_sector_fallbacks++;
                        note_fallback(h, R.seq);
                    }
                    const int si = (int)(&R - h->slots.data());
                    const int set = si / (h->lim.max_batch);
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    rc = ensure_free_blocks_idle(h, h->frame_block_bound); // (its k_voxelize cannot be replayed)
                    if (rc) return rc;
                    if (R.sector && (rc = forget_columns(h, R))) return rc;
                    // (k_tile has consumed the descriptors the columns that did finish handed out: nothing of the attempt is left)
                    rc = launch_stage_a_batch(h, si, 1);
                    if (rc) return rc;
                    HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[set], 0));
                    HIPCHK(h, hipMemcpyAsync(R.h_ctr, R.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    HIPCHK(h, hipMemcpyAsync(&h->P.g->fail_frame, &h->h_g->fail_frame, sizeof(int), hipMemcpyHostToDevice, h->stream));
                    R.sector = false;
                }
                if (h->pool_grow) h->h_g->err &= ~1u; // (fix_pool_short below)
                rc = check_queues(h, R);
                if (rc) return rc;
                if (!R.keys_exact) {
                    rc = order_hits_exact(h, R, R.h_ctr->u_hit, R.seq);
                    if (rc) return rc;
                    R.keys_exact = true;
                }
                const int si = (int)(&R - h->slots.data());
                if (R.sector) {
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    rc = fix_pool_short(h, R); // (k_tile found the pool full: the frame's blocks are created now)
                    if (rc) return rc;
                    R.F.flags |= MLM_FRAME_EXACT_KEYS;
                    h->h_frame_tab[si] = R.F;
                    HIPCHK(h, hipMemcpyAsync(h->d_frame_tab + si, h->h_frame_tab + si, sizeof(MlmFrame), hipMemcpyHostToDevice, h->stream));
                    rc = launch_apply_tiles(h, si, 1);
                    if (rc) return rc;
                } else {
                    launch_stage_bc(h, R, 0);
                }
                // the frame is finished before the next one starts (this is the rare path)
                HIPCHK(h, hipMemcpyAsync(R.h_ctr, R.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                HIPCHK(h, hipGetLastError());
                // (a LATER pending frame's k_tile may have found the pool full when the frames were first submitted — the flag is sticky on the
                // device until that frame's turn comes, fix_pool_short; with everything confirmed the top of the loop looks at it again)
                if (h->pool_grow) h->h_g->err &= ~1u;
                if (h->h_g->err) { // (pool full with growth off, a queue overflow)
                    rc = check_queues(h, R);
                    if (rc) return rc;
                }
            }
            h->h_g->fail_frame = 0x7FFFFFFF; // (every pending frame is applied)
            HIPCHK(h, hipMemcpyAsync(&h->P.g->fail_frame, &h->h_g->fail_frame, sizeof(int), hipMemcpyHostToDevice, h->stream));
            continue; // (the loop's synchronisation confirms them)
        }
        h->n_spec_miss++;
        rc = check_queues(h, S);
        if (rc) return rc;
        rc = order_hits_exact(h, S, S.h_ctr->u_hit, S.seq);
        if (rc) return rc;
        S.keys_exact = true;
        launch_stage_bc(h, S, 0);
        HIPCHK(h, hipMemcpyAsync(S.h_ctr, S.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
        // the later frames evaluated their device-side check against the OLD threshold: re-arm it from the host with
        // the new policy state (their unique-hit counts are known), then resubmit their Stage B/C
        int ff = 0x7FFFFFFF;
        for (size_t j = h->pending.size(); j-- > 1;)
            if (h->pending[j]->h_ctr->u_hit > h->hit_pol._M_next_resize) ff = h->pending[j]->seq;
        HIPCHK(h, hipStreamSynchronize(h->stream)); // h_g is about to be rewritten
        h->h_g->fail_frame = ff;
        HIPCHK(h, hipMemcpyAsync(&h->P.g->fail_frame, &h->h_g->fail_frame, sizeof(int), hipMemcpyHostToDevice, h->stream));
        for (size_t j = 1; j < h->pending.size(); ++j) {
            MlmSlot &R = *h->pending[j];
            launch_stage_bc(h, R, h->hit_n_bkt);
            HIPCHK(h, hipMemcpyAsync(R.h_ctr, R.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
        }
    }
    for (int k = 0; k < MLM_SETS; ++k) h->set_pending[k] = 0;
    if (h->next_seq > 0x3FFFFFFF) { // nothing in flight: sequence numbers restart, so the bucket table must forget them
        h->next_seq = 0;
        for (auto &S : h->slots)
            if (S.P.tile_dir) HIPCHK(h, hipMemsetAsync(S.P.tile_dir, 0xFF, 4 * (size_t)S.P.n_tiles * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipMemsetAsync(h->P.bkt64, 0xFF, 2 * h->max_buckets * sizeof(unsigned long long), h->stream));
    }
    return MLM_OK;
}

// The batch submitted on slot set `set` is complete on the device?  Confirm it without draining the newer one.
int finish_set(mlm_handle *h, int set) {
    const int n = h->set_pending[set];
    if (n == 0) return MLM_OK;
    HIPCHK(h, hipEventSynchronize(h->batch_done[set]));
    const int last_seq = h->pending[(size_t)n - 1]->seq;
    if (h->h_gb[set]->fail_frame > last_seq && !h->h_gb[set]->err) {
        h->h_g->n_blocks = h->h_gb[set]->n_blocks;
        h->h_g->err = 0;
        h->set_pending[set] = 0;
        return confirm_front(h, n);
    }
    return drain(h);
}

// After a failed call: re-arm the device flags so that the handle stays usable.  MLM_ERR_CAPACITY leaves the map as
// far as the failing frame got (blocks that did not fit the pool are published as "pool full" and stay unusable; frames
// that touch only existing blocks integrate normally afterwards).
void clear_device_error(mlm_handle *h) {
    MlmGlobal g{};
    if (hipMemcpy(&g, h->P.g, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess) return;
    g.n_blocks = std::min<unsigned int>(g.n_blocks, (unsigned int)h->P.max_blocks);
    g.err = 0;
    g.fail_frame = 0x7FFFFFFF;
    hipMemcpy(h->P.g, &g, sizeof(g), hipMemcpyHostToDevice);
    *h->h_g = g;
    for (int k = 0; k < MLM_SETS; ++k) *h->h_gb[k] = g;
}

// ... and the hand-over counters of the frame slots: the frames that were dropped with the failed call may have left
// descriptors for the columns / tiles (their consumers did not run), which the next frame in the same slot must not inherit.
void wipe_frame_grids(mlm_handle *h) {
    if (!h->use_sectors) return;
    for (auto &S : h->slots) {
        if (S.P.tile_cols) hipMemsetAsync(S.P.tile_cols, 0, (size_t)S.P.n_tiles * S.P.tile_words * sizeof(uint32_t), h->stream);
        if (S.P.col_cnt) hipMemsetAsync(S.P.col_cnt, 0, (size_t)S.P.nPhi * sizeof(unsigned int), h->stream);
    }
    hipStreamSynchronize(h->stream);
}

// The handle-level half of "this call goes through the single-frame graph" (sector path, the handle's own stream, synchronous
// mode, no per-kernel timing, nothing in flight): ONE predicate for the stream a call's inputs are uploaded on and for the
// submission itself — the frame-level half (geometry, bucket table) is added by single_fast_ok once the frame is described.
inline bool fast_handle_ok(const mlm_handle *h) {
    return h->use_graph && !h->P.explore && !h->async_mode && h->own_stream && h->timing == 0 && h->use_sectors && h->sector_backoff == 0 &&
           h->hit_n_bkt > 1 && h->pending.empty();
}
bool single_fast_ok(const mlm_handle *h, int n) {
    if (n != 1 || !fast_handle_ok(h)) return false;
    const MlmSlot &S = h->slots[(size_t)(h->cur_set * h->lim.max_batch)];
    return S.F.width <= MLM_SEC_MAX_WIDTH && h->hit_n_bkt <= S.P.sbkt_cap && S.F.n > 0;
}
// the stream uploads of a call's inputs go to: the one its Stage A will run on (a frame-level veto of the graph path is
// repaired by run_slots with an event between the two streams)
inline hipStream_t upload_stream(const mlm_handle *h) {
    return (fast_handle_ok(h) || !h->async_mode) ? h->stream : h->stream_as[h->cur_set]; // (synchronous calls: everything on the main stream)
}
// The launch sequence of ONE frame on stream `st` (the sector path, everything on one stream): a prologue kernel takes the frame's
// parameters from pinned host memory and clears the slot's counters (no_prologue — a small frame in a slot whose counters the frame
// before left clear —: the first Stage A kernel fetches the parameters itself), Stage A, k_apply_single, whose last workgroup writes
// the counters, the map-wide flags and — last — the completion ticket back to pinned memory and clears the slot's counters
// (mlm_hand_back).  Kernels only: a copy or fill node is 4-5 us of a lone frame's time.  Issued directly, or captured into a graph.
hipError_t enqueue_single_frame(mlm_handle *h, int base, unsigned int nb, int big, bool no_prologue, hipStream_t st) {
    const MlmSlot &S = h->slots[(size_t)base];
    const MlmDev &P = S.P;
    if (no_prologue) { // (a small frame behind a frame that left the slot's counters clear: k_bin_sectors_hostf)
        const MlmFrame *hf = h->h_frame_tab + base;
        const unsigned int ln = staged_list_len(h, S);
        const int32_t *lp = ln ? S.F.pix : nullptr, *lr = ln ? S.F.raw : nullptr;
        if (S.mode == 0) hipLaunchKernelGGL(k_bin_sectors_hostf<0>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
        else if (S.mode == 1) hipLaunchKernelGGL(k_bin_sectors_hostf<1>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
        else hipLaunchKernelGGL(k_bin_sectors_hostf<2>, dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, hf, h->d_frame_tab + base, base, nb, lp, lr, ln);
    } else {
        hipLaunchKernelGGL(k_frame_prologue, dim3(1), dim3(128), 0, st, (const MlmFrame *)(h->h_frame_tab + base), h->d_frame_tab + base, h->d_ctr_all + base);
        if (S.mode == 0) hipLaunchKernelGGL((k_bin_sectors<0, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else if (S.mode == 1) hipLaunchKernelGGL((k_bin_sectors<1, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else hipLaunchKernelGGL((k_bin_sectors<2, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
    }
    const int row_w = S.mode == 0 ? S.F.width : 64;
    unsigned long long dm, rm;
    int ds, rs;
    div_magic((unsigned int)row_w, dm, ds);
    div_magic((unsigned int)P.nRho, rm, rs);
    // (a single frame is alone on the GPU: the 512-thread workgroup finishes a column sooner; the table is the same)
    if (h->sec_threads == 256 && P.sec_tab < 512u)
        hipLaunchKernelGGL((k_sector<false, 256>), dim3((unsigned int)P.nPhi, 1, 1), dim3(256), P.sec_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base,
                           S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big | (2 << 4), dm, ds); // (| 2 << 4: sixteen lanes per ray)
    else
        hipLaunchKernelGGL((k_sector<false, 512>), dim3((unsigned int)P.nPhi, 1, 1), dim3(512), P.sec_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base,
                           S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big | (2 << 4), dm, ds); // (| 2 << 4: sixteen lanes per ray)
    if (big)
        hipLaunchKernelGGL(k_sector_big<false>, dim3(h->big_grid), dim3(MLM_SEC_THREADS), P.sec_big_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base, 1,
                           S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, dm, ds);
    if (nb <= kFusedChainStrips) { // (a small frame on its own: the wave that ranks a cell runs its chain, no k_chain_lanes — k_rank<true>)
        hipLaunchKernelGGL(k_rank<true>, dim3(h->single_rank_grid, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base, S.mode == 0 ? S.F.width : 0, row_w, dm, ds, MlmExOrder{});
    } else {
        hipLaunchKernelGGL(k_rank<false>, dim3(256, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base, S.mode == 0 ? S.F.width : 0, row_w, dm, ds, MlmExOrder{});
        if (!h->no_spread)
            hipLaunchKernelGGL(k_chain_lanes<4>, dim3(h->single_chain_grid, 1, 1), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab, base, 64u);
    }
    hipLaunchKernelGGL(k_tile, dim3((unsigned int)(P.n_tiles <= 4096 ? P.n_tiles : 1024), 1, 1), dim3(MLM_TILE_THREADS), h->tile_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base);
    hipLaunchKernelGGL(k_apply_single, dim3(h->single_apply_grid, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base, h->h_ctr_all + base, h->h_g);
    return hipGetLastError();
}

// A single frame in synchronous mode: that sequence as ONE replay of a HIP graph (issued launch by launch it measured 8 us slower
// per call: profiles/README.md, r4o).  The calling thread then polls the ticket
// the last workgroup of k_apply_single writes (drain).
int submit_single_graph(mlm_handle *h, int base) {
    MlmSlot &S = h->slots[(size_t)base];
    const MlmDev &P = S.P;
    const int set = base / h->lim.max_batch;
    S.seq = h->next_seq++;
    S.F.seq = S.seq;
    S.F.rehash_thr = (unsigned int)std::min<size_t>(h->hit_pol._M_next_resize, 0xFFFFFFFFu);
    S.sector = true;
    S.keys_exact = false;
    const unsigned int nb = S.mode == 0 ? (unsigned int)(((S.F.width + 31) / 32) * ((S.F.height + 7) / 8)) : (unsigned int)(((size_t)S.F.n + 255) / 256);
    if (nb > P.nb_cap) {
        h->err = "frame geometry exceeds the queues sized from mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    const int big = P.sec_tab_big && h->big_armed > 0 ? 1 : 0;
    if (h->big_armed > 0) --h->big_armed;
    h->h_frame_tab[base] = S.F;
    h->h_g->pad = 0u; // (k_apply_single ends with a ticket in the host copy of the map-wide flags: drain polls it)
    h->wait_ticket = (unsigned int)S.F.seq + 1u;
    // The slot's last frame went through this graph and was applied: its last workgroup left the device counters clear and said so
    // (MLM_CTR_CLEARED; every other use of the slot copies the real counters over the mark, or withdraws it: launch_stage_a_*).  A small frame
    // then starts without the prologue kernel — its first kernel takes the parameters from pinned memory itself.
    const bool no_prologue = nb <= kHostFrameStrips && S.h_ctr->apply_done == MLM_CTR_CLEARED;
    S.h_ctr->apply_done = 0u; // (consumed: the counters are in use from here on)
    mlm_handle::SingleGraph *G = nullptr;
    for (auto &g : h->graphs)
        if (g.mode == S.mode && g.width == S.F.width && g.height == S.F.height && g.base == base && g.nb == nb && g.sec_tab == P.sec_tab && g.n_bkt == h->hit_n_bkt && g.big == big &&
            g.no_prologue == no_prologue && (!no_prologue || (g.list == (staged_list_len(h, S) ? S.F.pix : nullptr) && g.list_n == staged_list_len(h, S))))
            G = &g;
    if (!G) {
        if (h->graphs.size() >= 16) { // (a handful of frame geometries at most; the bucket count of the emulated container changes a dozen times per stream)
            for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
            h->graphs.clear();
        }
        hipStream_t st = h->stream;
        HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const hipError_t e = enqueue_single_frame(h, base, nb, big, no_prologue, st);
        hipGraph_t graph = nullptr;
        const hipError_t e2 = hipStreamEndCapture(st, &graph);
        if (e != hipSuccess || e2 != hipSuccess || !graph) {
            if (graph) hipGraphDestroy(graph);
            h->err = std::string("single-frame graph capture: ") + hipGetErrorString(e != hipSuccess ? e : e2);
            return MLM_ERR_HIP;
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t e3 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e3 != hipSuccess) {
            h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e3);
            return MLM_ERR_HIP;
        }
        h->graphs.push_back(mlm_handle::SingleGraph{S.mode, S.F.width, S.F.height, base, big, nb, P.sec_tab, h->hit_n_bkt, no_prologue, staged_list_len(h, S) ? S.F.pix : nullptr, staged_list_len(h, S), exec});
        G = &h->graphs.back();
    }
    clk_mark(h, 1);
    HIPCHK(h, hipGraphLaunch(G->exec, h->stream));
    clk_mark(h, 2);
    h->n_graph_launches++;
    h->pending.push_back(&S);
    h->set_pending[set] = 1;
    return MLM_OK;
}

// Integrate the frames already described in slots[base..base+n) (F, mode set), in order.
int run_slots_inner(mlm_handle *h, int n);
void mirror_eager(mlm_handle *h);
int run_slots(mlm_handle *h, int n) {
    const int rc = run_slots_inner(h, n);
    if (rc == MLM_OK) mirror_eager(h); // (a planner that queries after every frame: its refresh of the host mirror starts now, mlm_mirror.h)
    return rc;
}
int run_slots_inner(mlm_handle *h, int n) {
    (void)hipGetLastError(); // a stale error of unrelated HIP calls in this thread is not ours
    mirror_mark_frames(h, n); // (the host mirror of the map is stale inside these frames' reach: mlm_mirror.h)
    if (h->want_widen) {
        h->want_widen = false;
        h->ov_heavy = 0;
        const int rc = widen_sec_tab(h);
        if (rc) return rc;
    }
    if (h->timing == 1) { // per-call mode: the list describes the last call only
        h->ktimes.clear();
        h->kpool_used = 0;
    }
    if (h->P.explore) { // frontier mode: exact ordering of both containers, no speculation
        const int K = h->lim.max_batch;
        {
            // the call's inputs went up on one stream, its Stage A runs on another (synchronous calls: the main stream): order it behind
            hipStream_t target = h->async_mode ? h->stream_as[h->cur_set] : h->stream;
            if (h->last_upload && h->last_upload != target) {
                hipError_t e = hipSuccess;
                if (!h->upload_ev) e = hipEventCreateWithFlags(&h->upload_ev, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(h->upload_ev, h->last_upload);
                if (e == hipSuccess) e = hipStreamWaitEvent(target, h->upload_ev, 0);
                if (e != hipSuccess) {
                    h->err = std::string("ordering the upload: ") + hipGetErrorString(e);
                    return MLM_ERR_HIP;
                }
            }
            h->last_upload = nullptr;
        }
        {
            size_t in_flight = 0;
            for (const auto &b : h->ex_q) in_flight += (size_t)b.n;
            const int rc = ensure_free_blocks(h, (in_flight + (size_t)n) * h->frame_block_bound); // (no replay in this mode)
            if (rc) return rc;
        }
        const int base = h->cur_set * K;
        if (h->async_mode) {
            const int set = h->cur_set;
            int rc = explore_stage_a(h, base, n);
            if (rc == MLM_OK) {
                hipError_t e = hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, (size_t)n * sizeof(MlmCounters), hipMemcpyDeviceToHost,
                                              h->stream_as[set]);
                if (e == hipSuccess) e = hipEventRecord(h->ex_counts[set], h->stream_as[set]);
                if (e != hipSuccess) {
                    h->err = std::string("frontier batch: ") + hipGetErrorString(e);
                    rc = MLM_ERR_HIP;
                }
            }
            if (rc == MLM_OK) {
                h->ex_q.push_back(mlm_handle::ExBatch{set, n, false});
                for (size_t k = 0; k + 1 < h->ex_q.size() && rc == MLM_OK; ++k) // everything but the batch just submitted
                    if (!h->ex_q[k].bc_enqueued) rc = explore_enqueue_bc(h, h->ex_q[k]);
                h->cur_set = (set + 1) % h->n_sets;
                // the set that is filled next must have been confirmed (its host-side counters are reused)
                while (rc == MLM_OK && !h->ex_q.empty() && h->ex_q.front().set == h->cur_set) rc = explore_confirm_front(h);
            }
            if (rc != MLM_OK) explore_fail_epilogue(h);
            return rc;
        }
        // Stage A of all frames in one launch sequence (it does not depend on the map), one synchronisation to learn the
        // frames' hit/miss counts, then the map-dependent part frame by frame without further synchronisation
        auto sync_path = [&]() -> int {
            // one frame, both emulated containers past their first insertion, nothing deferred: the map-dependent part goes out behind the
            // frame's Stage A at once, guarded on the device by the condition the host checks afterwards (explore_spec_begin,
            // explore_stage_bc_spec); its last launch hands the counters back
            const bool spec = n == 1 && h->ex_spec && !h->ex_tail && h->hit_pol._M_next_resize >= 1 && h->miss_pol._M_next_resize >= 1 &&
                              h->hit_n_bkt <= h->max_buckets && h->miss_n_bkt <= h->max_buckets;
            h->ex_om.on = 0;
            int rc = spec ? explore_spec_begin(h) : MLM_OK;
            if (rc) return rc;
            rc = explore_stage_a(h, base, n, true); // (on the main stream: nothing to overlap with in a synchronous call)
            if (rc) return rc;
            if (!spec) HIPCHK(h, hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, (size_t)n * sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
            if (spec) {
                unsigned int thr[2];
                rc = explore_stage_bc_spec(h, base, thr);
                if (rc) return rc;
                HIPCHK(h, wait_for_ticket(h, h->stream));
                HIPCHK(h, hipGetLastError());
                MlmSlot &S = h->slots[(size_t)base];
                const MlmCounters &c = *S.h_ctr;
                if (c.sector_overflow == 0u && c.u_hit <= thr[0] && c.n_ex_miss <= thr[1]) {
                    S.ex_um = c.n_ex_miss;
                    h->stats.n_rehash_epochs = 1;
                    HIPCHK(h, hipEventRecord(h->set_free[h->cur_set], h->stream));
                    return explore_finish(h, base);
                }
                h->n_spec_miss++; // (nothing of the map-dependent part ran: the general path follows, with the counts on the host)
            } else {
                HIPCHK(h, mlm_spin_sync(h->stream));
                HIPCHK(h, hipGetLastError());
            }
            rc = explore_redo_overflows(h, base, n);
            if (rc) return rc;
            for (int j = 0; j < n; ++j) {
                rc = explore_stage_bc(h, base + j);
                if (rc) return rc;
            }
            rc = explore_end_batch(h);
            if (rc) return rc;
            HIPCHK(h, hipEventRecord(h->set_free[h->cur_set], h->stream));
            HIPCHK(h, mlm_spin_sync(h->stream));
            HIPCHK(h, hipGetLastError());
            for (int j = 0; j < n; ++j) {
                rc = explore_finish(h, base + j);
                if (rc) return rc;
            }
            return MLM_OK;
        };
        const int rc = sync_path();
        if (rc != MLM_OK) explore_fail_epilogue(h); // (the handle stays usable after MLM_ERR_CAPACITY in this mode too)
        return rc;
    }
    h->stats.n_rehash_epochs = 1;
    const int K = h->lim.max_batch;
    const int set = h->cur_set;
    int rc;
    const bool fast = single_fast_ok(h, n);
    {
        // the call's inputs went up on one stream, its Stage A may run on another: order it behind
        hipStream_t target = (fast || !h->async_mode) ? h->stream : h->stream_as[set];
        if (h->last_upload && h->last_upload != target) {
            hipError_t e = hipSuccess;
            if (!h->upload_ev) e = hipEventCreateWithFlags(&h->upload_ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(h->upload_ev, h->last_upload);
            if (e == hipSuccess) e = hipStreamWaitEvent(target, h->upload_ev, 0);
            if (e != hipSuccess) {
                h->err = std::string("ordering the upload: ") + hipGetErrorString(e);
                return MLM_ERR_HIP;
            }
        }
        h->last_upload = nullptr;
    }
    if (fast) {
        rc = submit_single_graph(h, set * K);
        if (rc == MLM_OK) rc = drain(h, true); // (the graph ends with the read-back of the map-wide flags)
    } else {
        rc = submit_batch(h, set * K, n);
        if (rc != MLM_OK) {
        } else if (h->async_mode && h->hit_n_bkt > 1) {
            // confirm the OLDEST batch in flight (the set that will be refilled next); the newer ones keep the GPU busy
            h->cur_set = (set + 1) % h->n_sets;
            rc = finish_set(h, h->cur_set);
        } else { // (also the very first batch of a stream: its first frame always grows the emulated container from empty)
            rc = drain(h);
        }
    }
    if (rc != MLM_OK) { // leave a defined state behind
        hipDeviceSynchronize();
        h->pending.clear();
        for (int k = 0; k < MLM_SETS; ++k) h->set_pending[k] = 0;
        clear_device_error(h);
        wipe_frame_grids(h);
    }
    return rc;
}
inline MlmSlot &cur_slot(mlm_handle *h, int j) { return h->slots[(size_t)(h->cur_set * (h->lim.max_batch) + j)]; }

} // namespace
