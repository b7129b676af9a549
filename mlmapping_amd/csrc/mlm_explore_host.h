// mlm_explore_host.h — host side of frontier mode (use_exploration_frontiers: true): exact ordering of both emulated containers,
// the map-dependent launches per frame, asynchronous batches.  Part of mlmap_hip.hip.
#pragma once
namespace {

// ---- frontier mode (use_exploration_frontiers: true): one frame at a time, exact ordering of BOTH containers ------
// Iteration-order keys of miss_idx_set (std::unordered_set<size_t>) into ex_key; same scheme as order_hits_exact.
int order_misses_exact(mlm_handle *h, MlmSlot &S, unsigned int U) {
    const MlmDev &P = S.P;
    const auto ep = plan_epochs_for(h->miss_pol, h->miss_n_bkt, U);
    if (h->miss_n_bkt > h->max_buckets) {
        h->err = "emulated bucket count exceeds capacity";
        return MLM_ERR_CAPACITY;
    }
    if (U == 0) return MLM_OK;
    const bool multi = ep.size() > 1;
    if (multi) {
        tlaunch(h, "k_ex_time_keys", k_ex_time_keys, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sk_in, h->sv_in);
        if (mlm_sort_pairs_u64_u32(h->sort_tmp, h->sort_tmp_bytes, h->sk_in, h->sk_out, h->sv_in, h->sv_out, U, h->stream) != 0) {
            h->err = "radix sort failed";
            return MLM_ERR_HIP;
        }
        tlaunch(h, "k_ex_assign_rank", k_ex_assign_rank, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sv_out, U, 1);
    }
    for (size_t e = 0; e < ep.size(); ++e) {
        const unsigned int m = (unsigned int)ep[e].first;
        const unsigned long long nb = ep[e].second;
        const bool final_pass = (e + 1 == ep.size());
        HIPCHK(h, hipMemsetAsync(P.bktm_first, 0xFF, nb * sizeof(uint32_t), h->stream));
        tlaunch(h, "k_ex_bucket_min", k_ex_bucket_min, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, nb, m, multi ? 1 : 0);
        tlaunch(h, "k_ex_make_keys", k_ex_make_keys, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, nb, m, multi ? 1 : 0,
                           final_pass ? 1 : 0, h->sk_in, h->sv_in);
        if (!final_pass) {
            if (mlm_sort_pairs_u64_u32(h->sort_tmp, h->sort_tmp_bytes, h->sk_in, h->sk_out, h->sv_in, h->sv_out, U, h->stream) != 0) {
                h->err = "radix sort failed";
                return MLM_ERR_HIP;
            }
            tlaunch(h, "k_ex_assign_rank", k_ex_assign_rank, dim3(kListGrid), dim3(MLM_BLOCK), 0, h->stream, P, U, h->sv_out, m, 0);
        }
    }
    return MLM_OK;
}

// Frontier mode, the part of a frame that needs the map (main stream, no host synchronisation): exact iteration order
// of both containers (the host replays the two rehash policies from the frame's counts in S.h_ctr), hits, then the
// miss-side frontier bookkeeping and the release scan.  Ends with the asynchronous read-back of the counters.
// the deferred tail of the previous frame as launches of its own (end of a batch, or before a general ordering replay)
void explore_flush_tail(mlm_handle *h) {
    if (!h->ex_tail) return;
    const dim3 blk(MLM_BLOCK);
    tlaunch(h, "k_ex_apply_misses", k_ex_apply_misses, dim3(kListGrid), blk, 0, h->stream, h->ex_tail->P, (MlmCounters *)nullptr, (MlmGlobal *)nullptr, 0u);
    tlaunch(h, "k_ex_release", k_ex_release, dim3(1024), blk, 0, h->stream, h->ex_tail->P);
    h->ex_tail = nullptr;
}

int explore_stage_bc(mlm_handle *h, int slot_index) {
    MlmSlot &S = h->slots[(size_t)slot_index];
    const MlmDev &P = S.P;
    hipStream_t st = h->stream;
    const dim3 blk(MLM_BLOCK);
    const unsigned int U = S.h_ctr->u_hit, UM = S.h_ctr->n_ex_miss;
    S.ex_um = UM;
    // iteration-order keys of both containers.  Neither rehashes in a typical frame: then one fused pair of launches
    // (tagged bucket-first tables, nothing to clear) that also carries the previous frame's miss phase and release scan;
    // otherwise the general epoch-by-epoch replay per container.
    int rc = MLM_OK;
    {
        std::__detail::_Prime_rehash_policy hp = h->hit_pol, mp = h->miss_pol;
        size_t hn = h->hit_n_bkt, mn = h->miss_n_bkt;
        const auto eh = plan_epochs_for(hp, hn, U);
        const auto em = plan_epochs_for(mp, mn, UM);
        if (eh.size() == 1 && em.size() == 1 && hn <= h->max_buckets && mn <= h->max_buckets) {
            h->hit_pol = hp;
            h->miss_pol = mp;
            h->hit_n_bkt = hn;
            h->miss_n_bkt = mn;
            h->stats.n_rehash_epochs = 1;
            const int tag = h->ex_tag++;
            if (h->ex_tag > 0x3FFFFFFF) { // tags restart: the tables must forget them
                h->ex_tag = 0;
                HIPCHK(h, hipMemsetAsync(h->P.bkt64, 0xFF, 2 * h->max_buckets * sizeof(unsigned long long), st));
            }
            const MlmDev &Pp = h->ex_tail ? h->ex_tail->P : P;
            const unsigned int rows = h->ex_tail ? 3u : 2u;
            tlaunch(h, "k_ex_order_min", k_ex_order_min, dim3(kListGrid, rows), blk, 0, st, P, U, UM, (unsigned long long)hn, (unsigned long long)mn, tag, Pp);
            tlaunch(h, "k_ex_order_keys", k_ex_order_keys, dim3(kListGrid, rows), blk, 0, st, P, U, UM, (unsigned long long)hn, (unsigned long long)mn, Pp);
            h->ex_tail = nullptr;
        } else {
            explore_flush_tail(h);
            rc = order_hits_exact(h, S, U, 0);
            if (rc) return rc;
            rc = order_misses_exact(h, S, UM);
            if (rc) return rc;
        }
    }
    tlaunch(h, "k_ex_register", k_ex_register, dim3(4 * kListGrid, 2), blk, 0, st, P, S.F); // hits: push on voxel lists; misses: count + tau
    tlaunch(h, "k_apply", k_apply, dim3(64, 1), blk, 0, st, P, 0, 1);           // hits: ordered replay, frontier erase on 'o'
    tlaunch(h, "k_ex_observe", k_ex_observe, dim3(4 * kListGrid), blk, 0, st, P, S.F);
    h->ex_tail = &S; // its miss phase and release scan ride with the next frame's ordering launches (or explore_flush_tail)
    return MLM_OK;
}
// ONE synchronous frame, enqueued before the host has seen its counts: the whole map-dependent part — ordering, registration, hits,
// observation, misses, release scan — behind the frame's Stage A without a synchronisation in between (the host used to wait for the
// counts, replay both rehash policies and only then enqueue seven launches: the GPU idled through a copy, a wake-up and the first
// launch's way down).  The kernels read the counts themselves and do nothing unless the frame fits both containers as they are
// (MlmDev::spec_on, mlm_ex_spec_skip); the host checks the same condition once the frame has drained (wait_for_ticket) and, if it did not hold — the
// first frames of a stream, while the emulated containers still grow, or a Stage A that left the sector path —, runs the general
// path with the counts it now has.  Returns with everything enqueued; thresholds in thr[2].
// explore_spec_begin (before the frame's Stage A is enqueued): thresholds, tag and bucket counts of the frame — h->ex_om, which the frame's
// k_rank takes along (launch_stage_a_sector) so that the bucket-first pass needs no launch of its own.
int explore_spec_begin(mlm_handle *h) {
    MlmExOrder &om = h->ex_om;
    om.thr_hit = (unsigned int)std::min<size_t>(h->hit_pol._M_next_resize, 0xFFFFFFFFu);
    om.thr_miss = (unsigned int)std::min<size_t>(h->miss_pol._M_next_resize, 0xFFFFFFFFu);
    if (h->ex_spec == 2) om.thr_hit = om.thr_miss = 0u; // (test knob: every frame with a hit or a miss cell misses the speculation)
    om.tag = h->ex_tag++;
    if (h->ex_tag > 0x3FFFFFFF) { // tags restart: the tables must forget them
        h->ex_tag = 0;
        HIPCHK(h, hipMemsetAsync(h->P.bkt64, 0xFF, 2 * h->max_buckets * sizeof(unsigned long long), h->stream));
    }
    om.nb_hit = h->hit_n_bkt;
    om.nb_miss = h->miss_n_bkt;
    om.on = 1;
    h->ex_om_launched = false;
    return MLM_OK;
}
int explore_stage_bc_spec(mlm_handle *h, int slot_index, unsigned int thr[2]) {
    MlmSlot &S = h->slots[(size_t)slot_index];
    hipStream_t st = h->stream;
    const dim3 blk(MLM_BLOCK);
    const MlmExOrder om = h->ex_om;
    h->ex_om.on = 0;
    thr[0] = om.thr_hit;
    thr[1] = om.thr_miss;
    MlmDev Ps = S.P;
    Ps.spec_on = 1;
    Ps.spec_hit_thr = thr[0];
    Ps.spec_miss_thr = thr[1];
    const int tag = om.tag;
    const unsigned long long hn = om.nb_hit, mn = om.nb_miss;
    if (!h->ex_om_launched) // (the frame's Stage A did not go through launch_stage_a_sector: the cell-table path)
        tlaunch(h, "k_ex_order_min", k_ex_order_min, dim3(kListGrid, 2), blk, 0, st, Ps, 0u, 0u, hn, mn, tag, Ps);
    h->ex_om_launched = false;
    tlaunch(h, "k_ex_order_keys", k_ex_order_keys, dim3(kListGrid, 2), blk, 0, st, Ps, 0u, 0u, hn, mn, Ps);
    tlaunch(h, "k_ex_register", k_ex_register, dim3(4 * kListGrid, 2), blk, 0, st, Ps, S.F);
    tlaunch(h, "k_apply", k_apply, dim3(64, 1), blk, 0, st, Ps, 0, 1);
    tlaunch(h, "k_ex_observe", k_ex_observe, dim3(4 * kListGrid), blk, 0, st, Ps, S.F);
    // (the frame's counters, the map-wide flags and the completion ticket go to the host with the last workgroup of the miss phase — 128
    // workgroups: the hand-back counts their arrivals on one word, ~10 ns each —; the release scan runs behind it)
    h->h_g->pad = 0u;
    h->wait_ticket = (unsigned int)S.F.pad2 + 1u;
    tlaunch(h, "k_ex_apply_misses", k_ex_apply_misses, dim3(128), blk, 0, st, Ps, h->h_ctr_all + slot_index, h->h_g, h->wait_ticket);
    tlaunch(h, "k_ex_release", k_ex_release, dim3(128), blk, 0, st, Ps);
    return MLM_OK;
}
// end of a batch (or of a single frame): the last frame's tail, the map-wide counters.  (Deferring that tail to the next synchronous
// call's ordering launches was tried in round 5: the next call then needs another slot set — cold lists — and came out 5 % slower.)
int explore_end_batch(mlm_handle *h) {
    explore_flush_tail(h);
    HIPCHK(h, hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
    return MLM_OK;
}
// after the stream has been synchronised
int explore_finish(mlm_handle *h, int slot_index) {
    MlmSlot &S = h->slots[(size_t)slot_index];
    const int rc = check_queues(h, S);
    if (rc) return rc;
    h->last_slot = slot_index;
    fill_stats(h, S);
    h->stats.n_miss_cells = S.ex_um;
    h->stats.hit_bucket_count = (int64_t)h->hit_n_bkt;
    return MLM_OK;
}

// Frontier mode, asynchronous submission: Stage A of a batch runs while the map-dependent part of the batch before it is
// enqueued — the host needs the frames' hit / miss counts (it replays both containers' rehash policies) before it can
// enqueue that part, so a batch's second half is always one call behind its first.
int explore_redo_overflows(mlm_handle *h, int base, int n);
int explore_enqueue_bc(mlm_handle *h, mlm_handle::ExBatch &b) {
    const int K = h->lim.max_batch, base = b.set * K;
    HIPCHK(h, hipEventSynchronize(h->ex_counts[b.set])); // the frames' counters are on the host
    {
        const int rc = explore_redo_overflows(h, base, b.n);
        if (rc) return rc;
    }
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[b.set], 0));
    for (int j = 0; j < b.n; ++j) {
        const int rc = explore_stage_bc(h, base + j);
        if (rc) return rc;
    }
    {
        const int rc = explore_end_batch(h);
        if (rc) return rc;
    }
    HIPCHK(h, hipEventRecord(h->set_free[b.set], h->stream));
    HIPCHK(h, hipEventRecord(h->ex_bc_done[b.set], h->stream));
    b.bc_enqueued = true;
    return MLM_OK;
}
int explore_confirm_front(mlm_handle *h) {
    const mlm_handle::ExBatch b = h->ex_q.front();
    h->ex_q.pop_front();
    const int K = h->lim.max_batch;
    HIPCHK(h, hipEventSynchronize(h->ex_bc_done[b.set]));
    HIPCHK(h, hipGetLastError());
    for (int j = 0; j < b.n; ++j) {
        const int rc = explore_finish(h, b.set * K + j);
        if (rc) return rc;
    }
    return MLM_OK;
}
void clear_device_error(mlm_handle *h);
// Frontier mode after a failed call: nothing stays queued, the deferred tail is dropped, the device flags are re-armed —
// the handle stays usable (what the default path's epilogue in run_slots does)
void explore_fail_epilogue(mlm_handle *h) {
    hipDeviceSynchronize();
    h->ex_q.clear();
    h->ex_tail = nullptr;
    clear_device_error(h);
}
int drain_explore(mlm_handle *h) {
    int rc = MLM_OK;
    for (auto &b : h->ex_q)
        if (!b.bc_enqueued && rc == MLM_OK) rc = explore_enqueue_bc(h, b);
    while (rc == MLM_OK && !h->ex_q.empty()) rc = explore_confirm_front(h);
    if (rc == MLM_OK && hipStreamSynchronize(h->stream) != hipSuccess) {
        h->err = "hipStreamSynchronize failed";
        rc = MLM_ERR_HIP;
    }
    if (rc != MLM_OK) explore_fail_epilogue(h);
    return rc;
}

// Frontier mode, Stage A of the slots base..base+n: by azimuth sector when the handle can (k_sector<true>), else (and for
// frames whose sector tables overflowed, explore_redo_overflows) on the cell-table path.
int explore_stage_a(mlm_handle *h, int base, int n, bool on_main = false) {
    const bool sectors = h->use_sectors && h->sector_backoff == 0 && h->slots[(size_t)base].F.width <= MLM_SEC_MAX_WIDTH;
    if (h->sector_backoff > 0) --h->sector_backoff;
    for (int j = 0; j < n; ++j) {
        MlmSlot &S = h->slots[(size_t)(base + j)];
        S.seq = 0;
        S.F.seq = 0;
        S.F.pad2 = (int)(h->ex_frame_no++ & 0x3FFFFFFF); // (frame counter for the MLM_SEC_FAIL_EVERY test hook)
        S.sector = sectors;
    }
    Timed t(h, on_main ? h->stream : h->stream_as[base / h->lim.max_batch], "stage_a_batch");
    return sectors ? launch_stage_a_sector(h, base, n, on_main) : launch_stage_a_batch(h, base, n, on_main);
}
// The frames' counters are on the host: those with an overflowed sector table get their Stage A redone on the cell-table
// path (nothing that depends on the map has been enqueued for them yet).  Returns with their new counters on the host.
int explore_redo_overflows(mlm_handle *h, int base, int n) {
    const int set = base / h->lim.max_batch;
    bool any = false;
    for (int j = 0; j < n; ++j) {
        MlmSlot &S = h->slots[(size_t)(base + j)];
        if (!S.sector || !S.h_ctr->sector_overflow) continue;
        h->n_sector_fallbacks++;
        note_fallback(h, S.F.pad2);
        S.sector = false;
        const int rc = launch_stage_a_batch(h, base + j, 1);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(S.h_ctr, S.P.ctr, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream_as[set]));
        any = true;
    }
    if (any) {
        HIPCHK(h, hipStreamSynchronize(h->stream_as[set]));
        HIPCHK(h, hipGetLastError());
    }
    return MLM_OK;
}

} // namespace
