// mlm_types.h — device-visible parameter blocks of the map-update path (gfx950).
#pragma once
#include <stdint.h>

#define MLM_EMPTY_T 0xFFFFFFFFu      // hit_t: cell not hit this frame
#define MLM_HT_EMPTY 0xFFFFFFFFFFFFFFFFull
#define MLM_TIME_SLOTS 21            // insertion slots per point: centre + (+d,-d) for d = 1..10
#define MLM_DIFF_RANGE 10            // get_odds_table rows = 2*10+1 (map_awareness.cpp:36)

struct MlmCounters {
    // per-frame (zeroed at the start of every frame)
    unsigned int n_points;    // points with raw != 0
    unsigned int u_hit;       // unique hit cells
    unsigned int chain_next;  // sector path: next ranked cell to hand to a wave of k_chain_lanes
    unsigned int n_oor;       // "point out range"
    unsigned int n_ov;        // sector path: columns whose cell table overflowed (MlmDev::ov_list), redone with the large table
    unsigned int n_miss_list; // entries of ml_cell (record_awareness only)
    unsigned int n_contrib;   // contributions stored for multi-type cells (segments of `contrib`)
    unsigned int n_multi;     // hit cells that received more than one kind of contribution
    unsigned int n_ex_rays;   // explore mode: queued rays
    unsigned int n_ex_miss;   // explore mode: unique miss cells
    unsigned int pool_short;  // sector path: k_tile could not create a block the frame touches (the pool is full): the frame is applied
                              // once the host has grown the pool and k_alloc_retry has filled the records' slots
    unsigned int n_big;       // multi-kind cells with more than 1024 contributions
    unsigned int n_groups;    // contribution groups kept in the blocks' own slices of `bnodes` (folded by k_collect_hits)
    unsigned int sector_overflow; // sector path: 3 = a list sized by need was too short: the host enlarges the frame slots and runs the frame's
                                  // Stage A again; 2 = Stage A gave up, the frame is redone by the cell-table path; 1 = columns whose cell table
                                  // overflowed wait on ov_list for a large-table pass that was not scheduled: the host runs it when it drains
    unsigned int n_refs;      // sector path: (record, kind) references of the multi-kind cells
    unsigned int n_unassigned;// contribution groups that overflowed a block's LDS buffer (booked by k_assign_nodes)
    unsigned int bin_exact;   // sector path: waves of k_bin_sectors that evaluated the reference's own sequence (a lane too near a cell boundary)
    unsigned int apply_done;  // single-frame graph: workgroups of k_apply_single that have finished (the last one reports to the host)
    unsigned int big_next;    // sector path, in the counters of a batch's FIRST frame: next (frame, column) task of k_sector_big to hand to a workgroup
    unsigned int ray_cnt[8][32]; // [k][0] = rays walked (statistic), partial sums spread by blockIdx & 7, 128 B apart;
                                 // [k][1] = device-scope atomics the frame's kernels issued (sector path, counted by k_sector)
    unsigned int touch_cnt[8][32]; // [k][0] = first-touched hit cells queued in sub-list k
    unsigned int mvox_cnt[8][32];  // cell-table path: [k][0] = voxels touched by misses, sub-list k; sector path: [0][0] voxel records,
                                   // [1][0] entries of vr_hit, [3][0] unique miss cells of the frame
    unsigned int umiss_part[8][32];// [k][0] = partial count of unique miss cells
    unsigned int node_cnt[8][32];  // [k][0] = contribution nodes allocated in region k
    unsigned int mc_cnt[8][32];    // [k][0] = unique miss cells queued in sub-list k
};
// map-wide state shared by all frame slots
struct MlmGlobal {
    unsigned int n_blocks;    // allocated blocks
    unsigned int err;         // sticky error bits (1 = block pool / hash table full, 2 = a per-frame queue overflowed)
    int fail_frame;           // first frame of the current submission whose hit container would rehash (speculation
                              // miss); INT_MAX = none.  Stage B/C kernels of frames >= it do nothing.
    unsigned int pad;
};
#define MLM_CTR_SCALARS 19u // scalar members in front of the six [8][32] arrays (k_apply_single hands those and the arrays' [k][0], [k][1] to the host)
static_assert(sizeof(MlmCounters) == (MLM_CTR_SCALARS + 6u * 8u * 32u) * 4u, "MlmCounters layout");
#define MLM_CTR_APPLY_DONE 17u       // word index of MlmCounters::apply_done
static_assert(offsetof(MlmCounters, apply_done) == MLM_CTR_APPLY_DONE * 4u && offsetof(MlmCounters, ray_cnt) == MLM_CTR_SCALARS * 4u, "MlmCounters layout");
#define MLM_CTR_CLEARED 0xC1EA4ED0u  // host copy of apply_done after a single-frame graph: the device counters of the slot are clear again
// the t-th of the 114 words of MlmCounters that are ever written: the scalars, then [k][0] and [k][1] of the six spread arrays
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned int mlm_ctr_live_word(unsigned int t) {
    const unsigned int u = t - MLM_CTR_SCALARS;
    return t < MLM_CTR_SCALARS ? t : MLM_CTR_SCALARS + (u >> 4) * 256u + ((u >> 1) & 7u) * 32u + (u & 1u);
}
#define MLM_RAY_LISTS 8
#define MLM_MISS_COPIES 8 // private copies of the miss bit mask (chosen by blockIdx): rays of the whole image converge
                          // on the words next to the sensor, and same-line atomics serialise (~11 ns each)

// The hit contributions one wave makes to one awareness cell with one kind (`sub`): lanes in `mask`, work items
// i = i00 + lane (linear modes) or i00 + (lane>>3)*W + (lane&7) (dense 8x8 pixel tile).  `pos` is where the group's
// insertion times start inside the cell's segment of MlmDev::contrib.
struct MlmNode {
    uint32_t cell;
    uint32_t pos;
    uint32_t i00_sub; // i00 | sub << 27
    uint32_t pad;
    unsigned long long mask;
};
#define MLM_NIL 0xFFFFFFFFu
// Frontier mode, a synchronous call's lone frame: the bucket-first pass of both emulated containers (k_ex_order_min) rides in the
// frame's k_rank — it needs the hit and miss lists, which are complete by then, not the map.  on = 0: not this launch.
struct MlmExOrder {
    unsigned long long nb_hit, nb_miss; // bucket counts of the two containers (neither rehashes in this frame: thr_*)
    int tag;                            // the frame's tag in the bucket-first tables (mlm_bkt_entry)
    unsigned int thr_hit, thr_miss;     // the frame is left to the general path when it has more unique hit / miss cells than this
    int on;
};

// What one k_bin_points block contributes to one awareness cell (its groups merged in LDS).  k_book_cells books the
// pair on the cell with three atomics and records the returned count in `base`: the position of the block's
// contributions inside the cell's segment (the groups' MlmNode::pos are relative to it).
struct MlmPair {
    uint32_t cell;
    uint32_t tmin;      // earliest insertion time
    uint32_t kmask;     // kinds
    uint32_t cnt;       // contributions
    uint32_t start_min; // explore mode: first point whose hit centre is the cell (MLM_EMPTY_T: none)
    uint32_t base;
};

// Per-frame state of one awareness cell, 16 bytes so that the three atomics that book a group on a cell and the
// reads of k_collect_hits touch ONE cache line.
struct MlmCell {
    uint32_t t;    // first-touch time of a hit cell (min over contributions), MLM_EMPTY_T = not hit this frame
    uint32_t cnt;  // number of contributions
    uint32_t mask; // bit s set = a contribution of insertion slot s (0 centre, 2d-1 "+d", 2d "-d")
    uint32_t seg;  // start of the cell's segment in `contrib` (MLM_NIL for single-kind cells)
};

// One voxel a frame touches (sector path, written by k_tile): everything the kernel that applies the frame needs comes with
// one 16-byte load.  (map_local.cpp:147-207: the voxel's hits in container order, then its misses.)  The voxel's block key and
// cell id are not stored: k_tile has created the block, so the record carries the voxel's ADDRESS in the pool; where the pool was
// full (at = -1) k_alloc_retry recomputes block and cell from the tile and `vt` after the host has grown the pool.
struct MlmVoxRec {
    int32_t at;              // slot * cells + cell id of the voxel in the block pool; -1: its block could not be created yet
    uint32_t km_vt;          // unique miss cells of the frame that fall into the voxel (low 16 bits) | voxel in its tile (column << 10 | layer) << 16
    uint32_t inc_first;      // n_hit == 1: float bits of that hit's log-odds increment (its order is irrelevant);
                             // n_hit >= 2: first of its n_hit entries in MlmDev::vr_hit
    uint32_t n_hit;          // unique hit cells that fall into it
};
// One hit of a voxel with several: increment, iteration-order key (k_rank's speculative one; on a replayed frame hl_key[pos]
// holds the exact key instead)
struct MlmVoxHit {
    unsigned long long key;
    float inc;
    uint32_t pos;            // index in hl_*
};

#define MLM_LV_SLOTS 8
struct MlmDev {
    // ---- awareness map constants (map_awareness.cpp:19-82)
    double dRho, dPhi, dZ, z_border_min;
    int nRho, nPhi, nZ, zc;    // zc = map_center_z_idx
    int nRhoPhi, nCells;
    int RW;                    // 32-bit words per (phi,z) row of the miss bit mask = ceil(nRho/32)
    int nMissWords;
    int visibility;
    int logit_exact;           // 1: the host's log10f equals mlm_glibc_log10f (mlm_host.h) -> hit increments with the reference's float bits
    // ---- local map constants (map_local.cpp:46-139)
    double d_sub, d_glb, d_sub_half;
    double inv_dRho, inv_dPhi, inv_dZ, inv_d_sub, inv_d_glb; // 1/d (rounded once): see mlm_quot in mlm_device.h
    int n, cells;
    float lo_min, lo_max, lo_miss, lo_sh;
    // ---- camera (mlmap.h:85-92)
    float cx, cy, fx, fy;
    double inv_factor;
    double inv_fx, inv_fy;     // 1 / fx, 1 / fy (k_bin_sectors' fast path: mlm_bin_point_fast)
    double inv_d_max;          // max(1 / dRho, 1 / dPhi, 1 / dZ): scale of that path's margin
    // ---- tables
    const float *odds_table;   // [21*nRho] get_odds_table (map_awareness.cpp:36-46)
    const float *sigma3;       // [nRho] 3*sigma_in_dr(rho)  (float, map_awareness.cpp:149)
    const uint32_t *sec_const; // k_sector's constant LDS tables as it lays them out (mlm_sec_lds: odds .. sigma): a strength byte per entry of
    unsigned int sec_const_words; // the odds table, then sigma3 — one copy, one trip to memory
    const double *cos_phi;     // [nPhi] cos/sin of the cell-centre azimuth (map_awareness.cpp:59-61)
    const double *sin_phi;
    // ---- per-frame awareness scratch
    MlmCell *cs;               // [nCells] per-cell frame state (see MlmCell)
    MlmNode *bnodes;           // [nb_cap][node_lds] groups of k_bin_points block b in slice b (count: blk_stats[4b+2]);
                               // MlmNode::pad = index of the group's MlmPair in `pairs`
    MlmPair *pairs;            // [nb_cap][agg_lds] (block, cell) pairs of block b in slice b (count: blk_stats[4b+3])
    unsigned int nb_cap;       // most k_bin_points blocks per frame
    MlmNode *nodes;            // [MLM_RAY_LISTS][node_cap] overflow: groups that did not fit a block's LDS buffer
    unsigned int node_cap;     // per region
    uint32_t *contrib;         // [contrib_cap] insertion times of the contributions of multi-kind cells, by cell
    unsigned int *blk_stats;   // [4*nb_cap] per k_bin_points block: points fed, points out of range, groups, pairs
    uint32_t *mt_list;         // [nCells] indices into the hit list of the multi-type cells
    uint4 *mt_rec;             // [nCells] per multi-type cell: {hit-list index, segment base, contribution count, first-touch time}
    uint32_t *mt_big;          // [nCells] those with more than 1024 contributions
    uint32_t *touched;         // [MLM_RAY_LISTS][touch_cap] hit cells in first-touch order of the GPU (arbitrary)
    unsigned int touch_cap;    // per sub-list
    uint8_t *subs;             // [contrib_cap] per multi-kind cell: contribution kinds in insertion-time order
                               // (segments start 16-byte aligned; same offsets as `contrib`)
    unsigned int contrib_cap;
    uint32_t *miss_bits;       // [MLM_MISS_COPIES][nMissWords] free cells, row-major (z,phi) rows of RW words, bit = rho;
                               // the mask of the frame is the OR of the copies (k_prepare_voxels)
    // ---- unique-hit list (capacity nCells)
    uint32_t *hl_cell;         // linear awareness cell idx
    uint32_t *hl_t;            // first-touch time
    float *hl_odd;             // noisy-OR odd
    float *hl_inc;             // logit(odd), the log-odds increment
    uint32_t *hl_base;         // segment start in `contrib`
    uint32_t *hl_cnt;          // segment length
    uint32_t *hl_vt;           // virtual insertion time (== hl_t when no rehash happened this frame)
    uint32_t *hl_arr;          // arrival index (rank of hl_t), only valid on rehash frames
    uint64_t *hl_key;          // iteration-order key: (bucket_first << 32) | vt ; larger = earlier in iteration
    int *hl_next;              // per-voxel pending list link (sector path: only hits beyond the voxel's direct slots, lv_hits)
    int *hl_vox;               // voxel address (slot*cells + cell id)
    uint32_t *bkt_first;       // [max buckets] min vt per hash bucket (exact path on rehash frames)
    unsigned long long *bkt64; // [max buckets] (~seq << 32 | min vt): speculative path, never cleared
    uint32_t *hl_bkt;          // bucket index of each unique hit (speculative path)
    unsigned long long *hl_bkey; // packed block key of the cell centre's world voxel
    uint32_t *hl_cid;          // cell id inside that block
    int *hl_slot;              // pool slot of that block if it already existed when Stage A looked, else -1
    int *mc_slot;              // same for the miss cells
    int *mc_vox;               // voxel address if the miss cell is its voxel's first miss of the frame, else -1
    unsigned long long *mc_bkey; // [MLM_RAY_LISTS][mc_cap] unique miss cells: packed block key ...
    uint32_t *mc_cid;          //                              ... and cell id
    unsigned int mc_cap;       // per sub-list
    uint32_t *ml_cell;         // unique-miss list (only with record_awareness)
    int record_awareness;
    // ---- hashed block table + pool
    unsigned long long *ht_keys; // [ht_mask+1] packed block key or MLM_HT_EMPTY
    int *ht_slot;              // [ht_mask+1] pool slot, -1 until published
    uint32_t ht_mask;
    int max_blocks;
    int *block_keys;           // [max_blocks*3]
    float *log_odds;           // [max_blocks*cells]
    uint8_t *occ;              // [max_blocks*cells] 'u','f','o'
    uint8_t *infl;             // [max_blocks*cells]
    int *vox_head;             // [2][max_blocks*cells] head of this frame's pending hit list, -1 = none; the copy is chosen
                               // by the frame's sequence number & 1, so that k_apply of frame f and k_voxelize of
                               // frame f+1 can run in one launch
    uint32_t *vox_miss;        // [2][max_blocks*cells] miss count of this frame
    size_t vox_stride;         // max_blocks*cells
    size_t bkt_stride;         // elements of one copy of bkt64 (also [2], same rule)
    int *miss_vox;             // [MLM_RAY_LISTS][mvox_cap] voxels touched by misses this frame
    unsigned int mvox_cap;
    // ---- exploration-frontier mode (use_exploration_frontiers: true) — map_local.cpp:7-33,208-232
    int explore;
    // LDS sizing of k_bin_points (host-chosen from the noise spread of the configuration)
    unsigned int node_lds, agg_lds, agg_shift, bin_lds_bytes;
    unsigned int bin_block;    // threads per k_bin_points block: 256 (32x8 pixel strip) or 1024 (32x32 tile)
    uint8_t *frnt;             // [max_blocks*cells] 1 = the cell is in its block's frontier set
    uint8_t *blk_collapsed;    // [max_blocks] 1 = block was "released" (vectors resized to 1: frozen, element 0 answers)
    uint8_t *blk_observed;     // [max_blocks] observed_subboxes of the current frame
    unsigned long long *vox_tau; // [max_blocks*cells] iteration-order key of the voxel's first miss this frame
    uint32_t *start_t;         // [nCells] first point whose hit centre is this cell (its ray's insertion time base)
    uint32_t *miss_t;          // [nCells] first insertion time of a miss cell: point*256 + ray step
    int32_t *ex_rays;          // [max_points][4] queued rays: rho, phi, z, point (-1: take start_t of the cell)
    uint32_t *ex_cell;         // unique miss cells of the frame (capacity nCells) ...
    uint32_t *ex_t;            // ... first insertion time
    uint32_t *ex_vt;           // ... virtual insertion time (rehash replay)
    uint32_t *ex_arr;          // ... arrival rank
    unsigned long long *ex_key;// ... iteration-order key (larger = earlier)
    int *ex_vox;               // ... voxel address (see mlm_ex_miss_tau_body)
    unsigned long long *ex_bkey; // ... packed block key of its world voxel
    uint32_t *ex_cid;          // ... cell id inside that block
    uint32_t *bktm_first;      // [max buckets of the miss container] min vt per bucket
    // ---- sector path (mlm_kernels_sector.h)
    unsigned int *col_cnt;     // [nPhi] chunk descriptors handed to each column this frame (reset by k_sector)
    uint32_t *col_chunks;      // [nPhi][chunk_cap][2] {first record in `bnodes`, record count} per (bin block, column) run
    unsigned int chunk_cap;
    unsigned int sec_tab, sec_lds_bytes; // LDS sizing of k_sector: cell table entries (power of two)
    unsigned int sec_tab_big, sec_big_lds_bytes; // ... of k_sector_big (0 entries: no second pass)
    uint32_t *ov_list;         // [nPhi] columns whose cell table overflowed in k_sector (count: MlmCounters::n_ov)
    unsigned int sec_fail_every;         // test hook (MLM_SEC_FAIL_EVERY=k): every k-th frame is made to fall back
    uint32_t *refs;            // [refs_cap] one 4-byte reference (mlm_ref_pack: row byte, kind, position relative to the cell's first pixel) per
                               // non-empty row of the lane mask of every contribution group of a multi-kind cell
    unsigned int refs_cap;
    uint32_t *mt_ref;          // [nCells][2] per multi-kind cell: {start in `refs`, count of its references}
    // Frame-local voxel grid: the voxels the awareness cylinder can reach, addressed relative to MlmFrame::lv_o and cut into
    // TILES of 2^tile_sh x 2^tile_sh voxels in x,y over the grid's whole height.  Which voxel — and so which tile — a cell
    // of an azimuth column falls into is geometry (no map needed), and along a column it depends on rho only: k_sector hands
    // every tile the runs of the column's miss cells and hits that fall into it (a descriptor per (column, tile): the
    // column's cells leave in one coalesced stream), k_tile groups a tile's cells by voxel in LDS and writes ONE record per
    // touched voxel, and the kernel that applies the frame reads those records.  No per-cell global atomic anywhere.
    int lv_nx, lv_ny, lv_nz;
    int tile_sh, n_tx, n_tiles;        // tile edge 2^tile_sh voxels; tiles per grid row; tiles per frame
    uint16_t *mc_list;         // [nPhi][nRho * nZ] the frame's unique miss cells as voxel-in-tile indices, every column in a stretch of its own;
                               // vt = ((y & m) << tile_sh | (x & m)) * lv_nz + z; a column's cells are contiguous, ordered by tile
    unsigned int mc_list_cap;
    uint16_t *hl_vt16;         // [nCells] vt of every unique hit (a column's hits are contiguous in hl_*, ordered by tile)
    unsigned int tile_combos;  // most blocks a tile of the frame-local grid overlaps (k_tile keeps their pool slots in LDS), multiple of 4
    uint32_t *tile_cols;       // [n_tiles][tile_words] column mask of each tile: bit phi = column phi left a descriptor this frame (set by
                               // k_sector with a fire-and-forget atomic, read and cleared by k_tile)
    unsigned int tile_words;   // ceil(nPhi / 32)
    uint32_t *tile_desc;       // [n_tiles][nPhi][4] slot [tile][phi] = {first miss cell in mc_list, count, first hit in hl_*, count} of
                               // column phi's run through the tile (a column's ray crosses a tile once)
    MlmVoxRec *vr_rec;         // [rec_cap] one record per voxel the frame touches (count: MlmCounters::mvox_cnt[0][0])
    unsigned int rec_cap;
    // capacities of the frame's lists that are sized by NEED (mlm_handle::need_sized; otherwise the worst case, nCells): a Stage A
    // that would overrun one of them gives the frame up with sector_overflow 3, the host enlarges the slots and runs it again
    unsigned int hl_cap;       // unique hits: hl_* (and hl_vt16)
    unsigned int mt_cap;       // cells whose contributions are ranked: mt_list, mt_rec, mt_ref
    unsigned int vh_cap;       // entries of vr_hit
    MlmVoxHit *vr_hit;         // [nCells] the hits of voxels with two or more, a voxel's hits contiguous (count: mvox_cnt[1][0])
    uint32_t *tile_dir;        // [n_tiles][4] {first record of the tile in vr_rec, count, frame sequence number, -}: written by k_tile,
                               // valid for the frame whose sequence number it carries (never cleared)
    unsigned long long *sbkt;  // [sbkt_cap] this slot's bucket-first table of the emulated hit container: (~seq << 32 | time),
    unsigned int sbkt_cap;     // never cleared (a newer frame's entries win the min)
    MlmCounters *ctr;          // this slot's per-frame counters
    MlmGlobal *g;
    // frontier mode, a synchronous frame whose map-dependent kernels are enqueued BEFORE the host has seen the frame's counts
    // (mlm_explore_host.h, explore_stage_bc_spec): they take the counts from `ctr` and do nothing unless the frame fits both emulated
    // containers without a rehash and its Stage A stayed on the sector path (mlm_ex_spec_skip); the host checks the same afterwards
    int spec_on;
    unsigned int spec_hit_thr, spec_miss_thr;
};

#define MLM_FRAME_EXACT_KEYS 1 // hl_key holds the exact iteration-order keys (order_hits_exact): no speculation check, keys from hl_key
#define MLM_FRAME_SKIP 2       // the frame has been applied by other means (redone on the cell-table path)
struct MlmFrame {
    // T_ls = T_wa^-1 * T_wb * T_bs (map_awareness.cpp:184-186), t_wa = t_wb
    double q_ls[4]; // w,x,y,z
    double t_ls[3];
    double t_wa[3];
    const uint16_t *img;   // device
    const int32_t *pix;    // device or null
    const int32_t *raw;    // device or null: the depth values of the listed pixels by list position (else read from img)
    const double *pts;     // device or null
    int width, height, row_stride;
    int n;                 // work items: pixels (dense), list length (indexed) or points
    int seq;               // sequence number of the frame (speculation bookkeeping)
    unsigned int rehash_thr; // the emulated hit container takes this many elements without a rehash (speculative Stage B)
    int lv_o[3];           // origin of the frame-local voxel grid (MlmDev::lv_state) in voxel coordinates
    int pad2;                  // frontier mode: running frame number (test hook MLM_SEC_FAIL_EVERY; seq stays 0 there)
    int flags;                 // sector path, k_apply_tiles: MLM_FRAME_EXACT_KEYS | MLM_FRAME_SKIP
    int pad3;
    // k_bin_sectors' cheap evaluation (mlm_bin_point_fast): the linear map of QuaternionBase::_transformVector for q_ls as a matrix
    // (I + 2 w [q]x + 2 [q]x^2, row major; exact for any q, unit or not), its gain 1 + 2 |w| |q_v|_1 + 2 |q_v|_1^2 (bound of the
    // intermediates per unit |v|), and |t_ls|_1 + 1
    double m_ls[9];
    double m_gain, t_l1;
};
