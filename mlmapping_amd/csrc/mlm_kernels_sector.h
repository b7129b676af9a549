// mlm_kernels_sector.h — Stage A by azimuth sector (the default path; mlm_kernels.h keeps the cell-table path that a
// frame falls back to when a sector's LDS tables overflow).
//
// Everything awareness_map_cylindrical::input_pc_pose does to one azimuth column phi stays inside that column: the noise
// spread of a hit moves along rho (and z) at the point's phi (map_awareness.cpp:149-168) and its ray runs radially inwards
// at that phi (map_awareness.cpp:243-274).  So the frame is cut into nPhi independent sectors:
//
//   k_bin_sectors   point -> (rho,phi,z), by a cheap FP64 evaluation with certified margins (mlm_bin_point_fast; the reference's own
//                   sequence for a wave with a lane too near a cell boundary); the lanes of a wave that share a centre cell become
//                   ONE 16-byte record (cell, tile origin, lane mask); a block buckets its <= 256 records by column in LDS and
//                   hands each (block, column) run to the column with one returning atomic (a chunk descriptor).  Nothing else
//                   leaves the block.
//   k_sector        one workgroup per column: per-cell hit bookkeeping (first-touch time, kinds, counts) in an LDS hash
//                   table, the column's miss bit mask in LDS (rays walked by an integer DDA), the unique-hit list
//                   (cells that received ONE kind of contribution, or enough strong ones, get their odd right away), and
//                   for every other cell one 4-byte reference per non-empty row of a record's lane mask.  Which world voxel a cell falls into is
//                   geometry that separates by axis inside a column (tables per rho and per z); the frame-local voxel grid
//                   is cut into tiles, a column crosses a tile once, so the column's hits and miss cells — ordered by rho —
//                   leave as one coalesced stream of voxel-in-tile indices plus ONE descriptor per tile it crosses, stored in the
//                   (tile, column) slot of the tile's table and announced by the column's bit in the tile's column mask.
//                   Global atomics: four list reservations per workgroup (the only ones it waits for), fire and forget: one
//                   bucket-min per hit, one mask bit per tile run.
//   k_rank          iteration-order keys of the frame's hits; one wave per multi-kind cell whose float chain depends on the
//                   order: rank its contributions by pixel (bitmap ranking fed with the records' 8x8 lane masks) and
//                   store the kinds in that order.
//   k_chain_lanes   replays the float noisy-OR chains, one cell per lane, lanes drawing cells dynamically.
//   k_tile          workgroups take the tiles of the frame-local voxel grid in turns (a column mask says which columns left a run):
//                   count the tile's miss cells and hits per voxel in LDS, create the blocks the frame touches there, and write
//                   ONE 16-byte record per touched voxel (address in the block pool, miss count | voxel-in-tile index, the
//                   increment of a single hit, hit count); the hits of voxels with several go next to each other.
//   k_apply_tiles   the part that needs the map, ONE launch per batch: the frame-local grids are aligned to tile boundaries, a
//                   workgroup owns a WORLD tile, keeps its voxels in LDS and walks the batch's frames in order — per record the
//                   voxel's hits in the reference's iteration order, then its misses.  k_apply_single: the same for one frame
//                   (a flat loop over the frame's records; the synchronous single-frame call and replays).
//
// Frontier mode (use_exploration_frontiers) uses k_bin_sectors, k_sector<true>, k_rank and k_chain_lanes and continues with its
// own map-dependent part (mlm_kernels_explore.h): there the miss container's iteration order matters as well, so the
// column keeps the first insertion time of every miss cell in LDS instead of a bit.
//
// Compared with the cell-table path this removes every per-cell, per-miss-word and per-voxel device-scope atomic (they are
// executed at the memory side, ~34 G/s for the whole chip, and their latency is what a column's workgroup used to wait
// for), the 8 global copies of the miss mask and their scan, and the scattered insertion-time stores (0.9 M per VGA frame).
#pragma once
#include "mlm_kernels.h"

#define MLM_SEC_THREADS 512 // threads of a column's workgroup (k_sector also exists with 256: MlmDev::sec_tab <= 1024)
#define MLM_SEC_WAVES (MLM_SEC_THREADS / 64)
#define MLM_SEC_COLS 256    // distinct columns one k_bin_sectors block can feed in the list modes: one per record at most (a pixel list
                            // scatters over the image); a dense 32x8 pixel strip spans a handful: 64 entries bucketed by one wave
#define MLM_SEC_CHUNKS 256  // chunk descriptors staged per pass of k_sector (a column of a VGA frame has ~50)
#define MLM_SEC_OUTER 0x80000000u // top bit of a column record's first word: the record only starts a ray (point outside the map)
#define MLM_SEC_RANK_WORDS (2 * MLM_BMP_ROWS) // u64 words of a wave's ranking bitmap (128 columns x 128 rows)

// one hit cell of the column while k_sector works on it (16 bytes: a 2 048-entry table is 32 KB of LDS)
struct MlmSecCell {
    uint32_t key;   // low 16 bits: z * nRho + rho (nZ * nRho < 65 536), MLM_NIL = empty; high 16 bits, set once the column's lists are
                    // built (cells that need their order): where the cell's references start, in units of MLM_SEC_REF_ALIGN references
                    // from the column's first
    uint32_t tmin;  // first-touch time (min over contributions)
    uint32_t kg;    // kinds (bits 0..20) | (record, kind) references of the cell << 21: counted up while the contributions are booked,
                    // counted down while the references are written (the count hands every writer its own stretch of the cell's segment)
    uint32_t cnt;   // contributions
};
#define MLM_SEC_REF_ALIGN 4u // a cell's references start at a multiple of this many (16 bytes) from the column's first
#define MLM_SEC_KEEP_MORE 15u // count of a kept record whose targets did not fit the kept registers (see k_sector)
#define MLM_SEC_KEY_MASK 0xFFFFu

#define MLM_SEC_KIND_BITS 21 // MlmSecCell::kg: 2 * MLM_DIFF_RANGE + 1 kinds below the reference count
#define MLM_SEC_KIND_MASK ((1u << MLM_SEC_KIND_BITS) - 1u)
#ifndef MLM_SEC_CNT_BITS
#define MLM_SEC_CNT_BITS 21 // MlmSecCell::cnt: contributions in the low bits (mlm_limits.max_points < 2^21 on this path: a full-HD depth image),
                            // the sum of their strengths above them (mod 2^11: a wrap only makes a cell look weaker than it is)
#endif
#define MLM_SEC_CNT_MASK ((1u << MLM_SEC_CNT_BITS) - 1u)
// Does the float noisy-OR chain of the cell (update_odds_hashmap, map_awareness.h:147-154: p <- 1 - (1 - p)(1 - a), each
// operation rounded) depend on the order of its contributions?  Not with one kind only.  And not once the contributions
// are strong enough to end at exactly 1.0f in ANY order.  Call a contribution with a >= 0.5 strong, of strength
// s = 1 (a < 0.75), 2 (a < 0.875) or 3: b = 1 - a is exact and <= 2^-s.  The first strong step of any order puts p into
// [0.5, 1], where 1 - p is exact: an integer k <= 2^24 in units of 2^-24.  From then on p never decreases; a strong step maps
// k to at most k (1 + 2^-24) / 2^s + 0.5 (the product m = k 2^-24 b is rounded to float, then 1 - m to the grid), any other
// step to at most k.  Unrolled over the strong steps before the LAST one of the order (fewer than 2^21: (1 + 2^-24)^(2^21) < 1.14;
// strengths summing to at least S - 3 with S the sum of all): k <= 1.14 (2^24 / 2^(S - 3) + 1) < 2, i.e. k <= 1 once S >= 28; the last strong step
// turns k <= 1 into m <= 2^-25 (exact), and 1 - m rounds to 1.0f (ties to even), which is absorbing.
// Such cells are finished without ranking their contributions (S is summed mod 2048 next to the count: a wrap only
// makes a cell look weaker than it is).
#define MLM_SEC_STRONG_ENOUGH 28u
__host__ __device__ __forceinline__ uint32_t mlm_sec_strength(float a) { return a >= 0.875f ? 3u : (a >= 0.75f ? 2u : (a >= 0.5f ? 1u : 0u)); }
// A cell with exactly TWO contributions needs no order either: p = 1 - (1 - a)(1 - b) whichever comes first (the first step sets
// p to the first value, the float product commutes) — the far cells of a frame, where a pixel's spread meets a neighbour's centre.
__device__ __forceinline__ bool mlm_sec_needs_order(const MlmSecCell &c) {
#ifdef MLM_EXP_KEEP_ONE_IN // (throughput experiment, results are wrong: only one ordered cell in N keeps its order — what would the pipeline gain if the rest were free?)
    if ((((c.key & MLM_SEC_KEY_MASK) * 2654435761u) >> 16) % MLM_EXP_KEEP_ONE_IN != 0u) return false;
#endif
    return __popc(c.kg & MLM_SEC_KIND_MASK) > 1 && (c.cnt >> MLM_SEC_CNT_BITS) < MLM_SEC_STRONG_ENOUGH && (c.cnt & MLM_SEC_CNT_MASK) > 2u;
}

// rows (bytes) of an 8x8 lane mask that hold a lane: a multi-kind cell gets one reference per such row of every contribution group
__device__ __forceinline__ uint32_t mlm_mask_rows(unsigned long long m) {
    m |= m >> 4;
    m |= m >> 2;
    m |= m >> 1;
    return (uint32_t)__popcll(m & 0x0101010101010101ull);
}

// A reference of a multi-kind cell = one non-empty row of the 8x8 lane mask of one contribution group, 4 bytes:
//   bits 0-7 the row's byte of the mask, 8-12 the kind, 13-31 where the row's first lane lies relative to the cell's FIRST pixel
//   (its earliest contribution, MlmSecCell::tmin: no contribution lies in a row above it):
//     dense images   (rows below the first pixel's) << 8 | 128 + (tile column - the first pixel's tile column)   11 + 8 bits
//     lists          (64-item rows below the first item's) << 3 | mask row      16 + 3 bits (2^22 items)
#define MLM_REF_DY_DENSE 2047u
#define MLM_REF_DY_LIST 65535u
// (xrel: the row's tile column relative to the tile column of the cell's first pixel, + MLM_REF_XREL0: an image may be any width, a cell's
// contributions lie within 1 016 pixels of its first one's column — else the frame gives the sector path up)
#define MLM_REF_XREL0 128u
__device__ __forceinline__ uint32_t mlm_ref_pack(uint32_t bits, uint32_t kind, bool dense, uint32_t dy0, uint32_t row, uint32_t xrel) {
    const uint32_t pos = dense ? ((dy0 + row) << 8) | xrel : (dy0 << 3) | row;
    return bits | (kind << 8) | (pos << 13);
}
// A hit record's tile origin (MlmSecRec, second word).  Dense images: (row >> 3) << 13 | column >> 3 of the wave's 8x8 pixel tile
// (both multiples of eight: images up to 65 528 pixels wide); lists: the wave's run of 64 items << 11.
#define MLM_REC_XT_BITS 13
#define MLM_REC_XT_MASK ((1u << MLM_REC_XT_BITS) - 1u)
#define MLM_SEC_MAX_WIDTH ((int)(MLM_REC_XT_MASK << 3)) // widest dense image of the sector path
// -> row byte, kind, rows below the cell's first pixel, column of the row's first lane
__device__ __forceinline__ void mlm_ref_unpack(uint32_t ref, bool dense, uint32_t &bits, uint32_t &kind, uint32_t &dy, uint32_t &x) {
    bits = ref & 0xFFu;
    kind = (ref >> 8) & 31u;
    const uint32_t pos = ref >> 13;
    dy = dense ? pos >> 8 : pos >> 3;
    x = dense ? (pos & 255u) << 3 : (pos & 7u) << 3;
}

__device__ __forceinline__ void mlm_sector_fail(const MlmDev &P, const MlmFrame &F, bool capacity = false) {
    // (3: a list sized by need was too short — the slots are enlarged and the frame's Stage A runs again; 2: the frame takes the
    // cell-table path; 1: see the column kernel's full-table branch)
    g_atomic_max(&mlm_gp(P.ctr)->sector_overflow, capacity ? 3u : 2u);
    // (frontier mode does not speculate: its host reads the flag before it enqueues what depends on the map)
    if (!P.explore) g_atomic_min(&mlm_gp(P.g)->fail_frame, F.seq);
}

// work item of thread `threadIdx.x` in strip `strip` of the frame (mlm_tile_item with the strip index instead of blockIdx.x):
// dense images are cut into strips of 32 x 8 pixels (a wave = an 8x8 tile), lists into runs of 256 items
template <int MODE> __device__ __forceinline__ MlmTile mlm_strip_item(const MlmFrame &F, unsigned int strip) {
    MlmTile t;
    if (MODE == 0) {
        const int tiles_x = (F.width + 31) >> 5;
        const int by = (int)strip / tiles_x;
        const int bx = (int)strip - by * tiles_x;
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        const int px = bx * 32 + (w & 3) * 8 + (l & 7);
        const int py = by * 8 + (l >> 3);
        t.valid = px < F.width && py < F.height;
        t.i = py * F.width + px;
    } else {
        t.i = (int)(strip * 256u + threadIdx.x);
        t.valid = t.i < F.n;
    }
    return t;
}

// S strips per workgroup, worked on TOGETHER (every phase runs for all S before the next starts).  The launches use S = 1: two
// and four strips in flight measured 1-4 % slower in the pipeline (profiles/r4b_*) — the kernel is bound by vector-instruction issue,
// not by the waits of a strip's chain.  A record never touches LDS: the lane that leads a group of pixels keeps it in registers until
// its place in the strip's slice is known (a lane leads at most one record) — what is staged per strip is the column table (phi,
// count, offset: 768 bytes for a dense strip).
template <int MODE, int S>
__device__ __forceinline__ void mlm_bin_sectors_body(const MlmDev &P, const MlmFrame &F, unsigned int n_strips, bool pre = false, int pre_pix = 0,
                                                     int pre_raw = 0) { // (pre: this thread's list entry has been fetched already, k_bin_sectors_hostf)
    constexpr uint32_t COLS = MODE == 0 ? 64u : (uint32_t)MLM_SEC_COLS; // entries of a strip's column table
    static_assert(MODE == 0 || S == 1, "list modes: one strip per workgroup (one column entry per thread)");
    __shared__ uint32_t s_col_phi[S][COLS], s_col_cnt[S][COLS], s_col_off[S][COLS];
    __shared__ unsigned int s_cnt[S][4], s_wsum[4];
    __shared__ unsigned int s_over;
    for (uint32_t e = threadIdx.x; e < S * COLS; e += 256u) {
        (&s_col_phi[0][0])[e] = MLM_NIL;
        (&s_col_cnt[0][0])[e] = 0;
    }
    if (threadIdx.x == 0) s_over = 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned int strip0 = blockIdx.x * (unsigned int)S;
    // ---- the strips' inputs (all loads in flight together)
    int item[S];
    bool have[S];
    double xs[S], ys[S], zs[S];
    uint16_t raw[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        MlmTile T = mlm_strip_item<MODE>(F, strip0 + (unsigned int)j);
        if (strip0 + (unsigned int)j >= n_strips) T.valid = false;
        item[j] = T.i;
        have[j] = T.valid;
        xs[j] = ys[j] = zs[j] = 0;
        raw[j] = 0;
        if (have[j]) {
            if (MODE == 2) {
                xs[j] = mlm_gp(F.pts)[3 * (size_t)T.i + 0];
                ys[j] = mlm_gp(F.pts)[3 * (size_t)T.i + 1];
                zs[j] = mlm_gp(F.pts)[3 * (size_t)T.i + 2];
            } else {
                const int pix = (MODE == 1) ? (pre ? pre_pix : mlm_gp(F.pix)[T.i]) : T.i;
                const int v = pix / F.width;
                const int u = pix - v * F.width;
                raw[j] = (MODE == 1 && pre) ? (uint16_t)pre_raw : (MODE == 1 && F.raw) ? (uint16_t)mlm_gp(F.raw)[T.i] : mlm_gp(F.img)[(size_t)v * F.row_stride + u];
                // mlmap.cpp:329,344-346: (size_t u - float cx_) is a float subtraction, the rest is double (the depth factor below)
                xs[j] = (double)((float)u - P.cx);
                ys[j] = (double)((float)v - P.cy);
            }
        }
    }
    __syncthreads(); // (the column tables are clear)
    // ---- bins, records (in registers), column buckets
    uint32_t rec_cell[S], rec_pos[S], rec_place[S]; // rec_place: column entry | position in its run << 16, MLM_NIL: no record
    unsigned long long rec_mask[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        if (MODE != 2 && have[j]) {
            if (raw[j] == 0) { // mlmap.cpp:338-341
                have[j] = false;
            } else {
                const double depth = raw[j] * P.inv_factor;
                xs[j] = xs[j] * depth; // (the division by the focal length follows below: x = (u - cx) * depth / fx, mlmap.cpp:344-346)
                ys[j] = ys[j] * depth;
                zs[j] = depth;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int rho = 0, phi = 0, zi = 0, c0 = -1;
        bool can_do_cast = false, inside = false, sure = true;
        // the bins by the cheap evaluation with certified margins (mlm_bin_point_fast) ...
#ifndef MLM_BIN_EXACT // (experiment builds: make alt ALT_FLAGS=-DMLM_BIN_EXACT evaluates the reference's sequence for every wave)
        if (have[j]) {
            const double vx = MODE == 2 ? xs[j] : xs[j] * P.inv_fx, vy = MODE == 2 ? ys[j] : ys[j] * P.inv_fy;
            inside = mlm_bin_point_fast(P, F, vx, vy, zs[j], (fabs(vx) + fabs(vy)) + fabs(zs[j]), rho, phi, zi, can_do_cast, sure);
        }
#else
        sure = !have[j];
#endif
        if (!__all(sure)) { // ... and, for a wave with a lane too close to a cell boundary to be sure, by the reference's own sequence
            if (lane == 0) g_atomic_add(&mlm_gp(P.ctr)->bin_exact, 1u); // (statistic: mlm_frame_stats.n_bin_exact_waves)
            if (have[j]) {
                // p_l = T_ls * p_s (map_awareness.cpp:222; se3.cpp:91-95)
                double x, y, z;
                mlm_quat_rot(F.q_ls, MODE == 2 ? xs[j] : xs[j] / P.fx, MODE == 2 ? ys[j] : ys[j] / P.fy, zs[j], x, y, z);
                x = x + F.t_ls[0];
                y = y + F.t_ls[1];
                z = z + F.t_ls[2];
                inside = mlm_bin_point(P, x, y, z, rho, phi, zi, can_do_cast);
            }
        }
        if (inside) c0 = zi * P.nRhoPhi + phi * P.nRho + rho;
        const uint32_t i00 = (uint32_t)mlm_readlane(item[j], 0); // work item of lane 0 of this wave (see MlmNode)
        // lanes of one centre cell -> one record, held by their lowest lane (= earliest insertion time)
        unsigned long long my_mask = 0;
        bool leader = false;
        mlm_wave_groups(c0, inside, [&](int, unsigned long long m) {
            leader = true;
            my_mask = m;
        });
        // points outside the map that can still cast (map_awareness.cpp:241,249-265): identical starts merged per wave
        const bool outer = have[j] && !inside && can_do_cast && P.visibility;
        bool ray_leader = false;
        {
            unsigned long long todo = __ballot(outer);
            while (todo) {
                const int ld = __ffsll((long long)todo) - 1;
                const int kr = mlm_readlane(rho, ld), kp = mlm_readlane(phi, ld), kz = mlm_readlane(zi, ld);
                const unsigned long long m = __ballot(outer && rho == kr && phi == kp && zi == kz);
                if (lane == ld) {
                    ray_leader = true;
                    my_mask = m;
                }
                todo &= ~m;
            }
        }
        rec_place[j] = MLM_NIL;
        rec_cell[j] = leader ? (uint32_t)zi << 16 | (uint32_t)rho : (MLM_SEC_OUTER | (uint32_t)rho); // (hit records: z and rho of the centre cell, nRho * nZ < 65 536)
        // hit records carry their tile's origin (MLM_REC_XT_BITS; list modes: 64 items = one row) for k_rank
        uint32_t yx = i00 >> 6 << 11;
        if (MODE == 0) { // (from the strip's place in the image: nothing is divided per wave)
            const uint32_t tiles_x = ((uint32_t)F.width + 31u) >> 5, strip = strip0 + (uint32_t)j;
            const uint32_t by = strip / tiles_x, bx = strip - by * tiles_x;
            yx = (by << MLM_REC_XT_BITS) | (bx * 4u + (uint32_t)wid);
        }
        // a record is 16 bytes (MlmSecRec).  Hit: centre cell z << 16 | rho, tile origin yx, lane mask (the wave's first work item
        // follows from yx: y0 * width + x0, lists (yx >> 11) << 6).  Ray of a point outside the map: MLM_SEC_OUTER | rho, z, and in the
        // mask's place its first point (the leader's own work item: frontier mode orders miss cells by it)
        rec_pos[j] = leader ? yx : (uint32_t)zi;
        rec_mask[j] = leader ? my_mask : (unsigned long long)(uint32_t)item[j];
        if (leader || ray_leader) { // bucket the record by column
            const uint32_t ph = (uint32_t)phi;
            uint32_t e = (ph * 2654435761u) >> (MODE == 0 ? 26 : 24);
            bool placed = false;
            for (uint32_t probe = 0; probe < COLS; ++probe) {
                const uint32_t prev = atomicCAS(&s_col_phi[j][e], MLM_NIL, ph);
                if (prev == MLM_NIL || prev == ph) {
                    placed = true;
                    break;
                }
                e = (e + 1) & (COLS - 1);
            }
            if (placed) rec_place[j] = e | (atomicAdd(&s_col_cnt[j][e], 1u) << 16);
            else s_over = 1; // (dense tiles only: more than 64 columns in one 32x8 pixel strip)
        }
        const unsigned int n_pts = (unsigned int)__popcll(__ballot(have[j]));
        const unsigned int n_oor = (unsigned int)__popcll(__ballot(have[j] && !(can_do_cast && P.visibility)));
        if (lane == 0) s_cnt[j][wid] = n_pts | (n_oor << 10);
    }
    __syncthreads();
    // ---- offsets of the columns' runs inside the strips' slices.  Dense strips: wave j does strip j (the other waves are done
    //      once their records are stored and do not wait for the atomic); list modes: one table entry per thread.
    uint32_t run_cnt = 0, run_off = 0, run_tot = 0;
    if (MODE == 0) {
        if (wid < S) {
            run_cnt = s_col_cnt[wid][lane];
            const uint32_t incl = mlm_wave_incl_scan(run_cnt);
            run_off = incl - run_cnt;
            run_tot = mlm_readlane(incl, 63);
            s_col_off[wid][lane] = run_off;
        }
    } else {
        run_cnt = s_col_cnt[0][threadIdx.x];
        run_off = mlm_wave_incl_scan(run_cnt);
        if (lane == 63) s_wsum[wid] = run_off;
        __syncthreads();
        for (int w = 0; w < wid; ++w) run_off += s_wsum[w];
        run_tot = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        run_off -= run_cnt;
        s_col_off[0][threadIdx.x] = run_off;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S; ++j) {
        if (rec_place[j] != MLM_NIL) {
            *((MLM_GLOBAL mlm_u32x4 *)mlm_gp(P.bnodes) + ((size_t)(strip0 + (unsigned int)j) * 256u + s_col_off[j][rec_place[j] & 0xFFFFu] + (rec_place[j] >> 16))) =
                mlm_u32x4{rec_cell[j], rec_pos[j], (uint32_t)rec_mask[j], (uint32_t)(rec_mask[j] >> 32)};
        }
    }
    const bool owner = MODE == 0 ? (wid < S && lane == 0) : threadIdx.x == 0; // the thread that knows strip `wid`'s record count
    if (owner) {
        const int j = MODE == 0 ? wid : 0;
        if (strip0 + (unsigned int)j < n_strips) {
            unsigned int pts = 0, oor = 0;
            for (unsigned int w = 0; w < 4; ++w) {
                pts += s_cnt[j][w] & 1023u;
                oor += s_cnt[j][w] >> 10;
            }
            *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.blk_stats) + 4 * (size_t)(strip0 + (unsigned int)j)) = mlm_u32x4{pts, oor, run_tot, 0u};
        }
        if (threadIdx.x == 0 && s_over) mlm_sector_fail(P, F);
    }
    if (MODE == 0 && wid >= S) return;
    if (run_cnt) { // one chunk descriptor per run, its place in the column's list from a returning atomic
        const int j = MODE == 0 ? wid : 0;
        const uint32_t ph = s_col_phi[j][MODE == 0 ? lane : (int)threadIdx.x];
        const unsigned int k = g_atomic_add(&mlm_gp(P.col_cnt)[ph], 1u);
        if (k < P.chunk_cap)
            *(MLM_GLOBAL mlm_u32x2 *)(mlm_gp(P.col_chunks) + 2 * ((size_t)ph * P.chunk_cap + k)) = mlm_u32x2{(strip0 + (unsigned int)j) * 256u + run_off, run_cnt};
        else
            mlm_sector_fail(P, F);
    }
}
template <int MODE, int S>
__global__ __launch_bounds__(256) void k_bin_sectors(MLM_SLOT_ARGS, unsigned int n_strips) {
    MLM_SLOT_SETUP
    mlm_bin_sectors_body<MODE, S>(P, F, n_strips);
}
// First node of a SMALL frame's graph (at most kHostFrameStrips strips: the reference's 500-sample callback): the frame's parameters
// come straight from pinned host memory — every workgroup fetches the 248 bytes across the link into LDS, the first one leaves the
// copy in the device-resident table for the kernels behind it — and the slot's counters are already clear (the last workgroup of the
// previous frame's k_apply_single cleared them after handing them to the host): no prologue kernel in front of this one.
template <int MODE>
__global__ __launch_bounds__(256) void k_bin_sectors_hostf(const MlmDev *__restrict__ slot_tab, const MlmFrame *host_frame, MlmFrame *dev_frame,
                                                           int slot_base, unsigned int n_strips, const int32_t *list_pix, const int32_t *list_raw,
                                                           unsigned int list_n) {
    const MlmDev &P = slot_tab[slot_base];
    __shared__ MlmFrame s_F;
    static_assert(sizeof(MlmFrame) % 4 == 0 && sizeof(MlmFrame) / 4 <= 256, "one word per thread");
    // (list_pix / list_raw: where the frame's pixel list and depths will turn out to be — the callback's pinned staging buffer, the same
    // for every call —, so that a thread's entry crosses the link together with the parameters instead of after them)
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    bool pre = MODE == 1 && list_pix && list_raw && i < list_n;
    int pre_pix = 0, pre_raw = 0;
    if (pre) {
        pre_pix = mlm_gp(list_pix)[i];
        pre_raw = mlm_gp(list_raw)[i];
    }
    if (threadIdx.x < sizeof(MlmFrame) / 4) {
        const uint32_t w = __hip_atomic_load((const uint32_t *)host_frame + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ((uint32_t *)&s_F)[threadIdx.x] = w;
        if (blockIdx.x == 0) ((uint32_t *)dev_frame)[threadIdx.x] = w;
    }
    __syncthreads();
    pre = pre && s_F.pix == list_pix && s_F.raw == list_raw;
    mlm_bin_sectors_body<MODE, 1>(P, s_F, n_strips, pre, pre_pix, pre_raw);
}

// find (or with INSERT create) the table entry of a column-local cell key; -1: table full
// (an insertion gives up after 96 probes: at the load a table is meant for, probe sequences are a handful long — a table that
// makes them longer is as good as full, and every further probe of a full table is a wasted LDS atomic)
template <bool INSERT>
__device__ __forceinline__ int mlm_sec_entry(MlmSecCell *tab, uint32_t tab_mask, uint32_t key) {
    uint32_t e = ((key * 2654435761u) >> 12) & tab_mask;
    const uint32_t max_probe = INSERT ? min(tab_mask, 95u) : tab_mask;
    for (uint32_t probe = 0; probe <= max_probe; ++probe) {
        if (INSERT) {
            const uint32_t prev = atomicCAS(&tab[e].key, MLM_NIL, key);
            if (prev == MLM_NIL || prev == key) return (int)e;
        } else {
            const uint32_t k = tab[e].key;
            if (k == MLM_NIL) return -1;
            if ((k & MLM_SEC_KEY_MASK) == key) return (int)e;
        }
        e = (e + 1) & tab_mask;
    }
    return -1;
}

// exclusive prefix sums over a workgroup of NW waves of four values per thread at once (one pair of barriers); v[] is
// replaced by this thread's offsets, total[] gets the sums.  s_w needs 4 * NW words.
template <int NW = MLM_SEC_WAVES>
__device__ __forceinline__ void mlm_block_excl_scan4(uint32_t (&v)[4], uint32_t *s_w, uint32_t (&total)[4]) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) incl[k] = mlm_wave_incl_scan(v[k]);
    __syncthreads(); // s_w may still be read from a previous use
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_w[4 * wid + k] = incl[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t t = s_w[4 * w + k];
            if (w < wid) off += t;
            tot += t;
        }
        total[k] = tot;
        v[k] = off + incl[k] - v[k];
    }
}
template <int NW = MLM_SEC_WAVES>
__device__ __forceinline__ uint32_t mlm_block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t *total) {
    uint32_t a[4] = {v, 0u, 0u, 0u}, t[4];
    mlm_block_excl_scan4<NW>(a, s_w, t);
    *total = t[0];
    return a[0];
}

// The targets of a hit-centre record: its own cell (kind 0) and the +-d neighbours (kinds 2d-1, 2d), update_hits
// map_awareness.cpp:135-171.  f(column-local key, kind, rho of the target).  s3 = 3 * sigma_in_dr(rho).
template <class F>
__device__ __forceinline__ void mlm_sec_targets(const MlmDev &P, int rho, int phi, int zi, float s3, F &&f) {
    f((uint32_t)(zi * P.nRho + rho), 0, rho);
    const double slope = rho > 0 ? (zi - P.zc) / (rho * 1.0) : 0.0;
    for (int d = 1; mlm_spread_active(P, rho, d, s3); ++d) {
        // neighbour cells of step d (mlm_spread_cells), as column-local keys
        int rz = mlm_cvt_int(round(zi + (d * slope)));
        if (0 <= rz && rz < P.nZ) f((uint32_t)(rz * P.nRho + rho + d), 2 * d - 1, rho + d);
        rz = mlm_cvt_int(round(zi - (d * slope)));
        if (0 <= rz && rz < P.nZ && rho - d >= 0) f((uint32_t)(rz * P.nRho + rho - d), 2 * d, rho - d);
    }
}

// LDS plan of k_sector (dynamic): the host computes the same offsets
struct MlmSecLds {
    uint32_t tab, miss, odds, sigma, rays, occ, multi, chunk, ray_p0, vox, aux, total;
};
// n_miss: words of the column's miss table (bit mask: nZ * RW; frontier mode keeps insertion times: nZ * nRho)
__host__ __device__ inline MlmSecLds mlm_sec_lds(uint32_t TAB, uint32_t n_miss, uint32_t n_rho, uint32_t n_z, bool explore) {
    MlmSecLds L;
    uint32_t o = 0;
    L.tab = o;      o += TAB * (uint32_t)sizeof(MlmSecCell);
    {
        // staged chunk descriptors of the record passes; later the list of occupied table entries (L.occ)
        uint32_t b = 2u * MLM_SEC_CHUNKS * 4u;
        if (b < TAB * 2u) b = TAB * 2u;
        L.chunk = o;
        o += (b + 15u) & ~15u;
    }
    L.odds = o;     o += ((2u * MLM_DIFF_RANGE + 1u) * n_rho + 3u) & ~3u; // a byte per entry of the odds table: its strength
    L.sigma = o;    o += ((n_rho + 3u) & ~3u) * 4u;
    L.miss = o;     o += ((n_miss + 3u) & ~3u) * 4u;
    L.rays = o;     o += TAB * 2u;                       // table entries that start a ray
    L.occ = L.chunk;                                     // occupied table entries (= the column's unique hits): in the chunk
                                                         // staging, idle between the first record pass and the second
    L.multi = o;    if (explore) o += TAB * 4u;          // frontier mode: per table entry, the first point whose hit centre is the cell
    L.ray_p0 = o;   if (explore) o += TAB * 4u;          // frontier mode: first point of every ray start
    o = (o + 15u) & ~15u;
    L.vox = o;      o += (n_rho + n_z) * 16u;            // world voxel per axis: x, y by rho; z by z (see k_sector)
    L.aux = o;      o += n_rho * 24u;                    // per tile run along rho: tile, first rho, hits (count, offset); per rho: miss cells
                                                         // (count -> fill cursor, offset)
    L.total = (o + 15u) & ~15u;
    return L;
}

// the per-block statistics of k_bin_sectors folded into the frame's counters (one wave)
__device__ __forceinline__ void mlm_fold_bin_stats(const MlmDev &P, int n_bin_blocks) {
    const int lane = threadIdx.x & 63;
    unsigned int a = 0, b = 0, g = 0;
    for (int j = lane; j < n_bin_blocks; j += 64) {
        const mlm_u32x4 st = *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.blk_stats) + 4 * (size_t)j);
        a += st.x;
        b += st.y;
        g += st.z;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
        g += __shfl_xor(g, off, 64);
    }
    if (lane == 0) {
        mlm_gp(P.ctr)->n_points = a;
        mlm_gp(P.ctr)->n_oor = b;
        mlm_gp(P.ctr)->n_groups = g;
    }
}

// EX: frontier mode (use_exploration_frontiers).  The miss container's iteration order matters there (mlm_kernels_explore.h),
// so a miss cell keeps its first insertion time — point index * 256 + step of the ray, map_awareness.cpp:266-274 — instead
// of a bit, and the map-dependent part is frontier mode's own (explore_stage_bc): the kernel ends with the unique hit list
// (+ world voxels) and the unique miss list (cell, time, world voxel), no frame-local grid.
// One azimuth column.  BIG: the second pass over the columns whose cell table overflowed in the first (k_sector_big: a table
// of MlmDev::sec_tab_big entries, one workgroup per CU) — the first pass leaves such a column untouched and puts it on the
// frame's overflow list instead of sending the whole frame to the cell-table path.
// NT: threads of the workgroup (512, or 256 for columns whose cell table has at most 1 024 entries: the smaller workgroup leaves
// wave slots and LDS of its CU to the other streams' kernels — measured +5 % frames/s in the pipeline although the kernel
// alone is 10 % slower).
template <bool EX, bool BIG, int NT>
__device__ __forceinline__ void mlm_sector_column(const MlmDev &P, const MlmFrame &F, const int phi, int tile_w, int n_bin_blocks, unsigned long long rho_m,
                                                  int rho_s, unsigned long long n_bkt, int col_flags, unsigned long long row_m, int row_s) {
    // col_flags: bit 0 — k_sector_big follows this launch; bits 4-7 — lanes per ray = 4 << that (a lone frame's columns have the lanes)
    const int big_armed = col_flags & 1;
    const uint32_t ray_sh = 2u + (((uint32_t)col_flags >> 4) & 15u);
    constexpr int PER_MAX = BIG ? 8 : 4; // cell-table entries per thread
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // (four columns in five of a camera frame hold nothing: the count comes through the scalar cache — sixteen columns share a
    // line of it — and an empty column's workgroup is gone after a fraction of a trip to memory)
#ifdef MLM_PHASE_PROF
    const long long fn_t0_ = clock64();
#endif
    const unsigned int nch_all = mlm_uniform_word(P.col_cnt + phi);
    if (nch_all == 0) return; // nothing fell into this column (uniform)
    // (the column's first chunk descriptors are requested now and arrive while the tables below are set up: one dependent trip
    // to memory less in a column's life; what lies beyond the count is not looked at)
    mlm_u32x2 chunk_first = mlm_u32x2{0u, 0u};
    constexpr uint32_t CH = NT < MLM_SEC_CHUNKS ? NT : MLM_SEC_CHUNKS; // chunk descriptors staged per pass: one per thread
    if (threadIdx.x < min(64u, min(nch_all, P.chunk_cap))) // (one wave's worth: a column of a VGA frame has ~50)
        chunk_first = *(const MLM_GLOBAL mlm_u32x2 *)(mlm_gp(P.col_chunks) + 2 * ((size_t)phi * P.chunk_cap + threadIdx.x));
    MLM_PHASE_BEGIN
#ifdef MLM_PHASE_PROF
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_phase[9], (unsigned long long)(ph_t_ - fn_t0_));
#endif
    const unsigned int nch = min(nch_all, P.chunk_cap);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t TAB = BIG ? P.sec_tab_big : P.sec_tab, NMISS = (uint32_t)(P.nZ * (EX ? P.nRho : P.RW));
    const MlmSecLds L = mlm_sec_lds(TAB, NMISS, (uint32_t)P.nRho, (uint32_t)P.nZ, EX);
    MlmSecCell *s_tab = (MlmSecCell *)(s_dyn + L.tab);
    uint32_t *s_miss = (uint32_t *)(s_dyn + L.miss);
    uint8_t *s_strength = (uint8_t *)(s_dyn + L.odds); // strength (mlm_sec_strength) of every entry of the odds table: what the booking pass needs of it
    float *s_sigma = (float *)(s_dyn + L.sigma);
    uint16_t *s_rays = (uint16_t *)(s_dyn + L.rays);
    uint16_t *s_occ = (uint16_t *)(s_dyn + L.occ);
    uint32_t *s_p0 = (uint32_t *)(s_dyn + L.multi); // (EX only) first point whose hit centre is the cell, per table entry
    uint32_t *s_chunk_first = (uint32_t *)(s_dyn + L.chunk);
    uint32_t *s_chunk_start = s_chunk_first + MLM_SEC_CHUNKS;
    uint32_t *s_ray_p0 = (uint32_t *)(s_dyn + L.ray_p0); // (EX only)
    int4 *s_vr = (int4 *)(s_dyn + L.vox), *s_vz = s_vr + P.nRho;
    // per tile run r along rho: its tile, its first rho, its hits (count, then offset in the column's hit list);
    // per rho: its unique miss cells (count, then fill cursor) and their offset in the column's miss list
    uint32_t *s_run_tile = (uint32_t *)(s_dyn + L.aux), *s_run_rho = s_run_tile + P.nRho, *s_run_hits = s_run_rho + P.nRho,
             *s_run_off = s_run_hits + P.nRho, *s_rho_miss = s_run_off + P.nRho, *s_rho_off = s_rho_miss + P.nRho;
    __shared__ int s_kr;
    __shared__ uint32_t s_w[4 * (NT / 64)];
    __shared__ uint32_t s_base[8];
    __shared__ unsigned int s_fail, s_nouter, s_tab_full;
    __shared__ uint32_t s_ref_ov[8], s_ref_ov_n; // table entries whose reference count wrapped
    const MLM_GLOBAL uint32_t *chunks = mlm_gp(P.col_chunks) + 2 * (size_t)phi * P.chunk_cap;
    const MLM_GLOBAL mlm_u32x4 *recs = (const MLM_GLOBAL mlm_u32x4 *)mlm_gp(P.bnodes); // 16-byte column records (k_bin_sectors)
    // flat record r of the staged chunks -> index into `bnodes`
    auto rec_index = [&](uint32_t r, uint32_t n_staged) -> uint32_t {
        uint32_t lo = 0, hi = n_staged; // largest c with start[c] <= r
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_chunk_start[mid] <= r) lo = mid;
            else hi = mid;
        }
        return s_chunk_first[lo] + (r - s_chunk_start[lo]);
    };
    // stage the chunk descriptors [c0, c0 + n_staged): first record and running record count of each; returns the records in all
    auto stage_chunks = [&](uint32_t c0, uint32_t n_staged) -> uint32_t {
        uint32_t total = 0;
        const uint32_t j = threadIdx.x;
        mlm_u32x2 d = mlm_u32x2{0u, 0u};
        if (j < n_staged) d = (c0 == 0 && j < 64u) ? chunk_first : *(const MLM_GLOBAL mlm_u32x2 *)(chunks + 2 * (size_t)(c0 + j));
        const uint32_t off = mlm_block_excl_scan<NT / 64>(d.y, s_w, &total);
        if (j < n_staged) {
            s_chunk_first[j] = d.x;
            s_chunk_start[j] = off;
        }
        __syncthreads();
        return total;
    };
    const double col_cos = mlm_uniform_f64(P.cos_phi + phi), col_sin = mlm_uniform_f64(P.sin_phi + phi);
    // (the tables that come from memory first: a wait for a later load is a wait for every earlier one, and nothing below may wait
    // for the records)
    // (strength bytes and sigma3 lie in memory as they lie in LDS — MlmDev::sec_const —: one copy, requested together with the chunk
    // descriptors; a lone frame's column would otherwise spend two more trips to memory here)
    for (uint32_t e = threadIdx.x; e < P.sec_const_words; e += NT) ((uint32_t *)(s_dyn + L.odds))[e] = mlm_gp(P.sec_const)[e];
    // The first descriptors are staged and every thread's first record is requested BEFORE the tables are set up: the records
    // (a trip to HBM, the longest wait of a column's life under load) arrive while the workgroup initialises its LDS.
    const uint32_t pre_total = stage_chunks(0, min(nch, CH));
    mlm_u32x4 pre_a = mlm_u32x4{0u, 0u, 0u, 0u};
    if (threadIdx.x < pre_total) pre_a = recs[rec_index(threadIdx.x, min(nch, CH))];
    MLM_PHASE(7);
    for (uint32_t e = threadIdx.x; e < TAB; e += NT) {
        s_tab[e].key = MLM_NIL;
        s_tab[e].tmin = MLM_EMPTY_T;
        s_tab[e].kg = 0;
        s_tab[e].cnt = 0;
        if (EX) s_p0[e] = MLM_EMPTY_T;
    }
    for (uint32_t e = threadIdx.x; e < NMISS; e += NT) s_miss[e] = EX ? MLM_EMPTY_T : 0u;
    for (uint32_t e = threadIdx.x; e < (uint32_t)P.nRho; e += NT) {
        s_run_hits[e] = 0;
        s_rho_miss[e] = 0;
    }
    if (threadIdx.x == 0) {
        s_fail = (nch_all > P.chunk_cap || (P.sec_fail_every && (unsigned int)(EX ? F.pad2 : F.seq) % P.sec_fail_every == 0)) ? 1u : 0u;
        s_nouter = 0;
        s_tab_full = 0;
        s_ref_ov_n = 0;
    }
    __syncthreads();
    // Which world voxel a cell of this column falls into (get_global_idx / get_subbox_id of its centre moved by T_wa,
    // map_local.cpp:151,180) separates by axis: x and y depend on rho only (phi is the column's), z on z only.  One
    // evaluation of the FP64 sequences per rho and per z instead of one per hit and miss cell.
    //   frontier mode:  s_vr[rho] = {gx, gy, cx | cy << 8 | out-of-range << 16, -},  s_vz[z] = {gz, cz | out-of-range << 16, -, -}
    //   default:        s_vr[rho] = {tile of the frame-local grid, in-tile (y, x) bits, -, tile run},  s_vz[z] = {grid z, -, -, -}
    //                   (a voxel-in-tile index is vt = s_vr[rho].y * lv_nz + s_vz[z].x).  A coordinate the reference's two
    //                   independent divisions put outside [0, n) (the id-0 quirk of get_subbox_id, map_local.h:170) cannot be
    //                   expressed per axis: such a column — one in ~1e13 evaluations — sends its frame to the cell-table path.
    for (uint32_t e = threadIdx.x; e < (uint32_t)(P.nRho + P.nZ); e += NT) {
        double wx, wy, wz;
        if (e < (uint32_t)P.nRho) {
            // (mlm_cell_center_w with the column's cosine and sine from the scalar cache: no vector load between the request for the
            // records and their use)
            const double center_rho = P.dRho / 2 + ((int)e * P.dRho);
            wx = center_rho * col_cos + F.t_wa[0];
            wy = center_rho * col_sin + F.t_wa[1];
            int gx, gy, cx, cy;
            mlm_voxel_axis(P, wx, gx, cx);
            mlm_voxel_axis(P, wy, gy, cy);
            const bool bad = cx < 0 || cy < 0 || cx >= P.n || cy >= P.n;
            if (EX) {
                s_vr[e] = make_int4(gx, gy, bad ? 0x10000 : (cx | cy << 8), 0);
            } else {
                const int x = gx * P.n + cx - F.lv_o[0], y = gy * P.n + cy - F.lv_o[1], m = (1 << P.tile_sh) - 1;
                if (bad || (unsigned)x >= (unsigned)P.lv_nx || (unsigned)y >= (unsigned)P.lv_ny) s_fail = 1;
                s_vr[e] = make_int4((y >> P.tile_sh) * P.n_tx + (x >> P.tile_sh), ((y & m) << P.tile_sh) | (x & m), 0, 0);
            }
        } else {
            const int z = (int)e - P.nRho;
            wz = (P.z_border_min + (P.dZ / 2) + (z * P.dZ)) + F.t_wa[2];
            int gz, cz;
            mlm_voxel_axis(P, wz, gz, cz);
            const bool bad = cz < 0 || cz >= P.n;
            if (EX) {
                s_vz[z] = make_int4(gz, bad ? 0x10000 : cz, 0, 0);
            } else {
                const int zz = gz * P.n + cz - F.lv_o[2];
                if (bad || (unsigned)zz >= (unsigned)P.lv_nz) s_fail = 1;
                s_vz[z] = make_int4(zz, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (!EX && wid == 0) { // runs of equal tile along rho (a ray crosses a tile once): ballot scan over the change flags
        int carry = 0;
        for (int i0 = 0; i0 < P.nRho; i0 += 64) {
            const int i = i0 + lane;
            const bool change = i < P.nRho && (i == 0 || s_vr[i].x != s_vr[i - 1].x);
            const unsigned long long m = __ballot(change);
            const int id = carry + (int)__popcll(m & ((2ull << lane) - 1ull)) - 1;
            if (i < P.nRho) {
                s_vr[i].w = id;
                if (change) {
                    s_run_tile[id] = (uint32_t)s_vr[i].x;
                    s_run_rho[id] = (uint32_t)i;
                }
            }
            carry += (int)__popcll(m);
        }
        if (lane == 0) s_kr = carry;
    }
    __syncthreads();
    const int n_run = EX ? 0 : s_kr;
    // frontier mode: world voxel of cell (rho, z) of this column — what mlm_voxel_of(mlm_cell_center_w(...)) gives
    auto cell_voxel = [&](int rho, int z, int &gx, int &gy, int &gz, int &cid) {
        const int4 vr = s_vr[rho], vz = s_vz[z];
        gx = vr.x;
        gy = vr.y;
        gz = vz.x;
        int cx = vr.z & 0xFF, cy = (vr.z >> 8) & 0xFF, cz = vz.y & 0xFF;
        if ((vr.z | vz.y) >> 16) cx = cy = cz = 0; // a coordinate outside [0, n): id 0 (mlm_voxel_of)
        cid = cz * P.n * P.n + cy * P.n + cx;
    };
    MLM_PHASE(0);
    const uint32_t tab_mask = TAB - 1;
    auto key_rz = [&](uint32_t key, int &rho, int &z) { // key = z * nRho + rho (exact division by multiplication, key < 2^27)
        z = (int)(((unsigned long long)key * rho_m) >> rho_s);
        rho = (int)key - z * P.nRho;
    };
    // pass = 0: book every record's contributions on the cells of the column (and walk the rays of points outside the
    // map); pass = 1: write a (record, kind) reference for every contribution group of a multi-kind cell.  A thread
    // keeps the record it handled first: a column with at most NT records (the usual case) is not read twice.
    uint32_t keep_cell = MLM_NIL, keep_yx = 0, keep_total = 0xFFFFFFFFu;
    unsigned long long keep_mask = 0;
    // ... and the record it handled second as it came (its targets are looked up again): a column with at most 2 NT records — nearly
    // every column of a camera frame — is not read from memory twice (the second read, with its descriptors staged again, was a
    // fifth of the kernel's wave-cycles: tools/sector_phase.py)
    uint32_t keep2_cell = MLM_NIL, keep2_yx = 0;
    unsigned long long keep2_mask = 0;
    // ... and the table entries of its targets, so that the second pass neither recomputes them nor probes the table:
    // (entry | kind << 12) in 16 bits each, four in keep_lo, the fifth in keep_hi's low half, the count in its high half
    // (MLM_SEC_KEEP_MORE: more than five targets or a kind above 15 — the record is recomputed)
    unsigned long long keep_lo = 0;
    uint32_t keep_hi = 0;
    // the non-empty rows of a group's lane mask, once per record (three bits each): the same for every target cell
    auto rows_of = [&](unsigned long long mask, uint32_t &rows3, uint32_t &n_rows) {
        rows3 = 0, n_rows = 0;
#pragma unroll
        for (uint32_t row = 0; row < 8u; ++row)
            if ((uint32_t)(mask >> (8u * row)) & 0xFFu) rows3 |= row << (3u * n_rows++);
    };
    auto emit_refs = [&](int e, uint32_t sub, uint32_t yx, unsigned long long mask, uint32_t rows3, uint32_t n_rows) {
        if (!mlm_sec_needs_order(s_tab[e])) return;
        // one 4-byte reference per non-empty ROW of the group's lane mask (mlm_ref_pack): row byte, kind, and the row's
        // position relative to the cell's first pixel — what k_rank needs, with no empty rows in its rounds (a group
        // touches two or three of its eight rows)
        // (the count of the cell's references, counted up by the booking pass, is counted down here: every group gets its own
        // stretch of the cell's segment; the order of the references inside a segment does not matter)
        const uint32_t left = atomicSub(&s_tab[e].kg, n_rows << MLM_SEC_KIND_BITS) >> MLM_SEC_KIND_BITS;
        const uint32_t at = s_base[2] + (s_tab[e].key >> 16) * MLM_SEC_REF_ALIGN + (left - n_rows);
        const uint32_t pix0 = s_tab[e].tmin / MLM_TIME_SLOTS;
        const uint32_t y0c = tile_w > 0 ? (uint32_t)(((unsigned long long)pix0 * row_m) >> row_s) : pix0 >> 6;
        const uint32_t dy0 = (tile_w > 0 ? (yx >> MLM_REC_XT_BITS) << 3 : yx >> 11) - y0c; // (>= 0: the cell's first pixel is its contributions' smallest)
        // (dense images: the tile's column relative to the first pixel's tile column)
        const uint32_t xrel = tile_w > 0 ? (yx & MLM_REC_XT_MASK) - ((pix0 - y0c * (uint32_t)tile_w) >> 3) + MLM_REF_XREL0 : 0u;
        if (dy0 + 7u > (tile_w > 0 ? MLM_REF_DY_DENSE : MLM_REF_DY_LIST) || xrel > 255u) {
            s_fail = 1; // (a cell whose pixels lie more than 2 047 rows below or 1 016 columns beside its first one: not expressible — the frame falls back)
        } else if (at + n_rows <= P.refs_cap) {
            MLM_GLOBAL uint32_t *dst = mlm_gp(P.refs) + at;
            for (uint32_t k = 0; k < n_rows; ++k) {
                const uint32_t row = (rows3 >> (3u * k)) & 7u;
                dst[k] = mlm_ref_pack((uint32_t)(mask >> (8u * row)) & 0xFFu, sub, tile_w > 0, dy0, row, xrel);
            }
        }
    };
    auto refs_of = [&](uint32_t cell, uint32_t yx, unsigned long long mask) {
        const int z = (int)(cell >> 16), rho = (int)(cell & 0xFFFFu);
        uint32_t rows3, n_rows;
        rows_of(mask, rows3, n_rows);
        mlm_sec_targets(P, rho, phi, z, s_sigma[rho], [&](uint32_t key, int sub, int) {
            const int e = mlm_sec_entry<false>(s_tab, tab_mask, key);
            if (e >= 0) emit_refs(e, (uint32_t)sub, yx, mask, rows3, n_rows);
        });
    };
    auto refs_of_kept = [&]() {
        const uint32_t nt = keep_hi >> 16;
        if (nt == MLM_SEC_KEEP_MORE) {
            refs_of(keep_cell, keep_yx, keep_mask);
            return;
        }
        uint32_t rows3, n_rows;
        rows_of(keep_mask, rows3, n_rows);
        for (uint32_t k = 0; k < nt; ++k) {
            const uint32_t t = k < 4u ? (uint32_t)(keep_lo >> (16u * k)) & 0xFFFFu : keep_hi & 0xFFFFu;
            emit_refs((int)(t & 0xFFFu), t >> 12, keep_yx, keep_mask, rows3, n_rows);
        }
    };
    auto for_records = [&](int pass) {
        if (pass == 1 && nch <= CH && keep_total <= 2u * NT) {
            if (keep_cell != MLM_NIL) refs_of_kept();
            if (keep2_cell != MLM_NIL) refs_of(keep2_cell, keep2_yx, keep2_mask);
            return;
        }
        for (uint32_t c0 = 0; c0 < nch; c0 += CH) {
            const uint32_t n_staged = min(nch - c0, CH);
            uint32_t total = pre_total; // (pass 0, first descriptors: staged before the set-up)
            if (pass != 0 || c0 != 0) {
                __syncthreads();
                total = stage_chunks(c0, n_staged);
            }
            for (uint32_t r = threadIdx.x; r < total; r += NT) {
                if (pass == 0 && *(volatile unsigned int *)&s_tab_full) break; // (the column is given up: nothing more to book)
                mlm_u32x4 a = pre_a;
                if (pass != 0 || c0 != 0 || r != threadIdx.x) a = recs[rec_index(r, n_staged)];
                const unsigned long long rec_mask = (unsigned long long)a.z | (unsigned long long)a.w << 32;
#ifdef MLM_PHASE_PROF
                if (pass == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    MLM_PHASE(8);
                }
#endif
                if (a.x & MLM_SEC_OUTER) {
                    if (pass == 0) { // ray of a point outside the map: one lane walks it into the LDS mask
                        int rho = (int)(a.x & ~MLM_SEC_OUTER), z = (int)a.y;
                        const double slope = (rho > 0) ? (z - P.zc) / (rho * 1.0) : 0.0;
                        if (rho >= P.nRho) {
                            z = mlm_cvt_int(round(z - ((rho - P.nRho + 1) * slope)));
                            rho = P.nRho - 1;
                        }
                        const uint32_t p0 = a.z; // the record's first point (EX: several records may start the same ray, the minimum wins)
                        for (int rr = 1; rr < rho; ++rr) {
                            const int zr = mlm_cvt_int(round(z - ((rho - rr) * slope)));
                            if (0 <= zr && zr < P.nZ) {
                                if (EX) atomicMin(&s_miss[zr * P.nRho + rr], p0 * 256u + (uint32_t)(rho - rr - 1));
                                else atomicOr(&s_miss[zr * P.RW + (rr >> 5)], 1u << (rr & 31));
                            }
                        }
                        atomicAdd(&s_nouter, 1u);
                    }
                    continue;
                }
                const uint32_t cell = a.x;
                if (pass == 1) {
                    refs_of(cell, a.y, rec_mask);
                    continue;
                }
                const int z = (int)(cell >> 16), rho = (int)(cell & 0xFFFFu);
                {
                    const unsigned long long mask = rec_mask;
                    const bool kept = c0 == 0 && r == threadIdx.x;
                    if (kept) {
                        keep_cell = cell;
                        keep_yx = a.y;
                        keep_mask = mask;
                    }
                    if (c0 == 0 && r == threadIdx.x + (uint32_t)NT) {
                        keep2_cell = cell;
                        keep2_yx = a.y;
                        keep2_mask = mask;
                    }
                    uint32_t nt = 0;
                    const int l0 = __ffsll((long long)mask) - 1; // lowest lane = earliest insertion time of the record
                    // (the wave's first work item from the tile origin: dense y0 * width + x0, lists 64 items per "row")
                    const uint32_t i00 = tile_w > 0 ? ((a.y >> MLM_REC_XT_BITS) << 3) * (uint32_t)tile_w + ((a.y & MLM_REC_XT_MASK) << 3) : (a.y >> 11) << 6;
                    const uint32_t i_first = i00 + (tile_w > 0 ? (uint32_t)((l0 >> 3) * tile_w + (l0 & 7)) : (uint32_t)l0);
                    const uint32_t cnt = (uint32_t)__popcll(mask), n_rows = mlm_mask_rows(mask);
                    mlm_sec_targets(P, rho, phi, z, s_sigma[rho], [&](uint32_t key, int sub, int rho_t) {
                        const int e = mlm_sec_entry<true>(s_tab, tab_mask, key);
                        if (e < 0) {
                            s_tab_full = 1;
                            return;
                        }
                        if (kept) {
                            const uint32_t t = (uint32_t)e | (uint32_t)sub << 12;
                            if (nt < 4u) keep_lo |= (unsigned long long)t << (16u * nt);
                            else keep_hi |= t & 0xFFFFu;
                            if (nt >= 5u || sub > 15) nt = MLM_SEC_KEEP_MORE - 1u;
                            ++nt;
                        }
                        atomicMin(&s_tab[e].tmin, i_first * MLM_TIME_SLOTS + (uint32_t)sub);
                        atomicOr(&s_tab[e].kg, 1u << sub);
                        // contributions, and in the upper 11 bits (mod 2048) the sum of their strengths (mlm_sec_needs_order)
                        const uint32_t strength = s_strength[mlm_contribution_index(P, rho_t, sub)];
                        atomicAdd(&s_tab[e].cnt, cnt | ((cnt * strength) << MLM_SEC_CNT_BITS));
                        if ((atomicAdd(&s_tab[e].kg, n_rows << MLM_SEC_KIND_BITS) >> MLM_SEC_KIND_BITS) + n_rows > (0xFFFFFFFFu >> MLM_SEC_KIND_BITS)) {
                            // more than 2 047 references: the count has wrapped.  That happens to cells a few decimetres in front of
                            // the sensor (thousands of pixels), which have one kind or saturate and need no references at all: the entry
                            // is remembered and the frame only gives up if such a cell does need its order (checked below)
                            const uint32_t k = atomicAdd(&s_ref_ov_n, 1u);
                            if (k < 8u) s_ref_ov[k] = (uint32_t)e;
                        }
                        if (EX && sub == 0) atomicMin(&s_p0[e], i_first);
                    });
                    if (kept) keep_hi |= nt << 16;
                }
            }
            if (pass == 0 && c0 == 0) keep_total = total;
        }
    };
    __syncthreads();
    for_records(0);
    __syncthreads();
    if (s_tab_full) { // (uniform) nothing has left the workgroup yet
        if (threadIdx.x == 0) {
            if (!BIG && P.sec_tab_big && !s_fail && (big_armed || !EX)) {
                // the column keeps its chunk descriptors and waits for the pass with the large table.  big_armed: k_sector_big follows
                // this launch; else the frame is flagged (sector_overflow = 1): nothing of it is applied, the host runs the large-table
                // pass and the rest of Stage A for it when it drains (redo_overflow_columns) and schedules the pass for the batches to
                // come — no frame goes to the cell-table path because a scene's first crowded column surprised the small table
                const unsigned int k = g_atomic_add(&mlm_gp(P.ctr)->n_ov, 1u);
                mlm_gp(P.ov_list)[k] = (uint32_t)phi; // (k < nPhi: a column is listed once)
                if (!big_armed) {
                    g_atomic_max(&mlm_gp(P.ctr)->sector_overflow, 1u);
                    g_atomic_min(&mlm_gp(P.g)->fail_frame, F.seq);
                }
            } else {
                mlm_sector_fail(P, F);
                mlm_gp(P.col_cnt)[phi] = 0;
            }
        }
        return;
    }
    if (threadIdx.x == 0 && s_ref_ov_n) { // (see the booking pass: a wrapped reference count only matters for a cell that needs its order)
        if (s_ref_ov_n > 8u) s_fail = 1;
        for (uint32_t k = 0; k < min(s_ref_ov_n, 8u); ++k)
            if (mlm_sec_needs_order(s_tab[s_ref_ov[k]])) s_fail = 1;
    }
    MLM_PHASE(1);
    // ---- lists of the occupied entries (= the column's unique hits), of those with several kinds, of the ray starts;
    //      the column's reservations in the frame's lists (one round trip)
    //      The hit list is ordered by tile run (the hits of one tile are contiguous): rank inside the run from a returning
    //      LDS atomic, the runs' offsets from the same block scan that places the other lists.
    const uint32_t per = TAB / NT; // entries e = threadIdx.x * per + q: contiguous per thread (per <= PER_MAX)
    uint32_t v[4] = {0u, 0u, 0u, 0u}; // occupied, multi, ray starts, hits of tile run `threadIdx.x`
    uint32_t w[4] = {0u, 0u, 0u, 0u}; // reference slots (a cell's count rounded up to MLM_SEC_REF_ALIGN) / ordered-kinds slots of this thread's multi-kind cells
    uint32_t hk[PER_MAX];           // this thread's entries: rank among their tile run's hits
    uint32_t p0r[EX ? PER_MAX : 1]; // (EX) first point whose centre is the cell
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)PER_MAX; ++q) {
        if (q >= per) break;
        const MlmSecCell &c = s_tab[threadIdx.x * per + q];
        if (EX) p0r[EX ? q : 0] = 0;
        if (c.key == MLM_NIL) continue;
        ++v[0];
        v[2] += c.kg & 1u;
        if (mlm_sec_needs_order(c)) {
            ++v[1];
            w[0] += ((c.kg >> MLM_SEC_KIND_BITS) + MLM_SEC_REF_ALIGN - 1u) & ~(MLM_SEC_REF_ALIGN - 1u);
            w[1] += ((c.cnt & MLM_SEC_CNT_MASK) + 15u) & ~15u;
        }
        if (EX) { // (frontier mode has no tiles: the hits keep the table's order)
            hk[q] = v[0] - 1u;
            p0r[EX ? q : 0] = s_p0[threadIdx.x * per + q];
        } else {
            int rho, z;
            key_rz(c.key, rho, z);
            hk[q] = atomicAdd(&s_run_hits[s_vr[rho].w], 1u);
        }
    }
    __syncthreads(); // (the runs' hit counts are complete)
    if ((int)threadIdx.x < n_run) v[3] = s_run_hits[threadIdx.x]; // (n_run <= nRho <= NT)
    uint32_t tot[4], wtot[4];
    mlm_block_excl_scan4<NT / 64>(v, s_w, tot);
    mlm_block_excl_scan4<NT / 64>(w, s_w, wtot);
    if ((int)threadIdx.x < n_run) s_run_off[threadIdx.x] = v[3];
    const uint32_t n_occ = tot[0], n_multi = tot[1], n_rays = P.visibility ? tot[2] : 0u;
    const uint32_t tot_refs = wtot[0], tot_subs = wtot[1];
    if (threadIdx.x == 0) {
        s_base[0] = n_occ ? g_atomic_add(&mlm_gp(P.ctr)->u_hit, n_occ) : 0u;
        s_base[1] = n_multi ? g_atomic_add(&mlm_gp(P.ctr)->n_multi, n_multi) : 0u;
        s_base[2] = tot_refs ? g_atomic_add(&mlm_gp(P.ctr)->n_refs, tot_refs) : 0u;
        s_base[3] = tot_subs ? g_atomic_add(&mlm_gp(P.ctr)->n_contrib, tot_subs) : 0u;
        // (a cell's segment start is kept in 16 bits, in units of MLM_SEC_REF_ALIGN references from the column's first)
        if (tot_refs >= 65536u * MLM_SEC_REF_ALIGN) s_fail = 1;
        // (the frame's lists are sized by need: a column that does not fit writes nothing — s_fail 2, the slots are enlarged and the
        // frame's Stage A runs again)
        if (s_base[0] + n_occ > P.hl_cap || s_base[1] + n_multi > P.mt_cap || s_base[2] + tot_refs > P.refs_cap || s_base[3] + tot_subs > P.contrib_cap)
            s_fail = 2;
    }
    __syncthreads(); // (the runs' offsets and the column's places in the frame's lists are visible)
    MLM_PHASE(2);
    if (s_fail) { // a table of this column overflowed: the frame is redone (uniform branch)
        if (threadIdx.x == 0) {
            mlm_sector_fail(P, F, s_fail == 2u);
            mlm_gp(P.col_cnt)[phi] = 0;
        }
        return;
    }
    // ---- lists of the occupied entries (= the column's unique hits, ordered by tile run: the hits of one tile are contiguous;
    //      rank inside the run from the returning LDS atomic above, the runs' offsets from the block scan) and of the ray starts;
    //      multi-kind cells that need their order: segments in `refs` and `subs`, descriptors for k_rank / k_chain_lanes — the
    //      thread that owns a table entry knows all of it (places from the scans above)
    {
        uint32_t o_multi = v[1], o_rays = v[2], o_refs = w[0], o_subs = w[1];
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)PER_MAX; ++q) {
            if (q >= per) break;
            const uint32_t e = threadIdx.x * per + q;
            MlmSecCell &c = s_tab[e];
            if (c.key == MLM_NIL) continue;
            if (c.kg & 1u) {
                if (EX) s_ray_p0[o_rays] = p0r[EX ? q : 0];
                s_rays[o_rays++] = (uint16_t)e;
            }
            int c_rho, c_z;
            key_rz(c.key, c_rho, c_z);
            const uint32_t place = (EX ? v[0] : s_run_off[s_vr[c_rho].w]) + hk[q];
            s_occ[place] = (uint16_t)e;
            if (mlm_sec_needs_order(c)) {
                const uint32_t n_ref = c.kg >> MLM_SEC_KIND_BITS, n_con = c.cnt & MLM_SEC_CNT_MASK;
                const uint32_t pos = s_base[0] + place, m = s_base[1] + o_multi, g_refs = s_base[2] + o_refs, g_subs = s_base[3] + o_subs;
                // (contributions | rho << 20: what k_chain_lanes needs of the cell comes with one load)
                *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.mt_rec) + m) = mlm_u32x4{pos, g_subs, n_con | ((uint32_t)c_rho << MLM_SEC_CNT_BITS), c.tmin};
                *(MLM_GLOBAL mlm_u32x2 *)(mlm_gp(P.mt_ref) + 2 * (size_t)m) = mlm_u32x2{g_refs, n_ref};
                c.key = (c.key & MLM_SEC_KEY_MASK) | ((o_refs / MLM_SEC_REF_ALIGN) << 16);
                ++o_multi;
                o_refs += (n_ref + MLM_SEC_REF_ALIGN - 1u) & ~(MLM_SEC_REF_ALIGN - 1u);
                o_subs += (n_con + 15u) & ~15u;
            }
        }
    }
    __syncthreads();
    // ---- rays of the cells that hold a hit centre (every point of one (rho,phi,z) cell casts the identical ray,
    //      map_awareness.cpp:243-274: once per cell), FOUR LANES per ray (sixteen for a lone frame), each a share of its steps.
    //      z' = round(z - k (z - zc) / rho) for k = 1 .. rho-1 is followed by an integer DDA on
    //      N_k = 2 (z rho - k (z - zc)) + rho: z' = floor(N_k / 2 rho) unless N_k is a multiple of 2 rho — the exact value
    //      is then a half-integer and only the reference's own FP64 sequence (slope = dz / rho rounded, k * slope rounded,
    //      z - .. rounded, round half away) says which way it goes (those steps are collected and evaluated afterwards);
    //      everywhere else that sequence is at most ~1e-12 away from the exact value, which is at least 1 / (2 rho) away
    //      from the next half-integer.  Consecutive steps that fall into one word of the mask are merged in a register.
    const uint32_t ray_lanes = 1u << ray_sh;
    for (uint32_t i0 = 0; i0 < (n_rays << ray_sh); i0 += NT) {
        const uint32_t it = i0 + threadIdx.x;
        int rho = 0, z = 0;
        uint32_t t0 = 0; // EX: insertion time of the ray's step k is t0 + k - 1
        if (it < (n_rays << ray_sh)) {
            key_rz(s_tab[s_rays[it >> ray_sh]].key & MLM_SEC_KEY_MASK, rho, z);
            if (EX) t0 = s_ray_p0[it >> ray_sh] * 256u;
        }
        const int seg = (rho + (int)ray_lanes - 2) >> ray_sh; // steps k = 1 .. rho-1 in `ray_lanes` segments of `seg`
        const int k_lo = 1 + (int)(it & (ray_lanes - 1u)) * seg, k_hi = min(rho, k_lo + seg); // [k_lo, k_hi)
        const int dz = z - P.zc, two_rho = 2 * rho;
        // per step: N -= 2 dz = sq * 2 rho + fr with 0 <= fr < 2 rho
        int sq = 0, fr = 0, q = 0, rem = 0;
        if (k_lo < k_hi) {
            auto floor_div = [&](int a, int &quo, int &r) { // a = quo * two_rho + r, 0 <= r < two_rho (|a| < 2^24)
                quo = (int)floorf((float)a / (float)two_rho);
                r = a - quo * two_rho;
                if (r < 0) {
                    r += two_rho;
                    --quo;
                } else if (r >= two_rho) {
                    r -= two_rho;
                    ++quo;
                }
            };
            floor_div(2 * dz, sq, fr);
            floor_div(rho * (2 * z + 1) - (k_lo - 1) * 2 * dz, q, rem); // N at k_lo - 1
        }
        int cur_w = -1;
        uint32_t cur_bits = 0;
        unsigned long long ties = 0;
        for (int k = k_lo; k < k_hi; ++k) {
            const int r = rho - k;
            rem -= fr;
            q -= sq;
            if (rem < 0) {
                rem += two_rho;
                --q;
            }
            if (rem == 0) { // exact tie: the reference's FP64 sequence decides (below)
                ties |= 1ull << ((k - k_lo) & 63);
                if (k - k_lo < 64) continue;
            }
            int zr = q;
            if (rem == 0) zr = mlm_cvt_int(round(z - (k * (dz / (rho * 1.0))))); // (segments longer than 64 steps: nRho > 256)
            if (EX) {
                if (0 <= zr && zr < P.nZ) atomicMin(&s_miss[zr * P.nRho + r], t0 + (uint32_t)(k - 1));
                continue;
            }
            const int w = (0 <= zr && zr < P.nZ) ? zr * P.RW + (r >> 5) : -1;
            if (w != cur_w) {
                if (cur_w >= 0) atomicOr(&s_miss[cur_w], cur_bits);
                cur_w = w;
                cur_bits = 0;
            }
            cur_bits |= 1u << (r & 31);
        }
        if (!EX && cur_w >= 0) atomicOr(&s_miss[cur_w], cur_bits);
        if (__any(ties != 0)) {
            const double slope = (rho > 0) ? dz / (rho * 1.0) : 0.0;
            while (ties) {
                const int k = k_lo + __ffsll((long long)ties) - 1;
                ties &= ties - 1;
                const int r = rho - k;
                const int zr = mlm_cvt_int(round(z - (k * slope)));
                if (0 <= zr && zr < P.nZ) {
                    if (EX) atomicMin(&s_miss[zr * P.nRho + r], t0 + (uint32_t)(k - 1));
                    else atomicOr(&s_miss[zr * P.RW + (r >> 5)], 1u << (r & 31));
                }
            }
        }
    }
    __syncthreads(); // (the miss mask is complete)
    MLM_PHASE(3);
    // ---- the column's unique hits (ordered by tile run): cell, first-touch time, voxel-in-tile index; single-kind cells get
    //      their odd and increment here (multi-kind cells: k_rank / k_chain_lanes)
    const unsigned int sl = blockIdx.x & 7;
    for (uint32_t i = threadIdx.x; i < n_occ; i += NT) {
        const MlmSecCell c = s_tab[s_occ[i]];
        int rho, z;
        key_rz(c.key & MLM_SEC_KEY_MASK, rho, z);
        const uint32_t pos = s_base[0] + i;
        mlm_gp(P.hl_cell)[pos] = (uint32_t)(z * P.nRhoPhi + phi * P.nRho + rho);
        mlm_gp(P.hl_t)[pos] = c.tmin;
        mlm_gp(P.hl_vt)[pos] = c.tmin; // (the replay kernels re-rank it: k_assign_rank)
        if (!mlm_sec_needs_order(c)) {
            // one kind: cnt applications of one value (update_odds_hashmap, map_awareness.h:147-154), 1.0f is absorbing;
            // several kinds with enough strong contributions: 1.0f in any order; two contributions: their product (mlm_sec_needs_order)
            const float a = mlm_gp(P.odds_table)[mlm_contribution_index(P, rho, __ffs((int)(c.kg & MLM_SEC_KIND_MASK)) - 1)]; // (one value per hit cell: from memory)
            float p = a;
            if (__popc(c.kg & MLM_SEC_KIND_MASK) > 1) {
                if ((c.cnt >> MLM_SEC_CNT_BITS) >= MLM_SEC_STRONG_ENOUGH) {
                    p = 1.0f;
                } else { // two contributions of two kinds: the same value in either order
                    const uint32_t rest = c.kg & MLM_SEC_KIND_MASK & ((c.kg & MLM_SEC_KIND_MASK) - 1u);
                    const float b = mlm_gp(P.odds_table)[mlm_contribution_index(P, rho, __ffs((int)rest) - 1)];
                    p = 1 - (1 - a) * (1 - b);
                }
            } else {
                for (uint32_t j = 1; j < (c.cnt & MLM_SEC_CNT_MASK) && p != 1.0f; ++j) p = 1 - (1 - p) * (1 - a);
            }
            if (EX || P.record_awareness) mlm_gp(P.hl_odd)[pos] = p; // (the odd itself is only read back by mlm_get_awareness_hits)
            mlm_gp(P.hl_inc)[pos] = mlm_logit(P, p);
        }
        if (EX) { // frontier mode: the hit's world voxel (its kernels look the block up themselves)
            int gx, gy, gz, cid;
            cell_voxel(rho, z, gx, gy, gz, cid);
            mlm_gp(P.hl_bkey)[pos] = mlm_pack_key(gx, gy, gz);
            mlm_gp(P.hl_cid)[pos] = (uint32_t)cid;
            mlm_gp(P.hl_slot)[pos] = -1;
            continue;
        }
        // its voxel (k_tile groups the frame's hits and misses by voxel, tile by tile) and the bucket-first time of the
        // emulated container (iteration order, see Stage B in mlm_kernels.h) for the bucket count the frame was submitted with
        mlm_gp(P.hl_vt16)[pos] = (uint16_t)(s_vr[rho].y * P.lv_nz + s_vz[z].x);
        // (a fire-and-forget atomic: measured 32-byte transactions, 1.2 MB per VGA frame with k_rank's read-back — walking the hit
        // list once per class of buckets with the minima in LDS instead, profiles/r4c, cost 4.6 MB of re-reads and 2.5 us per frame)
        const unsigned long long b = mlm_hash_rpz(rho, phi, z) % n_bkt;
        mlm_gp(P.hl_bkt)[pos] = (uint32_t)b;
        if (b < P.sbkt_cap) g_atomic_min(&mlm_gp(P.sbkt)[b], mlm_bkt_entry(F.seq, c.tmin));
    }
    MLM_PHASE(4);
    // ---- references of the multi-kind cells (their fill cursors were set above)
    if (n_multi) for_records(1);
    MLM_PHASE(5);
    if (EX) {
        // frontier mode: the unique miss list with insertion times and world voxels (what k_ex_collect_misses leaves);
        // a thread's cells take consecutive places
        uint32_t vm = 0;
        for (uint32_t w = threadIdx.x; w < NMISS; w += NT) vm += s_miss[w] != MLM_EMPTY_T ? 1u : 0u;
        uint32_t total;
        const uint32_t my_off = mlm_block_excl_scan<NT / 64>(vm, s_w, &total);
        if (threadIdx.x == 0) {
            s_base[5] = total ? g_atomic_add(&mlm_gp(P.ctr)->n_ex_miss, total) : 0u;
            if (n_rays + s_nouter) g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[sl][0], n_rays + s_nouter); // statistic only
            g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[sl][1], nch_all + 8u); // device-scope atomics on account of this column
            mlm_gp(P.col_cnt)[phi] = 0; // consumed: clean for the slot's next frame
        }
        __syncthreads();
        uint32_t at = s_base[5] + my_off;
        for (uint32_t w = threadIdx.x; w < NMISS; w += NT) {
            const uint32_t t = s_miss[w];
            if (t == MLM_EMPTY_T) continue;
            int rho, z;
            key_rz(w, rho, z);
            int gx, gy, gz, m_cid;
            cell_voxel(rho, z, gx, gy, gz, m_cid);
            mlm_gp(P.ex_cell)[at] = (uint32_t)(z * P.nRhoPhi + phi * P.nRho + rho);
            mlm_gp(P.ex_t)[at] = t;
            mlm_gp(P.ex_vt)[at] = t;
            mlm_gp(P.ex_bkey)[at] = mlm_pack_key(gx, gy, gz);
            mlm_gp(P.ex_cid)[at] = (uint32_t)m_cid;
            ++at;
        }
    } else {
        // ---- the column's unique miss cells (its bit mask) leave as voxel-in-tile indices, ordered by rho and therefore by
        //      tile run: every miss adds the same constant (map_local.cpp:188-192), only the count per voxel matters, and
        //      k_tile counts.  Miss cells per rho (LDS atomics), their offsets (one block scan), the list in LDS (the cell
        //      table's space is idle by now), one coalesced copy into the column's own stretch of the frame's miss list; then the
        //      descriptors of its tile runs (below).
        // (a byte of a mask word per thread: the bits of a word are a serial chain of LDS atomics)
        for (uint32_t it = threadIdx.x; it < 4u * NMISS; it += NT) {
            const uint32_t w = it >> 2, part = (it & 3u) * 8u;
            uint32_t bits = (s_miss[w] >> part) & 0xFFu;
            const uint32_t z = w / (uint32_t)P.RW, rho0 = (w - z * (uint32_t)P.RW) * 32u + part;
            while (bits) {
                atomicAdd(&s_rho_miss[rho0 + (uint32_t)__ffs((int)bits) - 1u], 1u);
                bits &= bits - 1;
            }
        }
        __syncthreads();
        uint32_t total = 0;
        {
            const uint32_t c = (int)threadIdx.x < P.nRho ? s_rho_miss[threadIdx.x] : 0u; // (nRho <= NT on this path)
            const uint32_t off = mlm_block_excl_scan<NT / 64>(c, s_w, &total);
            if ((int)threadIdx.x < P.nRho) {
                s_rho_off[threadIdx.x] = off;
                s_rho_miss[threadIdx.x] = 0; // (now the fill cursor)
            }
        }
        if (threadIdx.x == 0) {
            // (the column's place in the frame's miss list is its own: room for every cell of a column; the counter only counts)
            if (total) __hip_atomic_fetch_add(&mlm_gp(P.ctr)->mvox_cnt[3][0], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base[7] = (total && P.record_awareness) ? g_atomic_add(&mlm_gp(P.ctr)->n_miss_list, total) : 0u;
            if (n_rays + s_nouter) g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[sl][0], n_rays + s_nouter); // statistic only
            // device-scope atomics on account of this column: its chunk descriptors (k_bin_sectors) and the four list reservations
            // (the ones something waits for), and fire-and-forget: a bucket-min per hit, a mask bit per tile run, four counters
            g_atomic_add(&mlm_gp(P.ctr)->ray_cnt[sl][1], nch_all + 8u + n_occ + (uint32_t)n_run);
            mlm_gp(P.col_cnt)[phi] = 0; // consumed: clean for the slot's next frame
        }
        __syncthreads();
        const uint32_t m_base = (uint32_t)phi * (uint32_t)(P.nRho * P.nZ);
        uint16_t *s_cells = (uint16_t *)s_tab; // [total] (nZ * nRho * 2 bytes <= the table's, checked by the host)
        if (total && !s_fail) {
            const uint32_t rec_base = s_base[7];
            for (uint32_t it = threadIdx.x; it < 4u * NMISS; it += NT) {
                const uint32_t w = it >> 2, part = (it & 3u) * 8u;
                uint32_t bits = (s_miss[w] >> part) & 0xFFu;
                const uint32_t z = w / (uint32_t)P.RW, rho0 = (w - z * (uint32_t)P.RW) * 32u + part;
                const uint32_t zz = (uint32_t)s_vz[z].x;
                while (bits) {
                    const uint32_t rho = rho0 + (uint32_t)__ffs((int)bits) - 1u;
                    bits &= bits - 1;
                    const uint32_t at = s_rho_off[rho] + atomicAdd(&s_rho_miss[rho], 1u);
                    s_cells[at] = (uint16_t)((uint32_t)s_vr[rho].y * (uint32_t)P.lv_nz + zz);
                    if (P.record_awareness) mlm_gp(P.ml_cell)[rec_base + at] = z * (uint32_t)P.nRhoPhi + (uint32_t)phi * (uint32_t)P.nRho + rho;
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < total; i += NT) mlm_gp(P.mc_list)[m_base + i] = s_cells[i];
        }
        // ONE descriptor per tile run — {first miss cell, count, first hit, count} — in the column's OWN slot of the tile's
        // descriptor table, announced by the column's bit in the tile's column mask (a fire-and-forget atomic): nothing here waits
        // for memory.  k_tile reads the mask, takes the descriptors of the columns it names and clears it.
        if ((int)threadIdx.x < n_run && !s_fail) {
            const uint32_t r = threadIdx.x, rho_a = s_run_rho[r], rho_b = (int)r + 1 < n_run ? s_run_rho[r + 1] : (uint32_t)P.nRho;
            const uint32_t m_first = s_rho_off[rho_a], m_cnt = (rho_b < (uint32_t)P.nRho ? s_rho_off[rho_b] : total) - m_first;
            const uint32_t h_cnt = s_run_hits[r];
            if (m_cnt | h_cnt) {
                const uint32_t tile = s_run_tile[r];
                *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.tile_desc) + 4 * ((size_t)tile * (size_t)P.nPhi + (size_t)phi)) =
                    mlm_u32x4{m_base + m_first, m_cnt, s_base[0] + s_run_off[r], h_cnt};
                __hip_atomic_fetch_or(mlm_gp(P.tile_cols) + (size_t)tile * P.tile_words + ((uint32_t)phi >> 5), 1u << ((uint32_t)phi & 31u), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_fail) mlm_sector_fail(P, F); // (a cell outside the frame-local grid, a tile with too many runs)
    MLM_PHASE(6);
    MLM_PHASE_END
}
template <bool EX, int NT>
// (256 threads: eight workgroups per CU = eight waves per SIMD, 64 VGPRs; 512 threads — tables above 1 024 entries: at most three
// workgroups per CU by LDS, or a frame on its own — six waves per SIMD, 80 VGPRs)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 8 : 6))) void k_sector(MLM_SLOT_ARGS, int tile_w, int n_bin_blocks, unsigned long long rho_m,
                                                                                       int rho_s, unsigned long long n_bkt, int big_armed, unsigned long long row_m, int row_s) {
    // (row_m, row_s: exact division of a pixel index by the image width — the row of a cell's first pixel, mlm_ref_pack)
    MLM_SLOT_SETUP
    MLM_SPAN_BEGIN(0)
    if (blockIdx.x == 0 && threadIdx.x < 64) mlm_fold_bin_stats(P, n_bin_blocks);
    mlm_sector_column<EX, false, NT>(P, F, (int)blockIdx.x, tile_w, n_bin_blocks, rho_m, rho_s, n_bkt, big_armed, row_m, row_s);
    MLM_SPAN_END(0)
}
// The columns on the overflow lists of a batch's frames, with the large cell table (dynamic LDS of MlmDev::sec_big_lds_bytes:
// one workgroup per CU).  ONE launch per batch, a fixed number of workgroups that share all (frame, column) tasks.  A kernel
// that needs most of a CU's LDS waits for the other streams' workgroups to leave, so the host launches it only while the
// scene calls for it (it arms the pass when a frame had to fall back because of a full table, and disarms it after a
// while without overflows); k_sector is told whether the pass follows (big_armed) — if not, a full table still sends the
// frame to the cell-table path.  Scenes where every pixel lands
// in a cell of its own ("scatter") overflow the small table in every column.
template <bool EX>
__global__ __launch_bounds__(MLM_SEC_THREADS) void k_sector_big(const MlmDev *__restrict__ slot_tab, const MlmFrame *__restrict__ frame_tab, int slot_base,
                                                                int n_frames, int tile_w, int n_bin_blocks, unsigned long long rho_m, int rho_s,
                                                                unsigned long long n_bkt, unsigned long long row_m, int row_s) {
    __shared__ unsigned int s_first[65]; // exclusive prefix of the frames' overflow counts (n_frames <= 64)
    if (threadIdx.x < 64) {
        const int j = (int)threadIdx.x;
        unsigned int c = 0;
        if (j < n_frames) c = min(mlm_gp(slot_tab[slot_base + j].ctr)->n_ov, (unsigned int)slot_tab[slot_base + j].nPhi);
        const unsigned int incl = mlm_wave_incl_scan(c);
        s_first[j] = incl - c;
        if (j == 63) s_first[64] = incl;
    }
    __syncthreads();
    const unsigned int n_tasks = s_first[64];
    // (tasks are drawn from a counter, not dealt out by index: a crowded column takes several times as long as a sparse one, and the
    // heavy columns of consecutive frames are the same columns — dealt out by index they would meet in the same workgroups)
    __shared__ unsigned int s_task;
    for (;;) { // (uniform)
        __syncthreads(); // (the previous column's shared state, and s_task, are no longer read)
        if (threadIdx.x == 0) s_task = g_atomic_add(&mlm_gp(slot_tab[slot_base].ctr)->big_next, 1u);
        __syncthreads();
        const unsigned int t = s_task;
        if (t >= n_tasks) break;
        int j = 0;
        while (j + 1 < n_frames && s_first[j + 1] <= t) ++j;
        const MlmDev &P = slot_tab[slot_base + j];
        const MlmFrame &F = frame_tab[slot_base + j];
        mlm_sector_column<EX, true, MLM_SEC_THREADS>(P, F, (int)mlm_gp(P.ov_list)[t - s_first[j]], tile_w, n_bin_blocks, rho_m, rho_s, n_bkt, 0, row_m, row_s);
    }
}

// One wave per multi-kind hit cell: order the cell's contributions by insertion time (= pixel order: a pixel contributes
// to a cell at most once) and store their kinds in that order for k_chain_lanes.  Bitmap ranking as k_sort_contribs, but fed with
// the records' 8x8 lane masks (eight row bytes per (record, kind) reference) instead of one key per contribution.
// tile_w > 0: dense 8x8 pixel tiles of an image of that width; 0: linear work items (see MlmNode).  row_w, div_m, div_s:
// rows of the ranking bitmap and the exact division by row_w.
#ifdef MLM_RANK_WPE // (experiment builds: make alt ALT_FLAGS=-DMLM_RANK_WPE=8)
#define MLM_RANK_ATTR __attribute__((amdgpu_waves_per_eu(MLM_RANK_WPE)))
#else
#define MLM_RANK_ATTR
#endif
// CHAIN (the launches of a frame on its own): the wave that has ranked a cell runs its float chain at once, from the kinds it has just
// put in order, and k_chain_lanes is not launched — a lone frame's cells are a handful per wave, so the chain by one lane per cell
// (a quarter of a microsecond) is cheaper than a kernel boundary plus a kernel that starts from memory again; in a batch, where a wave
// ranks dozens of cells, one lane per cell would idle the other sixty-three and k_chain_lanes' cell-per-lane replay stays.
template <bool CHAIN>
__global__ __launch_bounds__(MLM_BLOCK) MLM_RANK_ATTR void k_rank(MLM_SLOT_ARGS, int tile_w, int row_w, unsigned long long div_m, int div_s, const MlmExOrder om) {
    MLM_SLOT_SETUP
    __shared__ __attribute__((aligned(16))) unsigned long long s_rows[MLM_BLOCK / 64][MLM_SEC_RANK_WORDS];
    __shared__ uint16_t s_pref[MLM_BLOCK / 64][MLM_SEC_RANK_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_kinds[MLM_BLOCK / 64][1024 + 64]; // ordered kinds of cells with n <= 1024 (+ a spare byte per lane)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // (a frame whose Stage A gave up — sector_overflow 2 — is redone from its Stage A on: a column that failed after its reservations has
    // counted its cells in n_multi without writing their descriptors, so nothing of this frame may be ranked from mt_rec / mt_ref)
    const bool gave_up = mlm_gp(P.ctr)->sector_overflow >= 2u; // (also: the counters of such a frame may exceed its lists' capacities)
    const unsigned int n_cells = gave_up ? 0u : mlm_gp(P.ctr)->n_multi;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    if (!P.explore) {
        // Iteration-order keys of the frame's unique hits, valid if the frame fits the emulated container without a rehash
        // (k_apply_tiles checks): (first insertion time of the hit's bucket, its own insertion time) — the bucket-first
        // table of this slot is complete now that every column of the frame has been through k_sector
        const unsigned int n_hit = gave_up ? 0u : mlm_gp(P.ctr)->u_hit;
        for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_hit; i += gridDim.x * blockDim.x) {
            const uint32_t b = mlm_gp(P.hl_bkt)[i];
            const unsigned long long first = b < P.sbkt_cap ? mlm_gp(P.sbkt)[b] & 0xFFFFFFFFull : 0ull;
            mlm_gp(P.hl_key)[i] = ((first + 1ull) << 32) | (unsigned long long)mlm_gp(P.hl_vt)[i];
        }
    } else if (om.on) {
        // frontier mode, a lone frame whose map-dependent launches follow without the host having seen its counts: the bucket-first
        // pass of both containers (what k_ex_order_min does, one launch later otherwise) — fire-and-forget atomics in front of the ranking
        const MLM_GLOBAL MlmCounters *c = mlm_gp(P.ctr);
        const unsigned int n_hit = c->u_hit, n_miss = c->n_ex_miss;
        if (c->sector_overflow == 0u && n_hit <= om.thr_hit && n_miss <= om.thr_miss) { // (else: mlm_ex_spec_skip holds the frame's other launches back)
            const unsigned int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
            for (unsigned int i = i0; i < n_hit; i += stride) {
                int rho, phi, z;
                mlm_cell_rpz(P, P.hl_cell[i], rho, phi, z);
                atomicMin(&P.bkt64[mlm_hash_rpz(rho, phi, z) % om.nb_hit], mlm_bkt_entry(om.tag, P.hl_vt[i]));
            }
            for (unsigned int i = i0; i < n_miss; i += stride)
                atomicMin(&P.bkt64[P.bkt_stride + (unsigned long long)P.ex_cell[i] % om.nb_miss], mlm_bkt_entry(om.tag, P.ex_vt[i]));
        }
    }
    MLM_LDS unsigned long long *rows = mlm_lp(s_rows[wid]);
    volatile MLM_LDS uint16_t *pref = mlm_lp(s_pref[wid]);
    for (int j = lane; j < MLM_SEC_RANK_WORDS; j += 64) rows[j] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const MLM_GLOBAL uint32_t *refs = mlm_gp(P.refs);
    // reference p of a cell = one non-empty row of one contribution group: the row's byte of the group's lane mask, the position
    // of the row's first lane relative to the cell's first pixel, kind (mlm_ref_pack)
    // (a reference travels as its one packed word — the rounds held in registers one pair of cells ahead cost a register each —
    // and is taken apart where it is used; 0 = none: a real reference has a non-empty row byte)
    auto load_ref = [&](const mlm_u32x2 &rf, uint32_t p) -> uint32_t { return p < rf.y ? refs[(size_t)(rf.x + p)] : 0u; };
    // The same for the rounds held in registers one pair ahead: issued UNCONDITIONALLY from a clamped index, the range check is made
    // where the word is used (ref_of).  The compiler counts outstanding loads (s_waitcnt vmcnt(N)) only through straight-line code:
    // behind a per-lane branch it has to wait for everything, i.e. for the prefetches it has just issued.
    auto load_ref_ahead = [&](const mlm_u32x2 &rf, uint32_t p) -> uint32_t { return refs[(size_t)rf.x + (p < rf.y ? p : 0u)]; };
    auto ref_of = [&](uint32_t word, uint32_t p, uint32_t n_refs) -> uint32_t { return p < n_refs ? word : 0u; };
    // (yx: rows below the cell's first pixel << 11 | column of the row's first lane)
    auto unpack = [&](uint32_t ref, uint32_t &bits, uint32_t &yx, uint32_t &sub) {
        uint32_t dy, x;
        mlm_ref_unpack(ref, tile_w > 0, bits, sub, dy, x);
        yx = (dy << 11) | x;
    };
    auto load_pair = [&](const mlm_u32x2 &rf, uint32_t p, uint32_t &bits, uint32_t &yx, uint32_t &sub) { unpack(load_ref(rf, p), bits, yx, sub); };
    // (CHAIN) the float noisy-OR chain of ONE cell by the calling lane (update_odds_hashmap, map_awareness.h:147-154) over its n kinds in
    // pixel order, read with kind_at(j); hit-list place and rho from the cell's descriptor — what k_chain_lanes does for a lane's cell
    // (the cell's 21 possible odds — one per kind, at its rho — are put into LDS by the lanes of the half / wave first: a look-up per step
    // from memory would make a chain of 77 steps 77 trips)
    __shared__ float s_odd[CHAIN ? MLM_BLOCK / 64 : 1][2][32];
    auto chain_odds = [&](const mlm_u32x4 &rec, int h, int l) { // lanes l = 0 .. 31 of the cell's half h (both halves for a whole-wave cell: h = 0)
        const int rho = (int)(rec.z >> MLM_SEC_CNT_BITS), d = (l + 1) >> 1, rho_s = l == 0 ? rho : ((l & 1) ? rho - d : rho + d);
        if (l < 32) s_odd[CHAIN ? wid : 0][h][l] = (l <= 2 * MLM_DIFF_RANGE && rho_s >= 0 && rho_s < P.nRho) ? mlm_gp(P.odds_table)[mlm_contribution_index(P, rho, l)] : 0.0f;
    };
    auto chain_cell = [&](const mlm_u32x4 &rec, int h, uint32_t n, auto &&kind_at) {
        volatile MLM_LDS float *odd = mlm_lp(s_odd[CHAIN ? wid : 0][h]);
        float p = 0.0f;
        for (uint32_t j = 0; j < n && p != 1.0f; ++j) { // (1.0f is absorbing)
            const float a = odd[kind_at(j) & 31u];
            p = j == 0u ? a : 1 - (1 - p) * (1 - a);
        }
        const uint32_t pos = rec.x;
        if (P.record_awareness || P.explore) mlm_gp(P.hl_odd)[pos] = p;
        mlm_gp(P.hl_inc)[pos] = mlm_logit(P, p);
    };
    auto process = [&](const mlm_u32x4 &rec, const mlm_u32x2 &rf, const uint32_t (&r_raw)[4]) {
        uint32_t r_bits[4], r_yx[4], r_sub[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack(ref_of(r_raw[q], (uint32_t)lane + 64u * q, rf.y), r_bits[q], r_yx[q], r_sub[q]);
        // (the descriptor is the same in every lane: scalar registers, uniform branches)
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.y), n = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.z) & MLM_SEC_CNT_MASK,
                       n_refs = (uint32_t)__builtin_amdgcn_readfirstlane((int)rf.y);
        const uint32_t pix0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec.w) / MLM_TIME_SLOTS; // the cell's first work item: smallest row of the window
        const uint32_t y0 = (uint32_t)(((unsigned long long)pix0 * div_m) >> div_s);
        // (left edge of the 128-column window, multiple of 8: a record's row byte never straddles a word; dense images: the references' columns
        // are relative to the first pixel's tile column, MLM_REF_XREL0)
        const int xlo = tile_w > 0 ? (int)(MLM_REF_XREL0 << 3) - 64 : (int)((pix0 - y0 * (uint32_t)row_w) & ~7u) - 64;
        MLM_GLOBAL uint8_t *S = mlm_gp(P.subs) + soff;
        volatile MLM_LDS uint8_t *SL = mlm_lp(s_kinds[wid]);
        const bool staged = n <= 1024u; // kinds are collected in LDS and leave as whole dwords
        const int rounds = (int)min(4u, (n_refs + 63u) >> 6); // (uniform) rounds of 64 pairs held in registers
        bool bad = n > 0xFFFFu;
        auto locate = [&](uint32_t yx, uint32_t &wi, uint32_t &sh) -> bool {
            const int dx = (int)(yx & 2047u) - xlo;
            const uint32_t dy = yx >> 11;
            wi = 2 * dy + ((uint32_t)dx >> 6);
            sh = (uint32_t)dx & 63u;
            return dx >= 0 && dx <= 120 && dy < MLM_BMP_ROWS;
        };
        uint32_t l_wi[4], l_sh[4];
        if (!bad) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                l_wi[q] = 0;
                l_sh[q] = 0;
                if (q < rounds && r_bits[q]) {
                    if (locate(r_yx[q], l_wi[q], l_sh[q])) __hip_atomic_fetch_or(&rows[l_wi[q]], (unsigned long long)r_bits[q] << l_sh[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else bad = true;
                }
            }
            for (uint32_t p = lane + 256u; p < n_refs; p += 64) {
                uint32_t b, px, sb, wi, sh;
                load_pair(rf, p, b, px, sb);
                if (b) {
                    if (locate(px, wi, sh)) __hip_atomic_fetch_or(&rows[wi], (unsigned long long)b << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else bad = true;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bad = __any(bad);
        uint32_t carry = 0;
        int used = MLM_SEC_RANK_WORDS;
        if (!bad) {
            int j0 = 0;
            for (; j0 < MLM_SEC_RANK_WORDS && carry < n; j0 += 64) {
                const uint32_t cw = (uint32_t)__popcll(((volatile MLM_LDS unsigned long long *)rows)[j0 + lane]);
                const uint32_t incl = mlm_wave_incl_scan(cw);
                pref[j0 + lane] = (uint16_t)(carry + incl - cw);
                carry += mlm_readlane(incl, 63);
            }
            if (carry == n) used = j0;
            else bad = true; // (two contributions on one pixel cannot happen; a count mismatch means a window miss)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!bad) {
            // a row's own set bits, in pixel order (other records' bits may lie between them): rank of bit b = contributions
            // before the segment + set bits of the segment below b
            auto place = [&](uint32_t bits, uint32_t wi, uint32_t sh, uint32_t sub, auto &&store) {
                const unsigned long long word = ((volatile MLM_LDS unsigned long long *)rows)[wi];
                const uint32_t before = pref[wi] + (uint32_t)__popcll(word & ((1ull << sh) - 1ull));
                const uint32_t seg = (uint32_t)(word >> sh) & 0xFFu; // every record's bits of this 8-pixel segment
                store(bits, before, seg, sub);
            };
            // staged (n <= 1024, the usual case): by bit position, eight straight-line steps without branches — a lane whose
            // bit is clear writes to a spare byte of its own behind the staged kinds
            auto store_lds = [&](uint32_t bits, uint32_t before, uint32_t seg, uint32_t sub) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const uint32_t at = before + (uint32_t)__popc(seg & ((1u << b) - 1u));
                    SL[(bits >> b) & 1u ? at : 1024u + (uint32_t)lane] = (uint8_t)sub;
                }
            };
            auto store_glb = [&](uint32_t bits, uint32_t before, uint32_t seg, uint32_t sub) {
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    S[before + (uint32_t)__popc(seg & ((1u << b) - 1u))] = (uint8_t)sub;
                }
            };
            if (staged) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < rounds) place(r_bits[q], l_wi[q], l_sh[q], r_sub[q], store_lds);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < rounds) place(r_bits[q], l_wi[q], l_sh[q], r_sub[q], store_glb);
            }
            for (uint32_t p = lane + 256u; p < n_refs; p += 64) {
                uint32_t b, px, sb, wi, sh;
                load_pair(rf, p, b, px, sb);
                if (b) {
                    locate(px, wi, sh);
                    if (staged) place(b, wi, sh, sb, store_lds);
                    else place(b, wi, sh, sb, store_glb);
                }
            }
            if (staged) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (CHAIN) {
                    chain_odds(rec, 0, lane);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) chain_cell(rec, 0, n, [&](uint32_t j) -> uint32_t { return SL[j]; });
                } else {
                    for (uint32_t j = lane; j < (n + 3u) >> 2; j += 64) // (segments are padded to 16 bytes)
                        ((MLM_GLOBAL uint32_t *)S)[j] = ((volatile MLM_LDS uint32_t *)SL)[j];
                }
            } else if (CHAIN) { // (more than 1 024 contributions: the kinds went straight to memory)
                chain_odds(rec, 0, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) chain_cell(rec, 0, n, [&](uint32_t j) -> uint32_t { return __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); });
            }
        } else {
            // slow exact path (a contribution outside the bitmap window, or a huge cell): write every contribution's
            // position into the cell's segment of `contrib`, then rank by counting straight from memory
            MLM_GLOBAL uint32_t *K = mlm_gp(P.contrib) + soff;
            uint32_t base = 0;
            for (uint32_t p0 = 0; p0 < n_refs; p0 += 64) {
                uint32_t b, px, sb;
                load_pair(rf, p0 + lane, b, px, sb);
                // position of the row's first lane, counted from the first row of the cell (dense images: columns relative to the first
                // pixel's tile column, below 2 048 — any row stride above that keeps the order)
                const uint32_t item = (px >> 11) * (tile_w > 0 ? 4096u : (uint32_t)row_w) + (px & 2047u);
                const uint32_t cb = (uint32_t)__popc(b);
                const uint32_t incl = mlm_wave_incl_scan(cb);
                uint32_t at = base + incl - cb;
                while (b) {
                    const int bit = __ffs((int)b) - 1;
                    b &= b - 1;
                    K[at++] = ((item + (uint32_t)bit) << 5) | sb;
                }
                base += mlm_readlane(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < n; j += 64) {
                const uint32_t my = __hip_atomic_load(&K[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t r = 0;
                for (uint32_t q = 0; q < n; ++q) r += __hip_atomic_load(&K[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < my;
                S[r] = (uint8_t)(my & 31u);
            }
            if (CHAIN) {
                chain_odds(rec, 0, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) chain_cell(rec, 0, n, [&](uint32_t j) -> uint32_t { return __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); });
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < used; j += 64) rows[j] = 0ull; // clean for the next cell
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // ---- TWO cells per wave, one per half: a cell's work is counted in wave-instructions (about 160 with one round of
    //      references), not in lanes, and with the empty rows gone most cells fill half a wave at best.  Each half has its own
    //      64-row half of the bitmap, its own half of the staging buffer and its own per-cell values in vector registers; a cell
    //      that does not fit half a wave (more than 512 contributions, taller than 64 rows, a window miss) is redone by the whole
    //      wave with `process`.
    const int half = lane >> 5, hl = lane & 31;
    MLM_LDS unsigned long long *rows_h = rows + half * (MLM_SEC_RANK_WORDS / 2);
    volatile MLM_LDS uint16_t *pref_h = pref + half * (MLM_SEC_RANK_WORDS / 2);
    volatile MLM_LDS uint8_t *SL_h = mlm_lp(s_kinds[wid]) + half * 544; // 512 staged kinds + a spare byte per lane
    auto half_incl_scan = [](uint32_t v) { // inclusive prefix sum inside each half of the wave (mlm_wave_incl_scan without its last step)
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
        return (uint32_t)x;
    };
    // returns (per lane, equal inside a half): the half's cell still has to be done by the whole wave
    auto process_pair = [&](const mlm_u32x4 &rec, const mlm_u32x2 &rf, bool valid, const uint32_t (&r_raw)[4]) -> bool {
        uint32_t r_ref[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) r_ref[q] = ref_of(r_raw[q], (uint32_t)hl + 32u * q, valid ? rf.y : 0u);
        const uint32_t soff = rec.y, n = rec.z & MLM_SEC_CNT_MASK, n_refs = rf.y;
        const uint32_t pix0 = rec.w / MLM_TIME_SLOTS;
        const uint32_t y0 = (uint32_t)(((unsigned long long)pix0 * div_m) >> div_s);
        const int xlo = tile_w > 0 ? (int)(MLM_REF_XREL0 << 3) - 64 : (int)((pix0 - y0 * (uint32_t)row_w) & ~7u) - 64;
        const bool act = valid && n <= 512u;
#ifdef MLM_PHASE_PROF // (diagnostic build: how many contributions the ranked cells have — tools/rank_hist.py)
        if (valid && hl == 0) {
            const uint32_t bkt = n <= 3u ? 0u : (n <= 4u ? 1u : (n <= 8u ? 2u : (n <= 16u ? 3u : (n <= 32u ? 4u : (n <= 64u ? 5u : (n <= 128u ? 6u : (n <= 256u ? 7u : 8u)))))));
            atomicAdd(&g_mlm_span[9 + 2 * bkt], 1ull); // (odd slots: the multiples of four hold the spans' first-start minima)
            atomicAdd(&g_mlm_span[27], (unsigned long long)n_refs);
        }
        {   // (would the cell fit a bitmap of ONE 64-pixel word per row — segments -3 .. +4 around its first pixel's —, and 32 rows?)
            bool out64 = false, row32 = false;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r_ref[q] & 0xFFu) {
                    uint32_t b_, yx_, s_;
                    unpack(r_ref[q], b_, yx_, s_);
                    const int dxs = (int)(yx_ & 2047u) - (xlo + 40);
                    out64 = out64 || dxs < 0 || dxs > 56;
                    row32 = row32 || (yx_ >> 11) >= 32u;
                }
            const unsigned long long o_l = __ballot(valid && out64), r_l = __ballot(valid && row32);
            if (valid && hl == 0) {
                if ((uint32_t)(o_l >> (32 * half))) atomicAdd(&g_mlm_span[29], 1ull);
                if ((uint32_t)(r_l >> (32 * half))) atomicAdd(&g_mlm_span[31], 1ull);
            }
        }
#endif
        const int my_rounds = act ? (int)min(4u, (n_refs + 31u) >> 5) : 0;
        const int rounds = max(mlm_readlane(my_rounds, 0), mlm_readlane(my_rounds, 32)); // (uniform)
        bool bad = false;
        auto locate = [&](uint32_t yx, uint32_t &wi, uint32_t &sh) -> bool {
            const int dx = (int)(yx & 2047u) - xlo;
            const uint32_t dy = yx >> 11;
            wi = 2 * dy + ((uint32_t)dx >> 6);
            sh = (uint32_t)dx & 63u;
            return dx >= 0 && dx <= 120 && dy < MLM_BMP_ROWS / 2;
        };
        uint32_t l_ws[4]; // bitmap word << 6 | shift of each held reference (the packed reference itself is taken apart where it is used)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            l_ws[q] = 0;
            if (q < rounds && q < my_rounds && (r_ref[q] & 0xFFu)) {
                uint32_t bits, yx, sub, wi, sh;
                unpack(r_ref[q], bits, yx, sub);
                if (locate(yx, wi, sh)) {
                    __hip_atomic_fetch_or(&rows_h[wi], (unsigned long long)bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    l_ws[q] = (wi << 6) | sh;
                } else {
                    bad = true;
                }
            }
        }
        if (act)
            for (uint32_t p = (uint32_t)hl + 128u; p < n_refs; p += 32) {
                uint32_t b, px, sb, wi, sh;
                load_pair(rf, p, b, px, sb);
                if (b) {
                    if (locate(px, wi, sh)) __hip_atomic_fetch_or(&rows_h[wi], (unsigned long long)b << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else bad = true;
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long bad_lanes = __ballot(bad);
        bool ok = act && ((uint32_t)(bad_lanes >> (32 * half)) == 0u);
        uint32_t carry = 0, used = MLM_SEC_RANK_WORDS / 2;
        for (uint32_t j0 = 0; j0 < MLM_SEC_RANK_WORDS / 2; j0 += 32) {
            const bool need = ok && carry < n;
            if (!__any(need)) break;
            const uint32_t cw = need ? (uint32_t)__popcll(((volatile MLM_LDS unsigned long long *)rows_h)[j0 + hl]) : 0u;
            const uint32_t incl = half_incl_scan(cw);
            const uint32_t t0 = mlm_readlane(incl, 31), t1 = mlm_readlane(incl, 63);
            if (need) {
                pref_h[j0 + hl] = (uint16_t)(carry + incl - cw);
                carry += half ? t1 : t0;
                used = j0 + 32;
            }
        }
        ok = ok && carry == n; // (a count mismatch means a window miss: the whole wave redoes the cell)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto place = [&](uint32_t bits, uint32_t wi, uint32_t sh, uint32_t sub) {
            const unsigned long long word = ((volatile MLM_LDS unsigned long long *)rows_h)[wi];
            const uint32_t before = pref_h[wi] + (uint32_t)__popcll(word & ((1ull << sh) - 1ull));
            const uint32_t seg = (uint32_t)(word >> sh) & 0xFFu;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t at = before + (uint32_t)__popc(seg & ((1u << b) - 1u));
                SL_h[(bits >> b) & 1u ? at : 512u + (uint32_t)hl] = (uint8_t)sub;
            }
        };
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < rounds && ok && q < my_rounds) place(r_ref[q] & 0xFFu, l_ws[q] >> 6, l_ws[q] & 63u, (r_ref[q] >> 8) & 31u);
        if (ok)
            for (uint32_t p = (uint32_t)hl + 128u; p < n_refs; p += 32) {
                uint32_t b, px, sb, wi, sh;
                load_pair(rf, p, b, px, sb);
                if (b) {
                    locate(px, wi, sh);
                    place(b, wi, sh, sb);
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (CHAIN) { // (the staged kinds are complete: the barrier above)
            if (ok) chain_odds(rec, half, hl);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ok && hl == 0) chain_cell(rec, half, n, [&](uint32_t j) -> uint32_t { return SL_h[j]; });
        } else if (ok) {
            MLM_GLOBAL uint32_t *S32 = (MLM_GLOBAL uint32_t *)(mlm_gp(P.subs) + soff);
            for (uint32_t j = (uint32_t)hl; j < (n + 3u) >> 2; j += 32) S32[j] = ((volatile MLM_LDS uint32_t *)SL_h)[j]; // (segments are padded to 16 bytes)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (act)
            for (uint32_t j = (uint32_t)hl; j < (ok ? used : (uint32_t)(MLM_SEC_RANK_WORDS / 2)); j += 32) rows_h[j] = 0ull; // clean for the next cell
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return valid && !ok;
    };
    // one pair of cells ahead: descriptors two pairs ahead, the first four rounds of references one pair ahead — a cell is
    // otherwise a chain of four dependent memory round trips (descriptor, references, records, store).  Every load of the loop is
    // issued unconditionally (clamped indices; validity travels beside the data): what was requested while the previous pair was
    // worked on is waited for ONCE at the top of the loop, nothing is waited for in the middle of a pair.
    const unsigned int n_pairs = (n_cells + 1u) >> 1;
    if (wave >= n_pairs) return; // (uniform; n_cells >= 1 from here on)
    auto load_desc = [&](unsigned int pw, mlm_u32x4 &rec, mlm_u32x2 &rf) { // the half's cell of pair pw (clamped to the last cell)
        const unsigned int c = min(2u * min(pw, n_pairs - 1u) + (unsigned int)half, n_cells - 1u);
        rec = *(const MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.mt_rec) + c);
        rf = *(const MLM_GLOBAL mlm_u32x2 *)(mlm_gp(P.mt_ref) + 2 * (size_t)c);
    };
    auto desc_valid = [&](unsigned int pw) { return pw < n_pairs && 2u * pw + (unsigned int)half < n_cells; };
    mlm_u32x4 rec_cur, rec_nxt;
    mlm_u32x2 rf_cur, rf_nxt;
    uint32_t r_cur[4], r_nxt[4];
    bool v_cur = desc_valid(wave), v_nxt = desc_valid(wave + n_waves);
    load_desc(wave, rec_cur, rf_cur);
    load_desc(wave + n_waves, rec_nxt, rf_nxt);
    if (!v_cur) rf_cur.y = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) r_cur[q] = load_ref_ahead(rf_cur, (uint32_t)hl + 32u * q);
    for (unsigned int pw = wave; pw < n_pairs; pw += n_waves) {
        mlm_u32x4 rec_nn;
        mlm_u32x2 rf_nn;
        if (!v_nxt) rf_nxt.y = 0u; // (rf_nxt was requested a pair ago: this is the loop's one wait for memory)
#pragma unroll
        for (int q = 0; q < 4; ++q) r_nxt[q] = load_ref_ahead(rf_nxt, (uint32_t)hl + 32u * q);
        load_desc(pw + 2 * n_waves, rec_nn, rf_nn);
        const bool v_nn = desc_valid(pw + 2 * n_waves);
        const bool again = process_pair(rec_cur, rf_cur, v_cur, r_cur);
        const unsigned long long again_lanes = __ballot(again);
        for (int h = 0; h < 2; ++h) // (uniform) the whole wave on a cell that did not fit half of it
            if ((again_lanes >> (32 * h)) & 1ull) {
                mlm_u32x4 rec_f;
                mlm_u32x2 rf_f;
                rec_f.x = mlm_readlane(rec_cur.x, 32 * h);
                rec_f.y = mlm_readlane(rec_cur.y, 32 * h);
                rec_f.z = mlm_readlane(rec_cur.z, 32 * h);
                rec_f.w = mlm_readlane(rec_cur.w, 32 * h);
                rf_f.x = mlm_readlane(rf_cur.x, 32 * h);
                rf_f.y = mlm_readlane(rf_cur.y, 32 * h);
                uint32_t r_f[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r_f[q] = load_ref(rf_f, (uint32_t)lane + 64u * q);
                process(rec_f, rf_f, r_f);
            }
        rec_cur = rec_nxt;
        rf_cur = rf_nxt;
        rec_nxt = rec_nn;
        rf_nxt = rf_nn;
        v_cur = v_nxt;
        v_nxt = v_nn;
#pragma unroll
        for (int q = 0; q < 4; ++q) r_cur[q] = r_nxt[q];
    }
}

// The float noisy-OR chains of the ranked cells (update_odds_hashmap, map_awareness.h:147-154, over the kinds k_rank put in
// pixel order): one cell per LANE, and a lane that finishes its cell draws the next one — chains are 2 to several hundred
// steps long, so with a fixed cell per lane a wave would run as long as its longest chain with most lanes idle.  A wave
// reserves cells from the frame's counter `reserve` at a time (128 in a batch; 64 — one per lane — for a frame on its own, whose
// 8 000 cells would otherwise keep 66 waves busy with two cells per lane while the rest of the grid has none: a lone frame's
// k_chain_lanes lasts as long as a lane's cells do); loads are issued one round ahead of their use (cell descriptor,
// then 16 kinds per 16-byte load), so a round's arithmetic covers the next round's memory latency.  The odds table is kept
// transposed in LDS ([rho][kind], 32 kinds per row): one shift-add per lookup.  p == 1.0f is absorbing and ends a chain.
// R: 16-byte loads of kinds a lane keeps in flight per round.  1 in a batch (most chains are a handful of steps: the second load would be
// wasted on nearly every cell, and the pipeline hides the rounds); 4 for a frame on its own, whose k_chain_lanes lasts as long as
// its LONGEST chain has rounds — several hundred kinds are five rounds of 64 instead of twenty of 16, a microsecond each.
template <int R>
__global__ __launch_bounds__(MLM_BLOCK) void k_chain_lanes(MLM_SLOT_ARGS, unsigned int reserve) {
    MLM_SLOT_SETUP
    extern __shared__ float s_odds_t[]; // [nRho][32]: odd of a contribution of kind k into a cell at rho
    for (int j = threadIdx.x; j < P.nRho * 32; j += blockDim.x) {
        const int rho = j >> 5, k = j & 31;
        // (kind k of a contribution into a cell at rho comes from a centre d = (k + 1) / 2 cells nearer (odd k) or farther;
        // pairs whose centre would lie outside the map cannot occur)
        const int d = (k + 1) >> 1, rho_s = k == 0 ? rho : ((k & 1) ? rho - d : rho + d);
        s_odds_t[j] = (k <= 2 * MLM_DIFF_RANGE && rho_s >= 0 && rho_s < P.nRho) ? mlm_contribution_odd(P, P.odds_table, rho, k) : 0.0f;
    }
    __syncthreads();
    const unsigned int n_cells = mlm_gp(P.ctr)->sector_overflow >= 2u ? 0u : mlm_gp(P.ctr)->n_multi; // (see k_rank)
    if (n_cells == 0u) return; // (uniform)
    const int lane = threadIdx.x & 63;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    uint32_t loc_next = 0, loc_end = 0; // (uniform) the wave's reserved cells not handed to a lane yet
    bool exhausted = false;             // (uniform) the frame's counter is past the last cell
    int state = 0;                      // 0 idle, 1 descriptor in flight, 2 first kinds in flight, 3 running
    mlm_u32x4 desc_in = mlm_u32x4{0u, 0u, 0u, 0u}, kinds_in[R], kinds[R];
#pragma unroll
    for (int r = 0; r < R; ++r) kinds_in[r] = kinds[r] = desc_in;
    uint32_t pos = 0, base = 0, n = 0, j0 = 0, row = 0;
    float p = 0.0f;
    bool first = true;
    // a finished chain waits in its lane until enough lanes have one: the logit (an FP64 log10) then runs once for all of them
    bool pend = false;
    uint32_t pend_pos = 0;
    float pend_p = 0.0f;
    auto flush = [&]() {
        if (pend) {
            if (P.record_awareness || P.explore) mlm_gp(P.hl_odd)[pend_pos] = pend_p;
            mlm_gp(P.hl_inc)[pend_pos] = mlm_logit(P, pend_p);
            pend = false;
        }
    };
    for (;;) {
        // ---- what was requested in the previous round has arrived
        if (state == 2 || state == 3) {
#pragma unroll
            for (int r = 0; r < R; ++r) kinds[r] = kinds_in[r];
        }
        if (state == 2) state = 3;
        // (both loads of a round — the next sixteen kinds, the next descriptor — are issued by EVERY lane, from a harmless address where
        // the lane needs nothing: the compiler counts outstanding loads only through straight-line code, behind a per-lane branch it
        // waits for them on the spot instead of at the top of the next round)
        uint32_t k_at = 0u, k_left = 0u; // offset in `subs` of the kinds this lane requests this round, and how many of them are the cell's
        if (state == 1) {
            pos = desc_in.x;
            base = desc_in.y;
            n = desc_in.z & MLM_SEC_CNT_MASK;
            row = (desc_in.z >> MLM_SEC_CNT_BITS) * 32u;
            j0 = 0;
            first = true;
            p = 0.0f;
            k_at = base;
            k_left = n;
            state = 2;
        } else if (state == 3 && j0 + 16u * R < n) {
            k_at = base + j0 + 16u * R; // (one round ahead)
            k_left = n - (j0 + 16u * R);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) // (a cell's segment of `subs` is padded to whole 16-byte words; past it: the harmless address)
            kinds_in[r] = *(const MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.subs) + ((r == 0 || 16u * r < k_left) ? k_at + 16u * r : 0u));
        // ---- idle lanes draw cells
        const unsigned long long need = __ballot(state == 0);
        uint32_t d_at = 0u; // the descriptor this lane requests this round
        if (need && !exhausted) {
            if (loc_next >= loc_end) {
                uint32_t b = 0;
                if (lane == 0) b = g_atomic_add(&mlm_gp(P.ctr)->chain_next, reserve);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                loc_next = b;
                loc_end = min(b + reserve, n_cells);
                exhausted = b >= n_cells;
            }
            if (!exhausted) {
                const uint32_t idx = loc_next + (uint32_t)__popcll(need & lanes_below);
                if (state == 0 && idx < loc_end) {
                    d_at = idx;
                    state = 1;
                }
                loc_next = min(loc_end, loc_next + (uint32_t)__popcll(need));
            }
        }
        {   // (three of the descriptor's four words: a destination register the lane never reads is reused by the compiler at once,
            // and a write to a register with a load in flight waits for the load)
            const mlm_u32x3 d3 = *(const MLM_GLOBAL mlm_u32x3 *)(mlm_gp(P.mt_rec) + d_at);
            desc_in = mlm_u32x4{d3.x, d3.y, d3.z, 0u};
        }
        if (exhausted && !__any(state != 0)) break;
        if (__popcll(__ballot(pend)) >= 40) flush();
        // ---- sixteen steps of the running chains
        if (state == 3) {
            const uint32_t rem = n - j0; // >= 1
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r > 0 && 16u * r >= rem) break; // (per lane: the later words of a round are a long chain's)
                const uint32_t w[4] = {kinds[r].x, kinds[r].y, kinds[r].z, kinds[r].w};
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float a = s_odds_t[row + ((w[q >> 2] >> ((q & 3) * 8)) & 31u)]; // (padding bytes: any value, not used)
                    if ((uint32_t)(16 * r + q) < rem) {
                        p = first ? a : 1 - (1 - p) * (1 - a);
                        first = false;
                    }
                }
            }
            j0 += 16u * R;
        }
        const bool fin = state == 3 && (j0 >= n || p == 1.0f);
        if (__any(fin && pend)) flush();
        if (fin) {
            pend = true;
            pend_pos = pos;
            pend_p = p;
            state = 0;
        }
    }
    flush();
}

// floor(a / n) for n > 0
__device__ __forceinline__ int mlm_floor_div(int a, int n) {
    int q = a / n;
    if ((a % n) < 0) --q;
    return q;
}

// Group one tile's share of a frame by voxel (map independent, batched over the frames of a batch: blockIdx.z = frame slot,
// blockIdx.x = tile).  The tile's descriptors say where the columns left its miss cells and hits; the workgroup counts them
// per voxel in LDS (the tile's voxels are a few thousand 32-bit counters: misses in the low half, hits in the high half),
// compacts the touched voxels and writes ONE 32-byte record per voxel — block key, cell id, miss count, hit count, the
// increment of a single hit, the pool slot of the block if it exists already — into the frame's record list; the hits of
// voxels with several go next to each other into vr_hit with their iteration-order keys.  The kernel that applies the frame
// then needs two round trips per voxel: record -> (log-odds, class, hits).  Global atomics: two list reservations per tile.
#ifndef MLM_TILE_THREADS
#define MLM_TILE_THREADS 256
#endif
#define MLM_TILE_KEEP 2     // hits per thread kept in registers between the counting and the placing pass
#ifndef MLM_TILE_DESC
#define MLM_TILE_DESC 128
#endif
// ^ descriptors staged per pass (a tile of a VGA frame has ~15; 128 instead of 256: 3 KB of LDS, a fifth workgroup per CU)
#define MLM_TILE_WORDS 64    // most words of a tile's column mask: MlmDev::tile_words = ceil(nPhi / 32) (sector path: nPhi <= 2048)
#define MLM_TILE_COMBOS 2048 // most blocks a tile may overlap: limit of MlmDev::tile_combos (their pool slots are kept in LDS)
struct MlmTileLds {
    uint32_t cnt, place, desc, slot, ztab, total;
};
__host__ __device__ inline MlmTileLds mlm_tile_lds(uint32_t n_vox, uint32_t lv_nz, uint32_t combos) {
    MlmTileLds L;
    uint32_t o = 0;
    L.cnt = o;      o += n_vox * 4u;                 // per voxel: misses | hits << 16 (the hit half doubles as the fill cursor later)
    L.place = o;    o += n_vox * 4u;                 // per touched voxel: record index in the tile | (offset of its hits, 0xFFFF: one hit) << 16
    o = (o + 15u) & ~15u;
    L.desc = o;     o += MLM_TILE_DESC * 16u + MLM_TILE_DESC * 8u; // staged descriptors + exclusive prefixes (miss cells, hits)
    L.slot = o;     o += combos * 4u + combos;           // pool slot per overlapped block + "the frame touches it" flags (combos: multiple of 4)
    o = (o + 3u) & ~3u;
    L.ztab = o;     o += ((lv_nz + 1u) & ~1u) * 4u;  // per grid z: block index << 8 ... (gz, cz) packed
    L.total = (o + 15u) & ~15u;
    return L;
}
// colw: the tile's column mask (MlmDev::tile_words words, in LDS): bit phi = column phi left a descriptor in its slot
__device__ __forceinline__ void mlm_tile_one(const MlmDev &P, const MlmFrame &F, const unsigned int tile, const uint32_t *colw) {
    const MLM_GLOBAL mlm_u32x4 *descs = (const MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.tile_desc) + 4 * (size_t)tile * (size_t)P.nPhi);
    __shared__ uint32_t s_cpre[MLM_TILE_WORDS + 1]; // exclusive prefix of the mask words' popcounts
    if (threadIdx.x < 64) { // (one wave: MlmDev::tile_words <= 64)
        const uint32_t w = threadIdx.x < P.tile_words ? colw[threadIdx.x] : 0u;
        const uint32_t c = (uint32_t)__popc(w), incl = mlm_wave_incl_scan(c);
        if (threadIdx.x < P.tile_words) s_cpre[threadIdx.x] = incl - c;
        if (threadIdx.x == 63) s_cpre[P.tile_words] = incl;
        // consumed: clean for the slot's next frame (the words were read into LDS by the caller)
        if (threadIdx.x < P.tile_words && w) mlm_gp(P.tile_cols)[(size_t)tile * P.tile_words + threadIdx.x] = 0u;
    }
    __syncthreads();
    const unsigned int nd_all = s_cpre[P.tile_words];
    // the j-th column (in ascending phi) that left a descriptor
    auto nth_column = [&](uint32_t j) -> uint32_t {
        uint32_t lo = 0, hi = P.tile_words; // largest w with s_cpre[w] <= j
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cpre[mid] <= j) lo = mid;
            else hi = mid;
        }
        uint32_t bits = colw[lo];
        for (uint32_t k = j - s_cpre[lo]; k; --k) bits &= bits - 1;
        return lo * 32u + (uint32_t)__ffs((int)bits) - 1u;
    };
    mlm_u32x4 dd_first = mlm_u32x4{0u, 0u, 0u, 0u};
    if (threadIdx.x < min((unsigned int)MLM_TILE_DESC, nd_all)) dd_first = descs[nth_column(threadIdx.x)];
    MLM_TPHASE_BEGIN
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t edge = 1u << P.tile_sh, NV = edge * edge * (uint32_t)P.lv_nz;
    const MlmTileLds L = mlm_tile_lds(NV, (uint32_t)P.lv_nz, P.tile_combos);
    uint32_t *s_cnt = (uint32_t *)(s_dyn + L.cnt), *s_place = (uint32_t *)(s_dyn + L.place);
    mlm_u32x4 *s_desc = (mlm_u32x4 *)(s_dyn + L.desc);
    uint32_t *s_dm = (uint32_t *)(s_desc + MLM_TILE_DESC), *s_dh = s_dm + MLM_TILE_DESC;
    int *s_slot = (int *)(s_dyn + L.slot);
    uint8_t *s_touch = (uint8_t *)(s_slot + P.tile_combos);
    uint32_t *s_ztab = (uint32_t *)(s_dyn + L.ztab);
    __shared__ uint32_t s_w[4 * MLM_SEC_WAVES];
    __shared__ uint32_t s_base[4];
    __shared__ unsigned int s_fail, s_dead;
    __shared__ uint32_t s_xytab[16]; // per x (first `edge` entries) and per y: block index relative to the tile's first | cell coordinate << 16
    if (threadIdx.x == 0) {
        s_dead = mlm_gp(P.ctr)->sector_overflow >= 2u; // set by Stage A: the frame is redone on the cell-table path — only clean up
        s_fail = 0u;
    }
    for (uint32_t v = threadIdx.x; v < NV; v += MLM_TILE_THREADS) s_cnt[v] = 0u;
    // the tile's place in the world: grid coordinates of its corner, the blocks it overlaps
    const int ty = (int)(tile / (unsigned int)P.n_tx), tx = (int)tile - ty * P.n_tx;
    const int X0 = (tx << P.tile_sh) + F.lv_o[0], Y0 = (ty << P.tile_sh) + F.lv_o[1], Z0 = F.lv_o[2];
    const int gx0 = mlm_floor_div(X0, P.n), gy0 = mlm_floor_div(Y0, P.n), gz0 = mlm_floor_div(Z0, P.n);
    const int ngx = mlm_floor_div(X0 + (int)edge - 1, P.n) - gx0 + 1, ngy = mlm_floor_div(Y0 + (int)edge - 1, P.n) - gy0 + 1,
              ngz = mlm_floor_div(Z0 + P.lv_nz - 1, P.n) - gz0 + 1;
    const int n_combo = ngx * ngy * ngz; // (<= P.tile_combos: the host's bound for the grid's geometry)
    const bool probe = true;
    for (int c = threadIdx.x; c < n_combo; c += MLM_TILE_THREADS) s_touch[c] = 0;
    auto combo_slot = [&](int c) {
        const int bz = c / (ngx * ngy), r = c - bz * ngx * ngy, by = r / ngx, bx = r - by * ngx;
        return mlm_block_find(P, gx0 + bx, gy0 + by, gz0 + bz);
    };
    for (int z = threadIdx.x; z < P.lv_nz; z += MLM_TILE_THREADS) {
        const int Z = Z0 + z, gz = mlm_floor_div(Z, P.n);
        s_ztab[z] = (uint32_t)(gz - gz0) | ((uint32_t)(Z - gz * P.n) << 16);
    }
    if (threadIdx.x < 2u * edge) { // the same per x and per y of the tile: the loops over the voxels below divide nothing
        const bool is_y = threadIdx.x >= edge;
        const int C = (is_y ? Y0 : X0) + (int)(threadIdx.x & (edge - 1u)), g = mlm_floor_div(C, P.n);
        s_xytab[threadIdx.x] = (uint32_t)(g - (is_y ? gy0 : gx0)) | ((uint32_t)(C - g * P.n) << 16);
    }
    __syncthreads();
    MLM_TPHASE(0); // parameters, column mask / first descriptors, LDS clear
    if (s_dead) return;
    // the pool slots of the tile's blocks (in flight while the cells are counted; a tile overlaps a few dozen blocks)
    int slot0 = -1;
    if (probe) {
        if ((int)threadIdx.x < n_combo) slot0 = combo_slot((int)threadIdx.x);
        for (int c = threadIdx.x + MLM_TILE_THREADS; c < n_combo; c += MLM_TILE_THREADS) s_slot[c] = combo_slot(c);
    }
    const unsigned int nd = nd_all;
    // flat item j of the staged descriptors -> (descriptor, offset inside it); pre[] = exclusive prefix of the counts
    auto locate = [&](const uint32_t *pre, uint32_t n_staged, uint32_t j, uint32_t &d, uint32_t &o) {
        uint32_t lo = 0, hi = n_staged;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pre[mid] <= j) lo = mid;
            else hi = mid;
        }
        d = lo;
        o = j - pre[lo];
    };
    // pass over the descriptors; what == 0: count the miss cells and hits per voxel; what == 1: place the hits
    uint32_t rec_base = 0, hit_base = 0;
    uint32_t keep_pos[MLM_TILE_KEEP], keep_v[MLM_TILE_KEEP];
    float keep_inc[MLM_TILE_KEEP];
    unsigned long long keep_key[MLM_TILE_KEEP];
#pragma unroll
    for (int q = 0; q < MLM_TILE_KEEP; ++q) {
        keep_pos[q] = 0u;
        keep_v[q] = MLM_NIL;
        keep_inc[q] = 0.0f;
        keep_key[q] = 0ull;
    }
    auto for_cells = [&](int what) {
        for (uint32_t d0 = 0; d0 < nd; d0 += MLM_TILE_DESC) {
            const uint32_t n_staged = min(nd - d0, (uint32_t)MLM_TILE_DESC);
            __syncthreads();
            uint32_t a[4] = {0u, 0u, 0u, 0u}, t4[4];
            mlm_u32x4 dd = mlm_u32x4{0u, 0u, 0u, 0u};
            if (threadIdx.x < n_staged) {
                dd = d0 == 0 ? dd_first : descs[nth_column(d0 + threadIdx.x)];
                a[0] = dd.y;
                a[1] = dd.w;
            }
            mlm_block_excl_scan4<MLM_TILE_THREADS / 64>(a, s_w, t4);
            if (threadIdx.x < n_staged) {
                s_desc[threadIdx.x] = dd;
                s_dm[threadIdx.x] = a[0];
                s_dh[threadIdx.x] = a[1];
            }
            __syncthreads();
            if (what == 0)
                for (uint32_t j0 = threadIdx.x; j0 < t4[0]; j0 += 4u * MLM_TILE_THREADS) { // miss cells, four of a thread's loads in flight
                    uint32_t vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t j = j0 + (uint32_t)u * MLM_TILE_THREADS;
                        vv[u] = MLM_NIL;
                        if (j < t4[0]) {
                            uint32_t d, o;
                            locate(s_dm, n_staged, j, d, o);
                            vv[u] = mlm_gp(P.mc_list)[s_desc[d].x + o]; // (16-bit entries: never MLM_NIL)
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (vv[u] == MLM_NIL) continue;
                        if (vv[u] < NV) atomicAdd(&s_cnt[vv[u]], 1u);
                        else s_fail = 1;
                    }
                }
            // hits.  The first two of a thread (of the first staged pass: practically every tile) are kept in registers with their
            // increment and key — nothing of that depends on the map or on the counts — so the placing pass reads no memory
            auto place_hit = [&](uint32_t pos, uint32_t v, float inc, unsigned long long key) {
                const uint32_t pl = s_place[v];
                if ((pl >> 16) == 0xFFFFu) { // the voxel's only hit: its increment rides in the record (that word is this lane's)
                    MLM_GLOBAL MlmVoxRec *rec = mlm_gp(P.vr_rec) + rec_base + (pl & 0xFFFFu);
                    rec->inc_first = __float_as_uint(inc);
                } else {
                    const uint32_t k = (atomicSub(&s_cnt[v], 0x10000u) >> 16) - 1u; // (counts down: n_hit - 1 .. 0)
                    MLM_GLOBAL MlmVoxHit *hh = mlm_gp(P.vr_hit) + hit_base + (pl >> 16) + k;
                    *(MLM_GLOBAL mlm_u32x4 *)hh = mlm_u32x4{(uint32_t)key, (uint32_t)(key >> 32), __float_as_uint(inc), pos};
                }
            };
            uint32_t j_tail = threadIdx.x;
            if (d0 == 0) {
#pragma unroll
                for (int q = 0; q < MLM_TILE_KEEP; ++q) {
                    const uint32_t j = threadIdx.x + (uint32_t)q * MLM_TILE_THREADS;
                    if (j >= t4[1]) continue;
                    if (what == 0) {
                        uint32_t d, o;
                        locate(s_dh, n_staged, j, d, o);
                        const uint32_t pos = s_desc[d].z + o;
                        const uint32_t v = mlm_gp(P.hl_vt16)[pos];
                        keep_pos[q] = pos;
                        keep_v[q] = v;
                        keep_inc[q] = mlm_gp(P.hl_inc)[pos];
                        keep_key[q] = mlm_gp(P.hl_key)[pos];
                        if (v < NV) atomicAdd(&s_cnt[v], 0x10000u);
                        else s_fail = 1;
                    } else if (keep_v[q] < NV) {
                        place_hit(keep_pos[q], keep_v[q], keep_inc[q], keep_key[q]);
                    }
                }
                j_tail += MLM_TILE_KEEP * MLM_TILE_THREADS;
            }
            for (uint32_t j = j_tail; j < t4[1]; j += MLM_TILE_THREADS) {
                uint32_t d, o;
                locate(s_dh, n_staged, j, d, o);
                const uint32_t pos = s_desc[d].z + o;
                const uint32_t v = mlm_gp(P.hl_vt16)[pos];
                if (v >= NV) {
                    s_fail = 1;
                    continue;
                }
                if (what == 0) atomicAdd(&s_cnt[v], 0x10000u);
                else place_hit(pos, v, mlm_gp(P.hl_inc)[pos], mlm_gp(P.hl_key)[pos]);
            }
        }
    };
    for_cells(0);
    MLM_TPHASE(1); // counting pass (descriptor scan, miss cells, hits)
    if (probe && (int)threadIdx.x < n_combo) s_slot[threadIdx.x] = slot0;
    __syncthreads();
    MLM_TPHASE(2); // block lookups arrive
    const uint32_t per = (NV + MLM_TILE_THREADS - 1) / MLM_TILE_THREADS, v_lo = min(NV, threadIdx.x * per), v_hi = min(NV, v_lo + per);
    const uint32_t lvz = (uint32_t)P.lv_nz;
    auto combo_of = [&](uint32_t vxy, uint32_t zz) { // index of the block of tile voxel (vxy, zz) among the blocks the tile overlaps
        return ((int)(s_ztab[zz] & 0xFFFFu) * ngy + (int)(s_xytab[edge + (vxy >> P.tile_sh)] & 0xFFFFu)) * ngx + (int)(s_xytab[vxy & (edge - 1u)] & 0xFFFFu);
    };
    // ---- the blocks the frame touches in this tile exist before the frame is applied (allocate_ram, map_local.h:215-231: any
    //      hit or miss cell creates its block; creating it earlier than the reference would is invisible — it stays 0 / 'u' until
    //      the frame is applied).  A full pool flags the frame: k_apply_tiles stops in front of it, the host grows the pool and
    //      k_alloc_retry fills in the slots.
    {
        uint32_t vxy = v_lo / lvz, zz = v_lo - vxy * lvz;
        for (uint32_t v = v_lo; v < v_hi; ++v) {
            if (s_cnt[v]) s_touch[combo_of(vxy, zz)] = 1;
            if (++zz == lvz) {
                zz = 0;
                ++vxy;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n_combo; c += MLM_TILE_THREADS)
        if (s_touch[c] && s_slot[c] < 0) {
            const int bz = c / (ngx * ngy), r = c - bz * ngx * ngy, by = r / ngx, bx = r - by * ngx;
            const int sl = mlm_block_slot(P, mlm_pack_key(gx0 + bx, gy0 + by, gz0 + bz));
            s_slot[c] = sl;
            if (sl < 0) mlm_gp(P.ctr)->pool_short = 1u;
        }
    __syncthreads();
    MLM_TPHASE(3); // touched blocks, creation of the missing ones
    // ---- compaction: a thread's voxels are contiguous; touched voxels get consecutive records, voxels with several hits
    //      consecutive room in vr_hit
    uint32_t a[4] = {0u, 0u, 0u, 0u}, t4[4];
    for (uint32_t v = v_lo; v < v_hi; ++v) {
        const uint32_t c = s_cnt[v];
        if (c) {
            ++a[0];
            if ((c >> 16) >= 2u) a[1] += c >> 16;
        }
    }
    mlm_block_excl_scan4<MLM_TILE_THREADS / 64>(a, s_w, t4);
    if (threadIdx.x == 0) {
        if (t4[0] > 0xFFFFu || t4[1] >= 0xFFFFu) s_fail = 1; // (the packed places hold 16 bits each)
        s_base[0] = t4[0] ? g_atomic_add(&mlm_gp(P.ctr)->mvox_cnt[0][0], t4[0]) : 0u;
        s_base[1] = t4[1] ? g_atomic_add(&mlm_gp(P.ctr)->mvox_cnt[1][0], t4[1]) : 0u;
        if (s_base[0] + t4[0] > P.rec_cap || s_base[1] + t4[1] > P.vh_cap) s_fail = 2; // (lists sized by need: enlarged, then the frame's Stage A runs again)
    }
    __syncthreads();
    if (s_fail) { // (uniform) the frame is redone
        if (threadIdx.x == 0) mlm_sector_fail(P, F, s_fail == 2u);
        return;
    }
    MLM_TPHASE(4); // compaction scan, reservations
    rec_base = s_base[0];
    hit_base = s_base[1];
    if (threadIdx.x == 0) // the tile's records of this frame, for the workgroup that applies the world tile (k_apply_tiles)
        *(MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.tile_dir) + 4 * (size_t)tile) = mlm_u32x4{rec_base, t4[0], (uint32_t)F.seq, 0u};
    {
        uint32_t o_rec = a[0], o_hit = a[1];
        uint32_t vxy = v_lo / lvz, zz = v_lo - vxy * lvz;
        for (uint32_t v = v_lo; v < v_hi; ++v) {
            const uint32_t c = s_cnt[v];
            if (c) {
                const uint32_t nh = c >> 16;
                s_place[v] = o_rec | ((nh >= 2u ? o_hit : 0xFFFFu) << 16);
                const uint32_t xt = s_xytab[vxy & (edge - 1u)], yt = s_xytab[edge + (vxy >> P.tile_sh)], zt = s_ztab[zz];
                const int bx = (int)(xt & 0xFFFFu), by = (int)(yt & 0xFFFFu), bz = (int)(zt & 0xFFFFu);
                const int cid = ((int)(zt >> 16) * P.n + (int)(yt >> 16)) * P.n + (int)(xt >> 16);
                const int slot = s_slot[(bz * ngy + by) * ngx + bx];
                const int at = slot >= 0 ? slot * P.cells + cid : -1; // (< 2^31: alloc_pool; -1: k_alloc_retry once the pool has grown)
                MLM_GLOBAL MlmVoxRec *rec = mlm_gp(P.vr_rec) + rec_base + o_rec;
                // (lv_nz <= 1024, at most 64 columns per tile; one hit: its increment is written by that hit's lane in the placing pass)
                *(MLM_GLOBAL mlm_u32x4 *)rec = mlm_u32x4{(uint32_t)at, (c & 0xFFFFu) | (((vxy << 10) | zz) << 16), nh != 1u ? hit_base + o_hit : 0u, nh};
                ++o_rec;
                if (nh >= 2u) o_hit += nh;
            }
            if (++zz == lvz) {
                zz = 0;
                ++vxy;
            }
        }
    }
    MLM_TPHASE(5); // records
    for_cells(1); // (its first barrier makes the places visible)
    MLM_TPHASE(6); // placing pass
    MLM_TPHASE_END
}
// A workgroup takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ... of the frame-local grid: it reads the column masks of ALL its
// candidates in one go (a frame reaches a fraction of its grid's tiles: most masks are empty) and works on the tiles that some
// column announced itself to.  No list of touched tiles, no per-tile counter: the columns hand over with plain stores and one
// fire-and-forget atomic each (k_sector).
__global__ __launch_bounds__(MLM_TILE_THREADS) void k_tile(MLM_SLOT_ARGS) {
    MLM_SLOT_SETUP
    MLM_SPAN_BEGIN(1)
    __shared__ uint32_t s_cand[MLM_TILE_THREADS];
    __shared__ unsigned int s_state;
    if (threadIdx.x == 0) s_state = mlm_gp(P.ctr)->sector_overflow;
    __syncthreads();
    if (s_state == 1u) return; // columns of the frame still wait for the large-table pass: the tiles are grouped when the host has run it
    const uint32_t TW = P.tile_words, KC = (uint32_t)MLM_TILE_THREADS / TW, nt = (uint32_t)P.n_tiles; // candidates per round
    for (uint32_t k0 = 0; blockIdx.x + k0 * gridDim.x < nt; k0 += KC) { // (uniform)
        if (k0) __syncthreads(); // (the previous round's masks are no longer read)
        {
            const uint32_t k = threadIdx.x / TW, w = threadIdx.x - k * TW;
            const unsigned long long tile = blockIdx.x + (unsigned long long)(k0 + k) * gridDim.x;
            s_cand[threadIdx.x] = (k < KC && tile < nt) ? mlm_gp(P.tile_cols)[(size_t)tile * TW + w] : 0u;
        }
        __syncthreads();
        for (uint32_t k = 0; k < KC; ++k) {
            const unsigned long long tile = blockIdx.x + (unsigned long long)(k0 + k) * gridDim.x;
            if (tile >= nt) break;
            uint32_t any = 0;
            for (uint32_t w = 0; w < TW; ++w) any |= s_cand[k * TW + w];
            if (!any) continue; // (uniform: every thread reads the same words)
            mlm_tile_one(P, F, (unsigned int)tile, s_cand + k * TW);
            __syncthreads(); // (the tile's shared state is no longer read)
        }
    }
    MLM_SPAN_END(1)
}

// One voxel record applied to the voxel's state (L, o): its hits in the reference's iteration order (descending key,
// map_local.cpp:157-171), then its misses (map_local.cpp:188-203).  hits: the frame's vr_hit; xkeys: exact keys of a replayed
// frame (by hit-list position), else null.
__device__ __forceinline__ void mlm_apply_record(const MlmDev &P0, const mlm_u32x4 &r, const MLM_GLOBAL MlmVoxHit *hits,
                                                 const MLM_GLOBAL unsigned long long *xkeys, float &L, uint8_t &o) {
    const float lo_max = P0.lo_max, lo_sh = P0.lo_sh;
    const uint32_t km = r.y & 0xFFFFu, nh = r.w, first = r.z;
    auto hit = [&](float inc) { // map_local.cpp:157-171
        if (L < lo_max) {
            L = L + inc;
            L = L > lo_max ? lo_max : L;
        }
        if (L > lo_sh && o != 'o') o = 'o';
    };
    if (nh == 1u) { // the common case: a single contribution
        hit(__uint_as_float(r.z));
    } else if (nh) {
        // (key, increment) of all its hits, next to each other in vr_hit, ordered in registers by descending key (the
        // reference's iteration order); a replayed frame's exact keys come from hl_key
        const MLM_GLOBAL mlm_u32x4 *hh = (const MLM_GLOBAL mlm_u32x4 *)(hits + first);
        auto key_of = [&](const mlm_u32x4 &e) -> unsigned long long {
            return xkeys ? xkeys[e.w] : ((unsigned long long)e.x | ((unsigned long long)e.y << 32));
        };
        if (nh <= MLM_APPLY_REGS) {
            unsigned long long ks[MLM_APPLY_REGS];
            float vs[MLM_APPLY_REGS];
#pragma unroll
            for (int q = 0; q < MLM_APPLY_REGS; ++q) {
                ks[q] = 0; // real keys are never 0
                vs[q] = 0.0f;
            }
            for (uint32_t j = 0; j < nh; ++j) {
                const mlm_u32x4 e = hh[j];
                unsigned long long kk = key_of(e);
                float inc = __uint_as_float(e.z);
#pragma unroll
                for (int q = 0; q < MLM_APPLY_REGS; ++q) {
                    if (kk > ks[q]) {
                        const unsigned long long tk = ks[q];
                        const float tv = vs[q];
                        ks[q] = kk;
                        vs[q] = inc;
                        kk = tk;
                        inc = tv;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < MLM_APPLY_REGS; ++q)
                if ((uint32_t)q < nh) hit(vs[q]);
        } else {
            // more hits than registers: repeated selection of the next key straight from memory
            unsigned long long last = ~0ull;
            for (uint32_t done = 0; done < nh; ++done) {
                unsigned long long bestkey = 0;
                float bestinc = 0.0f;
                for (uint32_t j = 0; j < nh; ++j) {
                    const mlm_u32x4 e = hh[j];
                    const unsigned long long kk = key_of(e);
                    if (kk < last && kk > bestkey) {
                        bestkey = kk;
                        bestinc = __uint_as_float(e.z);
                    }
                }
                hit(bestinc);
                last = bestkey;
            }
        }
    }
    mlm_apply_misses(P0, L, o, km);
}

// The part that needs the map (local_map_cartesian::input_pc_pose_direct, map_local.cpp:143-237), sector path: ONE launch for a
// whole batch of frames.  The frames of a stream must be applied in order, but the only thing one frame's update depends on is
// the same VOXEL's state after the frame before — so the work is cut by space, not by time: the frame-local grids are aligned to
// tile boundaries, a frame-local tile is a world tile, and the workgroup that owns a world tile walks the batch's frames in
// order and applies, frame after frame, the voxel records k_tile wrote for that tile (a barrier between frames; no ordering
// between workgroups at all — two tiles never share a voxel, and the blocks they may share were created by k_tile).
// Per record: the voxel's hits in the reference's iteration order (descending key, map_local.cpp:157-171), then its misses
// (map_local.cpp:188-203) — the reference runs all hits before all misses.  Two dependent round trips per frame and tile:
// records -> (log-odds, class, the hits of a voxel with several).
// Every workgroup stops in front of the same frame, the first one that cannot be applied as it stands: its Stage A gave up
// (redone on the cell-table path by the host), it does not fit the emulated hit container without a rehash (the host computes
// exact keys: MLM_FRAME_EXACT_KEYS), or k_tile could not create its blocks (the host grows the pool).  g->fail_frame tells the
// host which (sticky: batches behind it do nothing).  f_begin: first frame of the slot range to apply (replays).
#ifndef MLM_APPLY_U
#define MLM_APPLY_U 2 // records of a thread loaded together (4: 170 VGPRs and 1 % fewer frames/s in the pipeline — the kernel is hidden behind Stage A, its footprint is not)
#endif
__global__ __launch_bounds__(MLM_BLOCK) void k_apply_tiles(const MlmDev *__restrict__ slot_tab, const MlmFrame *__restrict__ frame_tab, int slot_base,
                                                           int n_frames, int f_begin, int z_span) {
    __builtin_amdgcn_s_setprio(3); // the serial chain of the pipeline: its few waves issue ahead of Stage A's
    // The tile's voxels stay in LDS for the whole batch: a voxel is fetched from the map when a frame first touches it and
    // written back once, after the last frame — between frames only an LDS barrier stands (not a round trip to memory and
    // the wait for the stores' acknowledgement), and the map's lines are touched once per batch instead of once per frame.
    // The frames' grids differ in their z origin (the sensor's height): the LDS column of a tile spans lv_nz + z_span layers from
    // the lowest origin of the range — z_span = the spread of the range's origins, known to the host, which cuts a batch where
    // it would exceed a grid height (LDS left to the other streams' kernels is throughput: §5 of DESIGN.md).
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    // per frame of the range, gathered once (the loop over the frames touches no frame or slot parameters in memory):
    __shared__ const MlmVoxRec *s_rec[64];           // this tile's records of the frame
    __shared__ const MlmVoxHit *s_hit[64];           // the frame's hit pairs
    __shared__ const unsigned long long *s_key[64];  // its exact keys (replayed frames), else null
    __shared__ uint32_t s_count[64];
    __shared__ int s_dz[64];                         // the frame's z origin above the range's lowest
    __shared__ int s_wx0, s_wy0, s_stop, s_nwx, s_z0;
    const int lane = threadIdx.x & 63;
    const MlmDev &P0 = slot_tab[slot_base];
    const int sh = P0.tile_sh, lv_nz = P0.lv_nz;
    MLM_GLOBAL float *const pool_L = mlm_gp(P0.log_odds); // (the pool is the same in every slot)
    MLM_GLOBAL uint8_t *const pool_o = mlm_gp(P0.occ);
    // the batch's box of world tiles (every workgroup derives it from the frames' grid origins: no launch argument changes from
    // call to call, which lets a single frame's launch sequence be replayed as a graph)
    if (threadIdx.x < 64) {
        int wx = 0x7FFFFFFF, wy = 0x7FFFFFFF, wxm = -0x7FFFFFFF, z0 = 0x7FFFFFFF;
        bool ok = true;
        if (lane < n_frames) {
            const MlmFrame &F = frame_tab[slot_base + lane];
            const MLM_GLOBAL MlmCounters *c = mlm_gp(slot_tab[slot_base + lane].ctr);
            wx = F.lv_o[0] >> sh;
            wy = F.lv_o[1] >> sh;
            wxm = wx;
            z0 = F.lv_o[2];
            ok = (F.flags & MLM_FRAME_SKIP) ||
                 (c->sector_overflow == 0u && c->pool_short == 0u && ((F.flags & MLM_FRAME_EXACT_KEYS) || c->u_hit <= F.rehash_thr));
        }
        for (int off = 32; off > 0; off >>= 1) {
            wx = min(wx, __shfl_xor(wx, off, 64));
            wy = min(wy, __shfl_xor(wy, off, 64));
            wxm = max(wxm, __shfl_xor(wxm, off, 64));
            z0 = min(z0, __shfl_xor(z0, off, 64));
        }
        const unsigned long long bad = __ballot(!ok && lane >= f_begin && lane < n_frames);
        if (lane == 0) {
            s_wx0 = wx;
            s_wy0 = wy;
            s_z0 = z0;
            s_nwx = wxm - wx + P0.n_tx;
            int stop = bad ? __ffsll((long long)bad) - 1 : n_frames;
            // an EARLIER batch is to be replayed first: nothing of this one may be applied
            const int ff = __hip_atomic_load(&P0.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ff < frame_tab[slot_base + f_begin].seq) stop = f_begin;
            else if (stop < n_frames && blockIdx.x == 0) atomicMin(&P0.g->fail_frame, frame_tab[slot_base + stop].seq);
            s_stop = stop;
        }
    }
    __syncthreads();
    const int f_stop = s_stop;
    const uint32_t NZ = (uint32_t)(lv_nz + z_span), NV = NZ << (2 * sh);
    float *s_L = (float *)s_dyn;                         // [NV] log-odds of the tile's voxels (valid where s_at != NIL)
    uint32_t *s_at = (uint32_t *)(s_dyn + 4u * NV);      // [NV] the voxel's address in the pool (slot * cells + cell id), MLM_NIL: not fetched
    uint8_t *s_o = (uint8_t *)(s_dyn + 8u * NV);         // [NV] occupancy class
    for (uint32_t v = threadIdx.x; v < NV; v += blockDim.x) s_at[v] = MLM_NIL;
    const int WX = s_wx0 + (int)(blockIdx.x % (unsigned int)s_nwx), WY = s_wy0 + (int)(blockIdx.x / (unsigned int)s_nwx);
    // this world tile's records in every frame of the range (one lane per frame: the directory entries arrive together)
    if (threadIdx.x < 64) {
        uint32_t count = 0;
        const MlmVoxRec *rec = nullptr;
        const MlmVoxHit *hit = nullptr;
        const unsigned long long *key = nullptr;
        int dz = 0;
        if (lane >= f_begin && lane < f_stop) {
            const MlmFrame &F = frame_tab[slot_base + lane];
            const MlmDev &P = slot_tab[slot_base + lane];
            const int tx = WX - (F.lv_o[0] >> sh), ty = WY - (F.lv_o[1] >> sh);
            dz = F.lv_o[2] - s_z0;
            if (!(F.flags & MLM_FRAME_SKIP) && tx >= 0 && ty >= 0 && tx < P.n_tx && ty * P.n_tx + tx < P.n_tiles && dz >= 0 && dz <= z_span) {
                const mlm_u32x4 e = *(const MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.tile_dir) + 4 * (size_t)(ty * P.n_tx + tx));
                if (e.z == (uint32_t)F.seq) {
                    rec = P.vr_rec + e.x;
                    count = e.y;
                    hit = P.vr_hit;
                    key = (F.flags & MLM_FRAME_EXACT_KEYS) ? (const unsigned long long *)P.hl_key : nullptr;
                }
            }
        }
        s_rec[lane] = rec;
        s_hit[lane] = hit;
        s_key[lane] = key;
        s_count[lane] = count;
        s_dz[lane] = dz;
    }
    __syncthreads();
    // The records of the next frame that has any for this tile are fetched while the current frame is applied (they do not depend
    // on the map).  Two records per thread are held in registers; a tile with more takes the rest straight from memory.
    constexpr int PRE = 2;
    mlm_u32x4 nx[PRE];
    auto next_frame = [&](int f) {
        while (f < f_stop && s_count[f] == 0u) ++f;
        return f;
    };
    auto prefetch = [&](int f) {
        if (f >= f_stop) return;
        const MLM_GLOBAL mlm_u32x4 *recs = (const MLM_GLOBAL mlm_u32x4 *)mlm_gp(s_rec[f]);
        const uint32_t count = s_count[f];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const uint32_t i = threadIdx.x + (uint32_t)k * blockDim.x;
            if (i < count) nx[k] = recs[i];
        }
    };
    int f = next_frame(f_begin);
    prefetch(f);
    while (f < f_stop) {
        const uint32_t count = s_count[f];
        const MLM_GLOBAL mlm_u32x4 *recs = (const MLM_GLOBAL mlm_u32x4 *)mlm_gp(s_rec[f]);
        const MLM_GLOBAL MlmVoxHit *hits = mlm_gp(s_hit[f]);
        const MLM_GLOBAL unsigned long long *xkeys = mlm_gp(s_key[f]); // exact keys of a replayed frame, else null
        const uint32_t dz = (uint32_t)s_dz[f];
        mlm_u32x4 cur[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) cur[k] = nx[k];
        const int f_next = next_frame(f + 1);
        prefetch(f_next);
        // a thread's records in chunks of U: the records of a chunk are loaded together (the first two came with the previous
        // frame), then the voxels that are not in LDS yet are fetched together, then the chunk is applied — the round trips of
        // a chunk overlap instead of following each other
        constexpr int U = MLM_APPLY_U;
        for (uint32_t i0 = threadIdx.x, chunk = 0; i0 < count; i0 += blockDim.x * U, ++chunk) {
            mlm_u32x4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                if (chunk == 0 && u < PRE) r[u] = cur[u];
                else if (i < count) r[u] = recs[i];
            }
            float Lv[U];
            uint8_t ov[U];
            uint32_t vts[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = i0 + (uint32_t)u * blockDim.x;
                vts[u] = MLM_NIL;
                Lv[u] = 0.0f;
                ov[u] = 'u';
                if (i >= count) continue;
                const uint32_t vt = (r[u].y >> 26) * NZ + ((r[u].y >> 16) & 1023u) + dz; // (tile column, layer above the range's lowest origin)
                if ((int)r[u].x < 0 || vt >= NV) continue; // (cannot happen: a frame whose blocks could not be created is not applied)
                vts[u] = vt;
                if (s_at[vt] == MLM_NIL) { // first touch of the voxel in this batch (one record per voxel and frame: no other lane has it now)
                    const uint32_t at = r[u].x;
                    Lv[u] = pool_L[at];
                    ov[u] = pool_o[at];
                    s_at[vt] = at;
                } else {
                    Lv[u] = s_L[vt];
                    ov[u] = s_o[vt];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t vt = vts[u];
                if (vt == MLM_NIL) continue;
                float L = Lv[u];
                uint8_t o = ov[u];
                mlm_apply_record(P0, r[u], hits, xkeys, L, o);
                s_L[vt] = L;
                s_o[vt] = o;
            }
        }
        __syncthreads(); // (this frame's voxels are in LDS before the next frame reads them)
        f = f_next;
    }
    // the voxels the batch touched go back to the map
    for (uint32_t v = threadIdx.x; v < NV; v += blockDim.x) {
        const uint32_t at = s_at[v];
        if (at != MLM_NIL) {
            pool_L[at] = s_L[v];
            pool_o[at] = s_o[v];
        }
    }
}

// In front of a Stage A (first node of the single-frame graph; first launch of a batch's Stage A on the sector path): the frames'
// parameters come from pinned host memory — workgroup j copies frame j's into the device-resident table and clears the slot's
// counters (the words that are ever written: mlm_ctr_live_word) — one kernel instead of a copy and a fill (which is two fill kernels
// for the 6 KB of MlmCounters: 15 us in front of a lone frame); the counters and the map-wide flags go back the same way (the tail of
// k_apply_single, k_ex_apply_misses).
__global__ __launch_bounds__(128) void k_frame_prologue(const MlmFrame *host_frame, MlmFrame *dev_frame, MlmCounters *ctr) {
    const uint32_t *src = (const uint32_t *)(host_frame + blockIdx.x);
    uint32_t *dst = (uint32_t *)(dev_frame + blockIdx.x), *c = (uint32_t *)(ctr + blockIdx.x);
    for (unsigned int i = threadIdx.x; i < sizeof(MlmFrame) / 4; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x < MLM_CTR_SCALARS + 96u) c[mlm_ctr_live_word(threadIdx.x)] = 0u;
}

// The hand-back of a lone frame, by the workgroup of its last kernel that finishes last (counted in MlmCounters::apply_done): the live
// words of the slot's counters and the map-wide flags into pinned host memory, then — as the last store — the ticket in the host copy's
// MlmGlobal::pad, which the calling thread polls instead of sleeping in hipStreamSynchronize.  cleared: the device counters are left
// clear for the slot's next frame and the host is told so (MLM_CTR_CLEARED in the word it has no other use for).  Call with all threads.
__device__ __forceinline__ void mlm_hand_back(const MlmDev &P, MlmCounters *host_ctr, MlmGlobal *host_g, unsigned int ticket, bool cleared) {
    __shared__ unsigned int s_hb_last;
    __syncthreads();
    if (threadIdx.x == 0) s_hb_last = g_atomic_add(&mlm_gp(P.ctr)->apply_done, 1u) == gridDim.x * gridDim.y - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_hb_last) return;
    if (threadIdx.x < 128u) {
        const unsigned int t = threadIdx.x;
        if (t < MLM_CTR_SCALARS + 96u) {
            const unsigned int w = mlm_ctr_live_word(t);
            uint32_t val = __hip_atomic_load((const uint32_t *)P.ctr + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cleared) ((uint32_t *)P.ctr)[w] = 0u;
            if (w == MLM_CTR_APPLY_DONE) val = cleared ? MLM_CTR_CLEARED : 0u;
            ((uint32_t *)host_ctr)[w] = val;
        } else if (t >= 120u && t < 123u) {
            ((uint32_t *)host_g)[t - 120u] = __hip_atomic_load((const uint32_t *)P.g + (t - 120u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&host_g->pad, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same for ONE frame (a synchronous single-frame call — the reference's call pattern, src/mlmap.cpp:463-507 — and the replay
// of one frame): nothing is carried from frame to frame, so nothing is staged in LDS, and the tiles do not matter — the frame's
// records lie in one list.  What a single frame costs is the number of DEPENDENT trips to memory (a kernel's first touch of what
// another kernel wrote crosses the XCDs: several microseconds each), so every thread fetches its first record together with the
// frame's flags and counts instead of after them: parameters -> (flags, count, record) -> (log-odds, class) -> store.
// Stops in front of the frame under the same conditions as k_apply_tiles.
__global__ __launch_bounds__(MLM_BLOCK) void k_apply_single(MLM_SLOT_ARGS, MlmCounters *host_ctr, MlmGlobal *host_g) {
    MLM_SLOT_SETUP
    __builtin_amdgcn_s_setprio(3);
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    const MLM_GLOBAL mlm_u32x4 *recs = (const MLM_GLOBAL mlm_u32x4 *)mlm_gp(P.vr_rec);
    mlm_u32x4 r = mlm_u32x4{0u, 0u, 0u, 0u};
    if (i0 < P.rec_cap) r = recs[i0]; // (before the count is known: the list's memory is there either way)
    const MLM_GLOBAL MlmCounters *c = mlm_gp(P.ctr);
    const uint32_t total = min(c->mvox_cnt[0][0], P.rec_cap);
    const bool ok = c->sector_overflow == 0u && c->pool_short == 0u && ((F.flags & MLM_FRAME_EXACT_KEYS) || c->u_hit <= F.rehash_thr);
    bool run = !(F.flags & MLM_FRAME_SKIP) && __hip_atomic_load(&P.g->fail_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= F.seq; // (else: an earlier frame is to be replayed first)
    if (run && !ok) { // (uniform)
        if (i0 == 0) atomicMin(&P.g->fail_frame, F.seq);
        run = false;
    }
    if (run) {
        MLM_GLOBAL float *const pool_L = mlm_gp(P.log_odds);
        MLM_GLOBAL uint8_t *const pool_o = mlm_gp(P.occ);
        const MLM_GLOBAL MlmVoxHit *hits = mlm_gp(P.vr_hit);
        const MLM_GLOBAL unsigned long long *xkeys = (F.flags & MLM_FRAME_EXACT_KEYS) ? (const MLM_GLOBAL unsigned long long *)mlm_gp(P.hl_key) : nullptr;
        for (uint32_t i = i0; i < total; i += stride) {
            if (i != i0) r = recs[i];
            if ((int)r.x < 0) continue; // (no block: cannot happen, such a frame is not applied)
            const uint32_t at = r.x;
            float L = pool_L[at];
            uint8_t o = pool_o[at];
            mlm_apply_record(P, r, hits, xkeys, L, o);
            pool_L[at] = L;
            pool_o[at] = o;
        }
    }
    if (!host_ctr) return; // (a replayed frame: the host reads the counters back itself)
    // Last node of the single-frame graph.  No release fence in front of the arrival: nothing the host does on the ticket alone reads what
    // the workgroups stored into the map — whatever reads the map is a later kernel of the stream —, the counters were final before this
    // kernel, and the flag a frame that is not applied raises is an atomic of workgroup 0's first thread, fenced here (an agent-scope
    // fence costs a lone frame 2-4 us).  A frame that has been applied leaves its slot's counters clear (submit_single_graph: the
    // slot's next small frame starts without k_frame_prologue); one that was not keeps them — the host's replay reads them.
    if (!ok && threadIdx.x == 0) __threadfence();
    mlm_hand_back(P, host_ctr, host_g, (unsigned int)F.seq + 1u, run);
}

// After the host has grown the block pool: the blocks a frame's voxel records still lack (k_tile found the pool full) are created
// and the records' pool addresses filled in; pool_short is cleared if every one fitted.  A record carries no block key: its
// block and cell follow from its tile and voxel-in-tile index exactly as in k_tile (one workgroup walks the tiles in turns).
__global__ __launch_bounds__(MLM_BLOCK) void k_alloc_retry(const MlmDev P, const MlmFrame F) {
    bool failed = false;
    const uint32_t edge = 1u << P.tile_sh;
    for (unsigned int tile = blockIdx.x; tile < (unsigned int)P.n_tiles; tile += gridDim.x) {
        const mlm_u32x4 dir = *(const MLM_GLOBAL mlm_u32x4 *)(mlm_gp(P.tile_dir) + 4 * (size_t)tile);
        if (dir.z != (uint32_t)F.seq) continue; // (the tile has no records of this frame)
        const int ty = (int)(tile / (unsigned int)P.n_tx), tx = (int)tile - ty * P.n_tx;
        const int X0 = (tx << P.tile_sh) + F.lv_o[0], Y0 = (ty << P.tile_sh) + F.lv_o[1], Z0 = F.lv_o[2];
        for (unsigned int i = threadIdx.x; i < dir.y; i += blockDim.x) {
            MLM_GLOBAL MlmVoxRec *rec = mlm_gp(P.vr_rec) + dir.x + i;
            if (rec->at >= 0) continue;
            const uint32_t vt = rec->km_vt >> 16, vxy = vt >> 10, zz = vt & 1023u;
            const int X = X0 + (int)(vxy & (edge - 1u)), Y = Y0 + (int)(vxy >> P.tile_sh), Z = Z0 + (int)zz;
            const int gx = mlm_floor_div(X, P.n), gy = mlm_floor_div(Y, P.n), gz = mlm_floor_div(Z, P.n);
            const int cid = ((Z - gz * P.n) * P.n + (Y - gy * P.n)) * P.n + (X - gx * P.n);
            const int slot = mlm_block_slot(P, mlm_pack_key(gx, gy, gz));
            if (slot >= 0) rec->at = slot * P.cells + cid;
            else failed = true;
        }
    }
    if (__any(failed) && (threadIdx.x & 63) == 0) mlm_gp(P.ctr)->pool_short = 2u; // (2: still short after this pass)
}
// (second launch of the pair: turns "not seen short by the pass above" into "complete")
__global__ void k_alloc_retry_done(const MlmDev P) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mlm_gp(P.ctr)->pool_short = mlm_gp(P.ctr)->pool_short == 2u ? 1u : 0u;
}

// Test hook (mlm_debug_probe_seeds): the largest relative error of the v_rcp_f64 / v_rsq_f64 seeds and of their once-refined
// forms (mlm_rcp_approx, mlm_sqrt_approx) over n values spread over [2^-40, 2^40], against the correctly rounded operations.
// out[0..3]: bit patterns of the four maxima (non-negative doubles order like their bits).
__global__ void k_probe_seeds(unsigned long long *out, unsigned int n) {
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // a mantissa from a hash of i, an exponent that walks through the range
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        const double mant = 1.0 + (double)(h >> 12) * (1.0 / 4503599627370496.0); // [1, 2)
        const double v = ldexp(mant, (int)(i % 81u) - 40);
        const double inv = 1.0 / v, rt = sqrt(v);
        m0 = fmax(m0, fabs(__builtin_amdgcn_rcp(v) - inv) / inv);
        m1 = fmax(m1, fabs(mlm_rcp_approx(v) - inv) / inv);
        m2 = fmax(m2, fabs(__builtin_amdgcn_rsq(v) * v - rt) / rt);
        m3 = fmax(m3, fabs(mlm_sqrt_approx(v) - rt) / rt);
    }
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(m0));
    atomicMax(&out[1], (unsigned long long)__double_as_longlong(m1));
    atomicMax(&out[2], (unsigned long long)__double_as_longlong(m2));
    atomicMax(&out[3], (unsigned long long)__double_as_longlong(m3));
}
