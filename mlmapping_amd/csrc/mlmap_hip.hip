// mlmap_hip.hip — host side of libmlmap_hip.so: the C ABI of include/mlmap_hip.h.
//
// Host work per frame is O(1): the pose composition T_ls (map_awareness.cpp:184-186, mlm_host.h), the bookkeeping of
// the emulated libstdc++ rehash policy of hit_idx_odds_hashmap, and kernel launches.  All per-point / per-cell /
// per-voxel work runs in the kernels of mlm_kernels_sector.h (default path) and mlm_kernels.h / mlm_kernels_explore.h.  There is no
// CPU fallback: every entry point fails with MLM_ERR_HIP when the device is unavailable.
//
// ONE device translation unit (the kernels live in headers and are launched from here); the host driver is cut into parts that
// are included in this order: mlm_handle.h (the handle, knobs, launch helpers) -> mlm_stage_a.h (Stage A launches, exact
// ordering, statistics) -> mlm_explore_host.h (frontier mode) -> mlm_submit.h (submission, drain / replay, single-frame graph)
// -> mlm_resources.h (device memory: pool growth, frame slots, queries' launcher) -> mlm_mirror.h (host mirror of the map for small
// query batches); this file holds the extern "C" entry points.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <atomic>
#include <chrono>

#include "../../include/mlmap_hip.h"
#include "mlm_kernels_explore.h"
#include "mlm_kernels_sector.h"
#include "mlm_host.h"
#include "mlm_mapview.h"

#include "mlm_handle.h"
#include "mlm_stage_a.h"
#include "mlm_explore_host.h"
#include "mlm_submit.h"
#include "mlm_resources.h"
#include "mlm_mirror.h"

extern "C" {

int mlm_abi_version(void) { return MLM_ABI_VERSION; }

int mlm_debug_set(const char *name, long long value) {
    if (!name) return MLM_ERR_INVALID;
    bool known = false;
    for (const char *k : kKnobNames) known = known || strcmp(k, name) == 0;
    if (!known) return MLM_ERR_INVALID;
    KnobStore &k = knob_store();
    std::lock_guard<std::mutex> lock(k.mu);
    k.v[name] = value;
    return MLM_OK;
}
int mlm_debug_reset(void) {
    KnobStore &k = knob_store();
    std::lock_guard<std::mutex> lock(k.mu);
    k.v.clear();
    return MLM_OK;
}

int mlm_debug_probe_seeds(mlm_handle *h, double out4[4]) {
    if (!h || !out4) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long *d = nullptr;
    HIPCHK(h, hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_seeds, dim3(1024), dim3(256), 0, h->stream, d, 1u << 26);
        e = hipMemcpyAsync(out4, d, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) {
        h->err = std::string("mlm_debug_probe_seeds: ") + hipGetErrorString(e);
        return MLM_ERR_HIP;
    }
    return MLM_OK;
}

int mlm_debug_clocks(mlm_handle *h, double out_us[8], int reset) {
    if (!h || !out_us) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    for (int i = 0; i < 8; ++i) {
        out_us[i] = h->clk[i];
        if (reset) h->clk[i] = 0.0;
    }
    return MLM_OK;
}

int mlm_host_register(mlm_handle *h, const void *ptr, size_t bytes) {
    if (!h || !ptr || bytes == 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault));
    return MLM_OK;
}
int mlm_host_unregister(mlm_handle *h, const void *ptr) {
    if (!h || !ptr) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = drain(h); // (nothing may still be reading from it)
    if (rc) return rc;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipHostUnregister(const_cast<void *>(ptr)));
    return MLM_OK;
}

const char *mlm_last_error(mlm_handle *h) { return h ? h->err.c_str() : "null handle"; }

int mlm_create(const mlm_config *cfg, const mlm_limits *lim_in, int device, mlm_handle **out) {
    if (!cfg || !out) return MLM_ERR_INVALID;
    *out = nullptr;
    if (cfg->am_n_rho <= 1 || cfg->am_d_rho <= 0 || cfg->am_d_phi_deg <= 0 || cfg->am_d_z <= 0 || cfg->subbox_n <= 0 ||
        cfg->subbox_d_xyz <= 0 || cfg->am_n_z_below < 0 || cfg->am_n_z_over < 0)
        return MLM_ERR_INVALID;
    mlm_handle *h = new mlm_handle();
    *out = h; // returned even on failure so that mlm_last_error can be read; caller must mlm_destroy it
    long long kv = 0; // (value of a test / experiment knob, see mlm_debug_set)
    h->device = device;
    h->cfg = *cfg;
    if (lim_in) h->lim = *lim_in;
    if (h->lim.max_blocks <= 0) h->lim.max_blocks = 65536;
    if (h->lim.max_points <= 0) h->lim.max_points = 1280 * 720;
    if (h->lim.max_batch <= 0) h->lim.max_batch = 8;
    if (h->lim.max_batch > 64) h->lim.max_batch = 64;
    // (frontier mode: Stage A is batched too; the map-dependent part runs frame by frame, see run_slots)
    if ((long long)h->lim.max_points * 256 > 0xFFFFFFF0ll && cfg->use_exploration_frontiers) {
        h->err = "max_points too large for 32-bit miss insertion times";
        return MLM_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    HIPCHK(h, hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device < 0 || device >= ndev) {
        h->err = "no such HIP device";
        return MLM_ERR_HIP;
    }
    HIPCHK(h, hipSetDevice(device));
    {
        // Stage B+C (main stream) is the serial per-frame chain of short, latency-bound kernels; Stage A floods the
        // chip with wide kernels.  MLM_CU_SPLIT=k (default 0 = off) reserves the first k CUs for the main stream and
        // leaves the rest to Stage A, so that the chain is not stretched by queueing behind Stage A's waves.
        int lo = 0, hi = 0; // numerically lower = higher priority
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (knob("expand_block", kv)) h->expand_block = (unsigned int)std::max(64, (int)kv);
        if (knob("sort_block", kv)) h->sort_block = (unsigned int)std::min(256, std::max(64, (int)kv));
        if (knob("sort_grid", kv)) h->sort_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("chain_grid", kv)) h->chain_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("rank_grid", kv)) h->rank_grid = (unsigned int)std::max(1, (int)kv);
        else if ((long long)cfg->am_n_rho * cfg->am_n_z_below > 4000) h->rank_grid = 384; // (fine maps order a hundred thousand cells per frame: config 3 128 blocks 33.0 us, 256: 31.0, 512: 29.2)
        if (knob("collect_grid", kv)) h->collect_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("sc_block", kv)) h->sc_block = (unsigned int)std::min(256, std::max(64, (int)kv));
        if (knob("sc_grid", kv)) {
            h->sc_grid = (unsigned int)std::max(1, (int)kv);
            h->sc_grid_fixed = true;
        }
        h->cu_split = knob("cu_split", kv) ? (int)kv : 0;
        // MLM_CU_RESERVE=k: Stage A stays off the first k CUs, the main stream may use all of them
        if (knob("cu_reserve", kv)) h->cu_reserve = std::max(0, (int)kv);
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount;
        if (h->cu_split > 0 && h->cu_split < ncu) {
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int c = 0; c < h->cu_split; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
            HIPCHK(h, hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)mask.size(), mask.data()));
        } else {
            h->cu_split = 0;
            HIPCHK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, hi));
        }
    }
    for (int k = 0; k < MLM_SETS; ++k) {
        HIPCHK(h, hipEventCreateWithFlags(&h->batch_done[k], hipEventDisableTiming));
        HIPCHK(h, hipHostMalloc((void **)&h->h_gb[k], sizeof(MlmGlobal), hipHostMallocDefault));
        std::memset(h->h_gb[k], 0, sizeof(MlmGlobal));
    }

    MlmDev &P = h->P;
    // awareness constants, map_awareness.cpp:21-32
    P.dRho = cfg->am_d_rho;
    P.dPhi = cfg->am_d_phi_deg * M_PI / 180;
    P.dZ = cfg->am_d_z;
    P.nRho = cfg->am_n_rho;
    P.nPhi = static_cast<int>(360 / cfg->am_d_phi_deg);
    P.nZ = cfg->am_n_z_below + cfg->am_n_z_over + 1;
    P.zc = cfg->am_n_z_below;
    P.z_border_min = -(cfg->am_n_z_below * cfg->am_d_z) - 0.5 * cfg->am_d_z;
    P.nRhoPhi = P.nRho * P.nPhi;
    const long long ncells = (long long)P.nRhoPhi * P.nZ;
    if (ncells <= 0 || ncells > (1ll << 31) - 64) {
        h->err = "awareness map too large";
        return MLM_ERR_UNSUPPORTED;
    }
    if ((long long)h->lim.max_points * MLM_TIME_SLOTS > 0x7FFFFFF0ll) {
        h->err = "max_points too large for 32-bit insertion times";
        return MLM_ERR_UNSUPPORTED;
    }
    P.nCells = (int)ncells;
    P.RW = (P.nRho + 31) / 32;
    P.nMissWords = P.nZ * P.nPhi * P.RW;
    P.visibility = cfg->use_raycasting != 0;
    // local constants, map_local.cpp:56-62,126-130 (float casts as mlmap.cpp:77-81)
    P.d_sub = cfg->subbox_d_xyz;
    P.d_sub_half = P.d_sub * 0.5;
    P.n = cfg->subbox_n;
    P.d_glb = P.d_sub * P.n;
    P.inv_dRho = 1.0 / P.dRho;
    P.inv_dPhi = 1.0 / P.dPhi;
    P.inv_dZ = 1.0 / P.dZ;
    P.inv_d_max = std::max(P.inv_dRho, std::max(P.inv_dPhi, P.inv_dZ));
    P.inv_d_sub = 1.0 / P.d_sub;
    P.inv_d_glb = 1.0 / P.d_glb;
    P.cells = P.n * P.n * P.n;
    P.lo_min = static_cast<float>(cfg->log_odds_min);
    P.lo_max = static_cast<float>(cfg->log_odds_max);
    P.lo_miss = static_cast<float>(cfg->measurement_miss);
    P.lo_sh = static_cast<float>(cfg->occupied_sh);
    // camera, mlmap.cpp:15-18 (float members) and mlmap.h:85-86
    P.cx = (float)cfg->cam_cx;
    P.cy = (float)cfg->cam_cy;
    P.fx = (float)cfg->cam_fx;
    P.fy = (float)cfg->cam_fy;
    P.inv_factor = 1.0 / 1000.0;
    P.inv_fx = 1.0 / (double)P.fx;
    P.inv_fy = 1.0 / (double)P.fy;
    P.record_awareness = h->lim.record_awareness;
    P.node_lds = 448;
    P.agg_lds = 256;
    P.explore = cfg->use_exploration_frontiers != 0;
    P.max_blocks = h->lim.max_blocks;

    // T_bs, mlmap.cpp:22-25
    const double R[9] = {cfg->T_bs[0], cfg->T_bs[1], cfg->T_bs[2], cfg->T_bs[4], cfg->T_bs[5],
                         cfg->T_bs[6], cfg->T_bs[8], cfg->T_bs[9], cfg->T_bs[10]};
    h->q_bs = q_from_R(R);
    h->t_bs = D3{cfg->T_bs[3], cfg->T_bs[7], cfg->T_bs[11]};

    // tables (host libm, uploaded once)
    OddsModel om{cfg->am_d_rho, cfg->depth_noise_coe};
    h->odds_table.resize((size_t)21 * P.nRho);
    for (int d = -MLM_DIFF_RANGE; d <= MLM_DIFF_RANGE; ++d)
        for (int r = 0; r < P.nRho; ++r) h->odds_table[(size_t)(d + MLM_DIFF_RANGE) * P.nRho + r] = om.get_odds(d, (size_t)r);
    std::vector<float> sigma3(P.nRho);
    for (int r = 0; r < P.nRho; ++r) sigma3[r] = 3 * om.sigma_in_dr((size_t)r); // map_awareness.cpp:149
    {
        // The hit increment is the HOST libm's log10f(odd / (1 - odd)) in the reference (map_local.h:8, map_local.cpp:159).  If this
        // host's log10f is the one mlm_glibc_log10f restates (checked on the table's own logit arguments and a sweep of the range),
        // the kernels evaluate that restatement and every increment has the reference's float bits; else FP64 log10 rounded once.
        std::vector<float> ratios(h->odds_table.size());
        for (size_t i = 0; i < ratios.size(); ++i) ratios[i] = h->odds_table[i] / (1 - h->odds_table[i]);
        P.logit_exact = host_log10f_matches(ratios.data(), ratios.size()) ? 1 : 0;
        if (knob("logit_exact", kv)) P.logit_exact = P.logit_exact && (int)kv != 0;
    }
    {
        // a point spreads into 1 + 2*dmax cells; the wider the spread, the more groups and distinct cells a block produces
        int dmax = 0;
        for (int r = 0; r < P.nRho; ++r) {
            int d = 1;
            while ((float)d < sigma3[r] && r + d < P.nRho && d <= MLM_DIFF_RANGE) ++d;
            dmax = std::max(dmax, d - 1);
        }
        h->no_spread = dmax == 0; // (a point only ever hits its own cell: no cell collects several kinds, nothing to rank or replay)
        if (dmax > 4) {
            P.node_lds = 1024;
            P.agg_lds = 512;
        }
        // block size of k_bin_points and its LDS buffers (the sizes above are per 256 threads); experiment knobs
        P.bin_block = 256; // measured on config 2: 34.4k frames/s vs 33.6k with 512 and 31.9k with 1024
        if (knob("bin_block", kv)) P.bin_block = ((int)kv >= 1024) ? 1024u : ((int)kv >= 512 ? 512u : 256u);
        P.node_lds = P.node_lds * (P.bin_block / 256);
        P.agg_lds = P.agg_lds * (P.bin_block / 256);
        if (knob("node_lds", kv)) P.node_lds = (unsigned int)(int)kv;
        if (knob("agg_lds", kv)) P.agg_lds = (unsigned int)(int)kv; // power of two >= bin_block
        unsigned int lg = 0;
        while ((1u << lg) < P.agg_lds) ++lg;
        P.agg_shift = 32 - lg;
        P.bin_lds_bytes = P.node_lds * (unsigned int)sizeof(MlmNode) + P.agg_lds * (unsigned int)(sizeof(MlmCellAgg) + 4) +
                          MLM_RAY_LDS * 16;
        if (P.bin_lds_bytes > 48 * 1024) { // above the default dynamic-LDS limit
            HIPCHK(h, hipFuncSetAttribute((const void *)k_bin_points<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.bin_lds_bytes));
            HIPCHK(h, hipFuncSetAttribute((const void *)k_bin_points<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.bin_lds_bytes));
            HIPCHK(h, hipFuncSetAttribute((const void *)k_bin_points<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.bin_lds_bytes));
        }
    }
    if ((size_t)(2 * MLM_DIFF_RANGE + 1) * P.nRho * sizeof(float) > 64u * 1024u) { // (k_chain keeps the odds table in LDS)
        h->err = "am_n_Rho above 780 is not supported";
        return MLM_ERR_UNSUPPORTED;
    }
    {
        // sector path: LDS tables of one azimuth column (k_sector).  Cell table: a column rarely holds more hit cells than a
        // few per range step; references: (record, kind) pairs of its multi-kind cells.  A column that needs more makes
        // its frame fall back to the cell-table path.
        // (the hit cells of a column grow faster than its range steps — finer cells also mean more distinct cells per pixel footprint:
        // ~220 on config 2's 65 steps, ~1 000 on config 3's 130 — hence 6 entries per step for coarse maps, 12 for fine ones)
        unsigned int tab = 512;
        while (tab < (P.nRho > 100 ? 12u : 6u) * (unsigned int)P.nRho && tab < 2048u) tab <<= 1;
        if (knob("sec_tab", kv)) tab = (unsigned int)std::max(256, (int)kv); // power of two
        P.sec_tab = tab;
        // a column's workgroup: 256 threads where the table allows (at most 4 entries per thread, one thread per range step) — the
        // smaller workgroup and table leave wave slots and LDS of a CU to the other streams' kernels, which is worth more in the
        // pipeline (+5 % frames/s on config 2) than the 10 % the kernel loses alone
        h->sec_threads = (tab <= 1024u && P.nRho <= 256) ? 256 : 512;
        if (knob("sec_threads", kv)) h->sec_threads = ((int)kv <= 256 && tab <= 1024u && P.nRho <= 256) ? 256 : 512;
        if (P.sec_tab < (unsigned int)h->sec_threads) P.sec_tab = (unsigned int)h->sec_threads;
        if (knob("sec_backoff", kv)) h->sector_backoff_len = std::max(0, (int)kv);
        if (knob("sec_fail_every", kv)) P.sec_fail_every = (unsigned int)std::max(0, (int)kv);
        P.sec_lds_bytes = mlm_sec_lds(P.sec_tab, (unsigned int)(P.nZ * (P.explore ? P.nRho : P.RW)), (unsigned int)P.nRho, (unsigned int)P.nZ, P.explore).total;
        {
            // second pass for the columns that overflow that table: the largest table (up to 4096 entries = 8 per thread) that
            // fits a CU's LDS — an S1 column has 2 665 cells in all, so no scene overflows it there
            unsigned int big = 4096;
            while (big > P.sec_tab && mlm_sec_lds(big, (unsigned int)(P.nZ * (P.explore ? P.nRho : P.RW)), (unsigned int)P.nRho, (unsigned int)P.nZ, P.explore).total > 159u * 1024u) big >>= 1;
            if (knob("sec_tab_big", kv)) big = (unsigned int)(int)kv; // (0 or <= MLM_SEC_TAB: no second pass)
            P.sec_tab_big = big > P.sec_tab && big <= 8u * MLM_SEC_THREADS ? big : 0u;
            P.sec_big_lds_bytes = P.sec_tab_big ? mlm_sec_lds(P.sec_tab_big, (unsigned int)(P.nZ * (P.explore ? P.nRho : P.RW)), (unsigned int)P.nRho, (unsigned int)P.nZ, P.explore).total : 0u;
        }
        {
            // frame-local voxel grid: the awareness cylinder (radius nRho*dRho, height nZ*dZ) plus four voxels each side, cut
            // into tiles over its whole height: the largest edge (8, 4, 2, 1 voxels) with at most 4096 voxels per tile — k_tile counts
            // them in LDS, and the workgroup that applies a tile walks its records a few per thread and frame (measured: config 3's
            // 0.05 m map, 110 layers, runs k_tile and k_apply_tiles 1.6x faster with 4x4 columns than with 8x8; config 2's 51 layers
            // are best at 8x8)
            const double R = P.nRho * P.dRho;
            P.lv_nx = P.lv_ny = 2 * (int)std::ceil(R / P.d_sub) + 10;
            P.lv_nz = (int)std::ceil(P.nZ * P.dZ / P.d_sub) + 10;
            P.tile_sh = 3;
            while (P.tile_sh > 0 && ((size_t)P.lv_nz << (2 * P.tile_sh)) > 4096) --P.tile_sh;
            // (a handle sized for frame-by-frame calls — the default max_batch of 8 or less — takes 4x4 columns: a lone frame's k_tile
            // lasts as long as its busiest tile, the sensor's own, and a quarter of that tile is done sooner: the 500-sample callback
            // 104 -> 95 us, a dense VGA frame 184 -> 172 us; long batches keep 8x8, worth 3 % of their throughput)
            if (h->lim.max_batch <= 8 && P.tile_sh > 2) P.tile_sh = 2;
            if (knob("tile_sh", kv)) P.tile_sh = std::min(3, std::max(0, (int)kv));
            const int edge = 1 << P.tile_sh;
            P.lv_nx += edge; // (the grid's origin is snapped down to a tile boundary: frame_setup)
            P.lv_ny += edge;
            P.n_tx = (P.lv_nx + edge - 1) / edge;
            P.n_tiles = P.n_tx * ((P.lv_ny + edge - 1) / edge);
            P.tile_words = (unsigned int)(P.nPhi + 31) / 32u; // a column's ray crosses a tile once: one descriptor slot per (tile, column)
            // (a camera frame reaches about a quarter of its grid's tiles: a workgroup takes eight candidate tiles, most of them empty.
            // Measured on config 2, profiles/r4g: 121 workgroups per frame 95.1 k frames/s, 80: 97.0 k, 60: 97.7 k, 45: 98.6 k, 32: 98.3 k —
            // the kernel alone is no faster with fewer, but its 29 KB of LDS per workgroup are the other streams' kernels' room)
            h->tile_grid = (unsigned int)std::max(32, std::min(P.n_tiles, P.n_tiles / 8 + 1));
            if (knob("tile_grid", kv)) h->tile_grid = (unsigned int)std::max(1, std::min(P.n_tiles, (int)kv));
            {   // blocks a tile overlaps: an extent of e voxels starting anywhere touches at most (e - 1) / n + 2 blocks of n
                const long long cx = (edge - 1) / P.n + 2, cz = (P.lv_nz - 1) / P.n + 2;
                P.tile_combos = (unsigned int)std::min<long long>((cx * cx * cz + 3) & ~3ll, 1ll << 20);
            }
            h->tile_lds_bytes = mlm_tile_lds((unsigned int)(edge * edge * P.lv_nz), (unsigned int)P.lv_nz, P.tile_combos).total;
            h->apply_lds_bytes = (unsigned int)(edge * edge * 2 * P.lv_nz) * 9u + 16u; // (two grid heights of layers: frames of a range differ in z origin)
        }
        // (frontier mode: no tiles; its insertion times hold point index * 256 + ray step in 32 bits)
        // (k_sector lists a column's miss cells in its cell table's space; k_tile counts a voxel's misses and hits in 16 bits each:
        // no voxel may collect 2^16 cells — at most (d_sub / dRho + 2) (d_sub / dZ + 2) nPhi cell centres fall into one)
        const double cells_per_voxel = (std::ceil(P.d_sub / P.dRho) + 2) * (std::ceil(P.d_sub / P.dZ) + 2) * P.nPhi;
        h->use_sectors = P.bin_block == 256 && P.sec_lds_bytes <= 160u * 1024u - 1024u && h->lim.max_points < (1 << MLM_SEC_CNT_BITS) && P.n <= 255 &&
                         P.sec_tab <= 4u * (unsigned int)h->sec_threads && (size_t)P.nZ * P.RW * 64 <= (size_t)P.sec_tab * sizeof(MlmSecCell) && P.nZ * P.nRho < 65536 &&
                         P.nZ < 32768 /* a column record holds z << 16 | rho below its top bit, MLM_SEC_OUTER (nZ * nRho < 65536 only implies it for nRho >= 2) */ &&
                         P.nRho <= 512 /* k_chain_lanes: 128 bytes of LDS per rho; k_sector: one thread per rho */ &&
                         P.nPhi <= 32 * MLM_TILE_WORDS /* k_tile: a tile's column mask */ &&
                         (P.explore ? P.nRho <= 256
                                    : (((size_t)P.lv_nz << (2 * P.tile_sh)) <= 65536 && h->tile_lds_bytes <= 96u * 1024u && cells_per_voxel < 65536.0 &&
                                       P.n_tiles < (1 << 24) && P.lv_nz <= 1024 && h->apply_lds_bytes <= 150u * 1024u &&
                                       // (blocks one tile may overlap: their pool slots live in k_tile's LDS)
                                       P.tile_combos <= MLM_TILE_COMBOS));
        if (knob("sectors", kv)) h->use_sectors = h->use_sectors && (int)kv != 0;
        // (lean slots cost the worst-case scenes their batching — every frame overflowing its sector tables: 1.3k instead of
        // 4.4k frames/s on the "scatter" scene — so they are used on request, or when the full slots do not fit the device)
        h->lean = h->use_sectors;
        if (knob("lean_slots", kv)) h->lean = h->use_sectors && (int)kv != 0;
        if (getenv("MLM_DEBUG_CREATE"))
            fprintf(stderr, "[create] sector path %d: LDS %u bytes per column (table %u entries), frame-local grid %d x %d x %d in %d tiles of edge %d (%u bytes of LDS each)\n",
                    (int)h->use_sectors, P.sec_lds_bytes, P.sec_tab, P.lv_nx, P.lv_ny, P.lv_nz, P.n_tiles, 1 << P.tile_sh, h->tile_lds_bytes);
        if (h->use_sectors) {
            if (P.explore) { // (both instantiations: a frame on its own takes the 512-thread one where the table allows)
                HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_lds_bytes));
                HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_lds_bytes));
            } else { // (both: single frames take the 512-thread instantiation where the table allows)
                HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_lds_bytes));
                HIPCHK(h, hipFuncSetAttribute((const void *)k_sector<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_lds_bytes));
            }
            if (!P.explore) HIPCHK(h, hipFuncSetAttribute((const void *)k_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->tile_lds_bytes));
            if (!P.explore) HIPCHK(h, hipFuncSetAttribute((const void *)k_apply_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->apply_lds_bytes));
            if (P.sec_tab_big) {
                if (P.explore)
                    HIPCHK(h, hipFuncSetAttribute((const void *)k_sector_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_big_lds_bytes));
                else
                    HIPCHK(h, hipFuncSetAttribute((const void *)k_sector_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.sec_big_lds_bytes));
            }
        }
    }
    std::vector<double> cphi(P.nPhi), sphi(P.nPhi);
    for (int p = 0; p < P.nPhi; ++p) {
        const double center_phi = P.dPhi / 2 + (p * P.dPhi); // map_awareness.cpp:59
        cphi[p] = std::cos(center_phi);
        sphi[p] = std::sin(center_phi);
    }
    float *d_odds;
    float *d_s3;
    double *d_c, *d_s;
    int rc;
    if ((rc = dev_alloc(h, &d_odds, h->odds_table.size()))) return rc;
    if ((rc = dev_alloc(h, &d_s3, sigma3.size()))) return rc;
    if ((rc = dev_alloc(h, &d_c, cphi.size()))) return rc;
    if ((rc = dev_alloc(h, &d_s, sphi.size()))) return rc;
    HIPCHK(h, hipMemcpy(d_odds, h->odds_table.data(), h->odds_table.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_s3, sigma3.data(), sigma3.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_c, cphi.data(), cphi.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_s, sphi.data(), sphi.size() * sizeof(double), hipMemcpyHostToDevice));
    P.odds_table = d_odds;
    P.sigma3 = d_s3;
    {
        const size_t nb = ((h->odds_table.size() + 3) & ~(size_t)3), nw = nb / 4 + ((sigma3.size() + 3) & ~(size_t)3);
        std::vector<uint32_t> sc(nw, 0u);
        for (size_t e = 0; e < h->odds_table.size(); ++e) ((uint8_t *)sc.data())[e] = (uint8_t)mlm_sec_strength(h->odds_table[e]);
        std::memcpy(sc.data() + nb / 4, sigma3.data(), sigma3.size() * sizeof(float));
        uint32_t *d_sc;
        if ((rc = dev_alloc(h, &d_sc, nw))) return rc;
        HIPCHK(h, hipMemcpy(d_sc, sc.data(), nw * sizeof(uint32_t), hipMemcpyHostToDevice));
        P.sec_const = d_sc;
        P.sec_const_words = (unsigned int)nw;
    }
    P.cos_phi = d_c;
    P.sin_phi = d_s;

    const size_t NC = (size_t)P.nCells;
    // the emulated container can never hold more than nCells keys: bucket counts stay below the first
    // libstdc++ prime >= 2*nCells
    {
        std::__detail::_Prime_rehash_policy pol;
        h->max_buckets = pol._M_next_bkt(2 * NC + 2);
    }
    if ((rc = dev_alloc(h, &P.bkt_first, h->max_buckets))) return rc;
    P.bkt_stride = h->max_buckets;
    if ((rc = dev_alloc(h, &P.bkt64, 2 * h->max_buckets))) return rc;
    HIPCHK(h, hipMemset(P.bkt64, 0xFF, 2 * h->max_buckets * sizeof(unsigned long long)));

    // block table + pool (shared by all slots; grows on demand: grow_pool)
    if ((rc = dev_alloc(h, &P.g, 1))) return rc;
    if ((rc = alloc_pool(h, P, h->lim.max_blocks))) return rc;
    if (P.explore && (rc = dev_alloc(h, &P.bktm_first, h->max_buckets))) return rc;
    {
        MlmGlobal g0{};
        g0.fail_frame = 0x7FFFFFFF;
        HIPCHK(h, hipMemcpy(P.g, &g0, sizeof(MlmGlobal), hipMemcpyHostToDevice));
    }
    {
        // most blocks ONE frame can create: those that overlap the frame-local voxel grid (every cell of the awareness cylinder
        // falls inside it, else the frame is redone / rejected)
        auto nb = [&](int nv) { return (long long)(nv + P.n - 1) / P.n + 1; };
        const long long b = nb(P.lv_nx) * nb(P.lv_ny) * nb(P.lv_nz);
        h->frame_block_bound = (size_t)std::min<long long>(b, 1ll << 30);
        if (knob("pool_grow", kv)) h->pool_grow = (int)kv != 0;
        if (knob("graph", kv)) h->use_graph = (int)kv != 0;
        if (knob("big_grid", kv)) h->big_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("big_arm", kv)) h->big_arm_len = std::max(0, (int)kv);
        if (knob("ex_spec", kv)) h->ex_spec = (int)kv;
        if (knob("single_chain_grid", kv)) h->single_chain_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("single_rank_grid", kv)) h->single_rank_grid = (unsigned int)std::max(1, (int)kv);
        if (knob("mirror", kv)) h->mir.enabled = (int)kv != 0;
        if (knob("mirror_mb", kv)) h->mir.max_bytes = (size_t)std::max(0, (int)kv) << 20;
        if (knob("mirror_max", kv)) h->mir.max_clean = std::max(0, (int)kv), h->mir.max_dirty = std::min(h->mir.max_dirty, h->mir.max_clean);
    }
    HIPCHK(h, hipHostMalloc((void **)&h->h_g, sizeof(MlmGlobal), hipHostMallocDefault));
    std::memset(h->h_g, 0, sizeof(MlmGlobal));

    // sort buffers
    h->sort_tmp_bytes = mlm_sort_temp_bytes(NC);
    if ((rc = dev_alloc(h, &h->sk_in, NC))) return rc;
    if ((rc = dev_alloc(h, &h->sk_out, NC))) return rc;
    if ((rc = dev_alloc(h, &h->sv_in, NC))) return rc;
    if ((rc = dev_alloc(h, &h->sv_out, NC))) return rc;
    if ((rc = dev_alloc(h, (char **)&h->sort_tmp, h->sort_tmp_bytes))) return rc;

    // frame slots
    if (knob("slot_sets", kv)) h->n_sets = std::min(MLM_SETS, std::max(2, (int)kv));
    size_t NS = (size_t)h->lim.max_batch * h->n_sets; // one set being filled while the others are in flight
    h->slots.resize(NS);
    if ((rc = dev_alloc(h, &h->d_ctr_all, NS))) return rc;
    if ((rc = dev_alloc(h, &h->d_slot_tab, NS))) return rc;
    if ((rc = dev_alloc(h, &h->d_frame_tab, NS))) return rc;
    HIPCHK(h, hipMemset(h->d_ctr_all, 0, NS * sizeof(MlmCounters)));
    HIPCHK(h, hipHostMalloc((void **)&h->h_ctr_all, NS * sizeof(MlmCounters), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&h->h_frame_tab, NS * sizeof(MlmFrame), hipHostMallocDefault));
    std::memset(h->h_ctr_all, 0, NS * sizeof(MlmCounters));
    std::memset(h->h_frame_tab, 0, NS * sizeof(MlmFrame));
    if (h->cu_split > 0 || h->cu_reserve > 0) {
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int c = std::max(h->cu_split, h->cu_reserve); c < ncu; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
        for (int k = 0; k < MLM_SETS; ++k)
            HIPCHK(h, hipExtStreamCreateWithCUMask(&h->stream_as[k], (uint32_t)mask.size(), mask.data()));
    } else {
        for (int k = 0; k < MLM_SETS; ++k) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_as[k], hipStreamNonBlocking));

    }
    for (int k = 0; k < MLM_SETS; ++k) {
        HIPCHK(h, hipEventCreateWithFlags(&h->ex_counts[k], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ex_bc_done[k], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->stage_a_done[k], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->set_free[k], hipEventDisableTiming));
    }
    h->map_bytes = h->alloc_bytes;
    // frame slots sized by need (see mlm_handle::need_sized; knob "need_slots" = 0: every list at its worst case, as up to round 4)
    h->need_sized = h->lean && h->use_sectors && !P.explore;
    if (knob("need_slots", kv)) h->need_sized = h->need_sized && (int)kv != 0;
    slot_capacities(h, sigma3);
    {
        // the frame slots are most of the footprint (S1 ~1 GB, S3 ~3.8 GB each): three sets of max_batch if they fit the device
        // memory, else the same with lean slots (sector-path handles), else two sets (a few percent less throughput on config
        // 2), else the error says what would be needed
        const size_t mark = h->allocs.size();
        size_t got = 0;
        for (; got < NS; ++got)
            if ((rc = alloc_slot(h, h->slots[got], got, sigma3))) break;
        if (rc && h->use_sectors && !h->lean) {
            (void)hipGetLastError();
            for (size_t a = mark; a < h->allocs.size(); ++a)
                if (h->allocs[a]) hipFree(h->allocs[a]);
            h->allocs.resize(mark);
            h->alloc_bytes = h->map_bytes;
            for (auto &S : h->slots) // (the image staging buffers are not in `allocs`: ensure_img)
                if (S.d_img) hipFree(S.d_img);
            h->slots.assign(NS, MlmSlot{});
            h->lean = true;
            h->need_sized = !P.explore && !(knob("need_slots", kv) && (int)kv == 0);
            slot_capacities(h, sigma3);
            h->err.clear();
            rc = MLM_OK;
            for (got = 0; got < NS; ++got)
                if ((rc = alloc_slot(h, h->slots[got], got, sigma3))) break;
        }
        if (rc && got >= 2 * (size_t)h->lim.max_batch && h->n_sets > 2) {
            (void)hipGetLastError();
            // (keep the first two sets; give the partial third one back)
            size_t keep = mark;
            for (size_t i = 0; i < 2 * (size_t)h->lim.max_batch; ++i) keep = h->slots[i].alloc_end;
            for (size_t a = keep; a < h->allocs.size(); ++a)
                if (h->allocs[a]) hipFree(h->allocs[a]);
            h->allocs.resize(keep);
            for (size_t i = 2 * (size_t)h->lim.max_batch; i < NS; ++i)
                if (h->slots[i].d_img) {
                    hipFree(h->slots[i].d_img);
                    h->slots[i].d_img = nullptr;
                }
            h->n_sets = 2;
            NS = 2 * (size_t)h->lim.max_batch;
            h->slots.resize(NS);
            h->err.clear();
            rc = MLM_OK;
        }
        if (rc) {
            h->err += " (frame slots: lower mlm_limits.max_batch or max_points)";
            return rc;
        }
    }
    {
        std::vector<MlmDev> tab(NS);
        for (size_t i = 0; i < NS; ++i) tab[i] = h->slots[i].P;
        HIPCHK(h, hipMemcpy(h->d_slot_tab, tab.data(), NS * sizeof(MlmDev), hipMemcpyHostToDevice));
        if (h->lean) {
            const MlmDev &P0 = h->slots[0].P;
            if ((rc = dev_alloc(h, &h->fb_bnodes, (size_t)P0.nb_cap * P0.node_lds))) return rc;
            if ((rc = dev_alloc(h, &h->fb_pairs, (size_t)P0.nb_cap * P0.agg_lds))) return rc;
            if ((rc = dev_alloc(h, &h->fb_nodes, (size_t)MLM_RAY_LISTS * P0.node_cap))) return rc;
            for (size_t i = 0; i < NS; ++i) {
                make_fb_params(h, h->slots[i]);
                tab[i] = h->slots[i].Pfb;
            }
            if ((rc = dev_alloc(h, &h->d_slot_tab_fb, NS))) return rc;
            HIPCHK(h, hipMemcpy(h->d_slot_tab_fb, tab.data(), NS * sizeof(MlmDev), hipMemcpyHostToDevice));
            HIPCHK(h, hipEventCreateWithFlags(&h->fb_done, hipEventDisableTiming));
        }
    }
    if (!h->lean)
        for (auto &S : h->slots) S.Pfb = S.P;
    HIPCHK(h, hipDeviceSynchronize());
    if (getenv("MLM_DEBUG_CREATE"))
        fprintf(stderr, "[create] device memory: %.2f GB (%zu frame slots in %d sets of %d, %.3f GB each; the map and the shared tables %.2f GB)\n",
                h->alloc_bytes / 1e9, NS, h->n_sets, h->lim.max_batch, NS ? (h->alloc_bytes - h->map_bytes) / 1e9 / NS : 0.0, h->map_bytes / 1e9);
    return MLM_OK;
}

int mlm_destroy(mlm_handle *h) {
    if (!h) return MLM_ERR_INVALID;
    h->mu.lock(); // waits for a call in flight on another thread; the caller guarantees that none starts after this
    h->mu.unlock();
    if (!h->stream) { // creation failed before the device was touched
        delete h;
        return MLM_OK;
    }
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (void *p : h->allocs)
        if (p) hipFree(p);
    for (auto &S : h->slots) {
        if (S.d_img) hipFree(S.d_img);
        if (S.d_pix) hipFree(S.d_pix);
        if (S.d_pts) hipFree(S.d_pts);
    }
    if (h->h_ctr_all) hipHostFree(h->h_ctr_all);
    if (h->h_frame_tab) hipHostFree(h->h_frame_tab);
    for (int k = 0; k < MLM_SETS; ++k) {
        if (h->stage_a_done[k]) hipEventDestroy(h->stage_a_done[k]);
        if (h->ex_counts[k]) hipEventDestroy(h->ex_counts[k]);
        if (h->ex_bc_done[k]) hipEventDestroy(h->ex_bc_done[k]);
        if (h->set_free[k]) hipEventDestroy(h->set_free[k]);
    }
    for (int k = 0; k < MLM_SETS; ++k)
        if (h->stream_as[k]) hipStreamDestroy(h->stream_as[k]);

    for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
    if (h->h_stage) hipHostFree(h->h_stage);
    if (h->upload_ev) hipEventDestroy(h->upload_ev);
    if (h->host_read_ev) hipEventDestroy(h->host_read_ev);
    if (h->inputs_ready) hipEventDestroy(h->inputs_ready);
    if (h->fb_done) hipEventDestroy(h->fb_done);
    if (h->d_f32) hipFree(h->d_f32);
    for (int k = 0; k < MLM_SETS; ++k)
        if (h->d_img_set[k]) hipFree(h->d_img_set[k]);
    if (h->d_qpos) hipFree(h->d_qpos);
    if (h->d_qout) hipFree(h->d_qout);
    mirror_free(h);
    if (h->mir.eager_ev) hipEventDestroy(h->mir.eager_ev);
    if (h->mir.stat) hipHostFree(h->mir.stat);
    if (h->h_g) hipHostFree(h->h_g);
    for (int k = 0; k < MLM_SETS; ++k) {
        if (h->batch_done[k]) hipEventDestroy(h->batch_done[k]);
        if (h->h_gb[k]) hipHostFree(h->h_gb[k]);
    }
    for (auto &k : h->kpool) {
        hipEventDestroy(k.a);
        hipEventDestroy(k.b);
    }
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return MLM_OK;
}

int mlm_set_stream(mlm_handle *h, void *s) {
    if (!h) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = drain(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
    if (!h->inputs_ready) HIPCHK(h, hipEventCreateWithFlags(&h->inputs_ready, hipEventDisableTiming));
    return MLM_OK;
}

int mlm_integrate_depth_batch_dev(mlm_handle *h, const uint16_t *img_dev, int n_frames, size_t frame_stride, int width,
                                  int height, int row_stride, const double *q_wb, const double *t_wb) {
    if (!h || !img_dev || n_frames < 0 || !q_wb || !t_wb || width <= 0 || height <= 0 || row_stride < width)
        return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if ((long long)width * height > h->lim.max_points) {
        h->err = "frame has more points than mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const int K = h->lim.max_batch;
    for (int k0 = 0; k0 < n_frames; k0 += K) {
        const int n = std::min(K, n_frames - k0);
        for (int j = 0; j < n; ++j) {
            MlmSlot &S = cur_slot(h, j);
            S.F = MlmFrame{};
            frame_setup(h, q_wb + 4 * (size_t)(k0 + j), t_wb + 3 * (size_t)(k0 + j), S.F);
            S.F.img = img_dev + (size_t)(k0 + j) * frame_stride;
            S.F.width = width;
            S.F.height = height;
            S.F.row_stride = row_stride;
            S.F.n = width * height;
            S.mode = 0;
        }
        const int rc = run_slots(h, n);
        if (rc) return rc;
    }
    return MLM_OK;
}

int mlm_integrate_depth_batch(mlm_handle *h, const uint16_t *img_host, int n_frames, size_t frame_stride, int width,
                              int height, int row_stride, const double *q_wb, const double *t_wb) {
    if (!h || !img_host || n_frames < 0 || !q_wb || !t_wb || width <= 0 || height <= 0 || row_stride < width)
        return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if ((long long)width * height > h->lim.max_points) {
        h->err = "frame has more points than mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const int K = h->lim.max_batch;
    const size_t n_px = (size_t)row_stride * height;
    for (int k0 = 0; k0 < n_frames; k0 += K) {
        const int n = std::min(K, n_frames - k0);
        // frames that lie back to back in host memory go up in ONE copy into the slot set's batch buffer (64 copies of 0.6 MB each
        // reach a third of the link's rate); others frame by frame into the slots' own buffers
        const bool packed = frame_stride == n_px && n > 1;
        const int set = h->cur_set;
        hipStream_t up = h->async_mode ? h->stream_as[set] : h->stream; // (the stream this chunk's Stage A runs on, unless it is one frame: run_slots orders that case)
        if (packed) {
            if (h->img_set_cap[set] < (size_t)K * n_px) {
                if (h->d_img_set[set]) {
                    HIPCHK(h, hipStreamSynchronize(h->stream_as[set])); // (its last user is long done: the set is being refilled)
                    hipFree(h->d_img_set[set]);
                }
                h->d_img_set[set] = nullptr;
                h->img_set_cap[set] = 0;
                HIPCHK(h, hipMalloc((void **)&h->d_img_set[set], (size_t)K * n_px * sizeof(uint16_t)));
                h->img_set_cap[set] = (size_t)K * n_px;
            }
            HIPCHK(h, hipMemcpyAsync(h->d_img_set[set], img_host + (size_t)k0 * frame_stride, (size_t)n * n_px * sizeof(uint16_t), hipMemcpyHostToDevice, up));
        }
        for (int j = 0; j < n; ++j) {
            MlmSlot &S = cur_slot(h, j);
            if (!packed) {
                int rc = ensure_img(h, S, n_px);
                if (rc) return rc;
                HIPCHK(h, hipMemcpyAsync(S.d_img, img_host + (size_t)(k0 + j) * frame_stride, n_px * sizeof(uint16_t), hipMemcpyHostToDevice, up));
            }
            S.F = MlmFrame{};
            frame_setup(h, q_wb + 4 * (size_t)(k0 + j), t_wb + 3 * (size_t)(k0 + j), S.F);
            S.F.img = packed ? h->d_img_set[set] + (size_t)j * n_px : S.d_img;
            S.F.width = width;
            S.F.height = height;
            S.F.row_stride = row_stride;
            S.F.n = width * height;
            S.mode = 0;
        }
        int rc = borrowed_mark(h, up); // (asynchronous mode: the caller's frames are read by the copies above only)
        if (rc) return rc;
        // (the frames went up on the slot set's Stage A stream; a chunk of ONE frame in synchronous mode is submitted as the
        // single-frame graph on the MAIN stream: run_slots orders it behind the upload — without this the graph raced the copy,
        // found by tests/test_gpu_random_ops.py)
        h->last_upload = up;
        rc = run_slots(h, n);
        const int rc2 = borrowed_wait(h);
        if (rc || rc2) return rc ? rc : rc2;
    }
    return MLM_OK;
}

static int integrate_u16_dev(mlm_handle *h, const uint16_t *img_dev, int width, int height, int row_stride, const int32_t *pixel_idx_dev,
                             const int32_t *raw_dev, int n_idx, const double q_wb[4], const double t_wb[3]);
int mlm_integrate_depth_u16_dev(mlm_handle *h, const uint16_t *img_dev, int width, int height, int row_stride,
                                const int32_t *pixel_idx_dev, int n_idx, const double q_wb[4], const double t_wb[3]) {
    return integrate_u16_dev(h, img_dev, width, height, row_stride, pixel_idx_dev, nullptr, n_idx, q_wb, t_wb);
}
// raw_dev (with pixel_idx_dev): the listed pixels' depths by list position — the image itself is then never read
static int integrate_u16_dev(mlm_handle *h, const uint16_t *img_dev, int width, int height, int row_stride, const int32_t *pixel_idx_dev,
                             const int32_t *raw_dev, int n_idx, const double q_wb[4], const double t_wb[3]) {
    if (!h || !img_dev || width <= 0 || height <= 0 || row_stride < width || !q_wb || !t_wb) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if (pixel_idx_dev && n_idx < 0) return MLM_ERR_INVALID;
    const long long n = pixel_idx_dev ? n_idx : (long long)width * height;
    if (n > h->lim.max_points) {
        h->err = "frame has more points than mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    HIPCHK(h, hipSetDevice(h->device));
    MlmSlot &S = cur_slot(h, 0);
    S.F = MlmFrame{};
    frame_setup(h, q_wb, t_wb, S.F);
    S.F.img = img_dev;
    S.F.pix = pixel_idx_dev;
    S.F.raw = pixel_idx_dev ? raw_dev : nullptr;
    S.F.width = width;
    S.F.height = height;
    S.F.row_stride = row_stride;
    S.F.n = (int)n;
    S.mode = pixel_idx_dev ? 1 : 0;
    return run_slots(h, 1);
}

int mlm_integrate_depth_u16(mlm_handle *h, const uint16_t *img, int width, int height, int row_stride,
                            const int32_t *pixel_idx, int n_idx, const double q_wb[4], const double t_wb[3]) {
    if (!h || !img || width <= 0 || height <= 0 || row_stride < width) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    MlmSlot &S = cur_slot(h, 0);
    const size_t n_px = (size_t)row_stride * height;
    int rc = ensure_img(h, S, n_px);
    if (rc) return rc;
    hipStream_t up = upload_stream(h);
    h->last_upload = up;
    HIPCHK(h, hipMemcpyAsync(S.d_img, img, n_px * sizeof(uint16_t), hipMemcpyHostToDevice, up));
    if (pixel_idx) {
        if (n_idx < 0 || n_idx > h->lim.max_points) return MLM_ERR_CAPACITY;
        if ((rc = ensure_pix(h, S))) return rc;
        HIPCHK(h, hipMemcpyAsync(S.d_pix, pixel_idx, (size_t)n_idx * sizeof(int32_t), hipMemcpyHostToDevice, up));
    }
    if ((rc = borrowed_mark(h, up))) return rc;
    rc = mlm_integrate_depth_u16_dev(h, S.d_img, width, height, row_stride, pixel_idx ? S.d_pix : nullptr, n_idx, q_wb, t_wb);
    const int rc2 = borrowed_wait(h);
    return rc ? rc : rc2;
}

int mlm_integrate_callback(mlm_handle *h, const void *depth, int is_f32, int width, int height, double t_img,
                           const double odom_p[3], const double odom_q[4], const double odom_v[3], double t_odom,
                           const double imu_w[3], double t_imu, double latency, int sampled, double T_wb_out[7]) {
    if (!h || !depth || width <= 0 || height <= 0 || !odom_p || !odom_q || !odom_v || !imu_w) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    const size_t n_px = (size_t)width * height;
    if ((long long)n_px > h->lim.max_points) {
        h->err = "frame has more points than mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    HIPCHK(h, hipSetDevice(h->device));
    clk_mark(h, -1);
    // ---- pose latency compensation, mlmap.cpp:470-498 (mlm_host.h)
    double qa[4], ta[3];
    compensate_pose(odom_p, odom_q, odom_v, imu_w, t_img, t_odom, t_imu, latency, qa, ta);
    if (T_wb_out) {
        for (int i = 0; i < 4; ++i) T_wb_out[i] = qa[i];
        for (int i = 0; i < 3; ++i) T_wb_out[4 + i] = ta[i];
    }
    // ---- depth image: upload (and convert 32FC1 -> 16UC1 on the device)
    // (nothing in flight — the synchronous case: no read-back of the map-wide flags and no stream synchronisation just to learn that)
    // (frontier mode, nothing queued: the previous call has synchronised)
    int rc = (h->pending.empty() && !h->wait_ticket && (!h->P.explore || h->ex_q.empty())) ? MLM_OK : drain(h);
    if (rc) return rc;
    MlmSlot &S = cur_slot(h, 0);
    rc = ensure_img(h, S, n_px);
    if (rc) return rc;
    std::vector<int32_t> pix;
    if (sampled && h->cfg.sample_cnt > 0 && 2 * (size_t)h->cfg.sample_cnt <= (size_t)h->lim.max_points) {
        // project_depth, mlmap.cpp:311-349 (glibc rand(), v first, zeros skipped).  Only the sampled pixels are ever read
        // by the kernels, so only they travel: the host converts them (same float arithmetic as k_convert_f32_u16), a tiny
        // kernel drops them into the device image at their pixel positions.
        // The indices and depths are staged in a pinned buffer of the handle that the kernels read across the link.  In asynchronous
        // mode the frame is still in flight when this call returns: the buffer is free again because THIS call drained above before
        // sampling (see mlm_handle::h_stage for the invariant).
        const size_t want = (size_t)h->cfg.sample_cnt;
        if (h->stage_cap < 2 * want) {
            if (h->h_stage) hipHostFree(h->h_stage);
            h->h_stage = nullptr;
            h->stage_cap = 0;
            HIPCHK(h, hipHostMalloc((void **)&h->h_stage, 2 * want * sizeof(int32_t), hipHostMallocDefault));
            h->stage_cap = 2 * want;
        }
        size_t n_s = 0;
        int cnt = 0;
        const int max_iter = 2 * h->cfg.sample_cnt;
        int32_t *st_pix = h->h_stage, *st_raw = h->h_stage + want;
        // (the pixel positions a stretch ahead of their reads: every position drawn is one iteration of the reference's loop, and a
        // stretch is never longer than the iterations the loop is certain to make — rand() is called exactly as often, in the same
        // order; the reads, scattered over the frame, then miss the cache together instead of one after the other)
        size_t ahead[128];
        while (n_s < want && cnt < max_iter) {
            const size_t m = std::min<size_t>(std::min<size_t>(want - n_s, (size_t)(max_iter - cnt)), 128);
            for (size_t i = 0; i < m; ++i) {
                const size_t v = static_cast<size_t>(rand() % height);
                const size_t u = static_cast<size_t>(rand() % width);
                ahead[i] = v * (size_t)width + u;
                __builtin_prefetch(is_f32 ? (const void *)((const float *)depth + ahead[i]) : (const void *)((const uint16_t *)depth + ahead[i]));
            }
            cnt += (int)m;
            for (size_t i = 0; i < m; ++i) {
                const size_t at = ahead[i];
                const int r = is_f32 ? mlm_cv_f32_to_u16(((const float *)depth)[at]) : (int)((const uint16_t *)depth)[at];
                if (r == 0) continue;
                st_pix[n_s] = (int32_t)at;
                st_raw[n_s] = r;
                ++n_s;
            }
        }
        // (the 4 KB do not travel by a copy of their own — a call and a copy kernel, a tenth of the call: the pinned buffer is mapped
        // into the device's address space and k_bin_sectors, the list's only reader, takes the samples from it across the link)
        clk_mark(h, 0);
        rc = integrate_u16_dev(h, S.d_img, width, height, width, h->h_stage, h->h_stage + want, (int)n_s, qa, ta);
        clk_mark(h, 5);
        return rc;
    }
    std::vector<uint16_t> host_u16; // needed only by the sampler when the input is float
    if (is_f32) {
        // the converted frame stays in a buffer owned by the handle (no allocation per call)
        if (h->f32_cap < n_px) {
            if (h->d_f32) hipFree(h->d_f32);
            h->d_f32 = nullptr;
            h->f32_cap = 0;
            HIPCHK(h, hipMalloc((void **)&h->d_f32, n_px * sizeof(float)));
            h->f32_cap = n_px;
        }
        float *d_f = h->d_f32;
        hipError_t e = hipMemcpyAsync(d_f, depth, n_px * sizeof(float), hipMemcpyHostToDevice, h->stream_as[h->cur_set]);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_convert_f32_u16, dim3(grid_for(n_px)), dim3(MLM_BLOCK), 0, h->stream_as[h->cur_set], d_f, S.d_img, n_px);
            if (sampled) {
                host_u16.resize(n_px);
                e = hipMemcpyAsync(host_u16.data(), S.d_img, n_px * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream_as[h->cur_set]);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream_as[h->cur_set]); // `depth` is the caller's
        if (e != hipSuccess) {
            h->err = std::string("mlm_integrate_callback: ") + hipGetErrorString(e);
            return MLM_ERR_HIP;
        }
    } else {
        hipStream_t up = sampled ? h->stream_as[h->cur_set] : upload_stream(h); // (the sampled general path synchronises that stream below)
        HIPCHK(h, hipMemcpyAsync(S.d_img, depth, n_px * sizeof(uint16_t), hipMemcpyHostToDevice, up));
        h->last_upload = up;
        if (!sampled && (rc = borrowed_mark(h, up))) return rc;
    }
    if (sampled) { // (sample count larger than half the point capacity: the general path)
        const uint16_t *img = is_f32 ? host_u16.data() : (const uint16_t *)depth;
        const size_t want = (size_t)h->cfg.sample_cnt;
        int cnt = 0;
        const int max_iter = 2 * h->cfg.sample_cnt;
        while (pix.size() < want && cnt < max_iter) {
            cnt++;
            const size_t v = static_cast<size_t>(rand() % height);
            const size_t u = static_cast<size_t>(rand() % width);
            if (img[v * (size_t)width + u] == 0) continue;
            pix.push_back((int32_t)(v * (size_t)width + u));
        }
        if ((rc = ensure_pix(h, S))) return rc;
        if (!pix.empty())
            HIPCHK(h, hipMemcpyAsync(S.d_pix, pix.data(), pix.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream_as[h->cur_set]));
        HIPCHK(h, hipStreamSynchronize(h->stream_as[h->cur_set])); // pix is a local
    }
    clk_mark(h, 0);
    rc = mlm_integrate_depth_u16_dev(h, S.d_img, width, height, width, sampled ? S.d_pix : nullptr, (int)pix.size(), qa, ta);
    {
        const int rc2 = borrowed_wait(h);
        if (!rc) rc = rc2;
    }
    clk_mark(h, 5);
    return rc;
}

int mlm_integrate_points(mlm_handle *h, const double *xyz, int n, const double q_wb[4], const double t_wb[3]) {
    if (!h || (!xyz && n > 0) || n < 0 || !q_wb || !t_wb) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if (n > h->lim.max_points) {
        h->err = "frame has more points than mlm_limits.max_points";
        return MLM_ERR_CAPACITY;
    }
    HIPCHK(h, hipSetDevice(h->device));
    MlmSlot &S = cur_slot(h, 0);
    {
        const int rc = ensure_pts(h, S);
        if (rc) return rc;
    }
    if (n > 0) {
        hipStream_t up = upload_stream(h);
        h->last_upload = up;
        HIPCHK(h, hipMemcpyAsync(S.d_pts, xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, up));
        const int rc = borrowed_mark(h, up);
        if (rc) return rc;
    }
    S.F = MlmFrame{};
    frame_setup(h, q_wb, t_wb, S.F);
    S.F.pts = S.d_pts;
    S.F.n = n;
    S.F.width = 1;
    S.mode = 2;
    const int rc = run_slots(h, 1), rc2 = borrowed_wait(h);
    return rc ? rc : rc2;
}

int mlm_query_occupancy(mlm_handle *h, const double *pos, int n, int8_t *out) {
    return run_query(h, 0, pos, n, 0.f, 0, out, 1);
}
int mlm_query_occupancy_inflate(mlm_handle *h, const double *pos, int n, float inflate, int8_t *out) {
    return run_query(h, 1, pos, n, inflate, 0, out, 1);
}
int mlm_query_inflate_occupancy(mlm_handle *h, const double *pos, int n, int8_t *out) {
    return run_query(h, 2, pos, n, 0.f, 0, out, 1);
}
int mlm_query_odds(mlm_handle *h, const double *pos, int n, float *out) {
    return run_query(h, 3, pos, n, 0.f, 0, out, sizeof(float));
}
int mlm_query_odd_grad(mlm_handle *h, const double *pos, int n, int max_iter, double *out3) {
    if (max_iter < 0) return MLM_ERR_INVALID;
    return run_query(h, 4, pos, n, 0.f, max_iter, out3, 3 * sizeof(double));
}

int mlm_set_free_in_bound(mlm_handle *h, const double bmin[3], const double bmax[3]) {
    if (!h || !bmin || !bmax) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    // mlmap.cpp:392-396: `for (double x = min; x <= max; x += d)` — the accumulated coordinates, not i*d
    std::vector<double> ax[3];
    for (int a = 0; a < 3; ++a) {
        for (double v = bmin[a]; v <= bmax[a]; v += h->P.d_sub) {
            ax[a].push_back(v);
            if (ax[a].size() > (1u << 22)) return MLM_ERR_INVALID;
        }
        if (ax[a].empty()) return MLM_OK;
    }
    const size_t total = ax[0].size() * ax[1].size() * ax[2].size();
    const size_t na = ax[0].size() + ax[1].size() + ax[2].size();
    {
        const double wlo[3] = {ax[0].front(), ax[1].front(), ax[2].front()}, whi[3] = {ax[0].back(), ax[1].back(), ax[2].back()};
        mirror_mark_world(h, wlo, whi);
    }
    int rc = drain(h);
    if (rc) return rc;
    rc = ensure_query(h, (na + 2) / 3 + 1);
    if (rc) return rc;
    double *d = h->d_qpos;
    HIPCHK(h, hipMemcpyAsync(d, ax[0].data(), ax[0].size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + ax[0].size(), ax[1].data(), ax[1].size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + ax[0].size() + ax[1].size(), ax[2].data(), ax[2].size() * 8, hipMemcpyHostToDevice,
                             h->stream));
    hipLaunchKernelGGL(k_set_free, dim3(grid_for(total)), dim3(MLM_BLOCK), 0, h->stream, h->P, d, (int)ax[0].size(),
                       d + ax[0].size(), (int)ax[1].size(), d + ax[0].size() + ax[1].size(), (int)ax[2].size());
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int mlm_inflate_map(mlm_handle *h, const double ct_pos[3]) {
    if (!h || !ct_pos) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int R = h->cfg.inflate_n, G = h->cfg.inflate_global_n;
    if (R < 0 || G < 0 || R >= h->P.n || G > 16) {
        h->err = "inflate_n must be < subbox_n and inflate_global_n <= 16";
        return MLM_ERR_UNSUPPORTED;
    }
    int rc = drain(h);
    if (rc) return rc;
    rc = ensure_free_blocks_idle(h, (size_t)(2 * G + 3) * (2 * G + 3) * (2 * G + 3)); // (neighbour blocks are allocated: inflate_atpos)
    if (rc) return rc;
    // get_global_idx(ct_pos) (mlmap.cpp:289, map_local.h:148-152)
    const int cgx = (int)std::floor(ct_pos[0] / h->P.d_glb), cgy = (int)std::floor(ct_pos[1] / h->P.d_glb),
              cgz = (int)std::floor(ct_pos[2] / h->P.d_glb);
    const size_t total = (size_t)(2 * G + 1) * (2 * G + 1) * (2 * G + 1) * (size_t)h->P.cells;
    {
        // (the cube of blocks and the neighbours its dilation may reach; a position whose block index leaves int: anywhere)
        const double c[3] = {std::floor(ct_pos[0] / h->P.d_glb), std::floor(ct_pos[1] / h->P.d_glb), std::floor(ct_pos[2] / h->P.d_glb)};
        if (std::fabs(c[0]) < 1e6 && std::fabs(c[1]) < 1e6 && std::fabs(c[2]) < 1e6) {
            const int lo[3] = {cgx - G - 1, cgy - G - 1, cgz - G - 1}, hi[3] = {cgx + G + 1, cgy + G + 1, cgz + G + 1};
            mirror_mark_box(h, lo, hi);
        } else {
            mirror_mark_all(h);
        }
    }
    hipLaunchKernelGGL(k_inflate_reset, dim3(grid_for(total)), dim3(MLM_BLOCK), 0, h->stream, h->P, cgx, cgy, cgz, G);
    hipLaunchKernelGGL(k_inflate_spread, dim3(grid_for(total)), dim3(MLM_BLOCK), 0, h->stream, h->P, cgx, cgy, cgz, G, R,
                       0.1 /* flate_height, map_local.h:65 */);
    HIPCHK(h, hipGetLastError());
    rc = read_global(h);
    if (rc) return rc;
    if (h->h_g->err) {
        h->err = "block pool or block hash table full (raise mlm_limits.max_blocks)";
        clear_device_error(h);
        return MLM_ERR_CAPACITY;
    }
    return MLM_OK;
}

int mlm_block_count(mlm_handle *h, int *n_out) {
    if (!h || !n_out) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = read_global(h);
    if (rc) return rc;
    *n_out = (int)std::min<unsigned int>(h->h_g->n_blocks, (unsigned int)h->P.max_blocks);
    return MLM_OK;
}

int mlm_export_blocks(mlm_handle *h, int cap, int32_t *keys, float *log_odds, uint8_t *occ, uint8_t *infl, int *n_out) {
    if (!h || cap < 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    int n = 0;
    int rc = mlm_block_count(h, &n);
    if (rc) return rc;
    if (n_out) *n_out = n;
    const size_t m = (size_t)std::min(n, cap);
    const size_t C = (size_t)h->P.cells;
    if (m == 0) return MLM_OK;
    if (keys) HIPCHK(h, hipMemcpyAsync(keys, h->P.block_keys, m * 3 * sizeof(int), hipMemcpyDefault, h->stream));
    if (log_odds) HIPCHK(h, hipMemcpyAsync(log_odds, h->P.log_odds, m * C * sizeof(float), hipMemcpyDefault, h->stream));
    if (occ) HIPCHK(h, hipMemcpyAsync(occ, h->P.occ, m * C, hipMemcpyDefault, h->stream));
    if (infl) HIPCHK(h, hipMemcpyAsync(infl, h->P.infl, m * C, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int mlm_export_block_flags(mlm_handle *h, int cap, uint8_t *collapsed, int *n_out) {
    if (!h || cap < 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    int n = 0;
    int rc = mlm_block_count(h, &n);
    if (rc) return rc;
    if (n_out) *n_out = n;
    const size_t m = (size_t)std::min(n, cap);
    if (!m || !collapsed) return MLM_OK;
    if (h->P.explore)
        HIPCHK(h, hipMemcpy(collapsed, h->P.blk_collapsed, m, hipMemcpyDefault));
    else
        std::memset(collapsed, 0, m);
    return MLM_OK;
}

int mlm_export_frontier(mlm_handle *h, int cap, int32_t *keys_cell, int *n_out) {
    if (!h || cap < 0 || (cap > 0 && !keys_cell)) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if (n_out) *n_out = 0;
    if (!h->P.explore) return MLM_OK;
    int nb = 0;
    int rc = mlm_block_count(h, &nb);
    if (rc) return rc;
    int32_t *d_out = nullptr;
    unsigned int *d_cnt = nullptr;
    HIPCHK(h, hipMalloc((void **)&d_out, std::max<size_t>((size_t)cap * 4 * sizeof(int32_t), 16)));
    HIPCHK(h, hipMalloc((void **)&d_cnt, sizeof(unsigned int)));
    HIPCHK(h, hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), h->stream));
    if (nb > 0)
        hipLaunchKernelGGL(k_ex_export_frontier, dim3(512), dim3(MLM_BLOCK), 0, h->stream, h->P, (unsigned int)nb, d_out,
                           (unsigned int)cap, d_cnt);
    unsigned int cnt = 0;
    hipError_t e = hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && cnt && cap)
        e = hipMemcpy(keys_cell, d_out, (size_t)std::min<unsigned int>(cnt, (unsigned int)cap) * 4 * sizeof(int32_t), hipMemcpyDefault);
    hipFree(d_out);
    hipFree(d_cnt);
    if (e != hipSuccess) {
        h->err = std::string("mlm_export_frontier: ") + hipGetErrorString(e);
        return MLM_ERR_HIP;
    }
    if (n_out) *n_out = (int)cnt;
    return MLM_OK;
}

static int export_points(mlm_handle *h, int cap_points, float *xyz, int *n_out, int which, const char *what) {
    HIPCHK(h, hipSetDevice(h->device));
    if (n_out) *n_out = 0;
    if (which == 1 && !h->P.explore) return MLM_OK; // frontier sets only exist with use_exploration_frontiers
    int nb = 0;
    int rc = mlm_block_count(h, &nb);
    if (rc) return rc;
    float *d_xyz = nullptr;
    unsigned int *d_cnt = nullptr;
    HIPCHK(h, hipMalloc((void **)&d_xyz, std::max<size_t>((size_t)cap_points * 3 * sizeof(float), 16)));
    hipError_t e = hipMalloc((void **)&d_cnt, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), h->stream);
    if (e == hipSuccess && nb > 0)
        hipLaunchKernelGGL(k_export_global, dim3(1024), dim3(MLM_BLOCK), 0, h->stream, h->P, (unsigned int)nb, d_xyz,
                           (unsigned int)cap_points, d_cnt, which);
    unsigned int cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && cnt && cap_points)
        e = hipMemcpy(xyz, d_xyz, (size_t)std::min<unsigned int>(cnt, (unsigned int)cap_points) * 3 * sizeof(float),
                      hipMemcpyDefault);
    hipFree(d_xyz);
    if (d_cnt) hipFree(d_cnt);
    if (e != hipSuccess) {
        h->err = std::string(what) + ": " + hipGetErrorString(e);
        return MLM_ERR_HIP;
    }
    if (n_out) *n_out = (int)cnt;
    return MLM_OK;
}

int mlm_export_global_map(mlm_handle *h, int cap_points, float *xyz, int *n_out) {
    if (!h || cap_points < 0 || (cap_points > 0 && !xyz)) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    return export_points(h, cap_points, xyz, n_out, 0, "mlm_export_global_map");
}

int mlm_export_frontier_points(mlm_handle *h, int cap_points, float *xyz, int *n_out) {
    if (!h || cap_points < 0 || (cap_points > 0 && !xyz)) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    return export_points(h, cap_points, xyz, n_out, 1, "mlm_export_frontier_points");
}

int mlm_import_blocks(mlm_handle *h, int n, const int32_t *keys, const float *log_odds, const uint8_t *occ, const uint8_t *infl,
                      const uint8_t *collapsed) {
    if (!h || n < 0 || (n > 0 && !keys)) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc || n == 0) return rc;
    mirror_mark_all(h);
    rc = ensure_free_blocks_idle(h, (size_t)n);
    if (rc) return rc;
    const size_t C = (size_t)h->P.cells, N = (size_t)n;
    // one staging buffer: keys | slots | log_odds | occ | infl | collapsed (sources may be host or device memory)
    const size_t o_keys = 0, o_slots = o_keys + N * 12, o_lo = (o_slots + N * 4 + 15) & ~(size_t)15, o_occ = o_lo + N * C * 4,
                 o_infl = o_occ + N * C, o_col = o_infl + N * C, total = o_col + N;
    char *d = nullptr;
    HIPCHK(h, hipMalloc((void **)&d, total));
    hipError_t e = hipMemcpyAsync(d + o_keys, keys, N * 12, hipMemcpyDefault, h->stream);
    if (e == hipSuccess && log_odds) e = hipMemcpyAsync(d + o_lo, log_odds, N * C * 4, hipMemcpyDefault, h->stream);
    if (e == hipSuccess && occ) e = hipMemcpyAsync(d + o_occ, occ, N * C, hipMemcpyDefault, h->stream);
    if (e == hipSuccess && infl) e = hipMemcpyAsync(d + o_infl, infl, N * C, hipMemcpyDefault, h->stream);
    if (e == hipSuccess && collapsed) e = hipMemcpyAsync(d + o_col, collapsed, N, hipMemcpyDefault, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_import_slots, dim3(grid_for(N)), dim3(MLM_BLOCK), 0, h->stream, h->P, (const int32_t *)(d + o_keys), n,
                           (int *)(d + o_slots));
        hipLaunchKernelGGL(k_import_cells, dim3(std::min<unsigned int>(4096u, grid_for(N * C))), dim3(MLM_BLOCK), 0, h->stream, h->P,
                           (const int *)(d + o_slots), n, log_odds ? (const float *)(d + o_lo) : nullptr,
                           occ ? (const uint8_t *)(d + o_occ) : nullptr, infl ? (const uint8_t *)(d + o_infl) : nullptr,
                           collapsed ? (const uint8_t *)(d + o_col) : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) {
        h->err = std::string("mlm_import_blocks: ") + hipGetErrorString(e);
        return MLM_ERR_HIP;
    }
    rc = read_global(h);
    if (rc) return rc;
    if (h->h_g->err) {
        h->err = "block pool or block hash table full (raise mlm_limits.max_blocks)";
        clear_device_error(h);
        return MLM_ERR_CAPACITY;
    }
    return MLM_OK;
}

int mlm_merge_pack(mlm_handle *h, const int32_t *keys_dev, int n, float *log_odds_dev, uint8_t *seen_dev) {
    if (!h || n < 0 || (n > 0 && (!keys_dev || !log_odds_dev || !seen_dev))) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = drain(h);
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(k_merge_pack, dim3(std::min<unsigned int>(8192u, grid_for((size_t)n * h->P.cells))), dim3(MLM_BLOCK), 0, h->stream,
                       h->P, keys_dev, n, log_odds_dev, seen_dev);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int mlm_merge_finish(mlm_handle *h, float *log_odds_dev, const uint8_t *seen_dev, size_t n_cells, uint8_t *occ_dev) {
    if (!h || (n_cells > 0 && (!log_odds_dev || !seen_dev || !occ_dev))) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    if (n_cells == 0) return MLM_OK;
    hipLaunchKernelGGL(k_merge_finish, dim3(std::min<unsigned int>(8192u, grid_for(n_cells))), dim3(MLM_BLOCK), 0, h->stream, h->P,
                       log_odds_dev, seen_dev, n_cells, occ_dev);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int mlm_query_odds_at(mlm_handle *h, const int32_t *glb_id, const int32_t *subbox_id, int n, float *out) {
    if (!h || !glb_id || !subbox_id || !out || n < 0) return MLM_ERR_INVALID;
    for (int i = 0; i < n; ++i) // std::vector::operator[] out of range is undefined behaviour in the reference
        if (subbox_id[i] < 0 || subbox_id[i] >= h->P.cells) return MLM_ERR_INVALID;
    if (n == 0) return MLM_OK;
    MLM_LOCK(h);
    if (mirror_wanted(h, 3, n, 0)) { // (mlm_mirror.h)
        const int rc = mirror_sync(h);
        if (rc == MLM_OK) {
            for (int i = 0; i < n; ++i) out[i] = mir_odd_at(h, glb_id[3 * (size_t)i], glb_id[3 * (size_t)i + 1], glb_id[3 * (size_t)i + 2], subbox_id[i]);
            h->mir.n_host_queries += n;
            return MLM_OK;
        }
        if (!h->mir.alloc_failed) return rc;
    }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc) return rc;
    rc = ensure_query(h, (size_t)n);
    if (rc) return rc;
    int32_t *d_g = (int32_t *)h->d_qpos, *d_s = d_g + 3 * (size_t)n;
    HIPCHK(h, hipMemcpyAsync(d_g, glb_id, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_s, subbox_id, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_query_odds_at, dim3(grid_for((size_t)n)), dim3(MLM_BLOCK), 0, h->stream, h->P, d_g, d_s, n, (float *)h->d_qout);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_qout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLM_OK;
}

int mlm_sync(mlm_handle *h) {
    if (!h) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    return drain(h);
}

int mlm_set_async(mlm_handle *h, int on) {
    if (!h) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = drain(h);
    h->async_mode = on != 0;
    return rc;
}

int mlm_set_host_mirror_limit(mlm_handle *h, size_t max_bytes) {
    if (!h) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    MlmMirror &M = h->mir;
    M.max_bytes = max_bytes;
    if (M.alloc_failed) return MLM_OK; // (the planes could not be pinned at all: the kernel path stays)
    const size_t held = M.cap * ((size_t)h->P.cells * 6 + 13);
    if (held > max_bytes) {
        HIPCHK(h, hipSetDevice(h->device));
        mirror_free(h);
        M.over_limit = true;
        M.enabled = false;
    } else if (M.over_limit && max_bytes > 0) { // (a higher limit: the next small query tries again)
        M.over_limit = false;
        M.enabled = true;
    }
    return MLM_OK;
}

int mlm_get_frame_stats(mlm_handle *h, mlm_frame_stats *out) {
    if (!h || !out) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    *out = h->stats;
    out->n_host_queries = h->mir.n_host_queries;
    out->n_mirror_refreshes = h->mir.n_refresh;
    out->n_mirror_blocks = h->mir.n_copied;
    out->device_bytes = (int64_t)h->alloc_bytes;
    out->n_slot_grows = h->n_slot_grows;
    return MLM_OK;
}

int mlm_get_awareness_hits(mlm_handle *h, int cap, uint32_t *cell_idx, float *odds, uint32_t *t_first, int *n_out) {
    if (!h || cap < 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    {
        const int rc = drain(h);
        if (rc) return rc;
    }
    if (odds && !h->P.record_awareness && !h->P.explore) { // (the kernels keep the odds themselves only for this read-back)
        h->err = "mlm_limits.record_awareness was not set";
        return MLM_ERR_INVALID;
    }
    const MlmDev &P = eff_params(h, h->slots[(size_t)h->last_slot]);
    const size_t n = (size_t)h->stats.n_hit_cells;
    if (n_out) *n_out = (int)n;
    const size_t m = std::min<size_t>(n, (size_t)cap);
    if (m == 0) return MLM_OK;
    if (cell_idx) HIPCHK(h, hipMemcpy(cell_idx, P.hl_cell, m * 4, hipMemcpyDeviceToHost));
    if (odds) HIPCHK(h, hipMemcpy(odds, P.hl_odd, m * 4, hipMemcpyDeviceToHost));
    if (t_first) HIPCHK(h, hipMemcpy(t_first, P.hl_t, m * 4, hipMemcpyDeviceToHost));
    return MLM_OK;
}

int mlm_get_awareness_misses(mlm_handle *h, int cap, uint32_t *cell_idx, int *n_out) {
    if (!h || cap < 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    if (!h->P.record_awareness && !h->P.explore) {
        h->err = "mlm_limits.record_awareness was not set";
        return MLM_ERR_INVALID;
    }
    HIPCHK(h, hipSetDevice(h->device));
    const MlmDev &P = h->slots[(size_t)h->last_slot].P;
    const size_t n = (size_t)h->stats.n_miss_cells;
    if (n_out) *n_out = (int)n;
    const size_t m = std::min<size_t>(n, (size_t)cap);
    if (m && cell_idx) HIPCHK(h, hipMemcpy(cell_idx, P.explore ? P.ex_cell : P.ml_cell, m * 4, hipMemcpyDeviceToHost));
    return MLM_OK;
}

int mlm_get_T_ls(mlm_handle *h, double q[4], double t[3]) {
    if (!h || !q || !t) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    const MlmFrame &F = h->slots[(size_t)h->last_slot].F;
    for (int i = 0; i < 4; ++i) q[i] = F.q_ls[i];
    for (int i = 0; i < 3; ++i) t[i] = F.t_ls[i];
    return MLM_OK;
}

int mlm_get_odds_table(mlm_handle *h, float *out) {
    if (!h || !out) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    std::memcpy(out, h->odds_table.data(), h->odds_table.size() * sizeof(float));
    return MLM_OK;
}

int mlm_enable_kernel_timing(mlm_handle *h, int on) {
    if (!h) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = drain(h);
    h->timing = on;
    h->ktimes.clear();
    h->kpool_used = 0;
    return rc;
}

int mlm_set_timed_kernel(mlm_handle *h, const char *name, int every) {
    if (!h || !name || every < 1) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    h->timed_kernel = name;
    h->timed_every = (unsigned int)every;
    h->timed_count = 0;
    return MLM_OK;
}

int mlm_get_kernel_times(mlm_handle *h, int cap, const char **names, float *ms, int *n_out) {
    if (!h || cap < 0) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    HIPCHK(h, hipSetDevice(h->device));
    {
        const int rc = drain(h);
        if (rc) return rc;
    }
    HIPCHK(h, hipDeviceSynchronize());
    const int n = (int)h->ktimes.size();
    if (n_out) *n_out = n;
    for (int i = 0; i < std::min(n, cap); ++i) {
        float t = 0.f;
        hipEventElapsedTime(&t, h->ktimes[i].a, h->ktimes[i].b);
        if (names) names[i] = h->ktimes[i].name;
        if (ms) ms[i] = t;
    }
    if (h->timing >= 2 && cap >= n) { // accumulate modes: reading the list consumes it
        h->ktimes.clear();
        h->kpool_used = 0;
    }
    return MLM_OK;
}

} // extern "C"

#ifdef MLM_PHASE_PROF
// diagnostic build only: the same for k_tile (thread 0's clock per phase, summed over tiles and launches; cleared)
extern "C" int mlm_debug_tile_phases(unsigned long long *out8) {
    hipDeviceSynchronize();
    std::vector<unsigned long long> v((size_t)4096 * 8);
    if (hipMemcpyFromSymbol(v.data(), HIP_SYMBOL(g_mlm_tphase), v.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out8[k] = 0;
    for (size_t i = 0; i < v.size(); ++i) out8[i & 7] += v[i];
    std::fill(v.begin(), v.end(), 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mlm_tphase), v.data(), v.size() * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
// diagnostic build only: workgroup spans of k_sector [0..3] and k_tile [4..7] — first start, last end, longest workgroup (10 ns
// ticks), workgroups — since the last call (cleared)
extern "C" int mlm_debug_spans(unsigned long long *out32) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mlm_span), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[32];
    for (int k = 0; k < 32; ++k) z[k] = (k & 3) == 0 ? ~0ull : 0ull;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mlm_span), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int mlm_debug_wg_times(unsigned long long *out) { // [2][2048][2]
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlm_wg), 2 * 2048 * 2 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
// diagnostic build only: sum over blocks and clear the per-phase cycle counts of k_bin_points
extern "C" int mlm_debug_phases(unsigned long long *out16) {
    hipDeviceSynchronize();
    std::vector<unsigned long long> v((size_t)MLM_PHASE_BLOCKS * 16);
    if (hipMemcpyFromSymbol(v.data(), HIP_SYMBOL(g_mlm_phase), v.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
    for (int k = 0; k < 16; ++k) out16[k] = 0;
    for (size_t i = 0; i < v.size(); ++i) out16[i & 15] += v[i];
    std::fill(v.begin(), v.end(), 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mlm_phase), v.data(), v.size() * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
