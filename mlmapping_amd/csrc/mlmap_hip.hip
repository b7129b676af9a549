// mlmap_hip.hip — host side of libmlmap_hip.so: the C ABI of include/mlmap_hip.h.
//
// Host work per frame is O(1): the pose composition T_ls (map_awareness.cpp:184-186, mlm_host.h), the bookkeeping of
// the emulated libstdc++ rehash policy of hit_idx_odds_hashmap, and kernel launches.  All per-point / per-cell /
// per-voxel work runs in the kernels of mlm_kernels_sector.h (default path) and mlm_kernels.h / mlm_kernels_explore.h.  There is no CPU fallback: every entry point fails with
// MLM_ERR_HIP when the device is unavailable.
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

using namespace mlm_host;

extern "C" size_t mlm_sort_temp_bytes(size_t n);
extern "C" int mlm_sort_pairs_u64_u32(void *temp, size_t temp_bytes, const unsigned long long *kin,
                                      unsigned long long *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                                      hipStream_t stream);

namespace {

#ifndef MLM_SETS
#define MLM_SETS 3
#endif
// MLM_SETS slot sets: batches in flight (one being filled, one in Stage A, one draining).  Measured on config 2 with the sector
// path: round 2 44.7k frames/s with 2, 48.4k with 3, 45.3k with 4; round 3 74.0k / 80.9k / 78.7k

struct KernelTime {
    const char *name;
    hipEvent_t a, b;
};

// Test and experiment knobs (mlm_debug_set): named integers that mlm_create reads — launch geometries, forced fall-backs,
// simulated allocation failures.  Process-wide, not part of the drop-in contract; the library reads no environment variable
// for them (only the three diagnostic switches MLM_DEBUG_CREATE / MLM_DEBUG_ALLOC / MLM_DEBUG_DRAIN, which print).
const char *const kKnobNames[] = {"agg_lds", "big_arm", "big_grid", "bin_block", "bin_strips", "chain_grid", "collect_grid", "cu_reserve", "cu_split",
                                  "debug_fail_slot", "expand_block", "graph", "graph_copies", "lean_slots", "logit_exact", "node_lds", "pool_grow",
                                  "rank_grid", "sc_block", "sc_grid", "sec_backoff", "sec_fail_every", "sec_tab", "sec_tab_big", "sec_threads",
                                  "sectors", "slot_sets", "sort_block", "sort_grid", "tile_grid", "tile_sh"};
struct KnobStore {
    std::mutex mu;
    std::unordered_map<std::string, long long> v;
};
KnobStore &knob_store() {
    static KnobStore k;
    return k;
}
bool knob(const char *name, long long &out) {
    KnobStore &k = knob_store();
    std::lock_guard<std::mutex> lock(k.mu);
    auto it = k.v.find(name);
    if (it == k.v.end()) return false;
    out = it->second;
    return true;
}

} // namespace

// One frame in flight: private awareness scratch + unique-hit list, its own stream for Stage A.
struct MlmSlot {
    MlmDev P{};               // shared map pointers + this slot's scratch pointers
    MlmCounters *h_ctr = nullptr; // pinned mirror of P.ctr (points into mlm_handle::h_ctr_all)
    MlmFrame F{};
    int mode = 0;
    int seq = 0;              // sequence number of the frame it currently holds
    unsigned int nb = 0;      // k_bin_points blocks
    unsigned int ex_um = 0;   // frontier mode: unique miss cells of the frame it holds
    size_t alloc_end = 0;     // mlm_handle::allocs.size() once this slot was allocated
    bool sector = false;      // the frame it holds went through the sector path (Stage A and the frame-local voxel grid)
    bool keys_exact = false;  // hl_key of the frame it holds was written by order_hits_exact (a replay must not recompute it:
                              // the emulated container's policy state has moved on)
    uint16_t *d_img = nullptr; // staging for host images
    size_t img_cap = 0;
    int32_t *d_pix = nullptr;
    double *d_pts = nullptr;
};

struct mlm_handle {
    int device = 0;
    hipStream_t stream = nullptr; // main stream: Stage B/C (ordered across frames), queries
    bool own_stream = true;
    mlm_config cfg{};
    mlm_limits lim{};
    MlmDev P{};                   // template: constants + shared pointers
    std::vector<MlmSlot> slots;
    int last_slot = 0;
    Q4 q_bs{};
    D3 t_bs{};
    std::vector<float> odds_table;
    std::vector<void *> allocs;
    double *d_qpos = nullptr; // query positions
    void *d_qout = nullptr;
    size_t q_cap = 0;
    // sort buffers (rehash frames only)
    unsigned long long *sk_in = nullptr, *sk_out = nullptr;
    uint32_t *sv_in = nullptr, *sv_out = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    MlmGlobal *h_g = nullptr; // pinned mirror of P.g
    // emulated libstdc++ state of awareness_map->hit_idx_odds_hashmap: bucket count + rehash policy.
    // clear() (map_awareness.cpp:178) keeps both, so they persist across frames.
    size_t hit_n_bkt = 1;
    std::__detail::_Prime_rehash_policy hit_pol;
    size_t max_buckets = 0;
    // the same for awareness_map->miss_idx_set (only tracked in frontier mode, where its iteration order matters)
    size_t miss_n_bkt = 1;
    std::__detail::_Prime_rehash_policy miss_pol;
    mlm_frame_stats stats{};
    std::string err;
    int timing = 0;
    std::vector<KernelTime> ktimes;
    std::vector<KernelTime> kpool;
    size_t kpool_used = 0;
    // submission state (see submit_batch / drain)
    std::vector<MlmSlot *> pending;
    int next_seq = 0;
    int cur_set = 0;
    unsigned int expand_block = 256;         // threads per k_expand_nodes block (128 and 64 measured slower)
    unsigned int sort_block = 256;           // threads per k_sort_contribs<1024> block
    unsigned int sort_grid = 256;            // blocks per frame of k_sort_contribs<1024> in a batch
    unsigned int rank_grid = 128;            // blocks per frame of k_rank in a batch (config 2 with two cells per wave: 87.5k frames/s, 64: 87.7k, 256: 86.4k,
                                             // 512: 84.7k; MLM_RANK_GRID)
    // Lean frame slots (sector-path handles): the three large buffers only the cell-table Stage A uses — the block slices of
    // contribution nodes sized for its LDS overflow, the (block, cell) pairs and the node lists — exist ONCE per handle instead
    // of once per slot; a cell-table Stage A (a frame's fall-back, or a batch submitted while the sector path backs off) then
    // runs one frame at a time, ordered by fb_done across the sets' streams.
    bool lean = false;
    MlmDev *d_slot_tab_fb = nullptr;         // the slots' constants with those three buffers pointing at the shared ones
    hipEvent_t fb_done = nullptr;
    size_t map_bytes = 0;                    // ... of it the map, its tables and the buffers shared by all frame slots
    bool debug_alloc = getenv("MLM_DEBUG_ALLOC") != nullptr;
    size_t alloc_bytes = 0;                  // device memory the handle holds (MLM_DEBUG_CREATE prints it)
    unsigned int chain_grid = 0;             // blocks per frame of k_chain_lanes (0: from the last confirmed frame's ranked cells; MLM_CHAIN_GRID)
    unsigned int collect_grid = 16;          // blocks per sub-list of k_collect_hits (grid-stride loop)
    unsigned int sc_block = 64;              // threads per block of the per-frame apply kernels: single-wave blocks are placed as soon as any wave
                                             // slot frees between Stage A's workgroups (config 2: 66.5k frames/s, 128: 64.1k, 256: 57.2k)
    bool sc_grid_fixed = false;              // MLM_SC_GRID given: do not adapt
    unsigned int sc_grid = 80;               // blocks per list of k_apply_voxelize (grid-stride loops; 40..120 measured equal, 160 3 % slower)
    std::string timed_kernel = "k_bin_points"; // the kernel bracketed in timing mode 3 ...
    unsigned int timed_every = 1, timed_count = 0; // ... on every timed_every-th launch
    int n_sets = MLM_SETS;     // slot sets in use (2 when three do not fit the device memory)
    int set_pending[MLM_SETS] = {};
    bool async_mode = false;
    int cu_split = 0;
    int cu_reserve = 0;
    float *d_f32 = nullptr;                  // staging of a 32FC1 frame (mlm_integrate_callback)
    size_t f32_cap = 0;
    hipStream_t stream_as[MLM_SETS] = {};    // Stage A of whole batches, one stream per slot set (overlaps Stage B/C of the
                                             // previous batch and the tails of the other set's Stage A kernels)
    hipEvent_t stage_a_done[MLM_SETS] = {};
    hipEvent_t set_free[MLM_SETS] = {}; // main stream finished reading the set's Stage A outputs
    MlmDev *d_slot_tab = nullptr;            // [2K] device copies of the slots' MlmDev
    MlmFrame *d_frame_tab = nullptr;         // [2K] per-frame parameters of the frames in flight
    MlmFrame *h_frame_tab = nullptr;         // pinned staging of d_frame_tab
    MlmCounters *d_ctr_all = nullptr;        // [2K] contiguous per-slot counters
    MlmCounters *h_ctr_all = nullptr;        // pinned mirror
    hipEvent_t batch_done[MLM_SETS] = {};
    hipEvent_t inputs_ready = nullptr;       // caller-supplied stream only: orders Stage A after the caller's work on it
    MlmGlobal *h_gb[MLM_SETS] = {}; // pinned snapshots of P.g taken at the end of each batch
    bool use_sectors = true;   // Stage A by azimuth sector (mlm_kernels_sector.h); MLM_SECTORS=0: the cell-table path
    struct ExBatch {
        int set, n;
        bool bc_enqueued;
    };
    std::deque<ExBatch> ex_q;   // frontier mode, asynchronous submission: batches in flight, oldest first
    hipEvent_t ex_counts[MLM_SETS] = {}, ex_bc_done[MLM_SETS] = {};
    MlmSlot *ex_tail = nullptr; // frontier mode: the frame whose miss phase + release scan ride with the next frame's ordering launches
    int ex_tag = 0;            // frontier mode: per-frame tag of the bucket-first tables (k_ex_order_min)
    unsigned int ex_frame_no = 0; // frontier mode: frames submitted (MlmFrame::pad2)
    int sector_backoff_len = 16; // (MLM_SEC_BACKOFF)
    int sector_backoff = 0;    // batches that go straight to the cell-table path after a sector overflow (the scene does not fit the sectors' LDS tables: do not pay for both paths)
    long long n_sector_fallbacks = 0; // frames redone by the cell-table path because a sector's LDS tables overflowed
    std::recursive_mutex mu;   // serialises the entry points of this handle (see MLM_LOCK)
    long long n_spec_miss = 0; // frames replayed because the speculative "no rehash" plan did not hold
    bool pool_grow = true;     // the block pool grows on demand (MLM_POOL_GROW=0: fixed at mlm_limits.max_blocks, MLM_ERR_CAPACITY when full)
    size_t frame_block_bound = 0; // most blocks one frame can create
    unsigned int tile_lds_bytes = 0; // dynamic LDS of k_tile
    int ov_heavy = 0;                // confirmed batches in a row whose last frame had many overflowed columns
    bool want_widen = false;         // ... the cell table is doubled before the next submission (widen_sec_tab)
    int sec_threads = 512;           // threads of a column's workgroup (k_sector<.., 256 | 512>; MLM_SEC_THREADS)
    unsigned int wait_ticket = 0;    // nonzero: the single-frame graph in flight ends by writing this into h_g->pad (pinned)
    unsigned int single_apply_grid = 256; // workgroups of k_apply_single: a VGA frame's ~33 k voxel records, one per thread (more: in turns)
    unsigned int tile_grid = 0;      // workgroups of k_tile per frame of a batch: they walk the frame's touched tiles (MLM_TILE_GRID)
    unsigned int apply_lds_bytes = 0; // dynamic LDS of k_apply_tiles: 9 bytes per voxel of a tile
    unsigned int big_grid = 256;     // workgroups of k_sector_big per batch: one per CU (MLM_BIG_GRID)
    int big_armed = 0;               // batches (single frames) for which the pass with the large cell table stays scheduled
    int big_armed_from = 0;          // first frame (sequence number; frontier mode: frame number) submitted after it was scheduled
    int big_arm_len = 64;            // (MLM_BIG_ARM: 0 never schedules it)
    // Single frames in synchronous mode — the reference's own call pattern, one frame per depth callback (mlmap.cpp:463-507) — are
    // submitted as ONE replay of a HIP graph on the main stream (parameter upload, counter reset, six kernels, counter
    // read-back) instead of a dozen launches and copies spread over two streams: the call's cost is launch latency, not work.
    struct SingleGraph {
        int mode, width, height, base, big;
        unsigned int nb, sec_tab; // (sec_tab stands for the column kernel's launch geometry: threads and LDS follow from it)
        size_t n_bkt;
        hipGraphExec_t exec;
    };
    std::vector<SingleGraph> graphs;
    bool use_graph = true;       // (MLM_GRAPH=0: always the general submission)
    int graph_copies = 2;        // 0: issued eagerly around the graph, 2: as kernels of the graph reading / writing the pinned buffers; 1: the graph holds the parameter upload / counter reset / read-back as memcpy and memset nodes (MLM_GRAPH_COPIES=0: issued eagerly around it)
    hipStream_t last_upload = nullptr; // the stream the current call's inputs were uploaded on (run_slots orders Stage A behind it)
    hipEvent_t upload_ev = nullptr;
    long long n_graph_launches = 0;
    int32_t *h_stage = nullptr;  // pinned staging of the callback's sampled pixels (indices, then raw depths)
    size_t stage_cap = 0;
    long long n_pool_grows = 0;
    int bin_strips = 1;          // strips per k_bin_sectors workgroup for the dense frames of a batch (MLM_BIN_STRIPS=2|4: experiments)
    size_t grow_failed_at = 0;   // a pool of this many blocks did not fit the device (grow_pool does not retry it)
    MlmNode *fb_bnodes = nullptr, *fb_nodes = nullptr; // lean slots: the cell-table path's shared buffers
    MlmPair *fb_pairs = nullptr;
};

namespace {

#define HIPCHK(h, expr)                                                                                               \
    do {                                                                                                              \
        hipError_t e__ = (expr);                                                                                      \
        if (e__ != hipSuccess) {                                                                                      \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                            \
            return MLM_ERR_HIP;                                                                                       \
        }                                                                                                             \
    } while (0)

// Every entry point takes the handle's lock: integrate calls, queries and exports may come from different threads (the
// reference runs planner queries and the depth callback on an MT nodelet, nodelet_map.cpp:21); a query then observes the
// map of the last integrate call that returned.  Recursive because entry points call each other.
#define MLM_LOCK(h) std::lock_guard<std::recursive_mutex> lock__((h)->mu)

template <class T> int dev_alloc(mlm_handle *h, T **p, size_t n) {
    void *v = nullptr;
    HIPCHK(h, hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(T)));
    h->alloc_bytes += std::max<size_t>(n, 1) * sizeof(T);
    if (h->debug_alloc && n * sizeof(T) > (8u << 20)) fprintf(stderr, "[alloc] #%zu %.1f MB\n", h->allocs.size(), n * sizeof(T) / 1e6);
    h->allocs.push_back(v);
    *p = (T *)v;
    return MLM_OK;
}
inline unsigned int grid_for(size_t n) { return (unsigned int)((n + MLM_BLOCK - 1) / MLM_BLOCK); }
// blocks of k_bin_points for one frame (tile geometry: mlm_tile_item)
inline unsigned int bin_grid(const MlmDev &P, const MlmFrame &F, int mode) {
    const int tile_h = (int)(P.bin_block / 256) * 8;
    if (mode == 0) return (unsigned int)(((F.width + 31) / 32) * ((F.height + tile_h - 1) / tile_h));
    return (unsigned int)(((size_t)F.n + P.bin_block - 1) / P.bin_block);
}
constexpr unsigned int kListGrid = 256; // blocks of the grid-stride kernels that walk a device-sized list

struct Timed {
    mlm_handle *h;
    hipStream_t s;
    KernelTime *kt = nullptr;
    Timed(mlm_handle *hh, hipStream_t st, const char *name) : h(hh), s(st) {
        if (h->timing != 4) return; // mode 4: the two stage spans of a batch (plain event pairs on the stream)
        if (h->kpool_used == h->kpool.size()) {
            KernelTime k{name, nullptr, nullptr};
            hipEventCreate(&k.a);
            hipEventCreate(&k.b);
            h->kpool.push_back(k);
        }
        kt = &h->kpool[h->kpool_used++];
        kt->name = name;
        hipEventRecord(kt->a, s);
    }
    ~Timed() {
        if (kt) {
            hipEventRecord(kt->b, s);
            h->ktimes.push_back(*kt);
        }
    }
};

// Launch a kernel; when its launches are being timed, through hipExtLaunchKernelGGL with a start/stop event pair: the
// pair reports the kernel's own begin/end on the device (what rocprofv3 reports as its duration), not the time the
// stream waited for compute units behind the other streams' waves.
inline KernelTime *timing_slot(mlm_handle *h, const char *name) {
    if (!h->timing || h->timing == 4) return nullptr;
    if (h->timing == 3) { // one kernel only (bench: the dominant one), every `timed_every`-th launch of it
        if (strcmp(name, h->timed_kernel.c_str()) != 0) return nullptr;
        if (h->timed_count++ % h->timed_every != 0) return nullptr;
    }
    if (h->kpool_used == h->kpool.size()) {
        KernelTime k{name, nullptr, nullptr};
        hipEventCreate(&k.a);
        hipEventCreate(&k.b);
        h->kpool.push_back(k);
    }
    KernelTime *kt = &h->kpool[h->kpool_used++];
    kt->name = name;
    return kt;
}
template <class K, class... A>
inline void tlaunch(mlm_handle *h, const char *name, K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, A... args) {
    if (KernelTime *kt = timing_slot(h, name)) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)shmem, st, kt->a, kt->b, 0, args...);
        h->ktimes.push_back(*kt);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
    }
}

// A frame's Stage A gave up on the sector path.  First answer: schedule the pass with the large cell table for the batches to come
// (the usual reason is a column that overflowed the small table); if that pass was scheduled already, the scene does not fit
// the sector path at all: the next batches go straight to the cell-table path for a while.
inline void note_fallback(mlm_handle *h, int frame_no) {
    if (!(h->P.sec_tab_big && h->big_arm_len > 0)) {
        h->sector_backoff = h->sector_backoff_len;
    } else if (h->big_armed <= 0) {
        h->big_armed = h->big_arm_len;
        h->big_armed_from = h->P.explore ? (int)(h->ex_frame_no & 0x3FFFFFFF) : h->next_seq; // (frames submitted from now on have the pass behind them)
    } else if (frame_no >= h->big_armed_from) {
        h->sector_backoff = h->sector_backoff_len; // (it had the pass and gave up all the same)
    }
}
// lean slots of a sector-path handle outside frontier mode: the cell-table path's per-frame state exists once (alloc_slot)
inline bool share_ct(const mlm_handle *h) { return h->lean && !h->P.explore; }
int drain(mlm_handle *h, bool g_copied = false);
int grow_pool(mlm_handle *h, size_t want);
int ensure_free_blocks(mlm_handle *h, size_t need);
int widen_sec_tab(mlm_handle *h);
int ensure_free_blocks_idle(mlm_handle *h, size_t need);

// T_ls and t_wa of one frame (map_awareness.cpp:184-186)
void frame_setup(const mlm_handle *h, const double q_wb_in[4], const double t_wb_in[3], MlmFrame &F) {
    frame_pose(h->q_bs, h->t_bs, q_wb_in, t_wb_in, F.q_ls, F.t_ls, F.t_wa);
    F.rehash_thr = 0xFFFFFFFFu; // only the speculative Stage B arms the check (submit_batch)
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
    const MlmDev &P = S.P;
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
    for (int j = 0; j < n; ++j) h->h_frame_tab[base + j] = h->slots[(size_t)(base + j)].F;
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

// Stage A by azimuth sector (mlm_kernels_sector.h): two launches per batch.
int launch_stage_a_sector(mlm_handle *h, int base, int n) {
    const MlmSlot &S0 = h->slots[(size_t)base];
    const MlmDev &P = S0.P;
    const MlmFrame &F = S0.F;
    const int mode = S0.mode;
    const int set = base / (h->lim.max_batch);
    hipStream_t st = h->stream_as[set];
    HIPCHK(h, hipStreamWaitEvent(st, h->set_free[set], 0));
    if (!h->own_stream) { // see launch_stage_a_batch
        HIPCHK(h, hipEventRecord(h->inputs_ready, h->stream));
        HIPCHK(h, hipStreamWaitEvent(st, h->inputs_ready, 0));
    }
    for (int j = 0; j < n; ++j) h->h_frame_tab[base + j] = h->slots[(size_t)(base + j)].F;
    HIPCHK(h, hipMemcpyAsync(h->d_frame_tab + base, h->h_frame_tab + base, (size_t)n * sizeof(MlmFrame),
                             hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemsetAsync(h->d_ctr_all + base, 0, (size_t)n * sizeof(MlmCounters), st));
    unsigned int nb = 0;
    if (F.n > 0) {
        nb = mode == 0 ? (unsigned int)(((F.width + 31) / 32) * ((F.height + 7) / 8)) : (unsigned int)(((size_t)F.n + 255) / 256);
        if (nb > P.nb_cap) {
            h->err = "frame geometry exceeds the queues sized from mlm_limits.max_points";
            return MLM_ERR_CAPACITY;
        }
        // (one strip per workgroup: two or four strips worked on together are 5 % quicker with nothing else on the GPU — 3.22 -> 3.06
        // us per frame — and cost the pipeline 6 %: 91.0 -> 85.3 k frames/s, profiles/r4b; MLM_BIN_STRIPS selects them for experiments)
        if (mode == 0 && n > 1 && h->bin_strips == 4)
            tlaunch(h, "k_bin_sectors", k_bin_sectors<0, 4>, dim3((nb + 3) / 4, 1, n), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else if (mode == 0 && n > 1 && h->bin_strips == 2)
            tlaunch(h, "k_bin_sectors", k_bin_sectors<0, 2>, dim3((nb + 1) / 2, 1, n), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
        else if (mode == 0)
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
        tlaunch(h, "k_rank", k_rank, dim3(n > 4 ? h->rank_grid : 1024, 1, n), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base,
                mode == 0 ? F.width : 0, row_w, dm, ds);
        // blocks per frame: about 400 ranked cells per block in a batch (a lane that finishes a chain draws the next cell; each
        // block builds the transposed odds table in LDS), as many as the last confirmed frame had; single frames spread wider
        unsigned int cg = h->chain_grid;
        if (!cg) {
            const long long cells = std::max<long long>(1, h->stats.n_multi_cells);
            cg = n > 4 ? (unsigned int)std::min<long long>(64, std::max<long long>(8, cells / 400)) : (unsigned int)std::min<long long>(128, std::max<long long>(16, cells / 128));
        }
        tlaunch(h, "k_chain_lanes", k_chain_lanes, dim3(cg, 1, n), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab,
                base);
        // the frame's hits and misses grouped by voxel, tile by tile (needs the increments and keys of the kernels above)
        if (!P.explore)
            tlaunch(h, "k_tile", k_tile, dim3(n > 1 ? h->tile_grid : (unsigned int)(P.n_tiles <= 4096 ? P.n_tiles : 1024), 1, n), dim3(MLM_TILE_THREADS), h->tile_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base);
    }
    HIPCHK(h, hipEventRecord(h->stage_a_done[set], st));
    return MLM_OK;
}

// Stage B+C of one frame on the main stream.  n_bkt != 0: speculative single-epoch ordering inside k_voxelize;
// n_bkt == 0: hl_key was produced by order_hits_exact.
void launch_stage_bc(mlm_handle *h, MlmSlot &S, unsigned long long n_bkt) {
    const MlmDev &P = S.P;
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
    h->stats.block_capacity = h->P.max_blocks;
}

int check_queues(mlm_handle *h, const MlmSlot &S) {
    const MlmCounters &c = *S.h_ctr;
    const MlmDev &P = S.P;
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
    tlaunch(h, "k_ex_apply_misses", k_ex_apply_misses, dim3(kListGrid), blk, 0, h->stream, h->ex_tail->P);
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
// end of a batch (or of a single frame): the last frame's tail, the map-wide counters
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
int explore_stage_a(mlm_handle *h, int base, int n) {
    const bool sectors = h->use_sectors && h->sector_backoff == 0 && h->slots[(size_t)base].F.width <= 2040;
    if (h->sector_backoff > 0) --h->sector_backoff;
    for (int j = 0; j < n; ++j) {
        MlmSlot &S = h->slots[(size_t)(base + j)];
        S.seq = 0;
        S.F.seq = 0;
        S.F.pad2 = (int)(h->ex_frame_no++ & 0x3FFFFFFF); // (frame counter for the MLM_SEC_FAIL_EVERY test hook)
        S.sector = sectors;
    }
    Timed t(h, h->stream_as[base / h->lim.max_batch], "stage_a_batch");
    return sectors ? launch_stage_a_sector(h, base, n) : launch_stage_a_batch(h, base, n);
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
            tlaunch(h, "k_apply_single", k_apply_single, dim3(h->single_apply_grid, 1, 1), dim3(MLM_BLOCK), 0, h->stream, h->d_slot_tab, h->d_frame_tab, base + j0);
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
    // (the sector path packs a tile's image column into 8 bits of a reference: images up to 2040 pixels wide; its bucket-first
    // tables hold sbkt_cap buckets)
    const bool sectors = h->use_sectors && h->sector_backoff == 0 && h->slots[(size_t)base].F.width <= 2040 &&
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
        Timed t(h, h->stream_as[set], "stage_a_batch");
        rc = sectors ? launch_stage_a_sector(h, base, n) : launch_stage_a_batch(h, base, n);
    }
    if (rc) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[set], 0));
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
            tlaunch(h, "k_apply_voxelize", k_apply_voxelize, dim3(scg * (MLM_BLOCK / h->sc_block), 2 * (1 + MLM_RAY_LISTS)), dim3(h->sc_block), 0, h->stream, Sa.P,
                    Sa.F.seq, j > 0 ? 1 : 0, Sv.P, Sv.F, h->hit_n_bkt, j < n ? 1 : 0);
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
    return S.F.width <= 2040 && h->hit_n_bkt <= S.P.sbkt_cap && S.F.n > 0;
}
// the stream uploads of a call's inputs go to: the one its Stage A will run on (a frame-level veto of the graph path is
// repaired by run_slots with an event between the two streams)
inline hipStream_t upload_stream(const mlm_handle *h) { return fast_handle_ok(h) ? h->stream : h->stream_as[h->cur_set]; }
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
    mlm_handle::SingleGraph *G = nullptr;
    for (auto &g : h->graphs)
        if (g.mode == S.mode && g.width == S.F.width && g.height == S.F.height && g.base == base && g.nb == nb && g.sec_tab == P.sec_tab && g.n_bkt == h->hit_n_bkt && g.big == big) G = &g;
    if (!G) {
        if (h->graphs.size() >= 8) { // (a handful of frame geometries at most; the bucket count of the emulated container changes a dozen times per stream)
            for (auto &g : h->graphs) hipGraphExecDestroy(g.exec);
            h->graphs.clear();
        }
        hipStream_t st = h->stream;
        HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipSuccess;
        if (h->graph_copies == 1) {
            e = hipMemcpyAsync(h->d_frame_tab + base, h->h_frame_tab + base, sizeof(MlmFrame), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemsetAsync(h->d_ctr_all + base, 0, sizeof(MlmCounters), st);
        } else if (h->graph_copies == 2) {
            hipLaunchKernelGGL(k_frame_prologue, dim3(1), dim3(128), 0, st, (const MlmFrame *)(h->h_frame_tab + base), h->d_frame_tab + base, h->d_ctr_all + base);
        }
        if (e == hipSuccess) {
            if (S.mode == 0) hipLaunchKernelGGL((k_bin_sectors<0, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
            else if (S.mode == 1) hipLaunchKernelGGL((k_bin_sectors<1, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
            else hipLaunchKernelGGL((k_bin_sectors<2, 1>), dim3(nb, 1, 1), dim3(256), 0, st, h->d_slot_tab, h->d_frame_tab, base, nb);
            const int row_w = S.mode == 0 ? S.F.width : 64;
            unsigned long long dm, rm;
            int ds, rs;
            div_magic((unsigned int)row_w, dm, ds);
            div_magic((unsigned int)P.nRho, rm, rs);
            // (a single frame is alone on the GPU: the 512-thread workgroup finishes a column sooner; the table is the same)
            if (h->sec_threads == 256 && P.sec_tab < 512u)
                hipLaunchKernelGGL((k_sector<false, 256>), dim3((unsigned int)P.nPhi, 1, 1), dim3(256), P.sec_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base,
                                   S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big, dm, ds);
            else
                hipLaunchKernelGGL((k_sector<false, 512>), dim3((unsigned int)P.nPhi, 1, 1), dim3(512), P.sec_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base,
                                   S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, big, dm, ds);
            if (big)
                hipLaunchKernelGGL(k_sector_big<false>, dim3(h->big_grid), dim3(MLM_SEC_THREADS), P.sec_big_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base, 1,
                                   S.mode == 0 ? S.F.width : 0, (int)nb, rm, rs, (unsigned long long)h->hit_n_bkt, dm, ds);
            hipLaunchKernelGGL(k_rank, dim3(256, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base, S.mode == 0 ? S.F.width : 0, row_w, dm, ds);
            hipLaunchKernelGGL(k_chain_lanes, dim3(32, 1, 1), dim3(MLM_BLOCK), (size_t)32 * P.nRho * sizeof(float), st, h->d_slot_tab, h->d_frame_tab, base);
            hipLaunchKernelGGL(k_tile, dim3((unsigned int)(P.n_tiles <= 4096 ? P.n_tiles : 1024), 1, 1), dim3(MLM_TILE_THREADS), h->tile_lds_bytes, st, h->d_slot_tab, h->d_frame_tab, base);
            hipLaunchKernelGGL(k_apply_single, dim3(h->single_apply_grid, 1, 1), dim3(MLM_BLOCK), 0, st, h->d_slot_tab, h->d_frame_tab, base);
            e = hipGetLastError();
        }
        if (h->graph_copies == 1) {
            if (e == hipSuccess) e = hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, sizeof(MlmCounters), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, st);
        } else if (h->graph_copies == 2 && e == hipSuccess) {
            hipLaunchKernelGGL(k_frame_epilogue, dim3(1), dim3(128), 0, st, (const MlmCounters *)(h->d_ctr_all + base), h->h_ctr_all + base, (const MlmGlobal *)h->P.g, h->h_g,
                               (const MlmFrame *)(h->d_frame_tab + base));
            e = hipGetLastError();
        }
        hipGraph_t graph = nullptr;
        const hipError_t e2 = hipStreamEndCapture(st, &graph);
        if (e != hipSuccess || e2 != hipSuccess || !graph) {
            if (graph) hipGraphDestroy(graph);
            h->err = std::string("single-frame graph capture: ") + hipGetErrorString(e != hipSuccess ? e : e2);
            return MLM_ERR_HIP;
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e != hipSuccess) {
            h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
            return MLM_ERR_HIP;
        }
        h->graphs.push_back(mlm_handle::SingleGraph{S.mode, S.F.width, S.F.height, base, big, nb, P.sec_tab, h->hit_n_bkt, exec});
        G = &h->graphs.back();
    }
    h->h_frame_tab[base] = S.F;
    if (h->graph_copies == 0) {
        HIPCHK(h, hipMemcpyAsync(h->d_frame_tab + base, h->h_frame_tab + base, sizeof(MlmFrame), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_ctr_all + base, 0, sizeof(MlmCounters), h->stream));
    }
    if (h->graph_copies == 2) { // (the epilogue kernel ends with a ticket in the host copy of the map-wide flags: drain polls it)
        h->h_g->pad = 0u;
        h->wait_ticket = (unsigned int)S.F.seq + 1u;
    }
    HIPCHK(h, hipGraphLaunch(G->exec, h->stream));
    if (h->graph_copies == 0) {
        HIPCHK(h, hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_g, h->P.g, sizeof(MlmGlobal), hipMemcpyDeviceToHost, h->stream));
    }
    h->n_graph_launches++;
    h->pending.push_back(&S);
    h->set_pending[set] = 1;
    return MLM_OK;
}

// Integrate the frames already described in slots[base..base+n) (F, mode set), in order.
int run_slots(mlm_handle *h, int n) {
    (void)hipGetLastError(); // a stale error of unrelated HIP calls in this thread is not ours
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
            int rc = explore_stage_a(h, base, n);
            if (rc) return rc;
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_a_done[h->cur_set], 0));
            HIPCHK(h, hipMemcpyAsync(h->h_ctr_all + base, h->d_ctr_all + base, (size_t)n * sizeof(MlmCounters), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            HIPCHK(h, hipGetLastError());
            rc = explore_redo_overflows(h, base, n);
            if (rc) return rc;
            for (int j = 0; j < n; ++j) {
                rc = explore_stage_bc(h, base + j);
                if (rc) return rc;
            }
            rc = explore_end_batch(h);
            if (rc) return rc;
            HIPCHK(h, hipEventRecord(h->set_free[h->cur_set], h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
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
        hipStream_t target = fast ? h->stream : h->stream_as[set];
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

int run_query(mlm_handle *h, int mode, const double *pos, int n, float inflate, int max_iter, void *out,
              size_t out_elem) {
    if (!h || !pos || !out || n < 0) return MLM_ERR_INVALID;
    if (n == 0) return MLM_OK;
    MLM_LOCK(h);
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
int upload_slot_tab(mlm_handle *h) {
    std::vector<MlmDev> tab(h->slots.size());
    for (size_t i = 0; i < h->slots.size(); ++i) tab[i] = h->slots[i].P;
    HIPCHK(h, hipMemcpy(h->d_slot_tab, tab.data(), tab.size() * sizeof(MlmDev), hipMemcpyHostToDevice));
    if (h->lean && h->d_slot_tab_fb) {
        for (size_t i = 0; i < tab.size(); ++i) {
            tab[i].bnodes = h->fb_bnodes;
            tab[i].pairs = h->fb_pairs;
            tab[i].nodes = h->fb_nodes;
        }
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
    if ((rc = dev_alloc(h, &P.mt_list, NC))) return rc;
    if ((rc = dev_alloc(h, &P.mt_rec, NC))) return rc;
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
        P.contrib_cap = (unsigned int)cap_pad;
        if ((rc = dev_alloc(h, &P.contrib, cap_pad))) return rc;
        if ((rc = dev_alloc(h, &P.subs, cap_pad))) return rc;
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
        if ((rc = dev_alloc(h, &P.hl_vt16, NC + 8))) return rc;
        if ((rc = dev_alloc(h, &P.tile_cols, (size_t)P.n_tiles * P.tile_words))) return rc;
        HIPCHK(h, hipMemset(P.tile_cols, 0, (size_t)P.n_tiles * P.tile_words * sizeof(uint32_t)));
        if ((rc = dev_alloc(h, &P.tile_desc, 4 * (size_t)P.n_tiles * (size_t)P.nPhi))) return rc;
        // a frame touches at most one voxel per awareness cell, and no more voxels than its grid has
        P.rec_cap = (unsigned int)std::min<size_t>(NC, (size_t)P.lv_nx * P.lv_ny * P.lv_nz);
        if ((rc = dev_alloc(h, &P.vr_rec, (size_t)P.rec_cap))) return rc;
        if ((rc = dev_alloc(h, &P.vr_hit, NC))) return rc;
        if ((rc = dev_alloc(h, &P.tile_dir, 4 * (size_t)P.n_tiles))) return rc;
        HIPCHK(h, hipMemset(P.tile_dir, 0xFF, 4 * (size_t)P.n_tiles * sizeof(uint32_t))); // (no frame carries that sequence number)
        // bucket-first table of this slot: room for the emulated container of a frame with up to 2 * max_points unique
        // hit cells (more: the handle continues on the cell-table path)
        P.sbkt_cap = (unsigned int)std::min<size_t>(h->max_buckets, std::__detail::_Prime_rehash_policy()._M_next_bkt(4 * (size_t)h->lim.max_points + 2));
        if ((rc = dev_alloc(h, &P.sbkt, (size_t)P.sbkt_cap))) return rc;
        HIPCHK(h, hipMemset(P.sbkt, 0xFF, (size_t)P.sbkt_cap * sizeof(unsigned long long)));
    }
    // (a reference — one row of a group's lane mask — stands for at least one contribution; a cell's references start at a multiple
    // of MLM_SEC_REF_ALIGN, and a cell that needs references has at least two contributions)
    P.refs_cap = (unsigned int)std::min<size_t>(0xFFFFFFF0ull, max_contrib + (MLM_SEC_REF_ALIGN - 1) * std::min<size_t>(NC, max_contrib / 2) + 64);
    if ((rc = dev_alloc(h, &P.refs, h->use_sectors ? (size_t)P.refs_cap : 4))) return rc;
    if ((rc = dev_alloc(h, &P.mt_ref, h->use_sectors ? 2 * NC : 2))) return rc;
    HIPCHK(h, hipMemset(P.col_cnt, 0, (size_t)P.nPhi * sizeof(unsigned int)));
    if ((rc = dev_alloc(h, &P.hl_cell, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_t, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_odd, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_inc, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_base, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_cnt, NC))) return rc;
    if ((rc = dev_alloc(h, &P.hl_vt, NC))) return rc;
    MLM_CT_ALLOC(hl_arr, NC);
    if ((rc = dev_alloc(h, &P.hl_key, NC))) return rc;
    MLM_CT_ALLOC(hl_next, NC);
    MLM_CT_ALLOC(hl_vox, NC);
    if ((rc = dev_alloc(h, &P.hl_bkt, NC))) return rc;
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

} // namespace

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
        if (knob("graph_copies", kv)) h->graph_copies = (int)kv;
        if (knob("bin_strips", kv)) h->bin_strips = (int)kv == 2 ? 2 : ((int)kv == 4 ? 4 : 1);
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
                tab[i].bnodes = h->fb_bnodes;
                tab[i].pairs = h->fb_pairs;
                tab[i].nodes = h->fb_nodes;
            }
            if ((rc = dev_alloc(h, &h->d_slot_tab_fb, NS))) return rc;
            HIPCHK(h, hipMemcpy(h->d_slot_tab_fb, tab.data(), NS * sizeof(MlmDev), hipMemcpyHostToDevice));
            HIPCHK(h, hipEventCreateWithFlags(&h->fb_done, hipEventDisableTiming));
        }
    }
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
    if (h->inputs_ready) hipEventDestroy(h->inputs_ready);
    if (h->fb_done) hipEventDestroy(h->fb_done);
    if (h->d_f32) hipFree(h->d_f32);
    if (h->d_qpos) hipFree(h->d_qpos);
    if (h->d_qout) hipFree(h->d_qout);
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
        for (int j = 0; j < n; ++j) {
            MlmSlot &S = cur_slot(h, j);
            int rc = ensure_img(h, S, n_px);
            if (rc) return rc;
            HIPCHK(h, hipMemcpyAsync(S.d_img, img_host + (size_t)(k0 + j) * frame_stride, n_px * sizeof(uint16_t),
                                     hipMemcpyHostToDevice, h->stream_as[h->cur_set]));
            S.F = MlmFrame{};
            frame_setup(h, q_wb + 4 * (size_t)(k0 + j), t_wb + 3 * (size_t)(k0 + j), S.F);
            S.F.img = S.d_img;
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
    return mlm_integrate_depth_u16_dev(h, S.d_img, width, height, row_stride, pixel_idx ? S.d_pix : nullptr, n_idx, q_wb,
                                       t_wb);
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
    // ---- pose latency compensation, mlmap.cpp:470-498 (mlm_host.h)
    double qa[4], ta[3];
    compensate_pose(odom_p, odom_q, odom_v, imu_w, t_img, t_odom, t_imu, latency, qa, ta);
    if (T_wb_out) {
        for (int i = 0; i < 4; ++i) T_wb_out[i] = qa[i];
        for (int i = 0; i < 3; ++i) T_wb_out[4 + i] = ta[i];
    }
    // ---- depth image: upload (and convert 32FC1 -> 16UC1 on the device)
    int rc = drain(h);
    if (rc) return rc;
    MlmSlot &S = cur_slot(h, 0);
    rc = ensure_img(h, S, n_px);
    if (rc) return rc;
    std::vector<int32_t> pix;
    if (sampled && h->cfg.sample_cnt > 0 && 2 * (size_t)h->cfg.sample_cnt <= (size_t)h->lim.max_points) {
        // project_depth, mlmap.cpp:311-349 (glibc rand(), v first, zeros skipped).  Only the sampled pixels are ever read
        // by the kernels, so only they travel: the host converts them (same float arithmetic as k_convert_f32_u16), a tiny
        // kernel drops them into the device image at their pixel positions.
        // The indices and depths are staged in a pinned buffer of the handle (the call drains before it returns, so the buffer
        // is free again at the next call) and travel in one copy; the kernels take the depths from that list.
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
        while (n_s < want && cnt < max_iter) {
            cnt++;
            const size_t v = static_cast<size_t>(rand() % height);
            const size_t u = static_cast<size_t>(rand() % width);
            const size_t at = v * (size_t)width + u;
            int r;
            if (is_f32) {
                r = mlm_cv_f32_to_u16(((const float *)depth)[at]);
            } else {
                r = ((const uint16_t *)depth)[at];
            }
            if (r == 0) continue;
            st_pix[n_s] = (int32_t)at;
            st_raw[n_s] = r;
            ++n_s;
        }
        if ((rc = ensure_pix(h, S))) return rc;
        if (n_s) {
            hipStream_t up = upload_stream(h);
            h->last_upload = up;
            HIPCHK(h, hipMemcpyAsync(S.d_pix, h->h_stage, 2 * want * sizeof(int32_t), hipMemcpyHostToDevice, up));
        }
        return integrate_u16_dev(h, S.d_img, width, height, width, S.d_pix, S.d_pix + want, (int)n_s, qa, ta);
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
    return mlm_integrate_depth_u16_dev(h, S.d_img, width, height, width, sampled ? S.d_pix : nullptr, (int)pix.size(), qa, ta);
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
    }
    S.F = MlmFrame{};
    frame_setup(h, q_wb, t_wb, S.F);
    S.F.pts = S.d_pts;
    S.F.n = n;
    S.F.width = 1;
    S.mode = 2;
    return run_slots(h, 1);
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

int mlm_get_frame_stats(mlm_handle *h, mlm_frame_stats *out) {
    if (!h || !out) return MLM_ERR_INVALID;
    MLM_LOCK(h);
    *out = h->stats;
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
    const MlmDev &P = h->slots[(size_t)h->last_slot].P;
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
