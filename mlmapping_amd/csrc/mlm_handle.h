// mlm_handle.h — the handle behind the C ABI (struct mlm_handle, one frame slot, the test knobs) and the small helpers every part of
// the host driver uses: error macro, lock, device allocation, timed kernel launches.  Part of the single device TU mlmap_hip.hip.
#pragma once
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
const char *const kKnobNames[] = {"agg_lds", "big_arm", "big_grid", "bin_block", "chain_grid", "collect_grid", "cu_reserve", "cu_split",
                                  "debug_fail_slot", "ex_spec", "expand_block", "graph", "lean_slots", "logit_exact", "mirror", "mirror_max", "mirror_mb", "need_slots", "node_lds", "pool_grow",
                                  "rank_grid", "sc_block", "sc_grid", "sec_backoff", "sec_fail_every", "sec_tab", "sec_tab_big", "sec_threads",
                                  "sectors", "single_chain_grid", "single_rank_grid", "slot_sets", "sort_block", "sort_grid", "tile_grid", "tile_sh"};
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
    MlmDev Pfb{};              // the parameter block a CELL-TABLE Stage A / apply of this slot's frame runs with: the buffers that path shares
                               // across the slots (lean slots) and, where the slot's own lists are sized by need, the handle's one
                               // full-size set of them (eff_params)
    uint16_t *d_img = nullptr; // staging for host images
    size_t img_cap = 0;
    int32_t *d_pix = nullptr;
    double *d_pts = nullptr;
};

// Host-resident mirror of the map for small query batches (mlm_mirror.h): pinned host planes that mirror the block pool slot by
// slot, a host table block key -> slot, and the record of what changed on the device since the last refresh.
struct MlmMirror {
    bool enabled = true;           // knob "mirror" = 0: every query runs as a kernel
    bool alloc_failed = false;     // the pinned planes could not be allocated: disabled for good, queries run as kernels
    int max_clean = 256;           // largest batch answered on the host while the mirror is up to date (knob "mirror_max") ...
    int max_dirty = 32;            // ... and while it needs a refresh first (a large batch is then cheaper as one kernel)
    size_t cap = 0;                // blocks the planes hold
    size_t max_bytes = (size_t)1 << 30; // most pinned host memory the planes may take (mlm_set_host_mirror_limit; knob "mirror_mb"): a map
                                   // that needs more is queried by kernels only (over_limit) — nothing is pinned behind the caller's back without bound
    bool over_limit = false;
    float *lo = nullptr;           // [cap * cells] pinned, device-visible
    uint8_t *occ = nullptr, *infl = nullptr; // [cap * cells]
    uint8_t *col = nullptr;        // [cap] released ("collapsed") blocks: element 0 answers
    int *keys = nullptr;           // [cap * 3]
    unsigned int *stat = nullptr;  // pinned: [0] block count seen by the refresh kernel, [2 + i] blocks copied by its workgroup i
    mlm_host::MapView view;        // the planes above + the host table block key -> slot, with the reference's query inlines on top (mlm_mapview.h)
    unsigned int n_known = 0;      // blocks [0, n_known) are in the table and have valid planes
    bool dirty = true;             // the device map changed (or may have) since the last refresh ...
    mlm_host::DirtyBoxes boxes;    // ... anywhere, or inside these boxes of block indices (mlm_mapview.h)
    long long n_refresh = 0, n_copied = 0, n_host_queries = 0;
    // EAGER refresh (mirror_eager): a planner that asks after every frame pays the refresh — a launch, ~1 MB across the link, a
    // synchronisation: ~48 us — on its first query.  Once a query has found the mirror stale, every synchronous integrate call that
    // follows launches the refresh itself as it returns (nothing waits for it); the first query then only waits for an event that
    // has usually fired.  Switched off again by the first integrate call that finds the previous eager refresh unused.
    bool eager_on = false;         // queries do follow the integrate calls
    bool eager_pending = false;    // a refresh kernel is in flight or done, its results not yet taken in (mirror_collect)
    hipEvent_t eager_ev = nullptr; // behind that kernel
    long long q_at_eager = -1;     // n_host_queries when it was launched
    long long n_eager = 0;
};

// A frame on its own with at most this many strips of 256 points (sampled callbacks, point lists: 4 096 points) runs its cells' float chains
// inside k_rank<true> instead of launching k_chain_lanes — a launch of its own costs a lone frame the kernel boundary (1.5 us) and the few
// dependent trips to memory every kernel starts with (parameters, counts, its list); a dense frame's thousands of cells keep
// k_chain_lanes' cell-per-lane replay (measured: 128 -> 201 us for a dense VGA frame with the chains in k_rank)
constexpr unsigned int kFusedChainStrips = 16;
static inline double mlm_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
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
    uint16_t *d_img_set[MLM_SETS] = {};      // per slot set: device buffer of a whole batch of host frames that arrive back to back
    size_t img_set_cap[MLM_SETS] = {};       // (mlm_integrate_depth_batch: one upload per batch)
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
    double clk[8] = {}, clk_t = 0;   // host clocks of the single-frame path (mlm_debug_clocks): microseconds per section, summed over the calls
    MlmExOrder ex_om{};              // frontier mode: what the next lone frame's k_rank needs to run the bucket-first pass (explore_spec_begin; on = 0: nothing)
    bool ex_om_launched = false;     // ... and whether the frame's Stage A took it along (else explore_stage_bc_spec launches k_ex_order_min)
    unsigned int wait_ticket = 0;    // nonzero: the single-frame graph in flight ends by writing this into h_g->pad (pinned)
    unsigned int single_rank_grid = 256;   // workgroups of a lone frame's k_rank<true> (ranking + chains; knob "single_rank_grid")
    unsigned int single_chain_grid = 64;  // workgroups of a lone frame's k_chain_lanes: 256 waves x 64 cells cover a dense VGA frame's ranked cells in one turn
    unsigned int single_apply_grid = 256; // workgroups of k_apply_single: a VGA frame's ~33 k voxel records, one per thread (more: in turns)
    unsigned int tile_grid = 0;      // workgroups of k_tile per frame of a batch: they walk the frame's touched tiles (MLM_TILE_GRID)
    unsigned int apply_lds_bytes = 0; // dynamic LDS of k_apply_tiles: 9 bytes per voxel of a tile
    unsigned int big_grid = 256;     // workgroups of k_sector_big per batch: one per CU (MLM_BIG_GRID)
    int ex_spec = 1;                 // frontier mode: a synchronous frame's map-dependent part is enqueued before its counts are known (explore_stage_bc_spec);
                                     // knob "ex_spec": 0 never, 2 with thresholds of zero (every frame misses: the test of the way back)
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
        bool no_prologue; // the variant for a small frame whose slot counters are clear (k_bin_sectors_hostf first)
        const int32_t *list; // ... and the staged pixel list its first kernel prefetches (null: none), `list_n` entries in front of the depths
        unsigned int list_n;
        hipGraphExec_t exec;
    };
    std::vector<SingleGraph> graphs;
    bool use_graph = true;       // (MLM_GRAPH=0: always the general submission)
    bool no_spread = false;      // the noise model never spreads a hit beyond its own cell (3 sigma < 1 cell everywhere, e.g. the reference's default
                                 // depth_noise_coe 1e-6): k_chain_lanes has nothing to do and is not launched
    hipStream_t last_upload = nullptr; // the stream the current call's inputs were uploaded on (run_slots orders Stage A behind it)
    hipEvent_t upload_ev = nullptr;
    hipEvent_t host_read_ev = nullptr; // asynchronous mode: recorded behind the copies that read a caller's host buffer (borrowed_mark / borrowed_wait)
    bool host_read_marked = false;
    long long n_graph_launches = 0;
    // Pinned staging of the callback's sampled pixels (indices, then raw depths).  k_bin_sectors reads it ACROSS THE LINK while the
    // frame is in flight, so it may only be rewritten when nothing is in flight: mlm_integrate_callback drains before it samples
    // (its first statement after the pose), and no other entry point touches the buffer.  Keep it that way — or give every slot
    // set a staging buffer of its own — when adding shortcuts.
    int32_t *h_stage = nullptr;
    size_t stage_cap = 0;
    long long n_pool_grows = 0;
    long long n_big_redos = 0;   // frames whose overflowed columns were redone with the large table at drain time (redo_overflow_columns)
    size_t grow_failed_at = 0;   // a pool of this many blocks did not fit the device (grow_pool does not retry it)
    MlmNode *fb_bnodes = nullptr, *fb_nodes = nullptr; // lean slots: the cell-table path's shared buffers
    MlmPair *fb_pairs = nullptr;
    MlmMirror mir;               // host mirror for small query batches (mlm_mirror.h)
    // Frame slots sized by NEED (sector-path handles with lean slots, not frontier mode): the per-frame lists whose worst case is
    // "every awareness cell is a multi-kind hit" start at what a camera frame of max_points pixels needs and are doubled at a
    // drained point when a frame's Stage A runs out of room (grow_slots, like grow_pool for the block pool).  The cell-table path,
    // which a frame falls back to one at a time, gets ONE full-size set of those lists per handle, allocated at its first use.
    bool need_sized = false;
    struct SlotCaps {
        size_t hl = 0, mt = 0, vh = 0, rec = 0, refs = 0, sub = 0, sbkt = 0;
    } caps_worst, caps_now;
    long long n_slot_grows = 0;
    std::vector<std::pair<void *, size_t>> regrow_trash; // lists replaced by resize_slots, freed once the copies into the new ones are through
    struct CtFull { // the cell-table path's full-size lists (ensure_ct_full)
        bool ready = false;
        uint32_t *mt_list = nullptr, *contrib = nullptr, *hl_cell = nullptr, *hl_t = nullptr, *hl_base = nullptr, *hl_cnt = nullptr, *hl_vt = nullptr, *hl_bkt = nullptr;
        uint4 *mt_rec = nullptr;
        uint8_t *subs = nullptr;
        float *hl_odd = nullptr, *hl_inc = nullptr;
        uint64_t *hl_key = nullptr;
    } ct_full;
};

namespace {

#define HIPCHK(h, expr)                                                                                               \
    do {                                                                                                              \
        hipError_t e__ = (expr);                                                                                      \
        if (e__ != hipSuccess) {                                                                                      \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                            \
            (void)hipGetLastError(); /* (reported: a later hipGetLastError() of this thread must not find it again) */  \
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
constexpr unsigned int kHostFrameStrips = 16; // a frame of at most this many strips may start its graph without k_frame_prologue (submit_single_graph)
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
int grow_slots(mlm_handle *h, const MlmCounters &demand);
int grow_sbkt(mlm_handle *h, size_t buckets);
int ensure_ct_full(mlm_handle *h);
// the parameter block the kernels of slot S's frame run with on the path the frame is on (see MlmSlot::Pfb)
inline const MlmDev &eff_params(const mlm_handle *h, const MlmSlot &S) { return (S.sector || !h->lean) ? S.P : S.Pfb; }
void mirror_mark_all(mlm_handle *h);
void mirror_mark_box(mlm_handle *h, const int lo[3], const int hi[3]);
void mirror_mark_frames(mlm_handle *h, int n);

} // namespace

// host clock of the single-frame path: the time since the previous mark goes to section i (i < 0: start of a call)
static inline void clk_mark(mlm_handle *h, int i) {
    if (i >= 0 && h->clk_t == 0.0) return; // (not inside a clocked call: the marks of the submission and drain code are shared)
    const double t = mlm_now_us();
    if (i >= 0) h->clk[i] += t - h->clk_t;
    h->clk_t = i == 5 ? 0.0 : t; // (5: the call's last mark)
}

// Host buffers are BORROWED for the duration of a call (include/mlmap_hip.h).  In synchronous mode a call drains before it returns;
// in asynchronous mode it returns once the work is submitted — and a copy from a REGISTERED (pinned) host buffer is then still in
// flight (from pageable memory the runtime has staged it already).  The entry points that read host buffers therefore record an
// event right behind their copies (before the frame's kernels go onto that stream) and wait for it — for the upload only, not for
// the compute — before they return.
static inline int borrowed_mark(mlm_handle *h, hipStream_t up) {
    if (!h->async_mode) return MLM_OK;
    if (!h->host_read_ev) HIPCHK(h, hipEventCreateWithFlags(&h->host_read_ev, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->host_read_ev, up));
    h->host_read_marked = true;
    return MLM_OK;
}
static inline int borrowed_wait(mlm_handle *h) {
    if (!h->host_read_marked) return MLM_OK;
    h->host_read_marked = false;
    HIPCHK(h, hipEventSynchronize(h->host_read_ev));
    return MLM_OK;
}

// Wait for a stream the way a latency-bound caller wants it: hipStreamSynchronize may put the thread to sleep, and the wake-up costs
// more (50-100 us measured on frontier mode's per-frame calls) than the work it waits for; so poll the stream for up to a couple of
// milliseconds first (a frame's launches take 0.1-0.3 ms), then fall back to the blocking call.
static inline hipError_t mlm_spin_sync(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    return hipStreamSynchronize(st);
}
// ... and when the frame's last launch ends with a ticket in pinned memory (mlm_hand_back): poll that word — it arrives a few microseconds
// before the stream reads as idle; past a few milliseconds (or with no ticket expected) wait for the stream
static inline hipError_t wait_for_ticket(mlm_handle *h, hipStream_t st) {
    const unsigned int want = h->wait_ticket;
    h->wait_ticket = 0u;
    if (want && !h->timing) { // (per-kernel timing: the events behind the last launch must have completed too)
        const volatile unsigned int *ticket = &h->h_g->pad;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned int spins = 0;; ++spins) {
            if (*ticket == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return hipSuccess;
            }
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
    }
    return mlm_spin_sync(st);
}
