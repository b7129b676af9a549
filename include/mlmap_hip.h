/*
 * mlmap_hip.h — C ABI of the MI355X-native MLMapping map-update path (libmlmap_hip.so).
 *
 * This is the drop-in boundary for ONE path of the reference: mlmap::update_map()
 * (src/mlmap.cpp:382-386 = awareness_map_cylindrical::input_pc_pose, src/map_awareness.cpp:173-282,
 *  + local_map_cartesian::input_pc_pose_direct, src/map_local.cpp:143-237) and the query inlines
 * planners call on the result (include/mlmap.h:142-295, src/mlmap.cpp:388-407).
 *
 * The reference has no FFI: it is one C++ process.  Each entry point below names the C++ member it
 * replaces; INTEGRATION.md shows the mlmap-side stubs a maintainer would add, and
 * include/mlmap_facade.hpp is a header-only class with the reference's public names on top of this ABI.
 *
 * Conventions
 *  - every function returns 0 on success, a negative mlm_status otherwise; no exceptions cross the ABI
 *    (the reference has no error channel at all: yaml-cpp / vector::at exceptions kill the nodelet);
 *  - plain pointers and sizes only; host buffers are borrowed for the duration of the call;
 *  - one handle = one device + one HIP stream.  Every entry point takes the handle's lock, so a handle may be used
 *    from several threads (the reference serves planner queries and the depth callback from an MT nodelet,
 *    src/nodelet_map.cpp:21): calls are serialised, query_* calls observe the map as of the last integrate call
 *    that returned (in async mode they first wait for everything submitted).  mlm_last_error is per handle: read it
 *    on the thread that got the failure before that thread issues another call.  mlm_destroy must not race with
 *    other calls;
 *  - the block pool grows on demand (while it grows, the old and the new pool are resident together: about three times the old
 *    pool for a moment); after MLM_ERR_CAPACITY (device memory exhausted, a frame with more points than mlm_limits.max_points, or
 *    a pool fixed by the test knob "pool_grow" = 0) the handle stays usable: the map keeps what the failing call applied before it
 *    ran out of room, later frames integrate normally.  Blocks are created by the map-independent stage, up to three batches
 *    ahead of the stage that applies a frame: after a failed call the map may hold blocks of frames that were never applied —
 *    all 'u' / 0.0f, i.e. what the reference's allocate_ram leaves for a block nothing was integrated into — and
 *    mlm_frame_stats.n_blocks counts them;
 *  - poses are q_wb = (w,x,y,z) and t_wb of T_wb (body in world), exactly what mlmap.cpp:494 builds;
 *  - positions are world-frame doubles (Vec3 of include/common.h:22), n x 3 row-major.
 */
#ifndef MLMAP_HIP_H
#define MLMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLM_ABI_VERSION 6 /* 6: mlm_set_host_mirror_limit */

typedef enum mlm_status {
    MLM_OK = 0,
    MLM_ERR_INVALID = -1,     /* bad argument */
    MLM_ERR_HIP = -2,         /* HIP runtime error, see mlm_last_error */
    MLM_ERR_CAPACITY = -3,    /* block pool / point capacity exceeded (map unchanged by the failing frame's tail) */
    MLM_ERR_UNSUPPORTED = -4, /* configuration outside the supported envelope */
} mlm_status;

/* mlmap::getOccupancy return values, include/mlmap.h:109-114 */
enum { MLM_FREE = 1, MLM_OCCUPIED = 0, MLM_UNKNOWN = -1 };

/* The YAML keys mlmap::init_map reads (src/mlmap.cpp:10-33,75-85); same meaning, same units.
 * Doubles that the reference casts to float (mlmap.cpp:77-81, mlmap.h:92) are cast the same way inside. */
typedef struct mlm_config {
    double am_d_rho;       /* mlmapping_am_d_Rho */
    double am_d_phi_deg;   /* mlmapping_am_d_Phi_deg */
    double am_d_z;         /* mlmapping_am_d_Z */
    int32_t am_n_rho;      /* mlmapping_am_n_Rho */
    int32_t am_n_z_below;  /* mlmapping_am_n_Z_below */
    int32_t am_n_z_over;   /* mlmapping_am_n_Z_over */
    int32_t use_raycasting;/* mlmapping_use_raycasting */
    double depth_noise_coe;/* mlmapping_depth_noise_coe */
    double subbox_d_xyz;   /* mlmapping_subbox_d_xyz */
    int32_t subbox_n;      /* mlmapping_subbox_n */
    int32_t use_exploration_frontiers; /* use_exploration_frontiers */
    double log_odds_min;   /* mlmapping_lm_log_odds_min */
    double log_odds_max;   /* mlmapping_lm_log_odds_max */
    double measurement_hit;/* mlmapping_lm_measurement_hit (stored, never used: map_local.cpp:128,159) */
    double measurement_miss;/* mlmapping_lm_measurement_miss */
    double occupied_sh;    /* mlmapping_lm_occupied_sh */
    int32_t inflate_n;     /* mlmapping_inflate_n */
    int32_t inflate_global_n; /* mlmapping_inflate_global_n */
    int32_t apply_inflate; /* mlmapping_apply_inflate */
    int32_t sample_cnt;    /* mlmapping_sample_cnt */
    double cam_cx, cam_cy, cam_fx, cam_fy; /* mlmapping_cam_* */
    double T_bs[16];       /* T_B_S, 4x4 row major (include/yamlRead.h:16-24) */
} mlm_config;

/* Sizing of the device-resident state (no reference counterpart: the reference grows std containers). */
typedef struct mlm_limits {
    int32_t max_blocks;      /* INITIAL capacity of the hashed block pool (n^3 cells each); 0 = default 65536.  The pool
                              * grows on demand (table and pool re-allocated at twice the size, blocks copied, table rebuilt
                              * on the device) like the reference's observed_group_map; MLM_ERR_CAPACITY only when the
                              * device cannot hold the larger pool (or with the test knob "pool_grow" = 0) */
    int32_t max_points;      /* largest point count of one frame; 0 = 1280*720 */
    int32_t max_batch;       /* frames integrated per launch sequence (batch entry points); 0 = 8, at most 64;
                              * every frame in flight owns a slot of scratch — ~0.06 GB at 640x480 / 0.1 m, ~0.25 GB at 1280x720 /
                              * 0.05 m for camera scenes: its lists start at what frames of max_points pixels need and are enlarged
                              * when a frame needs more (worst case 0.29 GB / 1.97 GB) — three sets of them (one being filled, one
                              * in the map-independent stage, one draining) */
    int32_t record_awareness;/* keep per-frame hit/miss lists readable via mlm_get_awareness_* (tests) */
} mlm_limits;

typedef struct mlm_handle mlm_handle;

/* Counters of the last integrated frame (the reference exposes the same facts as container sizes:
 * hit_idx_odds_hashmap.size(), miss_idx_set.size(), observed_group_map.size()). */
typedef struct mlm_frame_stats {
    int64_t n_points;      /* points fed (raw != 0) */
    int64_t n_hit_cells;   /* unique awareness hit cells  == hit_idx_odds_hashmap.size() */
    int64_t n_miss_cells;  /* unique awareness miss cells == miss_idx_set.size() */
    int64_t n_out_of_range;/* "point out range" branch, map_awareness.cpp:277 */
    int64_t n_blocks;      /* observed_group_map.size() */
    int64_t n_rehash_epochs; /* libstdc++ rehash epochs replayed for the hit container this frame (1 = none) */
    int64_t hit_bucket_count;/* emulated hit_idx_odds_hashmap.bucket_count() after this frame */
    /* device-side work counters (no reference counterpart) */
    int64_t n_multi_cells;   /* hit cells that received more than one kind of contribution (need the ordered replay) */
    int64_t n_contrib_slots; /* 16-padded contribution slots reserved for those cells */
    int64_t n_groups;        /* (wave, cell, kind) contribution groups (merged per cell in LDS before any global atomic) */
    int64_t n_rays;          /* rays walked after de-duplication */
    int64_t n_spec_replays;  /* frames so far whose Stage B had to be replayed with a rehash plan */
    int64_t n_device_atomics;   /* device-scope atomics the frame's Stage A issued, counted by the kernels (0 on the cell-table path) */
    int64_t n_sector_fallbacks; /* frames so far redone by the cell-table path (an azimuth sector overflowed its LDS tables) */
    int64_t logit_bit_exact;    /* 1: hit increments log10f(odd / (1 - odd)) carry the float bits of this host's libm (map_local.h:8);
                                 * 0: unknown libm, increments are FP64 log10 rounded once (last-place differences possible) */
    int64_t n_pool_grows;       /* times the block pool has grown so far (allocate_ram never refuses: map_local.h:215-231) */
    int64_t block_capacity;     /* blocks the pool holds now */
    int64_t n_graph_launches;   /* single frames so far submitted as one HIP-graph replay (synchronous mode: the reference's one frame
                                 * per depth callback, src/mlmap.cpp:463-507) */
    int64_t n_bin_exact_waves;  /* waves of the frame whose bins came from the reference's own FP64 sequence because a lane lay too near
                                 * a cell boundary for the certified cheap evaluation (k_bin_sectors; usually 0) */
    /* host mirror of the map (ABI 5): batches of up to a few hundred positions are answered on the host like the reference's inline
     * queries (include/mlmap.h:170-295), from a pinned copy of the block planes that is refreshed after the map changed */
    int64_t n_host_queries;     /* positions answered from the host mirror so far */
    int64_t n_mirror_refreshes; /* times the mirror was brought up to date (one kernel + one synchronisation each) */
    int64_t n_mirror_blocks;    /* blocks copied to the host by those refreshes */
    int64_t device_bytes;       /* device memory the handle holds now (map + frame slots + tables) */
    int64_t n_slot_grows;       /* times the frame slots' lists — sized by what frames need, not by the worst case — were enlarged */
} mlm_frame_stats;

/* replaces mlmap::init_map (src/mlmap.cpp:3-149), minus ROS plumbing */
int mlm_create(const mlm_config *cfg, const mlm_limits *limits_or_null, int device, mlm_handle **out);
int mlm_destroy(mlm_handle *h);
const char *mlm_last_error(mlm_handle *h);
int mlm_abi_version(void);

/* Use an externally owned HIP stream (e.g. the framework's current stream) instead of the handle's own for the
 * map-dependent stage, queries and exports.  Device inputs of the *_dev entry points may then be produced by work
 * enqueued on that stream before the call: the per-frame stage that reads them (which runs on streams of the handle)
 * is ordered behind it.  With the handle's own stream (the default) device inputs must be complete before the call. */
int mlm_set_stream(mlm_handle *h, void *hip_stream);

/* replaces mlmap::project_depth + update_map (src/mlmap.cpp:311-349,382-386).
 * img: uint16 millimetres (the 16UC1 image of mlmap.cpp:477-484), row_stride in pixels.
 * pixel_idx == NULL: dense — every pixel with raw != 0 in row-major order (v outer).
 * pixel_idx != NULL: exactly those pixels (v*width+u) in list order, raw == 0 skipped — lets a host reproduce
 * the reference's rand() sampler (mlmap.cpp:322-327) outside and keep parity. */
int mlm_integrate_depth_u16(mlm_handle *h, const uint16_t *img_host, int width, int height, int row_stride,
                            const int32_t *pixel_idx, int n_idx, const double q_wb[4], const double t_wb[3]);
/* same with the image already resident in device memory (HBM) */
int mlm_integrate_depth_u16_dev(mlm_handle *h, const uint16_t *img_dev, int width, int height, int row_stride,
                                const int32_t *pixel_idx_dev, int n_idx, const double q_wb[4],
                                const double t_wb[3]);
/* K frames of one stream, device resident, frame k at img_dev + k*frame_stride (in pixels); poses 4K / 3K doubles.
 * Frames are integrated in order (the update is order dependent). */
int mlm_integrate_depth_batch_dev(mlm_handle *h, const uint16_t *img_dev, int n_frames, size_t frame_stride,
                                  int width, int height, int row_stride, const double *q_wb, const double *t_wb);
/* same with host-resident frames (frame k at img_host + k*frame_stride); uploads overlap with compute */
int mlm_integrate_depth_batch(mlm_handle *h, const uint16_t *img_host, int n_frames, size_t frame_stride, int width,
                              int height, int row_stride, const double *q_wb, const double *t_wb);
/* replaces the body of mlmap::depth_odom_input_callback (src/mlmap.cpp:463-532) for a ROS-free host: depth is the
 * sensor_msgs/Image payload (encoding 32FC1 metres -> converted x1000 to 16UC1 like cv::Mat::convertTo, mlmap.cpp:480-483:
 * the dense path on the device, the sampler's few pixels on the host with the same float arithmetic; or
 * 16UC1 millimetres), odom_* / imu_w the nav_msgs/Odometry pose+twist and sensor_msgs/Imu angular velocity, stamps in
 * seconds; the pose is forwarded to the image stamp by the reference's linear model (mlmap.cpp:485-498).
 * sampled != 0: project_depth's rand() sampler (<= sample_cnt pixels, glibc rand(), v first, mlmap.cpp:322-327);
 * 0: dense.  T_wb_out (optional): compensated pose, q (w,x,y,z) then t. */
int mlm_integrate_callback(mlm_handle *h, const void *depth_host, int is_f32, int width, int height, double t_img,
                           const double odom_p[3], const double odom_q[4], const double odom_v[3], double t_odom,
                           const double imu_w[3], double t_imu, double camera2odom_latency, int sampled,
                           double T_wb_out[7]);
/* replaces awareness_map_cylindrical::input_pc_pose(PC_s, T_wb) + input_pc_pose_direct on an explicit
 * sensor-frame point list (include/map_awareness.h:74) */
int mlm_integrate_points(mlm_handle *h, const double *xyz_s_host, int n, const double q_wb[4],
                         const double t_wb[3]);

/* queries: include/mlmap.h:170-193 / :142-169 / :195-211 / :213-225 / :237-295 */
int mlm_query_occupancy(mlm_handle *h, const double *pos, int n, int8_t *out);
int mlm_query_occupancy_inflate(mlm_handle *h, const double *pos, int n, float inflate, int8_t *out);
int mlm_query_inflate_occupancy(mlm_handle *h, const double *pos, int n, int8_t *out);
int mlm_query_odds(mlm_handle *h, const double *pos, int n, float *out);
int mlm_query_odd_grad(mlm_handle *h, const double *pos, int n, int max_iter, double *out3);
/* float mlmap::getOdd(const Vec3I &glb_id, size_t subbox_id), include/mlmap.h:227-235: glb_id n x 3 block indices,
 * subbox_id n cell ids in [0, subbox_n^3) (out of range is undefined behaviour in the reference: MLM_ERR_INVALID here) */
int mlm_query_odds_at(mlm_handle *h, const int32_t *glb_id, const int32_t *subbox_id, int n, float *out);
/* src/mlmap.cpp:388-407 */
int mlm_set_free_in_bound(mlm_handle *h, const double box_min[3], const double box_max[3]);
/* mlmap::inflate_map (src/mlmap.cpp:286-309) around vehicle position ct_pos */
int mlm_inflate_map(mlm_handle *h, const double ct_pos[3]);

/* map read-out: what visualisers get by iterating local_map->observed_group_map (rviz_vis.cpp:280-321) */
int mlm_block_count(mlm_handle *h, int *n_out);
/* keys [cap*3], log_odds [cap*cells], occ / infl [cap*cells] ('u','f','o'); any pointer may be NULL; destinations may
 * be host or device memory (the global-map merge exports straight into device tensors) */
int mlm_export_blocks(mlm_handle *h, int cap, int32_t *keys, float *log_odds, uint8_t *occ, uint8_t *infl,
                      int *n_out);
/* collapsed [cap]: 1 for blocks the release scan (map_local.cpp:208-232) froze — their vectors have size 1 in the
 * reference, element 0 answers queries; always 0 unless use_exploration_frontiers */
int mlm_export_block_flags(mlm_handle *h, int cap, uint8_t *collapsed, int *n_out);
/* frontier cells as (gx,gy,gz,cell id) quadruples = the /frontier cloud before centre conversion (rviz_vis.cpp:267-293) */
int mlm_export_frontier(mlm_handle *h, int cap, int32_t *keys_cell, int *n_out);
/* float xyz of inflated-'o' cell centres = PointCloud2 payload of /global_map (rviz_vis.cpp:296-327), unordered */
int mlm_export_global_map(mlm_handle *h, int cap_points, float *xyz, int *n_out);
/* float xyz of the frontier cells' centres = PointCloud2 payload of /frontier (rviz_vis.cpp:267-293,
 * subbox_id2xyz_glb include/map_local.h:201-206), unordered; empty unless use_exploration_frontiers */
int mlm_export_frontier_points(mlm_handle *h, int cap_points, float *xyz, int *n_out);
/* Load blocks into the map (no reference counterpart: the reference never persists or merges maps; this is how a
 * merged global map, mlmapping_amd/merge.py, is put back behind the query interface).  keys [n*3]; log_odds / occ /
 * infl [n*cells] and collapsed [n] as mlm_export_blocks / mlm_export_block_flags write them, any of them may be NULL
 * (that plane keeps its current content; new blocks start as allocate_ram leaves them: 0 / 'u' / 'u'); sources may
 * be host or device memory.  Blocks already present are overwritten cell by cell, others are created. */
int mlm_import_blocks(mlm_handle *h, int n, const int32_t *keys, const float *log_odds, const uint8_t *occ,
                      const uint8_t *infl, const uint8_t *collapsed);

/* The two device-side steps of the optional global-map merge across GPUs (mlmapping_amd/merge.py; no reference
 * counterpart: SURVEY.md §8e).  All pointers are device memory; both calls return when the buffers are written.
 * pack:   row b of log_odds_dev / seen_dev [n*cells] = this map's cells of block keys_dev[3b..3b+2] (0 / 0 where the
 *         map does not hold the block; seen = occupancy != 'u').  These rows are what the ranks exchange and sum.
 * finish: summed log-odds clamped to [log_odds_min, log_odds_max]; occ = 'o' above occupied_sh, else 'f' where any
 *         rank had seen the voxel, else 'u'. */
int mlm_merge_pack(mlm_handle *h, const int32_t *keys_dev, int n, float *log_odds_dev, uint8_t *seen_dev);
int mlm_merge_finish(mlm_handle *h, float *log_odds_dev, const uint8_t *seen_dev, size_t n_cells, uint8_t *occ_dev);

/* Pin a caller-owned host buffer (hipHostRegister) so that the host-buffer entry points (mlm_integrate_depth_batch,
 * mlm_integrate_depth_u16, mlm_integrate_callback, mlm_integrate_points) DMA straight from it: from pageable memory a copy is
 * staged by the HIP runtime at a third of the link's rate.  A replay tool registers its frame buffer once; the ROS callback
 * pattern (one frame per call) does not need it.  Registering does not change the lifetime rule: every entry point is done with
 * the buffer when it returns (see mlm_set_async).  Unregister before freeing the buffer (waits for everything submitted). */
int mlm_host_register(mlm_handle *h, const void *ptr, size_t bytes);
int mlm_host_unregister(mlm_handle *h, const void *ptr);

int mlm_sync(mlm_handle *h);
/* async = 1: integrate calls return once the work is SUBMITTED (up to three batches may be in flight, one per slot set); errors of
 * a batch and mlm_get_frame_stats lag by one call; mlm_sync, queries and exports wait for everything.  Default 0: integrate
 * calls return when the map is updated (a lone frame's call returns on a completion ticket its last map-updating kernel writes to pinned
 * memory; in frontier mode the release scan of map_local.cpp:208-232 — which marks blocks, not voxels — may still be running then:
 * everything that reads the map afterwards is ordered behind it).  Host buffers stay BORROWED FOR THE CALL in both modes: an asynchronous call returns
 * only after its copies out of the caller's buffer have completed (also from a buffer pinned with mlm_host_register, whose
 * copies are truly asynchronous) — the buffer may be refilled as soon as the call returns.  Device inputs of the *_dev entry
 * points are read by the frames' kernels and must stay unmodified until mlm_sync (or until three further batches were submitted). */
int mlm_set_async(mlm_handle *h, int on);
/* Small query batches (a planner asking position by position, include/mlmap.h:170-295) are answered from a pinned HOST copy of the
 * block planes (6 bytes per voxel + 13 per block), which grows with the map.  max_bytes bounds that pinned memory (default 1 GiB;
 * 0: no host copy at all): a map that needs more is queried by kernels only, as large batches always are — same answers, ~20 us
 * per call instead of ~0.05 us.  Takes effect at the next query; lowering it below what is pinned frees the copy. */
int mlm_set_host_mirror_limit(mlm_handle *h, size_t max_bytes);
int mlm_get_frame_stats(mlm_handle *h, mlm_frame_stats *out);

/* test hooks (need limits.record_awareness): unique hit cells (linear cell idx, odd, first-touch time) and
 * unique miss cells of the LAST frame, unordered */
int mlm_get_awareness_hits(mlm_handle *h, int cap, uint32_t *cell_idx, float *odds, uint32_t *t_first, int *n_out);
int mlm_get_awareness_misses(mlm_handle *h, int cap, uint32_t *cell_idx, int *n_out);
/* derived constants, for cross-checking against the oracle: T_ls of the last frame, odds table [21*n_rho] */
int mlm_get_T_ls(mlm_handle *h, double q[4], double t[3]);
int mlm_get_odds_table(mlm_handle *h, float *out);

/* Device time of the launches of the integrate calls, measured with HIP events on the streams the kernels run on
 * (milliseconds); names are static strings.  on = 1: the list describes the last call only; on = 2: it accumulates over
 * calls until read (mlm_get_kernel_times with cap >= n consumes it); on = 3: like 2 but only every `every`-th
 * launch of ONE kernel is bracketed (mlm_set_timed_kernel, default "k_bin_points", 1; events around every kernel cost
 * ~19 % throughput) — bench.py uses 2 on a few batches to find the dominant kernel and 3 on it over its timed region; on = 4: only the spans of a batch's Stage A and Stage B+C ("stage_a_batch",
 * "stage_bc_batch"; the latter starts when the main stream reaches it, i.e. after the previous batch's).
 * Stage A kernels are launched once per batch, so one entry of theirs covers all frames of that batch. */
int mlm_get_kernel_times(mlm_handle *h, int cap, const char **names, float *ms, int *n_out);
int mlm_enable_kernel_timing(mlm_handle *h, int on);
int mlm_set_timed_kernel(mlm_handle *h, const char *name, int every);

/* Test and experiment knobs — NOT part of the drop-in contract.  Named integers read by the NEXT mlm_create of this process:
 * forced fall-backs ("sec_fail_every", "sec_backoff", "sectors"), simulated allocation failures ("debug_fail_slot"), a fixed pool
 * ("pool_grow"), slot layout ("lean_slots", "slot_sets"), launch geometries ("sec_tab", "sec_threads", "rank_grid", ...; the full
 * list is kKnobNames in mlmapping_amd/csrc/mlm_handle.h).  Unknown names: MLM_ERR_INVALID.  mlm_debug_reset forgets them all.  The library reads
 * no environment variable for behaviour; MLM_DEBUG_CREATE / MLM_DEBUG_ALLOC / MLM_DEBUG_DRAIN only print diagnostics. */
int mlm_debug_set(const char *name, long long value);
int mlm_debug_reset(void);
/* Host clocks of the synchronous single-frame path of mlm_integrate_callback (a development aid, like the knobs): microseconds
 * summed over the calls since the last reset — [0] pose compensation, drain of the previous call, sampling; [1] frame set-up up
 * to the launch; [2] the launch call (hipGraphLaunch); [3] host work between launch and wait; [4] waiting for the frame's ticket;
 * [5] the rest of the call. */
int mlm_debug_clocks(mlm_handle *h, double out_us[8], int reset);
/* Test hook for the binning kernel's cheap arithmetic (mlm_bin_point_fast, mlm_device.h): the largest relative errors, over 2^26
 * values, of the hardware's reciprocal and reciprocal-square-root seeds [0], [2] and of their once-refined forms [1], [3] — the
 * error budget of the certified margins assumes [1], [3] <= 4e-12. */
int mlm_debug_probe_seeds(mlm_handle *h, double out4[4]);

#ifdef __cplusplus
}
#endif
#endif
