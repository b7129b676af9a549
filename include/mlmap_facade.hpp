// mlmap_facade.hpp — header-only C++ class with the reference's public `mlmap` names (include/mlmap.h:105-139)
// on top of the C ABI of mlmap_hip.h, so that a ROS host (or a planner that today does `#include <mlmap.h>`)
// can swap the CPU maps for the MI355X path without touching its call sites.
//
// No Eigen / ROS dependency: Vec3 is any type with operator[](int) -> double (Eigen::Vector3d qualifies).
// Not part of the hot path: every method is a thin forward to one mlm_* call.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "mlmap_hip.h"

namespace mlmap_hip {

struct Vec3d {
    double v[3];
    double operator[](int i) const { return v[i]; }
    double &operator[](int i) { return v[i]; }
};

// What visualisers read by iterating local_map->observed_group_map (src/rviz_vis.cpp:267-327, src/mlmap.cpp:226-276): a host
// SNAPSHOT with the reference's member names (include/map_local.h:42-60,92,201-213), so that code written against
// `local_map_cartesian *localmap` — `for (auto it = localmap->observed_group_map.begin(); ...)`, `it->second.inflate_occupancy`,
// `it->second.frontier`, `localmap->subbox_id2xyz_glb(it->first, id)` — compiles against `mlmap::local_map_snapshot().get()`.
struct Vec3I {
    int v[3];
    int operator[](int i) const { return v[i]; }
    int &operator[](int i) { return v[i]; }
    int size() const { return 3; }
    bool operator==(const Vec3I &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct PointF { // pcl::PointXYZ's payload
    float x, y, z;
};
struct local_map_view {
    struct VectorHasher { // include/map_local.h:42-52
        int operator()(const Vec3I &V) const {
            int hash = V.size();
            hash ^= V[0] + 0x9e3779b9 + (hash << 6) + (hash >> 2);
            hash ^= V[1] + 0x9e3779b9 + (hash << 6) + (hash >> 2);
            hash ^= V[2] + 0x9e3779b9 + (hash << 6) + (hash >> 2);
            return hash;
        }
    };
    struct subbox { // include/map_local.h:53-60; a released block holds ONE element per vector (map_local.cpp:221-226)
        std::vector<char> occupancy, inflate_occupancy;
        std::vector<float> log_odds;
        std::unordered_set<int> frontier;
    };
    std::unordered_map<Vec3I, subbox, VectorHasher> observed_group_map;
    double map_dxyz_obv_sub = 0, map_dxyz_obv_glb = 0, map_dxyz_obv_sub_half = 0;
    int subbox_nxyz = 0;
    // include/map_local.h:201-213
    PointF subbox_id2xyz_glb(const Vec3I &origin, int idx) const {
        const Vec3d c = subbox_id2xyz_glb_vec(origin, idx);
        return PointF{(float)c[0], (float)c[1], (float)c[2]};
    }
    Vec3d subbox_id2xyz_glb_vec(const Vec3I &origin, int idx) const {
        const int n = subbox_nxyz, cz = idx / (n * n), cy = (idx - cz * n * n) / n, cx = idx - cz * n * n - cy * n;
        return Vec3d{{origin[0] * map_dxyz_obv_glb + cx * map_dxyz_obv_sub + map_dxyz_obv_sub_half,
                      origin[1] * map_dxyz_obv_glb + cy * map_dxyz_obv_sub + map_dxyz_obv_sub_half,
                      origin[2] * map_dxyz_obv_glb + cz * map_dxyz_obv_sub + map_dxyz_obv_sub_half}};
    }
};

class mlmap {
  public:
    // include/mlmap.h:109-114
    enum { FREE = 1, OCCUPIED = 0, UNKNOWN = -1 };
    bool has_data = false;    // mlmap.h:115
    bool map_updated = false; // mlmap.h:116

    mlmap() = default;
    mlmap(const mlmap &) = delete;
    mlmap &operator=(const mlmap &) = delete;
    ~mlmap() {
        if (h_) mlm_destroy(h_);
    }

    // replaces init_map(ros::NodeHandle&) (src/mlmap.cpp:3-149): the caller reads the YAML keys into mlm_config
    void init_map(const mlm_config &cfg, int device = 0, const mlm_limits *limits = nullptr) {
        if (h_) mlm_destroy(h_);
        h_ = nullptr;
        cfg_ = cfg;
        const int rc = mlm_create(&cfg, limits, device, &h_);
        if (rc != MLM_OK) {
            std::string msg = h_ ? mlm_last_error(h_) : "";
            if (h_) mlm_destroy(h_);
            h_ = nullptr;
            throw std::runtime_error("mlm_create failed (" + std::to_string(rc) + "): " + msg);
        }
        width_ = height_ = 0;
    }

    // What depth_odom_input_callback hands to project_depth()/update_map() (src/mlmap.cpp:484,494,504-507):
    // the 16UC1 depth image and the compensated pose T_wb = (q_wb (w,x,y,z), t_wb).
    void set_depth_image(const uint16_t *img, int width, int height, int row_stride = 0) {
        img_ = img;
        width_ = width;
        height_ = height;
        stride_ = row_stride ? row_stride : width;
        has_data = true;
    }
    void set_pose(const double q_wb[4], const double t_wb[3]) {
        for (int i = 0; i < 4; ++i) q_[i] = q_wb[i];
        for (int i = 0; i < 3; ++i) t_[i] = t_wb[i];
    }
    // project_depth (src/mlmap.cpp:311-349): the reference draws <= sample_cnt pixels with rand(); a host that wants
    // bit parity with it passes the same pixel list, otherwise the whole image is integrated.
    void project_depth(const std::vector<int32_t> *pixel_idx = nullptr) {
        if (pixel_idx)
            pix_ = *pixel_idx;
        else
            pix_.clear();
        sampled_ = pixel_idx != nullptr;
    }
    // update_map (src/mlmap.cpp:382-386)
    void update_map() {
        check(mlm_integrate_depth_u16(h_, img_, width_, height_, stride_, sampled_ ? pix_.data() : nullptr,
                                      (int)pix_.size(), q_, t_),
              "mlm_integrate_depth_u16");
        map_updated = true;
    }

    // src/mlmap.cpp:388-407
    template <class V3> void setFree_map_in_bound(const V3 &box_min, const V3 &box_max) {
        const double a[3] = {box_min[0], box_min[1], box_min[2]}, b[3] = {box_max[0], box_max[1], box_max[2]};
        check(mlm_set_free_in_bound(h_, a, b), "mlm_set_free_in_bound");
    }
    // include/mlmap.h:170-193
    template <class V3> int getOccupancy(const V3 &pos_w) {
        const double p[3] = {pos_w[0], pos_w[1], pos_w[2]};
        int8_t r = 0;
        check(mlm_query_occupancy(h_, p, 1, &r), "mlm_query_occupancy");
        return r;
    }
    // include/mlmap.h:142-169
    template <class V3> int getOccupancy(const V3 &pos_w, float inflate) {
        const double p[3] = {pos_w[0], pos_w[1], pos_w[2]};
        int8_t r = 0;
        check(mlm_query_occupancy_inflate(h_, p, 1, inflate, &r), "mlm_query_occupancy_inflate");
        return r;
    }
    // include/mlmap.h:195-211
    template <class V3> int getInflateOccupancy(const V3 &pos_w) {
        const double p[3] = {pos_w[0], pos_w[1], pos_w[2]};
        int8_t r = 0;
        check(mlm_query_inflate_occupancy(h_, p, 1, &r), "mlm_query_inflate_occupancy");
        return r;
    }
    // include/mlmap.h:213-225
    template <class V3> float getOdd(const V3 &pos_w) {
        const double p[3] = {pos_w[0], pos_w[1], pos_w[2]};
        float r = 0.5f;
        check(mlm_query_odds(h_, p, 1, &r), "mlm_query_odds");
        return r;
    }
    // include/mlmap.h:227-235: float getOdd(const Vec3I &glb_id, size_t subbox_id)
    template <class V3I> float getOdd(const V3I &glb_id, size_t subbox_id) {
        const int32_t g[3] = {(int32_t)glb_id[0], (int32_t)glb_id[1], (int32_t)glb_id[2]};
        const int32_t c = (int32_t)subbox_id;
        float r = 0.5f;
        check(mlm_query_odds_at(h_, g, &c, 1, &r), "mlm_query_odds_at");
        return r;
    }
    // include/mlmap.h:237-295
    template <class V3> Vec3d getOddGrad(const V3 &pos_w, size_t max_iter = 5) {
        const double p[3] = {pos_w[0], pos_w[1], pos_w[2]};
        Vec3d g{};
        check(mlm_query_odd_grad(h_, p, 1, (int)max_iter, g.v), "mlm_query_odd_grad");
        return g;
    }
    // src/mlmap.cpp:286-309 (ct_pos = vehicle position, set by the odometry callback, mlmap.cpp:485-487)
    template <class V3> void inflate_map(const V3 &ct_pos) {
        const double p[3] = {ct_pos[0], ct_pos[1], ct_pos[2]};
        check(mlm_inflate_map(h_, p), "mlm_inflate_map");
    }

    // `local_map` for visualisers: a snapshot of observed_group_map (one D2H of the block planes; waits for everything submitted)
    std::shared_ptr<local_map_view> local_map_snapshot() {
        auto lm = std::make_shared<local_map_view>();
        const int n = cfg_.subbox_n, C = n * n * n;
        lm->subbox_nxyz = n;
        lm->map_dxyz_obv_sub = cfg_.subbox_d_xyz;
        lm->map_dxyz_obv_glb = cfg_.subbox_d_xyz * n; // map_local.cpp:60
        lm->map_dxyz_obv_sub_half = cfg_.subbox_d_xyz * 0.5;
        int nb = 0;
        check(mlm_block_count(h_, &nb), "mlm_block_count");
        std::vector<int32_t> keys((size_t)nb * 3);
        std::vector<float> lo((size_t)nb * C);
        std::vector<uint8_t> occ((size_t)nb * C), infl((size_t)nb * C), col((size_t)nb);
        int m = 0;
        if (nb) {
            check(mlm_export_blocks(h_, nb, keys.data(), lo.data(), occ.data(), infl.data(), &m), "mlm_export_blocks");
            check(mlm_export_block_flags(h_, nb, col.data(), &m), "mlm_export_block_flags");
        }
        lm->observed_group_map.reserve((size_t)nb);
        for (int b = 0; b < nb; ++b) {
            auto &sb = lm->observed_group_map[Vec3I{{keys[3 * (size_t)b], keys[3 * (size_t)b + 1], keys[3 * (size_t)b + 2]}}];
            const size_t cnt = col[(size_t)b] ? 1 : (size_t)C, o = (size_t)b * C;
            sb.occupancy.assign(occ.begin() + o, occ.begin() + o + cnt);
            sb.inflate_occupancy.assign(infl.begin() + o, infl.begin() + o + cnt);
            sb.log_odds.assign(lo.begin() + o, lo.begin() + o + cnt);
        }
        int nf = 0;
        check(mlm_export_frontier(h_, 0, nullptr, &nf), "mlm_export_frontier");
        if (nf) {
            std::vector<int32_t> fr((size_t)nf * 4);
            check(mlm_export_frontier(h_, nf, fr.data(), &nf), "mlm_export_frontier");
            for (int i = 0; i < nf; ++i) {
                auto it = lm->observed_group_map.find(Vec3I{{fr[4 * (size_t)i], fr[4 * (size_t)i + 1], fr[4 * (size_t)i + 2]}});
                if (it != lm->observed_group_map.end()) it->second.frontier.insert(fr[4 * (size_t)i + 3]);
            }
        }
        return lm;
    }

    // planners that query thousands of positions per cycle should use the batched entry points directly
    mlm_handle *handle() { return h_; }

  private:
    void check(int rc, const char *what) {
        if (rc != MLM_OK) throw std::runtime_error(std::string(what) + ": " + (h_ ? mlm_last_error(h_) : "no handle"));
    }
    mlm_handle *h_ = nullptr;
    mlm_config cfg_{};
    const uint16_t *img_ = nullptr;
    int width_ = 0, height_ = 0, stride_ = 0;
    double q_[4] = {1, 0, 0, 0}, t_[3] = {0, 0, 0};
    std::vector<int32_t> pix_;
    bool sampled_ = false;
};

} // namespace mlmap_hip
