// mlmap_facade.hpp — header-only C++ class with the reference's public `mlmap` names (include/mlmap.h:105-139)
// on top of the C ABI of mlmap_hip.h, so that a ROS host (or a planner that today does `#include <mlmap.h>`)
// can swap the CPU maps for the MI355X path without touching its call sites.
//
// No Eigen / ROS dependency: Vec3 is any type with operator[](int) -> double (Eigen::Vector3d qualifies).
// Not part of the hot path: every method is a thin forward to one mlm_* call.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mlmap_hip.h"

namespace mlmap_hip {

struct Vec3d {
    double v[3];
    double operator[](int i) const { return v[i]; }
    double &operator[](int i) { return v[i]; }
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

    // planners that query thousands of positions per cycle should use the batched entry points directly
    mlm_handle *handle() { return h_; }

  private:
    void check(int rc, const char *what) {
        if (rc != MLM_OK) throw std::runtime_error(std::string(what) + ": " + (h_ ? mlm_last_error(h_) : "no handle"));
    }
    mlm_handle *h_ = nullptr;
    const uint16_t *img_ = nullptr;
    int width_ = 0, height_ = 0, stride_ = 0;
    double q_[4] = {1, 0, 0, 0}, t_[3] = {0, 0, 0};
    std::vector<int32_t> pix_;
    bool sampled_ = false;
};

} // namespace mlmap_hip
