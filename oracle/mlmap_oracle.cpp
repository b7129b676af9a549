/*
 * mlmap_oracle.cpp — CPU oracle for the MLMapping per-frame map update.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see mlmap_oracle.h).  "parity unpinned" by reference tests
 * (none exist); pinned against the SURVEY.md §8d known-answer counts by tests/test_oracle_kat.py.
 *
 * This is a restatement, not a copy: no Eigen, no Sophus, no ROS, no PCL.  It keeps what the
 * reference's results depend on:
 *   - FP64 arithmetic in the reference's operation order, built with -ffp-contract=off and no -march
 *     (reference flags: CMakeLists.txt:4 "-std=c++17 -O3", i.e. SSE2, no FMA);
 *   - float/double mixing exactly where the reference mixes them;
 *   - std::unordered_map / std::unordered_set with the reference's hashers, because
 *     local_map_cartesian::input_pc_pose_direct applies contributions in container ITERATION order
 *     (map_local.cpp:147,176) and the result depends on that order (upper-only clamp + sticky 'o').
 * Eigen itself is not under /root/reference (find_package(Eigen3), version unpinned,
 * CMakeLists.txt:7); its quaternion formulas are restated from Eigen 3.3/3.4's generic
 * (non-SIMD) code paths: quat_product, normalize = coeffs / sqrt(squaredNorm),
 * _transformVector = v + w*(2 q x v) + q x (2 q x v), Quaternion(Matrix3) = Shepperd.
 *
 * Citations are relative to /root/reference.
 */
#include "mlmap_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct V3 {
    double x, y, z;
};
struct V3i {
    int x, y, z;
    bool operator==(const V3i &o) const { return x == o.x && y == o.y && z == o.z; }
    int operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    int &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3i operator+(const V3i &a, const V3i &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

/* x86-64 static_cast<int>(double) is cvttsd2si: NaN / out-of-range -> INT_MIN ("integer indefinite").
 * The reference relies on it implicitly (e.g. phi = NaN when x = y = 0, map_awareness.cpp:95). */
static inline int cvt_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int)v;
}

/* ---- Eigen::Quaterniond / Sophus SO3, SE3 (so3.cpp:36-96, se3.cpp:29-95) ------------------------ */
struct Quat {
    double w, x, y, z;
};
static inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
/* Which of Eigen's code paths the quaternion arithmetic follows (mlo_set_quat_arch).  Eigen is NOT under /root/reference (unpinned
 * system dependency, CMakeLists.txt:7), so both are restated from its published sources:
 *   0 (default) — the generic path: internal::quat_product<Architecture::Generic> and a sequential squaredNorm;
 *   1 — what an x86-64 build of the reference (SSE2 is baseline; CMakeLists.txt:4 sets no -march) gets from Eigen 3.3 / 3.4:
 *       quat_product<Architecture::SSE, ..., double> (Eigen/src/Geometry/arch/Geometry_SSE.h) evaluates, with two-lane packets,
 *           t1 = aw*b.xy + ay*b.zw,  t2 = az*b.xy - ax*b.zw,  res.xy = t1 + (-swap(t2).lane0, +swap(t2).lane1)
 *           t1 = aw*b.zw - ay*b.xy,  t2 = az*b.zw + ax*b.xy,  res.zw = t1 - (-swap(t2).lane0, +swap(t2).lane1)
 *       i.e. the same four products per coefficient in a different association, and the vectorised redux of squaredNorm adds the
 *       squares lane-wise first: (x^2 + z^2) + (y^2 + w^2).
 * tests/test_quat_arch.py reports how many T_ls bits and how many map cells differ between the two on the parity fixtures. */
static int g_quat_arch = 0;
static inline Quat quat_mul(const Quat &a, const Quat &b) {
    if (g_quat_arch == 1)
        return {(a.w * b.w - a.y * b.y) - (a.z * b.z + a.x * b.x), (a.w * b.x + a.y * b.z) - (a.z * b.y - a.x * b.w),
                (a.w * b.y + a.y * b.w) + (a.z * b.x - a.x * b.z), (a.w * b.z - a.y * b.x) + (a.z * b.w + a.x * b.y)};
    /* Eigen generic quat_product (Quaternion.h, internal::quat_product<Architecture::Generic>) */
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
/* QuaternionBase::normalize: m_coeffs /= norm(), coefficient order (x,y,z,w), sequential sum */
static inline Quat quat_normalized(const Quat &q) {
    double n = g_quat_arch == 1 ? std::sqrt((q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w))
                                : std::sqrt(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
/* QuaternionBase::_transformVector */
static inline V3 quat_rot(const Quat &q, const V3 &v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv = uv + uv;
    return (v + uv * q.w) + cross(qv, uv);
}
/* Eigen Quaternion(Matrix3) — quaternionbase_assign_impl<Mat,3,3> (Shepperd); m row-major 3x3 */
static Quat quat_from_R(const double m[9]) {
    auto M = [&](int r, int c) { return m[r * 3 + c]; };
    Quat q;
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (M(2, 1) - M(1, 2)) * t;
        q.y = (M(0, 2) - M(2, 0)) * t;
        q.z = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (M(k, j) - M(j, k)) * t;
        v[j] = (M(j, i) + M(i, j)) * t;
        v[k] = (M(k, i) + M(i, k)) * t;
        q.x = v[0];
        q.y = v[1];
        q.z = v[2];
    }
    return q;
}
struct SE3 {
    Quat q{1, 0, 0, 0};
    V3 t{0, 0, 0};
};
/* SE3::operator* (se3.cpp:59-66): t += so3*other.t ; so3 *= other.so3 (so3.cpp:73-78 renormalises) */
static inline SE3 se3_mul(const SE3 &a, const SE3 &b) {
    SE3 r;
    r.t = a.t + quat_rot(a.q, b.t);
    r.q = quat_normalized(quat_mul(a.q, b.q));
    return r;
}
/* SE3::inverse (se3.cpp:76-83); SO3::inverse = SO3(conjugate) which normalises (so3.cpp:43-47,86-90) */
static inline SE3 se3_inv(const SE3 &a) {
    SE3 r;
    r.q = quat_normalized(Quat{a.q.w, -a.q.x, -a.q.y, -a.q.z});
    r.t = quat_rot(r.q, a.t * -1.);
    return r;
}
/* SE3::operator*(Vector3d) (se3.cpp:91-95) */
static inline V3 se3_apply(const SE3 &a, const V3 &p) { return quat_rot(a.q, p) + a.t; }

/* ---- hashers ------------------------------------------------------------------------------------ */
/* VectorHasher, map_awareness.h:31-41 == map_local.h:42-52.  int result -> size_t sign-extends. */
struct VectorHasher {
    int operator()(const V3i &V) const {
        int hash = 3; /* V.size() */
        hash ^= V.x + 0x9e3779b9 + (hash << 6) + (hash >> 2);
        hash ^= V.y + 0x9e3779b9 + (hash << 6) + (hash >> 2);
        hash ^= V.z + 0x9e3779b9 + (hash << 6) + (hash >> 2);
        return hash;
    }
};

/* ---- awareness_map_cylindrical (map_awareness.h, map_awareness.cpp) ----------------------------- */
struct Awareness {
    int nRho_x_nPhi = 0;
    SE3 T_bs;
    bool visibility_check = true;
    double map_dRho = 0, map_dPhi = 0, map_dZ = 0, z_border_min = 0, noise_coe_ = 0;
    int map_nRho = 0, map_nPhi = 0, map_nZ = 0, map_center_z_idx = 0;
    int diff_range = 10;
    std::vector<std::vector<float>> get_odds_table;
    std::vector<double> cos_phi, sin_phi; /* per-phi factors of CYLINDRICAL_CELL::center_pt */
    SE3 T_wa, T_ls;
    std::unordered_map<V3i, float, VectorHasher> hit_idx_odds_hashmap;
    std::unordered_set<size_t> miss_idx_set;
    size_t out_of_range = 0;

    /* map_awareness.h:81-84 */
    size_t mapIdx(int Rho, int Phi, int z) const {
        return static_cast<size_t>(z * nRho_x_nPhi + Phi * map_nRho + Rho);
    }
    /* map_awareness.h:120-124 */
    float sigma_in_dr(size_t x) const {
        float dis = (x * map_dRho);
        return noise_coe_ * dis * dis / map_dRho;
    }
    /* map_awareness.h:126-146.  fabs(float)->float, exp(float)->expf (float overloads under
     * `using namespace std`). */
    float standard_ND(float x) const {
        double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
        double p = 0.3275911;
        int sign = 1;
        if (x < 0) sign = -1;
        x = std::fabs(x) / std::sqrt(2.0);
        double t = 1.0 / (1.0 + p * x);
        double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
        return 0.5 * (1.0 + sign * y);
    }
    /* map_awareness.cpp:119-132 */
    float get_odds(int diff, size_t r) const {
        if (r == 0) r = 1;
        float up = standard_ND(static_cast<float>(diff + 0.5) / sigma_in_dr(r));
        float down = standard_ND(static_cast<float>(diff - 0.5) / sigma_in_dr(r));
        float res = up - down < 0.001 ? 0.001 : up - down;
        res = res >= 0.999 ? 0.999 : res;
        return res;
    }
    /* per-cell constants of the fill loop, map_awareness.cpp:47-78, computed on demand */
    double slope_of(int rho, int z) const {
        if (rho > 0) return (z - map_center_z_idx) / (rho * 1.0);
        return 0;
    }
    V3 center_of(int rho, int phi, int z) const {
        double center_z = z_border_min + (map_dZ / 2) + (z * map_dZ);
        double center_rho = map_dRho / 2 + (rho * map_dRho);
        return {center_rho * cos_phi[phi], center_rho * sin_phi[phi], center_z};
    }
    V3 center_of_idx(size_t idx) const {
        int z = (int)(idx / (size_t)nRho_x_nPhi);
        int rem = (int)(idx % (size_t)nRho_x_nPhi);
        return center_of(rem % map_nRho, rem / map_nRho, z);
    }

    /* map_awareness.cpp:19-82 */
    void init_map(double d_Rho, double d_Phi_deg, double d_Z, int n_Rho, int n_z_below, int n_z_over,
                  bool apply_raycasting, double noise_coe) {
        noise_coe_ = noise_coe;
        map_dRho = d_Rho;
        map_dPhi = d_Phi_deg * M_PI / 180;
        map_dZ = d_Z;
        map_nRho = n_Rho;
        map_nPhi = static_cast<int>(360 / d_Phi_deg);
        map_nZ = n_z_below + n_z_over + 1;
        map_center_z_idx = n_z_below;
        z_border_min = -(n_z_below * d_Z) - 0.5 * d_Z;
        nRho_x_nPhi = map_nRho * map_nPhi;
        diff_range = 10;
        get_odds_table.clear();
        for (int diff = -diff_range; diff < diff_range + 1; diff++) {
            std::vector<float> line;
            for (int r = 0; r < n_Rho; r++) line.emplace_back(get_odds(diff, r));
            get_odds_table.emplace_back(line);
        }
        cos_phi.resize(map_nPhi);
        sin_phi.resize(map_nPhi);
        for (int phi = 0; phi < map_nPhi; phi++) {
            double center_phi = map_dPhi / 2 + (phi * map_dPhi);
            cos_phi[phi] = std::cos(center_phi);
            sin_phi[phi] = std::sin(center_phi);
        }
        visibility_check = apply_raycasting;
    }

    /* map_awareness.h:115-118 */
    static double fast_atan(double x) { return x * (45 - (x - 1) * (14 + 3.83 * x)); }
    /* map_awareness.h:86-113; `deg2rad` is the unparenthesised macro M_PI / 180 (map_awareness.h:7) */
    static double fast_atan2(double y, double x) {
        double input = y / x;
        double a_input = std::abs(input);
        double res;
        if (a_input > 1)
            res = std::copysign(M_PI / 180 * (90 - fast_atan(1 / a_input)), input);
        else
            res = std::copysign(M_PI / 180 * fast_atan(a_input), input);
        if (x > 0)
            return res;
        else if (y >= 0)
            return res + M_PI;
        else
            return res - M_PI;
    }
    /* map_awareness.cpp:84-107 */
    bool xyz2RhoPhiZwithBoderCheck(const V3 &xyz_l, V3i &rhophiz, bool &can_do_cast) const {
        double rho = std::sqrt(xyz_l.x * xyz_l.x + xyz_l.y * xyz_l.y); /* pow(x,2) == x*x */
        int rho_idx = cvt_int(rho / map_dRho);
        double phi = fast_atan2(xyz_l.y, xyz_l.x);
        if (phi < 0) phi += 2 * M_PI;
        int phi_idx = cvt_int(phi / map_dPhi);
        double z = xyz_l.z - z_border_min;
        int z_idx = cvt_int(std::floor(z / map_dZ));
        rhophiz = V3i{rho_idx, phi_idx, z_idx};
        can_do_cast = (rho_idx >= 0 && phi_idx >= 0 && phi_idx < map_nPhi);
        if (can_do_cast && z_idx >= 0 && rho_idx < map_nRho && z_idx < map_nZ) return true;
        return false;
    }
    /* map_awareness.h:147-154 — noisy-OR in float */
    void update_odds_hashmap(const V3i &rpz_idx, float odd) {
        if (hit_idx_odds_hashmap.find(rpz_idx) == hit_idx_odds_hashmap.end())
            hit_idx_odds_hashmap[rpz_idx] = odd;
        else
            hit_idx_odds_hashmap[rpz_idx] = 1 - (1 - hit_idx_odds_hashmap[rpz_idx]) * (1 - odd);
    }
    /* map_awareness.cpp:135-171.  The reference indexes get_odds_table[±diff_r + 10] without a bound
     * (UB once 3*sigma > 10); the oracle stops at diff_r == diff_range and documents it. */
    void update_hits(const V3i &rpz_idx) {
        int raycasting_z;
        double raycasting_rate = slope_of(rpz_idx.x, rpz_idx.z);
        float odd;
        update_odds_hashmap(rpz_idx, get_odds_table[0 + diff_range][rpz_idx.x]);
        for (auto diff_r = 1; diff_r < 3 * sigma_in_dr(rpz_idx.x) && (rpz_idx.x + diff_r < map_nRho); diff_r++) {
            if (diff_r > diff_range) break; /* documented deviation: reference is UB here */
            raycasting_z = cvt_int(std::round(rpz_idx.z + (diff_r * raycasting_rate)));
            odd = get_odds_table[diff_r + diff_range][rpz_idx.x];
            if (0 <= raycasting_z && raycasting_z < map_nZ)
                update_odds_hashmap(V3i{rpz_idx.x + diff_r, rpz_idx.y, raycasting_z}, odd);
            odd = get_odds_table[-diff_r + diff_range][rpz_idx.x];
            raycasting_z = cvt_int(std::round(rpz_idx.z - (diff_r * raycasting_rate)));
            if (0 <= raycasting_z && raycasting_z < map_nZ) {
                if (rpz_idx.x - diff_r < 0) continue; /* documented deviation: reference would key rho < 0 */
                update_odds_hashmap(V3i{rpz_idx.x - diff_r, rpz_idx.y, raycasting_z}, odd);
            }
        }
    }
    /* map_awareness.cpp:173-282 */
    void input_pc_pose(const std::vector<V3> &PC_s, const SE3 &T_wb) {
        hit_idx_odds_hashmap.clear();
        miss_idx_set.clear();
        out_of_range = 0;
        T_wa = SE3{quat_normalized(Quat{1, 0, 0, 0}), T_wb.t};
        SE3 T_ws = se3_mul(T_wb, T_bs);
        T_ls = se3_mul(se3_inv(T_wa), T_ws);
        for (const V3 &p_s : PC_s) {
            V3 p_l = se3_apply(T_ls, p_s);
            V3i rpz_idx;
            bool can_do_cast;
            bool inside_range = xyz2RhoPhiZwithBoderCheck(p_l, rpz_idx, can_do_cast);
            if (inside_range) update_hits(rpz_idx);
            if (can_do_cast && visibility_check) {
                double raycasting_rate;
                if (inside_range) {
                    raycasting_rate = slope_of(rpz_idx.x, rpz_idx.z);
                } else {
                    if (rpz_idx.x > 0)
                        raycasting_rate = (rpz_idx.z - map_center_z_idx) / (rpz_idx.x * 1.0);
                    else
                        raycasting_rate = 0;
                }
                if (rpz_idx.x >= map_nRho) {
                    rpz_idx.z = cvt_int(std::round(rpz_idx.z - ((rpz_idx.x - map_nRho + 1) * raycasting_rate)));
                    rpz_idx.x = map_nRho - 1;
                }
                for (int r = rpz_idx.x - 1; r > 0; r--) {
                    int diff_r = rpz_idx.x - r;
                    int raycasting_z = cvt_int(std::round(rpz_idx.z - (diff_r * raycasting_rate)));
                    if (0 <= raycasting_z && raycasting_z < map_nZ)
                        miss_idx_set.emplace(mapIdx(r, rpz_idx.y, raycasting_z));
                }
            } else {
                out_of_range++; /* reference prints "point out range" (map_awareness.cpp:277-278) */
            }
        }
    }
};

/* ---- local_map_cartesian (map_local.h, map_local.cpp) ------------------------------------------- */
struct Subbox { /* map_local.h:53-60 */
    std::vector<char> occupancy;
    std::vector<char> inflate_occupancy;
    std::vector<float> log_odds;
    std::unordered_set<int> frontier;
};
struct Nbr {
    int d[6][4];
};

struct LocalMap {
    float log_odds_hit = 0, log_odds_miss = 0, log_odds_occupied_sh = 0, log_odds_max = 0, log_odds_min = 0;
    int inflate_n = 3;
    bool apply_inflate = false;
    double flate_height = 0.1; /* map_local.h:65 */
    bool apply_explored_area = false;
    double map_dxyz_obv_glb = 0, map_dxyz_obv_sub = 0, map_dxyz_obv_sub_half = 0;
    size_t cell_num_subbox = 0;
    int subbox_nxyz = 0;
    int ram_expand_cnt = 0, obs_cnt = 0;
    std::vector<double> global_bd = std::vector<double>(6);
    std::vector<V3i> subbox_id2xyz_table;
    std::vector<V3> nbr_disp_real;
    std::unordered_map<V3i, Subbox, VectorHasher> observed_group_map;
    std::vector<Nbr> subbox_neighbors;
    std::unordered_set<V3i, VectorHasher> observed_subboxes;

    /* subbox_cell_id_table lookup (map_local.cpp:63-75); a missing key is default-inserted as 0 by
     * operator[] (map_local.h:170), which is what any out-of-[0,n) component yields. */
    size_t cell_id(const V3i &c) const {
        if (c.x < 0 || c.y < 0 || c.z < 0 || c.x >= subbox_nxyz || c.y >= subbox_nxyz || c.z >= subbox_nxyz) return 0;
        return (size_t)(c.z * subbox_nxyz * subbox_nxyz + c.y * subbox_nxyz + c.x);
    }
    /* map_local.cpp:46-139 */
    void init_map(double d_xyz_in, unsigned int subbox_n, float lo_min, float lo_max, float lo_hit, float lo_miss,
                  float lo_sh, bool if_apply_explor) {
        map_dxyz_obv_sub = d_xyz_in;
        map_dxyz_obv_sub_half = map_dxyz_obv_sub * 0.5;
        subbox_nxyz = subbox_n;
        map_dxyz_obv_glb = map_dxyz_obv_sub * subbox_nxyz;
        cell_num_subbox = (size_t)std::pow(subbox_nxyz, 3);
        for (int i = 0; i < subbox_nxyz; i++)
            for (int j = 0; j < subbox_nxyz; j++)
                for (int k = 0; k < subbox_nxyz; k++) subbox_id2xyz_table.push_back(V3i{k, j, i});
        const V3i nbr_disp[6] = {{0, 0, 1}, {0, 0, -1}, {0, 1, 0}, {0, -1, 0}, {1, 0, 0}, {-1, 0, 0}};
        nbr_disp_real = {{0, 0, map_dxyz_obv_sub}, {0, 0, -map_dxyz_obv_sub}, {0, map_dxyz_obv_sub, 0},
                         {0, -map_dxyz_obv_sub, 0}, {map_dxyz_obv_sub, 0, 0},  {-map_dxyz_obv_sub, 0, 0}};
        for (int i = 0; i < subbox_nxyz; i++)
            for (int j = 0; j < subbox_nxyz; j++)
                for (int k = 0; k < subbox_nxyz; k++) {
                    Nbr nbrs;
                    for (int n = 0; n < 6; n++) {
                        V3i glb_disp{0, 0, 0};
                        V3i temp_id = nbr_disp[n] + V3i{k, j, i};
                        for (int m = 0; m < 3; m++) {
                            if (temp_id[m] >= subbox_nxyz) {
                                glb_disp.at(m) = 1;
                                temp_id.at(m) = 0;
                            } else if (temp_id[m] < 0) {
                                glb_disp.at(m) = -1;
                                temp_id.at(m) = subbox_nxyz - 1;
                            }
                        }
                        nbrs.d[n][0] = glb_disp.x;
                        nbrs.d[n][1] = glb_disp.y;
                        nbrs.d[n][2] = glb_disp.z;
                        nbrs.d[n][3] = (int)cell_id(temp_id);
                    }
                    subbox_neighbors.push_back(nbrs);
                }
        apply_explored_area = if_apply_explor;
        global_bd = {-30, 30, -30, 30, 0, 5}; /* map_local.cpp:124 */
        log_odds_min = lo_min;
        log_odds_max = lo_max;
        log_odds_hit = lo_hit; /* stored, never used (map_local.cpp:128,159) */
        log_odds_miss = lo_miss;
        log_odds_occupied_sh = lo_sh;
    }
    /* map_local.h:148-152,167-173 */
    void get_global_idx(const V3 &pt_w, V3i &glb_idx, size_t &subbox_id) const {
        glb_idx = V3i{cvt_int(std::floor(pt_w.x / map_dxyz_obv_glb)), cvt_int(std::floor(pt_w.y / map_dxyz_obv_glb)),
                      cvt_int(std::floor(pt_w.z / map_dxyz_obv_glb))};
        subbox_id = cell_id(V3i{cvt_int(std::floor(pt_w.x / map_dxyz_obv_sub) - glb_idx.x * subbox_nxyz),
                                cvt_int(std::floor(pt_w.y / map_dxyz_obv_sub) - glb_idx.y * subbox_nxyz),
                                cvt_int(std::floor(pt_w.z / map_dxyz_obv_sub) - glb_idx.z * subbox_nxyz)});
    }
    /* map_local.h:160-165 */
    bool inside_exp_bd(const V3 &p) const {
        return (p.x >= global_bd[0] && p.x < global_bd[1] && p.y >= global_bd[2] && p.y < global_bd[3] &&
                p.z >= global_bd[4] && p.z < global_bd[5]);
    }
    /* map_local.h:208-213 */
    V3 subbox_id2xyz_glb_vec(const V3i &origin, int idx) const {
        const V3i &c = subbox_id2xyz_table[idx];
        return {origin.x * map_dxyz_obv_glb + c.x * map_dxyz_obv_sub + map_dxyz_obv_sub_half,
                origin.y * map_dxyz_obv_glb + c.y * map_dxyz_obv_sub + map_dxyz_obv_sub_half,
                origin.z * map_dxyz_obv_glb + c.z * map_dxyz_obv_sub + map_dxyz_obv_sub_half};
    }
    /* map_local.h:215-231 */
    bool allocate_ram(const V3i &glb_idx) {
        if (observed_group_map.find(glb_idx) == observed_group_map.end()) {
            observed_group_map[glb_idx].occupancy.resize(cell_num_subbox, 'u');
            observed_group_map[glb_idx].inflate_occupancy.resize(cell_num_subbox, 'u');
            observed_group_map[glb_idx].log_odds.resize(cell_num_subbox, 0);
            observed_group_map[glb_idx].frontier.clear();
            ram_expand_cnt++;
            return true;
        } else if (observed_group_map[glb_idx].occupancy.size() == 1)
            return false;
        return true;
    }
    /* map_local.cpp:7-33 */
    void update_observation(const V3i &glb_idx, size_t subbox_id, const V3 &pt_w) {
        if (!inside_exp_bd(pt_w)) return;
        observed_subboxes.emplace(glb_idx);
        for (int i = 0; i < 6; i++) {
            V3 pt_w_nb = pt_w + nbr_disp_real[i];
            const Nbr &nb = subbox_neighbors[subbox_id];
            V3i glb_idx_nb = glb_idx + V3i{nb.d[i][0], nb.d[i][1], nb.d[i][2]};
            size_t subbox_id_nb = (size_t)nb.d[i][3];
            if (inside_exp_bd(pt_w_nb) && allocate_ram(glb_idx_nb) &&
                (observed_group_map[glb_idx_nb].occupancy[subbox_id_nb] == 'u')) {
                observed_group_map[glb_idx_nb].frontier.emplace((int)subbox_id_nb);
                break;
            }
        }
    }
    /* map_local.h:233-264 */
    void inflate_atpos(const V3i &glb_idx, size_t subbox_id) {
        V3i off;
        for (off.x = -inflate_n; off.x <= inflate_n; off.x++)
            for (off.y = -inflate_n; off.y <= inflate_n; off.y++)
                for (off.z = -inflate_n; off.z <= inflate_n; off.z++) {
                    if (std::abs(off.x) + std::abs(off.y) + std::abs(off.z) > inflate_n) continue;
                    V3i sid = off + subbox_id2xyz_table[subbox_id];
                    bool expanded = false;
                    V3i g = glb_idx;
                    for (int m = 0; m < 3; m++) {
                        if (sid[m] >= subbox_nxyz) {
                            g.at(m) += 1;
                            sid.at(m) = sid[m] - subbox_nxyz;
                            expanded = true;
                        } else if (sid[m] < 0) {
                            g.at(m) += -1;
                            sid.at(m) = subbox_nxyz + sid[m];
                            expanded = true;
                        }
                    }
                    if ((expanded && allocate_ram(g)) || !expanded)
                        observed_group_map[g].inflate_occupancy[cell_id(sid)] = 'o';
                }
    }
    /* map_local.cpp:143-237.  `logit` is the macro log10((x)/(1-(x))) on a float -> log10f. */
    void input_pc_pose_direct(Awareness *a_map) {
        SE3 T_wa = a_map->T_wa;
        for (auto pair_ : a_map->hit_idx_odds_hashmap) {
            V3i glb_idx;
            size_t subbox_id;
            V3 p_w = se3_apply(T_wa, a_map->center_of(pair_.first.x, pair_.first.y, pair_.first.z));
            get_global_idx(p_w, glb_idx, subbox_id);
            if (allocate_ram(glb_idx)) {
                if (observed_group_map[glb_idx].log_odds[subbox_id] < log_odds_max) {
                    observed_group_map[glb_idx].log_odds[subbox_id] += std::log10((pair_.second) / (1 - (pair_.second)));
                    observed_group_map[glb_idx].log_odds[subbox_id] =
                        observed_group_map[glb_idx].log_odds[subbox_id] > log_odds_max
                            ? log_odds_max
                            : observed_group_map[glb_idx].log_odds[subbox_id];
                }
                if (observed_group_map[glb_idx].log_odds[subbox_id] > log_odds_occupied_sh &&
                    observed_group_map[glb_idx].occupancy[subbox_id] != 'o') {
                    observed_group_map[glb_idx].occupancy[subbox_id] = 'o';
                    if (apply_explored_area) observed_group_map[glb_idx].frontier.erase((int)subbox_id);
                    obs_cnt++;
                }
            }
        }
        for (auto idx : a_map->miss_idx_set) {
            V3i glb_idx;
            size_t subbox_id;
            V3 p_w = se3_apply(T_wa, a_map->center_of_idx(idx));
            get_global_idx(p_w, glb_idx, subbox_id);
            if (allocate_ram(glb_idx)) {
                if (observed_group_map[glb_idx].log_odds[subbox_id] >= log_odds_min) {
                    observed_group_map[glb_idx].log_odds[subbox_id] += log_odds_miss;
                    observed_group_map[glb_idx].log_odds[subbox_id] =
                        observed_group_map[glb_idx].log_odds[subbox_id] < log_odds_min
                            ? log_odds_min
                            : observed_group_map[glb_idx].log_odds[subbox_id];
                }
                if (observed_group_map[glb_idx].log_odds[subbox_id] < log_odds_occupied_sh &&
                    observed_group_map[glb_idx].occupancy[subbox_id] != 'f') {
                    if (observed_group_map[glb_idx].occupancy[subbox_id] == 'u' && apply_explored_area)
                        update_observation(glb_idx, subbox_id, p_w);
                    observed_group_map[glb_idx].occupancy[subbox_id] = 'f';
                    if (apply_explored_area) observed_group_map[glb_idx].frontier.erase((int)subbox_id);
                }
            }
        }
        for (auto glb_idx : observed_subboxes) {
            if (observed_group_map.find(glb_idx) != observed_group_map.end() &&
                observed_group_map[glb_idx].occupancy.size() > 1 && observed_group_map[glb_idx].frontier.empty()) {
                if (std::adjacent_find(observed_group_map[glb_idx].occupancy.begin(),
                                       observed_group_map[glb_idx].occupancy.end(),
                                       std::not_equal_to<char>()) == observed_group_map[glb_idx].occupancy.end()) {
                    observed_group_map[glb_idx].occupancy.resize(1);
                    observed_group_map[glb_idx].occupancy.shrink_to_fit();
                    observed_group_map[glb_idx].inflate_occupancy.resize(1);
                    observed_group_map[glb_idx].inflate_occupancy.shrink_to_fit();
                    observed_group_map[glb_idx].log_odds.resize(1);
                    observed_group_map[glb_idx].log_odds.shrink_to_fit();
                }
            }
        }
        observed_subboxes.clear();
    }
};

} // namespace

/* ---- mlmap (mlmap.h, mlmap.cpp) ------------------------------------------------------------------ */
struct mlo_handle {
    mlo_config cfg;
    Awareness am;
    LocalMap lm;
    float cx_, cy_, fx_, fy_; /* mlmap.h:92 */
    size_t pc_sample_cnt;
    int inflate_global_n = 2;
    std::vector<V3> pc_eigen;
    /* getOddGrad member scratch, mlmap.h:96-97 */
    std::vector<V3i> glb_idx_nb_list = std::vector<V3i>(6);
    std::vector<size_t> subbox_id_nb_list = std::vector<size_t>(6);
    const double inv_factor = 1.0 / 1000.0; /* mlmap.h:85-86 */

    enum { FREE = 1, OCCUPIED = 0, UNKNOWN = -1 }; /* mlmap.h:109-114 */

    /* mlmap.cpp:344-346: (size_t u - float cx_) is a float subtraction, then double */
    V3 backproject(size_t u, size_t v, uint16_t raw) const {
        double depth = raw * inv_factor;
        V3 pt;
        pt.x = (u - cx_) * depth / fx_;
        pt.y = (v - cy_) * depth / fy_;
        pt.z = depth;
        return pt;
    }
    /* mlmap.h:170-193 */
    int getOccupancy(const V3 &pos_w) {
        V3i glb_id;
        size_t subbox_id;
        char res;
        lm.get_global_idx(pos_w, glb_id, subbox_id);
        auto it = lm.observed_group_map.find(glb_id);
        if (it == lm.observed_group_map.end())
            return UNKNOWN;
        else if (it->second.occupancy.size() == 1)
            res = it->second.occupancy[0];
        else
            res = it->second.occupancy[subbox_id];
        if (res == 'o')
            return OCCUPIED;
        else if (res == 'f')
            return FREE;
        else
            return UNKNOWN;
    }
    /* mlmap.h:142-169 — 19-point stencil */
    int getOccupancyInflate(const V3 &p, float inflate) {
        const double f = inflate;
        const V3 offs[19] = {{0, 0, 0},   {0, 0, f},  {0, 0, -f},  {0, f, 0},  {0, -f, 0}, {f, 0, 0},  {-f, 0, 0},
                             {-f, f, 0},  {-f, -f, 0}, {f, f, 0},  {f, -f, 0}, {0, -f, f}, {0, -f, -f}, {0, f, f},
                             {0, f, -f},  {-f, 0, f},  {-f, 0, -f}, {f, 0, f}, {f, 0, -f}};
        for (int i = 0; i < 19; i++)
            if (getOccupancy(p + offs[i]) == OCCUPIED) return OCCUPIED;
        return FREE;
    }
    /* mlmap.h:195-211 */
    int getInflateOccupancy(const V3 &pos_w) {
        V3i glb_id;
        size_t subbox_id;
        lm.get_global_idx(pos_w, glb_id, subbox_id);
        auto it = lm.observed_group_map.find(glb_id);
        if (it == lm.observed_group_map.end())
            return UNKNOWN;
        else if (it->second.occupancy.size() == 1)
            return UNKNOWN;
        else {
            if (it->second.inflate_occupancy[subbox_id] == 'o')
                return OCCUPIED;
            else
                return UNKNOWN;
        }
    }
    /* logit_inv macro, mlmap.h:40: pow(int, float) promotes to double pow */
    static float logit_inv(float x) { return (float)(std::pow(10, x) / (1 + std::pow(10, x))); }
    /* mlmap.h:227-235 */
    float getOdd(const V3i &glb_id, size_t subbox_id) {
        auto it = lm.observed_group_map.find(glb_id);
        if (it == lm.observed_group_map.end())
            return 0.5;
        else if (it->second.log_odds.size() == 1)
            return logit_inv(it->second.log_odds[0]);
        else
            return logit_inv(it->second.log_odds[subbox_id]);
    }
    /* mlmap.h:213-225 */
    float getOdd(const V3 &pos_w) {
        V3i glb_id;
        size_t subbox_id;
        lm.get_global_idx(pos_w, glb_id, subbox_id);
        return getOdd(glb_id, subbox_id);
    }
    /* mlmap.h:237-295 */
    V3 getOddGrad(const V3 &pos_w, size_t max_iter) {
        V3i glb_id;
        size_t subbox_id;
        lm.get_global_idx(pos_w, glb_id, subbox_id);
        V3i glb_idx_nb{0, 0, 0}, glb_idx_nb_min{0, 0, 0};
        size_t subbox_id_nb = 0, subbox_id_nb_min = 0;
        float min_odd = getOdd(glb_id, subbox_id);
        float ori_odd = min_odd;
        float tmp_odd;
        bool flag = false;
        size_t iter;
        for (iter = 0; iter < max_iter && !flag; iter++) {
            for (int i = 0; i < 6; i++) {
                if (iter == 0) {
                    const Nbr &nb = lm.subbox_neighbors[subbox_id];
                    glb_idx_nb = glb_id + V3i{nb.d[i][0], nb.d[i][1], nb.d[i][2]};
                    subbox_id_nb = (size_t)nb.d[i][3];
                } else {
                    const Nbr &nb = lm.subbox_neighbors[subbox_id_nb_list[i]];
                    glb_idx_nb = glb_idx_nb_list[i] + V3i{nb.d[i][0], nb.d[i][1], nb.d[i][2]};
                    subbox_id_nb = (size_t)nb.d[i][3];
                }
                glb_idx_nb_list[i] = glb_idx_nb;
                subbox_id_nb_list[i] = subbox_id_nb;
                tmp_odd = getOdd(glb_idx_nb, subbox_id_nb);
                if (tmp_odd < min_odd) {
                    min_odd = tmp_odd;
                    glb_idx_nb_min = glb_idx_nb;
                    subbox_id_nb_min = subbox_id_nb;
                    flag = true;
                }
            }
        }
        if (flag)
            return (lm.subbox_id2xyz_glb_vec(glb_idx_nb_min, (int)subbox_id_nb_min) - pos_w) * (ori_odd - min_odd);
        else
            return V3{0.0, 0.0, 0.0};
    }
    /* mlmap.cpp:388-407 */
    void setFree_map_in_bound(const V3 &box_min, const V3 &box_max) {
        V3i glb_id;
        size_t subbox_id;
        for (double x = box_min.x; x <= box_max.x; x += lm.map_dxyz_obv_sub)
            for (double y = box_min.y; y <= box_max.y; y += lm.map_dxyz_obv_sub)
                for (double z = box_min.z; z <= box_max.z; z += lm.map_dxyz_obv_sub) {
                    lm.get_global_idx(V3{x, y, z}, glb_id, subbox_id);
                    auto it = lm.observed_group_map.find(glb_id);
                    if (it != lm.observed_group_map.end() && it->second.occupancy.size() > 1) {
                        it->second.occupancy[subbox_id] = 'f';
                        it->second.log_odds[subbox_id] = 0;
                    }
                }
    }
    /* mlmap.cpp:286-309 */
    void inflate_map(const V3 &ct_pos) {
        V3i ct_glb;
        size_t subbox_id;
        lm.get_global_idx(ct_pos, ct_glb, subbox_id);
        V3i off;
        for (off.x = -inflate_global_n; off.x <= inflate_global_n; off.x++)
            for (off.y = -inflate_global_n; off.y <= inflate_global_n; off.y++)
                for (off.z = -inflate_global_n; off.z <= inflate_global_n; off.z++) {
                    V3i temp_glb = off + ct_glb;
                    if (lm.observed_group_map.find(temp_glb) != lm.observed_group_map.end() &&
                        lm.observed_group_map[temp_glb].occupancy.size() > 1) {
                        lm.observed_group_map[temp_glb].inflate_occupancy.clear();
                        lm.observed_group_map[temp_glb].inflate_occupancy.resize(lm.cell_num_subbox, 'u');
                        for (size_t it = 0; it < lm.observed_group_map[temp_glb].occupancy.size(); it++)
                            if (lm.observed_group_map[temp_glb].occupancy[it] == 'o' &&
                                lm.subbox_id2xyz_glb_vec(temp_glb, (int)it).z > lm.flate_height)
                                lm.inflate_atpos(temp_glb, it);
                    }
                }
    }
    /* mlmap.cpp:382-386 */
    void update_map(const SE3 &T_wb) {
        am.input_pc_pose(pc_eigen, T_wb);
        lm.input_pc_pose_direct(&am);
    }
};

/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const Quat &q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1 - (txx + tyy);
}
/* SO3::logAndTheta, so3.cpp:127-165 (SMALL_EPS = 1e-10, so3.h:35) */
static V3 so3_log(const Quat &q) {
    const double SMALL_EPS = 1e-10;
    double n = std::sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
    double w = q.w;
    double squared_w = w * w;
    double two_atan_nbyw_by_n;
    if (n < SMALL_EPS) {
        two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
    } else {
        if (std::fabs(w) < SMALL_EPS) {
            if (w > 0)
                two_atan_nbyw_by_n = M_PI / n;
            else
                two_atan_nbyw_by_n = -M_PI / n;
        }
        two_atan_nbyw_by_n = 2 * std::atan(n / w) / n; /* (overwrites the branch above, as in the reference) */
    }
    return V3{two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z};
}
/* SO3::expAndTheta, so3.cpp:175-197; SO3(Quaterniond) normalises */
static Quat so3_exp(const V3 &omega) {
    const double SMALL_EPS = 1e-10;
    double theta = std::sqrt((omega.x * omega.x + omega.y * omega.y) + omega.z * omega.z);
    double half_theta = 0.5 * theta;
    double imag_factor;
    double real_factor = std::cos(half_theta);
    if (theta < SMALL_EPS) {
        double theta_sq = theta * theta;
        double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
        double sin_half_theta = std::sin(half_theta);
        imag_factor = sin_half_theta / theta;
    }
    return quat_normalized(Quat{real_factor, imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z});
}

static SE3 make_T_wb(const double q_wb[4], const double t_wb[3]) {
    /* mlmap.cpp:494: SE3(SO3, Vec3); SO3::exp returns a unit quaternion — the harness supplies q_wb
     * directly and it is normalised once as SO3(Quaterniond) does (so3.cpp:43-47). */
    SE3 T;
    T.q = quat_normalized(Quat{q_wb[0], q_wb[1], q_wb[2], q_wb[3]});
    T.t = V3{t_wb[0], t_wb[1], t_wb[2]};
    return T;
}

extern "C" {

void mlo_set_quat_arch(int arch) { g_quat_arch = arch == 1 ? 1 : 0; }

mlo_handle *mlo_create(const mlo_config *c) {
    mlo_handle *h = new mlo_handle();
    h->cfg = *c;
    /* mlmap.cpp:14-18 */
    h->pc_sample_cnt = (size_t)c->sample_cnt;
    h->cx_ = (float)c->cam_cx;
    h->cy_ = (float)c->cam_cy;
    h->fx_ = (float)c->cam_fx;
    h->fy_ = (float)c->cam_fy;
    h->inflate_global_n = c->inflate_global_n;
    /* mlmap.cpp:22-25: SE3(Matrix3d, Vector3d) -> Quaterniond(R), no normalisation (so3.cpp:39-40) */
    double R[9] = {c->T_bs[0], c->T_bs[1], c->T_bs[2], c->T_bs[4], c->T_bs[5], c->T_bs[6], c->T_bs[8], c->T_bs[9], c->T_bs[10]};
    h->am.T_bs.q = quat_from_R(R);
    h->am.T_bs.t = V3{c->T_bs[3], c->T_bs[7], c->T_bs[11]};
    h->am.init_map(c->am_d_rho, c->am_d_phi_deg, c->am_d_z, c->am_n_rho, c->am_n_z_below, c->am_n_z_over,
                   c->use_raycasting != 0, c->depth_noise_coe);
    /* mlmap.cpp:75-85 */
    h->lm.init_map(c->subbox_d_xyz, (unsigned int)c->subbox_n, static_cast<float>(c->log_odds_min),
                   static_cast<float>(c->log_odds_max), static_cast<float>(c->measurement_hit),
                   static_cast<float>(c->measurement_miss), static_cast<float>(c->occupied_sh),
                   c->use_exploration_frontiers != 0);
    h->lm.inflate_n = c->inflate_n;
    h->lm.apply_inflate = c->apply_inflate != 0;
    return h;
}
void mlo_destroy(mlo_handle *h) { delete h; }

int mlo_update_points(mlo_handle *h, const double *xyz, int n, const double q_wb[4], const double t_wb[3]) {
    h->pc_eigen.clear();
    for (int i = 0; i < n; i++) h->pc_eigen.push_back(V3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    h->update_map(make_T_wb(q_wb, t_wb));
    return n;
}
int mlo_awareness_points(mlo_handle *h, const double *xyz, int n, const double q_wb[4], const double t_wb[3]) {
    h->pc_eigen.clear();
    for (int i = 0; i < n; i++) h->pc_eigen.push_back(V3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    h->am.input_pc_pose(h->pc_eigen, make_T_wb(q_wb, t_wb));
    return n;
}
void mlo_local_from_awareness(mlo_handle *h) { h->lm.input_pc_pose_direct(&h->am); }

int mlo_project_dense(mlo_handle *h, const uint16_t *img, int rows, int cols, double *out) {
    int n = 0;
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            uint16_t raw = img[(size_t)v * cols + u];
            if (raw == 0) continue;
            V3 p = h->backproject((size_t)u, (size_t)v, raw);
            out[3 * n] = p.x;
            out[3 * n + 1] = p.y;
            out[3 * n + 2] = p.z;
            n++;
        }
    return n;
}
int mlo_update_depth_dense(mlo_handle *h, const uint16_t *img, int rows, int cols, const double q_wb[4],
                           const double t_wb[3]) {
    h->pc_eigen.clear();
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            uint16_t raw = img[(size_t)v * cols + u];
            if (raw == 0) continue;
            h->pc_eigen.emplace_back(h->backproject((size_t)u, (size_t)v, raw));
        }
    h->update_map(make_T_wb(q_wb, t_wb));
    return (int)h->pc_eigen.size();
}
int mlo_update_depth_indexed(mlo_handle *h, const uint16_t *img, int rows, int cols, const int32_t *pix, int n_pix,
                             const double q_wb[4], const double t_wb[3]) {
    (void)rows;
    h->pc_eigen.clear();
    for (int i = 0; i < n_pix; i++) {
        uint16_t raw = img[pix[i]];
        if (raw == 0) continue;
        h->pc_eigen.emplace_back(h->backproject((size_t)(pix[i] % cols), (size_t)(pix[i] / cols), raw));
    }
    h->update_map(make_T_wb(q_wb, t_wb));
    return (int)h->pc_eigen.size();
}
/* mlmap::project_depth, mlmap.cpp:311-349 (pc_eigen is cleared by the callback first, mlmap.cpp:469) */
int mlo_update_depth_sampled(mlo_handle *h, const uint16_t *img, int rows, int cols, const double q_wb[4],
                             const double t_wb[3]) {
    h->pc_eigen.clear();
    size_t u, v;
    int cnt = 0;
    int max_iter = 2 * h->pc_sample_cnt;
    while (h->pc_eigen.size() < h->pc_sample_cnt && cnt < max_iter) {
        cnt++;
        v = static_cast<size_t>(rand() % rows);
        u = static_cast<size_t>(rand() % cols);
        uint16_t raw = img[v * cols + u];
        if (raw == 0) continue;
        h->pc_eigen.emplace_back(h->backproject(u, v, raw));
    }
    h->update_map(make_T_wb(q_wb, t_wb));
    return (int)h->pc_eigen.size();
}

/* cv::Mat::convertTo(CV_16UC1, 1000) on a 32FC1 image (mlmap.cpp:482).  OpenCV is not in the reference tree (a
 * system dependency, version unpinned: CMakeLists.txt find_package(OpenCV)); its rule for float -> ushort with a scale
 * is dst = saturate_cast<ushort>(cvRound(src*alpha)) with the product in float.  cvRound on x86-64 is cvtss2si: round
 * half to even, and INT_MIN for NaN / Inf / anything outside int32; saturate_cast<ushort>(int) clamps to [0, 65535].
 * So non-finite pixels (REP-117 "no return") become 0 and are skipped by project_depth (mlmap.cpp:338-341). */
static inline uint16_t cv_f32_to_u16(float v) {
    const float s = v * 1000.0f;
    int r;
    if (!(s == s) || !(s < 2147483648.0f) || s < -2147483648.0f)
        r = INT32_MIN;
    else
        r = (int)std::lrintf(s);
    return (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
}

int mlo_callback(mlo_handle *h, const void *depth, int is_f32, int rows, int cols, double t_img, const double odom_p[3],
                 const double odom_q[4], const double odom_v[3], double t_odom, const double imu_w[3], double t_imu,
                 double camera2odom_latency, int sampled, double *T_wb_out) {
    /* mlmap.cpp:470-473 */
    double ros_time_gap_odom = t_img - t_odom;
    double ros_time_gap_imu = t_img - t_imu;
    double time_gap = ros_time_gap_imu - camera2odom_latency;
    std::vector<uint16_t> img((size_t)rows * cols);
    if (is_f32) {
        const float *f = (const float *)depth;
        for (size_t i = 0; i < img.size(); i++) img[i] = cv_f32_to_u16(f[i]);
    } else {
        std::memcpy(img.data(), depth, img.size() * sizeof(uint16_t));
    }
    /* mlmap.cpp:485-498 */
    V3 ct_pos{odom_p[0], odom_p[1], odom_p[2]};
    Quat rot_og = quat_normalized(Quat{odom_q[0], odom_q[1], odom_q[2], odom_q[3]});
    double R[9];
    quat_to_R(rot_og, R);
    V3 rot_dot{(R[0] * imu_w[0] + R[1] * imu_w[1]) + R[2] * imu_w[2], (R[3] * imu_w[0] + R[4] * imu_w[1]) + R[5] * imu_w[2],
               (R[6] * imu_w[0] + R[7] * imu_w[1]) + R[8] * imu_w[2]};
    V3 lg = so3_log(rot_og);
    V3 rot_cp{lg.x + time_gap * rot_dot.x, lg.y + time_gap * rot_dot.y, lg.z + time_gap * rot_dot.z};
    SE3 T_wb;
    T_wb.q = so3_exp(rot_cp);
    const double dtv = ros_time_gap_odom - camera2odom_latency;
    T_wb.t = V3{ct_pos.x + dtv * odom_v[0], ct_pos.y + dtv * odom_v[1], ct_pos.z + dtv * odom_v[2]};
    if (T_wb_out) {
        T_wb_out[0] = T_wb.q.w;
        T_wb_out[1] = T_wb.q.x;
        T_wb_out[2] = T_wb.q.y;
        T_wb_out[3] = T_wb.q.z;
        T_wb_out[4] = T_wb.t.x;
        T_wb_out[5] = T_wb.t.y;
        T_wb_out[6] = T_wb.t.z;
    }
    /* project_depth + update_map, mlmap.cpp:504-507 (pc_eigen cleared at :469) */
    h->pc_eigen.clear();
    if (sampled) {
        size_t u, v;
        int cnt = 0;
        int max_iter = 2 * h->pc_sample_cnt;
        while (h->pc_eigen.size() < h->pc_sample_cnt && cnt < max_iter) {
            cnt++;
            v = static_cast<size_t>(rand() % rows);
            u = static_cast<size_t>(rand() % cols);
            uint16_t raw = img[v * cols + u];
            if (raw == 0) continue;
            h->pc_eigen.emplace_back(h->backproject(u, v, raw));
        }
    } else {
        for (int v = 0; v < rows; v++)
            for (int u = 0; u < cols; u++) {
                uint16_t raw = img[(size_t)v * cols + u];
                if (raw == 0) continue;
                h->pc_eigen.emplace_back(h->backproject((size_t)u, (size_t)v, raw));
            }
    }
    h->update_map(T_wb);
    return (int)h->pc_eigen.size();
}

size_t mlo_hit_count(mlo_handle *h) { return h->am.hit_idx_odds_hashmap.size(); }
size_t mlo_miss_count(mlo_handle *h) { return h->am.miss_idx_set.size(); }
size_t mlo_hit_bucket_count(mlo_handle *h) { return h->am.hit_idx_odds_hashmap.bucket_count(); }
size_t mlo_out_of_range_count(mlo_handle *h) { return h->am.out_of_range; }
void mlo_get_hits(mlo_handle *h, int32_t *rpz, float *odds) {
    size_t i = 0;
    for (auto &kv : h->am.hit_idx_odds_hashmap) {
        rpz[3 * i] = kv.first.x;
        rpz[3 * i + 1] = kv.first.y;
        rpz[3 * i + 2] = kv.first.z;
        odds[i] = kv.second;
        i++;
    }
}
void mlo_get_misses(mlo_handle *h, uint64_t *idx) {
    size_t i = 0;
    for (auto v : h->am.miss_idx_set) idx[i++] = (uint64_t)v;
}
void mlo_get_T_ls(mlo_handle *h, double q[4], double t[3]) {
    q[0] = h->am.T_ls.q.w;
    q[1] = h->am.T_ls.q.x;
    q[2] = h->am.T_ls.q.y;
    q[3] = h->am.T_ls.q.z;
    t[0] = h->am.T_ls.t.x;
    t[1] = h->am.T_ls.t.y;
    t[2] = h->am.T_ls.t.z;
}
void mlo_get_odds_table(mlo_handle *h, float *out) {
    for (int d = 0; d < 21; d++)
        for (int r = 0; r < h->am.map_nRho; r++) out[d * h->am.map_nRho + r] = h->am.get_odds_table[d][r];
}
int mlo_n_phi(mlo_handle *h) { return h->am.map_nPhi; }
int mlo_n_z(mlo_handle *h) { return h->am.map_nZ; }

size_t mlo_block_count(mlo_handle *h) { return h->lm.observed_group_map.size(); }
void mlo_export_blocks(mlo_handle *h, int32_t *keys, uint8_t *collapsed, float *log_odds, char *occ, char *infl,
                       int32_t *frontier_cnt) {
    size_t i = 0, C = h->lm.cell_num_subbox;
    for (auto &kv : h->lm.observed_group_map) {
        keys[3 * i] = kv.first.x;
        keys[3 * i + 1] = kv.first.y;
        keys[3 * i + 2] = kv.first.z;
        const Subbox &b = kv.second;
        collapsed[i] = b.occupancy.size() == 1;
        size_t n = b.occupancy.size();
        std::memset(log_odds + i * C, 0, C * sizeof(float));
        std::memset(occ + i * C, 0, C);
        std::memset(infl + i * C, 0, C);
        for (size_t k = 0; k < n; k++) {
            log_odds[i * C + k] = b.log_odds[k];
            occ[i * C + k] = b.occupancy[k];
            infl[i * C + k] = b.inflate_occupancy[k];
        }
        if (frontier_cnt) frontier_cnt[i] = (int32_t)b.frontier.size();
        i++;
    }
}
size_t mlo_frontier_total(mlo_handle *h) {
    size_t n = 0;
    for (auto &kv : h->lm.observed_group_map) n += kv.second.frontier.size();
    return n;
}
void mlo_export_frontier(mlo_handle *h, int32_t *out) {
    size_t i = 0;
    for (auto &kv : h->lm.observed_group_map)
        for (int c : kv.second.frontier) {
            out[4 * i] = kv.first.x;
            out[4 * i + 1] = kv.first.y;
            out[4 * i + 2] = kv.first.z;
            out[4 * i + 3] = c;
            i++;
        }
}

void mlo_get_occupancy(mlo_handle *h, const double *pos, int n, int32_t *out) {
    for (int i = 0; i < n; i++) out[i] = h->getOccupancy(V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
}
void mlo_get_occupancy_inflate(mlo_handle *h, const double *pos, int n, float inflate, int32_t *out) {
    for (int i = 0; i < n; i++)
        out[i] = h->getOccupancyInflate(V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, inflate);
}
void mlo_get_inflate_occupancy(mlo_handle *h, const double *pos, int n, int32_t *out) {
    for (int i = 0; i < n; i++) out[i] = h->getInflateOccupancy(V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
}
void mlo_get_odd(mlo_handle *h, const double *pos, int n, float *out) {
    for (int i = 0; i < n; i++) out[i] = h->getOdd(V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
}
void mlo_get_odd_grad(mlo_handle *h, const double *pos, int n, int max_iter, double *out3) {
    for (int i = 0; i < n; i++) {
        V3 g = h->getOddGrad(V3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, (size_t)max_iter);
        out3[3 * i] = g.x;
        out3[3 * i + 1] = g.y;
        out3[3 * i + 2] = g.z;
    }
}
void mlo_set_free_in_bound(mlo_handle *h, const double bmin[3], const double bmax[3]) {
    h->setFree_map_in_bound(V3{bmin[0], bmin[1], bmin[2]}, V3{bmax[0], bmax[1], bmax[2]});
}
void mlo_inflate_map(mlo_handle *h, const double ct_pos[3]) { h->inflate_map(V3{ct_pos[0], ct_pos[1], ct_pos[2]}); }
/* rviz_vis.cpp:296-327 + map_local.h:201-206: PointP(float) of inflated 'o' cells, block iteration order */
size_t mlo_global_map_points(mlo_handle *h, float *xyz) {
    size_t n = 0;
    for (auto &kv : h->lm.observed_group_map) {
        int subbox_id = 0;
        for (char c : kv.second.inflate_occupancy) {
            if (c == 'o') {
                if (xyz) {
                    V3 p = h->lm.subbox_id2xyz_glb_vec(kv.first, subbox_id);
                    xyz[3 * n] = (float)p.x;
                    xyz[3 * n + 1] = (float)p.y;
                    xyz[3 * n + 2] = (float)p.z;
                }
                n++;
            }
            subbox_id++;
        }
    }
    return n;
}

/* rviz_vis.cpp:267-293 + map_local.h:201-206: PointP(float) of every frontier cell, block then set iteration order */
size_t mlo_frontier_points(mlo_handle *h, float *xyz) {
    size_t n = 0;
    for (auto &kv : h->lm.observed_group_map)
        for (int c : kv.second.frontier) {
            if (xyz) {
                V3 p = h->lm.subbox_id2xyz_glb_vec(kv.first, c);
                xyz[3 * n] = (float)p.x;
                xyz[3 * n + 1] = (float)p.y;
                xyz[3 * n + 2] = (float)p.z;
            }
            n++;
        }
    return n;
}
/* float mlmap::getOdd(const Vec3I &glb_id, size_t subbox_id), mlmap.h:227-235 */
void mlo_get_odd_at(mlo_handle *h, const int32_t *glb_id, const int32_t *subbox_id, int n, float *out) {
    for (int i = 0; i < n; i++)
        out[i] = h->getOdd(V3i{glb_id[3 * i], glb_id[3 * i + 1], glb_id[3 * i + 2]}, (size_t)subbox_id[i]);
}

/* cv::Mat::convertTo(CV_16UC1, 1000) per pixel (mlmap.cpp:482), see cv_f32_to_u16 */
void mlo_cv_f32_to_u16(const float *in, int n, uint16_t *out) {
    for (int i = 0; i < n; i++) out[i] = cv_f32_to_u16(in[i]);
}

/* ---- the SO3 / SE3 restatements on their own, for the property tests the reference holds for Sophus
 *      (3rdPartLib/Sophus/sophus/test_so3.cpp:14-110, test_se3.cpp:10-86).  q = (w,x,y,z). ---- */
void mlo_so3_from_quat(const double q[4], double out[4]) { /* SO3(Quaterniond): normalises, so3.cpp:43-47 */
    Quat r = quat_normalized(Quat{q[0], q[1], q[2], q[3]});
    out[0] = r.w, out[1] = r.x, out[2] = r.y, out[3] = r.z;
}
void mlo_so3_exp(const double omega[3], double out[4]) {
    Quat r = so3_exp(V3{omega[0], omega[1], omega[2]});
    out[0] = r.w, out[1] = r.x, out[2] = r.y, out[3] = r.z;
}
void mlo_so3_log(const double q[4], double out[3]) {
    V3 r = so3_log(Quat{q[0], q[1], q[2], q[3]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void mlo_so3_mul(const double a[4], const double b[4], double out[4]) { /* so3.cpp:73-78 */
    Quat r = quat_normalized(quat_mul(Quat{a[0], a[1], a[2], a[3]}, Quat{b[0], b[1], b[2], b[3]}));
    out[0] = r.w, out[1] = r.x, out[2] = r.y, out[3] = r.z;
}
void mlo_so3_matrix(const double q[4], double R[9]) { quat_to_R(Quat{q[0], q[1], q[2], q[3]}, R); }
/* T = (q, t) as 7 doubles */
void mlo_se3_mul(const double a[7], const double b[7], double out[7]) {
    SE3 A, B;
    A.q = Quat{a[0], a[1], a[2], a[3]}, A.t = V3{a[4], a[5], a[6]};
    B.q = Quat{b[0], b[1], b[2], b[3]}, B.t = V3{b[4], b[5], b[6]};
    SE3 r = se3_mul(A, B);
    out[0] = r.q.w, out[1] = r.q.x, out[2] = r.q.y, out[3] = r.q.z, out[4] = r.t.x, out[5] = r.t.y, out[6] = r.t.z;
}
void mlo_se3_inverse(const double a[7], double out[7]) {
    SE3 A;
    A.q = Quat{a[0], a[1], a[2], a[3]}, A.t = V3{a[4], a[5], a[6]};
    SE3 r = se3_inv(A);
    out[0] = r.q.w, out[1] = r.q.x, out[2] = r.q.y, out[3] = r.q.z, out[4] = r.t.x, out[5] = r.t.y, out[6] = r.t.z;
}
void mlo_se3_apply(const double a[7], const double p[3], double out[3]) {
    SE3 A;
    A.q = Quat{a[0], a[1], a[2], a[3]}, A.t = V3{a[4], a[5], a[6]};
    V3 r = se3_apply(A, V3{p[0], p[1], p[2]});
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

} /* extern "C" */
