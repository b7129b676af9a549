// mlm_host.h — the pure host arithmetic of libmlmap_hip.so: no HIP, no device memory, no handle.
//
// Everything here must reproduce the reference's FP64/float operation order bit for bit (the results feed the device
// kernels as constants): the Eigen::Quaterniond / Sophus SO3+SE3 pieces of the frame setup (so3.cpp:36-96,127-197,
// se3.cpp:29-95), the depth-noise odds table (map_awareness.cpp:36-46,119-132, map_awareness.h:120-146), the pose
// latency compensation of the depth callback (mlmap.cpp:485-498) and the replay of libstdc++'s rehash policy.
// Kept separate so that tests/test_host_math.py can build it with g++ -fsanitize=address,undefined on the CPU and check
// it against the oracle and against the property tests the reference holds for Sophus (test_so3.cpp, test_se3.cpp).
// Build with -ffp-contract=off (the reference is an SSE2 build without FMA, CMakeLists.txt:4).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <unordered_map>
#include <utility>
#include <vector>

#ifdef __HIPCC__
#define MLM_HD __host__ __device__
#else
#define MLM_HD
#endif

// cv::Mat::convertTo(CV_16UC1, 1000) of one 32FC1 pixel (mlmap.cpp:482): float product, cvRound (x86 cvtss2si: round
// half to even; NaN and anything that does not fit an int32 give INT_MIN), saturate_cast<ushort>(int).  So NaN, +-Inf
// (the REP-117 "no return" encoding) and |s| >= 2^31 become 0 — a pixel project_depth skips (mlmap.cpp:338-341) — while
// finite depths beyond 65.535 m saturate to 65535.
MLM_HD inline int mlm_cv_f32_to_u16(float v) {
    const float s = v * 1000.0f;
    if (!(s < 2147483648.0f)) return 0; // NaN, +Inf, >= 2^31: INT_MIN -> 0
    if (!(s > 0.0f)) return 0;          // negative (incl. -Inf) and zero
    const float r = rintf(s);
    return r > 65535.0f ? 65535 : (int)r;
}

// log10f of glibc 2.35 (Ubuntu 22.04: sysdeps/ieee754/flt-32/e_log10f.c on top of the table-driven logf of
// sysdeps/ieee754/flt-32/e_logf.c + logf_data.c), restated operation by operation: the hit increment of the reference is
// `logit(odd)` = log10f(odd / (1 - odd)) evaluated by the HOST's libm (map_local.h:8, map_local.cpp:159), and the log-odds
// it accumulates decide occupancy classes at a threshold, so the device must produce the same float bits.  Every step is
// an IEEE-754 double or float operation (no libm call), so the device result equals the host's bit for bit.  Checked
// against this image's libm over ALL positive finite floats (2^31 - 2^23 inputs: zero mismatches, with and without
// FMA contraction of the polynomial — the FMA ifunc variant of glibc's logf gives the same floats);
// tests/test_host_math.py repeats a strided sweep, and mlm_create compares the host's log10f with this function on the
// configuration's odds table and a sweep of the logit range before it lets the kernels use it (MlmDev::logit_exact).
// Precondition of mlm_glibc_logf_core: 0.5 <= x < 2 (what log10f hands it).
MLM_HD inline float mlm_glibc_logf_core(float x) {
    static const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const unsigned int ix = __builtin_bit_cast(unsigned int, x);
    if (ix == 0x3f800000u) return 0.0f;
    const unsigned int tmp = ix - 0x3f330000u;  // OFF
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int)tmp >> 23;               // arithmetic shift
    const unsigned int iz = ix - (tmp & (0x1ffu << 23));
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)__builtin_bit_cast(float, iz);
    const double r = z * invc - 1;              // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
MLM_HD inline float mlm_glibc_log10f(float x) {
    const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
    int hx = __builtin_bit_cast(int, x);
    int k = 0;
    if (hx < 0x00800000) { // x < 2^-126
        if ((hx & 0x7fffffff) == 0) return -two25 / __builtin_fabsf(x); // log(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);                           // log(-#) = NaN
        k -= 25;
        x *= two25; // subnormal: scale up
        hx = __builtin_bit_cast(int, x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int i = (int)(((unsigned int)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    const float xm = __builtin_bit_cast(float, hx);
    const float z = y * log10_2lo + ivln10 * mlm_glibc_logf_core(xm);
    return z + y * log10_2hi;
}

namespace mlm_host {

// Does this host's libm log10f (what the reference's logit macro calls) agree with mlm_glibc_log10f?  Checked on the values the
// caller cares about (`vals`) and on a strided sweep of the positive floats between 1e-4 and 1e4 (the logit's argument for odds in
// [0.001, 0.999] and the noisy-OR results above them).
inline bool host_log10f_matches(const float *vals, size_t n) {
    auto same = [](float v) {
        const float a = ::log10f(v), b = mlm_glibc_log10f(v);
        return __builtin_bit_cast(unsigned int, a) == __builtin_bit_cast(unsigned int, b);
    };
    for (size_t i = 0; i < n; ++i)
        if (!same(vals[i])) return false;
    const unsigned int lo = __builtin_bit_cast(unsigned int, 1e-4f), hi = __builtin_bit_cast(unsigned int, 1e4f);
    for (unsigned int u = lo; u < hi; u += 4099u)
        if (!same(__builtin_bit_cast(float, u))) return false;
    const float sp[] = {0.0f, 1.0f, __builtin_inff(), 1e-45f, 1e-39f, 3.4e38f};
    for (float v : sp)
        if (!same(v)) return false;
    return true;
}

// ---- Eigen::Quaterniond / Sophus::SE3 pieces of the frame setup (so3.cpp:36-96, se3.cpp:29-95) -----------------
struct Q4 {
    double w, x, y, z;
};
struct D3 {
    double x, y, z;
};
inline Q4 q_mul(const Q4 &a, const Q4 &b) { // Eigen generic quat_product
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Q4 q_norm(const Q4 &q) { // normalize(): coeffs / sqrt(x²+y²+z²+w²)
    const double n = std::sqrt(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
inline D3 q_rot(const Q4 &q, const D3 &v) { // _transformVector
    D3 uv{q.y * v.z - q.z * v.y, q.z * v.x - q.x * v.z, q.x * v.y - q.y * v.x};
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    const D3 c{q.y * uv.z - q.z * uv.y, q.z * uv.x - q.x * uv.z, q.x * uv.y - q.y * uv.x};
    return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
inline Q4 q_from_R(const double m[9]) { // Eigen Quaternion(Matrix3): Shepperd, no normalisation (so3.cpp:39-40)
    auto M = [&](int r, int c) { return m[r * 3 + c]; };
    Q4 q;
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
        const int j = (i + 1) % 3, k = (j + 1) % 3;
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

// ---- odds table (map_awareness.cpp:36-46,119-132; map_awareness.h:120-146) -------------------------------------
struct OddsModel {
    double dRho, noise;
    float sigma_in_dr(size_t x) const {
        float dis = (x * dRho);
        return noise * dis * dis / dRho;
    }
    static float standard_ND(float x) { // A&S 7.1.26; fabs/exp resolve to the float overloads
        const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
        const double p = 0.3275911;
        int sign = 1;
        if (x < 0) sign = -1;
        x = std::fabs(x) / std::sqrt(2.0);
        const double t = 1.0 / (1.0 + p * x);
        const double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
        return 0.5 * (1.0 + sign * y);
    }
    float get_odds(int diff, size_t r) const {
        if (r == 0) r = 1;
        const float up = standard_ND(static_cast<float>(diff + 0.5) / sigma_in_dr(r));
        const float down = standard_ND(static_cast<float>(diff - 0.5) / sigma_in_dr(r));
        float res = up - down < 0.001 ? 0.001 : up - down;
        res = res >= 0.999 ? 0.999 : res;
        return res;
    }
};

// Eigen Quaternion::toRotationMatrix (row major), as rot_og.matrix() in mlmap.cpp:492
inline void q_to_R(const Q4 &q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
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
// SO3::logAndTheta, so3.cpp:134-175 (this Sophus version overwrites its |w| < eps branch: there is no `else`)
inline D3 so3_log(const Q4 &q, double *theta_out = nullptr) {
    const double EPS = 1e-10; // SMALL_EPS, so3.h:35
    const double n = std::sqrt((q.x * q.x + q.y * q.y) + q.z * q.z), w = q.w;
    double f;
    if (n < EPS)
        f = 2. / w - 2. * (n * n) / (w * (w * w));
    else
        f = 2 * std::atan(n / w) / n;
    if (theta_out) *theta_out = f * n;
    return D3{f * q.x, f * q.y, f * q.z};
}
// SO3::expAndTheta, so3.cpp:177-202 (SO3(Quaterniond) normalises)
inline Q4 so3_exp(const D3 &omega) {
    const double EPS = 1e-10;
    const double theta = std::sqrt((omega.x * omega.x + omega.y * omega.y) + omega.z * omega.z);
    const double half = 0.5 * theta, re = std::cos(half);
    double im;
    if (theta < EPS) {
        const double t2 = theta * theta, t4 = t2 * t2;
        im = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
    } else {
        im = std::sin(half) / theta;
    }
    return q_norm(Q4{re, im * omega.x, im * omega.y, im * omega.z});
}
// SE3 = (unit quaternion, translation): se3.cpp:60-95
struct T7 {
    Q4 q;
    D3 t;
};
inline T7 se3_mul(const T7 &a, const T7 &b) { // se3.cpp:60-66
    const D3 r = q_rot(a.q, b.t);
    return T7{q_norm(q_mul(a.q, b.q)), D3{a.t.x + r.x, a.t.y + r.y, a.t.z + r.z}};
}
inline T7 se3_inverse(const T7 &a) { // se3.cpp:76-83
    const Q4 qi = q_norm(Q4{a.q.w, -a.q.x, -a.q.y, -a.q.z});
    return T7{qi, q_rot(qi, D3{a.t.x * -1., a.t.y * -1., a.t.z * -1.})};
}
inline D3 se3_apply(const T7 &a, const D3 &p) { // se3.cpp:91-95
    const D3 r = q_rot(a.q, p);
    return D3{r.x + a.t.x, r.y + a.t.y, r.z + a.t.z};
}

// Pose latency compensation of depth_odom_input_callback, mlmap.cpp:470-498: T_wb forwarded to the image stamp by the
// linear model (rotation in the Lie algebra).  Stamps in seconds; out = q (w,x,y,z) then t.
inline void compensate_pose(const double odom_p[3], const double odom_q[4], const double odom_v[3], const double imu_w[3],
                            double t_img, double t_odom, double t_imu, double latency, double q_out[4], double t_out[3]) {
    const double gap_odom = t_img - t_odom, gap_imu = t_img - t_imu;
    const double time_gap = gap_imu - latency;
    const Q4 q = q_norm(Q4{odom_q[0], odom_q[1], odom_q[2], odom_q[3]});
    double R[9];
    q_to_R(q, R);
    const D3 rot_dot{(R[0] * imu_w[0] + R[1] * imu_w[1]) + R[2] * imu_w[2], (R[3] * imu_w[0] + R[4] * imu_w[1]) + R[5] * imu_w[2],
                     (R[6] * imu_w[0] + R[7] * imu_w[1]) + R[8] * imu_w[2]};
    const D3 lg = so3_log(q);
    const D3 rot_cp{lg.x + time_gap * rot_dot.x, lg.y + time_gap * rot_dot.y, lg.z + time_gap * rot_dot.z};
    const Q4 q_wb = so3_exp(rot_cp);
    const double dtv = gap_odom - latency;
    q_out[0] = q_wb.w;
    q_out[1] = q_wb.x;
    q_out[2] = q_wb.y;
    q_out[3] = q_wb.z;
    for (int i = 0; i < 3; ++i) t_out[i] = odom_p[i] + dtv * odom_v[i];
}

// T_ls and t_wa of one frame (map_awareness.cpp:184-186) — SURVEY.md App. C1, evaluated in that order
inline void frame_pose(const Q4 &q_bs, const D3 &t_bs, const double q_wb_in[4], const double t_wb_in[3], double q_ls_out[4],
                       double t_ls_out[3], double t_wa_out[3]) {
    const Q4 q_wb = q_norm(Q4{q_wb_in[0], q_wb_in[1], q_wb_in[2], q_wb_in[3]}); // SO3(Quaterniond), so3.cpp:43-47
    const D3 t_wb{t_wb_in[0], t_wb_in[1], t_wb_in[2]};
    // T_wa = (I, t_wb)
    const Q4 q_wa = q_norm(Q4{1, 0, 0, 0});
    // T_ws = T_wb * T_bs
    const D3 r1 = q_rot(q_wb, t_bs);
    const D3 t_ws{t_wb.x + r1.x, t_wb.y + r1.y, t_wb.z + r1.z};
    const Q4 q_ws = q_norm(q_mul(q_wb, q_bs));
    // T_wa^-1
    const Q4 q_ai = q_norm(Q4{q_wa.w, -q_wa.x, -q_wa.y, -q_wa.z});
    const D3 t_ai = q_rot(q_ai, D3{t_wb.x * -1., t_wb.y * -1., t_wb.z * -1.});
    // T_ls = T_wa^-1 * T_ws
    const D3 r2 = q_rot(q_ai, t_ws);
    const Q4 q_ls = q_norm(q_mul(q_ai, q_ws));
    q_ls_out[0] = q_ls.w;
    q_ls_out[1] = q_ls.x;
    q_ls_out[2] = q_ls.y;
    q_ls_out[3] = q_ls.z;
    t_ls_out[0] = t_ai.x + r2.x;
    t_ls_out[1] = t_ai.y + r2.y;
    t_ls_out[2] = t_ai.z + r2.z;
    t_wa_out[0] = t_wb.x;
    t_wa_out[1] = t_wb.y;
    t_wa_out[2] = t_wb.z;
}

// exact floor(i / d) for i < 2^27 as (i * m) >> s (Granlund-Montgomery: m = ceil(2^(27+L) / d), L = ceil(log2 d))
inline void div_magic(unsigned int d, unsigned long long &m, int &s) {
    int L = 0;
    while ((1ull << L) < d) ++L;
    s = 27 + L;
    m = ((1ull << s) + d - 1) / d;
}

// Replay the rehash policy of libstdc++'s _Hashtable for `U` unique insertions into a cleared container.
// Returns the epochs: (number of elements present when the epoch ends, bucket count during the epoch).
// Uses the very policy object std::unordered_map uses, so it follows whatever libstdc++ this library is linked to.
inline std::vector<std::pair<size_t, size_t>> plan_epochs_for(std::__detail::_Prime_rehash_policy &pol, size_t &n_bkt, size_t U) {
    std::vector<std::pair<size_t, size_t>> ep;
    size_t n = n_bkt;
    size_t i = 0;
    while (i < U) {
        // _M_insert_unique_node: _M_need_rehash(bucket_count, element_count, 1) before linking the node
        const auto r = pol._M_need_rehash(n, i, 1);
        if (r.first) {
            if (i > 0) ep.emplace_back(i, n);
            n = r.second;
        }
        // the policy is inert while element_count + 1 <= _M_next_resize
        const size_t next = std::max<size_t>(i + 1, pol._M_next_resize);
        i = std::min(U, next);
    }
    ep.emplace_back(U, n);
    n_bkt = n;
    return ep;
}

} // namespace mlm_host
