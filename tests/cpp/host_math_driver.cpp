// Test driver for mlmapping_amd/csrc/mlm_host.h (the pure host arithmetic of libmlmap_hip.so), built by
// tests/test_host_math.py with g++ -fsanitize=address,undefined.  Prints every result as hex floats, one record per
// line ("tag index v0 v1 ..."), so that the test can compare them bit for bit with the oracle on the same inputs and
// check the properties the reference's Sophus tests state (3rdPartLib/Sophus/sophus/test_so3.cpp, test_se3.cpp).
// The rehash-policy replay is checked here against a real std::unordered_map (exit code 1 on a mismatch).
#include <cstdint>
#include <cstdio>
#include <unordered_set>

#include "mlm_host.h"

using namespace mlm_host;

static void rec(const char *tag, int i, const double *v, int n) {
    std::printf("%s %d", tag, i);
    for (int k = 0; k < n; ++k) std::printf(" %a", v[k]);
    std::printf("\n");
}
static void rec_q(const char *tag, int i, const Q4 &q) {
    const double v[4] = {q.w, q.x, q.y, q.z};
    rec(tag, i, v, 4);
}
static void rec_t(const char *tag, int i, const T7 &T) {
    const double v[7] = {T.q.w, T.q.x, T.q.y, T.q.z, T.t.x, T.t.y, T.t.z};
    rec(tag, i, v, 7);
}
static Q4 so3_mul(const Q4 &a, const Q4 &b) { return q_norm(q_mul(a, b)); } // so3.cpp:73-78

int main() {
    const double PI = 3.14159265358979323846;
    // ---- the SO3 cases of test_so3.cpp:17-30
    std::vector<Q4> so3s;
    so3s.push_back(q_norm(Q4{0.1e-11, 0., 1., 0.}));
    so3s.push_back(q_norm(Q4{-1, 0.00001, 0.0, 0.0}));
    so3s.push_back(so3_exp(D3{0.2, 0.5, 0.0}));
    so3s.push_back(so3_exp(D3{0.2, 0.5, -1.0}));
    so3s.push_back(so3_exp(D3{0., 0., 0.}));
    so3s.push_back(so3_exp(D3{0., 0., 0.00001}));
    so3s.push_back(so3_exp(D3{PI, 0, 0}));
    so3s.push_back(so3_mul(so3_mul(so3_exp(D3{0.2, 0.5, 0.0}), so3_exp(D3{PI, 0, 0})), so3_exp(D3{-0.2, -0.5, -0.0})));
    so3s.push_back(so3_mul(so3_mul(so3_exp(D3{0.3, 0.5, 0.1}), so3_exp(D3{PI, 0, 0})), so3_exp(D3{-0.3, -0.5, -0.1})));
    for (size_t i = 0; i < so3s.size(); ++i) {
        const Q4 &q = so3s[i];
        rec_q("so3", (int)i, q);
        double theta = 0;
        const D3 lg = so3_log(q, &theta);
        const double l[4] = {lg.x, lg.y, lg.z, theta};
        rec("so3_log", (int)i, l, 4);
        rec_q("so3_explog", (int)i, so3_exp(lg));
        double R[9];
        q_to_R(q, R);
        rec("so3_R", (int)i, R, 9);
        const D3 p = q_rot(q, D3{1, 2, 4});
        const double pv[3] = {p.x, p.y, p.z};
        rec("so3_p", (int)i, pv, 3);
        rec_q("so3_inv", (int)i, q_norm(Q4{q.w, -q.x, -q.y, -q.z}));
    }
    // ---- the SE3 cases of test_se3.cpp:13-27 (log/exp of SE3 is not on the path: transform, product, inverse)
    const double pi_f = 3.14159265;
    std::vector<T7> se3s;
    se3s.push_back(T7{so3_exp(D3{0.2, 0.5, 0.0}), D3{0, 0, 0}});
    se3s.push_back(T7{so3_exp(D3{0.2, 0.5, -1.0}), D3{10, 0, 0}});
    se3s.push_back(T7{so3_exp(D3{0., 0., 0.}), D3{0, 100, 5}});
    se3s.push_back(T7{so3_exp(D3{0., 0., 0.00001}), D3{0, 0, 0}});
    se3s.push_back(T7{so3_exp(D3{0., 0., 0.00001}), D3{0, -0.00000001, 0.0000000001}});
    se3s.push_back(T7{so3_exp(D3{0., 0., 0.00001}), D3{0.01, 0, 0}});
    se3s.push_back(T7{so3_exp(D3{pi_f, 0, 0}), D3{4, -5, 0}});
    se3s.push_back(se3_mul(se3_mul(T7{so3_exp(D3{0.2, 0.5, 0.0}), D3{0, 0, 0}}, T7{so3_exp(D3{pi_f, 0, 0}), D3{0, 0, 0}}),
                           T7{so3_exp(D3{-0.2, -0.5, -0.0}), D3{0, 0, 0}}));
    se3s.push_back(se3_mul(se3_mul(T7{so3_exp(D3{0.3, 0.5, 0.1}), D3{2, 0, -7}}, T7{so3_exp(D3{pi_f, 0, 0}), D3{0, 0, 0}}),
                           T7{so3_exp(D3{-0.3, -0.5, -0.1}), D3{0, 6, 0}}));
    for (size_t i = 0; i < se3s.size(); ++i) {
        rec_t("se3", (int)i, se3s[i]);
        rec_t("se3_inv", (int)i, se3_inverse(se3s[i]));
        rec_t("se3_mul_inv", (int)i, se3_mul(se3s[i], se3_inverse(se3s[i])));
        const D3 p = se3_apply(se3s[i], D3{1, 2, 4});
        const double pv[3] = {p.x, p.y, p.z};
        rec("se3_p", (int)i, pv, 3);
    }
    // ---- frame setup (map_awareness.cpp:184-186) with T_B_S of config_sim.yaml:51-55, poses from a fixed LCG
    const double Rbs[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
    const Q4 q_bs = q_from_R(Rbs);
    rec_q("q_bs", 0, q_bs);
    const D3 t_bs{0.12, 0, 0};
    uint64_t st = 12345;
    auto rnd = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    };
    for (int i = 0; i < 64; ++i) {
        const double q_in[4] = {rnd(), rnd(), rnd(), rnd()}, t_in[3] = {4 * rnd(), 4 * rnd(), 1.5 + rnd()};
        double out[10];
        frame_pose(q_bs, t_bs, q_in, t_in, out, out + 4, out + 7);
        double in[7] = {q_in[0], q_in[1], q_in[2], q_in[3], t_in[0], t_in[1], t_in[2]};
        rec("pose_in", i, in, 7);
        rec("pose_out", i, out, 10);
        const double v[3] = {0.3 * rnd(), 0.3 * rnd(), 0.1 * rnd()}, w[3] = {rnd(), rnd(), rnd()};
        double cq[4], ct[3];
        compensate_pose(t_in, q_in, v, w, 10.0 + i / 30.0, 10.0 + i / 30.0 - 0.004, 10.0 + i / 30.0 - 0.002, 0.085, cq, ct);
        double cin[6] = {v[0], v[1], v[2], w[0], w[1], w[2]};
        rec("comp_in", i, cin, 6);
        double cout[7] = {cq[0], cq[1], cq[2], cq[3], ct[0], ct[1], ct[2]};
        rec("comp_out", i, cout, 7);
    }
    // ---- odds table of S1 (dRho 0.1, noise 0.00375, nRho 65): map_awareness.cpp:36-46
    {
        OddsModel om{0.1, 0.00375};
        for (int d = -10; d <= 10; ++d) {
            std::vector<double> row;
            for (int r = 0; r < 65; ++r) row.push_back((double)om.get_odds(d, (size_t)r));
            rec("odds", d + 10, row.data(), (int)row.size());
        }
    }
    // ---- 32FC1 -> 16UC1 (mlmap.cpp:482)
    {
        const float in[] = {0.0f, 1.0f, 1.0005f, 1.0015f, 65.5354f, 65.536f, 70.0f, -1.0f, 2147483.5f, 2147483.75f, 3.0e6f,
                            __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 0.0004f, 0.0005f, 0.00051f};
        std::vector<double> out;
        for (float v : in) out.push_back((double)mlm_cv_f32_to_u16(v));
        rec("cvt", 0, out.data(), (int)out.size());
    }
    // ---- mlm_glibc_log10f against this host's libm log10f (what the reference's logit macro calls, map_local.h:8): every
    //      317th positive finite float (6.7 M inputs; the full 2^31 sweep was run once, zero mismatches), every float of the
    //      logit's usual argument range [0.5, 2), subnormals, zero, infinity
    {
        unsigned long long n = 0, bad = 0;
        auto chk = [&](unsigned int u) {
            const float v = __builtin_bit_cast(float, u), a = ::log10f(v), b = mlm_glibc_log10f(v);
            ++n;
            if (__builtin_bit_cast(unsigned int, a) != __builtin_bit_cast(unsigned int, b)) ++bad;
        };
        for (unsigned int u = 0; u <= 0x7f800000u; u += 317u) chk(u);
        for (unsigned int u = 0x3f000000u; u < 0x40000000u; u += 7u) chk(u);
        for (unsigned int u = 0; u < 4096u; ++u) chk(u);
        chk(0x7f800000u);
        const double v[2] = {(double)n, (double)bad};
        rec("log10f", 0, v, 2);
        if (bad) {
            std::fprintf(stderr, "mlm_glibc_log10f differs from libm log10f on %llu of %llu inputs\n", bad, n);
            return 1;
        }
        if (!host_log10f_matches(nullptr, 0)) return 1;
    }
    // ---- exact division by multiplication (k_sort_contribs): i / d == (i * m) >> s for i < 2^27
    for (unsigned int d : {1u, 3u, 64u, 333u, 640u, 641u, 1280u, 4096u, 99991u}) {
        unsigned long long m;
        int s;
        div_magic(d, m, s);
        for (unsigned long long i = 0; i < (1ull << 27); i += (i < 70000 ? 1 : 9973))
            if ((unsigned int)((i * m) >> s) != (unsigned int)(i / d)) {
                std::fprintf(stderr, "div_magic(%u) wrong at %llu\n", d, i);
                return 1;
            }
        if ((unsigned int)((((1ull << 27) - 1) * m) >> s) != (unsigned int)(((1ull << 27) - 1) / d)) return 1;
    }
    // ---- rehash-policy replay against a real std::unordered_set that is cleared between frames (clear() keeps the
    //      bucket array, map_awareness.cpp:178)
    {
        std::unordered_set<size_t> real;
        std::__detail::_Prime_rehash_policy pol;
        size_t n_bkt = 1;
        const size_t frames[] = {14154, 17000, 9000, 50, 100000, 100001, 136766, 3, 0, 250000};
        for (size_t U : frames) {
            real.clear();
            // epoch boundaries must be where the real container changes its bucket count
            std::vector<std::pair<size_t, size_t>> real_ep;
            size_t cur = real.bucket_count();
            for (size_t k = 0; k < U; ++k) {
                real.insert(k * 2654435761ull + 17);
                if (real.bucket_count() != cur) {
                    if (k > 0) real_ep.emplace_back(k, cur);
                    cur = real.bucket_count();
                }
            }
            real_ep.emplace_back(U, cur);
            const auto ep = plan_epochs_for(pol, n_bkt, U);
            if (n_bkt != real.bucket_count() || ep != real_ep) {
                std::fprintf(stderr, "rehash replay differs for U=%zu: %zu vs %zu buckets, %zu vs %zu epochs\n", U, n_bkt,
                             real.bucket_count(), ep.size(), real_ep.size());
                return 1;
            }
            const double v[2] = {(double)n_bkt, (double)ep.size()};
            rec("epochs", (int)U, v, 2);
        }
    }
    return 0;
}
