"""The pure host arithmetic of the product (mlmapping_amd/csrc/mlm_host.h: Eigen/Sophus pose pieces, latency compensation,
odds table, 32FC1 conversion, rehash-policy replay) built for the CPU with -fsanitize=address,undefined and checked

* bit for bit against the oracle's restatements on the same inputs, and
* against the property tests the reference itself holds for Sophus (3rdPartLib/Sophus/sophus/test_so3.cpp:14-110,
  test_se3.cpp:10-86): exp(log(R)) == R, theta in [-pi, pi], R*p == matrix*p, R*R^-1 == I, T*p == R p + t,
  T*T^-1 == I, all to SMALL_EPS = 1e-10 (so3.h:35) — the only reference-held checks that touch this path.

Both the oracle's functions and the product's are held to those properties.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL_EPS = 1e-10  # so3.h:35
PI = 3.14159265358979323846
PI_F = 3.14159265  # test_se3.cpp:12


@pytest.fixture(scope="module")
def rec(tmp_path_factory):
    exe = tmp_path_factory.mktemp("hm") / "host_math_driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-Wall", "-Werror", "-I", os.path.join(ROOT, "mlmapping_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "host_math_driver.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    r = {}
    for line in out.splitlines():
        f = line.split()
        r.setdefault(f[0], {})[int(f[1])] = np.array([float.fromhex(x) for x in f[2:]])
    return r


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _so3_cases(B):
    e, m = B.so3_exp, B.so3_mul
    return [B.so3_from_quat([0.1e-11, 0., 1., 0.]), B.so3_from_quat([-1, 0.00001, 0.0, 0.0]), e([0.2, 0.5, 0.0]),
            e([0.2, 0.5, -1.0]), e([0., 0., 0.]), e([0., 0., 0.00001]), e([PI, 0, 0]),
            m(m(e([0.2, 0.5, 0.0]), e([PI, 0, 0])), e([-0.2, -0.5, -0.0])),
            m(m(e([0.3, 0.5, 0.1]), e([PI, 0, 0])), e([-0.3, -0.5, -0.1]))]


def _se3_cases(B):
    e = B.so3_exp
    T = lambda q, t: np.concatenate([q, np.asarray(t, dtype=np.float64)])
    m = B.se3_mul
    return [T(e([0.2, 0.5, 0.0]), [0, 0, 0]), T(e([0.2, 0.5, -1.0]), [10, 0, 0]), T(e([0., 0., 0.]), [0, 100, 5]),
            T(e([0., 0., 0.00001]), [0, 0, 0]), T(e([0., 0., 0.00001]), [0, -0.00000001, 0.0000000001]),
            T(e([0., 0., 0.00001]), [0.01, 0, 0]), T(e([PI_F, 0, 0]), [4, -5, 0]),
            m(m(T(e([0.2, 0.5, 0.0]), [0, 0, 0]), T(e([PI_F, 0, 0]), [0, 0, 0])), T(e([-0.2, -0.5, -0.0]), [0, 0, 0])),
            m(m(T(e([0.3, 0.5, 0.1]), [2, 0, -7]), T(e([PI_F, 0, 0]), [0, 0, 0])), T(e([-0.3, -0.5, -0.1]), [0, 6, 0]))]


def _R(B, q):
    return B.so3_matrix(q)


def test_so3_properties_and_parity(rec, oracle_lib):
    from oracle import binding as B

    cases = _so3_cases(B)
    assert len(cases) == len(rec["so3"]) == 9
    p = np.array([1.0, 2.0, 4.0])
    for i, q in enumerate(cases):
        # product == oracle, bit for bit
        assert np.array_equal(_bits(rec["so3"][i]), _bits(q)), f"case {i}: quaternion"
        lg = B.so3_log(q)
        assert np.array_equal(_bits(rec["so3_log"][i][:3]), _bits(lg)), f"case {i}: log"
        assert np.array_equal(_bits(rec["so3_explog"][i]), _bits(B.so3_exp(lg))), f"case {i}: exp(log)"
        assert np.array_equal(_bits(rec["so3_R"][i].reshape(3, 3)), _bits(_R(B, q))), f"case {i}: matrix"
        # the reference's properties (test_so3.cpp:34-107), on the oracle's and on the product's numbers
        for what, qq, q2, th in (("oracle", q, B.so3_exp(lg), None), ("product", rec["so3"][i], rec["so3_explog"][i], rec["so3_log"][i][3])):
            R1, R2 = _R(B, qq), _R(B, q2)
            assert np.linalg.norm(R1 - R2) <= SMALL_EPS, f"{what} case {i}: SO3 - exp(log(SO3))"
            if th is not None:
                assert -np.pi <= th <= np.pi, f"{what} case {i}: log theta not in [-pi,pi]"
        theta = np.linalg.norm(lg)
        assert theta <= np.pi + 1e-12
        for what, rp in (("oracle", B.se3_apply(np.concatenate([q, [0, 0, 0]]), p)), ("product", rec["so3_p"][i])):
            assert np.linalg.norm(rp - _R(B, q) @ p) <= SMALL_EPS, f"{what} case {i}: transform vector"
        qi = B.se3_inverse(np.concatenate([q, [0, 0, 0]]))[:4]
        assert np.array_equal(_bits(rec["so3_inv"][i]), _bits(qi)), f"case {i}: inverse"
        assert np.linalg.norm(_R(B, q) @ _R(B, qi) - np.eye(3)) <= SMALL_EPS, f"case {i}: inverse property"


def test_se3_properties_and_parity(rec, oracle_lib):
    from oracle import binding as B

    cases = _se3_cases(B)
    p = np.array([1.0, 2.0, 4.0])
    for i, T in enumerate(cases):
        assert np.array_equal(_bits(rec["se3"][i]), _bits(T)), f"case {i}: SE3"
        Ti = B.se3_inverse(T)
        assert np.array_equal(_bits(rec["se3_inv"][i]), _bits(Ti)), f"case {i}: inverse"
        assert np.array_equal(_bits(rec["se3_mul_inv"][i]), _bits(B.se3_mul(T, Ti))), f"case {i}: T * T^-1"
        tp = B.se3_apply(T, p)
        assert np.array_equal(_bits(rec["se3_p"][i]), _bits(tp)), f"case {i}: T * p"
        # test_se3.cpp:47-84
        R, t = _R(B, T[:4]), T[4:]
        assert np.linalg.norm(tp - (R @ p + t)) <= SMALL_EPS, f"case {i}: transform vector"
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R, t
        Mi = np.eye(4)
        Mi[:3, :3], Mi[:3, 3] = _R(B, Ti[:4]), Ti[4:]
        # the reference compares with SMALL_EPS on cases whose translations reach 100: the worst entry is ~1e-14 here
        assert np.linalg.norm(M @ Mi - np.eye(4)) <= SMALL_EPS, f"case {i}: inverse"


def test_frame_pose_and_compensation_match_oracle(rec, oracle_lib):
    """T_ls / t_wa (map_awareness.cpp:184-186) and the latency-compensated T_wb (mlmap.cpp:485-498) of the product's host
    code, bit for bit against the oracle."""
    from mlmapping_amd.config import S1
    from oracle.binding import OracleMap

    cfg = S1.with_(width=4, height=4)
    cpu = OracleMap(cfg)
    assert np.array_equal(_bits(rec["q_bs"][0]), _bits([-0.5, 0.5, -0.5, 0.5]))
    depth = np.full((4, 4), 1000, dtype=np.uint16)
    for i in sorted(rec["pose_in"]):
        q, t = rec["pose_in"][i][:4], rec["pose_in"][i][4:]
        cpu.awareness_points(np.array([[0.1, 0.2, 1.0]]), q, t)
        cq, ct = cpu.T_ls()
        out = rec["pose_out"][i]
        assert np.array_equal(_bits(out[:4]), _bits(cq)) and np.array_equal(_bits(out[4:7]), _bits(ct)), f"pose {i}: T_ls"
        assert np.array_equal(_bits(out[7:]), _bits(t)), f"pose {i}: t_wa"
        v, w = rec["comp_in"][i][:3], rec["comp_in"][i][3:]
        tc = cpu.depth_odom_callback(depth, t_img=10.0 + i / 30.0, odom_p=t, odom_q=q, odom_v=v, t_odom=10.0 + i / 30.0 - 0.004,
                                     imu_w=w, t_imu=10.0 + i / 30.0 - 0.002, latency=0.085, sampled=False)
        assert np.array_equal(_bits(rec["comp_out"][i]), _bits(tc)), f"pose {i}: compensated T_wb"


def test_odds_table_and_conversion_match_oracle(rec, oracle_lib):
    from mlmapping_amd.config import S1
    from oracle import binding as B

    tab = B.OracleMap(S1).odds_table()
    got = np.stack([rec["odds"][d] for d in range(21)]).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), tab.view(np.uint32))
    inp = np.array([0.0, 1.0, 1.0005, 1.0015, 65.5354, 65.536, 70.0, -1.0, 2147483.5, 2147483.75, 3.0e6, np.inf, -np.inf, np.nan,
                    0.0004, 0.0005, 0.00051], dtype=np.float32)
    want = np.array([0, 1000, 1000, 1002, 65535, 65535, 65535, 0, 65535, 0, 0, 0, 0, 0, 0, 0, 1])
    assert np.array_equal(rec["cvt"][0].astype(np.int64), want)      # cvRound + saturate_cast<ushort> on x86
    assert np.array_equal(B.cv_f32_to_u16(inp).astype(np.int64), want)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1, 70, 100000), rng.integers(0, 66000, 50000) / 1000.0 + 0.0005]).astype(np.float32)
    # numpy restatement of the same rule: float product, round half to even, clamp
    s = x * np.float32(1000.0)
    ref = np.clip(np.rint(s), 0, 65535).astype(np.uint16)
    assert np.array_equal(B.cv_f32_to_u16(x), ref)


def test_log10f_restatement_matches_host_libm(rec):
    """mlm_glibc_log10f (the device's logit) equals this host's log10f bit for bit on the driver's sweep (the driver exits
    with an error on any mismatch); reference: logit macro map_local.h:8 evaluated by the host libm, map_local.cpp:159."""
    n, bad = rec["log10f"][0]
    assert n > 8e6 and bad == 0
