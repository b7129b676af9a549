"""The oracle leg of the two longest GPU parity tests as a process of its own, started when the session's collection contains those tests
(tests/conftest.py) so that it runs on another core WHILE the other GPU tests run: the CPU oracle integrates ~7 VGA frames a second, the
1 000-frame stream is two and a half minutes of it.  The tests pick the oracle's maps up from files (np.savez, one per checkpoint).
TEST INFRASTRUCTURE: like everything that touches oracle/, never imported by the product.

    python -m tests.oracle_worker long_stream <frames> <dir>     # tests/test_gpu_parity.py::test_long_stream_cfg2
    python -m tests.oracle_worker batch64 <dir>                  # tests/test_gpu_parity.py::test_bench_batch64_parity
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _dump(path, blocks):
    tmp = path + ".tmp.npz"
    np.savez(tmp, **blocks)
    os.replace(tmp, path)  # (the reader never sees a half-written file)


def long_stream_inputs(n):
    """frames, q, t of test_long_stream_cfg2 (the test and the worker must feed the same stream)"""
    from mlmapping_amd import synthetic as syn
    from mlmapping_amd.config import S1

    distinct = 32
    base = syn.room_depth(S1)
    frames = np.stack([syn.jitter_depth(base, k, seed=42) for k in range(distinct)])
    poses = syn.random_poses(n, seed=42)
    return frames, np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses])


def long_stream(n, out):
    from mlmapping_amd.config import S1
    from oracle.binding import OracleMap

    frames, q, t = long_stream_inputs(n)
    # the second, short map first: the test asks for it last, but it is quick
    B = 25
    c2 = OracleMap(S1)
    for k in range(min(4 * B, n)):
        c2.update_depth(frames[k % B], q[k], t[k])
    _dump(os.path.join(out, "batch_dev.npz"), c2.export_blocks())
    cpu = OracleMap(S1)
    for k in range(n):
        cpu.update_depth(frames[k % frames.shape[0]], q[k], t[k])
        if (k + 1) % 50 == 0:
            _dump(os.path.join(out, f"ckpt_{k + 1}.npz"), cpu.export_blocks())


def batch64(out):
    from bench import make_inputs
    from mlmapping_amd.config import S1
    from oracle.binding import OracleMap

    B, nb = 64, 3
    frames, q, t = make_inputs(S1, B, B * nb, seed=42)
    cpu = OracleMap(S1)
    for k in range(B * nb):
        cpu.update_depth(frames[k % B], q[k], t[k])
    _dump(os.path.join(out, "after_192.npz"), cpu.export_blocks())
    for k in range(B):
        cpu.update_depth(frames[k], q[k], t[k])
    _dump(os.path.join(out, "after_256.npz"), cpu.export_blocks())


if __name__ == "__main__":
    try:
        if sys.argv[1] == "long_stream":
            long_stream(int(sys.argv[2]), sys.argv[3])
        elif sys.argv[1] == "batch64":
            batch64(sys.argv[2])
        open(os.path.join(sys.argv[-1], "done"), "w").write("ok")
    except BaseException as e:  # the waiting test reports it
        open(os.path.join(sys.argv[-1], "failed"), "w").write(repr(e))
        raise
