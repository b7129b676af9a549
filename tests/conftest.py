import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A test that hangs (a kernel that never ends, a lost wake-up) must fail with the stacks of its threads instead of holding the
    # whole run until somebody's outer limit kills it without a line of output: 15 minutes per test where pytest-timeout is installed
    # (the longest test, the 1 000-frame stream, takes three).
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(900, method="thread"))


def pytest_collection_finish(session):
    # (after deselection by -m / -k: the workers only start for tests that will run — never in a CPU-only session)
    if not session.config.option.collectonly:
        _start_oracle_jobs(session.items)


# ---- the oracle legs of the two longest GPU tests run in processes of their own, started right after collection (they finish while
#      the other tests run; tests/oracle_worker.py)
_ORACLE_JOBS = {}


def _start_oracle_jobs(items):
    import subprocess
    import tempfile

    names = {it.name.split("[")[0] for it in items}
    want = []
    if "test_long_stream_cfg2" in names:
        n = int(os.environ.get("MLM_LONG_STREAM_FRAMES", "1000")) // 50 * 50
        want.append(("long_stream", ["long_stream", str(n)]))
    if "test_bench_batch64_parity" in names:
        want.append(("batch64", ["batch64"]))
    for key, argv in want:
        d = tempfile.mkdtemp(prefix=f"mlm_oracle_{key}_")
        p = subprocess.Popen([sys.executable, "-m", "tests.oracle_worker", *argv, d], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        _ORACLE_JOBS[key] = (p, d)


def pytest_sessionfinish(session, exitstatus):
    import shutil

    for p, d in _ORACLE_JOBS.values():
        if p.poll() is None:
            p.kill()
        shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="session")
def oracle_jobs():
    """{job: fetch(name, timeout)}: the oracle's map dumps of a job started at collection time, as dicts like OracleMap.export_blocks()"""
    import time

    import numpy as np

    def fetcher(key):
        p, d = _ORACLE_JOBS[key]

        def fetch(name, timeout=1500.0):
            path, t0 = os.path.join(d, name), time.time()
            while not os.path.exists(path):
                if os.path.exists(os.path.join(d, "failed")) or (p.poll() not in (None, 0)):
                    raise RuntimeError(f"oracle worker {key} failed: " + (p.stderr.read().decode(errors='replace')[-2000:] if p.stderr else ""))
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"oracle worker {key}: {name} not there after {timeout:.0f} s")
                time.sleep(0.05)
            with np.load(path) as z:
                out = {k: z[k] for k in z.files}
            os.remove(path)
            return out

        return fetch

    return {k: fetcher(k) for k in _ORACLE_JOBS}


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding

    binding.build()
    return binding.lib()


@pytest.fixture
def knobs():
    """Test / experiment knobs of the library (mlm_debug_set): read by the next mlm_create, forgotten again after the test."""
    from mlmapping_amd import mlmap

    class Knobs:
        def set(self, name, value):
            mlmap.debug_set(name, int(value))

    yield Knobs()
    mlmap.debug_reset()
