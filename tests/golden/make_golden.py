"""Regenerate tests/golden/oracle_digests.json: SHA-256 digests of the CPU oracle's outputs on the reference
configurations.  The reference ships no golden vectors; the counts of SURVEY §8d pin the oracle
(tests/test_oracle_kat.py) and these digests freeze its full output — every hit cell, odd, miss cell, block key,
occupancy byte and log-odds bit — so that a later edit of the oracle cannot drift silently.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1, S3, SDEF  # noqa: E402
from oracle.binding import OracleMap  # noqa: E402


def h(*arrays):
    m = hashlib.sha256()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def digest(m: OracleMap):
    cells, odds = m.hit_cells_sorted()
    b = m.export_blocks()
    return {"hits": h(cells, odds), "misses": h(np.sort(m.misses())), "n_hit": int(cells.size), "n_miss": int(m.misses().size),
            "blocks": h(b["keys"], b["collapsed"]), "occ": h(b["occ"]), "log_odds": h(b["log_odds"]), "infl": h(b["infl"]),
            "frontier": h(m.export_frontier()), "n_blocks": int(b["keys"].shape[0])}


CASES = {
    "cfg1_room_static_f0_f4": (S1, "room", "static", [0, 4]),
    "cfg2_jitter_random_f0_f3": (S1, "room_jitter", "random", [0, 3]),
    "sdef_translating_f0_f5": (SDEF, "room", "translating", [0, 5]),
    "cfg3_room_f0": (S3, "room", "static", [0]),
    "explore_s1n5_f0_f3": (S1.with_(use_exploration_frontiers=True, subbox_n=5), "room_jitter", "smooth", [0, 3]),
    "corridor_f0_f2": (S1, "corridor", "translating", [0, 2]),
}


def compute():
    out = {}
    for name, (cfg, scene, poses, frames) in CASES.items():
        m = OracleMap(cfg)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, scene, poses, max(frames) + 1)):
            m.update_depth(img, q, t)
            if k == max(frames):
                m.inflate_map(t)
            if k in frames:
                out[f"{name}/{k}"] = digest(m)
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
    json.dump(compute(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
