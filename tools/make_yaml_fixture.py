"""Generates tests/golden/reference_yaml_params.json: the parsed key/value pairs (no comments, no file text) of the two shipped
config files of the reference that carry the current keys — data, so that the YAML-parity test also runs where /root/reference
does not exist (the GPU box).  Run in the authoring container: python tools/make_yaml_fixture.py"""
import json
import os

import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd.config import load_reference_yaml  # noqa: E402

REF = "/root/reference/launch/config"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_yaml_params.json")
out = {}
for name in ("config_sim.yaml", "config2.yaml"):
    y = load_reference_yaml(os.path.join(REF, name))
    out[name] = {k: y[k] for k in sorted(y) if k.startswith("mlmapping_") or k in ("use_exploration_frontiers", "camera2odom_latency", "T_B_S")}
with open(OUT, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("wrote", OUT)
