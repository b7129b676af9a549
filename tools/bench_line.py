import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(round(d["value"]), r["kernel"], round(r["achieved"], 1), "GB/s", round(r["avg_launch_us"], 1), "us/launch",
      {k.split()[0] + ("*" if "timed" in k else ""): round(v, 1) for k, v in r["kernels_us_per_frame"].items()})
