import json,sys
d=json.loads(sys.stdin.read()); print(round(d["value"]), {k.split()[0]:round(v,1) for k,v in d["roofline"]["kernels_us_per_frame"].items()})
