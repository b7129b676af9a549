O=gpurun_out/r3af; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/latency_probe.py dense > $O/lat_dense.txt 2>&1
timeout 300 python tools/latency_probe.py sampled > $O/lat_sampled.txt 2>&1
timeout 300 python tools/latency_probe.py sampled 300 sdef > $O/lat_sampled_sdef.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
tail -n 3 $O/pytest.log
grep -v amdgpu $O/lat*.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3af/bench.json').read())
print(round(d['value']), d['extra'])
PY
