O=gpurun_out/r3ag; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
MLM_KT_BATCH=1 timeout 300 python tools/kernel_times.py 64 > $O/kt_single.txt 2>&1
timeout 300 python tools/kernel_times.py > $O/kt.txt 2>&1
timeout 300 python tools/latency_probe.py dense > $O/lat_dense.txt 2>&1
timeout 300 python tools/latency_probe.py sampled 300 sdef > $O/lat_sampled_sdef.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/bench_$i.json 2>/dev/null; done
tail -n 3 $O/pytest.log
grep -v amdgpu $O/kt_single.txt $O/kt.txt $O/lat*.txt
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f'gpurun_out/r3ag/bench_{i}.json').read())
    print(round(d['value']))
PY
