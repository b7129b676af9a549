O=gpurun_out/r3au; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
MLM_KT_BATCH=1 timeout 300 python tools/kernel_times.py 64 > $O/kernel_times_single.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench_full.json 2>/dev/null
tail -n 3 $O/pytest.log
grep -v amdgpu $O/kernel_times_single.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3au/bench_full.json').read()); print(round(d['value']), d['extra'])
PY
