O=gpurun_out/r3ai; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/kernel_times.py > $O/kt.txt 2>&1
timeout 300 python tools/kernel_times.py cfg3 4 > $O/kt_cfg3.txt 2>&1
timeout 300 python tools/kernel_times.py scatter 4 > $O/kt_scatter.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_$i.json 2>/dev/null; done
tail -n 5 $O/pytest.log
grep -v amdgpu $O/kt.txt $O/kt_cfg3.txt $O/kt_scatter.txt
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f'gpurun_out/r3ai/bench_{i}.json').read())
    print(round(d['value']), round(d['extra']['cfg3']['value']))
PY
