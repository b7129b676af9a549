O=gpurun_out/r3at; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/b_def$i.json 2>/dev/null; done
for i in 1 2; do MLMAP_HIP_LIB=$PWD/mlmapping_amd/lib/libmlmap_hip_alt.so timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/b_rankwpe8_$i.json 2>/dev/null; done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_full.json 2>/dev/null
MLM_DEBUG_CREATE=1 timeout 300 python tools/kernel_times.py > $O/kt.txt 2>&1
tail -n 3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3at/b*.json')):
    try:
        d=json.loads(open(f).read()); print(f, round(d['value']), d['path'], d.get('extra'))
    except Exception as e: print(f,'ERR',e)
PY
grep -v amdgpu $O/kt.txt
