O=gpurun_out/r3ap; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_full.json 2>/dev/null
MLM_RANK_GRID=512 timeout 300 python bench.py --no-cpu-baseline > $O/bench_full_rank512.json 2>/dev/null
tail -n 3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3ap/b*.json')):
    try:
        d=json.loads(open(f).read()); print(f, round(d['value']), d['path'], d.get('extra'))
    except Exception as e: print(f,'ERR',e)
PY
