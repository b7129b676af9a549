O=gpurun_out/r3al; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
run() { # name threads tab
  for i in 1 2; do MLM_SEC_THREADS=$2 MLM_SEC_TAB=$3 timeout 300 python bench.py --no-cpu-baseline --no-extra > $O/b_$1_$i.json 2>/dev/null; done
}
run t256_tab512 256 512
run t256_tab1024 256 1024
run t512_tab512 512 512
run t512_tab1024 512 1024
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null
timeout 300 python tools/kernel_times.py > $O/kt.txt 2>&1
tail -n 3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3al/b*.json')):
    try:
        d=json.loads(open(f).read()); print(f, round(d['value']), d['path'], d.get('extra',{}).get('cfg3',{}).get('value'), d.get('extra',{}).get('single_frame_us'))
    except Exception as e: print(f,'ERR',e)
PY
grep -v amdgpu $O/kt.txt
