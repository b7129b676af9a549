for K in "" "slot_sets=2" "rank_grid=192" "rank_grid=768" "tile_grid=64" "tile_grid=400" "cu_reserve=32" "sec_tab=1024"; do
  echo "== knobs: $K"
  MLM_KNOBS="$K" timeout 300 python bench.py --workload cfg3 --batch 32 --batches-per-step 3 --steps 10 --warmup 2 --distinct 32 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['value_p50']), d['path'], {k:round(v,1) for k,v in r['kernels_us_per_frame'].items() if 'alone' in k or 'instrumented' in k})"
done
