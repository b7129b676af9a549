# A/B of two builds of the library (default vs mlmapping_amd/lib/libmlmap_hip_alt.so, `make -C mlmapping_amd/csrc alt ALT_FLAGS=...`):
# config 3, config 2 and the scatter scene, two rounds each
R=$PWD
for round in 1 2; do
for L in "" "$R/mlmapping_amd/lib/libmlmap_hip_alt.so"; do
  echo "== lib: ${L:-default}"
  MLMAP_HIP_LIB=$L timeout 300 python bench.py --workload cfg3 --batch 32 --batches-per-step 3 --steps 10 --warmup 2 --distinct 32 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg3', round(d['value']), round(d['value_p50']), {k.split(' ')[0]:round(v,1) for k,v in r['kernels_us_per_frame'].items() if 'alone' in k})"
  MLMAP_HIP_LIB=$L timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg2', round(d['value']), round(d['value_p50']), {k.split(' ')[0]:round(v,2) for k,v in r['kernels_us_per_frame'].items() if 'alone' in k})"
  MLMAP_HIP_LIB=$L timeout 300 python tools/scatter_rate.py 2>/dev/null | tail -2
done
done
