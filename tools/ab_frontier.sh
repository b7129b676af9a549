# A/B of the default library vs libmlmap_hip_alt.so on frontier mode's rows (frame by frame, batches, the config2.yaml callback)
R=$PWD
for round in 1 2; do
for L in "" "$R/mlmapping_amd/lib/libmlmap_hip_alt.so"; do
  echo "== lib: ${L:-default}"
  MLMAP_HIP_LIB=$L python tools/frontier_latency.py 2>/dev/null | head -2
  MLMAP_HIP_LIB=$L python tools/bench_rows.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k.replace('frontier_mode_','').replace('_frames_per_s',''):round(v['gpu']) for k,v in d.items() if k.startswith('frontier')}, round(d['callback_sampled500_frames_per_s']['gpu_incl_pcie']))"
done
done
