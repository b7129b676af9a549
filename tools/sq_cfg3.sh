# SQ counter passes over the batched config-3 path (as tools/prof_round.sh does for config 2): gpurun_out/<tag>/pmc_sq_cfg3.json
TAG=${1:-r5a}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq3 -- python3 $R/tools/pmc_batch64.py cfg3 4 > $O/sq3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/sq32 -- python3 $R/tools/pmc_batch64.py cfg3 4 > $O/sq32.log 2>&1
S=$(find $O/sq3 -name '*counter_collection.csv' | head -1); S2=$(find $O/sq32 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_sq_json.py $S $S2 $O/pmc_sq_cfg3.json frames=64 > $O/pmc_sq_cfg3.log 2>&1
rm -rf $O/sq3 $O/sq32
cd $R
python - <<PY
import json
d=json.load(open("$O/pmc_sq_cfg3.json"))
for k,v in d["kernels"].items():
    if v.get("SQ_INSTS_VALU",0)>1e4: print(k, {a:round(b) if b>10 else round(b,3) for a,b in v.items() if a in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM","SQ_ACTIVE_INST_VALU","SQ_WAVES","wait_any_over_wave_cycles","active_inst_valu_over_wave_cycles","SQ_ACTIVE_INST_LDS","SQ_BUSY_CYCLES")})
PY
