# usage: bash tools/prof_round.sh [tag]   (everything under gpurun_out/<tag>/; copy the summaries to profiles/<tag>_*)
set -x
TAG=${1:-r3a}
R=$PWD
mkdir -p gpurun_out/$TAG
O=$R/gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/kernel_times.py > $O/kernel_times.txt 2>&1
timeout 300 python tools/kernel_times.py cfg3 4 > $O/kernel_times_cfg3.txt 2>&1
MLM_KT_BATCH=1 timeout 300 python tools/kernel_times.py 64 > $O/kernel_times_single.txt 2>&1
timeout 300 python tools/kernel_times.py frontier > $O/kernel_times_frontier.txt 2>&1
timeout 300 python tools/kernel_times.py scatter 4 > $O/kernel_times_scatter.txt 2>&1
timeout 900 python tools/bench_rows.py > $O/rows.json 2> $O/rows.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 60 > $O/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/pmc_workload.py 48 > $O/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/pmc_workload.py 48 > $O/pw.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/pmc_batch.py > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/pmc_batch.py > $O/sq2.log 2>&1
cd $R
F=$(find $O/pf -name '*counter_collection.csv' | head -1); W=$(find $O/pw -name '*counter_collection.csv' | head -1); S=$(find $O/sq -name '*counter_collection.csv' | head -1); S2=$(find $O/sq2 -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic_json.py $F $W $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
python tools/pmc_sq_json.py $S $S2 $O/pmc_sq.json > $O/pmc_sq.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
# raw counter CSVs are large: keep only summaries
rm -rf $O/pf $O/pw $O/sq $O/sq2 $O/kt
tail -3 $O/pytest.log; cat $O/bench.json
