# usage: bash tools/prof_round.sh [tag] [sections]   (everything under gpurun_out/<tag>/; copy the summaries to profiles/<tag>_*)
# sections (default "test bench times rows kt pmc sq"): test = pytest -m gpu; bench = bench.py; times = isolated kernel times;
# (kt also traces a config-3 run, sq also runs tools/sq_cfg3.sh)
# rows = the widened rows; kt = rocprofv3 kernel trace of bench.py; pmc = FETCH/WRITE passes over the batched path (cfg2 from a fresh
# stream and in the steady state after eight warm-up batches, cfg3) and
# over single frames; sq = SQ counter passes over the batched path
set -x
TAG=${1:-r4a}
SEC=${2:-"test bench times rows kt pmc sq"}
R=$PWD
mkdir -p gpurun_out/$TAG
O=$R/gpurun_out/$TAG
# what the set was taken on: bench.py compares it with the tree it runs in (roofline.stale_profiles); copy to profiles/<tag>_meta.json with the set
python -c "import bench, json, subprocess; print(json.dumps({'tag': '$TAG', 'csrc_sha16': bench.csrc_sha16()}))" > $O/meta.json
has() { case " $SEC " in *" $1 "*) return 0;; *) return 1;; esac; }
if has test; then timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; fi
if has bench; then timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; fi
if has times; then
timeout 300 python tools/kernel_times.py > $O/kernel_times.txt 2>&1
timeout 300 python tools/kernel_times.py cfg3 4 > $O/kernel_times_cfg3.txt 2>&1
MLM_KT_BATCH=1 timeout 300 python tools/kernel_times.py 64 > $O/kernel_times_single.txt 2>&1
timeout 300 python tools/kernel_times.py frontier > $O/kernel_times_frontier.txt 2>&1
timeout 300 python tools/kernel_times.py scatter 4 > $O/kernel_times_scatter.txt 2>&1
fi
if has rows; then timeout 900 python tools/bench_rows.py > $O/rows.json 2> $O/rows.err; fi
cd /tmp && export TMPDIR=/tmp
if has kt; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 12 > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 $R/bench.py --workload cfg3 --batch 32 --batches-per-step 3 --distinct 32 --no-cpu-baseline --no-extra --steps 12 > $O/kt3.log 2>&1
cp $(find $O/kt3 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_cfg3.csv; rm -rf $O/kt3
fi
if has pmc; then
NB=6
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bf -- python3 $R/tools/pmc_batch64.py $NB > $O/bf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bw -- python3 $R/tools/pmc_batch64.py $NB > $O/bw.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/sf -- python3 $R/tools/pmc_batch64.py $NB warm=8 > $O/sf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/sw -- python3 $R/tools/pmc_batch64.py $NB warm=8 > $O/sw.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cf -- python3 $R/tools/pmc_batch64.py cfg3 4 > $O/cf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cw -- python3 $R/tools/pmc_batch64.py cfg3 4 > $O/cw.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/pmc_workload.py 48 > $O/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/pmc_workload.py 48 > $O/pw.log 2>&1
cc() { find $O/$1 -name '*counter_collection.csv' | head -1; }
python $R/tools/pmc_traffic_json.py $(cc bf) $(cc bw) $O/pmc_traffic_batch.json frames=$((NB*64)) > $O/pmc_traffic_batch.log 2>&1
python $R/tools/pmc_traffic_json.py $(cc sf) $(cc sw) $O/pmc_traffic_batch_steady.json frames=$((NB*64)) > $O/pmc_traffic_batch_steady.log 2>&1
python $R/tools/pmc_traffic_json.py $(cc cf) $(cc cw) $O/pmc_traffic_batch_cfg3.json frames=$((4*16)) > $O/pmc_traffic_batch_cfg3.log 2>&1
python $R/tools/pmc_traffic_json.py $(cc pf) $(cc pw) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/bf $O/bw $O/sf $O/sw $O/cf $O/cw $O/pf $O/pw
fi
if has sq; then
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/pmc_batch64.py 3 > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/pmc_batch64.py 3 > $O/sq2.log 2>&1
S=$(find $O/sq -name '*counter_collection.csv' | head -1); S2=$(find $O/sq2 -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_sq_json.py $S $S2 $O/pmc_sq.json frames=192 > $O/pmc_sq.log 2>&1
rm -rf $O/sq $O/sq2
fi
cd $R
if has sq; then bash tools/sq_cfg3.sh $TAG > $O/sq_cfg3.log 2>&1; fi
# (raw counter CSVs are large: only the summaries are kept)
tail -3 $O/pytest.log 2>/dev/null; cat $O/bench.json 2>/dev/null
