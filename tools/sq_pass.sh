R=$PWD; mkdir -p gpurun_out/sq2; O=$R/gpurun_out/sq2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/pmc_batch.py > $O/sq.log 2>&1
cd $R
S=$(find $O/sq -name '*counter_collection.csv' | head -1)
python tools/pmc_sq_json.py $S $O/pmc_sq.json | head -8
rm -rf $O/sq
