set -x
O=gpurun_out/r3s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/kernel_times.py > $O/kernel_times.txt 2>&1
MLMAP_HIP_LIB=$PWD/mlmapping_amd/lib/libmlmap_hip_alt.so timeout 600 python bench.py --no-cpu-baseline > $O/bench_alt512.json 2> $O/bench_alt512.err
MLMAP_HIP_LIB=$PWD/mlmapping_amd/lib/libmlmap_hip_alt.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_alt512.log 2>&1
tail -3 $O/pytest.log $O/pytest_alt512.log; cat $O/bench.json; cat $O/bench_alt512.json
