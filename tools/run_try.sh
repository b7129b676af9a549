O=gpurun_out/r3z; mkdir -p $O
timeout 300 python tools/bin_probe.py > $O/bin_probe.txt 2>&1
cat $O/bin_probe.txt
