O=gpurun_out/r3ah; mkdir -p $O
timeout 300 python tools/sector_phase.py single > $O/phase_single.txt 2>&1
timeout 300 python tools/sector_phase.py > $O/phase.txt 2>&1
grep -v amdgpu $O/phase_single.txt $O/phase.txt
