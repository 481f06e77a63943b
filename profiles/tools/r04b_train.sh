#!/bin/bash
# Round 4 (second session): trainer after the paired-row stash / raw gather /
# hoisted step size -- parity tests, speed, phase stamps.
# gpurun -- bash profiles/tools/r04b_train.sh
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "emulator or training" > $O/train_tests.log 2>&1
tail -3 $O/train_tests.log
timeout 300 python profiles/tools/train_speed.py > $O/train_speed.txt 2>&1
cat $O/train_speed.txt
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg.so timeout 300 python profiles/tools/train_phases.py > $O/train_phases.txt 2>&1
cat $O/train_phases.txt
