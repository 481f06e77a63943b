#!/bin/bash
# Round 5 (fourth session): trainer with the output layer / delta 4 / delta 3 /
# delta 2 as one stage -- parity of the training tests, speed, phase stamps.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-fused_tail}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "emulator or training" > $O/train_tests_$T.log 2>&1
tail -3 $O/train_tests_$T.log
timeout 300 python profiles/tools/train_speed.py > $O/train_speed_$T.txt 2>&1
cat $O/train_speed_$T.txt
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg0.so timeout 300 python profiles/tools/train_phases.py > $O/train_phases_$T.txt 2>&1
cat $O/train_phases_$T.txt
