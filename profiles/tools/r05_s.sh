#!/bin/bash
# Round 5 (fourth session): trainer with the forward / backward pass
# rescheduled (sub-tiles for the small layers, operand loads from inside the
# MFMA chains) -- training parity, speed, phase stamps, the bench line and the
# exploration summary.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "emulator or training" > $O/train_tests_final.log 2>&1
tail -3 $O/train_tests_final.log
timeout 300 python profiles/tools/train_speed.py 2>&1 | grep -v amdgpu.ids > $O/train_speed_final.txt
cat $O/train_speed_final.txt
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg0.so timeout 300 python profiles/tools/train_phases.py 2>&1 | grep -v amdgpu.ids > $O/train_phases_final.txt
timeout 600 python bench.py > $O/bench_r05b.json 2> $O/bench_r05b.err
tail -c 1500 $O/bench_r05b.json
timeout 420 python profiles/tools/explore_profile.py 2>&1 | grep -E "^wall|^bounds|^ rows" | head -12 > $O/explore_summary_b.txt
head -3 $O/explore_summary_b.txt | cut -c1-500
