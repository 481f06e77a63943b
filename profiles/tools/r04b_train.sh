#!/bin/bash
# Round 4 (second session): trainer -- parity tests, speed, phase stamps.
# gpurun -- bash profiles/tools/r04b_train.sh [tag]
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
T=${1:-run}
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "emulator or training" > $O/train_tests_$T.log 2>&1
tail -3 $O/train_tests_$T.log
NB_TRAIN_DEBUG=1 timeout 120 python -c "
import torch
from nautilus_amd import emulator
x = torch.randn((2000, 50), dtype=torch.float64, device='cuda'); y = torch.rand(2000, dtype=torch.float64, device='cuda')
emulator.train_networks(x, y, [0], max_epochs=2)
" 2>&1 | grep trainer | head -60 > $O/train_schedule_$T.txt
timeout 300 python profiles/tools/train_speed.py > $O/train_speed_$T.txt 2>&1
cat $O/train_speed_$T.txt
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg0.so timeout 300 python profiles/tools/train_phases.py 100 8 > $O/train_phases_$T.txt 2>&1
cat $O/train_phases_$T.txt
