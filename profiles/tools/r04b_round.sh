#!/bin/bash
# trainer tests + same-box speed comparison + per-workgroup table
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "emulator or training" 2>&1 | tail -2
NB_TRAIN_DEBUG=1 timeout 120 python -c "
import torch
from nautilus_amd import emulator
for d in (50, 100):
    x = torch.randn((2000, d), dtype=torch.float64, device='cuda'); y = torch.rand(2000, dtype=torch.float64, device='cuda')
    emulator.train_networks(x, y, [0], max_epochs=2)
" 2>&1 | grep trainer > $O/train_schedule.txt
bash profiles/tools/r04b_ab.sh "$@" > /dev/null 2>&1
grep -E "^==|D=50 E=4 n=24000|D=100|D=20" $O/train_speed_ab.txt
(NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbgslots.so python profiles/tools/train_slots.py 50 4; NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbgslots.so python profiles/tools/train_slots.py 100 8) 2>&1 | grep -v amdgpu.ids > $O/train_slots.txt
cat $O/train_slots.txt
