"""Per-workgroup time table of the resident training kernel (library built with
-DNB_TRAIN_TIMING -DNB_TRAIN_SLOT_TIMING: profiles/tools/build_train_dbg.sh
slots): ticks per step between the two barriers of a step, for each of the 32
workgroups of network 0.  The workgroup with the smallest wait at a barrier is
the one the others wait for.
  NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbgslots.so python profiles/tools/train_slots.py [D E]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nautilus_amd import _lib, emulator  # noqa: E402

lib = _lib.load()
fn = getattr(lib, 'nb_dbg_train_slot_times')
d, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 4)
n_row = 24000
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda')
y = torch.rand(n_row, dtype=torch.float64, device='cuda')
emulator.train_networks(x, y, list(range(e)), max_epochs=2)
buf = (ctypes.c_longlong * 256)()
fn(buf)
torch.cuda.synchronize()
t = time.perf_counter()
emulator.train_networks(x, y, list(range(e)), max_epochs=32,
                        hparams=dict(n_iter_no_change=100000))
torch.cuda.synchronize()
dt = time.perf_counter() - t
fn(buf)
a = np.array(list(buf), dtype=float).reshape(32, 8)
steps = a[:, 4].max()
print('D=%d E=%d: %.2f us/step wall, %d steps' % (
    d, e, dt / (32 * ((n_row + 199) // 200)) * 1e6, steps))
print('workgroup   FB / early job   wait 1   late job   wait 2   sum')
for s in range(32):
    v = a[s, :4] / max(a[s, 4], 1)
    print('   %2d        %8.0f   %8.0f   %8.0f %8.0f %8.0f' % (
        s, v[0], v[1], v[2], v[3], v.sum()))
