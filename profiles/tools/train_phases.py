"""Per-phase ticks of the resident training kernel (debug library:
make debug DEFS=-DNB_TRAIN_TIMING; NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg.so
python profiles/tools/train_phases.py).  The stamps themselves cost a few us
per step: read the proportions."""
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
fn = getattr(lib, 'nb_dbg_train_times')
ORDER = [(0, 'loop top (incl. stamp folding)'),
         (11, 'fb weights issued, input block in LDS'), (12, 'fb L1'),
         (13, 'fb L2'), (14, 'fb L3'), (15, 'fb out + d4'), (16, 'fb d3'),
         (17, 'fb d2'), (18, 'fb d1'), (19, 'fb stash stores issued'),
         (1, 'fb return'), (2, 'barrier 1'),
         (33, 'g job record + layer decode'), (30, 'g operand loads issued'),
         (31, 'g MFMA chains -> LDS'), (36, 'g LDS barrier'),
         (32, 'g reduce + Adam + stores issued'), (3, 'g return'),
         (4, 'barrier 2 + next rows prefetch')]
if int(os.environ.get('NB_PHASE_SLOT', '0')) >= 13:
    # a workgroup without a row tile (library built with
    # -DNB_TRAIN_TIMING_SLOT=<slot>, profiles/tools/build_train_dbg.sh)
    ORDER = [(0, 'loop top (incl. stamp folding)'), (1, 'no FB work'),
             (5, 'wait for the upper stash'),
             (33, 'early: g job record + layer decode'),
             (30, 'early: g operand loads issued'),
             (31, 'early: g MFMA chains -> LDS'), (36, 'early: g LDS barrier'),
             (32, 'early: g reduce + Adam + stores issued'),
             (6, 'early job returns'), (2, 'wait for barrier 1'),
             (3, 'late job (if any)'), (4, 'barrier 2')]
CASES = [(50, 24000, 4), (100, 24000, 8)]
if len(sys.argv) > 1:
    CASES = [(int(sys.argv[1]), 24000, int(sys.argv[2]))]
for d, n_row, e in CASES:
    x = torch.randn((n_row, d), dtype=torch.float64, device='cuda')
    y = torch.rand(n_row, dtype=torch.float64, device='cuda')
    emulator.train_networks(x, y, list(range(e)), max_epochs=2)
    buf = (ctypes.c_longlong * 64)()
    fn(buf)
    n_ep = 32
    torch.cuda.synchronize()
    t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_ep,
                            hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    fn(buf)
    steps = n_ep * ((n_row + 199) // 200)
    ticks = np.array(list(buf), dtype=float) / steps
    total = sum(ticks[k] for k, _ in ORDER)
    print('D=%d E=%d: %.2f us/step wall; ticks/step %.0f' % (
        d, e, dt / steps * 1e6, total))
    if ORDER[1][0] == 11:
        fb = sum(ticks[k] for k in (11, 12, 13, 14, 15, 16, 17, 18, 19, 1))
        g = sum(ticks[k] for k in (33, 30, 31, 36, 32, 3))
        print('   FB %.0f   G %.0f' % (fb, g))
    for k, name in ORDER:
        print('      %-42s %8.0f' % (name, ticks[k]))
