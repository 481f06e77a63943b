"""Per-phase ticks of the resident training kernel (debug library:
make debug DEFS=-DNB_TRAIN_TIMING)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nautilus_amd import emulator, _lib
lib = _lib.load()
fn = getattr(lib, 'nb_dbg_train_times')
for d, nrow in [(50, 24000), (100, 24000)]:
    X = torch.randn((nrow, d), dtype=torch.float64, device='cuda')
    y = torch.rand(nrow, dtype=torch.float64, device='cuda')
    emulator.train_networks(X, y, [0, 1, 2, 3], max_epochs=2)
    buf = (ctypes.c_longlong * 64)()
    fn(buf)
    ne = 32
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(X, y, [0, 1, 2, 3], max_epochs=ne, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    fn(buf)
    steps = ne * ((nrow + 199) // 200)
    t = np.array(list(buf), dtype=float) / steps
    fbn = ['weights issued, input block in LDS', 'L1', 'L2',
           'L3', 'out + d4', 'd3', 'd2', 'd1 (+D1 store issued)',
           'stash stores issued']
    fb_total = t[11:20].sum() + t[1]
    g_total = t[30:37].sum() + t[3]
    total = t[0] + fb_total + t[2] + g_total + t[4]
    print('D=%d: %.2f us/step wall; ticks/step %.0f (%.2f GHz if ticks are '
          'core cycles)' % (d, dt / steps * 1e6, total,
                            total / (dt / steps * 1e9)))
    print('   %-36s %8.0f' % ('loop top (incl. stamp folding)', t[0]))
    print('   %-36s %8.0f' % ('FB', fb_total))
    for i in range(9):
        print('      fb %-34s %8.0f' % (fbn[i], t[11 + i]))
    print('      fb %-32s %8.0f' % ('return', t[1]))
    print('   %-36s %8.0f' % ('barrier 1 (+ adam_lr)', t[2]))
    print('   %-36s %8.0f' % ('G', g_total))
    for nm, k in [('job record + layer decode', 33),
                  ('operand loads issued', 30),
                  ('MFMA chains -> LDS', 31), ('LDS barrier', 36),
                  ('reduce + Adam + stores issued', 32)]:
        print('      g %-33s %8.0f' % (nm, t[k]))
    print('      g %-33s %8.0f' % ('return (+ loss fold elsewhere)', t[3]))
    print('   %-36s %8.0f' % ('barrier 2 + next rows prefetch', t[4]))
