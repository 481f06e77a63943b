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
    v = np.array(list(buf))[:8]
    fb = np.array(list(buf))[10:22]
    fbn = ['weights issue + sA0 zero', 'input fill', 'L1', 'L2', 'L3', 'out + d4', 'd3', 'd2', 'd1 (+D1 store)', '-']
    names = ['loop top', 'FB (gather + fwd/bwd + stash)', 'barrier 1', 'G (dW + Adam)', 'barrier 2 + prefetch']
    print('D=%d: %.2f us/step wall; ticks/step %.0f (= %.2f us at 2.4 GHz)' % (d, dt / steps * 1e6, v.sum() / steps, v.sum() / steps / 2400))
    for i in range(5):
        print('   %-32s %8.0f ticks/step' % (names[i], v[i] / steps))
    for i in range(10):
        print('      fb %-28s %8.0f' % (fbn[i], fb[i] / steps))
    gg = np.array(list(buf))[30:33]
    for nm, val in zip(['g: decode + loads issued', 'g: MFMA chain (waits for operands)', 'g: Adam + stores'], gg):
        print('      %-36s %8.0f' % (nm, val / steps))
