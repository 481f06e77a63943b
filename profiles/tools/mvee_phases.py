"""Phase cycle stamps of the MVEE sweep kernel (needs the debug library:
make debug DEFS=-DNB_MVEE_TIMING; NAUTILUS_HIP_LIB=.../libnautilus_hip_dbg.so):
call 5 of a fit, workgroup 0, plus the wall time of whole fits (HIP events)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nautilus_amd import device, _lib
import ctypes as C
lib = _lib.load()
rng = np.random.default_rng(0)
names = ['A0 stage', 'A1 merge1', 'A2 merge2', 'A3 gather', 'A4 W', 'A5 G0', 'A6 seq', 'A7 Y', 'A8 Pnew', 'B1 g', 'B2 sel1', 'B3 sel2']
shapes = [(2000, 50, 1), (2000, 50, 2), (1000, 50, 2), (500, 20, 2), (10000, 100, 1)]
for n, d, nb in shapes:
    xs, us = [], []
    for b in range(nb):
        pts = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, size=d) + 0.3
        xw, _ = device.whiten(torch.from_numpy(pts).cuda())
        xs.append(xw)
        us.append(torch.empty(n, dtype=torch.float64, device='cuda'))
    n_arr = (C.c_int64 * nb)(*[n] * nb)
    work = torch.zeros(int(lib.nb_mvee_work_doubles(nb, n, d, 20)), dtype=torch.float64, device='cuda')

    def fit():
        _lib.check(lib.nb_mvee_khachiyan(
            nb, (C.c_void_p * nb)(*[x.data_ptr() for x in xs]), n_arr, d, 100, 20,
            (C.c_void_p * nb)(*[u.data_ptr() for u in us]), C.c_void_p(work.data_ptr()), None))
    fit()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fit()
    e1.record()
    torch.cuda.synchronize()
    m = d + 1
    off = 2 * m * m + 2 + 2 * 32 * 20 + (2 * 32 * 21 + 1) // 2 + 1
    st = work[off:off + 32].cpu().numpy()
    print('n=%d d=%d sets=%d: %.3f ms per fit (101 sweep launches), scale=%g accepted=%g' % (
        n, d, nb, e0.elapsed_time(e1) / 10, st[0], st[1]))
    t = st[8:21]
    if t[12] > t[0] > 0:
        for k in range(12):
            print('  %-10s %8.0f cycles' % (names[k], t[k + 1] - t[k]))
        print('  total      %8.0f' % (t[12] - t[0]))
