"""Phase cycle stamps of the MVEE sweep kernel (needs the debug library:
make debug DEFS=-DNB_MVEE_TIMING; NAUTILUS_HIP_LIB=.../libnautilus_hip_dbg.so)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nautilus_amd import device, _lib
import ctypes as C
lib = _lib.load()
rng = np.random.default_rng(0)
names = ['A0 stage', 'A1 merge1', 'A2 merge2', 'A3 gather', 'A4 W', 'A5 G0', 'A6 seq', 'A7 Y', 'A8 Pnew', 'B1 g', 'B2 sel1', 'B3 sel2']
for n, d in [(2000, 50), (10000, 100)]:
    pts = rng.normal(size=(n, d)) * rng.uniform(0.5, 2.0, size=d) + 0.3
    x = torch.from_numpy(pts).cuda()
    u = torch.empty(n, dtype=torch.float64, device='cuda')
    work = torch.zeros(lib.nb_mvee_weights_work_doubles(n, d, 20), dtype=torch.float64, device='cuda')
    _lib.check(lib.nb_mvee_weights(C.c_void_p(x.data_ptr()), n, d, 100, 20, C.c_void_p(u.data_ptr()), C.c_void_p(work.data_ptr()), None))
    torch.cuda.synchronize()
    m = d + 1
    off = 2 * 128 + ((n * d + 1) & ~1) + 2 * m * m + 2 + 2 * 32 * 20 + (2 * 32 * 21 + 1) // 2 + 1
    st = work[off:off + 32].cpu().numpy()
    print('n=%d d=%d scale=%g accepted=%g' % (n, d, st[0], st[1]))
    t = st[8:21]
    for k in range(12):
        print('  %-10s %8.0f cycles' % (names[k], t[k + 1] - t[k]))
    print('  total      %8.0f' % (t[12] - t[0]))
