# dense score / accept timing of the pipelined eval kernel vs the general one
# usage: python profiles/tools/fast_time.py [D ...]   (NAUTILUS_HIP_LIB selects the library)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch, time
torch.manual_seed(0)
from oracle import mlp_oracle as mo
from nautilus_amd import device
dims = [int(v) for v in sys.argv[1:]] or [50]
for d in dims:
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d)); cov = A @ A.T / d + np.eye(d); B = np.linalg.cholesky(cov * 0.02)
    nets = [mo.glorot_init(d, e)[:2] for e in range(4)]
    ell = device.member(0.5 * np.ones(d), B)
    nbd = device.DeviceBound(d, [ell], None, False, [dict(ellipsoid=ell, score_predict_min=0.0, mlp=dict(mean=np.zeros(d), scale=np.ones(d), nets=nets))])
    n = 1 << 20
    x = torch.rand((n, d), dtype=torch.float64, device='cuda')
    out = None
    for _ in range(3): out = nbd.neural_score(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): nbd.neural_score(x)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    flop = n * (d * (d + 1) + 4 * 2 * (100 * d + 6020))
    print('D=%d score: %.3f ms  %.1f TFLOP/s algorithmic (%.3f of 78.6)  checksum %.12e' % (d, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6, float(out[1].sum())))
    # proposals of the bound itself
    xs = nbd.propose(7, 0, n) if hasattr(nbd, 'propose') else x
    for _ in range(2): fl = nbd.accept(7, 0, xs)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10): fl = nbd.accept(7, 0, xs)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    f = fl.cpu().numpy()
    print('D=%d accept: %.3f ms  flags1 %d flags3 %d' % (d, ms, int((f & 1).sum()), int((f == 3).sum())))
