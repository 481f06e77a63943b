import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(R,'tests')); sys.path.insert(0,R)
import numpy as np, torch
from helpers import upload
from oracle import bounds_oracle as bo
def timeit(fn, n=10, warm=2):
    # median of synchronised calls (and the slowest: a process that ran
    # kernels of another n_dim before sees single calls stall for tens of
    # milliseconds -- profiles/r05/slow_mode_probe.txt -- which a mean hides)
    for _ in range(warm): fn()
    each = []
    for _ in range(n):
        torch.cuda.synchronize(); t=time.perf_counter()
        fn()
        torch.cuda.synchronize(); each.append(time.perf_counter()-t)
    timeit.slowest = max(each)
    return float(np.median(each))
for d in ([int(v) for v in sys.argv[1:]] or (20, 49, 50, 64, 100)):
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d,d)); cov = A@A.T/d + np.eye(d); B = np.linalg.cholesky(cov*0.02)
    ell = bo.OEllipsoid.from_params(0.5*np.ones(d), B)
    b = upload(ell)
    n = int(os.environ.get('NB_STREAM_N', 1 << 24))
    x = torch.rand((n,d), dtype=torch.float64, device='cuda')
    x[::2] = 0.5 + 0.6*(x[::2]-0.5)
    m1 = b.contains_stream(x); m2 = b.contains(x[:200000])
    t1 = timeit(lambda: b.contains_stream(x))
    gb = n*(8*d+1)/1e9
    print('D=%d stream: %.3f ms %.1f GB/s (%.1f%% of 8TB/s) %.2f Gpt/s  mismatch vs mfma %d inside %.3f, slowest call %.3f ms' % (d, t1*1e3, gb/t1, gb/t1/80, n/t1/1e9, int((m1[:200000]!=m2).sum()), float(m1.double().mean()), timeit.slowest*1e3))
