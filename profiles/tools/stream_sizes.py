"""nb_ell_stream_kernel at one dimension for several point counts (VERDICT
r3 item 7: does the process-to-process spread at D = 100 come with the size
of the array?).  python profiles/tools/stream_sizes.py D [log2 n ...]
Reports GB/s against the 8 TB/s HBM peak and TFLOP/s (algorithmic D (D + 1)
flop per point) against the 78.6 TF fp64 MFMA peak: beyond n_dim = 78 the
kernel is bound by the matrix cores, not by HBM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nautilus_amd import bounds as nb  # noqa: E402

d = int(sys.argv[1])
sizes = [int(v) for v in sys.argv[2:]] or [22, 23, 24]
rng = np.random.default_rng(d)
a = rng.normal(size=(d, d))
cov = (a @ a.T / d + np.eye(d)) * 0.02
B = np.linalg.cholesky(cov)
ell = nb.Ellipsoid.from_params(0.5 * np.ones(d), B, np.linalg.inv(B),
                               np.linalg.inv(cov))
dev = ell.device_bound()
for lg in sizes:
    n = 1 << lg
    x = torch.rand((n, d), dtype=torch.float64, device='cuda')
    x[::2] = 0.5 + 0.6 * (x[::2] - 0.5)
    for _ in range(3):
        dev.contains_stream(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        mask = dev.contains_stream(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    gbs = n * (8 * d + 1) / dt / 1e9
    tf = n * d * (d + 1.0) / dt / 1e12
    print('D=%d n=2^%d (%.1f GB): %.3f ms  %.0f GB/s = %.3f of 8 TB/s  '
          '%.1f TFLOP/s = %.3f of 78.6  (address %#x)' % (
              d, lg, n * 8 * d / 1e9, dt * 1e3, gbs, gbs / 8000, tf,
              tf / 78.6, x.data_ptr()), flush=True)
    del x, mask
    torch.cuda.empty_cache()
