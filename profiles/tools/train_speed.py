"""Emulator training speed (nb_mlp_train.hip + the host's shuffle streams):
microseconds per Adam step of the slowest network of an ensemble, (a)
differenced (a long minus a short run: the kernel alone) and (b) over a whole
fit as the sampler sees it (host start-up, shuffles, uploads, status reads
included).  python profiles/tools/train_speed.py [--big]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch  # noqa: E402
from nautilus_amd import emulator  # noqa: E402

CASES = [(50, 24000, 4), (20, 8000, 4), (100, 24000, 8), (30, 30000, 4),
         (50, 2000, 4)]
if '--big' in sys.argv:
    # config-5 size: 8 networks x 2 x 10^5 rows (the host's shuffles used to
    # take longer than the GPU's epochs there)
    CASES = [(100, 200000, 8), (50, 180000, 4)]


def fit(x, y, e, n_epochs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs,
                            hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize()
    return time.perf_counter() - t


for d, n_row, e in CASES:
    x = torch.randn((n_row, d), dtype=torch.float64, device='cuda')
    y = torch.rand(n_row, dtype=torch.float64, device='cuda')
    fit(x, y, e, 2)
    steps = (n_row + 199) // 200
    short, long_ = (16, 64) if n_row >= 100000 else (32, 160)
    t_s, t_l = fit(x, y, e, short), fit(x, y, e, long_)
    print('D=%d E=%d n=%d: %.2f us/step (differenced), %.2f us/step over a '
          'whole fit of %d epochs' % (
              d, e, n_row, (t_l - t_s) / ((long_ - short) * steps) * 1e6,
              t_l / (long_ * steps) * 1e6, long_), flush=True)
