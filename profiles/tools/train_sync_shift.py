"""Which part of the trainer's block makes the step time follow address bit
13: the barrier counters or the pool?  (NB_TRAIN_BLOCK_SHIFT moves the whole
block inside an allocation of fixed size, NB_TRAIN_SYNC_SHIFT the counters
alone.)  Decisive on the tree before the per-XCD arena of barrier records
(profiles/r04/second_session/train_sync_shift_before_arena.txt); since then
the counters live in the arena, NB_TRAIN_SYNC_SHIFT only moves the error
mirror, and every line of this tool reads the same (..._with_arena.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nautilus_amd import emulator
def fit(x, y, e, n_epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    emulator.train_networks(x, y, list(range(e)), max_epochs=n_epochs, hparams=dict(n_iter_no_change=100000))
    torch.cuda.synchronize(); return time.perf_counter() - t
d, n_row, e = 50, 24000, 4
x = torch.randn((n_row, d), dtype=torch.float64, device='cuda'); y = torch.rand(n_row, dtype=torch.float64, device='cuda')
os.environ['NB_TRAIN_BLOCK_SHIFT'] = '0'
fit(x, y, e, 2)
for bs, ss in [(0, None), (8, None), (0, 0), (0, 8), (8, 0), (8, 8), (16, None), (24, None), (0, None), (8, None)]:
    os.environ['NB_TRAIN_BLOCK_SHIFT'] = str(bs)
    if ss is None: os.environ.pop('NB_TRAIN_SYNC_SHIFT', None)
    else: os.environ['NB_TRAIN_SYNC_SHIFT'] = str(ss)
    t_s, t_l = fit(x, y, e, 32), fit(x, y, e, 96)
    print('block +%2d KB, counters at %-9s %.2f' % (bs, 'start:' if ss is None else '16+%d KB:' % ss, (t_l - t_s) / (64 * 120) * 1e6), flush=True)
