"""Where does the time of the D = 100 staged acceptance go in a process that
ran the D = 50 one before (the "slow mode" of rounds 3-5)?  Per-call wall
times with a synchronisation after EVERY call, under a few variations.

    python profiles/tools/slow_mode_probe.py MODE
      plain      D = 50 calls, then D = 100 calls
      alone      D = 100 only
      trim       D = 50, free its tensors + torch.cuda.empty_cache(), D = 100
      half       D = 50, then D = 100 on 2^19 proposals
      reverse    D = 100, then D = 50, then D = 100 again
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
sys.argv, mode = sys.argv[:1] + ['0'], sys.argv[1]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location(
    'accept_bench_defs', os.path.join(os.path.dirname(__file__),
                                      'accept_bench.py'))
src = open(spec.origin).read().split('# arguments: D or D:K:M')[0]
defs = {'__file__': spec.origin, '__name__': 'accept_bench_defs'}
exec(compile(src, spec.origin, 'exec'), defs)
build = defs['build']


def run(d, n, tag, reps=6):
    bound = build(d, 4, 4)
    dev = bound.device_bound()
    x = dev.propose(7, 0, n)
    dev.accept(7, 0, x)
    torch.cuda.synchronize()
    times = []
    for r in range(reps):
        t0 = time.perf_counter()
        flags = dev.accept(7, 0, x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        times.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    # the same calls queued back to back, one synchronisation at the end
    t0 = time.perf_counter()
    for r in range(reps):
        flags = dev.accept(7, 0, x)
    torch.cuda.synchronize()
    queued = 1e3 * (time.perf_counter() - t0) / reps
    print('%s D=%d n=%d: launch + sync per call (ms): %s | queued: %.2f ms '
          'per call' % (tag, d, n, ' '.join('%.2f+%.2f' % t for t in times),
                        queued), flush=True)
    return bound, dev, x, flags


N = 1 << 20
if mode == 'alone':
    run(100, N, mode)
elif mode == 'plain':
    keep = run(50, N, mode)
    run(100, N, mode)
elif mode == 'trim':
    keep = run(50, N, mode)
    del keep
    from nautilus_amd import device
    device._SCRATCH.clear()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    run(100, N, mode)
elif mode == 'half':
    keep = run(50, N, mode)
    run(100, N // 2, mode)
elif mode == 'reverse':
    a = run(100, N, mode)
    b = run(50, N, mode)
    c = run(100, N, mode)
print('%s: torch allocated %.0f MB reserved %.0f MB' % (
    mode, torch.cuda.memory_allocated() / 2**20,
    torch.cuda.memory_reserved() / 2**20))
