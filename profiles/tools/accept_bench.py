"""Acceptance of a multi-modal bound (K = 4 outer members, M = 4 neural bounds,
E = 4 networks each) through the staged route (nb_accept_staged: geometric
kernel + candidate lists + ONE batched emulator launch, no host round trip).
Reports the rate of the whole call and of the emulator stage alone against
the fp64 MFMA peak, and the HIP-event time of the stages."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
from nautilus_amd import bounds as nb, device  # noqa: E402
from nautilus_amd.emulator import NeuralNetworkEmulator, Network  # noqa: E402

PEAK = 78.6


def glorot(rs, d):
    units = [d, 100, 50, 20, 1]
    coefs, icpts = [], []
    for a, b in zip(units[:-1], units[1:]):
        lim = np.sqrt(6.0 / (a + b))
        coefs.append(rs.uniform(-lim, lim, (a, b)))
        icpts.append(rs.uniform(-lim, lim, b))
    return Network(coefs, icpts)


def build(d, k, e, seed=0, k_members=None, m_neural=None):
    rs = np.random.RandomState(seed)
    centres = 0.25 + 0.5 * rs.rand(k, d)
    members, neural = [], []
    for c in centres:
        a = rs.normal(size=(d, d)) * 0.02 / np.sqrt(d) + 0.04 * np.eye(d)
        cov = a @ a.T
        B = np.linalg.cholesky(cov)
        B_inv = np.linalg.inv(B)
        A = np.linalg.inv(cov)
        ell = nb.Ellipsoid.from_params(c, B, B_inv, A)
        members.append(ell)
        emu = NeuralNetworkEmulator.from_weights(
            np.zeros(d), np.ones(d), [glorot(rs, d) for _ in range(e)])
        neural.append(nb.NeuralBound.from_parts(
            nb.Ellipsoid.from_params(c, B, B_inv, A), emu, 0.0))
    members = members[:k_members or k]
    neural = neural[:m_neural or k]
    outer = nb.Union.from_members(members, unit=True)
    outer.log_v_all = np.array([m.log_v for m in members])
    return nb.NautilusBound.from_parts(outer, neural,
                                       rng=np.random.default_rng(1))


# arguments: D or D:K:M (default 50 100, K = M = 4); one process per
# dimension gives the cleanest numbers (r03_kernels_prof.sh)
CASES = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or \
    [(50,), (100,)]
for case in CASES:
    d = case[0]
    k, m = (case[1], case[2]) if len(case) == 3 else (4, 4)
    e = 4
    bound = build(d, max(k, m), e, k_members=k, m_neural=m)
    dev = bound.device_bound()
    n = 1 << 20
    x = dev.propose(7, 0, n)
    dev.accept(7, 0, x)                      # warm-up
    reps = int(os.environ.get('NB_ACCEPT_REPS', 20))
    with device.EvalCounters() as counters:
        torch.cuda.synchronize()
        prof = None
        if os.environ.get('NB_ACCEPT_CPROFILE'):
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        # every call timed on its own (synchronised): the median is the
        # kernels' time; a process that ran kernels of another n_dim before
        # sees single calls stall for 20-80 ms (the device drops its clocks
        # and ramps back over the next calls, profiles/r05/
        # slow_mode_probe.txt) -- averaged over five queued calls that was
        # the "slow mode" of rounds 3-4
        each = []
        for r in range(reps):
            t0 = time.perf_counter()
            flags = dev.accept(7, 0, x)
            torch.cuda.synchronize()
            each.append(time.perf_counter() - t0)
        dt = float(np.median(each))
        # ... and queued back to back behind one synchronisation
        t0 = time.perf_counter()
        for r in range(reps):
            flags = dev.accept(7, 0, x)
        torch.cuda.synchronize()
        dt_queued = (time.perf_counter() - t0) / reps
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats('tottime').print_stats(12)
        work = counters.read()
    reps_counted = 2 * reps
    flops = ((work['outer_point_evals'] + work['ellipsoid_point_evals']) *
             d * (d + 1) + work['emulator_point_evals'] * 2.0 *
             (100 * d + 6020)) / reps_counted
    kept = float((flags & 2).bool().double().mean())
    print('D=%d K=%d M=%d E=%d, %d proposals: %.2f ms per call (median of %d '
          'synchronised calls; slowest %.2f, queued mean %.2f), %.1f TFLOP/s '
          '= %.3f of the fp64 MFMA peak (accepted %.3f, emulator evaluations '
          'per proposal %.2f)' % (
              d, k, m, e, n, dt * 1e3, reps, max(each) * 1e3, dt_queued * 1e3,
              flops / dt / 1e12, flops / dt / 1e12 / PEAK, kept,
              work['emulator_point_evals'] / reps_counted / n / e),
          flush=True)
    # the emulator stage alone: all proposals of one mode, gathered
    idx = torch.arange(n, device='cuda')
    out = torch.empty((n, 2), dtype=torch.float64, device='cuda')
    from nautilus_amd import _lib
    lib = _lib.load()
    for r in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.nb_neural_score_rows(
            dev._h, 1, 0, x.data_ptr(), idx.data_ptr(), n, out.data_ptr(),
            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    f2 = n * (d * (d + 1) + e * 2.0 * (100 * d + 6020))
    for r in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.nb_neural_score_rows(
            dev._h, 1, 0, x.data_ptr(), None, n, out.data_ptr(),
            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        dt0 = time.perf_counter() - t0
    print('   the same rows without an index list: %.2f ms, %.1f TFLOP/s = '
          '%.3f' % (dt0 * 1e3, f2 / dt0 / 1e12, f2 / dt0 / 1e12 / PEAK),
          flush=True)
    print('   gathered emulator scores (neural bound 1 of 4, %d rows by '
          'index): %.2f ms, %.1f TFLOP/s = %.3f' % (
              n, dt * 1e3, f2 / dt / 1e12, f2 / dt / 1e12 / PEAK), flush=True)
