"""Worker of the sharded-run tests: one rank of a whole ``Sampler.run()`` with
every batch of every phase (exploration, pre-fill of new bounds, sampling) and
the emulator ensembles spread over the ranks.  Launched by
``torch.distributed.run``; rank 0 prints one JSON line."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def host_like(x):
    # normalised 4-D Gaussian: analytic log Z = 0
    return (-0.5 * np.sum(((x - 0.5) / 0.1)**2, axis=-1) -
            x.shape[-1] * np.log(0.1 * np.sqrt(2 * np.pi)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--host-likelihood', action='store_true')
    ap.add_argument('--blobs', action='store_true')
    ap.add_argument('--n-networks', type=int, default=2)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    os.environ.setdefault('NB_TRAIN_TWO_LAUNCH', '1')   # ranks share one GPU
    os.environ.setdefault('NB_GMM_MAX_WGS', '1')
    dist.init_process_group('gloo')
    from nautilus_amd import GaussianLikelihood, Sampler, unit_prior
    from nautilus_amd.parallel import ShardedComm
    comm = ShardedComm()
    d = 4
    # the construction of a bound, sharded (networks dealt out over the ranks,
    # weights exchanged) against the same construction on this rank alone:
    # identical parameters, bit for bit (neural.py:93-96)
    from nautilus_amd.bounds import NautilusBound
    rng = np.random.default_rng(5)
    cloud = 0.5 + 0.12 * rng.normal(size=(1200, d))
    cloud_l = -np.sum((cloud - 0.5)**2, axis=1)

    def build(c):
        return NautilusBound.compute(
            cloud, cloud_l, np.median(cloud_l), np.log(0.05), n_networks=2,
            rng=np.random.default_rng(9), comm=c)
    sharded, alone = build(comm), build(None)

    def params(b):
        out = []
        for nb in b.neural_bounds:
            out += [nb.outer_bound.c, nb.outer_bound.B, [nb.score_predict_min]]
            for net in nb.emulator.neural_networks:
                out += [np.ravel(w) for w in net.coefs_ + net.intercepts_]
        return np.concatenate([np.ravel(np.asarray(v, float)) for v in out])
    construction_identical = bool(np.array_equal(params(sharded),
                                                 params(alone)))
    if args.host_likelihood:
        prior, like = (lambda u: u), host_like
    else:
        prior, like = unit_prior, GaussianLikelihood(np.full(d, 0.5),
                                                     np.eye(d) * 0.01)
    if args.blobs:
        # two blobs per point that say which point they belong to
        plain = like
        if args.host_likelihood:
            def like(x):
                return (plain(x), x[:, 0].astype(np.float32),
                        (1000 * x[:, 1]).astype(np.int16))
        else:
            def like(x):
                return (plain(x), x[:, 0].to(torch.float32),
                        (1000 * x[:, 1]).to(torch.int16))
            like.device = True
    s = Sampler(prior, like, n_dim=d, n_live=400, n_networks=args.n_networks,
                vectorized=True, seed=args.seed, n_batch=400,
                comm=comm)
    ok = s.run(n_eff=3000, discard_exploration=True, timeout=500)
    # every rank must hold the same state
    state = np.concatenate([s.shell_n_sample, s.shell_n, [s.n_like],
                            np.concatenate(s.log_l)])
    digest = hashlib.sha1(np.ascontiguousarray(state).tobytes()).hexdigest()
    comm.assert_identical([float(int(digest[:12], 16)), s.log_z, s.n_eff],
                          'cuda', 'sampler state')
    blobs_follow = None
    if args.blobs:
        pts, _, _, blobs = s.posterior(return_blobs=True)
        blobs_follow = bool(
            len(pts) == len(blobs) and
            np.all(pts[:, 0].astype(np.float32) == blobs['blob_0']) and
            np.all((1000 * pts[:, 1]).astype(np.int16) == blobs['blob_1']))
        n_blob = sum(len(b) for b in s.blobs)
        comm.assert_identical([float(n_blob), float(blobs_follow)], 'cuda',
                              'blobs')
        blobs_follow = blobs_follow and n_blob == sum(
            len(v) for v in s.log_l)
    if comm.rank == 0:
        nets = [n for b in s.bounds[1:] for nbd in b.neural_bounds
                if nbd.emulator is not None
                for n in nbd.emulator.neural_networks]
        print(json.dumps(dict(
            ok=bool(ok), world=comm.world, log_z=float(s.log_z),
            n_eff=float(s.n_eff), n_like=int(s.n_like),
            n_bounds=len(s.bounds),
            shell_n_sample=[int(v) for v in s.shell_n_sample],
            shell_n=[int(v) for v in s.shell_n],
            n_networks=len(nets),
            construction_identical=construction_identical,
            blobs_follow=blobs_follow)), flush=True)
    comm.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
