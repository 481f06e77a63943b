"""World-size-2 test of the sharding collectives on the CPU (gloo)."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nautilus_amd import parallel
    comm = parallel.ShardedComm()
    assert (comm.rank, comm.world) == (rank, world)

    # a "batch": each rank contributes n_local rows drawn from its own stream
    n_local, d = parallel.split_batch(8, world), 3
    rng = np.random.default_rng(parallel.rank_key(1234, rank))
    rows = torch.from_numpy(rng.random((n_local, d)))
    log_l = torch.from_numpy(-rng.random(n_local))
    counts = [10 + rank, 100 * (rank + 1), 7, 0, 1]
    all_rows, all_ll, totals = parallel.shard_shell_batch(comm, rows, log_l,
                                                          counts)
    assert all_rows.shape == (8, d) and all_ll.shape == (8,)
    # rank order is preserved and every rank sees the same result
    assert torch.equal(all_rows[rank * n_local:(rank + 1) * n_local], rows)
    assert torch.equal(all_ll[rank * n_local:(rank + 1) * n_local], log_l)
    assert totals == [21, 300, 14, 0, 2]
    comm.assert_identical([float(all_rows.sum()), float(all_ll.sum())],
                          'cpu', 'gathered batch')
    assert comm.max_float(float(rank), 'cpu') == world - 1
    with pytest.raises(RuntimeError):
        comm.assert_identical([float(rank)], 'cpu')
    with pytest.raises(ValueError):
        parallel.split_batch(7, world)
    # zero-padded rows summed over the ranks = bit-exact gather (the weight
    # exchange of the sharded ensemble training)
    blob = torch.zeros((4, 5), dtype=torch.float64)
    mine = [g for g in range(4) if g % world == rank]
    for g in mine:
        blob[g] = torch.from_numpy(np.random.default_rng(g).normal(size=5))
    total = comm.sum_rows(blob)
    for g in range(4):
        assert np.array_equal(total[g].numpy(),
                              np.random.default_rng(g).normal(size=5))
    # gathers in flight land in the order they were issued (the points of
    # the sharded sampling phase), interleaved with a synchronous collective
    h1 = comm.gather_rows_async(rows)
    h2 = comm.gather_rows_async(2.0 * rows)
    again = comm.gather_rows(log_l[:, None])[:, 0]
    assert torch.equal(again, all_ll)
    assert torch.equal(h1.wait(), all_rows)
    assert torch.equal(h2.wait(), 2.0 * all_rows)
    assert torch.equal(h1.wait(), all_rows)          # idempotent
    # blobs of a sharded batch follow their points (sampler.py:875-904 for
    # the layouts): rank-major like gather_rows, or interleaved like the host
    # likelihood of a batch every rank holds
    from types import SimpleNamespace
    from nautilus_amd.sampler import Sampler
    for dtype in (np.float32, [('a', '|S10'), ('b', np.int16)], (np.int64, 3),
                  (np.float64, (2, 3))):
        dt = np.dtype(dtype)
        # (one sampler has one blob layout: the row shape is agreed on once)
        me = SimpleNamespace(comm=comm)

        def blob_of(i):
            if dt.names:
                return (str(i).encode(), i)
            return np.full(dt.shape, i) if dt.shape else i
        base = dt.base if dt.shape else dt
        mine_rows = range(rank * n_local, (rank + 1) * n_local)
        local = np.array([blob_of(i) for i in mine_rows], dtype=base)
        got = Sampler._gather_blobs(me, np.squeeze(local), n_local)
        want = np.array([blob_of(i) for i in range(world * n_local)],
                        dtype=base)
        assert got.dtype == want.dtype and np.array_equal(got, want)
        n = 7                                        # not a multiple of world
        per = -(-n // world)
        local = np.array([blob_of(i) for i in range(rank, n, world)],
                         dtype=base)
        got = Sampler._gather_blobs(me, np.squeeze(local), per,
                                    interleaved_to=n)
        want = np.array([blob_of(i) for i in range(n)], dtype=base)
        assert got.dtype == want.dtype and np.array_equal(got, want)
        # a share of ONE row (squeezed: its row axis is gone) on the last
        # rank, several on the others: every rank ends with the same shape
        n = world + 1
        per = -(-n // world)
        local = np.array([blob_of(i) for i in range(rank, n, world)],
                         dtype=base)
        got = Sampler._gather_blobs(SimpleNamespace(comm=comm),
                                    np.squeeze(local), per, interleaved_to=n)
        want = np.array([blob_of(i) for i in range(n)], dtype=base)
        assert got.shape == want.shape and np.array_equal(got, want)
    # collective stop decision: true everywhere if true anywhere
    assert comm.any_flag(rank == 1) is True
    assert comm.any_flag(False) is False
    comm.barrier()
    np.save(os.path.join(out_dir, 'rows_%d.npy' % rank), all_rows.numpy())
    dist.destroy_process_group()


def test_sharded_batch_two_ranks(tmp_path):
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / 'rows_0.npy')
    b = np.load(tmp_path / 'rows_1.npy')
    assert np.array_equal(a, b)
    # different ranks drew different points
    assert not np.array_equal(a[:4], a[4:])


def test_rank_keys_are_distinct():
    from nautilus_amd import parallel
    keys = {parallel.rank_key(987654321, r) for r in range(8)}
    assert len(keys) == 8
    assert parallel.rank_key(987654321, 0) == 987654321
    assert all(0 <= k < 2**63 for k in keys)


def test_network_weight_rows_round_trip():
    """The rows exchanged by ``train_ensembles_sharded`` hold a network
    completely (host logic, no GPU)."""
    from nautilus_amd import emulator
    rs = np.random.RandomState(3)
    coefs, intercepts = emulator._glorot(7, rs)
    net = emulator.Network(coefs, intercepts, 41, [0.5, 0.25, 0.125])
    row = emulator._pack_network(net, 7)
    assert row.shape == (2 + sum(c.size for c in coefs) +
                         sum(b.size for b in intercepts),)
    back = emulator._unpack_network(row, 7)
    assert back.n_iter_ == 41 and back.loss_curve_ == [0.125]
    for a, b in zip(coefs + intercepts, back.coefs_ + back.intercepts_):
        assert np.array_equal(a, b)
