"""Handles of bounds living in HBM and the launch wrappers around the C ABI.

torch is used for device memory and streams only; every computation on the
path runs in the hand-written HIP kernels of ``libnautilus_hip.so``.
"""

import ctypes as C
import math
import os
import time

import numpy as np
import torch
from scipy.linalg import solve_triangular

from . import _lib


# (the raw handle of torch's current stream: torch.cuda.current_stream()
# builds a Stream object, ~10 us, and the hot loop asks ~10 times a step.  The
# two accessors are private torch API -- resolved once here, with the public
# route for a torch build that lacks them)
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_device = torch._C._cuda_getDevice
except AttributeError:                                 # pragma: no cover
    _raw_stream = None


def _stream():
    if _raw_stream is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(_raw_stream(_cur_device()))


def _itemsize(dtype):
    try:
        return dtype.itemsize                          # torch >= 2.1
    except AttributeError:                             # pragma: no cover
        return torch.empty((), dtype=dtype).element_size()


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


def as_device_points(x, n_dim=None):
    """numpy / torch (n, D) -> contiguous float64 cuda tensor."""
    if isinstance(x, torch.Tensor):
        t = x.to(device='cuda', dtype=torch.float64)
    else:
        t = torch.from_numpy(_f64(x)).cuda()
    if t.dim() == 1:
        t = t.unsqueeze(0)
    t = t.contiguous()
    if n_dim is not None and t.shape[1] != n_dim:
        raise ValueError('points have %d columns, bound has n_dim=%d' %
                         (t.shape[1], n_dim))
    return t


def member(c=None, B=None, B_inv=None, idx_ell=None, free_dims=False):
    """Description of one union member (Ellipsoid / Mixture / cube)."""
    if c is None:
        return dict(n_ell=0, c=np.zeros(0), B=np.zeros((0, 0)),
                    B_inv=np.zeros((0, 0)), idx_ell=None, free_dims=free_dims)
    c = _f64(c)
    B = _f64(B)
    B_inv = _f64(np.linalg.inv(B) if B_inv is None else B_inv)
    # exact zeros above the diagonal (the kernels exploit the triangle)
    B_inv = np.tril(B_inv)
    return dict(n_ell=len(c), c=c, B=np.tril(B), B_inv=B_inv,
                idx_ell=None if idx_ell is None else
                np.ascontiguousarray(idx_ell, dtype=np.int32),
                free_dims=free_dims)


class DeviceBound:
    """One bound (any type of the reference) uploaded to HBM."""

    def __init__(self, n_dim, members=(), log_v_all=None, unit_cube=False,
                 neural=(), shift=None):
        """``shift``: optional (periodic indices, centers) of the bound's
        PhaseShift (reference bounds/nautilus.py:91-96)."""
        lib = _lib.load()
        self.n_dim = int(n_dim)
        self.n_members = len(members)
        self.n_neural = len(neural)
        keep = []          # keep numpy buffers alive during the call

        def fill_member(md, src):
            md.n_ell = src['n_ell']
            md.free_dims = 1 if src.get('free_dims') else 0
            if src['idx_ell'] is not None:
                keep.append(src['idx_ell'])
                md.idx_ell = src['idx_ell'].ctypes.data_as(_lib.c_int32_p)
            else:
                md.idx_ell = None
            for key in ('c', 'B', 'B_inv'):
                arr = _f64(src[key])
                keep.append(arr)
                setattr(md, key, _dp(arr))

        m_arr = (_lib.MemberDesc * max(1, self.n_members))()
        for md, src in zip(m_arr, members):
            fill_member(md, src)
        n_arr = (_lib.NeuralDesc * max(1, self.n_neural))()
        self.n_networks = 0
        # emulator threshold of every neural bound (bounds/neural.py:125), as
        # the blob holds it
        self.thresholds = [float(src.get('score_predict_min', 0.0)) - 1e-9
                           for src in neural]
        self.dense_need = None     # share of proposals that reach an emulator
        for nd, src in zip(n_arr, neural):
            fill_member(nd.ellipsoid, src['ellipsoid'])
            nd.score_predict_min = float(src.get('score_predict_min', 0.0))
            # largest squared semi-axis (exact, with a safety margin)
            nd.radius2 = float(np.linalg.norm(src['ellipsoid']['B'], 2)**2 *
                               (1.0 + 1e-9))
            mlp = src.get('mlp')
            if mlp is None:
                nd.mlp = None
                continue
            e = len(mlp['nets'])
            self.n_networks = e
            md = _lib.MlpDesc()
            md.n_networks = e
            mean, scale = _f64(mlp['mean']), _f64(mlp['scale'])
            keep += [mean, scale]
            md.mean, md.scale = _dp(mean), _dp(scale)
            cp = (_lib.c_double_p * (4 * e))()
            ip = (_lib.c_double_p * (4 * e))()
            from .emulator import pad_network
            for i, (coefs, intercepts) in enumerate(mlp['nets']):
                coefs, intercepts = pad_network(coefs, intercepts, self.n_dim)
                for k in range(4):
                    w, b = _f64(coefs[k]), _f64(intercepts[k])
                    keep += [w, b]
                    cp[4 * i + k], ip[4 * i + k] = _dp(w), _dp(b)
            md.coefs, md.intercepts = cp, ip
            keep += [md, cp, ip]
            nd.mlp = C.pointer(md)

        desc = _lib.BoundDesc()
        desc.n_dim = self.n_dim
        desc.n_members = self.n_members
        desc.members = m_arr
        lv = _f64(np.zeros(self.n_members) if log_v_all is None
                  else log_v_all)
        desc.log_v_all = _dp(lv)
        desc.unit_cube = 1 if unit_cube else 0
        desc.n_neural = self.n_neural
        desc.neural = n_arr
        desc.n_periodic = 0
        if shift is not None and len(shift[0]) > 0:
            per = np.ascontiguousarray(shift[0], dtype=np.int32)
            cen = _f64(shift[1])
            keep += [per, cen]
            desc.n_periodic = len(per)
            desc.periodic = per.ctypes.data_as(_lib.c_int32_p)
            desc.centers = _dp(cen)
        handle = C.c_void_p()
        _lib.check(lib.nb_bound_create(C.byref(desc), C.byref(handle)))
        self._h = handle
        self._lib = lib
        del keep

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.nb_bound_destroy(h)
            self._h = None

    @property
    def nbytes(self):
        return self._lib.nb_bound_nbytes(self._h)

    # -- queries ---------------------------------------------------------
    def contains(self, x):
        x = as_device_points(x, self.n_dim)
        return (self._self_list().eval(x, GEOM_ANY)[0] & GS_INSIDE) != 0

    def _self_list(self):
        lst = self.__dict__.get('_list')
        if lst is None:
            lst = self._list = DeviceBoundList([self])
        return lst

    def contains_stream(self, x):
        x = as_device_points(x, self.n_dim)
        mask = torch.empty(x.shape[0], dtype=torch.uint8, device='cuda')
        _lib.check(self._lib.nb_ellipsoid_contains_stream(
            self._h, _ptr(x), x.shape[0], _ptr(mask), _stream()))
        return mask.bool()

    def transform(self, x):
        """B_inv (x - c) of an Ellipsoid / NeuralBound ellipsoid (reference
        basic.py:340) -- ``nb_ellipsoid_transform``."""
        x = as_device_points(x, self.n_dim)
        y = torch.empty_like(x)
        _lib.check(self._lib.nb_ellipsoid_transform(
            self._h, _ptr(x), x.shape[0], _ptr(y), _stream()))
        return y

    def member_count(self, x):
        x = as_device_points(x, self.n_dim)
        cnt = torch.empty(x.shape[0], dtype=torch.uint8, device='cuda')
        _lib.check(self._lib.nb_member_count(self._h, _ptr(x), x.shape[0],
                                             _ptr(cnt), _stream()))
        return cnt

    def neural_score(self, x):
        """(r2, score) of neural bound 0."""
        x = as_device_points(x, self.n_dim)
        out = torch.empty((x.shape[0], 2), dtype=torch.float64, device='cuda')
        _lib.check(self._lib.nb_neural_score(self._h, _ptr(x), x.shape[0],
                                             _ptr(out), _stream()))
        return out[:, 0], out[:, 1]

    def propose(self, seed, offset, n, reuse=False):
        x = _buffer('propose', (n, self.n_dim), torch.float64, reuse)
        _lib.check(self._lib.nb_propose(self._h, seed, offset, n, _ptr(x),
                                        _stream()))
        return x

    def accept(self, seed, offset, x, reuse=False):
        """Flags of ``sample`` (bit 0: kept by the outer union's acceptance
        draw, bit 1: accepted) for the proposals ``x`` of stream position
        ``offset``.  Two routes, same decisions: the fused kernel (cube test,
        acceptance draw, ellipsoid and emulators of ONE neural bound in one
        pass over dense tiles) where most proposals reach the emulator, and
        the staged route (``nb_accept_staged``: geometric stage, candidate
        lists and ONE batched emulator launch, all counts on the device) for
        several outer members or neural bounds -- and for bounds whose
        proposals mostly die in the geometric tests (a funnel's envelope
        sticks far out of the unit cube): there the emulators only see the
        survivors.  The route of a bound is chosen ONCE, from the share of
        its first launch's proposals that reached an emulator."""
        fused_ok = (self.n_neural == 1 and self.n_members <= 1 and
                    self.n_networks >= 1)
        if fused_ok and self.dense_need is not None and self.dense_need > 0.5:
            flags = _buffer('accept', (x.shape[0],), torch.uint8, reuse)
            _lib.check(self._lib.nb_accept(self._h, seed, offset, _ptr(x),
                                           x.shape[0], _ptr(flags),
                                           _stream()))
            DISPATCHES['nb_eval_fast_kernel'] += 1
            return flags
        n = x.shape[0]
        flags = _buffer('accept', (n,), torch.uint8, reuse)
        need = self._lib.nb_accept_staged_work_bytes(self._h, n)
        work = _buffer('staged_work', (need,), torch.uint8, True)
        off = C.c_int64(0)
        _lib.check(self._lib.nb_accept_staged(
            self._h, seed, offset, _ptr(x), n, _ptr(flags), _ptr(work), need,
            C.byref(off), _stream()))
        DISPATCHES['nb_cand_kernel'] += 1
        DISPATCHES['nb_eval_fast_kernel'] += 1 if self.n_networks else 0
        if fused_ok and self.dense_need is None:
            # (the one host read of this route: the first launch of a bound
            # that could also take the fused kernel)
            totals = work[off.value:off.value + 4 * self.n_neural].view(
                torch.int32)
            self.dense_need = float(totals.sum()) / max(1, n)
        return flags

    def sample_launch(self, seed, offset, n_draw, mask=2, reuse=False,
                      out_role='compact'):
        """One launch of the device ``sample`` pipeline: draw, accept,
        compact.  Returns (points, counters) with counters = int64 tensor
        [n kept by the outer union, n kept in total] still on the device.
        ``reuse=True`` (the bounds' refill loops): the launch works in the
        process-wide scratch buffers and the returned rows are only valid
        until the next such launch (the next launch with the same
        ``out_role``: a refill in flight keeps its rows in a role of its
        own)."""
        x = self.propose(seed, offset, n_draw, reuse)
        flags = self.accept(seed, offset, x, reuse)
        out, counts, _ = compact_rows(x, flags, mask, reuse=reuse,
                                      out_role=out_role)
        return out, counts


class DeviceBoundList:
    """Device array of bounds for multi-bound queries."""

    def __init__(self, bounds):
        lib = _lib.load()
        self.bounds = list(bounds)
        arr = (C.c_void_p * max(1, len(self.bounds)))(
            *[b._h for b in self.bounds])
        handle = C.c_void_p()
        _lib.check(lib.nb_boundlist_create(arr, len(self.bounds),
                                           C.byref(handle)))
        self._h = handle
        self._lib = lib
        self.n_dim = self.bounds[0].n_dim if self.bounds else None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.nb_boundlist_destroy(h)
            self._h = None

    def eval(self, x, mode, reuse=False):
        """(status bytes, first containing bound or None) of the rows of the
        cuda tensor ``x`` against the list -- ``nb_list_eval``: geometric
        stage, candidate lists, ONE batched emulator launch; nothing returns
        to the host in between.  Rows are processed in slabs whose work space
        stays below WORK_BYTES.  ``reuse``: the results live in the
        process-wide scratch buffers (valid until the next such call)."""
        n = x.shape[0]
        st = _buffer('list_status', (n,), torch.uint8, reuse)
        first = (_buffer('list_first', (n,), torch.int32, reuse)
                 if mode == GEOM_FIRST else None)
        if n == 0:
            return st, first
        if len(self.bounds) == 0:
            st.zero_()
            if first is not None:
                first.fill_(NO_BOUND)
            return st, first
        slab = n
        while slab > 4096 and self._lib.nb_list_eval_work_bytes(
                self._h, slab) > WORK_BYTES:
            slab = (slab + 1) // 2
        # (the work space is not monotone in the row count: the last, shorter
        # slab may need more than a full one)
        need = max(self._lib.nb_list_eval_work_bytes(self._h, slab),
                   self._lib.nb_list_eval_work_bytes(
                       self._h, n - (n - 1) // slab * slab))
        work = _buffer('staged_work', (need,), torch.uint8, True)
        for lo in range(0, n, slab):
            k = min(slab, n - lo)
            _lib.check(self._lib.nb_list_eval(
                self._h, mode, _ptr(x[lo:]), k, _ptr(st[lo:]),
                _ptr(first[lo:]) if first is not None else None, _ptr(work),
                need, _stream()))
            DISPATCHES['nb_cand_kernel'] += 1
            DISPATCHES['nb_eval_fast_kernel'] += 1
        return st, first

    def inside_flags(self, x, reuse=False):
        """Status bytes of the rows of ``x`` for the compaction kernels: bit
        GS_INSIDE is set where any bound of the list contains the row."""
        return self.eval(as_device_points(x, self.n_dim), GEOM_ANY, reuse)[0]

    def contains_any(self, x, as_flags=False):
        """mask[i] = any bound of the list contains x[i] (uint8 0 / 1 with
        ``as_flags``)."""
        x = as_device_points(x, self.n_dim)
        st, _ = self.eval(x, GEOM_ANY)
        inside = (st & GS_INSIDE) != 0
        return inside.to(torch.uint8) if as_flags else inside

    def first_containing(self, x):
        """Position of the first bound of the list that contains x[i], -1 if
        none does."""
        x = as_device_points(x, self.n_dim)
        _, first = self.eval(x, GEOM_FIRST)
        return torch.where(first == NO_BOUND, torch.full_like(first, -1),
                           first)


GEOM_ANY, GEOM_FIRST, GEOM_SAMPLE = 0, 1, 2
NO_BOUND = 2**31 - 1       # nb_list_eval: no bound of the list contains the row
WORK_BYTES = 256 << 20     # candidate lists of one slab of rows
# kernel dispatches of the bound evaluation since import (bench.py: which
# dispatches of a profiled run belong to the timed region)
DISPATCHES = dict(nb_eval_fast_kernel=0, nb_cand_kernel=0)
GS_OUTER, GS_INSIDE = 1, 2


MAX_DIM = 128          # n_dim limit of the device kernels


def _work(n_doubles):
    return torch.empty(int(n_doubles), dtype=torch.float64, device='cuda')


def mvee_weights(x, n_max=100, n_batch=20):
    """Weights u of the batched Khachiyan iteration (reference
    bounds/basic.py:175-232) for the rows of the cuda tensor / array ``x``
    -- ``nb_mvee_weights``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    u = torch.empty(n, dtype=torch.float64, device='cuda')
    work = _work(lib.nb_mvee_weights_work_doubles(n, d, n_batch))
    _lib.check(lib.nb_mvee_weights(_ptr(x), n, d, n_max, n_batch, _ptr(u),
                                   _ptr(work), _stream()))
    return u


def weighted_moments(x, w=None, scale=1.0):
    """scale * sum_i w_i q_i q_i^T with q_i = (x_i, 1) as an (n_dim+1)^2 cuda
    tensor -- ``nb_weighted_moments``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    out = torch.empty((d + 1, d + 1), dtype=torch.float64, device='cuda')
    work = _work(lib.nb_moments_work_doubles(n, d))
    _lib.check(lib.nb_weighted_moments(
        _ptr(x), _ptr(w) if w is not None else None, n, d, float(scale),
        _ptr(out), _ptr(work), _stream()))
    return out


def quadform_max(x, p):
    """max_i q_i^T P q_i, q_i = (x_i, 1), as a one-element cuda tensor --
    ``nb_quadform_max``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    p = torch.as_tensor(p, dtype=torch.float64).to('cuda').contiguous()
    out = torch.empty(1, dtype=torch.float64, device='cuda')
    work = _work(lib.nb_quadform_work_doubles())
    _lib.check(lib.nb_quadform_max(_ptr(x), n, d, _ptr(p), _ptr(out),
                                   _ptr(work), _stream()))
    return out


def whiten(x):
    """(xw, mean, sd, w) with xw = w ((x - mean) / sd) of zero mean and unit
    covariance -- ``nb_whiten``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    xw = torch.empty_like(x)
    stats = torch.empty(2 * d + d * d, dtype=torch.float64, device='cuda')
    work = _work(lib.nb_whiten_work_doubles(n, d))
    _lib.check(lib.nb_whiten(
        _ptr(x), n, d, _ptr(xw), _ptr(stats), _ptr(stats[d:]),
        _ptr(stats[2 * d:]), _ptr(work), _stream()))
    return xw, stats


def mvee_fit_batch(point_sets, n_max=100, n_batch=20):
    """minimum_volume_enclosing_ellipsoid (reference bounds/basic.py:175-241)
    for several point sets of one dimension at once.

    Every set is whitened on the device (``nb_whiten``; the iteration is
    affine invariant), the Khachiyan iterations of all sets advance side by
    side in the same launches (``nb_mvee_khachiyan``), centre and covariance
    follow from the weights with one matrix-core pass
    (``nb_weighted_moments``, basic.py:233-234) and the scaling from the
    largest quadratic form (``nb_quadform_max``, basic.py:236).  Only
    O(n_dim^2) numbers per set cross PCIe.  Returns a list of (c, A, A_inv)
    like the reference."""
    lib = _lib.load()
    xs, stats, us = [], [], []
    for pts in point_sets:
        xw, st = whiten(pts)
        xs.append(xw)
        stats.append(st)
        us.append(torch.empty(xw.shape[0], dtype=torch.float64,
                              device='cuda'))
    nb = len(xs)
    d = xs[0].shape[1]
    n_arr = (C.c_int64 * nb)(*[x.shape[0] for x in xs])
    work = _work(lib.nb_mvee_work_doubles(nb, max(n_arr), d, n_batch))
    _lib.check(lib.nb_mvee_khachiyan(
        nb, (C.c_void_p * nb)(*[x.data_ptr() for x in xs]), n_arr, d, n_max,
        n_batch, (C.c_void_p * nb)(*[u.data_ptr() for u in us]), _ptr(work),
        _stream()))
    moments = torch.stack([weighted_moments(x, u) for x, u in zip(xs, us)])
    moments = moments.cpu().numpy()
    stats = torch.stack(stats).cpu().numpy()
    # host: O(n_dim^2) numbers per set only
    p_all, c_all, cov_all = [], [], []
    for s in moments:
        su = s[d, d]
        c = s[d, :d] / su                               # basic.py:233
        cov = s[:d, :d] / su - np.outer(c, c)           # basic.py:234
        cov = 0.5 * (cov + cov.T)
        v = np.empty((d + 1, d + 1))
        v[:d, :d] = s[:d, :d] / su
        v[d, :d] = v[:d, d] = c
        v[d, d] = 1.0
        p_all.append(np.linalg.inv(v))
        c_all.append(c)
        cov_all.append(cov)
    gmax = torch.cat([quadform_max(x, p) for x, p in zip(xs, p_all)])
    gmax = gmax.cpu().numpy()
    out = []
    for b in range(nb):
        mean, sd = stats[b, :d], stats[b, d:2 * d]
        w = stats[b, 2 * d:].reshape(d, d)
        # x = mean + back xw with back = diag(sd) w^-1 (lower triangular)
        back = sd[:, None] * solve_triangular(w, np.eye(d), lower=True)
        fwd = w / sd[None, :]                           # back^-1
        scale = gmax[b] - 1.0                           # basic.py:236
        c = mean + back @ c_all[b]
        a_inv = back @ cov_all[b] @ back.T * scale
        a = fwd.T @ np.linalg.inv(cov_all[b]) @ fwd / scale
        out.append((c, 0.5 * (a + a.T), 0.5 * (a_inv + a_inv.T)))
    return out


def standardize(x):
    """(mean, scale, (x - mean) / scale) of the rows of a cuda tensor
    (reference neural.py:74-77) -- ``nb_standardize``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    mean = torch.empty(d, dtype=torch.float64, device='cuda')
    scale = torch.empty(d, dtype=torch.float64, device='cuda')
    out = torch.empty_like(x)
    _lib.check(lib.nb_standardize(_ptr(x), n, d, _ptr(mean), _ptr(scale),
                                  _ptr(out), _stream()))
    return mean, scale, out


def prior_transform(u, kind, loc, scale):
    """x = dist.isf(1 - u) column by column (reference prior.py:85-120) for
    uniform (kind 0) / normal (kind 1) parameters -- ``nb_prior_transform``."""
    lib = _lib.load()
    u = as_device_points(u)
    n, d = u.shape
    out = torch.empty_like(u)
    kind = np.ascontiguousarray(kind, dtype=np.uint8)
    loc, scale = _f64(loc), _f64(scale)
    _lib.check(lib.nb_prior_transform(
        _ptr(u), n, d, kind.ctypes.data_as(C.c_void_p), _dp(loc), _dp(scale),
        _ptr(out), _stream()))
    return out




def gmm_fit(x, n_init=10, seed=0, tol=1e-3, reg_covar=1e-6, max_iter=100,
            init_labels=None):
    """Two-component full-covariance Gaussian mixture fits (one per restart)
    on the device -- ``nb_gmm_fit``.  Returns a list of dicts with
    ``lower_bound, n_iter, converged, failed, weights, means, covariances``."""
    lib = _lib.load()
    x = as_device_points(x)
    n, d = x.shape
    stride = lib.nb_gmm_out_doubles(d)
    out = torch.zeros(n_init * stride, dtype=torch.float64, device='cuda')
    scratch = torch.empty(lib.nb_gmm_work_doubles(n, d, n_init),
                          dtype=torch.float64, device='cuda')
    lab = None
    if init_labels is not None:
        lab = torch.from_numpy(np.ascontiguousarray(
            init_labels, dtype=np.int32).reshape(n_init, n)).cuda()
    _lib.check(lib.nb_gmm_fit(
        _ptr(x), n, d, n_init, int(seed) & (2**64 - 1), float(tol),
        float(reg_covar), int(max_iter), _ptr(lab) if lab is not None else None,
        _ptr(out), _ptr(scratch), _stream()))
    rec = out.cpu().numpy().reshape(n_init, stride)
    per = lib.nb_gmm_scratch_doubles(n, d)
    off = lib.nb_gmm_logp_offset(d)
    fits = []
    for i, r in enumerate(rec):
        fits.append(dict(
            lower_bound=float(r[0]), n_iter=int(r[1]), converged=bool(r[2]),
            failed=bool(r[3]), weights=r[4:6].copy(),
            means=r[6:6 + 2 * d].reshape(2, d).copy(),
            covariances=r[6 + 2 * d:].reshape(2, d, d).copy(),
            # (2, n) on the device: log(w_k N(x_i; mu_k, Sigma_k)) under the
            # returned parameters
            logp=scratch[i * per + off:i * per + off + 2 * n].view(2, n)))
    return fits


def phase_shift_(x, periodic, centers, inverse=False):
    """PhaseShift.transform (reference bounds/periodic.py:50-72) applied in
    place to the rows of the cuda tensor ``x``."""
    lib = _lib.load()
    if x.shape[0] == 0 or len(periodic) == 0:
        return x
    assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
    per = np.ascontiguousarray(periodic, dtype=np.int32)
    cen = _f64(centers)
    _lib.check(lib.nb_phase_shift(
        _ptr(x), x.shape[0], x.shape[1], len(per),
        per.ctypes.data_as(_lib.c_int32_p), _dp(cen), 1 if inverse else 0,
        _stream()))
    return x


_SCRATCH = {}


def _buffer(role, shape, dtype, reuse):
    """A fresh tensor, or (reuse=True) a view of the grow-only scratch buffer
    of that role.  The refill loops of the bounds draw a different number of
    proposals in every launch (~1 GB of proposals at n_dim = 50); fresh
    allocations of ever-changing sizes keep sending the caching allocator back
    to hipMalloc -- tens of milliseconds each -- in the middle of a run."""
    if not reuse:
        return torch.empty(shape, dtype=dtype, device='cuda')
    n_bytes = math.prod(shape) * _itemsize(dtype)
    buf = _SCRATCH.get(role)
    if buf is None or buf.numel() < n_bytes:
        _SCRATCH[role] = buf = None        # release before growing
        buf = torch.empty(n_bytes + n_bytes // 4 + 512, dtype=torch.uint8,
                          device='cuda')
        _SCRATCH[role] = buf
    return buf[:n_bytes].view(dtype).view(shape)


def compact_rows(x, flags, mask=1, want_index=False, reuse=False, flip=0,
                 out_role='compact'):
    """Stable compaction of the rows of ``x`` with ((flags ^ flip) & mask)
    != 0.

    Returns (rows, counts, src_idx): ``rows`` has x.shape[0] allocated rows of
    which the first counts[1] are valid; counts is an int64 device tensor
    [rows with bit0, rows kept]."""
    lib = _lib.load()
    n, d = x.shape
    out = _buffer(out_role, (n, d), torch.float64, reuse)
    # (written by the scan kernel, or zeroed by the launcher for n = 0)
    counts = torch.empty(2, dtype=torch.int64, device='cuda')
    scratch = _buffer('compact_scratch',
                      (max(16, lib.nb_compact_scratch_bytes(n)),),
                      torch.uint8, reuse)
    src = (_buffer('compact_index', (n,), torch.int64, reuse)
           if want_index else None)
    _lib.check(lib.nb_compact_rows(
        _ptr(x), _ptr(flags), mask, flip, n, d, _ptr(out),
        _ptr(src) if src is not None else None, _ptr(counts), _ptr(scratch),
        _stream()))
    return out, counts, src


def shell_stats(log_l, threshold=-np.inf):
    """(logsumexp(l), logsumexp(2l), max l, #(l >= threshold)) on the device
    (nautilus/sampler.py:927-943, 1144)."""
    lib = _lib.load()
    n = log_l.shape[0]
    out = torch.empty(4, dtype=torch.float64, device='cuda')
    scratch = torch.empty(max(16, lib.nb_shell_stats_scratch_bytes(n)),
                          dtype=torch.uint8, device='cuda')
    _lib.check(lib.nb_shell_stats(_ptr(log_l), n, float(threshold), _ptr(out),
                                  _ptr(scratch), _stream()))
    return out


class LivePool:
    """The n_live largest log-likelihoods of the exploration phase, kept on
    the device (``nb_live_append`` / ``nb_live_select`` / ``nb_live_stats``;
    reference nautilus/sampler.py:1147-1190 sorts every stored log L on
    every iteration)."""

    def __init__(self, k, tensors=()):
        """Pool of the ``k`` largest values of ``tensors`` (all stored log L
        at construction time; sized for them once, then for k + a few
        batches)."""
        self._lib = _lib.load()
        self.k = int(k)
        self._host = np.array([-np.inf, 0.0, 0.0, 0.0])
        self._alloc(sum(int(t.shape[0]) for t in tensors) + 16)
        for t in tensors:
            self.add(t)
        if self.dirty:
            self.select()
        keep = self.bufs[self.cur][:int(self.counts[self.cur])].clone()
        thr = self.thr.clone()
        # on a likelihood plateau every tied value at the threshold stays in
        # the pool (n_gt + n_eq values, not n_live): leave room for them and
        # for as many again before the next rebuild
        self._alloc(max(4 * self.k, 2 * int(keep.shape[0])) + (1 << 17))
        self.bufs[0][:keep.shape[0]] = keep
        self.counts[0] = keep.shape[0]
        self.thr.copy_(thr)

    def _alloc(self, capacity):
        self.cap = int(capacity)
        self.bufs = [torch.empty(self.cap, dtype=torch.float64, device='cuda')
                     for _ in range(2)]
        self.counts = torch.zeros(3, dtype=torch.int32, device='cuda')
        self.thr = torch.full((1,), -np.inf, dtype=torch.float64,
                              device='cuda')
        self.stats = torch.zeros(4, dtype=torch.float64, device='cuda')
        self.cur = 0
        self.dirty = False

    def add(self, log_l):
        """Append the values of a batch that reach the current threshold."""
        if log_l.shape[0] == 0:
            return
        log_l = log_l.contiguous()
        _lib.check(self._lib.nb_live_append(
            _ptr(log_l), log_l.shape[0], _ptr(self.thr), _ptr(self.bufs[
                self.cur]), _ptr(self.counts[self.cur:]), self.cap,
            _ptr(self.counts[2:]), _stream()))
        self.dirty = True

    def select(self):
        """(threshold, #above, #equal) of the k-th largest value; the pool
        shrinks to the values that reach it."""
        if self.dirty:
            nxt = 1 - self.cur
            _lib.check(self._lib.nb_live_select(
                _ptr(self.bufs[self.cur]), _ptr(self.counts[self.cur:]),
                self.cap, self.k, _ptr(self.bufs[nxt]),
                _ptr(self.counts[nxt:]), _ptr(self.thr), _ptr(self.stats),
                _stream()))
            self.cur = nxt
            self.dirty = False
            self._host = torch.cat([self.stats[:3],
                                    self.counts[2:].double()]).cpu().numpy()
            if self._host[3] != 0:
                raise OverflowError('live pool capacity exceeded')
        return float(self._host[0]), int(self._host[1]), int(self._host[2])

    def select_with_stats(self, shells):
        """``select`` and ``shell_stats(shells)`` with ONE wait: the selection
        kernel leaves the new threshold on the device, the per-shell
        reductions read it there, and everything comes to the host in one
        copy.  Returns ((threshold, #above, #equal), rows)."""
        if not self.dirty:
            return self.select(), self.shell_stats(shells)
        nxt = 1 - self.cur
        _lib.check(self._lib.nb_live_select(
            _ptr(self.bufs[self.cur]), _ptr(self.counts[self.cur:]),
            self.cap, self.k, _ptr(self.bufs[nxt]),
            _ptr(self.counts[nxt:]), _ptr(self.thr), _ptr(self.stats),
            _stream()))
        self.cur = nxt
        self.dirty = False
        out = torch.zeros((max(1, len(shells)), 4), dtype=torch.float64,
                          device='cuda')
        for row, ll in zip(out, shells):
            _lib.check(self._lib.nb_live_stats(
                _ptr(ll), ll.shape[0], _ptr(self.thr), _ptr(row), _stream()))
        host = torch.cat([self.stats[:3], self.counts[2:].double(),
                          out[:len(shells), :3].reshape(-1)]).cpu().numpy()
        self._host = host[:4]
        if self._host[3] != 0:
            raise OverflowError('live pool capacity exceeded')
        return ((float(host[0]), int(host[1]), int(host[2])),
                host[4:].reshape(len(shells), 3))

    def smallest_above(self):
        """Smallest pooled value strictly above the threshold (the likelihood
        plateau rule of add_bound, sampler.py:1012-1020); call after
        ``select``."""
        vals = self.bufs[self.cur][:int(self.counts[self.cur])]
        return float(vals[vals > self.thr].min())

    def shell_stats(self, shells):
        """Rows (#above, logsumexp above, #equal) for the log-L tensors of
        ``shells`` against the current threshold -- one device reduction per
        shell, ONE copy to the host."""
        out = torch.zeros((max(1, len(shells)), 4), dtype=torch.float64,
                          device='cuda')
        for row, ll in zip(out, shells):
            _lib.check(self._lib.nb_live_stats(
                _ptr(ll), ll.shape[0], _ptr(self.thr), _ptr(row), _stream()))
        return out[:len(shells), :3].cpu().numpy()


def philox_uniform(seed, offset, block, tag, n):
    lib = _lib.load()
    u = torch.empty((n, 2), dtype=torch.float64, device='cuda')
    _lib.check(lib.nb_philox_uniform(seed, offset, block, tag, n, _ptr(u),
                                     _stream()))
    return u


def mfma_f64_peak(iters=20000):
    lib = _lib.load()
    out = C.c_double(0.0)
    _lib.check(lib.nb_mfma_f64_peak(iters, C.byref(out)))
    return out.value


class EvalCounters:
    """Algorithmic-work counters of the bound-evaluation kernel (bench.py)."""

    def __init__(self):
        self._lib = _lib.load()
        self.buf = torch.zeros(4, dtype=torch.int64, device='cuda')

    def __enter__(self):
        self.buf.zero_()
        _lib.check(self._lib.nb_set_eval_counters(_ptr(self.buf)))
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        _lib.check(self._lib.nb_set_eval_counters(None))

    def read(self):
        c = self.buf.cpu().numpy()
        return dict(outer_point_evals=int(c[0]), ellipsoid_point_evals=int(c[1]),
                    emulator_point_evals=int(c[2]))


class KernelTimer:
    """Per-kernel-family wall time on the current stream with HIP events
    (torch.cuda.Event records on torch's current stream, which is the stream
    every launch of this module uses)."""

    active = None

    def __init__(self):
        self.events = {}
        self.folded = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def _fold(self, name):
        # long runs launch 10^5 kernels: keep a bounded number of live events
        pairs = self.events.pop(name, [])
        if pairs:
            pairs[-1][1].synchronize()
            acc = self.folded.setdefault(name, [0, 0.0])
            acc[0] += len(pairs)
            acc[1] += sum(a.elapsed_time(b) for a, b in pairs)

    def totals(self):
        torch.cuda.synchronize()
        for name in list(self.events):
            self._fold(name)
        return {name: dict(launches=acc[0], ms=acc[1])
                for name, acc in self.folded.items()}


def _timed(name):
    def deco(fn):
        def wrapper(*args, **kwargs):
            timer = KernelTimer.active
            if timer is None:
                return fn(*args, **kwargs)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(*args, **kwargs)
            b.record()
            pairs = timer.events.setdefault(name, [])
            pairs.append((a, b))
            if len(pairs) >= 1024:
                timer._fold(name)
            return out
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        return wrapper
    return deco


# HIP-event time per kernel family (bench.py).  'bound_eval' = everything that
# evaluates bounds on the matrix cores -- the fused acceptance kernel
# (nb_eval_fast_kernel), the geometric stage (nb_cand_kernel) and the gathered
# emulator scores behind it.
DeviceBound.contains = _timed('bound_eval')(DeviceBound.contains)
DeviceBound.accept = _timed('bound_eval')(DeviceBound.accept)
DeviceBound.neural_score = _timed('bound_eval')(DeviceBound.neural_score)
DeviceBound.propose = _timed('nb_draw_kernel')(DeviceBound.propose)
DeviceBound.contains_stream = _timed('nb_ell_stream_kernel')(
    DeviceBound.contains_stream)
DeviceBoundList.contains_any = _timed('bound_eval')(
    DeviceBoundList.contains_any)
DeviceBoundList.inside_flags = _timed('bound_eval')(
    DeviceBoundList.inside_flags)
DeviceBoundList.first_containing = _timed('bound_eval')(
    DeviceBoundList.first_containing)
compact_rows = _timed('nb_compact')(compact_rows)
shell_stats = _timed('nb_lse')(shell_stats)
