"""Likelihood emulator: an ensemble of (100, 50, 20) ReLU MLPs trained and
evaluated on the MI355X matrix cores.

Mirrors ``nautilus.neural.NeuralNetworkEmulator`` (reference
nautilus/neural.py:35-187): ``train(x, y, n_networks, neural_network_kwargs,
pool)`` / ``predict(x)`` / attributes ``mean``, ``scale``,
``neural_networks``.  The training algorithm is scikit-learn's
``MLPRegressor.fit`` as the reference configures it (neural.py:79-81); weight
initialisation and the per-epoch shuffles are drawn on the host from
``numpy.random.RandomState(i)`` exactly as scikit-learn does, so network ``i``
sees the same initial weights and minibatch order as in the reference; all
arithmetic (forward, backprop, Adam) runs in ``nb_mlp_train.hip``.
"""

import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib, device

HIDDEN = (100, 50, 20)
EPOCH_CHUNK = 16     # epochs per kernel launch (host prepares the next chunk
                     # of shuffles while the GPU trains)
CHUNK_BYTES = 32 << 20   # ... fewer where a chunk's orders would exceed this


class Network:
    """Weights of one trained network in scikit-learn layout."""

    def __init__(self, coefs, intercepts, n_iter=0, loss_curve=None):
        self.coefs_ = coefs
        self.intercepts_ = intercepts
        self.n_iter_ = n_iter
        self.loss_curve_ = [] if loss_curve is None else list(loss_curve)
        self.n_layers_ = 5


def check_hidden(hidden):
    """The architectures the device kernels hold: three hidden layers of at
    most (100, 50, 20) units -- the reference's default
    (nautilus/neural.py:79-81) and anything narrower.  The kernels' tiles are
    those of the default; a narrower layer is the default with zero weights
    for the missing units, which is exact for prediction AND for training: a
    missing unit has pre-activation 0, ReLU derivative 0 and feeds zero
    weights, so every gradient and every Adam moment of its weights stays 0
    (emulator.pad_network).  Anything else -- other depths, wider layers --
    raises ValueError (the reference passes any MLPRegressor option
    through, neural.py:79-83)."""
    try:
        hidden = tuple(int(h) for h in np.atleast_1d(hidden))
    except (TypeError, ValueError):
        raise ValueError('hidden_layer_sizes=%r is not a sequence of '
                         'integers' % (hidden,))
    if len(hidden) != 3 or any(h < 1 or h > m for h, m in zip(hidden, HIDDEN)):
        raise ValueError(
            'nautilus_amd holds emulators with three hidden layers of at most '
            '%r units on the device (the reference default and anything '
            'narrower); hidden_layer_sizes=%r is not supported' %
            (HIDDEN, hidden))
    return hidden


def pad_network(coefs, intercepts, n_dim):
    """Weights of a (narrower) network in the device's default shapes: zero
    rows / columns for the missing units."""
    units = [n_dim, *HIDDEN, 1]
    cs, bs = [], []
    for k, (a, b) in enumerate(zip(units[:-1], units[1:])):
        w = np.asarray(coefs[k], dtype=np.float64)
        v = np.asarray(intercepts[k], dtype=np.float64)
        if w.shape == (a, b):
            cs.append(w)
            bs.append(v)
            continue
        full = np.zeros((a, b))
        full[:w.shape[0], :w.shape[1]] = w
        fb = np.zeros(b)
        fb[:v.shape[0]] = v
        cs.append(full)
        bs.append(fb)
    return cs, bs


def hidden_of(coefs):
    return tuple(int(c.shape[1]) for c in coefs[:-1])


def _glorot(n_in, rs, hidden=HIDDEN):
    """sklearn/_multilayer_perceptron.py:441-456."""
    units = [n_in, *hidden, 1]
    coefs, intercepts = [], []
    for fan_in, fan_out in zip(units[:-1], units[1:]):
        bound = np.sqrt(6.0 / (fan_in + fan_out))
        coefs.append(rs.uniform(-bound, bound, (fan_in, fan_out)))
        intercepts.append(rs.uniform(-bound, bound, fan_out))
    return coefs, intercepts


def _weight_pointers(nets):
    e = len(nets)
    cp = (_lib.c_double_p * (4 * e))()
    ip = (_lib.c_double_p * (4 * e))()
    keep = []
    for i, (coefs, intercepts) in enumerate(nets):
        coefs, intercepts = pad_network(coefs, intercepts,
                                        np.shape(coefs[0])[0])
        for k in range(4):
            w = np.ascontiguousarray(coefs[k], dtype=np.float64)
            b = np.ascontiguousarray(intercepts[k], dtype=np.float64)
            keep += [w, b]
            cp[4 * i + k] = w.ctypes.data_as(_lib.c_double_p)
            ip[4 * i + k] = b.ctypes.data_as(_lib.c_double_p)
    return cp, ip, keep


class Trainer:
    """Thin object wrapper of the ``nb_trainer_*`` C ABI."""

    def __init__(self, x_dev, y_dev, init_nets, hparams=None):
        """``x_dev`` / ``y_dev``: one training set for all networks, or lists
        with the set of every network (a fleet: the networks of several
        ensembles in one trainer, ``nb_trainer_create_fleet``)."""
        lib = _lib.load()
        self._lib = lib
        self.e = len(init_nets)
        fleet = isinstance(x_dev, (list, tuple))
        self.xs = list(x_dev) if fleet else [x_dev] * self.e
        self.ys = list(y_dev) if fleet else [y_dev] * self.e   # kept alive
        self.ns = [int(x.shape[0]) for x in self.xs]
        self.n, self.n_dim = self.xs[0].shape
        cp, ip, keep = _weight_pointers(init_nets)
        h = C.c_void_p()
        if fleet:
            ns = (C.c_int64 * self.e)(*self.ns)
            xp = (C.c_void_p * self.e)(*[x.data_ptr() for x in self.xs])
            yp = (C.c_void_p * self.e)(*[y.data_ptr() for y in self.ys])
            _lib.check(lib.nb_trainer_create_fleet(
                self.n_dim, self.e, ns, xp, yp, cp, ip, C.byref(h)))
        else:
            _lib.check(lib.nb_trainer_create(
                self.n_dim, self.e, self.n, C.c_void_p(x_dev.data_ptr()),
                C.c_void_p(y_dev.data_ptr()), cp, ip, C.byref(h)))
        self.fleet = fleet
        self._h = h
        if hparams:
            hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, epsilon=1e-8,
                      batch=200, max_iter=10000, n_iter_no_change=10, tol=0.0)
            hp.update(hparams)
            _lib.check(lib.nb_trainer_set_hparams(
                h, hp['lr'], hp['beta1'], hp['beta2'], hp['epsilon'],
                hp['batch'], hp['max_iter'], hp['n_iter_no_change'],
                hp['tol']))

    def run(self, perms, sync=True):
        """perms: int32 array (E, n_epochs, n), or for a fleet a list of
        (n_epochs, n_i) arrays.  Returns per-network status (n_iter, negative
        once stopped) if sync else None."""
        status = (C.c_int32 * self.e)()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.fleet:
            # one upload: the shuffles of all networks back to back
            n_epochs = perms[0].shape[0]
            flat = np.concatenate([np.ascontiguousarray(
                p, dtype=np.int32).reshape(-1) for p in perms])
            perms_dev = torch.from_numpy(flat).cuda()
            offs = np.concatenate([[0], np.cumsum(
                [p.size for p in perms])[:-1]])
            ptrs = (C.c_void_p * self.e)(*[
                perms_dev.data_ptr() + 4 * int(o) for o in offs])
            _lib.check(self._lib.nb_trainer_run_fleet(
                self._h, ptrs, n_epochs, status if sync else None, stream))
        else:
            perms_dev = torch.from_numpy(
                np.ascontiguousarray(perms, dtype=np.int32)).cuda()
            _lib.check(self._lib.nb_trainer_run(
                self._h, C.c_void_p(perms_dev.data_ptr()), perms.shape[1],
                status if sync else None, stream))
        self._perms = getattr(self, '_perms', [])[-1:] + [perms_dev]
        return np.array(status[:]) if sync else None

    def run_async(self, perms_dev, offsets, n_epochs):
        """Enqueue ``n_epochs`` epochs whose orders sit in the int32 cuda
        tensor ``perms_dev`` (network i from element offsets[i]); returns the
        ticket for ``wait`` (``nb_trainer_run_async``)."""
        ptrs = (C.c_void_p * self.e)(*[
            perms_dev.data_ptr() + 4 * int(o) for o in offsets])
        ticket = C.c_int64()
        _lib.check(self._lib.nb_trainer_run_async(
            self._h, ptrs, n_epochs,
            C.c_void_p(torch.cuda.current_stream().cuda_stream),
            C.byref(ticket)))
        return ticket.value

    def wait(self, ticket):
        """Per-network status behind the chunk of ``ticket`` (n_iter, negative
        once stopped); later chunks may still be queued."""
        status = (C.c_int32 * self.e)()
        _lib.check(self._lib.nb_trainer_wait(self._h, ticket, status))
        return np.array(status[:])

    def status(self):
        """Wait for the enqueued epochs; per-network n_iter (negative once
        the network has stopped)."""
        status = (C.c_int32 * self.e)()
        _lib.check(self._lib.nb_trainer_status(
            self._h, status,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return np.array(status[:])

    def loss_curve(self, net, n):
        out = np.zeros(max(1, n))
        _lib.check(self._lib.nb_trainer_loss_curve(
            self._h, net, out.ctypes.data_as(_lib.c_double_p), len(out)))
        return out[:n]

    def weights(self, net):
        units = [self.n_dim, *HIDDEN, 1]
        coefs = [np.zeros((a, b)) for a, b in zip(units[:-1], units[1:])]
        intercepts = [np.zeros(b) for b in units[1:]]
        cp = (_lib.c_double_p * 4)(*[w.ctypes.data_as(_lib.c_double_p)
                                      for w in coefs])
        ip = (_lib.c_double_p * 4)(*[w.ctypes.data_as(_lib.c_double_p)
                                      for w in intercepts])
        _lib.check(self._lib.nb_trainer_weights(self._h, net, cp, ip))
        return coefs, intercepts

    def close(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.nb_trainer_destroy(h)
            self._h = None

    def __del__(self):
        self.close()


def _hparams_from_kwargs(kwargs):
    """Translate MLPRegressor keyword arguments (neural.py:79-88)."""
    known = dict(learning_rate_init='lr', beta_1='beta1', beta_2='beta2',
                 epsilon='epsilon', batch_size='batch', max_iter='max_iter',
                 n_iter_no_change='n_iter_no_change', tol='tol')
    fixed = dict(alpha=0, activation='relu', solver='adam', shuffle=True,
                 early_stopping=False)
    hp = {}
    for key, val in kwargs.items():
        if key == 'random_state':
            warnings.warn("The 'random_state' keyword argument passed to the"
                          " neural network is ignored.", Warning, stacklevel=3)
        elif key in known:
            hp[known[key]] = val
        elif key == 'hidden_layer_sizes':
            hp['hidden'] = check_hidden(val)
        elif key in fixed:
            if val != fixed[key]:
                raise ValueError(
                    'nautilus_amd trains ReLU networks with Adam and no '
                    'weight decay on the device (the reference default); '
                    '%s=%r is not supported' % (key, val))
        else:
            raise ValueError(
                'MLPRegressor option %r is not supported on the device' % key)
    return hp


def check_network_kwargs(kwargs):
    """Raise ValueError for options the device trainer does not hold
    (``Sampler.__init__`` calls this, so that a run does not fail at its
    first ``add_bound``)."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        _hparams_from_kwargs(dict(kwargs))


class NeuralNetworkEmulator:
    """Drop-in for ``nautilus.neural.NeuralNetworkEmulator``."""

    @classmethod
    def train(cls, x, y, n_networks=4, neural_network_kwargs={}, pool=None,
              comm=None):
        """neural.py:50-98.  ``x`` / ``y`` may be numpy arrays or cuda
        tensors; ``pool`` is accepted for API compatibility (the networks
        train concurrently on the GPU, one XCD each); ``comm`` deals them out
        over the GPUs of a sharded run."""
        return cls.train_many([(x, y)], n_networks, neural_network_kwargs,
                              comm=comm)[0]

    @classmethod
    def train_many(cls, data, n_networks=4, neural_network_kwargs={},
                   comm=None):
        """``train`` for several (x, y) sets at once; the ensembles train
        concurrently on separate streams.  With ``comm`` (a
        ``parallel.ShardedComm``) network g of the flattened (ensemble,
        seed) list trains on rank g mod world and the weights are exchanged
        afterwards (reference neural.py:93-96: networks mapped over the
        pool)."""
        hp = _hparams_from_kwargs(dict(neural_network_kwargs))
        emus, jobs = [], []
        for x, y in data:
            emu = cls()
            xt = device.as_device_points(x)
            yt = (y if isinstance(y, torch.Tensor) else
                  torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64))
                  ).to('cuda', torch.float64).contiguous()
            mean, scale, xs = device.standardize(xt)     # neural.py:74-77
            emu.mean = mean.cpu().numpy()
            emu.scale = scale.cpu().numpy()
            emu._dev = None
            emus.append(emu)
            jobs.append(dict(xs=xs, y=yt, seeds=list(range(n_networks)),
                             hparams=hp))
        results = (train_ensembles(jobs) if comm is None or comm.world == 1
                   else train_ensembles_sharded(jobs, comm))
        user = {k: v for k, v in dict(neural_network_kwargs).items()
                if k != 'random_state'}
        for emu, (nets, stats) in zip(emus, results):
            emu.neural_networks, emu.trainer_stats = nets, stats
            for net in nets:
                net.sk_params = dict(user)       # checkpoint attributes
                net.t_ = int(net.n_iter_) * int(stats['n_rows'])
        return emus

    @classmethod
    def from_weights(cls, mean, scale, networks):
        emu = cls()
        emu.mean = np.asarray(mean, float)
        emu.scale = np.asarray(scale, float)
        emu.neural_networks = list(networks)
        emu._dev = None
        return emu

    def mlp_desc(self):
        return dict(mean=self.mean, scale=self.scale,
                    nets=[(n.coefs_, n.intercepts_)
                          for n in self.neural_networks])

    def predict_device(self, x):
        if self._dev is None:
            d = len(self.mean)
            ident = device.member(np.zeros(d), np.eye(d), np.eye(d))
            self._dev = device.DeviceBound(d, [], None, False, [dict(
                ellipsoid=ident, score_predict_min=0.0, mlp=self.mlp_desc())])
        return self._dev.neural_score(x)[1]

    def predict(self, x):
        """neural.py:100-116; numpy in -> numpy out, tensor in -> tensor."""
        out = self.predict_device(x)
        return out if isinstance(x, torch.Tensor) else out.cpu().numpy()

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev'] = None
        return state


class _ShuffleStreams:
    """The per-network minibatch orders of scikit-learn's fit loop
    (_multilayer_perceptron.py:700-704: ``sample_idx = shuffle(sample_idx,
    random_state=self._random_state)`` before every epoch): numpy's legacy
    MT19937 ``RandomState.shuffle``, composed epoch after epoch.  The states
    are taken over from the ``RandomState`` objects that drew the initial
    weights and advanced by ``nb_host_shuffle_epochs`` (one native thread per
    network; bit for bit numpy's stream, tests/test_host_logic.py)."""

    def __init__(self, states, ns):
        self._lib = _lib.load()
        self.e = len(states)
        self.ns = [int(n) for n in ns]
        self.keys, self.pos = [], (C.c_int32 * self.e)()
        for i, rs in enumerate(states):
            kind, key, pos = rs.get_state()[:3]
            assert kind == 'MT19937'
            self.keys.append(np.ascontiguousarray(key, dtype=np.uint32).copy())
            self.pos[i] = int(pos)
        self.orders = [np.arange(n, dtype=np.int32) for n in self.ns]
        self.offsets = np.concatenate([[0], np.cumsum(self.ns)[:-1]])

    def fill(self, out, n_epochs, active):
        """Write the next ``n_epochs`` orders of every active network into the
        flat int32 array ``out`` (network i at offsets[i] * n_epochs); inactive
        networks keep their generator where it is and repeat their order."""
        e = self.e
        key_p = (C.c_void_p * e)(*[k.ctypes.data for k in self.keys])
        ord_p = (C.c_void_p * e)(*[o.ctypes.data for o in self.orders])
        n_of = (C.c_int64 * e)(*self.ns)
        out_p = (C.c_void_p * e)()
        base = out.ctypes.data
        for i in range(e):
            at = int(self.offsets[i]) * n_epochs
            if active[i]:
                out_p[i] = base + 4 * at
            else:
                out_p[i] = None
                out[at:at + n_epochs * self.ns[i]].reshape(
                    n_epochs, self.ns[i])[:] = self.orders[i]
        _lib.check(self._lib.nb_host_shuffle_epochs(
            e, key_p, self.pos, n_of, n_epochs, ord_p, out_p))


class _TrainJob:
    """Networks in flight -- one ensemble, or a fleet of several ensembles
    with a training set each -- : the trainer, the per-network shuffle
    streams and the bookkeeping of the chunked epoch loop.  Two chunks of
    epochs are kept in flight: chunk k + 1 is enqueued (its orders drawn and
    uploaded) before the status of chunk k is read, so the GPU does not wait
    for the host between chunks; a network that stops inside chunk k skips
    chunk k + 1 on the device."""

    def __init__(self, members, hparams, max_epochs, stream):
        """``members``: list of dicts (xs, y, seeds, permutations, init), one
        per ensemble."""
        self.members = members
        self.stream = stream
        self.hidden = tuple((hparams or {}).get('hidden', HIDDEN))
        d = members[0]['xs'].shape[1]
        self.owner, states, self.perm_src = [], [], []
        xs, ys, nets0 = [], [], []
        for k, m in enumerate(members):
            for j, seed in enumerate(m['seeds']):
                rs = np.random.RandomState(seed)
                init = m.get('init')
                # (the Glorot draw always advances the stream, as in
                # scikit-learn, also when the weights are then replaced)
                drawn = _glorot(d, rs, (hparams or {}).get('hidden', HIDDEN))
                nets0.append(drawn if init is None else init[j])
                self.owner.append(k)
                states.append(rs)
                perms = m.get('permutations')
                self.perm_src.append(None if perms is None else perms[j])
                xs.append(m['xs'])
                ys.append(m['y'])
        self.e = len(self.owner)
        self.ns = [int(x.shape[0]) for x in xs]
        self.fleet = len(members) > 1
        with torch.cuda.stream(stream):
            self.trainer = (Trainer(xs, ys, nets0, hparams) if self.fleet
                            else Trainer(xs[0], ys[0], nets0, hparams))
        self.max_iter = (hparams or {}).get('max_iter', 10000)
        if max_epochs is not None:
            self.max_iter = min(self.max_iter, max_epochs)
        self.shuffles = _ShuffleStreams(states, self.ns)
        # epochs per launch: EPOCH_CHUNK, fewer where the orders of a chunk
        # would exceed CHUNK_BYTES (config 5: 8 networks x 2 x 10^5 rows)
        per_epoch = 4 * sum(self.ns)
        self.chunk = int(max(2, min(EPOCH_CHUNK, CHUNK_BYTES // per_epoch)))
        self.status = np.zeros(self.e, dtype=int)
        self.done_epochs = 0
        self.tickets = []            # chunks in flight (oldest first)
        self.finished = False
        self._ring = []              # (pinned host orders, device orders)

    def _buffers(self, n_items):
        """Pinned host / device buffers of a chunk; a slot is reused once its
        chunk has been waited for (at most three are alive)."""
        if len(self._ring) < 3:
            pin = torch.empty(n_items, dtype=torch.int32,
                              pin_memory=torch.cuda.is_available())
            dev = torch.empty(n_items, dtype=torch.int32, device='cuda')
            self._ring.append((pin, dev))
            return pin, dev
        slot = self._ring.pop(0)
        self._ring.append(slot)
        if slot[0].numel() < n_items:
            raise RuntimeError('chunk larger than the first one')
        return slot

    def next_chunk(self):
        """Orders of the next chunk of epochs as (pinned flat int32 tensor,
        device tensor, epochs), or None.  Host work that overlaps with the GPU
        training the chunks in flight."""
        chunk = min(self.chunk, self.max_iter - self.done_epochs)
        if chunk <= 0:
            return None
        pin, dev = self._buffers(self.chunk * sum(self.ns))
        out = pin.numpy()
        active = [self.status[i] >= 0 and self.perm_src[i] is None
                  for i in range(self.e)]
        self.shuffles.fill(out, chunk, active)
        for i in range(self.e):
            if self.perm_src[i] is not None:         # orders given (tests)
                at = int(self.shuffles.offsets[i]) * chunk
                rows = out[at:at + chunk * self.ns[i]].reshape(chunk,
                                                               self.ns[i])
                for ep in range(chunk):
                    rows[ep] = np.asarray(
                        self.perm_src[i][self.done_epochs + ep])
        return pin, dev, chunk

    def step(self):
        """Enqueue the next chunk behind the one in flight, then collect the
        status of the oldest.  Returns False once all networks are done."""
        if self.finished:
            return False
        nxt = self.next_chunk() if np.any(self.status >= 0) else None
        with torch.cuda.stream(self.stream):
            if nxt is not None:
                pin, dev, chunk = nxt
                n_items = chunk * sum(self.ns)
                dev[:n_items].copy_(pin[:n_items], non_blocking=True)
                self.tickets.append(self.trainer.run_async(
                    dev, self.shuffles.offsets * chunk, chunk))
                self.done_epochs += chunk
            # the chunk just enqueued stays queued behind the one whose
            # status is read now; with nothing new, everything is collected
            keep = 1 if nxt is not None else 0
            while len(self.tickets) > keep:
                self.status = self.trainer.wait(self.tickets.pop(0))
            if not np.any(self.status >= 0):
                # all stopped: the queued chunk exits at once on the device
                while self.tickets:
                    self.status = self.trainer.wait(self.tickets.pop(0))
        if not self.tickets:
            self.finished = True
            return False
        return True

    def release(self):
        """Destroy the trainer (frees its XCDs and device buffers)."""
        self.trainer.close()
        self._ring = []

    def results(self):
        """One (networks, stats) pair per ensemble."""
        out = [([], dict(n_iter=[], n_rows=int(m['xs'].shape[0])))
               for m in self.members]
        with torch.cuda.stream(self.stream):
            for i in range(self.e):
                n_iter = abs(int(self.status[i]))
                coefs, intercepts = self.trainer.weights(i)
                if self.hidden != HIDDEN:
                    units = [coefs[0].shape[0], *self.hidden, 1]
                    coefs = [np.ascontiguousarray(c[:a, :b]) for c, a, b in
                             zip(coefs, units[:-1], units[1:])]
                    intercepts = [np.ascontiguousarray(v[:b]) for v, b in
                                  zip(intercepts, units[1:])]
                net = Network(coefs, intercepts, n_iter,
                              self.trainer.loss_curve(i, n_iter))
                # scikit-learn's sample counter (_multilayer_perceptron.py:728)
                net.t_ = n_iter * self.ns[i]
                nets, stats = out[self.owner[i]]
                nets.append(net)
                stats['n_iter'].append(n_iter)
        return out


MAX_RESIDENT = 16    # networks of one resident launch (two per XCD)


def train_ensembles(jobs):
    """Train several ensembles (the neural bounds of a multi-modal
    NautilusBound).  The resident training kernel takes up to 16 networks per
    launch -- every network on the 32 CUs of an XCD, two networks per XCD
    beyond eight -- each with the training set of its ensemble, so the
    ensembles are packed into fleets of at most 16 networks and the fleets
    train one after the other.  (Separate trainers side by side would leave
    all but the first two ensembles with two launches per Adam step next to
    the resident kernels: measured 80-120 us per step instead of 18.)  With
    NB_TRAIN_TWO_LAUNCH set (several processes on one GPU) every ensemble
    gets a trainer and a stream of its own.  ``jobs``: list of dicts with
    keys xs, y, seeds and optionally hparams, permutations, init,
    max_epochs."""
    import os
    main = torch.cuda.current_stream()
    out = [None] * len(jobs)
    if len(jobs) == 0:
        return out

    def finish(job, keys):
        while job.step():
            pass
        for k, res in zip(keys, job.results()):
            out[k] = res
        main.wait_stream(job.stream)
        job.release()

    same = all((j.get('hparams') or {}) == (jobs[0].get('hparams') or {}) and
               j.get('max_epochs') == jobs[0].get('max_epochs') for j in jobs)
    if os.environ.get('NB_TRAIN_TWO_LAUNCH') or not same:
        running = []
        for k, job in enumerate(jobs):
            stream = main if len(jobs) == 1 else torch.cuda.Stream()
            stream.wait_stream(main)
            running.append((k, _TrainJob([job], job.get('hparams'),
                                         job.get('max_epochs'), stream)))
        active = list(running)
        while active:
            active = [(k, j) for k, j in active if j.step()]
        for k, j in running:
            finish(j, [k])
        return out
    # fleets of at most MAX_RESIDENT networks, ensembles in order
    fleets, cur, size = [], [], 0
    for k, job in enumerate(jobs):
        e = len(job['seeds'])
        if cur and size + e > MAX_RESIDENT:
            fleets.append(cur)
            cur, size = [], 0
        cur.append(k)
        size += e
    fleets.append(cur)
    for keys in fleets:
        try:
            job = _TrainJob([jobs[k] for k in keys], jobs[0].get('hparams'),
                            jobs[0].get('max_epochs'), main)
        except RuntimeError as err:
            # A fleet (several training sets in one trainer) needs the
            # resident kernel; where that is not to be had -- the XCD
            # placement probe fails on this GPU / partition mode, or other
            # live trainers hold the XCDs -- every ensemble gets a trainer of
            # its own, which can fall back to two launches per step.
            if len(keys) == 1 or getattr(err, 'code', None) != \
                    _lib.ERR_UNSUPPORTED:
                raise
            for k in keys:
                finish(_TrainJob([jobs[k]], jobs[k].get('hparams'),
                                 jobs[k].get('max_epochs'), main), [k])
            continue
        finish(job, keys)
    return out


def train_networks(xs, y, seeds, hparams=None, permutations=None,
                   init=None, max_epochs=None):
    """Train ``len(seeds)`` networks on standardised inputs ``xs`` (cuda
    tensor) concurrently.  ``permutations`` (list over networks of lists of
    per-epoch orders) and ``init`` override the RandomState draws (tests)."""
    return train_ensembles([dict(xs=xs, y=y, seeds=seeds, hparams=hparams,
                                 permutations=permutations, init=init,
                                 max_epochs=max_epochs)])[0]


def _pack_network(net, n_dim):
    """[n_iter, final loss, coefs..., intercepts...] as one float64 row."""
    parts = [np.array([net.n_iter_, net.loss_curve_[-1]
                       if len(net.loss_curve_) else 0.0])]
    coefs, intercepts = pad_network(net.coefs_, net.intercepts_, n_dim)
    parts += [np.ravel(c) for c in coefs]
    parts += [np.ravel(b) for b in intercepts]
    return np.concatenate(parts)


def _unpack_network(row, n_dim, hidden=HIDDEN):
    units = [n_dim, *HIDDEN, 1]
    real = [n_dim, *hidden, 1]
    pos = 2
    coefs, intercepts = [], []
    for a, b, ra, rb in zip(units[:-1], units[1:], real[:-1], real[1:]):
        coefs.append(row[pos:pos + a * b].reshape(a, b)[:ra, :rb].copy())
        pos += a * b
    for b, rb in zip(units[1:], real[1:]):
        intercepts.append(row[pos:pos + b][:rb].copy())
        pos += b
    return Network(coefs, intercepts, int(row[0]), [float(row[1])])


def train_ensembles_sharded(jobs, comm):
    """``train_ensembles`` with the networks dealt out over the ranks of
    ``comm``: network g (ensembles in order, seeds in order) belongs to rank
    g mod world.  Every rank trains its share as smaller ensembles, then one
    all-reduce of the zero-padded weight rows brings all networks to all
    ranks.  Training is deterministic given (data, seed), so the result is
    bit for bit that of ``train_ensembles``; only the last entry of each
    remote network's loss curve travels."""
    flat = [(j, s) for j, job in enumerate(jobs) for s in job['seeds']]
    mine = [g for g in range(len(flat)) if g % comm.world == comm.rank]
    local_jobs, local_of = [], {}
    for g in mine:
        j, seed = flat[g]
        key = j
        if key not in local_of:
            local_of[key] = len(local_jobs)
            job = dict(jobs[j])
            job['seeds'] = []
            for opt in ('permutations', 'init'):
                if job.get(opt) is not None:
                    job[opt] = []
            local_jobs.append(job)
        lj = local_jobs[local_of[key]]
        lj['seeds'].append(seed)
        pos = jobs[j]['seeds'].index(seed)
        for opt in ('permutations', 'init'):
            if jobs[j].get(opt) is not None:
                lj[opt].append(jobs[j][opt][pos])
    trained = train_ensembles(local_jobs) if local_jobs else []
    n_dim = jobs[0]['xs'].shape[1]
    units = [n_dim, *HIDDEN, 1]
    width = 2 + sum(a * b + b for a, b in zip(units[:-1], units[1:]))
    rows = np.zeros((len(flat), width))
    for g in mine:
        j, seed = flat[g]
        nets, _ = trained[local_of[j]]
        net = nets[local_jobs[local_of[j]]['seeds'].index(seed)]
        rows[g] = _pack_network(net, n_dim)
    rows = comm.sum_rows(torch.from_numpy(rows).cuda()).cpu().numpy()
    out, g = [], 0
    for job in jobs:
        nets = []
        for _ in job['seeds']:
            nets.append(_unpack_network(
                rows[g], n_dim,
                tuple((job.get('hparams') or {}).get('hidden', HIDDEN))))
            g += 1
        out.append((nets, dict(n_iter=[n.n_iter_ for n in nets],
                               n_rows=job['xs'].shape[0])))
    return out
