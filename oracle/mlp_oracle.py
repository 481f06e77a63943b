"""Oracle restatement of the emulator: nautilus/neural.py plus the part of
scikit-learn's ``MLPRegressor`` it drives.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Third-party note (prompt section (3)): the arithmetic of the emulator lives in
scikit-learn (``pyproject.toml:12`` declares ``scikit-learn>=0.22.0``, no lock
file; 1.7.2 is installed in the build container).  This file restates the
published algorithm of ``MLPRegressor(hidden_layer_sizes=(100, 50, 20),
activation='relu', solver='adam', alpha=0, learning_rate_init=1e-2,
max_iter=10000, tol=0, n_iter_no_change=10, batch_size='auto', shuffle=True,
beta_1=0.9, beta_2=0.999, epsilon=1e-8)`` as configured at
nautilus/neural.py:79-81, citing ``sklearn/neural_network/...`` lines, and is
pinned against ``MLPRegressor.fit`` itself by ``tests/golden/emulator_*.npz``
(weights, loss curve, n_iter_, predictions).
"""

import numpy as np
from threadpoolctl import threadpool_limits

HIDDEN = (100, 50, 20)


def layer_sizes(n_in, hidden=HIDDEN):
    return [n_in, *hidden, 1]


def glorot_init(n_in, random_state, hidden=HIDDEN):
    """sklearn/_multilayer_perceptron.py:441-456: per layer, weights then
    biases, both U(-b, b) with b = sqrt(6 / (fan_in + fan_out)), drawn from
    ``RandomState(random_state)``.  Returns (coefs, intercepts, rs) so the
    caller can keep drawing epoch shuffles from the same stream."""
    rs = np.random.RandomState(random_state)
    units = layer_sizes(n_in, hidden)
    coefs, intercepts = [], []
    for fan_in, fan_out in zip(units[:-1], units[1:]):
        b = np.sqrt(6.0 / (fan_in + fan_out))
        coefs.append(rs.uniform(-b, b, (fan_in, fan_out)))
        intercepts.append(rs.uniform(-b, b, fan_out))
    return coefs, intercepts, rs


def epoch_permutation(rs, sample_idx):
    """sklearn/_multilayer_perceptron.py:700-704 -> sklearn.utils.shuffle ->
    resample(replace=False): ``indices = arange(n); rs.shuffle(indices)`` and
    the previous order is indexed with it (permutations compose)."""
    idx = np.arange(len(sample_idx))
    rs.shuffle(idx)
    return sample_idx[idx]


def forward(x, coefs, intercepts):
    """sklearn/_multilayer_perceptron.py:185-220 (ReLU hidden, identity out).
    Returns the list of layer activations."""
    acts = [x]
    last = len(coefs) - 1
    for i, (w, b) in enumerate(zip(coefs, intercepts)):
        a = acts[-1] @ w
        a += b
        if i != last:
            np.maximum(a, 0, out=a)
        acts.append(a)
    return acts


def loss_and_grads(x, y, coefs, intercepts):
    """One minibatch of ``_backprop`` (sklearn/_multilayer_perceptron.py:
    297-389) with alpha = 0 and squared loss (sklearn/neural_network/_base.py:
    187-189).  ``y`` has shape (n, 1)."""
    n = x.shape[0]
    acts = forward(x, coefs, intercepts)
    loss = 0.5 * np.average((y - acts[-1])**2, axis=0).mean()
    n_layers = len(coefs)
    cg = [None] * n_layers
    ig = [None] * n_layers
    delta = acts[-1] - y
    for i in range(n_layers - 1, -1, -1):
        cg[i] = (acts[i].T @ delta) / n
        ig[i] = np.sum(delta, axis=0) / n
        if i > 0:
            delta = delta @ coefs[i].T
            delta[acts[i] == 0] = 0
    return loss, cg, ig


class Adam:
    """sklearn/neural_network/_stochastic_optimizers.py:255-287."""

    def __init__(self, params, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8):
        self.params = params
        self.lr0, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.ms = [np.zeros_like(p) for p in params]
        self.vs = [np.zeros_like(p) for p in params]

    def step(self, grads):
        self.t += 1
        self.ms = [self.b1 * m + (1 - self.b1) * g
                   for m, g in zip(self.ms, grads)]
        self.vs = [self.b2 * v + (1 - self.b2) * (g**2)
                   for v, g in zip(self.vs, grads)]
        lr = self.lr0 * np.sqrt(1 - self.b2**self.t) / (1 - self.b1**self.t)
        for p, m, v in zip(self.params, self.ms, self.vs):
            p += -lr * m / (np.sqrt(v) + self.eps)


class Network:
    """One fitted MLP (what ``train_network`` returns, nautilus/neural.py:10-32)."""

    def __init__(self, coefs, intercepts, n_iter=0, loss_curve=()):
        self.coefs = coefs
        self.intercepts = intercepts
        self.n_iter = n_iter
        self.loss_curve = list(loss_curve)

    def predict(self, x):
        return forward(x, self.coefs, self.intercepts)[-1].ravel()


@threadpool_limits.wrap(limits=1)                       # neural.py:10
def fit_network(x, y, random_state, max_iter=10000, n_iter_no_change=10,
                tol=0.0, batch_size=200, lr=1e-2, permutations=None,
                init=None, hidden=HIDDEN):
    """``MLPRegressor(random_state=i, ...).fit(x, y)`` restated
    (sklearn/_multilayer_perceptron.py:620-760).

    permutations : optional list of per-epoch sample orders; when given they
        replace the MT19937 shuffles (used to drive the HIP kernel and the
        oracle with the same minibatch order).
    init : optional (coefs, intercepts) replacing the Glorot draw.
    hidden : ``hidden_layer_sizes`` (neural.py:79-83 passes it through).
    """
    n, d = x.shape
    y2 = y.reshape(-1, 1)
    coefs, intercepts, rs = glorot_init(d, random_state, hidden)
    if init is not None:
        coefs = [np.array(c, float) for c in init[0]]
        intercepts = [np.array(c, float) for c in init[1]]
    opt = Adam(coefs + intercepts, lr=lr)
    bs = min(batch_size, n)
    sample_idx = np.arange(n, dtype=int)
    best, stale = np.inf, 0
    curve = []
    for epoch in range(max_iter):
        if permutations is not None:
            if epoch >= len(permutations):
                break
            sample_idx = np.asarray(permutations[epoch])
        else:
            sample_idx = epoch_permutation(rs, sample_idx)
        acc = 0.0
        for start in range(0, n, bs):                    # gen_batches
            rows = sample_idx[start:start + bs]
            loss, cg, ig = loss_and_grads(x[rows], y2[rows], coefs,
                                          intercepts)
            acc += loss * len(rows)
            opt.step(cg + ig)
        curve.append(acc / n)
        # _update_no_improvement_count, :819-822 with tol = 0
        if curve[-1] > best - tol:
            stale += 1
        else:
            stale = 0
        if curve[-1] < best:
            best = curve[-1]
        if stale > n_iter_no_change:                     # :755
            break
    return Network(coefs, intercepts, n_iter=len(curve), loss_curve=curve)


class Emulator:
    """nautilus/neural.py:35-187 (train 50-98, predict 100-116)."""

    @classmethod
    def train(cls, x, y, n_networks=4, neural_network_kwargs={}, pool=None):
        self = cls()
        self.mean = np.mean(x, axis=0)
        self.scale = np.std(x, axis=0)
        kw = dict(max_iter=10000, n_iter_no_change=10, tol=0.0,
                  lr=1e-2)
        for key, val in neural_network_kwargs.items():
            if key == 'learning_rate_init':
                kw['lr'] = val
            elif key == 'hidden_layer_sizes':
                kw['hidden'] = tuple(val)
            elif key in kw:
                kw[key] = val
            elif key != 'random_state':
                raise ValueError('oracle emulator does not restate sklearn '
                                 'option %r' % key)
        xs = (x - self.mean) / self.scale
        self.networks = [fit_network(xs, y, i, **kw)
                         for i in range(n_networks)]
        return self

    @classmethod
    def from_weights(cls, mean, scale, nets):
        self = cls()
        self.mean = np.asarray(mean, float)
        self.scale = np.asarray(scale, float)
        self.networks = [Network(c, b) for c, b in nets]
        return self

    def predict(self, x):
        xs = (x - self.mean) / self.scale
        return np.mean([net.predict(xs) for net in self.networks], axis=0)
