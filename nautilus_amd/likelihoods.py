"""Device likelihoods for the BASELINE benchmark problems (SURVEY.md section 8
row A15): callables marked ``device = True`` receive the batch as a cuda
tensor of unit-cube points and return a cuda tensor of log-likelihoods, so the
shell-filling loop never leaves the GPU.  Any other callable handed to
``Sampler`` is evaluated on the host exactly as in the reference.
"""

import numpy as np
import torch

from . import device


def unit_prior(x):
    """Identity prior transform on the unit cube (works on numpy arrays and
    cuda tensors)."""
    return x


unit_prior.device = True


class GaussianLikelihood:
    """Multivariate normal log-density  -1/2 (x-mu)^T Sigma^-1 (x-mu) + const.

    The quadratic form is the same lower-triangular contraction as
    ``Ellipsoid.contains`` (|L^-1 (x - mu)|^2 with Sigma = L L^T) and runs on
    the matrix cores through ``nb_neural_score``."""

    device = True

    def __init__(self, mean, cov, normalised=True):
        mean = np.asarray(mean, float)
        cov = np.atleast_2d(np.asarray(cov, float))
        d = len(mean)
        if cov.shape == (1, 1) and d > 1:
            cov = np.eye(d) * cov[0, 0]
        chol = np.linalg.cholesky(cov)
        self.n_dim = d
        self.mean, self.cov = mean, cov
        self.log_norm = (-0.5 * (d * np.log(2 * np.pi) +
                                 2 * np.sum(np.log(np.diag(chol))))
                         if normalised else 0.0)
        self._chol = chol
        self._dev = None

    def _bound(self):
        if self._dev is None:
            self._dev = device.DeviceBound(
                self.n_dim, [], None, False,
                [dict(ellipsoid=device.member(self.mean, self._chol))])
        return self._dev

    def __call__(self, x):
        r2, _ = self._bound().neural_score(x)
        out = self.log_norm - 0.5 * r2
        return out if isinstance(x, torch.Tensor) else out.cpu().numpy()

    def numpy(self, x):
        """Pure-numpy evaluation (CPU baseline / oracle runs)."""
        y = np.linalg.solve(self._chol, (np.atleast_2d(x) - self.mean).T)
        return self.log_norm - 0.5 * np.sum(y**2, axis=0)

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev'] = None
        return state


class GaussianMixtureLikelihood:
    """Equal-weight mixture of isotropic Gaussians (BASELINE config 4)."""

    device = True

    def __init__(self, means, sigma):
        self.means = np.atleast_2d(np.asarray(means, float))
        self.sigma = float(sigma)
        d = self.means.shape[1]
        self.parts = [GaussianLikelihood(m, np.eye(d) * sigma**2)
                      for m in self.means]

    def __call__(self, x):
        xs = device.as_device_points(x)
        stack = torch.stack([p(xs) for p in self.parts])
        out = torch.logsumexp(stack, dim=0) - np.log(len(self.parts))
        return out if isinstance(x, torch.Tensor) else out.cpu().numpy()

    def numpy(self, x):
        from scipy.special import logsumexp
        return logsumexp([p.numpy(x) for p in self.parts], axis=0) - \
            np.log(len(self.parts))


class RosenbrockLikelihood:
    """Rosenbrock function on x = low + (high - low) u (BASELINE config 3):
    log L = -sum_i [a (x_{i+1} - x_i^2)^2 + (1 - x_i)^2] --
    ``nb_loglike_rosenbrock``."""

    device = True

    def __init__(self, n_dim, low=-5.0, high=5.0, a=100.0):
        self.n_dim = int(n_dim)
        self.low, self.high, self.a = float(low), float(high), float(a)

    def __call__(self, x):
        from . import _lib
        lib = _lib.load()
        xs = device.as_device_points(x, self.n_dim)
        out = torch.empty(xs.shape[0], dtype=torch.float64, device='cuda')
        _lib.check(lib.nb_loglike_rosenbrock(
            device._ptr(xs), xs.shape[0], self.n_dim, self.low, self.high,
            self.a, device._ptr(out), device._stream()))
        return out if isinstance(x, torch.Tensor) else out.cpu().numpy()

    def numpy(self, x):
        """Pure-numpy evaluation (CPU baseline / oracle runs / tests)."""
        x = self.low + (self.high - self.low) * np.atleast_2d(x)
        return -np.sum(self.a * (x[:, 1:] - x[:, :-1]**2)**2 +
                       (1.0 - x[:, :-1])**2, axis=1)


class FunnelLikelihood:
    """Neal's funnel in n_dim dimensions on the unit cube (BASELINE config 5;
    the reference's tests/test_sampler.py:311-314 has the 2-D case):
    x_0 ~ N(mu, sigma0^2), x_i ~ N(mu, (exp(k (x_0 - mu)) / c)^2) for i > 0 --
    ``nb_loglike_funnel``."""

    device = True

    def __init__(self, n_dim, mu=0.5, sigma0=0.1, k=20.0, c=100.0):
        self.n_dim = int(n_dim)
        self.mu, self.sigma0 = float(mu), float(sigma0)
        self.k, self.c = float(k), float(c)

    def __call__(self, x):
        from . import _lib
        lib = _lib.load()
        xs = device.as_device_points(x, self.n_dim)
        out = torch.empty(xs.shape[0], dtype=torch.float64, device='cuda')
        _lib.check(lib.nb_loglike_funnel(
            device._ptr(xs), xs.shape[0], self.n_dim, self.mu, self.sigma0,
            self.k, self.c, device._ptr(out), device._stream()))
        return out if isinstance(x, torch.Tensor) else out.cpu().numpy()

    def numpy(self, x):
        from scipy.stats import norm
        x = np.atleast_2d(x)
        s = np.exp(self.k * (x[:, 0] - self.mu)) / self.c
        return (norm.logpdf(x[:, 0], loc=self.mu, scale=self.sigma0) +
                np.sum(norm.logpdf(x[:, 1:], loc=self.mu,
                                   scale=s[:, None]), axis=1))
