"""The five BASELINE.json workloads as ready
made (likelihood, sampler settings) pairs: synthetic problems on the unit
cube with an identity prior and device likelihoods, shared by
``examples/run_config.py``, ``bench.py`` and the end-to-end tests."""

import numpy as np

from .likelihoods import (FunnelLikelihood, GaussianLikelihood,
                          GaussianMixtureLikelihood, RosenbrockLikelihood)

NAMES = ('C1', 'C2', 'C3', 'C4', 'C5')


def baseline_config(name):
    """dict(likelihood, n_dim, n_live, n_networks, n_batch, analytic_log_z or
    None, description).  ``n_batch`` is the batch size the runs in
    profiles/ and tests/test_configs_gpu.py use: large where the evidence
    does not depend on it (the Gaussians), the reference's order of magnitude
    for the Rosenbrock function, whose evidence does -- every batch of the
    exploration phase goes to the newest bound, so a batch larger than
    ``n_update`` makes every shell thicker, and the volume an emulator cuts off
    wrongly grows with it (30-D: log Z = -138.34 / -138.23 / -138.31 / -138.72
    at n_batch 100 / 256 / 1024 / 8192, reference at its default 100: -138.22,
    exact -137.49; docs/history/round2.md)."""
    if name == 'C1':
        # README example of the reference: 3-D Gaussian
        return dict(
            likelihood=GaussianLikelihood([0.4, 0.5, 0.6], 0.01 * np.eye(3)),
            n_dim=3, n_live=1000, n_networks=4, n_batch=512,
            analytic_log_z=-6.4e-5,
            description='3-dim multivariate Gaussian (README example), '
                        'n_live=1000')
    if name == 'C2':
        d, s = 20, 0.05
        cov = s**2 * (0.5 * np.ones((d, d)) + 0.5 * np.eye(d))
        return dict(
            likelihood=GaussianLikelihood(np.full(d, 0.5), cov), n_dim=d,
            n_live=2000, n_networks=4, n_batch=256, analytic_log_z=0.0,
            description='20-dim correlated Gaussian, n_live=2000')
    if name == 'C3':
        return dict(
            likelihood=RosenbrockLikelihood(30), n_dim=30, n_live=3000,
            n_networks=4, n_batch=256, analytic_log_z=-137.4875,
            description='30-dim Rosenbrock on [-5, 5]^30, n_live=3000')
    if name == 'C4':
        means = 0.25 + 0.5 * np.random.default_rng(3).random((4, 50))
        return dict(
            likelihood=GaussianMixtureLikelihood(means, 0.02), n_dim=50,
            n_live=5000, n_networks=4, n_batch=16384, analytic_log_z=0.0,
            means=means, description='50-dim 4-mode Gaussian mixture, n_live=5000')
    if name.startswith('C4-D'):
        # the config-4 problem at a dimension the REFERENCE finishes on a CPU
        # core (tests/golden/make_golden_mixture.py runs it with n_live 2000)
        d = int(name[4:])
        means = 0.25 + 0.5 * np.random.default_rng(3).random((4, d))
        return dict(
            likelihood=GaussianMixtureLikelihood(means, 0.02), n_dim=d,
            n_live=2000, n_networks=4, n_batch=100, analytic_log_z=0.0,
            means=means, description='%d-dim 4-mode Gaussian mixture, '
                                     'n_live=2000 (config 4 at a dimension '
                                     'the reference finishes)' % d)
    if name == 'C5':
        return dict(
            likelihood=FunnelLikelihood(100), n_dim=100, n_live=10000,
            n_networks=8, n_batch=8192,
            analytic_log_z=funnel_log_z(100),
            description='100-dim Neal funnel, n_live=10000, n_networks=8')
    if name.startswith('C5-D'):
        # the config-5 problem at a smaller dimension (same likelihood family,
        # networks and live points): the sizes at which a run FINISHES -- the
        # exploration has to walk down the funnel to x_0 ~ 0.27 (1 % of the
        # mass below), where the iso-likelihood region has shrunk by ~7.8 nats
        # per dimension; at ~0.7 nats per bound that is ~100 bounds at D = 10,
        # ~170 at D = 20, ~250 at D = 30 and ~830 at D = 100 (measured: 8.0-8.4
        # bounds per dimension; docs/history/round4.md)
        d = int(name[4:])
        return dict(
            likelihood=FunnelLikelihood(d), n_dim=d, n_live=10000,
            n_networks=8, n_batch=8192, analytic_log_z=funnel_log_z(d),
            description='%d-dim Neal funnel, n_live=10000, n_networks=8 '
                        '(config 5 at a dimension that finishes)' % d)
    raise ValueError('unknown BASELINE configuration %r' % (name,))


def funnel_log_z(n_dim, mu=0.5, sigma0=0.1, k=20.0, c=100.0):
    """Evidence of the n_dim funnel on the unit cube: the likelihood is a
    normalised density, so Z is the probability mass inside the cube -- the
    quantity the reference's own funnel test estimates from 10^6 draws
    (tests/test_sampler.py:316-322), here by quadrature over x_0:
    Z = int_0^1 N(x_0; mu, sigma0) [Phi((1 - mu) / s) - Phi(-mu / s)]^(n_dim-1)
    dx_0 with s = exp(k (x_0 - mu)) / c.  n_dim = 100: -0.07595 (the 10^6-draw
    estimate with the conditional probabilities integrated out: -0.07597)."""
    from scipy.integrate import quad
    from scipy.stats import norm

    def integrand(x0):
        s = np.exp(k * (x0 - mu)) / c
        p = norm.cdf((1.0 - mu) / s) - norm.cdf(-mu / s)
        return norm.pdf(x0, mu, sigma0) * p ** (n_dim - 1)
    z, _ = quad(integrand, 0.0, 1.0, epsabs=1e-13, epsrel=1e-13, limit=500)
    return float(np.log(z))


def funnel_moments(n_dim, mu=0.5, sigma0=0.1, k=20.0, c=100.0):
    """(E[x_0], Var[x_0]) of the funnel posterior on the unit cube, by the same
    quadrature as ``funnel_log_z``: the cube cuts the wide end of the funnel
    (x_0 > mu, where the other coordinates leave the cube), so the mean sits
    below mu -- 0.4895 / 0.4879 / 0.4871 / 0.4862 at n_dim 10 / 20 / 30 / 50."""
    from scipy.integrate import quad
    from scipy.stats import norm

    def moment(p):
        def integrand(x0):
            s = np.exp(k * (x0 - mu)) / c
            q = norm.cdf((1.0 - mu) / s) - norm.cdf(-mu / s)
            return x0**p * norm.pdf(x0, mu, sigma0) * q ** (n_dim - 1)
        return quad(integrand, 0.0, 1.0, epsabs=1e-13, epsrel=1e-13,
                    limit=500)[0]
    z, m1, m2 = moment(0), moment(1), moment(2)
    return float(m1 / z), float(m2 / z - (m1 / z)**2)


def headline_config(n_dim=50):
    """The BASELINE.json ``metric`` case: single-mode 50-D Gaussian,
    mu = 0.5, sigma = 0.05, analytic log Z = 0."""
    return dict(
        likelihood=GaussianLikelihood(np.full(n_dim, 0.5),
                                      np.eye(n_dim) * 0.05**2),
        n_dim=n_dim, n_live=2000, n_networks=4, analytic_log_z=0.0,
        description='%d-dim Gaussian mu=0.5 sigma=0.05' % n_dim)
