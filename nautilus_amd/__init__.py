"""nautilus_amd: the shell-filling hot path of johannesulf/nautilus
(importance nested sampling) on AMD MI355X -- hand-written HIP kernels for
gfx950 behind the reference's ``Sampler`` / ``Prior`` / bound interfaces.

Importing the package does not touch the GPU; the HIP library is loaded on
first use and its absence is an error (there is no CPU fallback).
"""

__version__ = '0.1.0'

_LAZY = {
    'Sampler': 'sampler', 'Prior': 'prior', 'NautilusPool': 'pool',
    'UnitCube': 'bounds', 'Ellipsoid': 'bounds',
    'UnitCubeEllipsoidMixture': 'bounds', 'Union': 'bounds',
    'NeuralBound': 'bounds', 'NautilusBound': 'bounds',
    'PhaseShift': 'bounds',
    'NeuralNetworkEmulator': 'emulator',
    'GaussianLikelihood': 'likelihoods',
    'GaussianMixtureLikelihood': 'likelihoods', 'unit_prior': 'likelihoods',
    'RosenbrockLikelihood': 'likelihoods', 'FunnelLikelihood': 'likelihoods',
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module('.' + _LAZY[name], __name__)
        return getattr(mod, name)
    raise AttributeError(name)


__all__ = sorted(_LAZY)
