"""HDF5 checkpoints in the reference's layout (``Sampler.write`` /
``write_shell_update`` / the resume branch of ``Sampler.__init__``,
nautilus/sampler.py:330-371, 1253-1377, and the ``write`` / ``read`` /
``update`` methods of every bound: bounds/basic.py:100-137, 396-436, 657-711,
bounds/union.py:345-429, bounds/neural.py:127-169, bounds/nautilus.py:
306-380, bounds/periodic.py:74-107, neural.py:118-187).

Group and attribute names are the reference's, so a file is readable by either
implementation.  The device path adds two attributes per sampling bound --
``amd_philox_seed`` / ``amd_philox_offset``, the position of its Philox
proposal stream -- so that a resumed run continues bit for bit; a file
without them (written by the reference) draws a fresh stream key from the
restored generator.

``h5py`` is imported on first use, like in the reference: without it
``filepath=`` raises ImportError.
"""

from pathlib import Path

import numpy as np
import torch

from . import bounds as nb
from .emulator import Network, NeuralNetworkEmulator

SAMPLER_ATTRS = ['n_dim', 'n_live', 'n_update', 'n_like_new_bound',
                 'enlarge_per_dim', 'n_points_min', 'split_threshold',
                 'n_networks', 'n_batch', 'vectorized', 'pass_dict', 'n_like',
                 'explored', '_discard_exploration', 'shell_n',
                 'shell_n_sample', 'shell_n_eff', 'shell_log_l_min',
                 'shell_log_l', 'shell_log_v', 'shell_n_sample_exp',
                 'shell_end_exp', 'n_update_iter', 'n_like_iter']
UPDATE_ATTRS = ['n_like', 'shell_n', 'shell_n_sample', 'shell_n_eff',
                'shell_log_l_min', 'shell_log_l', 'shell_log_v',
                'n_update_iter', 'n_like_iter']
RESUME_ATTRS = ['n_like', 'explored', '_discard_exploration', 'shell_n',
                'shell_n_sample', 'shell_n_eff', 'shell_log_l_min',
                'shell_log_l', 'shell_log_v', 'shell_n_sample_exp',
                'shell_end_exp', 'n_update_iter', 'n_like_iter']


def h5py():
    import h5py as module
    return module


# ---------------------------------------------------------------------------
# proposal stream + FIFO of a sampling bound
# ---------------------------------------------------------------------------

def _write_stream(bound, group):
    group.attrs['amd_philox_seed'] = str(bound._stream.seed)
    group.attrs['amd_philox_offset'] = str(bound._stream.offset)


def _read_stream(bound, group, rng):
    bound.rng = np.random.default_rng() if rng is None else rng
    stream = nb._PhiloxStream.__new__(nb._PhiloxStream)
    if 'amd_philox_seed' in group.attrs:
        stream.seed = int(group.attrs['amd_philox_seed'])
        stream.offset = int(group.attrs['amd_philox_offset'])
    else:
        stream.rekey(bound.rng)
    bound._stream = stream


def _set_queue(bound, rows):
    bound._fifo = None
    rows = np.asarray(rows, dtype=float).reshape(-1, int(bound.n_dim))
    if len(rows):
        bound._queue().push(torch.from_numpy(np.ascontiguousarray(rows)).cuda())


# ---------------------------------------------------------------------------
# bounds
# ---------------------------------------------------------------------------

def write_bound(bound, group):
    if isinstance(bound, nb.UnitCube):
        group.attrs['type'] = 'UnitCube'
        group.attrs['n_dim'] = bound.n_dim
        _write_stream(bound, group)
    elif isinstance(bound, nb.Ellipsoid):
        group.attrs['type'] = 'Ellipsoid'
        for key in ['n_dim', 'c', 'A', 'B', 'B_inv']:
            group.attrs[key] = getattr(bound, key)
        _write_stream(bound, group)
    elif isinstance(bound, nb.UnitCubeEllipsoidMixture):
        group.attrs['type'] = 'UnitCubeEllipsoidMixture'
        group.attrs['n_dim'] = bound.n_dim
        group.create_dataset('dim_cube', data=bound.dim_cube)
        if bound.cube is not None:
            write_bound(bound.cube, group.create_group('cube'))
        if bound.ellipsoid is not None:
            write_bound(bound.ellipsoid, group.create_group('ellipsoid'))
        _write_stream(bound, group)
    elif isinstance(bound, nb.Union):
        group.attrs['type'] = 'MultiEllipsoid'
        for key in ['n_dim', 'log_v_all', 'enlarge_per_dim', 'n_points_min',
                    'n_sample', 'n_reject']:
            group.attrs[key] = getattr(bound, key)
        group.attrs['unit'] = bound.cube is not None
        if bound.cube is not None:
            write_bound(bound.cube, group.create_group('cube'))
        group.attrs['bound_class'] = bound.bounds[0].__class__.__name__
        for i, member in enumerate(bound.bounds):
            write_bound(member, group.create_group('bound_{}'.format(i)))
        for i, pts in enumerate(bound.points_bounds):
            group.create_dataset('points_bound_{}'.format(i), data=pts)
        group.create_dataset('points', data=bound.points,
                             maxshape=(None, bound.n_dim))
        group.attrs['amd_block'] = np.asarray(bound.block, dtype=bool)
        _write_stream(bound, group)
    elif isinstance(bound, nb.NeuralBound):
        group.attrs['n_dim'] = bound.n_dim
        group.attrs['score_predict_min'] = bound.score_predict_min
        write_bound(bound.outer_bound, group.create_group('outer_bound'))
        if bound.emulator is not None:
            write_emulator(bound.emulator, group.create_group('emulator'))
    elif isinstance(bound, nb.NautilusBound):
        group.attrs['type'] = 'NautilusBound'
        group.attrs['n_dim'] = bound.n_dim
        if bound.shift is not None:
            write_bound(bound.shift, group.create_group('shift'))
        group.attrs['n_neural_bounds'] = len(bound.neural_bounds)
        for i, neural in enumerate(bound.neural_bounds):
            write_bound(neural, group.create_group(
                'neural_bound_{}'.format(i)))
        write_bound(bound.outer_bound, group.create_group('outer_bound'))
        group.create_dataset('points', data=_file_points(bound),
                             maxshape=(None, bound.n_dim))
        if bound.shift is not None:
            # the device queue itself (sampler frame): a resumed run of THIS
            # implementation continues bit for bit, which a shift / unshift
            # round trip through 'points' cannot guarantee
            group.create_dataset('amd_points', data=bound.points,
                                 maxshape=(None, bound.n_dim))
        group.attrs['n_sample'] = bound.n_sample
        group.attrs['n_reject'] = bound.n_reject
        _write_stream(bound, group)
    elif isinstance(bound, nb.PhaseShift):
        group.attrs['type'] = 'PhaseShift'
        group.attrs['periodic'] = bound.periodic
        group.attrs['centers'] = bound.centers
    else:
        raise TypeError('cannot write {}'.format(type(bound).__name__))


def _file_points(bound):
    """Queued points of a bound in the frame the reference stores them in:
    a NautilusBound with periodic parameters keeps ``self.points`` in the
    SHIFTED frame and undoes the shift when it hands points out
    (bounds/nautilus.py:239-243), while the device queue holds sampler-frame
    rows (bounds.py ``_fill``).  Forward shift on the way to the file."""
    pts = bound.points
    shift = getattr(bound, 'shift', None)
    if shift is not None and len(pts) > 0:
        pts = shift.transform(pts)
    return pts


def _queue_points(bound, pts):
    """File frame -> sampler frame (inverse of ``_file_points``)."""
    shift = getattr(bound, 'shift', None)
    if shift is not None and len(pts) > 0:
        pts = shift.transform(pts, inverse=True)
    return pts


def update_bound(bound, group):
    """bounds/union.py:374-385, bounds/nautilus.py:328-342."""
    group.attrs['n_sample'] = bound.n_sample
    group.attrs['n_reject'] = bound.n_reject
    if isinstance(bound, nb.NautilusBound):
        update_bound(bound.outer_bound, group['outer_bound'])
    pts = _file_points(bound)
    group['points'].resize(pts.shape)
    group['points'][...] = pts
    if 'amd_points' in group:
        raw = bound.points
        group['amd_points'].resize(raw.shape)
        group['amd_points'][...] = raw
    _write_stream(bound, group)


def read_bound(cls, group, rng=None):
    bound = cls.__new__(cls)
    if cls is nb.UnitCube:
        bound.n_dim = int(group.attrs['n_dim'])
        _read_stream(bound, group, rng)
    elif cls is nb.Ellipsoid:
        bound.n_dim = int(group.attrs['n_dim'])
        for key in ['c', 'A', 'B', 'B_inv']:
            setattr(bound, key, np.array(group.attrs[key], dtype=float))
        _read_stream(bound, group, rng)
    elif cls is nb.UnitCubeEllipsoidMixture:
        bound.n_dim = int(group.attrs['n_dim'])
        bound.dim_cube = np.array(group['dim_cube'], dtype=bool)
        _read_stream(bound, group, rng)
        bound.cube = (read_bound(nb.UnitCube, group['cube'], bound.rng)
                      if np.any(bound.dim_cube) else None)
        bound.ellipsoid = (read_bound(nb.Ellipsoid, group['ellipsoid'],
                                      bound.rng)
                           if not np.all(bound.dim_cube) else None)
    elif cls is nb.Union:
        bound.n_dim = int(group.attrs['n_dim'])
        bound.log_v_all = np.array(group.attrs['log_v_all'], dtype=float)
        bound.enlarge_per_dim = float(group.attrs['enlarge_per_dim'])
        bound.n_points_min = int(group.attrs['n_points_min'])
        _read_stream(bound, group, rng)
        bound.n_sample = int(group.attrs['n_sample'])
        bound.n_reject = int(group.attrs['n_reject'])
        bound.cube = (read_bound(nb.UnitCube, group['cube'], bound.rng)
                      if group.attrs['unit'] else None)
        member = (nb.Ellipsoid if group.attrs['bound_class'] == 'Ellipsoid'
                  else nb.UnitCubeEllipsoidMixture)
        k = len(bound.log_v_all)
        bound.bounds = [read_bound(member, group['bound_{}'.format(i)],
                                   bound.rng) for i in range(k)]
        bound.points_bounds = [np.array(group['points_bound_{}'.format(i)])
                               for i in range(k)]
        if 'amd_block' in group.attrs:
            bound.block = np.array(group.attrs['amd_block'], dtype=bool)
        else:
            bound.block = np.array([len(p) < 2 * bound.n_points_min
                                    for p in bound.points_bounds])
        _set_queue(bound, np.array(group['points']))
    elif cls is nb.NeuralBound:
        bound.n_dim = int(group.attrs['n_dim'])
        bound.score_predict_min = float(group.attrs['score_predict_min'])
        bound.outer_bound = read_bound(nb.Ellipsoid, group['outer_bound'],
                                       rng)
        bound.emulator = (read_emulator(group['emulator'])
                          if 'emulator' in group else None)
    elif cls is nb.NautilusBound:
        bound.n_dim = int(group.attrs['n_dim'])
        _read_stream(bound, group, rng)
        bound.shift = (read_bound(nb.PhaseShift, group['shift'])
                       if 'shift' in group else None)
        bound.neural_bounds = []
        while 'neural_bound_{}'.format(len(bound.neural_bounds)) in group:
            bound.neural_bounds.append(read_bound(
                nb.NeuralBound,
                group['neural_bound_{}'.format(len(bound.neural_bounds))],
                bound.rng))
        bound.outer_bound = read_bound(nb.Union, group['outer_bound'],
                                       bound.rng)
        # the composite draws through its own pipeline (bounds.py)
        bound.outer_bound._queue().clear()
        bound.n_sample = int(group.attrs['n_sample'])
        bound.n_reject = int(group.attrs['n_reject'])
        file_pts = np.array(group['points'], dtype=float)
        raw = (np.array(group['amd_points'], dtype=float)
               if 'amd_points' in group and 'amd_philox_seed' in group.attrs
               else None)
        # 'amd_points' is only trusted while it still describes 'points': the
        # reference's update() (bounds/nautilus.py:328-342) rewrites 'points'
        # alone when it continues a file written here
        if (raw is not None and raw.shape == file_pts.shape and
                (len(raw) == 0 or np.allclose(
                    bound.shift.transform(raw) if bound.shift is not None
                    else raw, file_pts, rtol=0, atol=1e-12))):
            _set_queue(bound, raw)
        else:                            # written / continued by the reference
            _set_queue(bound, _queue_points(bound, file_pts))
    elif cls is nb.PhaseShift:
        bound.periodic = np.array(group.attrs['periodic'])
        bound.centers = np.array(group.attrs['centers'], dtype=float)
    else:
        raise TypeError('cannot read {}'.format(cls.__name__))
    return bound


# ---------------------------------------------------------------------------
# emulator (neural.py:118-187)
# ---------------------------------------------------------------------------

# what MLPRegressor.__dict__ holds after the reference's fit (neural.py:79-88
# defaults; scikit-learn 1.7) and neural.py:128-137 therefore writes as
# attributes: constructor parameters first, fitted state after them
_SK_PARAMS = dict(
    activation='relu', solver='adam', alpha=0, batch_size='auto',
    learning_rate='constant', learning_rate_init=1e-2, power_t=0.5,
    max_iter=10000, loss='squared_error', hidden_layer_sizes=(100, 50, 20),
    shuffle=True, random_state=None, tol=0, verbose=False, warm_start=False,
    momentum=0.9, nesterovs_momentum=True, early_stopping=False,
    validation_fraction=0.1, beta_1=0.9, beta_2=0.999, epsilon=1e-8,
    n_iter_no_change=10, max_fun=15000)
_SK_KWARGS = dict(learning_rate_init='learning_rate_init', beta_1='beta_1',
                  beta_2='beta_2', epsilon='epsilon', batch_size='batch_size',
                  max_iter='max_iter', n_iter_no_change='n_iter_no_change',
                  tol='tol')


def _network_attrs(net, index):
    """Attribute dict of network ``index`` as the reference writes it."""
    attrs = dict(_SK_PARAMS)
    attrs['random_state'] = index                    # neural.py:31
    attrs.update(getattr(net, 'sk_params', {}))
    curve = np.asarray(net.loss_curve_, dtype=float)
    attrs.update(
        n_features_in_=int(np.shape(net.coefs_[0])[0]), n_outputs_=1,
        n_iter_=int(net.n_iter_),
        t_=int(getattr(net, 't_', 0)), n_layers_=int(net.n_layers_),
        out_activation_='identity', loss_curve_=curve,
        _no_improvement_count=int(getattr(
            net, '_no_improvement_count',
            len(curve) - 1 - int(np.argmin(curve)) if len(curve) else 0)),
        best_loss_=float(np.min(curve)) if len(curve) else np.inf,
        loss_=float(curve[-1]) if len(curve) else np.inf)
    attrs.update(getattr(net, 'sk_state', {}))
    return attrs


def write_emulator(emu, group):
    group.attrs['n_networks'] = len(emu.neural_networks)
    for i, net in enumerate(emu.neural_networks):
        # everything the reference's loop over MLPRegressor.__dict__ writes
        # (what has no HDF5 equivalent is skipped there as well), so that the
        # reference loads the file into MLPRegressor objects
        for key, val in _network_attrs(net, i).items():
            try:
                group.attrs['{}_{}'.format(key, i)] = val
            except (TypeError, ValueError):
                pass
        for k in range(net.n_layers_ - 1):
            group.create_dataset('coefs_{}_{}'.format(k, i),
                                 data=net.coefs_[k])
            group.create_dataset('intercepts_{}_{}'.format(k, i),
                                 data=net.intercepts_[k])
    group.create_dataset('mean', data=emu.mean)
    group.create_dataset('scale', data=emu.scale)


def read_emulator(group):
    nets = []
    for i in range(int(group.attrs['n_networks'])):
        n_layers = int(group.attrs['n_layers__{}'.format(i)])
        coefs = [np.array(group['coefs_{}_{}'.format(k, i)])
                 for k in range(n_layers - 1)]
        icpts = [np.array(group['intercepts_{}_{}'.format(k, i)])
                 for k in range(n_layers - 1)]
        # the remaining attributes of network i (neural.py:176-178) travel
        # along unchanged, so that a later write() reproduces the file
        stored = {}
        for key in group.attrs.keys():
            name, _, idx = key.rpartition('_')
            if idx == str(i) and name:
                stored[name] = group.attrs[key]
        net = Network(coefs, icpts, int(stored.get('n_iter_', 0)),
                      stored.get('loss_curve_'))
        net.sk_params = {k: stored[k] for k in _SK_PARAMS if k in stored}
        net.sk_state = {k: stored[k] for k in
                        ('t_', '_no_improvement_count', 'best_loss_', 'loss_')
                        if k in stored}
        nets.append(net)
    return NeuralNetworkEmulator.from_weights(
        np.array(group['mean']), np.array(group['scale']), nets)


# ---------------------------------------------------------------------------
# sampler
# ---------------------------------------------------------------------------

def _write_rng(sampler, group):
    state = sampler.rng.bit_generator.state
    group.attrs['rng_state'] = str(state['state']['state'])
    group.attrs['rng_inc'] = str(state['state']['inc'])
    group.attrs['rng_has_uint32'] = state['has_uint32']
    group.attrs['rng_uinteger'] = state['uinteger']


def write_sampler(sampler, filepath, overwrite=False):
    """sampler.py:1253-1332."""
    filepath = Path(filepath)
    if filepath.suffix not in ['.h5', '.hdf5']:
        raise ValueError("File ending must '.h5' or '.hdf5'.")
    lib = h5py()
    if filepath.exists():
        if not overwrite:
            raise RuntimeError(
                'File {} already exists.'.format(str(filepath)))
        filepath.unlink()
    filepath.parent.mkdir(parents=True, exist_ok=True)
    fstream = lib.File(filepath, 'x')
    group = fstream.create_group('sampler')
    for key in SAMPLER_ATTRS:
        group.attrs[key] = getattr(sampler, key)
    for key, val in sampler.neural_network_kwargs.items():
        group.attrs['neural_network_{}'.format(key)] = val
    points = sampler.points
    maxshape = None
    for shell in range(len(sampler.bounds)):
        group.create_dataset('points_{}'.format(shell), data=points[shell],
                             maxshape=(None, sampler.n_dim))
        group.create_dataset('log_l_{}'.format(shell),
                             data=sampler.log_l[shell], maxshape=(None, ))
        if sampler.blobs is not None:
            maxshape = (None, ) + tuple(sampler.blobs[shell].shape[1:])
            group.create_dataset('blobs_{}'.format(shell),
                                 data=sampler.blobs[shell], maxshape=maxshape)
    group.create_dataset('points_t', data=sampler.points_t,
                         maxshape=(None, sampler.n_dim))
    group.create_dataset('shell_t', data=sampler.shell_t, maxshape=(None, ))
    group.create_dataset('log_l_t', data=sampler.log_l_t, maxshape=(None, ))
    if sampler.blobs_t is not None:
        group.create_dataset('blobs_t', data=sampler.blobs_t,
                             maxshape=maxshape)
    for i, bound in enumerate(sampler.bounds):
        write_bound(bound, fstream.create_group('bound_{}'.format(i)))
    _write_rng(sampler, group)
    fstream.close()


def write_shell_update(sampler, filepath, shell):
    """sampler.py:1334-1377."""
    if shell < 0:
        shell = len(sampler.bounds) + shell
    fstream = h5py().File(Path(filepath), 'r+')
    group = fstream['sampler']
    for key in UPDATE_ATTRS:
        group.attrs[key] = getattr(sampler, key)
    pts = sampler._pts[shell].view().cpu().numpy()
    group['points_{}'.format(shell)].resize(pts.shape)
    group['points_{}'.format(shell)][...] = pts
    group['log_l_{}'.format(shell)].resize(sampler.log_l[shell].shape)
    group['log_l_{}'.format(shell)][...] = sampler.log_l[shell]
    if sampler.blobs is not None:
        group['blobs_{}'.format(shell)].resize(sampler.blobs[shell].shape)
        group['blobs_{}'.format(shell)][...] = sampler.blobs[shell]
    for key in ['points_t', 'shell_t', 'log_l_t', 'blobs_t']:
        val = getattr(sampler, key)
        if val is not None:
            group[key].resize(val.shape)
            group[key][...] = val
    if isinstance(sampler.bounds[shell], nb.NautilusBound):
        update_bound(sampler.bounds[shell],
                     fstream['bound_{}'.format(shell)])
    else:
        _write_stream(sampler.bounds[shell],
                      fstream['bound_{}'.format(shell)])
    _write_rng(sampler, group)
    fstream.close()


def read_sampler(sampler, filepath):
    """The resume branch of Sampler.__init__ (sampler.py:330-371); fills the
    freshly constructed ``sampler``."""
    from .sampler import _Grow
    with h5py().File(Path(filepath), 'r') as fstream:
        group = fstream['sampler']
        sampler.rng.bit_generator.state = dict(
            bit_generator='PCG64',
            state=dict(state=int(group.attrs['rng_state']),
                       inc=int(group.attrs['rng_inc'])),
            has_uint32=int(group.attrs['rng_has_uint32']),
            uinteger=int(group.attrs['rng_uinteger']))
        for key in RESUME_ATTRS:
            val = group.attrs[key]
            if key in ('n_like', 'n_update_iter', 'n_like_iter'):
                val = int(val)
            elif key in ('explored', '_discard_exploration'):
                val = bool(val)
            else:
                val = np.array(val)
            setattr(sampler, key, val)
        sampler._pts, sampler._ll_dev, sampler.log_l = [], [], []
        for shell in range(len(sampler.shell_n)):
            pts = np.array(group['points_{}'.format(shell)], dtype=float)
            log_l = np.array(group['log_l_{}'.format(shell)], dtype=float)
            grow_p, grow_l = _Grow(sampler.n_dim), _Grow()
            if len(log_l):
                grow_p.append(torch.from_numpy(
                    np.ascontiguousarray(pts)).cuda())
                grow_l.append(torch.from_numpy(log_l).cuda())
            sampler._pts.append(grow_p)
            sampler._ll_dev.append(grow_l)
            sampler.log_l.append(log_l)
            if 'blobs_{}'.format(shell) in group:
                if shell == 0:
                    sampler.blobs = []
                sampler.blobs.append(np.array(
                    group['blobs_{}'.format(shell)]))
                if shell == 0:
                    sampler.blobs_dtype = sampler.blobs[-1].dtype
        if 'shell_t' in group:
            sampler.shell_t = np.array(group['shell_t'])
        if 'log_l_t' in group:
            sampler.log_l_t = np.array(group['log_l_t'])
        if 'points_t' in group:
            sampler._pts_t = torch.from_numpy(np.ascontiguousarray(np.array(
                group['points_t'], dtype=float).reshape(
                    -1, sampler.n_dim))).cuda()
        if 'blobs_t' in group:
            sampler.blobs_t = np.array(group['blobs_t'])
        sampler.bounds = [read_bound(nb.UnitCube, fstream['bound_0'],
                                     rng=sampler.rng)]
        for i in range(1, len(sampler.shell_n)):
            sampler.bounds.append(read_bound(
                nb.NautilusBound, fstream['bound_{}'.format(i)],
                rng=sampler.rng))
    sampler._later = {}
