"""Philox tier of the oracle: the reference's accept/reject algorithm fed from
the counter-based streams the HIP kernels use.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

The reference draws from one shared PCG64 generator in chunks of 1000
(nautilus/bounds/union.py:305-323, nautilus/bounds/nautilus.py:214-222).  A
GPU cannot reproduce a sequential generator, so the device path (DESIGN.md
"RNG contract") replaces *where the random numbers come from* but not what is
done with them:

* proposal ``g`` (a 64-bit global index) owns the Philox4x32-10 streams
  ``philox(key=seed, counter=(g_lo, g_hi, block, tag))``;
* the union member is picked per proposal by inverse CDF on
  ``softmax(log_v_all)`` -- the same distribution as the reference's
  ``multinomial`` split followed by ``shuffle`` (union.py:308-315);
* the uniform-in-ellipsoid map, unit-cube clip, overlap count ``k`` and the
  ``u > 1 - 1/k`` acceptance are the reference's (basic.py:376-381,
  union.py:313-319);
* survivors are kept in proposal order (stable compaction), the counters are
  the reference's ``n_sample`` / ``n_reject`` at both levels.

Stream layout per proposal (a Philox call yields four 32-bit words):
    tag 0, block 0 : (u_member, u_accept)        two 53-bit uniforms in [0, 1)
    tag 0, block 1 : (u_radius, unused)
    tag 1, block q : two Box-Muller pairs -> normals 4q .. 4q+3 (ellipsoid
                     columns); every word w is one uniform (w + 1/2) / 2^32
                     in (0, 1): words (0, 1) -> normals 4q, 4q+1, words
                     (2, 3) -> 4q+2, 4q+3
    tag 2, block j : uniforms for cube columns 2j, 2j+1 (53 bits each)
(Round 4: the normals took a 53-bit uniform per word pair before, one Philox
call per Box-Muller pair; the integer rounds were half of the draw kernel.)
"""

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

TAG_CTRL, TAG_NORMAL, TAG_CUBE = 0, 1, 2


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-10 (Salmon et al. 2011).  Counter words are uint32 arrays
    (broadcastable), key words python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK
                      for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def to_unit(hi, lo):
    """53-bit uniform in [0, 1) from two 32-bit words."""
    return ((hi >> np.uint32(5)).astype(np.float64) * 67108864.0 +
            (lo >> np.uint32(6)).astype(np.float64)) / 9007199254740992.0


def uniform_pair(seed, g, block, tag):
    """Two uniforms for proposal index array ``g`` (uint64)."""
    g = np.asarray(g, dtype=np.uint64)
    w = philox4x32(g & MASK, g >> np.uint64(32), block, tag,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return to_unit(w[0], w[1]), to_unit(w[2], w[3])


def unit32(w):
    """Uniform in (0, 1) from one 32-bit word: (w + 1/2) / 2^32."""
    return (w.astype(np.float64) + 0.5) / 4294967296.0


def normal_quad(seed, g, block):
    """Two Box-Muller pairs from block ``block`` of the tag-1 stream: the
    normals 4 block .. 4 block + 3 of proposals ``g``."""
    g = np.asarray(g, dtype=np.uint64)
    w = philox4x32(g & MASK, g >> np.uint64(32), block, TAG_NORMAL,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = []
    for a, b in ((w[0], w[1]), (w[2], w[3])):
        r = np.sqrt(-2.0 * np.log(unit32(a)))
        t = 2.0 * np.pi * unit32(b)
        out += [r * np.cos(t), r * np.sin(t)]
    return out


def member_cdf(log_v_all):
    """Inverse-CDF table over softmax(log_v_all) (union.py:308)."""
    lv = np.asarray(log_v_all, float)
    p = np.exp(lv - np.max(lv))
    cdf = np.cumsum(p / np.sum(p))
    cdf[-1] = 1.0
    return cdf


def _member_parts(member):
    """(dim_cube mask, ellipsoid or None) of an OEllipsoid / OMixture / OCube."""
    if hasattr(member, 'dim_cube'):
        return member.dim_cube, member.ellipsoid
    if hasattr(member, 'B'):
        return np.zeros(member.n_dim, dtype=bool), member
    return np.ones(member.n_dim, dtype=bool), None


def union_propose(union, seed, offset, n_draw):
    """Raw proposals of the device path for global indices
    ``offset .. offset + n_draw - 1``.

    Returns (x, keep, k): the proposed points, the acceptance flag of the
    outer union (cube clip AND overlap acceptance) and the overlap count.
    """
    g = np.uint64(offset) + np.arange(n_draw, dtype=np.uint64)
    d = union.n_dim
    u_member, u_accept = uniform_pair(seed, g, 0, TAG_CTRL)
    u_radius, _ = uniform_pair(seed, g, 1, TAG_CTRL)
    cdf = member_cdf(union.log_v_all)
    member = np.minimum(np.searchsorted(cdf, u_member, side='right'),
                        len(cdf) - 1)
    x = np.zeros((n_draw, d))
    for m, bound in enumerate(union.bounds):
        rows = np.flatnonzero(member == m)
        if len(rows) == 0:
            continue
        dim_cube, ell = _member_parts(bound)
        gi = g[rows]
        if ell is not None:
            de = ell.n_dim
            z = np.zeros((len(rows), de))
            for q in range((de + 3) // 4):
                for k, col in enumerate(normal_quad(seed, gi, q)):
                    if 4 * q + k < de:
                        z[:, 4 * q + k] = col
            z = z / np.sqrt(np.sum(z**2, axis=1))[:, None]
            z *= (u_radius[rows]**(1.0 / de))[:, None]
            x[np.ix_(rows, np.flatnonzero(~dim_cube))] = \
                ell.transform(z, inverse=True)
        cols = np.flatnonzero(dim_cube)
        for j in range((len(cols) + 1) // 2):
            a, b = uniform_pair(seed, gi, j, TAG_CUBE)
            x[rows, cols[2 * j]] = a
            if 2 * j + 1 < len(cols):
                x[rows, cols[2 * j + 1]] = b
    in_cube = (np.all((x >= 0) & (x < 1), axis=1) if union.cube is not None
               else np.ones(n_draw, dtype=bool))
    k = union.member_count(x)
    with np.errstate(divide='ignore'):
        keep = in_cube & (u_accept > 1 - 1.0 / k)
    return x, keep, k


def union_sample(union, seed, offset, n_draw):
    """Device-path ``Union.sample`` for one launch: survivors in proposal
    order plus the counter increments (n_sample, n_reject)."""
    x, keep, _ = union_propose(union, seed, offset, n_draw)
    return x[keep], n_draw, n_draw - int(np.sum(keep))


def nautilus_sample(bound, seed, offset, n_draw):
    """Device-path ``NautilusBound.sample`` for one launch.

    Returns (points, counters) with counters = [outer n_sample, outer
    n_reject, n_sample, n_reject] increments (nautilus.py:221-222,
    union.py:322-323)."""
    x, n_s, n_r = union_sample(bound.outer_bound, seed, offset, n_draw)
    if len(bound.neural_bounds) > 0 and len(x) > 0:
        ok = bound.neural_contains(x)
    else:
        ok = np.ones(len(x), dtype=bool)
    return x[ok], np.array([n_s, n_r, len(x), len(x) - int(np.sum(ok))],
                           dtype=np.int64)
