"""CPU oracle for the nautilus shell-filling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may
be imported from ``tests/``, from ``__graft_entry__.smoke()`` and from the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the
reported CPU baseline, never as the thing that is measured or shipped.  The
product package (``nautilus_amd``) never imports it.

The oracle is a plain numpy restatement of the reference algorithm
(johannesulf/nautilus v1.0.6, every function cites the reference file:line it
follows).  It is pinned against golden vectors generated from the reference
itself (``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` is mounted) -- see ``tests/test_oracle_golden.py``.

Two RNG tiers (SURVEY.md section 4 take-away 1):

* ``numpy.random.Generator`` tier -- draws in the reference's order, so the
  golden vectors match bit-for-bit (or to a few ulp where BLAS is involved).
* Philox tier (``oracle.philox``) -- the same accept/reject algorithm fed from
  the counter-based Philox4x32-10 streams the HIP kernels use, so that device
  output can be compared point by point.
"""
