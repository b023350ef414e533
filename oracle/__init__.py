"""CPU oracle for the CFD-surrogate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cfd_amd/`` (the product) may import,
call, link or execute anything in this package.  The only permitted users are
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / reported baseline, never as the
thing that is measured or shipped.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.resample``  -- restates ``pixels_coords_testing.py:151-227`` in numpy
  fp64 and is PINNED: ``tests/golden/resample_*.npz`` were produced by the real
  reference function run in the build container
  (``tests/golden/make_resample_golden.py``) and the restatement matches them
  bit for bit, except on exact-distance ties between different-valued source
  points, where scipy's cKDTree choice is traversal dependent (documented,
  asserted to be the only place a mismatch may occur).
* ``oracle.nets``      -- restates the Keras graphs of ``U/models``, ``P/models``,
  ``DeltaP/models`` in PyTorch-CPU fp32.  PARITY UNPINNED: TensorFlow/Keras is
  not installed in the build container and the reference holds no tests,
  fixtures or golden outputs for these graphs; every Keras default used is
  listed in ``oracle/keras_semantics.py`` so one place needs fixing if a later
  session can run real Keras.
* ``oracle.dataset``   -- restates ``utils.load_data``'s per-case arithmetic; its signed distance
  (``oracle/sdf_fmm.c``, built by ``oracle/build.py``) restates scikit-fmm's published fast-marching
  scheme (scikit-fmm is absent here).  PARITY UNPINNED, checked against analytic shapes and
  hand-computed values of the scheme only.
"""
