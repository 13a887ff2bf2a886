"""CPU oracle for the per-tile preprocess -> decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is a product path: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product (``merfish3d-analysis_amd``) never imports this package and fails
loudly when its HIP library is missing.

Each module is a NumPy/SciPy restatement of the reference algorithm
(QI2lab/merfish3d-analysis v0.13.0), citing the reference ``file:line`` it
follows.  Pinning status (see DESIGN.md "Oracle pinning"):

* ``rlgc_ref``       - pinned bit-for-bit against the reference's own
                       ``utils/rlgc.py`` executed in the authoring container
                       with NumPy standing in for CuPy
                       (``tests/golden/make_golden.py`` -> ``rlgc_*.npz``).
* ``decode_ref``     - pixel decode pinned the same way
                       (``decode_pixels_*.npz``); the distance step restates
                       cuVS' expanded euclidean form - third-party arithmetic,
                       "parity unpinned" at the ulp level.
* ``registration_ref`` - warps pinned against the reference executed with
                       ``scipy.ndimage`` standing in for ``cupyx.scipy.ndimage``;
                       phase correlation restates scikit-image's published
                       algorithm and is pinned by the reference tests'
                       known-answer shifts only.
* ``sofima_ref``     - reference-owned glue pinned by direct import; the
                       third-party SOFIMA estimator arithmetic is
                       "parity unpinned" (tolerance pins only).
* ``ccl_ref``        - scikit-image label/regionprops semantics restated;
                       "parity unpinned" (no reference test, library absent).
"""
