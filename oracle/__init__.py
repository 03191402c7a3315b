"""CPU oracle for the two FEABAS hot paths (NCC matcher, FEM relaxation).

TEST INFRASTRUCTURE ONLY.  This package is a numpy/scipy restatement of the
reference algorithms (each function cites the reference file:line it follows).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- and there only as the checker.  The product
package ``feabas_amd`` never imports it and has no CPU fallback: it fails
loudly when the HIP library is missing.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the reference's own functions (``tests/golden/make_golden.py``
imports ``/root/reference`` in the build container; fixtures are committed as
``tests/golden/*.npz``) by ``tests/test_oracle_golden.py``.

Third-party arithmetic used by the reference and therefore by the oracle:
``scipy.fft`` (pocketfft), ``scipy.ndimage.gaussian_filter1d``,
``scipy.sparse`` -- versions unpinned by the reference (setup.py:10-27);
scipy 1.15.3 / numpy 2.2.6 here.
"""
