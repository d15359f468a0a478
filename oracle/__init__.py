"""CPU oracle for the AP-VAST hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker (or as the timed CPU baseline), never
as the thing measured or shipped.  The product path (``ap_vast_unofficial_amd``)
calls the HIP library through its C ABI and raises if that library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imported the reference's
``Python/apvast.py`` in the build container and wrote ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every oracle function against them.
The MATLAB dialect and ``perceptual=True`` (third-party ``libdetectability``,
unpinned, absent) are NOT pinned -- see DESIGN.md.
"""
