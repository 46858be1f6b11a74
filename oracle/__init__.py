"""CPU oracle for the DPRNN separator forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as a fallback for the HIP path.

* ``oracle.dprnn_numpy``     -- explicit-formula numpy restatement (fp32 or fp64)
* ``oracle.dprnn_torch_cpu`` -- the same algorithm on stock torch CPU ops
                                (what the reference itself dispatches to); used
                                for full-size cases and as the timed CPU baseline
* ``oracle.weights``         -- version-independent numpy weight recipe shared by
                                the golden-vector generator and the tests
* ``oracle.sisdr``           -- fp64 SI-SDR / PIT restatement

Parity pin: every function here is checked against golden vectors produced by
importing the reference's own model classes (tests/golden/make_golden.py,
run in the authoring container; vectors committed under tests/golden/).
"""
