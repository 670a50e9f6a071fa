"""CPU oracle for cellulus_amd — TEST INFRASTRUCTURE ONLY.

Plain torch/numpy/C restatements of the reference's algorithms on the hot path,
each citing the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``cellulus_amd``) never does.
"""
