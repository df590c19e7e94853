"""TEST INFRASTRUCTURE ONLY -- the parity oracle for the sampling hot path.

Nothing in ``cleandiffuser_amd`` may import from here.  Allowed callers: ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg (as the thing that *checks* / is timed beside the product, never as
the product).  Contents:

* ``ref_import.py``  -- imports the real reference from /root/reference (build container only; absent on GPU boxes)
* ``gen_golden.py``  -- runs the imported reference and writes ``tests/golden/*.npz`` (committed fixtures)
* ``torch_port.py``  -- functional CPU restatement of the reference algorithm (ATen fp32 ops, state_dict driven)
* ``c/``             -- plain-C restatement of the JannerUNet1d forward + solver step (no ATen at all)

Parity pinning: the reference's own tests hold no golden numbers for this path, so the oracle is
pinned against outputs of the reference itself, generated here by ``gen_golden.py`` and committed under tests/golden/.
"""
