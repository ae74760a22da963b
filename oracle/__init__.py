"""CPU oracle for the ORV denoising hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package - as the checker, never as the thing measured or shipped.  ``orv_amd`` (the product)
never imports it and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
* ORV-authored logic (oracle/dit.py, oracle/pipeline.py): PINNED by tests/golden/*.safetensors, which
  were generated in the build container by running the reference's own code
  (oracle/ref_harness.py + oracle/gen_golden.py; the reference cannot travel to the GPU box).
* diffusers leaf arithmetic (oracle/leaf.py): PARITY UNPINNED - diffusers (>=0.32 effective) is absent
  from /root/reference and not installable here; restated from its published algorithm.
"""
