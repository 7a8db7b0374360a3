"""CPU oracle for the EVREAL hot path -- TEST INFRASTRUCTURE ONLY.

Every function here is a CPU restatement (numpy / torch-CPU / plain C) of one
piece of the reference's hot path, citing the reference file:line it follows
(paths relative to the reference repo root, ercanburak/EVREAL @ v2).

Rules (see DESIGN.md "Oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from this package;
  * the product path (evreal_amd/) never imports it and never falls back to it;
  * pinning: each restatement is checked in tests/ against golden vectors made by
    importing the real reference in the build container
    (tests/golden/make_golden.py).  The scikit-image (MSE/SSIM) and pyiqa
    (LPIPS) arithmetic is NOT in the reference tree and not installed, so those
    restatements are "parity unpinned" (they follow the published algorithm and
    the kwargs at utils/eval_metrics.py:83,96).
"""
