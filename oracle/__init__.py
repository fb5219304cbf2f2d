"""TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is the checker for the HIP forward path:

* ``ref_loader``   - imports the *real* reference from /root/reference (only in
                     the build container; the GPU box has no /root/reference).
* ``weights``      - deterministic, torch-RNG-free state_dict generator with the
                     reference's exact key names / shapes.
* ``fsnp_numpy``   - dtype-generic numpy restatement (fp32 and the fp64 yard-stick).
* ``fsnp_torch``   - torch-CPU restatement (same ATen/oneDNN kernels the reference
                     runs on) used for parity and as bench.py's ``cpu_baseline``.
* ``make_golden``  - generates tests/golden/*.npz from the real reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (fullsubnet_plus_amd/) never does.
"""
