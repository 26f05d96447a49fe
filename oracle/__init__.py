"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU fp32 / numpy / plain C) of the reference algorithms on the
LiDARCrafter denoising hot path.  Only tests/, __graft_entry__.smoke() and the `cpu_baseline`
leg of bench.py may import anything from here, and only as the checker / reported baseline --
never as the thing measured or shipped.  The product path (lidarcrafter_amd, lidargen) never
imports this package and has no CPU fallback.

Pinning: the reference ships NO tests / golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
tests/golden/make_fixtures.py (imports /root/reference read-only) and committed under
tests/golden/*.npz.  tests/test_oracle_vs_golden.py checks every function here against them.
The RoI voxel-pool restatement (oracle/roipool.py) follows the CUDA kernel text and has no
runnable reference here: "parity unpinned" for that function only.
"""
