"""CPU oracle for the D3GA deform-and-rasterize hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``d3ga_amd/`` or ``compat/`` may import
this package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle.deform`` / ``oracle.camera`` -- PINNED against golden vectors captured
    from the reference's own Python (``tools/gen_golden.py`` -> ``tests/golden/*.npz``).
  * ``oracle.raster_torch`` / ``oracle/raster_c`` -- PARITY UNPINNED: the reference's
    rasterizer (graphdeco-inria/diff-gaussian-rasterization, branch ``dr_aa``, SHA not
    recorded in /root/reference/.gitmodules:9-12) is an un-vendored CUDA submodule with no
    tests or golden vectors in the reference tree.  These restate the published 3DGS
    algorithm (Kerbl et al. 2023) and are anchored on the reference's call site
    (renderer.py:79-141) plus known-answer tests.  Since round 2 the EWA Jacobian (off-axis, anisotropic),
    the 1.3 tan(fov/2) guard band and the radius rule are additionally pinned from FIRST PRINCIPLES
    (tests/test_known_answers.py: a float64 NUMERICAL Jacobian of the reference's own camera map -- no code shared
    with either restatement), and ``raster_c.margins`` reports where the algorithm's two discontinuities
    (alpha < 1/255, T(1-alpha) < 1e-4) sit within float32 noise of their thresholds, so that the parity tests can
    hold everything else to the strict bars with no allowance (tests/util.py:Parity).
  * ``oracle.bary`` -- convention pinned by submodules/tetrahedralize/include/tet/tetrahedron.h:46-101;
    tetra_sampler.compute_bary itself is un-vendored (parity unpinned for out-of-cage points).
"""
