"""Oracle: point-in-tetrahedron test and barycentric weights (numpy float64).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates submodules/tetrahedralize/include/tet/tetrahedron.h:46-101 of the reference (same_side,
point_in_tet, barycentric via signed-volume ratios, weights ordered a,b,c,d) -- the only in-tree definition of
the convention that the un-vendored tetra_sampler.compute_bary (lib/cage.py:325-327) must follow for points
inside the cage.  Behaviour for points outside every tet is NOT pinned by the reference (parity unpinned);
this oracle uses "largest minimum weight", the decision recorded in DESIGN.md.
"""
import numpy as np


def _stp(a, b, c):
    return np.einsum("...i,...i->...", a, np.cross(b, c))


def barycentric(a, b, c, d, p):
    """tetrahedron.h:77-101.  a,b,c,d,p (...,3) -> (...,4)."""
    vap, vbp = p - a, p - b
    vab, vac, vad = b - a, c - a, d - a
    vbc, vbd = c - b, d - b
    va6 = _stp(vbp, vbd, vbc)
    vb6 = _stp(vap, vac, vad)
    vc6 = _stp(vap, vad, vab)
    vd6 = _stp(vap, vab, vac)
    v6 = 1.0 / _stp(vab, vac, vad)
    return np.stack([va6 * v6, vb6 * v6, vc6 * v6, vd6 * v6], -1)


def same_side(v1, v2, v3, v4, p):
    """tetrahedron.h:46-57."""
    n = np.cross(v2 - v1, v3 - v1)
    return np.signbit(np.einsum("...i,...i->...", n, v4 - v1)) == np.signbit(np.einsum("...i,...i->...", n, p - v1))


def point_in_tet(v1, v2, v3, v4, p):
    """tetrahedron.h:59-71."""
    return (same_side(v1, v2, v3, v4, p) & same_side(v2, v3, v4, v1, p) & same_side(v3, v4, v1, v2, p)
            & same_side(v4, v1, v2, v3, p))


def compute_bary(points, corners, chunk=256):
    """points (P,3), corners (T,4,3) -> barys (P,4), tetra_id (P,), active (P,) [inside some tet]."""
    points = np.asarray(points, np.float64)
    corners = np.asarray(corners, np.float64)
    P = points.shape[0]
    barys = np.zeros((P, 4))
    tid = np.zeros(P, np.int64)
    best = np.full(P, -np.inf)
    a, b, c, d = (corners[None, :, k] for k in range(4))
    for s in range(0, P, chunk):
        p = points[s:s + chunk, None, :]
        w = barycentric(a, b, c, d, p)                       # (n,T,4)
        mn = w.min(-1)
        j = mn.argmax(1)                                     # first maximum = lowest tet index
        r = np.arange(p.shape[0])
        barys[s:s + chunk] = w[r, j]
        tid[s:s + chunk] = j
        best[s:s + chunk] = mn[r, j]
    return barys, tid, best >= 0
