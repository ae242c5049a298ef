"""`tetra_sampler`-compatible container and point location (lib/cage.py:17,299-346 of the reference use
`Tetra(path)`, `.points/.tetras/.triangles/.tetra_faces/.triangle_to_tetra`, `.gradient(x)`, `.get_triangles(v)`,
`.n()` and `compute_bary(points, tetras (T,4,3), triangles, tri_to_tetra, cage)`).

The original package (github.com/Zielon/sampler) is an un-vendored submodule; conventions that nothing in the
reference pins are decided here and documented in DESIGN.md:
  * `gradient` returns the edge vectors (v3-v0, v2-v0, v1-v0) as COLUMNS (the in-tree analogue lib/tet_mesh.py:88-94),
    so that Ds @ inv(Dm) is the deformation gradient (a rigid cage rotation R gives J = R);
  * a point outside every tet is assigned the tet with the largest minimum barycentric weight.
"""
import numpy as np
import math

import torch

from . import _lib
from ._lib import check, dptr, require_cuda, stream_handle


def read_medit_mesh(path):
    """Minimal ASCII Medit .mesh reader: returns (vertices (V,3) f32, triangles (F,3) i64, tetrahedra (T,4) i64),
    0-based.  (The reference reads the same file through meshio, lib/tet_mesh.py:18-32.)"""
    with open(path, "r") as f:
        tok = f.read().split()
    out = {"Vertices": (3, np.float64), "Triangles": (3, np.int64), "Tetrahedra": (4, np.int64)}
    data = {}
    i = 0
    while i < len(tok):
        t = tok[i]
        if t in out:
            width, dt = out[t]
            n = int(tok[i + 1])
            raw = np.asarray(tok[i + 2:i + 2 + n * (width + 1)], dtype=np.float64).reshape(n, width + 1)
            data[t] = raw[:, :width].astype(dt)
            i += 2 + n * (width + 1)
        elif t == "End":
            break
        else:
            i += 1
    verts = data.get("Vertices", np.zeros((0, 3))).astype(np.float32)
    tris = data.get("Triangles", np.zeros((0, 3), np.int64)) - 1
    tets = data.get("Tetrahedra", np.zeros((0, 4), np.int64)) - 1
    return verts, tris, tets


def boundary_triangles(tets):
    """Faces that belong to exactly one tet, and the tet each belongs to."""
    combos = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    faces = tets[:, combos].reshape(-1, 3)
    owner = np.repeat(np.arange(tets.shape[0]), 4)
    key = np.sort(faces, axis=1)
    _, inv, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    mask = counts[inv.reshape(-1)] == 1
    return faces[mask], owner[mask]


class Tetra:
    def __init__(self, path=None, points=None, tetras=None, device="cuda"):
        if path is not None:
            v, tris, t = read_medit_mesh(path)
        else:
            v, t = np.asarray(points, np.float32), np.asarray(tetras, np.int64)
        tri, owner = boundary_triangles(t)
        dev = torch.device(device)
        self.points = torch.from_numpy(v).to(dev)
        self.tetras = torch.from_numpy(t).to(dev)
        self.triangles = torch.from_numpy(tri).to(dev)
        self.tetra_faces = torch.from_numpy(t[:, [[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]]].reshape(-1, 3)).to(dev)
        self.triangle_to_tetra = torch.from_numpy(owner).to(dev)

    def n(self):
        return self.points.shape[0]

    def gradient(self, x):
        """(N,4,3) tet corners -> (N,3,3), columns (v3-v0, v2-v0, v1-v0)."""
        return torch.stack([x[:, 3] - x[:, 0], x[:, 2] - x[:, 0], x[:, 1] - x[:, 0]], dim=2)

    def get_triangles(self, vertices):
        return vertices[self.triangles]


def _grid_csr(lo_cells, hi_cells, dims):
    """CSR (cell_start (ncell+1,) int32, items (N,) int32) of the items whose inclusive integer cell boxes [lo, hi] (n,3)
    overlap each cell of a dims = (nx, ny, nz) grid.  Device-side: expand every item into its cells, sort by cell."""
    dev = lo_cells.device
    nx, ny, nz = dims
    ext = (hi_cells - lo_cells + 1).clamp_min(0)
    cnt = ext[:, 0] * ext[:, 1] * ext[:, 2]
    item = torch.repeat_interleave(torch.arange(lo_cells.shape[0], device=dev), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(item.shape[0], device=dev) - first[item]              # running index inside the item's box
    ex, ey = ext[item, 0], ext[item, 1]
    cx = lo_cells[item, 0] + k % ex
    cy = lo_cells[item, 1] + (k // ex) % ey
    cz = lo_cells[item, 2] + k // (ex * ey)
    cell = (cz * ny + cy) * nx + cx
    order = torch.argsort(cell, stable=True)                               # stable: items stay ascending inside a cell
    counts = torch.bincount(cell, minlength=nx * ny * nz)
    start = torch.zeros(nx * ny * nz + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(counts, 0)
    return start.int().contiguous(), item[order].int().contiguous()


def compute_bary(points, tetras, triangles=None, tri_to_tetra=None, cage=None, method="grid"):
    """(points (P,3), tetras (T,4,3) corner coordinates, ...) -> (barys (P,4) f32, tetra_id (P,) int64, active (P,) bool).
    `triangles`, `tri_to_tetra` and `cage` are accepted for signature compatibility (lib/cage.py:325-327).
    method="grid" (default): a uniform grid over the cage prunes the candidates of every point to the tets whose bounding
    boxes overlap its cell; the few points no candidate contains (outside the cage) go through the exhaustive search, so
    the result equals method="exhaustive" (O(P T)) bit for bit."""
    require_cuda(points, tetras)
    pts = points.detach().float().contiguous()
    cor = tetras.detach().float().contiguous()
    P, T = pts.shape[0], cor.shape[0]
    barys = torch.empty((P, 4), dtype=torch.float32, device=pts.device)
    tid = torch.empty((P,), dtype=torch.int32, device=pts.device)
    act = torch.empty((P,), dtype=torch.uint8, device=pts.device)
    L = _lib.lib()
    if method == "exhaustive" or P == 0 or T < 64:
        check(L.d3ga_compute_bary(P, T, dptr(pts), dptr(cor), dptr(barys), dptr(tid), dptr(act), stream_handle()),
              "d3ga_compute_bary")
        return barys, tid.long(), act.bool()
    # grid over the cage: cell edge ~ twice the mean tet box edge, at most 128 cells per axis
    lo, hi = cor.amin(1), cor.amax(1)                                       # (T,3) tet boxes
    glo, ghi = lo.amin(0), hi.amax(0)
    mean_edge = float((hi - lo).mean())
    span = (ghi - glo).cpu()
    h = max(2.0 * mean_edge, float(span.max()) / 128.0, 1e-12)
    dims = [max(1, int(math.ceil(float(span[a]) / h)) + 1) for a in range(3)]
    pad = 1e-5 * float(span.max()) + 1e-9                                   # a point ON a face is inside the inflated box
    origin = (glo - pad).cpu()
    o = origin.to(pts.device)
    lo_c = ((lo - pad - o) / h).floor().long().clamp_min(0)
    hi_c = ((hi + pad - o) / h).floor().long()
    hi_c = torch.minimum(hi_c, torch.tensor([d - 1 for d in dims], device=pts.device))
    cell_start, cell_tets = _grid_csr(lo_c, hi_c, dims)
    import ctypes
    origin_h = (ctypes.c_float * 4)(float(origin[0]), float(origin[1]), float(origin[2]), h)
    cdims = (ctypes.c_int32 * 3)(*dims)
    minw = torch.empty((P,), dtype=torch.float32, device=pts.device)
    check(L.d3ga_compute_bary_grid(P, dptr(pts), dptr(cor), dptr(cell_start), dptr(cell_tets), origin_h, cdims, dptr(barys),
                                   dptr(tid), dptr(minw), stream_handle()), "d3ga_compute_bary_grid")
    outside = torch.nonzero(~(minw >= 0.0)).reshape(-1)                     # no containing candidate (or NaN): exhaustive
    act = (minw >= 0.0)
    if outside.numel():
        sub = pts[outside].contiguous()
        n = sub.shape[0]
        b2 = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
        t2 = torch.empty((n,), dtype=torch.int32, device=pts.device)
        a2 = torch.empty((n,), dtype=torch.uint8, device=pts.device)
        check(L.d3ga_compute_bary(n, T, dptr(sub), dptr(cor), dptr(b2), dptr(t2), dptr(a2), stream_handle()), "d3ga_compute_bary")
        barys[outside] = b2
        tid[outside] = t2
        act[outside] = a2.bool()
    return barys, tid.long(), act


def knn_mean_dist2(points, method="grid"):
    """(P,3) -> (P,) mean squared distance to the 3 nearest neighbours: `simple_knn._C.distCUDA2(points)`
    (models/mesh_net.py:66) == `knn_points(p[None], p[None], K=4)[0][0, :, 1:].mean(-1)` (models/cage_net.py:66).
    method="grid" (default): points bucketed into a uniform grid (~4 per cell), exact ring search; "exhaustive": O(P^2)."""
    require_cuda(points)
    pts = points.detach().float().contiguous()
    P = pts.shape[0]
    out = torch.empty((P,), dtype=torch.float32, device=pts.device)
    L = _lib.lib()
    if method == "exhaustive" or P < 2048:
        check(L.d3ga_knn3_mean_dist2(P, dptr(pts), dptr(out), stream_handle()), "d3ga_knn3_mean_dist2")
        return out
    glo, ghi = pts.amin(0), pts.amax(0)
    span = (ghi - glo).cpu()
    vol = float(torch.clamp(span, min=1e-9).prod())
    h = max((vol / (P / 4.0)) ** (1.0 / 3.0), float(span.max()) / 256.0, 1e-12)
    dims = [max(1, int(math.ceil(float(span[a]) / h)) + 1) for a in range(3)]
    origin = glo.cpu()
    c = ((pts - glo) / h).floor().long()
    c = torch.minimum(c.clamp_min(0), torch.tensor([d - 1 for d in dims], device=pts.device))
    cell_start, cell_points = _grid_csr(c, c, dims)
    import ctypes
    origin_h = (ctypes.c_float * 4)(float(origin[0]), float(origin[1]), float(origin[2]), h)
    cdims = (ctypes.c_int32 * 3)(*dims)
    check(L.d3ga_knn3_mean_dist2_grid(P, dptr(pts), dptr(cell_start), dptr(cell_points), origin_h, cdims, dptr(out),
                                      stream_handle()), "d3ga_knn3_mean_dist2_grid")
    return out


distCUDA2 = knn_mean_dist2


def spatial_order(points, bits=10):
    """(P,3) -> (P,) int64 permutation that numbers the points along a Morton (Z-order) curve of their bounding box.

    Why (docs/LOG.md sec. 4 *Index order*): every stage of the frame works on blocks of CONSECUTIVE Gaussians -- 256 per workgroup
    in the per-Gaussian kernels, whose tile-histogram / scatter windows in LDS cover the union of the block's rectangles --
    and the compositing stages gather 16-byte geometry records by Gaussian id in list order.  With spatially coherent
    numbering a block's window is a few dozen tiles and a tile list's records share cache lines; with a random numbering
    the frame at C3 takes 0.57 instead of 0.41 ms (`bench.py --gaussian-order random`).  The reference keeps the order of its
    initial point cloud (lib/cage.py:325) and appends on densification (utils/geometry.py:107); a model is free to number
    its Gaussians as it likes, so apply this ONCE to the initial points (before `compute_bary`, so that every per-Gaussian
    buffer and parameter is created in this order), and again when points have been appended:
        order = spatial_order(init_points);  init_points = init_points[order]
    Works on any device (torch ops only; init-time)."""
    pts = points.detach().to(torch.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError(f"spatial_order: (P,3) points expected, got {tuple(points.shape)}")
    if not 1 <= bits <= 20:
        raise ValueError("spatial_order: 1..20 bits per axis")
    if pts.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int64, device=points.device)
    lo, hi = pts.amin(0), pts.amax(0)
    q = ((pts - lo) / torch.clamp(hi - lo, min=1e-30) * (2 ** bits - 1)).round().to(torch.int64).clamp_(0, 2 ** bits - 1)
    code = torch.zeros(pts.shape[0], dtype=torch.int64, device=points.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.sort(code, stable=True)[1]


def gaussian_ply_columns(features_dc, features_rest, scaling, rotation):
    """Column names of the 3DGS-style PLY export (models/cage_net.py:111-122 describe_ply)."""
    cols = ["x", "y", "z", "nx", "ny", "nz"]
    cols += [f"f_dc_{i}" for i in range(features_dc.shape[1] * features_dc.shape[2])]
    cols += [f"f_rest_{i}" for i in range(features_rest.shape[1] * features_rest.shape[2])]
    cols.append("opacity")
    cols += [f"scale_{i}" for i in range(scaling.shape[1])]
    cols += [f"rot_{i}" for i in range(rotation.shape[1])]
    return cols


def gaussian_ply_arrays(features_dc, features_rest, opacities, scaling, rotation):
    """(f_dc, f_rest, opacities, scale, rotation) numpy blocks in the layout of models/cage_net.py:124-132 get_ply."""
    f_dc = features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    return (f_dc, f_rest, opacities.detach().cpu().numpy(), scaling.detach().cpu().numpy(),
            rotation.detach().cpu().numpy())


def save_gaussian_ply(path, xyz, features_dc, features_rest, opacities, scaling, rotation, normals=None):
    """Writes the 3DGS-style point cloud the reference describes with `describe_ply` / `get_ply` (models/cage_net.py:111-132,
    models/mesh_net.py:98-119): one `vertex` element, binary little-endian float32 properties in the order
    x y z nx ny nz | f_dc_* | f_rest_* | opacity | scale_* | rot_*, the feature blocks channel-major (`transpose(1, 2)`) as in
    `get_ply`.  The reference builds the column list and the blocks but leaves the file writing to plyfile-based viewers'
    converters; this is that writer without the plyfile dependency (same bytes plyfile's PlyElement.describe produces)."""
    import numpy as np
    cols = gaussian_ply_columns(features_dc, features_rest, scaling, rotation)
    f_dc, f_rest, op, sc, rot = gaussian_ply_arrays(features_dc, features_rest, opacities, scaling, rotation)
    xyz = xyz.detach().cpu().numpy() if hasattr(xyz, "detach") else np.asarray(xyz)
    nrm = np.zeros_like(xyz) if normals is None else (normals.detach().cpu().numpy() if hasattr(normals, "detach") else np.asarray(normals))
    table = np.concatenate([xyz, nrm, f_dc, f_rest, op.reshape(len(xyz), -1), sc, rot], axis=1).astype("<f4")
    assert table.shape[1] == len(cols), (table.shape, len(cols))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(xyz)
    header += "".join(f"property float {c}\n" for c in cols) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())
    return cols


def load_gaussian_ply(path):
    """-> dict(columns, xyz (P,3), normals (P,3), features_dc (P,1,3), features_rest (P,R,3), opacities (P,1), scaling (P,S),
    rotation (P,4)) as float32 torch tensors: the inverse of save_gaussian_ply (binary little-endian or ascii, float
    properties)."""
    import numpy as np
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").split("\n")
    if lines[0].strip() != "ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt = [l.split()[1] for l in lines if l.startswith("format")][0]
    n = int([l.split()[2] for l in lines if l.startswith("element vertex")][0])
    props = [l.split() for l in lines if l.startswith("property")]
    if any(p[1] not in ("float", "float32") for p in props):
        raise ValueError(f"{path}: only float32 properties are supported")
    cols = [p[2] for p in props]
    if fmt == "binary_little_endian":
        table = np.frombuffer(raw, dtype="<f4", count=n * len(cols), offset=end).reshape(n, len(cols))
    elif fmt == "ascii":
        table = np.array(raw[end:].split(), dtype=np.float32)[:n * len(cols)].reshape(n, len(cols))
    else:
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    col = {c: i for i, c in enumerate(cols)}
    pick = lambda prefix: [col[c] for c in sorted((c for c in cols if c.startswith(prefix)), key=lambda c: int(c.rsplit("_", 1)[1]))]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    dc, rest = table[:, pick("f_dc_")], table[:, pick("f_rest_")]
    return {"columns": cols, "xyz": t(table[:, [col["x"], col["y"], col["z"]]]),
            "normals": t(table[:, [col["nx"], col["ny"], col["nz"]]]),
            "features_dc": t(dc.reshape(n, 3, -1)).transpose(1, 2).contiguous(),          # file is channel-major
            "features_rest": t(rest.reshape(n, 3, -1)).transpose(1, 2).contiguous(),
            "opacities": t(table[:, [col["opacity"]]]), "scaling": t(table[:, pick("scale_")]),
            "rotation": t(table[:, pick("rot_")])}
