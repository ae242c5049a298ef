"""Seeded synthetic workloads for the deform-and-rasterize path.

Everything is generated on the host with numpy (bit-reproducible across boxes) and returned as CPU torch
tensors; callers move them to the device.  Shapes follow the reference:
cage vertices/tets as in lib/cage.py:310-337 buffers, Gaussian parameters as in models/cage_net.py:57-77,
``batch`` camera keys as produced by lib/batch.py:186-231 and consumed by lib/cameras.py:14-26.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

C0 = 0.28209479177387814  # utils/sh_utils.py:7


@dataclass
class Workload:
    name: str
    n_gaussians: int
    n_cages: int
    width: int
    height: int
    sh_degree: int = 3
    lattice: int = 19          # cells per axis per cage -> (n+1)^3 vertices, 6 n^3 tets
    sigma_px: float = 1.6      # target median projected std-dev in pixels at 1080p
    n_joints: int = 55
    skin_k: int = 4


WORKLOADS = {
    # BASELINE.json configs[0..4]
    "C1": Workload("C1: 10k Gaussians, 1 cage, 256x256", 10_000, 1, 256, 256, lattice=8),
    "C2": Workload("C2: 100k Gaussians, 3 cages, 1920x1080", 100_000, 3, 1920, 1080),
    "C3": Workload("C3: 500k Gaussians, 3 cages, 1920x1080, SH deg 3, fwd+bwd", 500_000, 3, 1920, 1080),
    "C4": Workload("C4: actor02-shaped 135k Gaussians, 3 cages, 747x1022", 135_000, 3, 747, 1022),
    "C5": Workload("C5: 2M Gaussians, 8 cages, 3840x2160", 2_000_000, 8, 3840, 2160),
    # tiny cases for tests / smoke
    "T0": Workload("T0: 600 Gaussians, 1 cage, 96x80", 600, 1, 96, 80, lattice=3),
    "T1": Workload("T1: 3000 Gaussians, 2 cages, 160x128", 3000, 2, 160, 128, lattice=4),
}


def kuhn_cage(n, lo, hi, jitter, rng):
    """Jittered (n+1)^3 lattice in the box [lo,hi], each cell split into 6 tets (Kuhn).  Vectorised."""
    g = np.linspace(0.0, 1.0, n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    pts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    pts = lo + pts * (hi - lo) + rng.uniform(-jitter, jitter, pts.shape) * (hi - lo) / n
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = np.stack([i, j, k], -1).reshape(-1, 3)                          # (n^3,3)
    stride = np.array([(n + 1) * (n + 1), n + 1, 1])
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    tets = []
    for p in perms:
        c = base.copy()
        path = [c @ stride]
        for ax in p:
            c = c.copy()
            c[:, ax] += 1
            path.append(c @ stride)
        tets.append(np.stack(path, 1))
    tets = np.stack(tets, 1).reshape(-1, 4)
    return pts.astype(np.float32), tets.astype(np.int32)


def _cage_boxes(n_cages):
    """Boxes tiling a 0.6 x 1.8 x 0.4 m body volume (y is up)."""
    lo, hi = np.array([-0.3, -0.9, -0.2]), np.array([0.3, 0.9, 0.2])
    boxes = []
    if n_cages == 8:
        for a in range(2):
            for b in range(4):
                l = np.array([lo[0] + a * 0.3, lo[1] + b * 0.45, lo[2]])
                boxes.append((l, l + np.array([0.3, 0.45, 0.4])))
    else:
        ys = np.linspace(lo[1], hi[1], n_cages + 1)
        for b in range(n_cages):
            boxes.append((np.array([lo[0], ys[b], lo[2]]), np.array([hi[0], ys[b + 1], hi[2]])))
    return boxes


def rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def make_skeleton(n_joints, rng, max_angle=0.5):
    """Synthetic articulated pose: joints scattered in the body box, each with a rotation about itself."""
    pos = rng.uniform([-0.3, -0.9, -0.2], [0.3, 0.9, 0.2], size=(n_joints, 3))
    A = np.zeros((n_joints, 4, 4), np.float32)
    for j in range(n_joints):
        rv = rng.normal(size=3)
        rv = rv / np.linalg.norm(rv) * rng.uniform(0, max_angle) * 0.3
        R = rodrigues(rv)
        A[j, :3, :3] = R
        A[j, :3, 3] = pos[j] - R @ pos[j] + rng.normal(size=3) * 0.01
        A[j, 3, 3] = 1
    return pos.astype(np.float32), A


def pose_matrices(joint_pos, rng, max_angle=0.5):
    """Another pose of the skeleton `make_skeleton` placed (the same joints, new rotations about them): (J,4,4) float32 -- the
    frames of a batch differ in exactly this (lib/smplman.py:155-171 takes the joint transforms of the frame)."""
    pos = np.asarray(joint_pos, np.float64)
    A = np.zeros((pos.shape[0], 4, 4), np.float32)
    for j in range(pos.shape[0]):
        rv = rng.normal(size=3)
        rv = rv / np.linalg.norm(rv) * rng.uniform(0, max_angle) * 0.3
        R = rodrigues(rv)
        A[j, :3, :3] = R
        A[j, :3, 3] = pos[j] - R @ pos[j] + rng.normal(size=3) * 0.01
        A[j, 3, 3] = 1
    return A


def skin_weights(verts, joint_pos, k):
    d = ((verts[:, None, :] - joint_pos[None]) ** 2).sum(-1)               # (V,J)
    idx = np.argsort(d, axis=1)[:, :k]
    w = 1.0 / (np.take_along_axis(d, idx, 1) + 1e-3)
    w = w / w.sum(1, keepdims=True)
    return idx.astype(np.int32), w.astype(np.float32)


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """World->camera rotation (rows = camera x (right), y (down), z (forward)) and translation."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    Rw2c = np.stack([r, d, f], 0)
    return Rw2c, -Rw2c @ eye


def make_batch(width, height, azimuth=0.0, dist=3.0, fill=0.85, body_h=1.8, frame_id=0, camera_id=0, cx=None, cy=None):
    """``batch`` dict with the keys renderer.render / lib.cameras.batch_to_camera consume.
    Optional principal point (cx, cy) reproduces the symmetric-FoV crop trick of lib/batch.py:186-198."""
    eye = np.array([dist * math.sin(azimuth), 0.0, -dist * math.cos(azimuth)])
    Rw2c, t = look_at(eye, [0.0, 0.0, 0.0])
    fy = fill * height * dist / body_h
    fx = fy
    W, H = width, height
    if cx is None:
        cx, cy = W // 2, H // 2
    left_w, right_w, top_h, bottom_h = cx, W - cx, cy, H - cy
    w, h = int(2 * max(left_w, right_w)), int(2 * max(top_h, bottom_h))
    return {
        "camera_id": camera_id, "frame_id": frame_id,
        "R": Rw2c.T.copy(), "T": t,                       # lib/batch.py:202: R is the transposed w2c rotation
        "FoVx": 2 * math.atan(w / (2 * fx)), "FoVy": 2 * math.atan(h / (2 * fy)),
        "width": w, "height": h,
        "crop": np.array([left_w, right_w, top_h, bottom_h, W, H]),
    }


PER_GAUSSIAN = ("tetra_id", "barys", "scaling", "rotation", "opacity_logit", "features_dc", "features_rest", "rgb")


def permute_gaussians(sc, seed=5):
    """The same scene with its Gaussians in RANDOM index order.  make_scene() numbers them by tetrahedron (spatially coherent
    blocks of 256); the reference takes the order of its initial point cloud (`compute_bary(init_points, ...)`, lib/cage.py:325,
    and appends on densification, utils/geometry.py:107) -- coherent along the template mesh, not sorted.  A random order is the
    worst case for everything that works on blocks of consecutive Gaussians (bench.py --gaussian-order random)."""
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(sc["tetra_id"].shape[0]))
    out = dict(sc)
    for k in PER_GAUSSIAN:
        out[k] = sc[k][perm].contiguous()
    return out


def reorder_gaussians(sc, order):
    out = dict(sc)
    for k in PER_GAUSSIAN:
        out[k] = sc[k][order].contiguous()
    return out


def canonical_centres(sc):
    """(P,3) canonical positions of the Gaussians: barycentric mean of their tetrahedron's canonical corners."""
    corners = sc["canon_points"][sc["tetras"].long()[sc["tetra_id"].long()]]           # (P,4,3)
    return (corners * sc["barys"][:, :, None]).sum(1)


def make_scene(wl, seed=17):
    """Returns a dict of CPU tensors describing one avatar: cages, skinning, Gaussians."""
    if isinstance(wl, str):
        wl = WORKLOADS[wl]
    rng = np.random.default_rng(seed)
    boxes = _cage_boxes(wl.n_cages)
    pts_all, tets_all, voff = [], [], 0
    tet_counts = []
    for lo, hi in boxes:
        p, t = kuhn_cage(wl.lattice, lo, hi, 0.2, rng)
        pts_all.append(p)
        tets_all.append(t + voff)
        voff += p.shape[0]
        tet_counts.append(t.shape[0])
    canon = np.concatenate(pts_all, 0)
    tetras = np.concatenate(tets_all, 0)
    T = tetras.shape[0]
    P = wl.n_gaussians
    tetra_id = np.sort(rng.integers(0, T, size=P)).astype(np.int32)
    barys = rng.dirichlet(np.ones(4), size=P).astype(np.float32)
    rotation = rng.normal(size=(P, 4)).astype(np.float32)
    # world sigma such that the projected std-dev is sigma_px at the workload's focal length (the 4K
    # stress config keeps the 1080p world size, i.e. twice the footprint in pixels)
    h_eff = min(wl.height, 1080)
    sigma0 = wl.sigma_px * 3.0 / (0.85 * h_eff * 3.0 / 1.8)
    scaling = np.log(sigma0 * np.exp(rng.normal(size=(P, 3)) * 0.3)).astype(np.float32)
    opacity_logit = rng.normal(size=(P, 1)).astype(np.float32)
    opacity_logit = np.minimum(opacity_logit, math.log(0.98 / 0.02)).astype(np.float32)
    M = 16                       # max_sh_degree 3 -> (P,1,3)+(P,15,3), models/cage_net.py:60-77
    features_dc = ((rng.uniform(size=(P, 1, 3)) - 0.5) / C0).astype(np.float32)
    features_rest = (rng.normal(size=(P, M - 1, 3)) * 0.05).astype(np.float32)
    rgb = rng.uniform(size=(P, 3)).astype(np.float32)
    joint_pos, A = make_skeleton(wl.n_joints, rng)
    skin_idx, skin_w = skin_weights(canon, joint_pos, wl.skin_k)
    t = torch.from_numpy
    return {
        "workload": wl,
        "canon_points": t(canon), "tetras": t(tetras), "tetra_id": t(tetra_id), "barys": t(barys),
        "scaling": t(scaling), "rotation": t(rotation), "opacity_logit": t(opacity_logit),
        "features_dc": t(features_dc), "features_rest": t(features_rest), "rgb": t(rgb),
        "joint_mats": t(A), "joint_pos": t(joint_pos), "skin_idx": t(skin_idx), "skin_w": t(skin_w),
        "delta_node": t((rng.normal(size=canon.shape) * 0.002).astype(np.float32)),
    }
