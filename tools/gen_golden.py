#!/usr/bin/env python3
"""Generate golden fixtures by running the reference's OWN Python on CPU.

Runs ONLY in the build container (needs /root/reference).  The reference never travels: only the
small .npz files written under tests/golden/ are committed, together with this script.

Method: absent third-party roots are replaced by MagicMock packages, the
hard-coded ``.cuda()`` / ``device="cuda"`` are neutralised, and the reference's functions are then
called as they are:
  G1 camera   lib.cameras.Camera                        -> camera_cases.npz
  G2 cov      utils.general_utils.build_scaling_rotation / strip_symmetric -> cov_cases.npz
  G3 deform   models.cage_net.CageNet.forward (unbound, stand-in geometry) + autograd grads
              + lib.cage.CageBase.compute_def_grad / fem_energy       -> deform_case*.npz
  G4 boundary renderer.render up to the rasterizer call (recording stub)   -> boundary_cases.npz
  G5 SH       utils.sh_utils constants                                     -> sh_consts.npz
  G6 lbs      lib.smplman.Smplman.deform (unbound)                         -> lbs_case.npz
  G7 ply      models.cage_net.CageNet.describe_ply / get_ply (unbound)     -> ply_case.npz
  G8 lbs (Goliath)  lbsmodel.body_model.states_to_matrix + LinearBlendSkinning.skinning (unbound; 8-sparse
              indices / weights) + autograd gradient w.r.t. the vertices          -> lbs_goliath_case.npz
The only stand-in with numerical content is ``Tetra.gradient`` (un-vendored tetra_sampler): it is
written here as the column-edge matrix of lib/tet_mesh.py:88-94 (the reference's in-tree analogue).
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

ABSENT = {"loguru", "pytorch3d", "kornia", "trimesh", "meshio", "tetra_sampler", "tinycudann", "mcubes",
          "open3d", "cv2", "colour", "torchvision", "omegaconf", "diff_gaussian_rasterization",
          "simple_knn", "lpips", "plyfile", "rtree", "ffmpeg", "matplotlib", "networkx_stub"}


class _Stub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock()
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def install_harness():
    sys.path.insert(0, REF)
    sys.meta_path.insert(0, _Stub())
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fn in ("zeros", "ones", "eye", "tensor", "zeros_like"):
        f = getattr(torch, fn)
        setattr(torch, fn, (lambda f: lambda *a, **k: f(*a, **{kk: v for kk, v in k.items()
                                                              if not (kk == "device" and v == "cuda")}))(f))


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """w2c rotation (rows = camera axes, +z forward, +y down) and translation."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    Rw2c = np.stack([r, d, f], 0)
    t = -Rw2c @ eye
    return Rw2c, t


def kuhn_lattice(n, lo, hi, jitter, rng):
    """(n+1)^3 jittered lattice, 6 tets per cell (Kuhn split)."""
    g = np.linspace(0, 1, n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    pts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    pts = lo + pts * (hi - lo)
    pts = pts + rng.uniform(-jitter, jitter, pts.shape) * (hi - lo) / n

    def vid(i, j, k):
        return (i * (n + 1) + j) * (n + 1) + k
    tets = []
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    for i in range(n):
        for j in range(n):
            for k in range(n):
                for p in perms:
                    c = [i, j, k]
                    path = [vid(*c)]
                    for ax in p:
                        c[ax] += 1
                        path.append(vid(*c))
                    tets.append(path)
    return pts.astype(np.float32), np.asarray(tets, np.int64)


def gen_camera(cameras_mod):
    rng = np.random.default_rng(17)
    cases = {}
    n = 6
    for i in range(n):
        eye = rng.normal(size=3) * 2.0 + np.array([0, 0, -3.0])
        Rw2c, t = look_at(eye, rng.normal(size=3) * 0.2)
        R = Rw2c.T.copy()                      # reference convention: R is the TRANSPOSED w2c rotation
        fovx, fovy = rng.uniform(0.4, 1.4), rng.uniform(0.4, 1.4)
        cam = cameras_mod.Camera(colmap_id=i, R=R, T=t, FoVx=fovx, FoVy=fovy, image_name="x", uid=i,
                                 width=640, height=480)
        cases[f"R{i}"] = R
        cases[f"T{i}"] = t
        cases[f"fov{i}"] = np.array([fovx, fovy])
        cases[f"wv{i}"] = cam.world_view_transform.numpy()
        cases[f"proj{i}"] = cam.projection_matrix.numpy()
        cases[f"full{i}"] = cam.full_proj_transform.numpy()
        cases[f"center{i}"] = cam.camera_center.numpy()
    cases["n"] = np.array(n)
    np.savez(os.path.join(OUT, "camera_cases.npz"), **cases)


def gen_cov(gu):
    g = torch.Generator().manual_seed(17)
    s = torch.exp(torch.randn(64, 3, generator=g) * 0.5 - 3.0)
    q = torch.randn(64, 4, generator=g)
    L = gu.build_scaling_rotation(s, q)
    cov = gu.strip_symmetric(L @ L.transpose(1, 2))
    R = gu.build_rotation(q)
    np.savez(os.path.join(OUT, "cov_cases.npz"), scales=s.numpy(), rotations=q.numpy(), L=L.numpy(),
             cov6=cov.numpy(), R=R.numpy())


def make_geometry(CageBase, pts, tets, tetra_id, barys, posed, dtype):
    pts_t = torch.from_numpy(pts).to(dtype)
    tets_t = torch.from_numpy(tets)

    def gradient(x):  # stand-in for tetra_sampler.Tetra.gradient; lib/tet_mesh.py:88-94 column layout
        return torch.stack([x[:, 3] - x[:, 0], x[:, 2] - x[:, 0], x[:, 1] - x[:, 0]], dim=2)

    cage = SimpleNamespace(points=pts_t, tetras=tets_t, triangles=torch.zeros(1, 3, dtype=torch.long),
                           gradient=gradient)
    geo = SimpleNamespace(cage=cage, tetra_id=torch.from_numpy(tetra_id), barys=torch.from_numpy(barys).to(dtype))
    geo.canonical_gradient = torch.linalg.inv(gradient(pts_t[tets_t][geo.tetra_id]))     # lib/cage.py:329
    geo.Dn_inv = torch.linalg.inv(gradient(pts_t[tets_t]))                               # lib/cage.py:312-313
    geo.compute_def_grad = lambda tp: CageBase.compute_def_grad(geo, tp)
    geo.fem_energy = lambda p: CageBase.fem_energy(geo, p)
    geo.get = lambda lbs, delta=None: (posed if delta is None else posed + delta)[None]
    return geo


def gen_deform(cn, CageBase, name, n_cell, P, seed, dtype, use_shs):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    pts, tets = kuhn_lattice(n_cell, np.array([-0.3, -0.9, -0.2]), np.array([0.3, 0.9, 0.2]), 0.2, rng)
    T = tets.shape[0]
    tetra_id = np.sort(rng.integers(0, T, size=P)).astype(np.int64)
    barys = rng.dirichlet(np.ones(4), size=P).astype(np.float32)
    # posed cage: rotate + shear + noise
    A = np.eye(3) + rng.normal(size=(3, 3)) * 0.15
    posed_np = (pts @ A.T + rng.normal(size=pts.shape) * 0.01 + np.array([0.05, 0.1, -0.02])).astype(np.float32)
    posed = torch.from_numpy(posed_np).to(dtype).requires_grad_(True)
    geo = make_geometry(CageBase, pts, tets, tetra_id, barys, posed, dtype)

    scaling = (torch.randn(P, 3, generator=g) * 0.3 - 4.0).to(dtype).requires_grad_(True)
    rotation = torch.randn(P, 4, generator=g).to(dtype).requires_grad_(True)
    d_bary = (torch.tanh(torch.randn(P, 4, generator=g)) * 0.05).to(dtype).requires_grad_(True)
    d_rot = (torch.randn(P, 4, generator=g) * 0.1).to(dtype)
    d_scale = (torch.randn(P, 3, generator=g) * 0.1).to(dtype)
    d_node = (torch.randn(pts.shape[0], 3, generator=g) * 0.005).to(dtype)
    M = 16
    feats_dc = torch.rand(P, 1, 3, generator=g).to(dtype)
    feats_rest = (torch.randn(P, M - 1, 3, generator=g) * 0.05).to(dtype)
    opac = torch.randn(P, 1, generator=g).to(dtype)

    Rw2c, t = look_at([0.5, 0.2, -3.0], [0, 0, 0])
    batch = {"camera_id": 0, "R": Rw2c.T.copy(), "T": t, "FoVx": 0.8, "FoVy": 0.9, "frame_id": 0,
             "width": 256, "height": 256, "lbs": torch.zeros(87),
             "frame_encoding": None, "camera_encoding": None}

    self = SimpleNamespace()
    self.geometry = geo
    self.training = False
    self.deformation_field = lambda canon, cond: d_node
    self.canonical_field = lambda rot, scale, bary, cond: (d_bary, d_rot, d_scale)
    self.get_cond = lambda b, n: b["lbs"]
    self.tet_offset_pre_lbs = False
    self.scaling = scaling
    self.rotation = rotation
    self.scaling_activation = torch.exp
    self.rotation_activation = torch.nn.functional.normalize
    self.get_rotation = torch.nn.functional.normalize(rotation)
    self.get_scales = torch.exp(scaling)
    self.get_colors_feat = torch.zeros(P, 4)
    self.build_covariance_from_scaling_rotation = \
        lambda s, r: cn.CageNet.build_covariance_from_scaling_rotation(self, s, r)
    self.silhouette_color = torch.tensor([1.0, 0.0, 0.0])
    self.use_SHS = lambda: use_shs
    self.get_features = torch.cat((feats_dc, feats_rest), dim=1)
    self.get_opacity = torch.sigmoid(opac)
    self.cage_config = SimpleNamespace(cage_name="body")
    rgb_fixed = torch.rand(P, 3, generator=g)
    self.color_field = lambda *a: (rgb_fixed, torch.sigmoid(opac))

    pkg = cn.CageNet.forward(self, batch)

    gm = torch.randn(P, 3, generator=g).to(dtype)
    gc = torch.randn(P, 6, generator=g).to(dtype)
    loss = (pkg["means3D"] * gm).sum() + (pkg["cov3D_precomp"] * gc).sum()
    loss.backward()
    # the forward applies d_node in posed space (tet_offset_pre_lbs False) -> tetpoints = posed + d_node
    np.savez(
        os.path.join(OUT, f"{name}.npz"),
        canon_points=pts, tetras=tets, tetra_id=tetra_id, barys=barys,
        tetpoints=(posed.detach() + d_node).numpy(),
        canon_barys=(geo.barys + d_bary.detach()).numpy(),
        scales=torch.exp(scaling.detach() + d_scale).numpy(),
        rotations=torch.nn.functional.normalize(rotation.detach() + d_rot).numpy(),
        scaling_param=scaling.detach().numpy(), rotation_param=rotation.detach().numpy(),
        d_scale=d_scale.numpy(), d_rot=d_rot.numpy(),
        canonical_gradient=geo.canonical_gradient.numpy(), Dn_inv=geo.Dn_inv.numpy(),
        means3D=pkg["means3D"].detach().numpy(), cov3D_precomp=pkg["cov3D_precomp"].detach().numpy(),
        fm_energy=pkg["fm_energy"].detach().numpy(), scale_energy=pkg["scale_energy"].detach().numpy(),
        canonical_means3D=pkg["canonical_means3D"].detach().numpy(),
        shs=pkg["shs"].detach().numpy() if pkg["shs"] is not None else np.zeros(0),
        opacities=pkg["opacities"].detach().numpy(),
        silhouette_rgb=pkg["silhouette_rgb"].numpy(),
        up_grad_means=gm.numpy(), up_grad_cov=gc.numpy(),
        grad_tetpoints=posed.grad.numpy(), grad_barys=d_bary.grad.numpy(),
        grad_scaling_param=scaling.grad.numpy(), grad_rotation_param=rotation.grad.numpy(),
    )


class _Recorder:
    calls = []


def gen_boundary(renderer):
    class Settings:
        def __init__(self, **kw):
            self.kw = kw

    class Rasterizer:
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, **kw):
            _Recorder.calls.append((self.s.kw, kw))
            h, w = self.s.kw["image_height"], self.s.kw["image_width"]
            img = torch.arange(3 * h * w, dtype=torch.float32).reshape(3, h, w)
            return (img, None, None)

    renderer.GaussianRasterizationSettings = Settings
    renderer.GaussianRasterizer = Rasterizer
    renderer.cuda_timer = lambda *a, **k: __import__("contextlib").nullcontext()

    rng = np.random.default_rng(5)
    P = 7
    out = {}
    # replicate lib/batch.py:186-198 crop arithmetic for a principal point (cx, cy) in a WxH image
    cases = [(200, 240, 90, 130), (200, 240, 120, 100), (64, 48, 32, 24)]
    kinds = ["sh", "precomp", "silhouette"]
    ci = 0
    for (W, H, cx, cy) in cases:
        left_w, right_w, top_h, bottom_h = cx, W - cx, cy, H - cy
        w, h = int(2 * max(left_w, right_w)), int(2 * max(top_h, bottom_h))
        crop = np.array([left_w, right_w, top_h, bottom_h, W, H])
        Rw2c, t = look_at([0.3, 0.1, -2.5], [0, 0, 0])
        fx = fy = 300.0
        batch = {"camera_id": 1, "R": Rw2c.T.copy(), "T": t, "FoVx": 2 * math.atan(w / (2 * fx)),
                 "FoVy": 2 * math.atan(h / (2 * fy)), "frame_id": 3, "width": w, "height": h, "crop": crop}
        for kind in kinds:
            pkg = {"means3D": torch.from_numpy(rng.normal(size=(P, 3)).astype(np.float32)).requires_grad_(True),
                   "cov3D_precomp": torch.rand(P, 6).requires_grad_(True),
                   "opacities": torch.rand(P, 1).requires_grad_(True),
                   "shs": torch.rand(P, 16, 3) if kind == "sh" else None,
                   "rgb": torch.rand(P, 3) if kind != "sh" else None,
                   "sh_degree": 2}
            _Recorder.calls.clear()
            if kind == "silhouette":
                res = renderer.render(batch, pkg, torch.zeros(3), colors_precomp=torch.rand(P, 3),
                                      detach=["position", "covariance"])
            else:
                res = renderer.render(batch, pkg, torch.ones(3))
            s, kw = _Recorder.calls[0]
            pre = f"c{ci}_"
            out[pre + "kind"] = np.array(kind)
            out[pre + "batch_R"] = batch["R"]
            out[pre + "batch_T"] = batch["T"]
            out[pre + "batch_fov"] = np.array([batch["FoVx"], batch["FoVy"]])
            out[pre + "batch_wh"] = np.array([w, h])
            out[pre + "crop"] = crop
            out[pre + "out_shape"] = np.array(res["render"].shape)
            out[pre + "out_first"] = res["render"][:, 0, 0].numpy()
            out[pre + "out_last"] = res["render"][:, -1, -1].numpy()
            for k in ("image_height", "image_width", "tanfovx", "tanfovy", "scale_modifier", "sh_degree"):
                out[pre + "s_" + k] = np.array(s[k])
            for k in ("prefiltered", "debug", "antialiasing"):
                out[pre + "s_" + k] = np.array(bool(s[k]))
            out[pre + "s_viewmatrix"] = s["viewmatrix"].numpy()
            out[pre + "s_projmatrix"] = s["projmatrix"].numpy()
            out[pre + "s_campos"] = s["campos"].numpy()
            out[pre + "s_bg"] = s["bg"].numpy()
            for k, v in kw.items():
                out[pre + "arg_" + k + "_none"] = np.array(v is None)
                if v is not None:
                    out[pre + "arg_" + k + "_shape"] = np.array(v.shape)
                    out[pre + "arg_" + k + "_requires_grad"] = np.array(bool(v.requires_grad))
            ci += 1
    out["n"] = np.array(ci)
    np.savez(os.path.join(OUT, "boundary_cases.npz"), **out)


def gen_sh(sh_utils):
    np.savez(os.path.join(OUT, "sh_consts.npz"), C0=np.array(sh_utils.C0), C1=np.array(sh_utils.C1),
             C2=np.array(sh_utils.C2), C3=np.array(sh_utils.C3),
             rgb2sh_half=np.array(sh_utils.RGB2SH(np.array([0.0, 0.5, 1.0]))))


def gen_lbs(smplman_mod):
    g = torch.Generator().manual_seed(23)
    V, J = 40, 6
    w = torch.rand(V, J, generator=g)
    w = w / w.sum(1, keepdim=True)
    A = torch.eye(4)[None].repeat(J, 1, 1)
    A[:, :3, :] += torch.randn(J, 3, 4, generator=g) * 0.2
    tmpl = torch.randn(1, V, 3, generator=g)
    delta = torch.randn(V, 3, generator=g) * 0.01
    Rh_mat = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    Th = torch.randn(1, 3, generator=g)
    smplman_mod.batch_rodrigues = lambda rh: Rh_mat[None]      # un-vendored tetra_sampler.lbs helper
    ns = SimpleNamespace(lbs_module=SimpleNamespace(J_regressor=torch.zeros(J, 1)), skin_weights=w,
                         body_template_vertices=tmpl, nn_ids=torch.arange(V))
    ns.to_homo = lambda v: smplman_mod.Smplman.to_homo(ns, v)
    out = smplman_mod.Smplman.deform(ns, A[None], torch.zeros(1, V, 3), torch.zeros(1, 3), Th, delta)
    np.savez(os.path.join(OUT, "lbs_case.npz"), weights=w.numpy(), A=A.numpy(), template=tmpl[0].numpy(),
             delta=delta.numpy(), Rh=Rh_mat.numpy(), Th=Th[0].numpy(), out=out[0].numpy())


def gen_lbs_goliath(bm):
    """lbsmodel/body_model.py:208-234 (LinearBlendSkinning.skinning, called unbound on a namespace that holds the 8-sparse
    skin_indices / skin_weights buffers) on top of :350-387 (states_to_matrix): seeded bind / target skeleton states
    (translation 3 | quaternion xyzw 4 | scale 1), J = 30 joints, two poses; the skinned vertices, the (J,3,4) matrices and the
    reference's autograd gradient w.r.t. the vertices for a fixed upstream gradient."""
    g = torch.Generator().manual_seed(29)
    V, J, K, B = 600, 30, 8, 2

    def states(n):
        q = torch.randn(n, J, 4, generator=g)
        q = q / q.norm(dim=2, keepdim=True)
        t = torch.randn(n, J, 3, generator=g) * 0.5
        sc = torch.exp(0.2 * torch.randn(n, J, 1, generator=g))
        return torch.cat([t, q, sc], dim=2)
    bind, target = states(1), states(B)
    idx = torch.randint(0, J, (V, K), generator=g)
    w = torch.rand(V, K, generator=g)
    w[:, 5:] = 0.0                                   # rows with fewer than 8 influences carry zero weights (the file format pads)
    w = w / w.sum(1, keepdim=True)
    verts = torch.randn(1, V, 3, generator=g).requires_grad_(True)
    ns = SimpleNamespace(skin_indices=idx, skin_weights=w)
    mat = bm.states_to_matrix(bind, target)
    out = bm.LinearBlendSkinning.skinning(ns, bind, verts, target)
    gout = torch.randn(out.shape, generator=g)
    (out * gout).sum().backward()
    np.savez(os.path.join(OUT, "lbs_goliath_case.npz"), bind_state=bind.numpy(), target_states=target.numpy(),
             skin_indices=idx.numpy().astype(np.int32), skin_weights=w.numpy(), vertices=verts.detach()[0].numpy(),
             mat=mat.detach().numpy(), out=out.detach().numpy(), grad_out=gout.numpy(), grad_vertices=verts.grad[0].numpy())


def gen_losses(loss_utils):
    """utils/loss_utils.py: l1_loss (:29) and ssim (:59-86) of the reference on seeded images, values and autograd
    gradients w.r.t. the first image (sizes chosen to cross tile borders of the HIP kernel: not multiples of 16)."""
    g = torch.Generator().manual_seed(23)
    out = {}
    for name, (C, H, W) in {"a": (3, 37, 53), "b": (1, 16, 16), "c": (3, 64, 96)}.items():
        gt = torch.rand(C, H, W, generator=g)
        pred = (gt + 0.25 * torch.randn(C, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
        val = loss_utils.ssim(pred, gt)
        (grad,) = torch.autograd.grad(val, pred)
        l1 = loss_utils.l1_loss(pred, gt)
        out.update({f"{name}_pred": pred.detach().numpy(), f"{name}_gt": gt.numpy(), f"{name}_ssim": val.detach().numpy(),
                    f"{name}_ssim_grad": grad.numpy(), f"{name}_l1": l1.detach().numpy()})
    np.savez(os.path.join(OUT, "loss_cases.npz"), **out)


def gen_fields(mlp_mod):
    """models/mlp.py: DeformationField (:39-71) and CanonicalField (:74-110) of the reference with the widths of
    configs/actorshq_actor02.yml (n_nodes 128, n_layers 3): weights, inputs, outputs and autograd gradients w.r.t. the
    per-row inputs, the pose vector and every weight, for a fixed upstream gradient."""
    from types import SimpleNamespace as NS

    class Cfg(dict):
        __getattr__ = dict.__getitem__

        def get(self, k, d=None):
            return dict.get(self, k, d)

    config = Cfg(is_smpl_body=False, deform_mlp=Cfg(n_layers=3, n_nodes=128, scale=0.2),
                 canon_mlp=Cfg(n_layers=3, n_nodes=128, scale_bary=0.25, scale_rot=0.25, scale_scale=0.25))
    cage_config = Cfg(cage_name="body", node_scale=0.07)
    torch.manual_seed(31)
    out = {}
    g = torch.Generator().manual_seed(32)
    # --- CanonicalField: z = [pose(98) | rots(4) | scales(3) | barys(4)] -> 128 x (1+3) -> 11
    cf = mlp_mod.CanonicalField(config, cage_config)
    P = 300
    barys = torch.rand(P, 4, generator=g).requires_grad_(True)
    rots = torch.randn(P, 4, generator=g).requires_grad_(True)
    scales = (0.1 * torch.randn(P, 3, generator=g)).requires_grad_(True)
    pose = (0.3 * torch.randn(98, generator=g)).requires_grad_(True)
    # called as (rot, scale, bary, cond) against the signature (barys, rots, scales, pose): cage_net.py:199-204
    d_bary, d_rot, d_scale = cf(rots, scales, barys, pose)
    up = [torch.randn(*t.shape, generator=g) for t in (d_bary, d_rot, d_scale)]
    params = list(cf.parameters())
    grads = torch.autograd.grad([d_bary, d_rot, d_scale], [barys, rots, scales, pose] + params, up)
    out.update(cf_barys=barys.detach().numpy(), cf_rots=rots.detach().numpy(), cf_scales=scales.detach().numpy(),
               cf_pose=pose.detach().numpy(), cf_d_bary=d_bary.detach().numpy(), cf_d_rot=d_rot.detach().numpy(),
               cf_d_scale=d_scale.detach().numpy(), cf_up0=up[0].numpy(), cf_up1=up[1].numpy(), cf_up2=up[2].numpy(),
               cf_g_barys=grads[0].numpy(), cf_g_rots=grads[1].numpy(), cf_g_scales=grads[2].numpy(),
               cf_g_pose=grads[3].numpy())
    for (name, prm), gr in zip(cf.named_parameters(), grads[4:]):
        out[f"cf_w_{name}"] = prm.detach().numpy()
        out[f"cf_gw_{name}"] = gr.numpy()
    # --- DeformationField: z = [pose(98) | embed_7(canonical)(45)] -> 128 x (1+3) -> 3, tanh * node_scale
    df = mlp_mod.DeformationField(config, cage_config)
    V = 211
    canon = torch.randn(V, 3, generator=g).requires_grad_(True)
    pose2 = (0.3 * torch.randn(98, generator=g)).requires_grad_(True)
    delta = df(canon, pose2)
    up2 = torch.randn(V, 3, generator=g)
    params = list(df.parameters())
    grads = torch.autograd.grad(delta, [canon, pose2] + params, up2)
    out.update(df_canon=canon.detach().numpy(), df_pose=pose2.detach().numpy(), df_delta=delta.detach().numpy(),
               df_up=up2.numpy(), df_g_canon=grads[0].numpy(), df_g_pose=grads[1].numpy())
    for (name, prm), gr in zip(df.named_parameters(), grads[2:]):
        out[f"df_w_{name}"] = prm.detach().numpy()
        out[f"df_gw_{name}"] = gr.numpy()
    # --- ShadowDecoder: z = [pose[6:](92) | embed_7(template)(45)] -> 128 x (1+3) -> 1, sigmoid   (models/mlp.py:262-297)
    config["shadow_mlp"] = Cfg(n_layers=3, n_nodes=128)
    tmpl = torch.randn(97, 3, generator=g)
    sd = mlp_mod.ShadowDecoder(config, tmpl)
    pose3 = (0.3 * torch.randn(104, generator=g)).requires_grad_(True)      # pose[6:] must have 98 entries (mlp.py:271,288)
    ao = sd(pose3)
    up3 = torch.randn(97, 1, generator=g)
    params = list(sd.parameters())
    grads = torch.autograd.grad(ao, [pose3] + params, up3)
    out.update(sd_template=tmpl.numpy(), sd_pose=pose3.detach().numpy(), sd_ao=ao.detach().numpy(), sd_up=up3.numpy(),
               sd_g_pose=grads[0].numpy())
    for (name, prm), gr in zip(sd.named_parameters(), grads[1:]):
        out[f"sd_w_{name}"] = prm.detach().numpy()
        out[f"sd_gw_{name}"] = gr.numpy()
    # --- FaceDecoder: keypoints (n,3) flattened -> 128 x (1+3) -> n_output   (models/mlp.py:235-259)
    config["face_mlp"] = Cfg(n_layers=3, n_nodes=128, n_output=128)
    fd = mlp_mod.FaceDecoder(config, 33)
    kpt = torch.randn(33, 3, generator=g).requires_grad_(True)
    code = fd(kpt)
    up4 = torch.randn(128, generator=g)
    params = list(fd.parameters())
    grads = torch.autograd.grad(code, [kpt] + params, up4)
    out.update(fd_kpt=kpt.detach().numpy(), fd_code=code.detach().numpy(), fd_up=up4.numpy(), fd_g_kpt=grads[0].numpy())
    for (name, prm), gr in zip(fd.named_parameters(), grads[1:]):
        out[f"fd_w_{name}"] = prm.detach().numpy()
        out[f"fd_gw_{name}"] = gr.numpy()
    # --- ColorField (the colour path of configs/actorshq_actor02.yml: use_shs false, use_pose, use_view_enc, frame
    # embedder 32): the reference's own module, with tiny-cuda-nn's un-vendored direction encoding replaced by the
    # oracle's stand-in (so everything BUT that encoding is pinned)   (models/mlp.py:152-232)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import mlp as om

    class _Enc:
        n_output_dims = 16

        def __init__(self, n_input_dims, encoding_config):
            assert n_input_dims == 3 and encoding_config["nested"][0]["degree"] == 4

        def __call__(self, x):
            return om.sh4_direction_encoding(x)

    mlp_mod.tcnn = NS(Encoding=_Enc)
    config["color_mlp"] = Cfg(n_layers=4, n_nodes=128, n_features=64, use_pose=True, use_view_enc=True)
    config["frame_embedder"] = Cfg(n_dims=32)
    config["train"] = Cfg()
    col = mlp_mod.ColorField(config, cage_config)
    Pc = 260
    feat = (0.33 * torch.rand(Pc, 64, generator=g)).requires_grad_(True)
    pose4 = (0.3 * torch.randn(98, generator=g)).requires_grad_(True)
    vd = torch.nn.functional.normalize(torch.randn(Pc, 3, generator=g), dim=1).requires_grad_(True)
    frame = (0.5 * torch.randn(32, generator=g)).requires_grad_(True)
    rgb, opa = col(feat, pose4, vd, frame, None, None)     # (shadow would make z one column wider than n_input: mlp.py:188)
    upc = [torch.randn(Pc, 3, generator=g), torch.randn(Pc, 1, generator=g)]
    params = list(col.parameters())
    grads = torch.autograd.grad([rgb, opa], [feat, pose4, vd, frame] + params, upc)
    out.update(col_feat=feat.detach().numpy(), col_pose=pose4.detach().numpy(), col_viewdir=vd.detach().numpy(),
               col_frame=frame.detach().numpy(), col_rgb=rgb.detach().numpy(),
               col_opacity=opa.detach().numpy(), col_up0=upc[0].numpy(), col_up1=upc[1].numpy(),
               col_g_feat=grads[0].numpy(), col_g_pose=grads[1].numpy(), col_g_viewdir=grads[2].numpy(),
               col_g_frame=grads[3].numpy())
    for (name, prm), gr in zip(col.named_parameters(), grads[4:]):
        out[f"col_w_{name}"] = prm.detach().numpy()
        out[f"col_gw_{name}"] = gr.numpy()
    np.savez(os.path.join(OUT, "field_cases.npz"), **out)


def gen_ply(cn):
    """G7: CageNet.describe_ply / get_ply (models/cage_net.py:111-132), called unbound on a stand-in carrying the five
    parameter tensors -> the column names and the five blocks of the 3DGS-style export."""
    g = torch.Generator().manual_seed(23)
    P = 37
    self = SimpleNamespace(features_dc=torch.randn(P, 1, 3, generator=g), features_rest=torch.randn(P, 15, 3, generator=g),
                           opacities=torch.randn(P, 1, generator=g), scaling=torch.randn(P, 3, generator=g),
                           rotation=torch.randn(P, 4, generator=g))
    cols = cn.CageNet.describe_ply(self)
    f_dc, f_rest, op, sc, rot = cn.CageNet.get_ply(self)
    np.savez(os.path.join(OUT, "ply_case.npz"), columns=np.array(cols), features_dc=self.features_dc.numpy(),
             features_rest=self.features_rest.numpy(), opacities=self.opacities.numpy(), scaling=self.scaling.numpy(),
             rotation=self.rotation.numpy(), f_dc=f_dc, f_rest=f_rest, opacity=op, scale=sc, rot=rot)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_harness()
    import models.cage_net as cn
    from lib.cage import CageBase
    import lib.cameras as cameras_mod
    import utils.general_utils as gu
    import utils.sh_utils as sh_utils
    import renderer
    import lib.smplman as smplman_mod

    if sys.argv[1:] == ["ply"]:                 # one section only (the other fixtures are left untouched)
        gen_ply(cn)
        return
    if sys.argv[1:] == ["lbs_goliath"]:
        import lbsmodel.body_model as bm
        gen_lbs_goliath(bm)
        return
    gen_ply(cn)
    gen_camera(cameras_mod)
    gen_cov(gu)
    gen_deform(cn, CageBase, "deform_case0", n_cell=3, P=257, seed=17, dtype=torch.float32, use_shs=True)
    gen_deform(cn, CageBase, "deform_case1", n_cell=4, P=1000, seed=18, dtype=torch.float32, use_shs=False)
    gen_boundary(renderer)
    gen_sh(sh_utils)
    import utils.loss_utils as loss_utils
    gen_losses(loss_utils)
    import models.mlp as mlp_mod
    gen_fields(mlp_mod)
    gen_lbs(smplman_mod)
    import lbsmodel.body_model as bm
    gen_lbs_goliath(bm)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
