"""Property tests (hypothesis) and float64 gradchecks of the oracle -- the test strategy SURVEY.md sec. 4 derives from the
reference: barycentrics sum to one and reproduce the point (tetrahedron.h:77-101), rigid cage motion gives J = R and
cov' = R Sigma R^T (cage_net.py:218-230), paste() crop sizes (renderer.py:36-47, lib/batch.py:191-198), the symmetric
6-pack round trip (general_utils.py:24-35), and analytic-vs-numeric gradients of the differentiable restatements."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from d3ga_amd.dist import shard_views
from oracle import bary as ob
from oracle import camera as oc
from oracle import deform as od
from oracle import raster_torch as rt

SET = settings(max_examples=25, deadline=None, derandomize=True)


def _rot(rng):
    q = rng.normal(size=4)
    return od.quat_to_rotmat(torch.tensor(q / np.linalg.norm(q))[None])[0].numpy()


@SET
@given(st.integers(0, 2 ** 31 - 1))
def test_barycentrics_sum_to_one_and_reproduce_the_point(seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(4, 3))
    if abs(np.linalg.det(v[1:] - v[0])) < 1e-3:
        return
    w_in = rng.dirichlet(np.ones(4))
    p = w_in @ v
    w = ob.barycentric(v[0], v[1], v[2], v[3], p)
    np.testing.assert_allclose(w.sum(), 1.0, atol=1e-9)
    np.testing.assert_allclose(w @ v, p, atol=1e-9)
    np.testing.assert_allclose(w, w_in, atol=1e-8)
    assert bool(ob.point_in_tet(v[0], v[1], v[2], v[3], p))
    far = v.mean(0) + 50.0 * (v[0] - v.mean(0))
    assert not bool(ob.point_in_tet(v[0], v[1], v[2], v[3], far))


@SET
@given(st.integers(0, 2 ** 31 - 1))
def test_rigid_cage_motion_rotates_means_and_covariances(seed):
    rng = np.random.default_rng(seed)
    V = 8
    canon = torch.tensor(rng.normal(size=(V, 3)))
    tetras = torch.tensor([[0, 1, 2, 3], [4, 5, 6, 7], [0, 2, 4, 6]])
    vol = torch.linalg.det(od.tet_edge_matrix(canon[tetras]))
    if float(vol.abs().min()) < 1e-2:
        return
    P = 16
    tid = torch.tensor(rng.integers(0, 3, size=P))
    barys = torch.tensor(rng.dirichlet(np.ones(4), size=P))
    scales = torch.tensor(np.exp(rng.normal(size=(P, 3)) * 0.3))
    rots = torch.tensor(rng.normal(size=(P, 4)))
    cg = od.canonical_gradient(canon, tetras, tid)
    m0, c0 = od.cage_deform(canon, tetras, tid, barys, cg, scales, rots)
    R, t = torch.tensor(_rot(rng)), torch.tensor(rng.normal(size=3))
    m1, c1 = od.cage_deform(canon @ R.T + t, tetras, tid, barys, cg, scales, rots)
    np.testing.assert_allclose(m1.numpy(), (m0 @ R.T + t).numpy(), atol=1e-9)
    S0 = od.unpack_sym6(c0)
    np.testing.assert_allclose(od.unpack_sym6(c1).numpy(), (R @ S0 @ R.T).numpy(), atol=1e-8)
    # uniform scaling of the cage by s scales the covariance by s^2
    _, c2 = od.cage_deform(1.1 * canon, tetras, tid, barys, cg, scales, rots)
    np.testing.assert_allclose(c2.numpy(), 1.21 * c0.numpy(), rtol=1e-9, atol=1e-12)


@SET
@given(st.integers(0, 2 ** 31 - 1))
def test_sym6_round_trip_and_psd(seed):
    rng = np.random.default_rng(seed)
    s = torch.tensor(np.exp(rng.normal(size=(5, 3))))
    q = torch.tensor(rng.normal(size=(5, 4)))
    S = od.covariance_from_scale_rot(s, q)
    np.testing.assert_allclose(od.unpack_sym6(od.pack_sym6(S)).numpy(), S.numpy(), atol=1e-12)      # symmetric: exact up to S_ij vs S_ji
    assert float(torch.linalg.eigvalsh(S).min()) > -1e-9
    np.testing.assert_allclose(np.sort(torch.linalg.eigvalsh(S).numpy(), 1), np.sort((s ** 2).numpy(), 1), rtol=1e-8)


@SET
@given(st.integers(8, 64), st.integers(8, 64), st.integers(0, 12), st.integers(0, 12), st.booleans(), st.booleans())
def test_paste_undoes_the_symmetric_padding(w, h, pad_w, pad_h, left, top):
    """lib/batch.py:186-198 pads the image to put the principal point in the centre; paste() crops it back."""
    crop = [pad_w if left else 0, 0 if left else pad_w, pad_h if top else 0, 0 if top else pad_h, w, h]
    if pad_w == 0:
        crop[0], crop[1] = 1, 0
    if pad_h == 0:
        crop[2], crop[3] = 1, 0
    img = np.arange(3 * (h + pad_h) * (w + pad_w), dtype=np.float32).reshape(3, h + pad_h, w + pad_w)
    out = oc.paste(img, crop)
    assert out.shape == (3, h, w) == (3,) + oc.paste_shape(crop, h + pad_h, w + pad_w)
    x0 = 0 if crop[0] > crop[1] else pad_w
    y0 = 0 if crop[2] > crop[3] else pad_h
    np.testing.assert_array_equal(out, img[:, y0:y0 + h, x0:x0 + w])


@SET
@given(st.integers(1, 40), st.integers(1, 9))
def test_shard_views_partitions_the_views(n, world):
    got = sorted(v for r in range(world) for v in shard_views(n, r, world))
    assert got == list(range(n))
    sizes = [len(shard_views(n, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def test_gradcheck_deform_and_fem_float64():
    g = torch.Generator().manual_seed(3)
    canon = torch.randn(6, 3, generator=g, dtype=torch.float64)
    tetras = torch.tensor([[0, 1, 2, 3], [2, 3, 4, 5]])
    tid = torch.tensor([0, 1, 1, 0, 1])
    barys = torch.rand(5, 4, generator=g, dtype=torch.float64)
    cg = od.canonical_gradient(canon, tetras, tid)
    tp = (canon + 0.1 * torch.randn(6, 3, generator=g, dtype=torch.float64)).requires_grad_(True)
    b = barys.clone().requires_grad_(True)
    s = torch.rand(5, 3, generator=g, dtype=torch.float64).add(0.3).requires_grad_(True)
    q = torch.randn(5, 4, generator=g, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, bb, c, d: od.cage_deform(a, tetras, tid, bb, cg, c, d), (tp, b, s, q),
                                    eps=1e-6, atol=1e-6)
    Dn_inv = torch.linalg.inv(od.tet_edge_matrix(canon[tetras]))
    assert torch.autograd.gradcheck(lambda a: od.fem_energy(a, tetras, Dn_inv), (tp,), eps=1e-6, atol=1e-6)


def test_gradcheck_dense_rasterizer_float64():
    """The autograd oracle (which the C oracle's hand-derived backward is checked against) on a 3-Gaussian scene:
    analytic vs numeric gradients w.r.t. means, covariance, opacity and SH, away from the alpha / T thresholds."""
    dt = torch.float64
    cam = oc.camera(np.eye(3), np.array([0.0, 0.0, 4.0]), 0.6, 0.6)
    view, proj, campos = (torch.tensor(cam[k], dtype=dt) for k in ("world_view_transform", "full_proj_transform",
                                                                  "camera_center"))
    means = torch.tensor([[0.05, 0.02, 0.0], [-0.2, 0.15, 0.3], [0.18, -0.12, -0.2]], dtype=dt, requires_grad=True)
    L = torch.tensor([[[0.20, 0, 0], [0.03, 0.16, 0], [0.01, -0.02, 0.18]]] * 3, dtype=dt)
    cov = od.pack_sym6(L @ L.transpose(1, 2)).clone().requires_grad_(True)
    op = torch.tensor([[0.6], [0.4], [0.5]], dtype=dt, requires_grad=True)
    g = torch.Generator().manual_seed(1)
    sh = (0.3 * torch.randn(3, 4, 3, generator=g, dtype=dt)).add(0.2).requires_grad_(True)
    W = H = 12
    wgt = torch.randn(3, H, W, generator=g, dtype=dt)
    bg = torch.tensor([0.2, 0.4, 0.6], dtype=dt)

    def f(m, c, o, s):
        img, _ = rt.rasterize(m, o, bg, view, proj, campos, cam["tanfovx"], cam["tanfovy"], W, H, cov3D_precomp=c,
                              shs=s, sh_degree=1)
        return (img * wgt).sum()

    assert torch.autograd.gradcheck(f, (means, cov, op, sh), eps=1e-6, atol=1e-5, rtol=1e-4)
