"""Known answers that share NO code with the rasterizer restatements ("harden parity" item c).

The HIP kernels and the C oracle both restate the un-vendored `diff_gaussian_rasterization` from the same recollection of
upstream (closed-form EWA Jacobian, guard band, radius rule, conic backward).  A constant mis-remembered identically on
both sides would keep every HIP-vs-oracle test green.  The checks below derive their expected values from FIRST
PRINCIPLES in float64 instead:

  * the 2D covariance of a splat = J Sigma_cam J^T + 0.3 I with J a NUMERICAL Jacobian (central differences) of the
    camera-space -> pixel map, that map itself built from the reference's own camera matrices (oracle/camera.py is pinned
    to lib/cameras.py by golden vectors) -- no closed-form `fx/tz, -fx tx/tz^2`, no `limx`/`txtz`;
  * the guard band: the same numerical Jacobian evaluated at the point whose x/z, y/z are clamped to 1.3 tan(fov/2)
    (upstream's frustum guard), for a splat far off-axis;
  * the radius rule ceil(3 sqrt(mid + sqrt(max(0.1, mid^2 - det))));
  * the backward: central finite differences of the HIP forward itself in a regime with no discontinuity (every alpha
    >= 1/255 on every pixel, no saturation, no 0.99 clamp), against the HIP backward.
The CPU tests run the same known answers through the C oracle (`-m "not gpu"`), the GPU tests through the C ABI.
"""
import math

import numpy as np
import pytest
import torch

from d3ga_amd import synthetic as syn
from oracle import camera as oc
from oracle import raster_c as rc


# ---------------------------------------------------------------------------------------------------------------------
# first-principles reference (float64, numpy)
# ---------------------------------------------------------------------------------------------------------------------
def _camera(W, H, azimuth=0.35):
    b = syn.make_batch(W, H, azimuth=azimuth)
    cam = oc.camera(b["R"], b["T"], b["FoVx"], b["FoVy"])
    V = cam["world_view_transform"].astype(np.float64)       # row-vector convention: [p,1] @ V
    F = cam["full_proj_transform"].astype(np.float64)
    Pm = np.linalg.inv(V) @ F                                 # camera space -> clip space
    return b, cam, V, Pm


def _pix_of_cam(t, Pm, W, H):
    """camera-space point -> pixel coordinates through the reference's projection matrix; pixel = ((ndc + 1) S - 1) / 2."""
    h = np.concatenate([t, [1.0]]) @ Pm
    ndc = h[:2] / h[3]
    return np.array([((ndc[0] + 1.0) * W - 1.0) * 0.5, ((ndc[1] + 1.0) * H - 1.0) * 0.5])


def _numeric_jacobian(t, Pm, W, H):
    J = np.zeros((2, 3))
    for k in range(3):
        h = 1e-6 * max(1.0, abs(t[k]))
        e = np.zeros(3); e[k] = h
        J[:, k] = (_pix_of_cam(t + e, Pm, W, H) - _pix_of_cam(t - e, Pm, W, H)) / (2 * h)
    return J


def _expected_splat(p_world, cov6, opacity, cam, V, Pm, W, H, guard=True):
    """(alpha image (H,W) float64, radius, pixel centre, cov2D) of ONE Gaussian from first principles."""
    t = np.concatenate([p_world, [1.0]]) @ V
    t = t[:3]
    S = np.array([[cov6[0], cov6[1], cov6[2]], [cov6[1], cov6[3], cov6[4]], [cov6[2], cov6[4], cov6[5]]], np.float64)
    R = V[:3, :3]                                             # t = p @ R  ->  Sigma_cam = R^T Sigma R
    S_cam = R.T @ S @ R
    tj = t.copy()
    if guard:                                                 # the Jacobian is taken at the guard-band-clamped point
        lx, ly = 1.3 * cam["tanfovx"], 1.3 * cam["tanfovy"]
        tj[0] = np.clip(t[0] / t[2], -lx, lx) * t[2]
        tj[1] = np.clip(t[1] / t[2], -ly, ly) * t[2]
    J = _numeric_jacobian(tj, Pm, W, H)
    c2 = J @ S_cam @ J.T + 0.3 * np.eye(2)
    centre = _pix_of_cam(t, Pm, W, H)
    mid = 0.5 * (c2[0, 0] + c2[1, 1])
    det = np.linalg.det(c2)
    radius = math.ceil(3.0 * math.sqrt(mid + math.sqrt(max(0.1, mid * mid - det))))
    ci = np.linalg.inv(c2)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    dx, dy = centre[0] - xs, centre[1] - ys
    q = ci[0, 0] * dx * dx + 2 * ci[0, 1] * dx * dy + ci[1, 1] * dy * dy
    alpha = np.minimum(0.99, opacity * np.exp(-0.5 * q))
    alpha[alpha < 1.0 / 255.0] = 0.0
    return alpha, radius, centre, c2


def _world_of_cam(t, V):
    return (np.concatenate([t, [1.0]]) @ np.linalg.inv(V))[:3]


def _aniso_cov6(scales, axis, angle):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rm = np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * K @ K
    S = Rm @ np.diag(np.square(scales)) @ Rm.T
    return np.array([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])


CASES = {
    # name: (camera-space position as (x/z / tanfovx, y/z / tanfovy, z), scales, rotation axis, angle, opacity)
    # opacity 0.3: alpha at 3 sigma = 0.3 e^-4.5 < 1/255, so the tile rectangle (radius rule) cannot truncate the footprint
    "off_axis_anisotropic": ((0.62, -0.48, 3.1), (0.09, 0.03, 0.015), (0.3, 1.0, 0.5), 0.9, 0.3),
    "near_centre_elongated": ((0.08, 0.05, 2.7), (0.12, 0.012, 0.012), (0.0, 0.2, 1.0), 0.6, 0.3),
    # far outside the frustum sideways (x/z = 1.6 tanfovx > 1.3 tanfovx) but wide enough to reach into the image
    "guard_band": ((1.6, 0.2, 3.0), (0.45, 0.40, 0.30), (1.0, 0.3, 0.2), 0.4, 0.3),
    "guard_band_corner": ((-1.45, -1.5, 2.8), (0.5, 0.45, 0.4), (0.2, 1.0, 0.1), 1.1, 0.3),
}
W_, H_ = 96, 80


def _case_inputs(name):
    pos, scales, axis, angle, o = CASES[name]
    b, cam, V, Pm = _camera(W_, H_)
    z = pos[2]
    t = np.array([pos[0] * cam["tanfovx"] * z, pos[1] * cam["tanfovy"] * z, z])
    p = _world_of_cam(t, V)
    cov6 = _aniso_cov6(scales, axis, angle)
    return b, cam, V, Pm, p, cov6, o


def _check_alpha_image(name, render):
    b, cam, V, Pm, p, cov6, o = _case_inputs(name)
    alpha, radius, centre, c2 = _expected_splat(p, cov6, o, cam, V, Pm, b["width"], b["height"])
    assert (alpha > 0).sum() > 30, "the case must put a visible footprint into the image"
    color = np.array([1.0, 0.5, 0.25], np.float32)
    bg = np.array([0.0, 0.0, 0.0], np.float32)
    img, rad = render(p.astype(np.float32)[None], cov6.astype(np.float32)[None], np.array([[o]], np.float32), color[None], bg,
                      cam, b["width"], b["height"])
    assert int(rad[0]) == radius, (int(rad[0]), radius)
    # pixels within float32 noise of the 1/255 cut-off may fall either way: everything else must match to 2e-5
    q_alpha = o * np.exp(-0.5 * _quad(centre, c2, b["width"], b["height"]))
    stable = np.abs(q_alpha * 255.0 - 1.0) > 1e-3
    for ch in range(3):
        np.testing.assert_allclose(img[ch][stable], (alpha * color[ch])[stable], atol=2e-5, rtol=0)


def _quad(centre, c2, W, H):
    ci = np.linalg.inv(c2)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    dx, dy = centre[0] - xs, centre[1] - ys
    return ci[0, 0] * dx * dx + 2 * ci[0, 1] * dx * dy + ci[1, 1] * dy * dy


def _render_oracle(means, cov6, op, color, bg, cam, W, H):
    img, radii, _, _ = rc.forward(means, op, bg, cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                                  cam["tanfovx"], cam["tanfovy"], W, H, cov3D_precomp=cov6, colors_precomp=color)
    return img, radii


def _render_hip(means, cov6, op, color, bg, cam, W, H):
    from d3ga_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(bg),
                                       scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]),
                                       projmatrix=t(cam["full_proj_transform"]), sh_degree=0, campos=t(cam["camera_center"]),
                                       prefiltered=False, debug=False)
    with torch.no_grad():
        img, radii, _ = GaussianRasterizer(st)(means3D=t(means), means2D=None, opacities=t(op), colors_precomp=t(color),
                                               cov3D_precomp=t(cov6))
    return img.cpu().numpy(), radii.cpu().numpy()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_splat_footprint_from_first_principles(name):
    _check_alpha_image(name, _render_oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_splat_footprint_from_first_principles(name):
    _check_alpha_image(name, _render_hip)


def test_guard_band_cases_really_clamp():
    """The guard-band cases must differ measurably from the unclamped Jacobian, or they would pin nothing."""
    for name in ("guard_band", "guard_band_corner"):
        b, cam, V, Pm, p, cov6, o = _case_inputs(name)
        a1, _, _, c_clamped = _expected_splat(p, cov6, o, cam, V, Pm, b["width"], b["height"], guard=True)
        a0, _, _, c_free = _expected_splat(p, cov6, o, cam, V, Pm, b["width"], b["height"], guard=False)
        assert np.abs(c_clamped - c_free).max() > 0.03 * np.abs(c_free).max()
        assert np.abs(a1 - a0).max() > 2e-3          # 100x the 2e-5 bar of the footprint tests


def test_radius_floor_known_answer():
    """Isotropic footprint: mid^2 - det = 0 < 0.1, so the radius is ceil(3 sqrt(s + sqrt(0.1))) with s = (f sigma / z)^2 + 0.3."""
    W = H = 48
    b, cam, V, Pm = _camera(W, H, azimuth=0.0)
    fx = b["width"] / (2 * cam["tanfovx"])
    for sigma in (0.01, 0.03, 0.07):
        z = 3.0
        p = _world_of_cam(np.array([0.0, 0.0, z]), V)
        cov6 = np.array([sigma ** 2, 0, 0, sigma ** 2, 0, sigma ** 2])
        _, radii = _render_oracle(p.astype(np.float32)[None], cov6.astype(np.float32)[None], np.array([[0.3]], np.float32),
                                  np.ones((1, 3), np.float32), np.zeros(3, np.float32), cam, b["width"], b["height"])
        s = (fx * sigma / z) ** 2 + 0.3
        assert int(radii[0]) == math.ceil(3 * math.sqrt(s + math.sqrt(0.1))), sigma


# ---------------------------------------------------------------------------------------------------------------------
# backward: the HIP backward against finite differences of the HIP forward, smooth regime
# ---------------------------------------------------------------------------------------------------------------------
def _smooth_scene(seed=3, n=6, W=32, H=32):
    """A few very wide, half-transparent splats: alpha >= 1/255 on EVERY pixel for every splat (sigma ~ image size), no
    saturation (T >= 0.5^6), no 0.99 clamp -- the rendered image is a smooth function of every input."""
    rng = np.random.default_rng(seed)
    b, cam, V, Pm = _camera(W, H, azimuth=0.2)
    z = rng.uniform(2.6, 3.4, n)
    t = np.stack([rng.uniform(-0.5, 0.5, n) * cam["tanfovx"] * z, rng.uniform(-0.5, 0.5, n) * cam["tanfovy"] * z, z], 1)
    means = np.stack([_world_of_cam(ti, V) for ti in t])
    fx = b["width"] / (2 * cam["tanfovx"])
    sig = rng.uniform(20.0, 30.0, (n, 3)) * (z[:, None] / fx)                   # 20-30 px on screen: 3 sigma_min > the image diagonal
    cov6 = np.stack([_aniso_cov6(sig[i], rng.normal(size=3), rng.uniform(0, 3.0)) for i in range(n)])
    op = rng.uniform(0.35, 0.5, (n, 1))
    col = rng.uniform(0.1, 0.9, (n, 3))
    return b, cam, means.astype(np.float32), cov6.astype(np.float32), op.astype(np.float32), col.astype(np.float32)


@pytest.mark.gpu
def test_hip_backward_matches_finite_differences_of_hip_forward():
    from d3ga_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    b, cam, means, cov6, op, col = _smooth_scene()
    W, H = b["width"], b["height"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                       bg=t(np.array([0.2, 0.4, 0.6], np.float32)), scale_modifier=1.0,
                                       viewmatrix=t(cam["world_view_transform"]), projmatrix=t(cam["full_proj_transform"]),
                                       sh_degree=0, campos=t(cam["camera_center"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(st)
    wts = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4)).to(dev)
    x = {"means3D": t(means), "cov3D_precomp": t(cov6), "opacities": t(op), "colors_precomp": t(col)}

    def loss(**over):
        a = dict(x); a.update(over)
        img, _, _ = rast(means2D=None, **a)
        return (img.double() * wts.double()).sum()

    # the regime really is smooth: every splat reaches every pixel, nothing saturates
    with torch.no_grad():
        img, radii, _ = rast(means2D=None, **x)
    assert int(radii.min()) >= max(W, H), "splats must cover the whole image"
    leaves = {k: v.clone().requires_grad_(True) for k, v in x.items()}
    loss(**leaves).backward()
    rng = np.random.default_rng(5)
    with torch.no_grad():
        # rounding of the float32 forward as seen by the loss (4 sigma of a random walk over the pixels)
        eps_loss = 4.0 * 6e-8 * float(((img.double() * wts.double()) ** 2).sum().sqrt())
    for k, v in x.items():
        g = leaves[k].grad.double()
        scale = float(v.abs().mean())
        for trial in range(4):
            d = torch.from_numpy(rng.normal(size=tuple(v.shape))).to(dev)
            if trial < 2:
                d = d.abs() * torch.sign(g)          # aligned with the gradient: no cancellation, the relative bar bites
            h = 2e-3 * scale / float(d.abs().mean())
            with torch.no_grad():
                fd = (loss(**{k: (v.double() + h * d).float()}) - loss(**{k: (v.double() - h * d).float()})) / (2 * h)
            an = (g * d).sum()
            # 2e-3 relative (truncation O(h^2) included) + the finite-difference noise of a float32 forward
            tol = 2e-3 * abs(float(an)) + eps_loss / h
            assert abs(float(fd - an)) <= tol, (k, trial, float(fd), float(an), tol)
            if trial < 2:
                assert eps_loss / h < 0.1 * abs(float(an)), "the aligned trials must be dominated by the relative bar"


def test_spatial_order_is_a_permutation_along_a_z_curve():
    """tetra.spatial_order: a permutation; points of one octant of the bounding box are numbered consecutively (the top three
    bits of the Morton code are the octant), ties keep their input order, degenerate inputs are accepted."""
    import torch
    from d3ga_amd.tetra import spatial_order
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(5000, 3, generator=g) * torch.tensor([2.0, 0.5, 1.0]) - 0.3
    order = spatial_order(pts)
    assert sorted(order.tolist()) == list(range(5000))
    lo, hi = pts.amin(0), pts.amax(0)
    octant = (((pts - lo) / (hi - lo) * 1023).round().long() >> 9)          # top bit per axis
    o = (octant[:, 0] + 2 * octant[:, 1] + 4 * octant[:, 2])[order]
    assert bool((o[1:] >= o[:-1]).all())
    # neighbours in the new numbering are close: mean distance between consecutive points far below the random order's
    d_new = (pts[order][1:] - pts[order][:-1]).norm(dim=1).mean()
    d_old = (pts[1:] - pts[:-1]).norm(dim=1).mean()
    assert float(d_new) < 0.2 * float(d_old)
    same = torch.zeros(7, 3)
    assert spatial_order(same).tolist() == list(range(7))                  # all codes equal: stable
    assert spatial_order(torch.zeros(0, 3)).numel() == 0


def test_merge_plan_of_the_deform_backward_reproduces_the_vertex_sums():
    """cage_deform.merge_plan (host logic of d3ga_cage_deform_bwd_merged): emulating the kernel on the CPU -- items dropped at
    item_pos inside their workgroup, every segment summed, a vertex's partials added -- gives index_add over the items, for a
    sorted binding, a random one, a partial last workgroup and a single workgroup."""
    import torch
    from d3ga_amd import cage_deform as cd
    g = torch.Generator().manual_seed(0)
    for T, V, P, sort in ((200, 90, 1500, True), (300, 120, 1024, False), (50, 30, 77, True), (10, 8, 256, False)):
        tetras = torch.randint(0, V, (T, 4), generator=g, dtype=torch.int32)
        tid = torch.randint(0, T, (P,), generator=g)
        tid = (torch.sort(tid)[0] if sort else tid).to(torch.int32)
        pl = cd.merge_plan(tetras, tid, V)
        vid = tetras.long()[tid.long()].reshape(-1)
        vals = torch.randn(4 * P, 3, generator=g, dtype=torch.float64)
        pos = pl["item_pos"].long().reshape(-1) & 0xffff
        begin = pl["seg_begin"].long() & 0xffff
        partials = torch.zeros(pl["n_segments"], 3, dtype=torch.float64)
        for b in range((P + 255) // 256):
            n = 4 * min(256, P - 256 * b)
            buf = torch.zeros(1024, 3, dtype=torch.float64)
            p_b = pos[1024 * b:1024 * b + n]
            assert sorted(p_b.tolist()) == list(range(n))                      # a permutation of the workgroup's slots
            buf[p_b] = vals[1024 * b:1024 * b + n]
            g0, g1 = int(pl["seg_ptr"][b]), int(pl["seg_ptr"][b + 1])
            for s_ in range(g0, g1):
                e = int(begin[s_ + 1]) if s_ + 1 < g1 else n
                partials[s_] = buf[int(begin[s_]):e].sum(0)
        out = torch.zeros(V, 3, dtype=torch.float64)
        vs, vp = pl["vert_start"].long(), pl["vert_parts"].long()
        for v in range(V):
            out[v] = partials[vp[vs[v]:vs[v + 1]]].sum(0)
        ref = torch.zeros(V, 3, dtype=torch.float64).index_add_(0, vid, vals)
        assert float((out - ref).abs().max()) < 1e-12
        if sort:
            assert pl["n_segments"] < 2 * P                                    # coherent numbering: runs are long
