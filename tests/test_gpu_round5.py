"""Round 5: the backward's block-list prefix, the opt-in two-launch forward, ABI 101 refusals (GPU)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import scene_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _render(name, scale_mult, seed=3):
    from d3ga_amd import rasterizer as R
    from test_gpu_parity import _settings
    inp = scene_inputs(name, scale_mult=scale_mult)
    bg = torch.tensor([0.3, 0.6, 0.1])
    rast = R.GaussianRasterizer(_settings(inp, bg, 3))
    leaves = {k: inp[k].to(DEV).clone().requires_grad_(True) for k in ("means3D", "cov6", "opacities", "shs")}
    img, radii, _ = rast(means3D=leaves["means3D"], means2D=None, opacities=leaves["opacities"], shs=leaves["shs"],
                         cov3D_precomp=leaves["cov6"])
    gpix = torch.randn(img.shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
    return inp, R, img, radii, leaves, gpix


@pytest.mark.parametrize("name,scale_mult", [("T1", 3.0), ("C1", 1.0), ("T1", 8.0)])
def test_block_count_is_the_prefix_up_to_the_last_blended_entry(name, scale_mult):
    """blk_count of a block = number of entries of its list up to (and including) the last one some pixel of the block blended:
    the entry in front of that cut carries the largest n_contrib of the block's 16 pixels, nothing behind it is needed by the
    backward.  A block none of whose pixels blended anything has count 0.  (Rounds 2-4 handed the backward every emitted entry.)"""
    inp, R, img, _, _, _ = _render(name, scale_mult)
    W, H = inp["W"], inp["H"]
    torch.cuda.synchronize()
    cnt, lst = R.last_block_lists()
    _, ncontrib = R.last_termination()
    start, _, _ = R.last_tile_lists(W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    cnt, lst, ncontrib, start = cnt.cpu().numpy(), lst.cpu().numpy(), ncontrib.cpu().numpy().astype(np.int64), start.cpu().numpy()
    pad = np.zeros((gy * 16, gx * 16), np.int64)
    pad[:H, :W] = ncontrib
    checked = 0
    for t in range(gx * gy):
        b0, e0 = int(start[t]), int(start[t + 1])
        if e0 == b0:
            continue
        ty, tx = divmod(t, gx)
        for b in range(16):
            q, r = b >> 2, b & 3
            bx, by = 2 * (q & 1) + (r & 1), 2 * (q >> 1) + (r >> 1)
            if tx * 16 + 8 * (q & 1) >= W or ty * 16 + 8 * (q >> 1) >= H:
                continue                                   # (the forward writes the counts of quadrants that start inside the image)
            last = int(pad[ty * 16 + 4 * by: ty * 16 + 4 * by + 4, tx * 16 + 4 * bx: tx * 16 + 4 * bx + 4].max())
            c = int(cnt[t, b])
            if last == 0:
                assert c == 0
                continue
            entries = lst[16 * b0 + b * (e0 - b0): 16 * b0 + b * (e0 - b0) + c]
            assert c > 0 and int(entries[-1, 0]) == last, (t, b, c, last, entries[-3:])
            assert (np.diff(entries[:, 0]) > 0).all()        # list order = position order
            checked += 1
    assert checked > 0


_CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from test_gpu_round5 import _render
out = {{}}
for name, sm in (("T1", 3.0), ("C1", 1.0), ("T1", 8.0)):
    inp, R, img, radii, leaves, gpix = _render(name, sm)
    (img * gpix).sum().backward()
    torch.cuda.synchronize()
    T, n = R.last_termination()
    out[f"{{name}}_{{sm}}_img"] = img.detach().cpu().numpy(); out[f"{{name}}_{{sm}}_n"] = n.cpu().numpy()
    for k, v in leaves.items():
        out[f"{{name}}_{{sm}}_{{k}}"] = v.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_backward_without_the_precomputed_covariance_is_refused():
    """ABI 101: the forward keeps no copy of a precomputed covariance, so a per-Gaussian backward that is handed
    neither cov3D_precomp nor (scales, rotations) returns D3GA_E_NULL instead of reading uninitialised records."""
    from d3ga_amd import _lib
    L = _lib.lib()
    assert L.d3ga_version() == _lib.ABI_VERSION
    prm = _lib.RasterParams(P=16, M=0, sh_degree=0, W=64, H=64, tanfovx=1.0, tanfovy=1.0, scale_modifier=1.0, antialiasing=0,
                            prefiltered=0, debug=0, opacity_activation=0, forward_only=0, acc_self_clearing=0)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    p = ctypes.c_void_p(buf.data_ptr())
    st = L.d3ga_raster_preprocess_bwd(ctypes.byref(prm), p, None, None, None, None, p, p, p, p, p, p, None, None, None, p, None, None,
                                      None, None)
    assert st == -1, st                                      # D3GA_E_NULL


def test_lbs_cage_deform_equals_the_two_operators():
    """d3ga_amd.cage_deform.lbs_cage_deform (LBS + cage deform as one autograd node; the LBS backward formed in the vertex-gather
    launch: d3ga_cage_deform_bwd_merged_lbs) against lbs_cage() followed by cage_deform(): identical outputs, the same gradients
    (the sums are formed in the same order: bit for bit but for the skinning's own reduction), also with a second route into
    the posed vertices (an extra term on tetpoints, as the FEM regulariser adds one)."""
    from d3ga_amd.cage_deform import cage_deform, lbs_cage, lbs_cage_deform, canonical_gradient
    from d3ga_amd import synthetic as syn
    sc = syn.make_scene("T1", seed=5)
    d = lambda t: t.to(DEV)
    canon, tetras, tid, barys = d(sc["canon_points"]), d(sc["tetras"]), d(sc["tetra_id"]), d(sc["barys"])
    jm, si, sw = d(sc["joint_mats"]), d(sc["skin_idx"]), d(sc["skin_w"])
    cg = canonical_gradient(canon, tetras, tid).contiguous()
    g = torch.Generator().manual_seed(11)
    Rh = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].to(DEV)
    Th = torch.randn(3, generator=g).to(DEV)
    P, V = barys.shape[0], canon.shape[0]
    wm, wc, wt = torch.randn(P, 3, generator=g).to(DEV), torch.randn(P, 6, generator=g).to(DEV), torch.randn(V, 3, generator=g).to(DEV)
    res = {}
    for fused in (False, True):
        leaves = dict(delta=d(sc["delta_node"]).clone().requires_grad_(True), scaling=d(sc["scaling"]).clone().requires_grad_(True),
                      rot=d(sc["rotation"]).clone().requires_grad_(True), dbary=torch.zeros(P, 4, device=DEV, requires_grad=True))
        if fused:
            m, c, tp = lbs_cage_deform(canon, leaves["delta"], jm, si, sw, tetras, tid, barys, cg, leaves["scaling"], leaves["rot"],
                                       delta_barys=leaves["dbary"], scale_activation="exp", Rh=Rh, Th=Th)
        else:
            tp = lbs_cage(canon, leaves["delta"], jm, si, sw, Rh, Th)
            m, c = cage_deform(tp, tetras, tid, barys, cg, leaves["scaling"], leaves["rot"], delta_barys=leaves["dbary"],
                               scale_activation="exp")
        ((m * wm).sum() + (c * wc).sum() + (tp * wt).sum()).backward()
        torch.cuda.synchronize()
        res[fused] = dict(m=m.detach(), c=c.detach(), tp=tp.detach(), **{k: v.grad for k, v in leaves.items()})
    a, b = res[False], res[True]
    for k in ("m", "c", "tp", "scaling", "rot", "dbary"):
        assert torch.equal(a[k], b[k]), k
    scale = float(a["delta"].abs().max())
    assert float((a["delta"] - b["delta"]).abs().max()) <= 2e-6 * scale
    assert float(b["delta"].abs().max()) > 0


def test_backward_with_split_heavy_tiles_equals_the_unsplit_one(tmp_path):
    """Round 5 (last session): the heaviest tiles of the work order get TWO workgroups each in the compositing backward (their blocks
    dealt by rank parity, two blocks per wavefront walked over two DPP rows with wave-wide scans: the R = 2 walk), by default as
    many as the order kernel counts (D3GA_CNT_HEAVY).  The forward is untouched, so images and termination are bit-identical;
    gradients are the same sums in another order (2e-5 of the largest element).  The knob `bwd_split` (d3ga_debug_set, applied by
    d3ga_amd/_lib.py from D3GA_KNOBS) = 7 splits seven tiles whatever their length (a split tile with short or empty halves), 100000
    every tile the grid has (capped by the tile count)."""
    files = {}
    for split in ("0", "-1", "7", "100000"):
        f = str(tmp_path / f"split{split}.npz")
        env = dict(os.environ, D3GA_KNOBS=f"bwd_split={split}")
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        files[split] = np.load(f)
    a = files["0"]
    for split in ("-1", "7", "100000"):
        b = files[split]
        for k in a.files:
            if k.endswith(("_n", "_img")):
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{split} {k}")
            else:
                scale = np.abs(a[k]).max() + 1e-30
                assert np.abs(a[k] - b[k]).max() <= 2e-5 * scale, (split, k, np.abs(a[k] - b[k]).max() / scale)


def test_order_kernel_counts_the_tiles_the_backward_splits():
    """D3GA_CNT_HEAVY (ABI 103) = (non-empty tiles + 9) // 10 while the frame has at most 4096 non-empty tiles."""
    from d3ga_amd import rasterizer as rz
    inp, R, img, radii, leaves, gpix = _render("C1", 1.0)
    torch.cuda.synchronize()
    start = rz.last_tile_lists(inp["W"], inp["H"])[0].cpu().numpy().astype(np.int64)
    nonempty = int((np.diff(start) > 0).sum())
    assert 0 < nonempty <= 4096
    counters = rz._last[torch.cuda.current_device()][0][:32].view(torch.int32).cpu().numpy()
    assert int(counters[7]) == (nonempty + 9) // 10, (counters, nonempty)


@pytest.mark.parametrize("deg,M", [(3, 16), (1, 16), (1, 4), (2, 9), (2, 12), (0, 1)])
def test_view_direction_term_of_the_mean_gradient_comes_from_the_forwards_jacobian(deg, M):
    """Round 5: preprocess leaves d(SH colour)/d(unit view direction) per Gaussian (GeomBuf::dcol) and preprocess_bwd forms the
    direction term of dL/dmean from it without reading the coefficients.  Isolated here: the same scene rendered once with the SH
    coefficients and once with colors_precomp = the colours those coefficients give -- same alphas, same pixel gradient, so the two
    dL/dmeans3D differ by exactly  sum_c dL/dcolour_c . d colour_c / d mean  (clamped channels excluded), which torch forms in
    float64 from the oracle's eval_sh.  Also: the SH gradient itself and dL/dcolour agree between the two renders.
    (deg, M): M = 16 and 12 take the staged path with full / partial rows and the Jacobian record, M = 4 the staged path with short
    rows, M = 9 and 1 (3 M not a multiple of 4) the direct path, where the backward still reads the coefficients; degree 0 has no
    view dependence at all.)"""
    from d3ga_amd import rasterizer as R
    from oracle import raster_torch as rt
    from test_gpu_parity import _settings
    inp = scene_inputs("T1", scale_mult=3.0)
    shs = inp["shs"][:, :M].clone()
    shs[::3, 0, :] = -2.5                                   # some channels clamp at 0 (no gradient through them)
    shs[:, 1:] *= 4.0                                       # a strong view dependence: the term under test is not a rounding-level share
    bg = torch.tensor([0.1, 0.2, 0.3])
    rast = R.GaussianRasterizer(_settings(inp, bg, deg))
    m64 = inp["means3D"].double()
    d = m64 - inp["campos"].double()[None]
    col_raw = rt.eval_sh(deg, shs.double(), d / d.norm(dim=1, keepdim=True)) + 0.5
    col = col_raw.clamp_min(0.0).float()
    gpix = torch.randn(3, inp["H"], inp["W"], generator=torch.Generator().manual_seed(9)).to(DEV)

    def run(**colour):
        leaves = {k: v.to(DEV).clone().requires_grad_(True) for k, v in dict(means3D=inp["means3D"], **colour).items()}
        img, radii, _ = rast(means2D=None, opacities=inp["opacities"].to(DEV), cov3D_precomp=inp["cov6"].to(DEV), **leaves)
        (img * gpix).sum().backward()
        torch.cuda.synchronize()
        return {k: v.grad.cpu() for k, v in leaves.items()}, radii.cpu()
    g_sh, radii = run(shs=shs)
    g_pre, _ = run(colors_precomp=col)
    gcol = g_pre["colors_precomp"].double()
    assert float(gcol.abs().max()) > 0 and int((radii > 0).sum()) > 100
    # expected direction term, float64: clamped channels pass no gradient (upstream's clamped[] flags)
    mm = m64.clone().requires_grad_(True)
    dd = mm - inp["campos"].double()[None]
    c = rt.eval_sh(deg, shs.double(), dd / dd.norm(dim=1, keepdim=True)) + 0.5
    live = (col_raw > 0).double()
    if deg > 0:
        (c * gcol * live).sum().backward()
    want = mm.grad if mm.grad is not None else torch.zeros_like(m64)      # (degree 0: the colour does not see the direction)
    got = (g_sh["means3D"].double() - g_pre["means3D"].double())
    scale_term, scale_all = float(want.abs().max()), float(g_pre["means3D"].abs().max())
    if deg > 0:
        assert scale_term > 1e-3 * scale_all                 # the term is visible beside the geometry term it rides on
    err = float((got - want).abs().max())
    assert err <= 2e-3 * scale_term + 4e-6 * scale_all, (err, scale_term, scale_all)
    # the SH gradient is basis x (clamp-masked dL/dcolour), the colour gradient of the precomputed render is the unmasked one
    B = torch.zeros(shs.shape[0], shs.shape[1], dtype=torch.float64)
    for k in range((deg + 1) ** 2):
        e = torch.zeros(shs.shape[0], shs.shape[1], 3, dtype=torch.float64)
        e[:, k, :] = 1.0
        B[:, k] = rt.eval_sh(deg, e, (d / d.norm(dim=1, keepdim=True)))[:, 0]
    want_sh = B[:, :, None] * (gcol * live)[:, None, :]
    assert float((g_sh["shs"].double() - want_sh).abs().max()) <= 1e-4 * float(want_sh.abs().max())


@pytest.mark.parametrize("deg,M", [(3, 16), (2, 12), (1, 4), (2, 9)])
def test_inference_and_training_forward_give_the_same_image_bit_for_bit(deg, M):
    """The forward a backward follows also leaves d(colour)/d(direction) (another instantiation of preprocess_kernel, other
    registers live around the SH sum): basis, direction and the SH sum are contraction-free / explicit fmaf chains so that its
    colours -- and the image -- are bit-identical to the inference forward's, for every staging path."""
    from d3ga_amd import rasterizer as R
    from test_gpu_parity import _settings
    inp = scene_inputs("T1", scale_mult=3.0)
    rast = R.GaussianRasterizer(_settings(inp, torch.tensor([0.2, 0.4, 0.6]), deg))
    args = dict(means3D=inp["means3D"].to(DEV), means2D=None, opacities=inp["opacities"].to(DEV),
                shs=inp["shs"][:, :M].contiguous().to(DEV), cov3D_precomp=inp["cov6"].to(DEV))
    with torch.no_grad():
        img_inf, radii_inf, _ = rast(**args)
    img_g, radii_g, _ = rast(**dict(args, means3D=args["means3D"].clone().requires_grad_(True)))
    assert torch.equal(img_inf, img_g.detach()) and torch.equal(radii_inf, radii_g)
    assert float(img_inf.std()) > 0
