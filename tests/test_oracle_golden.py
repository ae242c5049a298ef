"""The CPU oracle against golden vectors captured from the reference's own Python (tools/gen_golden.py)."""
import numpy as np
import torch

from oracle import camera as oc
from oracle import deform as od
from oracle import raster_torch as rt


def test_camera_matches_reference(golden):
    g = golden("camera_cases.npz")
    for i in range(int(g["n"])):
        fovx, fovy = g[f"fov{i}"]
        cam = oc.camera(g[f"R{i}"], g[f"T{i}"], float(fovx), float(fovy))
        np.testing.assert_allclose(cam["world_view_transform"], g[f"wv{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(cam["projection_matrix"], g[f"proj{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(cam["full_proj_transform"], g[f"full{i}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam["camera_center"], g[f"center{i}"], rtol=1e-5, atol=1e-5)


def test_cov_helpers_match_reference(golden):
    g = golden("cov_cases.npz")
    s, q = torch.from_numpy(g["scales"]), torch.from_numpy(g["rotations"])
    np.testing.assert_allclose(od.quat_to_rotmat(q).numpy(), g["R"], atol=1e-6)
    cov = od.pack_sym6(od.covariance_from_scale_rot(s, q))
    np.testing.assert_allclose(cov.numpy(), g["cov6"], rtol=1e-5, atol=1e-9)


def _deform_case(g, dtype):
    t = lambda k: torch.from_numpy(g[k])
    tp = t("tetpoints").to(dtype).requires_grad_(True)
    barys = t("canon_barys").to(dtype).requires_grad_(True)
    scaling = t("scaling_param").to(dtype).requires_grad_(True)
    rotation = t("rotation_param").to(dtype).requires_grad_(True)
    scales = torch.exp(scaling + t("d_scale").to(dtype))
    rots = torch.nn.functional.normalize(rotation + t("d_rot").to(dtype))
    cg = od.canonical_gradient(t("canon_points").to(dtype), t("tetras"), t("tetra_id"))
    means, cov6 = od.cage_deform(tp, t("tetras"), t("tetra_id"), barys, cg, scales, rots)
    loss = (means * t("up_grad_means").to(dtype)).sum() + (cov6 * t("up_grad_cov").to(dtype)).sum()
    loss.backward()
    return means, cov6, cg, tp, barys, scaling, rotation


def test_deform_matches_reference_forward_and_grads(golden):
    for name in ("deform_case0.npz", "deform_case1.npz"):
        g = golden(name)
        means, cov6, cg, tp, barys, scaling, rotation = _deform_case(g, torch.float32)
        np.testing.assert_allclose(cg.numpy(), g["canonical_gradient"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(means.detach().numpy(), g["means3D"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cov6.detach().numpy(), g["cov3D_precomp"], rtol=2e-4, atol=1e-10)
        scale = lambda a: np.abs(a).max()
        for mine, ref in ((tp.grad, g["grad_tetpoints"]), (barys.grad, g["grad_barys"]),
                          (scaling.grad, g["grad_scaling_param"]), (rotation.grad, g["grad_rotation_param"])):
            assert np.abs(mine.numpy() - ref).max() <= 1e-3 * scale(ref) + 1e-9


def test_deform_float64_agrees_with_float32_reference(golden):
    g = golden("deform_case0.npz")
    means, cov6, *_ = _deform_case(g, torch.float64)
    np.testing.assert_allclose(means.detach().numpy(), g["means3D"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cov6.detach().numpy(), g["cov3D_precomp"], rtol=2e-4, atol=1e-10)


def test_fem_energy_matches_reference(golden):
    g = golden("deform_case0.npz")
    e = od.fem_energy(torch.from_numpy(g["tetpoints"]), torch.from_numpy(g["tetras"]), torch.from_numpy(g["Dn_inv"]))
    np.testing.assert_allclose(e.mean().numpy(), g["fm_energy"][0], rtol=1e-4)


def test_canonical_means_and_shs_layout(golden):
    g = golden("deform_case0.npz")
    assert g["shs"].shape == (g["means3D"].shape[0], 16, 3)
    assert g["opacities"].shape == (g["means3D"].shape[0], 1)
    cm = (torch.from_numpy(g["canon_points"])[torch.from_numpy(g["tetras"])][torch.from_numpy(g["tetra_id"])]
          * torch.from_numpy(g["barys"])[:, :, None]).sum(1)
    np.testing.assert_allclose(cm.numpy(), g["canonical_means3D"], rtol=1e-5, atol=1e-6)


def test_lbs_matches_reference(golden):
    g = golden("lbs_case.npz")
    V, J = g["weights"].shape
    idx = torch.arange(J, dtype=torch.int32)[None].repeat(V, 1)           # dense weights as a K=J sparse table
    out = od.lbs_cage(torch.from_numpy(g["template"]), torch.from_numpy(g["delta"]), torch.from_numpy(g["A"]), idx,
                      torch.from_numpy(g["weights"]), torch.from_numpy(g["Rh"]), torch.from_numpy(g["Th"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-5)


def test_sh_constants_match_reference(golden):
    g = golden("sh_consts.npz")
    assert rt.SH_C0 == float(g["C0"]) and rt.SH_C1 == float(g["C1"])
    np.testing.assert_array_equal(np.array(rt.SH_C2), g["C2"])
    np.testing.assert_array_equal(np.array(rt.SH_C3), g["C3"])


def test_rigid_cage_motion_property():
    """Rigid motion of the cage => J = R, cov' = R Sigma R^T, means' = R means + t."""
    from d3ga_amd import synthetic as syn
    sc = syn.make_scene("T0")
    canon, tets, tid = sc["canon_points"].double(), sc["tetras"], sc["tetra_id"]
    cg = od.canonical_gradient(canon, tets, tid)
    R = torch.from_numpy(syn.rodrigues(np.array([0.3, -0.5, 0.2]))).double()
    tvec = torch.tensor([0.1, -0.2, 0.3], dtype=torch.float64)
    scales, rots = torch.exp(sc["scaling"]).double(), sc["rotation"].double()
    m0, c0 = od.cage_deform(canon, tets, tid, sc["barys"].double(), cg, scales, rots)
    m1, c1 = od.cage_deform(canon @ R.T + tvec, tets, tid, sc["barys"].double(), cg, scales, rots)
    np.testing.assert_allclose(m1.numpy(), (m0 @ R.T + tvec).numpy(), atol=1e-6)   # float32 barys sum to 1 +- 1e-7
    S0 = od.unpack_sym6(c0)
    np.testing.assert_allclose(od.unpack_sym6(c1).numpy(), (R @ S0 @ R.T).numpy(), atol=1e-12)
    np.testing.assert_allclose(c0.numpy(), od.pack_sym6(od.covariance_from_scale_rot(scales, rots)).numpy(), atol=1e-12)


def test_loss_oracle_matches_reference_losses(golden):
    """oracle/losses.py against values and autograd gradients of the reference's own l1_loss / ssim
    (utils/loss_utils.py:29,59-86), tests/golden/loss_cases.npz."""
    import torch
    from oracle import losses as ol
    g = golden("loss_cases.npz")
    for name in ("a", "b", "c"):
        pred = torch.from_numpy(g[f"{name}_pred"]).requires_grad_(True)
        gt = torch.from_numpy(g[f"{name}_gt"])
        v = ol.ssim(pred, gt)
        (gr,) = torch.autograd.grad(v, pred)
        np.testing.assert_allclose(v.detach().numpy(), g[f"{name}_ssim"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(gr.numpy(), g[f"{name}_ssim_grad"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(ol.l1_loss(pred, gt).detach().numpy(), g[f"{name}_l1"], rtol=0, atol=1e-7)


def _field_weights(g, prefix, n_hidden=4):
    hidden = [(torch.from_numpy(g[f"{prefix}_w_network.{i}.weight"]), torch.from_numpy(g[f"{prefix}_w_network.{i}.bias"]))
              for i in range(n_hidden)]
    return hidden, torch.from_numpy(g[f"{prefix}_w_output.weight"]), torch.from_numpy(g[f"{prefix}_w_output.bias"])


def test_field_oracle_matches_reference_fields(golden):
    """oracle/mlp.py against outputs and autograd gradients of the reference's own CanonicalField / DeformationField
    (models/mlp.py:39-110), tests/golden/field_cases.npz -- including the argument-order quirk of cage_net.py:199-204."""
    from oracle import mlp as om
    g = golden("field_cases.npz")
    hidden, ow, ob = _field_weights(g, "cf")
    leaf = lambda k: torch.from_numpy(g[k]).requires_grad_(True)
    barys, rots, scales, pose = leaf("cf_barys"), leaf("cf_rots"), leaf("cf_scales"), leaf("cf_pose")
    # the reference calls canonical_field(rotation, scales, barys, cond) against (barys, rots, scales, pose)
    outs = om.canonical_field(rots, scales, barys, pose, hidden, ow, ob)
    for o, k in zip(outs, ("cf_d_bary", "cf_d_rot", "cf_d_scale")):
        np.testing.assert_allclose(o.detach().numpy(), g[k], rtol=1e-5, atol=1e-6)
    grads = torch.autograd.grad(list(outs), [barys, rots, scales, pose], [torch.from_numpy(g[f"cf_up{i}"]) for i in range(3)])
    for gr, k in zip(grads, ("cf_g_barys", "cf_g_rots", "cf_g_scales", "cf_g_pose")):
        np.testing.assert_allclose(gr.numpy(), g[k], rtol=1e-4, atol=1e-6)
    hidden, ow, ob = _field_weights(g, "df")
    canon, pose = leaf("df_canon"), leaf("df_pose")
    delta = om.deformation_field(canon, pose, hidden, ow, ob, scaling=0.07)
    np.testing.assert_allclose(delta.detach().numpy(), g["df_delta"], rtol=1e-5, atol=1e-7)
    gc, gp = torch.autograd.grad(delta, [canon, pose], torch.from_numpy(g["df_up"]))
    np.testing.assert_allclose(gc.numpy(), g["df_g_canon"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gp.numpy(), g["df_g_pose"], rtol=1e-4, atol=1e-6)
    # ShadowDecoder / FaceDecoder (models/mlp.py:235-297)
    hidden, ow, ob = _field_weights(g, "sd")
    pose = leaf("sd_pose")
    ao = om.shadow_decoder(torch.from_numpy(g["sd_template"]), pose, hidden, ow, ob)
    np.testing.assert_allclose(ao.detach().numpy(), g["sd_ao"], rtol=1e-5, atol=1e-7)
    (gp,) = torch.autograd.grad(ao, [pose], torch.from_numpy(g["sd_up"]))
    np.testing.assert_allclose(gp.numpy(), g["sd_g_pose"], rtol=1e-4, atol=1e-7)
    hidden, ow, ob = _field_weights(g, "fd")
    kpt = leaf("fd_kpt")
    code = om.face_decoder(kpt, hidden, ow, ob)
    np.testing.assert_allclose(code.detach().numpy(), g["fd_code"], rtol=1e-5, atol=1e-6)
    (gk,) = torch.autograd.grad(code, [kpt], torch.from_numpy(g["fd_up"]))
    np.testing.assert_allclose(gk.numpy(), g["fd_g_kpt"], rtol=1e-4, atol=1e-7)
    # ColorField (models/mlp.py:152-232) -- the reference module with tiny-cuda-nn's encoding replaced by the oracle's stand-in
    hidden, ow, ob = _field_weights(g, "col", n_hidden=5)
    feat, pose, vd, frame = leaf("col_feat"), leaf("col_pose"), leaf("col_viewdir"), leaf("col_frame")
    rgb, opa = om.color_field(feat, pose, vd, frame, None, None, hidden, ow, ob)
    np.testing.assert_allclose(rgb.detach().numpy(), g["col_rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(opa.detach().numpy(), g["col_opacity"], rtol=1e-5, atol=1e-6)
    grads = torch.autograd.grad([rgb, opa], [feat, pose, vd, frame], [torch.from_numpy(g["col_up0"]), torch.from_numpy(g["col_up1"])])
    for gr, k in zip(grads, ("col_g_feat", "col_g_pose", "col_g_viewdir", "col_g_frame")):
        np.testing.assert_allclose(gr.numpy(), g[k], rtol=1e-4, atol=1e-6)


def test_goliath_skinning_golden(golden):
    """D0's second cited source (lbsmodel/body_model.py:208-234 LinearBlendSkinning.skinning on :350-387 states_to_matrix, the
    8-sparse form): the oracle's skeleton_matrices + lbs_cage and the product's host-side skeleton_matrices against the fixture
    tools/gen_golden.py (G8) generated by calling the reference's own functions."""
    from d3ga_amd.cage_deform import skeleton_matrices
    g = golden("lbs_goliath_case.npz")
    t = lambda k: torch.from_numpy(g[k])
    M = od.skeleton_matrices(t("bind_state").double(), t("target_states").double())
    np.testing.assert_allclose(M[:, :, :3, :].numpy(), g["mat"], rtol=1e-5, atol=3e-6)
    assert float((M[:, :, 3, :] - torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64)).abs().max()) < 1e-12
    Mh = skeleton_matrices(t("bind_state"), t("target_states"))
    np.testing.assert_allclose(Mh[:, :, :3, :].numpy(), g["mat"], rtol=1e-5, atol=3e-6)
    v = t("vertices").double().requires_grad_(True)
    tot = 0.0
    for b in range(g["target_states"].shape[0]):
        out = od.lbs_cage(v, None, M[b], t("skin_indices"), t("skin_weights").double())
        np.testing.assert_allclose(out.detach().numpy(), g["out"][b], rtol=1e-5, atol=3e-6)
        tot = tot + (out * t("grad_out")[b].double()).sum()
    tot.backward()
    np.testing.assert_allclose(v.grad.numpy(), g["grad_vertices"], rtol=1e-4, atol=2e-6)
