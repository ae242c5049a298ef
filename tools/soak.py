"""A short real training run on synthetic data: Adam on the avatar parameters and the three field networks, the reference's
step (RGB + silhouette from one pass, L1 + SSIM, cage deform, LBS) -- the loss must fall, everything must stay finite, the
binning capacity must never overflow, and the split-weight cache must follow the optimizer's in-place updates."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from d3ga_amd import rasterizer as R
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
frame = bench.Frame(wl, dev, 0)
# a reachable target: the render of a perturbed avatar
with torch.no_grad():
    from d3ga_amd.renderer import render
    from d3ga_amd.cage_deform import cage_deform, lbs_cage
    p = frame.params
    tp = lbs_cage(frame.canon, p["delta_node"] + 0.01 * torch.randn_like(p["delta_node"]), frame.joint_mats, frame.skin_idx, frame.skin_w)
    m, c = cage_deform(tp, frame.tetras, frame.tetra_id, frame.barys0, frame.canon_grad, p["scaling"] + 0.1, p["rotation"], scale_activation="exp")
    tgt_pkg = {"means3D": m, "cov3D_precomp": c, "opacities": torch.sigmoid(p["opacity"] + 0.5),
               "shs": p["features"] * 0.8, "rgb": None, "sh_degree": frame.sh_degree}
    frame.target = render(frame.batch, tgt_pkg, frame.bg)["render"].clone()
    # the silhouette target is the perturbed avatar's own coverage (white on black), not a threshold of the RGB target: over a
    # white background that threshold is 1 everywhere and pulls every Gaussian across the whole screen
    frame.sil_rgb = torch.ones(m.shape[0], 3, device=dev)
    frame.bg0 = torch.zeros_like(frame.bg)
    frame.sil_target = render(frame.batch, tgt_pkg, frame.bg0, colors_precomp=frame.sil_rgb)["render"].clone()
frame.train_step(with_fields="color", pair=True)             # creates the networks
params = list(frame.params.values()) + frame.field_params + [frame.color_feat, frame.frame_enc]
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
opt = torch.optim.Adam(params, lr=lr)
names = list(frame.params.keys()) + [f"net{i}" for i in range(len(frame.field_params))] + ["color_feat", "frame_enc"]
losses, t0 = [], time.time()
for it in range(steps):
    opt.zero_grad(set_to_none=True)
    # with the reference's scale regulariser on the effective scales (cage_net.py:226, weight 175 as in train.py:203): without
    # it the synthetic objective drifts to its degenerate optimum, screen-filling Gaussians (D -> P x tiles)
    loss = frame.train_step(with_fields="color", pair=True, scale_weight=175.0)
    if os.environ.get("SOAK_TRACE"):
        torch.cuda.synchronize()
        bad = [n for n, q in zip(names, params) if q.grad is not None and not torch.isfinite(q.grad).all()]
        if bad or not torch.isfinite(loss):
            print("first non-finite at step", it, "loss", float(loss), "grads", bad[:12], flush=True)
            with torch.no_grad():
                d_bary, d_rot, d_scale = frame.canon_field(frame.params["rotation"], frame.params["scaling"], frame.barys0, frame.pose)
                raw = frame.params["scaling"] + d_scale
                q = frame.params["rotation"] + d_rot
                print("    raw log-scale max", float(raw.max()), "min", float(raw.min()), "| quaternion norm min", float(q.norm(dim=1).min()),
                      "| d_scale absmax", float(d_scale.abs().max()), "d_rot absmax", float(d_rot.abs().max()), flush=True)
            for n, q in zip(names, params):
                print("   ", n, "param finite", bool(torch.isfinite(q).all()), "absmax", float(q.detach().abs().max()), flush=True)
            sys.exit(1)
    opt.step()
    if it % 25 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        cnt = R.last_counters()
        bad = [n for n, q in zip(names, params) if q.grad is not None and not torch.isfinite(q.grad).all()]
        losses.append(float(loss.detach()))
        print(f"step {it:4d} loss {float(loss):.5f} D {cnt['D']} overflow {cnt['overflow']} nonfinite-grads {bad}", flush=True)
        assert not cnt["overflow"] and not bad and torch.isfinite(loss)
torch.cuda.synchronize()
print(f"{steps} steps in {time.time() - t0:.1f} s; loss {losses[0]:.5f} -> {losses[-1]:.5f}")
assert losses[-1] < 0.8 * losses[0], "the loss did not fall"
print("SOAK OK")
