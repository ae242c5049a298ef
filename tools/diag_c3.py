"""Diagnostic: where do GPU and oracle gradients differ most at C3?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import scene_inputs
from oracle import raster_c as rc
from d3ga_amd import rasterizer as R
DEV = "cuda"
inp = scene_inputs("C3")
bg = torch.tensor([1.0, 1.0, 1.0])
npy = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())
cu = lambda t: t.detach().to(DEV).clone().requires_grad_(True)
means, cov, op, sh = (cu(inp[k]) for k in ("means3D", "cov6", "opacities", "shs"))
m2d = torch.zeros_like(means, requires_grad=True)
s = R.GaussianRasterizationSettings(image_height=inp["H"], image_width=inp["W"], tanfovx=inp["cam"]["tanfovx"],
    tanfovy=inp["cam"]["tanfovy"], bg=bg.to(DEV), scale_modifier=1.0, viewmatrix=inp["view"].to(DEV),
    projmatrix=inp["proj"].to(DEV), sh_degree=3, campos=inp["campos"].to(DEV), prefiltered=False, debug=False, antialiasing=False)
rast = R.GaussianRasterizer(s)
color, radii, _ = rast(means3D=means, means2D=m2d, opacities=op, shs=sh, cov3D_precomp=cov)
gpix = torch.randn(3, inp["H"], inp["W"], generator=torch.Generator().manual_seed(11))
(color * gpix.to(DEV)).sum().backward()
cam = inp["cam"]
ocolor, oradii, _, ctx = rc.forward(npy(inp["means3D"]), npy(inp["opacities"]), npy(bg), cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], inp["W"], inp["H"], cov3D_precomp=npy(inp["cov6"]), shs=npy(inp["shs"]), sh_degree=3)
og = rc.backward(ctx, npy(gpix))
d = np.abs(npy(color) - ocolor)
print("image max diff", d.max(), "pixels > 1e-4:", (d > 1e-4).sum())
geom = rc.geom(ctx)
for name, mine, ref in (("means3D", means.grad, og["means3D"]), ("means2D", m2d.grad, og["means2D"]), ("cov3D", cov.grad, og["cov3D"]), ("opac", op.grad, og["opacities"]), ("sh", sh.grad, og["shs"])):
    a, b = npy(mine).reshape(len(ref), -1), ref.reshape(len(ref), -1)
    diff = np.abs(a - b).max(1)
    print(name, "rel", diff.max() / np.abs(b).max(), "max|ref|", np.abs(b).max())
    idx = np.argsort(-diff)[:5]
    for i in idx:
        print("   i", i, "diff", diff[i], "gpu", a[i][:3], "ref", b[i][:3], "radius", oradii[i], "opacity", float(inp["opacities"][i]), "depth", geom["depth"][i], "xy", geom["xy"][i])
