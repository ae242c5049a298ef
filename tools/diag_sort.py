import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from util import scene_inputs
from oracle import raster_c as rc
from d3ga_amd import rasterizer as R
DEV="cuda"
inp = scene_inputs("C1", scale_mult=30.0)
bg = torch.tensor([0.0, 0.3, 0.6])
s = R.GaussianRasterizationSettings(image_height=inp["H"], image_width=inp["W"], tanfovx=inp["cam"]["tanfovx"], tanfovy=inp["cam"]["tanfovy"], bg=bg.to(DEV), scale_modifier=1.0, viewmatrix=inp["view"].to(DEV), projmatrix=inp["proj"].to(DEV), sh_degree=0, campos=inp["campos"].to(DEV), prefiltered=False, debug=False, antialiasing=False)
rast = R.GaussianRasterizer(s)
with torch.no_grad():
    rast(means3D=inp["means3D"].to(DEV), means2D=None, opacities=inp["opacities"].to(DEV), colors_precomp=inp["rgb"].to(DEV), cov3D_precomp=inp["cov6"].to(DEV))
torch.cuda.synchronize()
binning, cap = R._last[0]
print("counters", binning[:32].view(torch.int32).tolist(), "cap", cap)
start, plist, keys = R.last_tile_lists(inp["W"], inp["H"])
npy = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())
cam = inp["cam"]
_,_,_,ctx = rc.forward(npy(inp["means3D"]), npy(inp["opacities"]), npy(bg), cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"], cam["tanfovy"], inp["W"], inp["H"], cov3D_precomp=npy(inp["cov6"]), colors_precomp=npy(inp["rgb"]))
ostart, olist = rc.tile_lists(ctx)
st = npy(start); pl = npy(plist)
bad=[]
for t in range(len(st)-1):
    a,b = st[t], st[t+1]
    if not np.array_equal(pl[a:b], olist[a:b]): bad.append((t, b-a, int((pl[a:b]==0).mean()*100)))
print("n tiles", len(st)-1, "bad", len(bad), bad[:20])
cnts = np.diff(st); print("counts min/max", cnts.min(), cnts.max(), "n>8192:", (cnts>8192).sum(), "2048<n<=8192:", ((cnts>2048)&(cnts<=8192)).sum())
