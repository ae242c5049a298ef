"""Elimination timing of d3ga_raster_preprocess at a workload (needs a D3GA_DIAG=1 build): full / no histogram /
SH staging only."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from d3ga_amd import _lib, rasterizer as R  # noqa: E402
from d3ga_amd._lib import RasterParams, dptr, stream_handle  # noqa: E402
from d3ga_amd.cameras import batch_to_camera  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda", 0)
f = bench.Frame(wl, dev, 0)
from d3ga_amd.cage_deform import cage_deform, lbs_cage  # noqa: E402
with torch.no_grad():
    p = f.params
    tp = lbs_cage(f.canon, p["delta_node"], f.joint_mats, f.skin_idx, f.skin_w)
    means, cov6 = cage_deform(tp, f.tetras, f.tetra_id, f.barys0, f.canon_grad, torch.exp(p["scaling"]), p["rotation"])
    op = torch.sigmoid(p["opacity"]).contiguous()
    sh = p["features"].detach().contiguous()
cam = batch_to_camera(f.batch, device=dev)
P, W, H = means.shape[0], f.batch["width"], f.batch["height"]
cap = 4 * P
geom, binning, img = R._scratch(P, W, H, cap, dev)
radii = torch.empty(P, dtype=torch.int32, device=dev)
L = _lib.lib()
for name, dbg in (("full", 0), ("no histogram", 0x100), ("SH staging only", 0x200)):
    prm = RasterParams(P=P, M=16, sh_degree=3, W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, scale_modifier=1.0,
                       antialiasing=0, prefiltered=0, debug=dbg)
    def run():
        return L.d3ga_raster_preprocess(ctypes.byref(prm), dptr(means), dptr(sh), None, dptr(op), None, None, dptr(cov6),
                                        dptr(cam.world_view_transform), dptr(cam.full_proj_transform),
                                        dptr(cam.camera_center), dptr(geom), dptr(binning), cap, dptr(radii), stream_handle())
    for _ in range(5):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us (incl. the counter memset)")
