#!/usr/bin/env python3
"""bench.py -- fwd+bwd frames/s of the D3GA deform-and-rasterize hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" on a rank = one frame of the hot path over synthetic inputs already resident in HBM:
    LBS of the cage (D0) -> cage_deform (D1-D5) -> render() (R1-R4, SH degree 3) -> L1 loss -> backward through the
    rasterizer (R5-R6), the deform (A1) and the LBS, down to the avatar parameters
    (cage-vertex offsets, barycentric offsets, scaling, rotation, opacity, SH).
With N ranks every rank renders its own view of the same pose (camera sharding, weak scaling) and the parameter
gradients are summed with one RCCL all-reduce per step.  value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     dominant compositing kernel vs the HBM roofline (algorithmic bytes of SURVEY.md sec. 8d / HIP-event time)
  "cpu_baseline": the oracle (CPU port: torch deform + C rasterizer) timed on this box's host cores, baseline only.
"""
import argparse
import json
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable in a copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--stage-burst", type=int, default=4,
                    help="stage events: launches of each compositing kernel between its two HIP events (queue kept full: the "
                         "quotient is the kernel's duration, not duration + dispatch latency; 1 = rounds 1-4 behaviour)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for tests)")
    ap.add_argument("--single-device", action="store_true", help="testing only: all ranks share cuda:0")
    ap.add_argument("--reduce", choices=("cut", "params"), default="cut",
                    help="N>1 gradient exchange: 'cut' = summed at the rasterizer's inputs with the SH gradient in factored "
                         "form (dist.ViewShardedGrads); 'params' = one all-reduce per parameter tensor (dist.GradReducer)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the 2-render training-step variant")
    ap.add_argument("--train-step", choices=("color",), default=None,
                    help="instead of the frame: the reference's REAL training step (configs/actorshq_actor02.yml, use_shs false) -- two "
                         "renders, DeformationField / CanonicalField / ColorField, L1 + SSIM + silhouette + scale regulariser, "
                         "clip_grad_norm_, Adam -- on N ranks, one view per rank, with the parameter gradients averaged by per-bucket "
                         "asynchronous all-reduces issued while the backward still runs (dist.BucketedGradReducer).  BASELINE "
                         "config 4; default workload C4 (135k Gaussians, 748x1022)")
    ap.add_argument("--field-mlp", action="store_true",
                    help="instead of the frame: the per-Gaussian CanonicalField network (SURVEY sec. 8f rank 1) forward + "
                         "backward at the workload's Gaussian count, roofline against the f32 MFMA peak")
    ap.add_argument("--no-graph", action="store_true",
                    help="time eager launches instead of replaying the captured hipGraph of one step")
    ap.add_argument("--force-cut", action="store_true",
                    help="diagnostic, one GPU: run the N>1 step (cut exchange over a one-rank nccl group, two hipGraphs) to "
                         "measure what the camera-sharded step costs per rank apart from the wire time")
    ap.add_argument("--settle", type=int, default=200,
                    help="untimed steps of the timed kind before the W warm-up / K timed steps (steady-state clocks and caches)")
    ap.add_argument("--watchdog", type=float, default=180.0,
                    help="N > 1: seconds after which device work that has not completed (a collective that never returns) "
                         "ends this rank with exit code 124 instead of hanging the job (0: off)")
    ap.add_argument("--graph-collectives", action="store_true",
                    help="N>1: capture the step INCLUDING the gradient exchange (RCCL collectives) in the hipGraph and use it "
                         "unconditionally (no probe)")
    ap.add_argument("--no-graph-collectives", action="store_true",
                    help="N>1: never capture a collective (default since round 4: ONE hipGraph incl. the RCCL collectives is probed "
                         "first -- 3 replays must reproduce the eager step's gradients on every rank, under the watchdog -- and kept "
                         "when it is the fastest; this flag restores the round-3 choice between two graphs and eager)")
    ap.add_argument("--fixed-camera", action="store_true",
                    help="N=1: replay the captured step with ONE camera / target (round-1 behaviour) instead of writing a new "
                         "camera and target into the step's slots before every replay")
    ap.add_argument("--scale-mult", type=float, default=1.0,
                    help="sensitivity: multiply the Gaussians' world-space scales (D/P grows ~quadratically)")
    ap.add_argument("--fill", type=float, default=0.85,
                    help="sensitivity: fraction of the image height the 1.8 m body fills (0.85 = SURVEY sec. 8d recipe)")
    ap.add_argument("--gaussian-order", choices=("tet", "random", "morton"), default="tet",
                    help="sensitivity: 'random' = the scene's Gaussians in random index order (default: numbered by tetrahedron); "
                         "'morton' = that random order numbered again by d3ga_amd.tetra.spatial_order")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="N>1 (or --force-cut): every rank renders k views of the pose per step and the gradients of all of them leave "
                         "in ONE exchange (the deform runs once per rank and step); value counts N x k frames per step")
    ap.add_argument("--batch-views", type=int, default=4,
                    help="N = 1: also time the VIEW-BATCHED step (k cameras of the pose in one grid per rasterizer stage: d3ga_amd/raster_views.py) "
                         "and report it as `batched_views` beside the single-view headline; 0 = off")
    ap.add_argument("--sequential-views", action="store_true",
                    help="--views-per-rank k as k sequential renders (rounds 4-5) instead of one view-batched pass")
    ap.add_argument("--fresh-scratch", action="store_true",
                    help="allocate and clear the backward's gradient accumulator per call instead of keeping a self-clearing one "
                         "(rasterizer.set_accumulator_policy)")
    ap.add_argument("--no-fused-l1", action="store_true",
                    help="render() and l1_loss() as two operators instead of d3ga_amd.renderer.render_l1 (same loss and gradients; "
                         "the gradient image then makes a round trip through HBM)")
    ap.add_argument("--no-fused-lbs", action="store_true",
                    help="lbs_cage() and cage_deform() as two operators instead of d3ga_amd.cage_deform.lbs_cage_deform (same outputs and "
                         "gradients; the LBS backward then takes a launch of its own)")
    ap.add_argument("--init-timing", action="store_true",
                    help="also time the init-time helpers at the workload's size: compute_bary (point -> tet + barycentrics, "
                         "lib/cage.py:325-327) and the 3-NN scale seed (models/cage_net.py:66), uniform grid vs exhaustive")
    ap.add_argument("--canon-grad", choices=("per-gaussian", "per-tet"), default=os.environ.get("D3GA_BENCH_CANON_GRAD", "per-gaussian"),
                    help="layout of the inverse canonical gradient handed to cage_deform: 'per-gaussian' = (P,3,3), the form the reference "
                         "stores (lib/cage.py:329) and therefore the drop-in path (default since round 5, ADVICE r4); 'per-tet' = one matrix per "
                         "tetrahedron read through tetra_id (an extension: -3 us per frame at C3)")
    ap.add_argument("--pmc", action="store_true",
                    help="collect the HBM / SQ counters of the compositing kernels with rocprofv3 (separate --pmc passes of this "
                         "same command, MI355X_MICROARCH.md) into profiles/pmc_<workload>.json, then run normally")
    return ap.parse_args()


class Frame:
    """Avatar parameters + one camera, resident on the device."""

    def __init__(self, wl_name, dev, view_index, n_views=8, scale_mult=1.0, fill=0.85, order="tet", canon_grad="per-gaussian"):
        from d3ga_amd import synthetic as syn
        from d3ga_amd.cage_deform import canonical_gradient
        self.syn = syn
        sc = syn.make_scene(wl_name)
        if order in ("random", "morton"):
            sc = syn.permute_gaussians(sc)
        if order == "morton":                   # ... and numbered again along a Z-order curve (tetra.spatial_order): the remedy
            from d3ga_amd.tetra import spatial_order
            sc = syn.reorder_gaussians(sc, spatial_order(syn.canonical_centres(sc)))
        self.wl = sc["workload"]
        d = lambda t: t.to(dev)
        self.canon, self.tetras, self.tetra_id = d(sc["canon_points"]), d(sc["tetras"]), d(sc["tetra_id"])
        self.barys0 = d(sc["barys"])
        self.joint_mats, self.skin_idx, self.skin_w = d(sc["joint_mats"]), d(sc["skin_idx"]), d(sc["skin_w"])
        self.joint_pos = sc["joint_pos"].numpy()
        # the inverse canonical gradient per GAUSSIAN, as the reference stores it (lib/cage.py:329): the drop-in layout is the one
        # that is timed.  --canon-grad per-tet: one matrix per TETRAHEDRON read through tetra_id (extension, round 4)
        self.canon_grad_mode = canon_grad
        if canon_grad == "per-tet":
            from d3ga_amd.cage_deform import canonical_gradient_per_tet
            self.canon_grad = canonical_gradient_per_tet(self.canon, self.tetras).contiguous()
        else:
            self.canon_grad = canonical_gradient(self.canon, self.tetras, self.tetra_id).contiguous()
        P = self.barys0.shape[0]
        par = lambda t: d(t).clone().requires_grad_(True)
        self.params = {
            "delta_node": par(sc["delta_node"]),
            "delta_bary": torch.zeros(P, 4, device=dev, requires_grad=True),
            "scaling": par(sc["scaling"] + math.log(scale_mult)),
            "rotation": par(sc["rotation"]),
            "opacity": par(sc["opacity_logit"]),
            # one contiguous SH buffer (features_dc | features_rest), instead of torch.cat per frame (cage_net.py:155-159)
            "features": par(torch.cat([sc["features_dc"], sc["features_rest"]], 1)),
        }
        self.n_views, self.fill, self.dev = n_views, fill, dev
        self.batch = syn.make_batch(self.wl.width, self.wl.height, azimuth=2 * math.pi * view_index / n_views,
                                    camera_id=view_index, fill=fill)
        self.bg = torch.ones(3, device=dev)
        g = torch.Generator().manual_seed(100 + view_index)
        self.target = torch.rand(3, self.wl.height, self.wl.width, generator=g).to(dev)
        self.views = None              # camera_cycle(): the n_views cameras + targets a trainer would draw from
        self.sh_degree = self.wl.sh_degree
        self.grad_sync = None          # d3ga_amd.dist.ViewShardedGrads when the views are sharded over ranks

    def camera_cycle(self):
        """The reference draws a new camera (and its ground-truth image) every step (datasets/actorshq_dataset.py:229): keep
        the n_views cameras of this pose and their targets resident, put a CameraSlot into the batch and make `target` a
        static buffer -- a captured step is then replayed with `graph.replay(camera=..., target=...)`."""
        from d3ga_amd.cameras import CameraSlot
        W, H = self.batch["width"], self.batch["height"]
        self.views = []
        for v in range(self.n_views):
            b = self.syn.make_batch(self.wl.width, self.wl.height, azimuth=2 * math.pi * v / self.n_views, camera_id=v, fill=self.fill)
            g = torch.Generator().manual_seed(100 + v)
            self.views.append((b, torch.rand(3, self.wl.height, self.wl.width, generator=g).to(self.dev)))
        self.slot = CameraSlot(W, H, device=self.dev, cells=1).set(self.batch)      # cell 0: the address of this step's target
        self.batch = dict(self.batch, camera_slot=self.slot)
        self.target = self.target.clone()              # static buffer: the training-step variants copy their target into it
        # the frame step reads its target through a TensorSlot: the n_views images are resident, a replay repoints the
        # loss kernels (8 bytes) instead of copying 25 MB into a static buffer
        from d3ga_amd.graph import TensorSlot
        self.target_slot = TensorSlot(self.views[0][1], arena=self.slot, index=0)      # travels with the camera's H2D copy
        return self.views

    def upstream(self):
        """Parameters -> what enters the rasterizer (view-independent: the same on every rank of a camera-sharded run)."""
        from d3ga_amd.cage_deform import cage_deform, lbs_cage, lbs_cage_deform
        p = self.params
        # canon_barys = barys + delta_bary, scales = exp(scaling) (cage_net.py:213-214): fused into the deform kernels
        if self.fused_lbs:      # LBS + cage deform as one autograd node: dL/d(delta_node) formed in the vertex-gather launch (round 5)
            means, cov6, _ = lbs_cage_deform(self.canon, p["delta_node"], self.joint_mats, self.skin_idx, self.skin_w, self.tetras,
                                             self.tetra_id, self.barys0, self.canon_grad, p["scaling"], p["rotation"],
                                             delta_barys=p["delta_bary"], scale_activation="exp",
                                             gradient_per_tet=self.canon_grad_mode == "per-tet")
        else:
            tetpoints = lbs_cage(self.canon, p["delta_node"], self.joint_mats, self.skin_idx, self.skin_w)
            means, cov6 = cage_deform(tetpoints, self.tetras, self.tetra_id, self.barys0, self.canon_grad, p["scaling"],
                                      p["rotation"], delta_barys=p["delta_bary"], scale_activation="exp",
                                      gradient_per_tet=self.canon_grad_mode == "per-tet")
        # opacity = sigmoid(opacities) (cage_net.py:247): fused into the per-Gaussian kernels (pkg["opacity_logits"])
        return {"means3D": means, "cov3D_precomp": cov6, "opacity_logits": p["opacity"],
                "shs": p["features"], "rgb": None, "sh_degree": self.sh_degree}

    fused_lbs = True                # --no-fused-lbs: lbs_cage() and cage_deform() as two operators (one more launch in the backward)
    fused_l1 = True                 # --no-fused-l1: render() + l1_loss() as two operators (a (3,H,W) gradient image in between)

    def loss_from(self, pkg):
        from d3ga_amd.losses import l1_loss
        from d3ga_amd.renderer import render, render_l1
        # mean |img - target| (utils/loss_utils.py:29); with camera_cycle() the target is whatever the slot names
        if getattr(self, "my_views", None) and getattr(self, "batch_my_views", True):
            # --views-per-rank k: this rank's k cameras of the pose in ONE view-batched pass (round 6; --sequential-views: k renders)
            from d3ga_amd.renderer import render_views
            if getattr(self, "_my_cams", None) is None:
                from d3ga_amd.raster_views import CameraBatch
                b0 = self.my_views[0][0]
                self._my_cams = CameraBatch(len(self.my_views), int(b0["width"]), int(b0["height"]), device=self.dev).set([b for b, _ in self.my_views])
                self._my_targets = torch.stack([t for _, t in self.my_views]).contiguous()
            return render_views(None, pkg, self.bg, targets=self._my_targets, cameras=self._my_cams, grad_sync=self.grad_sync)["l1"]
        if getattr(self, "my_views", None):     # k cameras of the pose, one package, the mean of their losses, k sequential renders
            tot = None
            for b, t in self.my_views:
                l = render_l1(b, pkg, self.bg, t, grad_sync=self.grad_sync)["l1"]
                tot = l if tot is None else tot + l
            return tot / len(self.my_views)
        target = getattr(self, "target_slot", None) or self.target
        if self.fused_l1:           # the same loss and gradients from ONE operator: dL/dimage is formed inside the compositing backward
            return render_l1(self.batch, pkg, self.bg, target, grad_sync=self.grad_sync)["l1"]
        img = render(self.batch, pkg, self.bg, grad_sync=self.grad_sync)["render"]
        return l1_loss(img, target)

    def step(self):
        loss = self.loss_from(self.upstream())
        # dL/dloss = 1 from a persistent device scalar: `loss.backward()` alone makes autograd fill a fresh ones_like(loss) --
        # one more kernel node per step (4.6 us of the captured step at C3, profiles/r04_bench_C3_kernel_stats.csv)
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device:
            one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(one)
        return loss


    def train_step(self, with_fields=False, pair=False, scale_weight=0.0):
        """The reference's training step renders twice (models/trainer.py:102-110): RGB, then a silhouette pass with a
        constant per-Gaussian colour on a black background; losses as in train.py:190-193 (L1 + SSIM on RGB, L1 on the
        silhouette).  with_fields: the cage-vertex offsets and the per-Gaussian (delta_bary, delta_rot, delta_scale) come
        from the DeformationField / CanonicalField networks as in models/cage_net.py:197-215 instead of free parameters.
        scale_weight: weight of the reference's scale regulariser mean(scales^2) (cage_net.py:226; train.py:203 uses 175) on
        the EFFECTIVE scales exp(scaling + delta_scale) -- without it the synthetic objective has a degenerate optimum,
        screen-filling Gaussians (tools/soak.py)."""
        from d3ga_amd.cage_deform import cage_deform, lbs_cage
        from d3ga_amd.losses import l1_loss, l1_ssim
        from d3ga_amd.renderer import render, render_pair
        p = self.params
        if not hasattr(self, "sil_rgb"):
            P = self.barys0.shape[0]
            self.sil_rgb = torch.ones(P, 3, device=self.bg.device)
            self.sil_target = (self.target.mean(0, keepdim=True) > 0.5).float().expand(3, -1, -1).contiguous()
            self.bg0 = torch.zeros_like(self.bg)
        if with_fields:         # True: geometry networks; "color": geometry + colour networks
            if not hasattr(self, "canon_field"):
                from d3ga_amd.mlp import CanonicalField, DeformationField
                torch.manual_seed(17)
                dev = self.bg.device
                self.canon_field, self.deform_field = CanonicalField().to(dev), DeformationField(scaling=0.07).to(dev)
                self.pose = 0.3 * torch.randn(98, device=dev)
                self.deform_field.set_constant_input(self.canon)      # the canonical cage vertices are a buffer of the model (cage_net.py:197)
                self.field_params = list(self.canon_field.parameters()) + list(self.deform_field.parameters())
            for q in self.field_params:
                q.grad = None
            delta_node = self.deform_field(self.canon, self.pose)                                  # cage_net.py:197
            d_bary, d_rot, d_scale = self.canon_field(p["rotation"], p["scaling"], self.barys0, self.pose)   # :199-204
            tetpoints = lbs_cage(self.canon, delta_node, self.joint_mats, self.skin_idx, self.skin_w)
            log_scales = p["scaling"] + d_scale
            means, cov6 = cage_deform(tetpoints, self.tetras, self.tetra_id, self.barys0, self.canon_grad,
                                      log_scales, p["rotation"] + d_rot, delta_barys=d_bary,
                                      scale_activation="exp", gradient_per_tet=self.canon_grad_mode == "per-tet")                                      # :213-230
        else:
            tetpoints = lbs_cage(self.canon, p["delta_node"], self.joint_mats, self.skin_idx, self.skin_w)
            log_scales = p["scaling"]
            means, cov6 = cage_deform(tetpoints, self.tetras, self.tetra_id, self.barys0, self.canon_grad, p["scaling"],
                                      p["rotation"], delta_barys=p["delta_bary"], scale_activation="exp",
                                      gradient_per_tet=self.canon_grad_mode == "per-tet")
        pkg = {"means3D": means, "cov3D_precomp": cov6, "opacity_logits": p["opacity"],
               "shs": p["features"], "rgb": None, "sh_degree": self.sh_degree}
        if with_fields == "color":
            from d3ga_amd.mlp import view_directions
            # configs/actorshq_actor02.yml (use_shs false): colour AND opacity from ColorField on per-Gaussian features,
            # the pose, the view direction and the frame embedding (cage_net.py:232-258)
            if not hasattr(self, "color_field"):
                from d3ga_amd.mlp import ColorField
                dev = self.bg.device
                self.color_field = ColorField().to(dev)
                self.field_params += list(self.color_field.parameters())
                self.color_feat = (0.33 * torch.rand(self.barys0.shape[0], 64, device=dev)).requires_grad_(True)   # :59
                self.frame_enc = (0.1 * torch.randn(32, device=dev)).requires_grad_(True)
                from d3ga_amd.cameras import batch_to_camera
                self.cam_center = batch_to_camera(self.batch, device=dev).camera_center.reshape(1, 3)
            self.color_feat.grad = None
            self.frame_enc.grad = None
            viewdirs = view_directions(means, self.cam_center)                                     # :233-235
            rgb, opac = self.color_field(self.color_feat, self.pose, viewdirs, frame_encoding=self.frame_enc)
            pkg = {"means3D": means, "cov3D_precomp": cov6, "opacities": opac, "shs": None, "rgb": rgb,
                   "sh_degree": self.sh_degree}
        if pair:       # both images from ONE compositing pass (d3ga_amd.renderer.render_pair; same images, same gradients)
            both = render_pair(self.batch, pkg, self.bg, self.sil_rgb, self.bg0, grad_sync=self.grad_sync)
            img, sil = both["render"], both["render2"]
        else:
            from d3ga_amd.rasterizer import geometry_reuse
            with geometry_reuse():       # the silhouette pass reuses the RGB pass's projection / binning / sort (opt-in)
                img = render(self.batch, pkg, self.bg, grad_sync=self.grad_sync)["render"]
                sil = render(self.batch, pkg, self.bg0, colors_precomp=self.sil_rgb, grad_sync=self.grad_sync)["render"]
        # train.py:190-193: (1 - lambda) L1 + lambda (1 - SSIM) on the RGB image, L1 on the silhouette
        lam = 0.2
        rgb_l1, rgb_ssim = l1_ssim(img, self.target)          # one fused kernel each way
        loss = (1.0 - lam) * rgb_l1 + lam * (1.0 - rgb_ssim) + l1_loss(sil, self.sil_target)
        if scale_weight:
            loss = loss + scale_weight * torch.exp(2.0 * log_scales).mean()          # cage_net.py:226, train.py:203
        loss.backward()
        return loss


class _CutShim:
    """What CapturedCutStep's eager flow needs of `self` (used unbound for the k-views-per-rank eager step)."""
    _ALIASES = {"opacity_logits": "opacities", "rgb": "colors_precomp"}

    def __init__(self, sync):
        self.sync = sync

    def _to_the_cut(self, upstream, loss_fn):
        from d3ga_amd.graph import CapturedCutStep
        return CapturedCutStep._to_the_cut(self, upstream, loss_fn)

    def _from_the_cut(self, up, pkg):
        from d3ga_amd.graph import CapturedCutStep
        return CapturedCutStep._from_the_cut(self, up, pkg)


def collect_pmc(args):
    """bench.py --pmc: the HBM / SQ counters of every d3ga kernel of THIS workload, collected with rocprofv3 in separate
    --pmc passes of this same command (FETCH_SIZE and WRITE_SIZE cannot share a pass; counters perturb timing, so the
    passes only count) and written to profiles/pmc_<workload>.json, where the normal run below picks them up."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "grbm": ["GRBM_GUI_ACTIVE"],
              "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
                     "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
              "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY",
                      "SQ_ACTIVE_INST_ANY"]}
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for name, counters in passes.items():
        d = tempfile.mkdtemp(prefix=f"d3ga_pmc_{name}_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--kernel-include-regex", "d3ga", "--output-format", "csv",
               "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "5",
               "--warmup", "3", "--no-cpu-baseline", "--no-train-step", "--no-stage-events", "--no-graph", "--fixed-camera", "--batch-views", "0",
               "--scale-mult", str(args.scale_mult), "--fill", str(args.fill)]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        files = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            print(f"[bench --pmc] pass {name} failed: rc {r.returncode} {r.stderr[-300:]}", file=sys.stderr)
            shutil.rmtree(d, ignore_errors=True)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: round(sum(x) / len(x), 1) for c, x in v.items()})
        shutil.rmtree(d, ignore_errors=True)
    if out:
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        suffix = "" if (args.scale_mult == 1.0 and args.fill == 0.85) else f"_s{args.scale_mult:g}_f{args.fill:g}"
        out["_meta"] = {"lib_sha256": lib_sha256(), "command": "bench.py --pmc (separate rocprofv3 --pmc passes of the same command)",
                        "workload": args.workload}
        json.dump(out, open(os.path.join(ROOT, "profiles", f"pmc_{args.workload}{suffix}.json"), "w"), indent=1)
    return out


def lib_sha256():
    """sha256 of the libd3ga_hip.so this process runs -- PMC summaries are stamped with it, so that a bench line can say whether
    the counters it quotes were collected on the library it timed (`roofline.traffic_stale`)."""
    import hashlib
    from d3ga_amd import _lib
    try:
        return hashlib.sha256(open(_lib._PATH, "rb").read()).hexdigest()
    except Exception:
        return None


def pmc_summary(workload, suffix=""):
    """(dict kernel -> counters, source path, stale?) of the newest PMC summary for this workload: profiles/pmc_<wl>.json (written
    by `bench.py --pmc` in this checkout) or the newest committed profiles/rNN_pmc_<wl>.json."""
    import glob
    cands = [os.path.join(ROOT, "profiles", f"pmc_{workload}{suffix}.json")]
    cands += sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{workload}{suffix}.json")), reverse=True)
    for pj in cands:
        if os.path.exists(pj):
            d = json.load(open(pj))
            meta = d.pop("_meta", {})
            stale = meta.get("lib_sha256") != lib_sha256()
            return d, os.path.relpath(pj, ROOT), stale
    return {}, None, None


def valu_issue_model():
    """Calibrated VALU issue costs (tools/micro/valu_issue.hip -> profiles/r02_valu_issue_pmc.json, cycles per wave-instruction
    per SIMD at 8 waves/SIMD) and the static instruction mix of the compositing backward's loop (tools/isa_mix.py ->
    profiles/r02_composite_bwd_mix.json).  Returns (average cycles per VALU instruction of that kernel, dict) or (None, None)."""
    try:
        cal = {(r["kind"], r["waves_per_simd"]): r for r in json.load(open(os.path.join(ROOT, "profiles", "r02_valu_issue_pmc.json")))}
        cyc = lambda kind: cal[(kind, 8)]["cycles_per_wave_inst_per_simd"]
        cost = {"plain": cyc("v_fma_f32"), "dpp": cyc("v_add_f32_dpp"), "trans": cyc("v_exp_f32"), "packed": cyc("v_pk_fma_f32")}
        mixes = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_composite_bwd_mix.json"))
        mix = json.load(open(os.path.join(ROOT, "profiles", mixes[-1])))
        cost["vop3_other"] = mix.get("vop3_other_cycles", 1.6 * cost["plain"])
        n = sum(mix["counts"].values())
        avg = sum(mix["counts"][k] * cost[k] for k in mix["counts"]) / n
        return avg, {"cycles_per_class": {k: round(v, 2) for k, v in cost.items()}, "mix": mix["counts"]}
    except Exception:
        return None, None


def lane_efficiency(workload):
    """Lane efficiency of the two compositing kernels -- useful (entry, pixel) pairs / issued lane slots -- from the newest
    profiles/r*_composite_diag_<workload>.json (tools/diag_scan.py / tools/diag_fwd.py on the counter build of the library:
    the product build carries no counters).  None when no such file exists."""
    try:
        pd = os.path.join(ROOT, "profiles")
        f = sorted(x for x in os.listdir(pd) if x.endswith(f"_composite_diag_{workload}.json"))
        if not f:
            return None
        d = json.load(open(os.path.join(pd, f[-1])))
        return {"backward": d["backward"].get("lane_efficiency"), "forward": d["forward"].get("lane_efficiency"),
                "definition": "useful (entry, pixel) pairs / lane slots issued by the pixel-step (backward: wave groups x 64 lanes x 16 steps) "
                              "and blend (forward: iterations x 2 entries x 64 lanes) loops", "source": "profiles/" + f[-1],
                "valid_pairs": d["backward"].get("valid_entry_pixel_pairs")}
    except Exception:
        return None


def init_timing(frame):
    """SURVEY sec. 8f-3 at the workload's size: every Gaussian located in the canonical cage (compute_bary) and the 3-nearest-
    neighbour scale seed over the Gaussian centres, uniform-grid search against the exhaustive kernels (same outputs)."""
    from d3ga_amd.tetra import compute_bary, knn_mean_dist2
    corners = frame.canon[frame.tetras.long()].contiguous()                                 # (T,4,3)
    pts = (corners[frame.tetra_id.long()] * frame.barys0[:, :, None]).sum(1).contiguous()   # the Gaussians' canonical centres
    P, T = pts.shape[0], corners.shape[0]

    def timed(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return time.perf_counter() - t0, r
    out = {"points": P, "tets": T}
    tg, rg = timed(lambda: compute_bary(pts, corners, method="grid"))
    out["compute_bary_grid_s"] = round(tg, 4)
    n_ex = P if P * T <= 2e11 else max(1, int(2e11 // T))              # bound the exhaustive leg to ~2e11 point-in-tet tests
    te, re_ = timed(lambda: compute_bary(pts[:n_ex].contiguous(), corners, method="exhaustive"))
    out["compute_bary_exhaustive_s"] = round(te * P / n_ex, 4)
    out["compute_bary_exhaustive_sample"] = n_ex
    out["compute_bary_same_result"] = bool(torch.equal(rg[1][:n_ex], re_[1]) and torch.equal(rg[0][:n_ex], re_[0]))
    out["recovered_tet_fraction"] = round(float((rg[1] == frame.tetra_id.long()).float().mean()), 4)
    tk, rk = timed(lambda: knn_mean_dist2(pts, method="grid"))
    out["knn3_grid_s"] = round(tk, 4)
    n_ex = P if P <= 300_000 else 300_000
    if n_ex == P:
        tke, rke = timed(lambda: knn_mean_dist2(pts, method="exhaustive"))
        out["knn3_exhaustive_s"] = round(tke, 4)
        out["knn3_same_result"] = bool(torch.equal(rk, rke))
    else:
        tke, _ = timed(lambda: knn_mean_dist2(pts[:n_ex].contiguous(), method="exhaustive"))
        out["knn3_exhaustive_s"] = round(tke * (P / n_ex) ** 2, 4)
        out["knn3_exhaustive_note"] = f"extrapolated quadratically from {n_ex} points"
    return out


def measured_copy_gbs(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes / time): the practical HBM ceiling quoted beside the 8 TB/s peak."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def deform_gpu_comparison(frame, reps=50):
    """Like-for-like for the fused deform kernels: the reference's unfused tensor program
    (oracle/deform.py, the restatement of cage_net.py:213-230) run by ATen on the SAME GPU, against
    d3ga_lbs_cage + d3ga_cage_deform, forward + backward each.  Baseline leg only."""
    from d3ga_amd.cage_deform import cage_deform, lbs_cage
    from oracle import deform as od
    p = frame.params
    tetras, tetra_id = frame.tetras.long(), frame.tetra_id.long()
    # (the reference's form of the canonical gradient: gathered per Gaussian at init, lib/cage.py:329)
    canon_grad_pg = frame.canon_grad if frame.canon_grad.shape[0] == frame.barys0.shape[0] else frame.canon_grad[tetra_id].contiguous()

    def fused():
        tp = lbs_cage(frame.canon, p["delta_node"], frame.joint_mats, frame.skin_idx, frame.skin_w)
        m, c = cage_deform(tp, frame.tetras, frame.tetra_id, frame.barys0, frame.canon_grad, p["scaling"], p["rotation"],
                           delta_barys=p["delta_bary"], scale_activation="exp", gradient_per_tet=frame.canon_grad_mode == "per-tet")
        (m.sum() + c.sum()).backward()

    def unfused():
        tp = od.lbs_cage(frame.canon, p["delta_node"], frame.joint_mats, frame.skin_idx.long(), frame.skin_w)
        m, c = od.cage_deform(tp, tetras, tetra_id, frame.barys0 + p["delta_bary"], canon_grad_pg,
                              torch.exp(p["scaling"]), p["rotation"])
        (m.sum() + c.sum()).backward()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 4)

    res = {}
    for name, fn in (("fused_hip_ms", fused), ("unfused_torch_gpu_ms", unfused)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        res[name] = timed(fn)
    res["fused_hip_note"] = "eager: a host-bound loop (its ~63 us of kernels sit inside ~0.2 ms of Python launches and ATen sums)"
    # the same two legs as captured hipGraphs: what the GPU needs, without the host
    for name, fn in (("fused_hip_graph_ms", fused), ("unfused_torch_gpu_graph_ms", unfused)):
        try:
            for q in p.values():
                q.grad = None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for q in p.values():
                q.grad = None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay()
            torch.cuda.synchronize()
            res[name] = timed(g.replay)
            del g
        except Exception as e:  # noqa: BLE001
            res[name] = f"capture failed: {type(e).__name__}"
    for q in p.values():
        q.grad = None
    return res


def usable_cpus():
    """Hardware threads this process may actually use: the affinity mask, capped by a cgroup CPU quota if there is one
    (os.cpu_count() reports the whole host inside a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(wl_name, budget_s=20.0, threads=None):
    """The oracle (CPU port of the path) on this box's host cores: torch deform fwd+bwd + C rasterizer fwd+bwd."""
    from d3ga_amd import synthetic as syn
    from oracle import camera as oc
    from oracle import deform as od
    from oracle import raster_c as rc
    # threads actually used.  All hardware threads of the box are tried as well (budget split): on a 256-thread host torch's
    # intra-op pool and libgomp oversubscribe badly, so `value` / `cores` report the FASTER of the two settings and
    # `host_threads` / `all_threads_frames_per_s` state the other
    host_threads = usable_cpus()
    cores = min(host_threads, 32) if threads is None else threads
    torch.set_num_threads(cores)
    rc.set_threads(cores)
    sc = syn.make_scene(wl_name)
    wl = sc["workload"]
    b = syn.make_batch(wl.width, wl.height)
    cam = oc.camera(b["R"], b["T"], b["FoVx"], b["FoVy"])
    cg = od.canonical_gradient(sc["canon_points"], sc["tetras"].long(), sc["tetra_id"].long())
    shs = torch.cat([sc["features_dc"], sc["features_rest"]], 1).numpy()
    op = torch.sigmoid(sc["opacity_logit"]).numpy()
    gpix = np.random.default_rng(0).normal(size=(3, b["height"], b["width"])).astype(np.float32)
    bg = np.ones(3, np.float32)
    times, deform_times = [], []
    t_start = time.time()
    while True:
        t0 = time.time()
        delta = sc["delta_node"].clone().requires_grad_(True)
        scaling = sc["scaling"].clone().requires_grad_(True)
        rot = sc["rotation"].clone().requires_grad_(True)
        barys = sc["barys"].clone().requires_grad_(True)
        tp = od.lbs_cage(sc["canon_points"], delta, sc["joint_mats"], sc["skin_idx"], sc["skin_w"])
        means, cov6 = od.cage_deform(tp, sc["tetras"], sc["tetra_id"], barys, cg, torch.exp(scaling), rot)
        t1 = time.time()
        color, radii, _, ctx = rc.forward(means.detach().numpy(), op, bg, cam["world_view_transform"],
                                          cam["full_proj_transform"], cam["camera_center"], cam["tanfovx"],
                                          cam["tanfovy"], b["width"], b["height"], cov3D_precomp=cov6.detach().numpy(),
                                          shs=shs, sh_degree=wl.sh_degree)
        g = rc.backward(ctx, gpix)
        t2 = time.time()
        ((means * torch.from_numpy(g["means3D"])).sum() + (cov6 * torch.from_numpy(g["cov3D"])).sum()).backward()
        t3 = time.time()
        times.append(t3 - t0)
        deform_times.append((t1 - t0) + (t3 - t2))
        el = time.time() - t_start          # a bounded sample: ~10 s of CPU work (at least 5 frames, at most budget_s)
        if el > budget_s or (el > 10.0 and len(times) >= 5) or len(times) >= 60:
            break
    best = min(times)
    if threads is None and host_threads > cores:
        try:
            alt = cpu_baseline(wl_name, budget_s=budget_s / 2, threads=host_threads)
        except Exception:  # noqa: BLE001
            alt = None
        mine = {"value": round(1.0 / best, 4), "cores": cores, "frames": len(times), "deform": round(1.0 / min(deform_times), 3)}
        other = None if alt is None else {"value": alt["value"], "cores": host_threads, "frames": None,
                                          "deform": alt["deform_only_frames_per_s"]}
        win = mine if other is None or mine["value"] >= other["value"] else other
        return {"value": win["value"], "unit": "frames/s", "cores": win["cores"], "kind": "port", "host_threads": host_threads,
                "frames_per_s_by_threads": {str(cores): mine["value"], **({str(host_threads): other["value"]} if other else {})},
                "sample": f"{len(times)} full frames of {wl.name} at {cores} threads (+ a second sample at all {host_threads} hardware "
                          "threads) -- oracle: torch CPU deform fwd+bwd + OpenMP C rasterizer fwd+bwd, best frame; the faster setting is `value`",
                "deform_only_frames_per_s": win["deform"]}
    return {"value": round(1.0 / best, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full frames of {wl.name} (oracle: torch CPU deform fwd+bwd + OpenMP C rasterizer fwd+bwd), best of {len(times)}",
            "deform_only_frames_per_s": round(1.0 / min(deform_times), 3)}


def field_mlp_bench(args):
    """CanonicalField (models/mlp.py:74-110, widths of configs/actorshq_actor02.yml) forward + backward to inputs and weights
    over the workload's Gaussians: rows/s, achieved HBM rate (and the f32-equivalent FLOP rate), the same module under ATen on this GPU and the oracle on
    the host cores."""
    from d3ga_amd import synthetic as syn
    from d3ga_amd.mlp import CanonicalField
    from oracle import mlp as om
    dev = torch.device("cuda", 0)
    P = syn.WORKLOADS[args.workload].n_gaussians
    torch.manual_seed(17)
    cf = CanonicalField().to(dev)
    g = torch.Generator().manual_seed(17)
    barys = torch.rand(P, 4, generator=g).to(dev).requires_grad_(True)
    rots = torch.randn(P, 4, generator=g).to(dev).requires_grad_(True)
    scales = (0.1 * torch.randn(P, 3, generator=g)).to(dev).requires_grad_(True)
    pose = (0.3 * torch.randn(98, generator=g)).to(dev)
    leaves = list(cf.parameters()) + [barys, rots, scales]
    hidden = [(l.weight, l.bias) for l in cf.network]

    def step(fn):
        for t in leaves:
            t.grad = None
        o = fn()
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()

    fused = lambda: step(lambda: cf(rots, scales, barys, pose))
    aten = lambda: step(lambda: om.canonical_field(rots, scales, barys, pose, hidden, cf.output.weight, cf.output.bias))

    def timed(fn, n):
        for _ in range(max(args.warmup, 3)):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = timed(fused, args.steps)
    ms_aten = timed(aten, max(5, args.steps // 4))
    flops = 3 * 2.0 * P * (11 * 128 + 3 * 128 * 128 + 128 * 11)         # forward + input-gradient + weight-gradient GEMMs
    # algorithmic bytes per row (f32 = 4 B): every GEMM reads its row operand(s) and writes its result once; sign bits 16 B
    widths = [11, 128, 128, 128, 128, 11]                                # 109 -> 128 x4 -> 11 with the 98 pose columns folded
    from d3ga_amd import mlp as _mlp
    fused_fwd = bool(_mlp._FUSED["enabled"])
    # fused forward (one launch per trunk): the input is read once, every layer's output is written once, nothing is read back
    fwd_b = (4 * widths[0] + sum(4 * b for b in widths[1:]) + 16 * 4) if fused_fwd \
        else sum(4 * (a + b) for a, b in zip(widths[:-1], widths[1:])) + 16 * 4
    dx_b = sum(4 * (a + b) for a, b in zip(widths[:-1], widths[1:])) + 16 * 4
    wg_b = sum(4 * (a + b) for a, b in zip(widths[:-1], widths[1:]))
    hbm_bytes = float(P) * (fwd_b + dx_b + wg_b)
    out = {"metric": f"CanonicalField fwd+bwd rows/sec ({P} Gaussians, 109->128x4->11, f32)", "value": round(P / (ms * 1e-3), 1),
           "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"CanonicalField over the Gaussians of {args.workload}", "rows": P},
           # the split-bf16 products (6 per f32 product, 16x the f32 MFMA rate) leave the layers HBM-bound
           "roofline": {"kernel": ("chain_fwd_kernel + " if fused_fwd else "") + "linear_kernel + wgrad_kernel", "fused_forward": fused_fwd, "bound": "hbm", "achieved": round(hbm_bytes / (ms * 1e-3) / 1e9, 1),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(hbm_bytes / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
                        "f32_equivalent_tflops": round(flops / (ms * 1e-3) / 1e12, 2), "f32_mfma_peak_tflops": 157.3,
                        "arithmetic": "exact 3-way bf16 split of every f32 operand, 6 products on v_mfma_f32_32x32x16_bf16, "
                                      "f32 accumulate (error of an f32 fmaf chain)"},
           "same_gpu_aten_ms": round(ms_aten, 4)}
    if not args.no_cpu_baseline:
        n = min(P, 50_000)
        cpu = lambda t: t.detach()[:n].cpu().requires_grad_(True)
        cb, cr, cs, cp = cpu(barys), cpu(rots), cpu(scales), pose.cpu()
        ch = [(w.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)) for w, b in hidden]
        cow, cob = cf.output.weight.detach().cpu().requires_grad_(True), cf.output.bias.detach().cpu().requires_grad_(True)
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        best = 1e9
        for _ in range(5):
            t0 = time.time()
            o = om.canonical_field(cr, cs, cb, cp, ch, cow, cob)
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()
            best = min(best, time.time() - t0)
        out["cpu_baseline"] = {"value": round(n / best, 1), "unit": "rows/s", "cores": cores, "kind": "port",
                               "sample": f"{n} rows, oracle/mlp.py (torch CPU), best of 5"}
    print(json.dumps(out))


def field_mlp_summary(P, dev, steps=20, warmup=5):
    """The `field_mlp` object of the default bench line: CanonicalField (models/mlp.py:74-110: 109 -> 128 x 4 -> 11 with the 98 pose
    columns folded into the first bias) forward + backward to inputs AND weights over the workload's P Gaussians -- the unit that is
    85 % of the reference-faithful training step.  Here the matrix cores are the right roofline: every f32
    product is six bf16 MFMA products (exact 3-way split), so the MFMA work is 6 x the f32-equivalent FLOPs; peak = the dense bf16
    rate of MI355X_MICROARCH.md (2.5 PFLOP/s).  The HBM side of the same launches is reported beside it (the weight gradients and
    the per-layer outputs are HBM-bound)."""
    from d3ga_amd.mlp import CanonicalField
    torch.manual_seed(17)
    cf = CanonicalField().to(dev)
    g = torch.Generator().manual_seed(17)
    barys = torch.rand(P, 4, generator=g).to(dev).requires_grad_(True)
    rots = torch.randn(P, 4, generator=g).to(dev).requires_grad_(True)
    scales = (0.1 * torch.randn(P, 3, generator=g)).to(dev).requires_grad_(True)
    pose = (0.3 * torch.randn(98, generator=g)).to(dev)
    leaves = list(cf.parameters()) + [barys, rots, scales]

    def step():
        for t in leaves:
            t.grad = None
        o = cf(rots, scales, barys, pose)
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    flops = 3 * 2.0 * P * (11 * 128 + 3 * 128 * 128 + 128 * 11)         # forward + input-gradient + weight-gradient GEMMs, f32-equivalent
    widths = [11, 128, 128, 128, 128, 11]
    fwd_b = 4 * widths[0] + sum(4 * b for b in widths[1:]) + 16 * 4      # fused forward: input once, every layer's output once, sign bits
    dx_b = sum(4 * (a + b) for a, b in zip(widths[:-1], widths[1:])) + 16 * 4
    wg_b = sum(4 * (a + b) for a, b in zip(widths[:-1], widths[1:]))
    hbm = float(P) * (fwd_b + dx_b + wg_b)
    mfma_tf = 6.0 * flops / (ms * 1e-3) / 1e12
    return {"unit_of_work": f"CanonicalField forward + backward (inputs and weights), {P} rows, 109 -> 128 x 4 -> 11, f32-equivalent",
            "ms": round(ms, 4), "rows_per_s": round(P / (ms * 1e-3), 1), "f32_equivalent_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
            "roofline": {"bound": "mfma", "achieved": round(mfma_tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(mfma_tf / 2500.0, 4),
                         "traffic": None, "note": "6 bf16 MFMA products per f32 product (exact 3-way split, v_mfma_f32_32x32x16_bf16); "
                                                  "peak = dense bf16, MI355X_MICROARCH.md"},
            "hbm": {"alg_bytes": int(hbm), "achieved_GBs": round(hbm / (ms * 1e-3) / 1e9, 1), "frac_hbm_peak": round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def color_train_bench(args):
    """`--train-step color`: BASELINE configs[3] -- the actor02-shaped training step on N ranks (camera sharding, parameter-level
    gradient exchange: the rasterizer's inputs are view-dependent there, so the cut exchange of the SH frame is invalid)."""
    from d3ga_amd import dist as ddist
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = ddist.init_process_group(args.backend)
    if world != max(args.gpus, 1):
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line under the wrong N", file=sys.stderr)
        sys.exit(3)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if world > 1 and not args.single_device and torch.cuda.device_count() < world:
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import d3ga_amd
    d3ga_amd.lib()
    from d3ga_amd import rasterizer as R
    wl_name = args.workload if args.workload != "C3" or "--workload" in sys.argv else "C4"
    frame = Frame(wl_name, dev, view_index=rank % 8, canon_grad=args.canon_grad)
    R.set_accumulator_policy("persistent")
    kw = dict(with_fields="color", pair=True, scale_weight=175.0)
    frame.train_step(**kw)                                 # creates the three networks (seeded: identical on every rank), sizes the scratch
    cnt = R.last_counters()
    d_max = torch.tensor([float(cnt["D"])], device=dev)
    if world > 1:
        torch.distributed.all_reduce(d_max, op=torch.distributed.ReduceOp.MAX)
    R.set_capacity_policy("static", int(2.0 * float(d_max.item())) + 4096)      # (the avatar is OPTIMISED here: its splats grow)
    p = frame.params
    nets = {"color_field": list(frame.color_field.parameters()), "canonical_field": list(frame.canon_field.parameters()),
            "deformation_field": list(frame.deform_field.parameters())}
    # buckets in the order their gradients complete in the backward (the colour network sits last in the forward)
    buckets = [nets["color_field"], [frame.color_feat], [frame.frame_enc], nets["canonical_field"], [p["rotation"]], [p["scaling"]],
               nets["deformation_field"]]
    leaves = [q for b in buckets for q in b]
    red = ddist.BucketedGradReducer(buckets, timing=True)
    make_opt = lambda: torch.optim.Adam(leaves, lr=1e-4, capturable=True)       # (capturable: the step may live in a hipGraph)
    opt = make_opt()
    losses = []

    def one_step():
        red.begin_step()
        loss = frame.train_step(**kw)
        red.finish()
        torch.nn.utils.clip_grad_norm_(leaves, 1.0)       # AFTER the reduce (models/trainer.py:188)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    can_capture = not args.no_graph and not args.single_device and (world == 1 or (torch.distributed.get_backend() == "nccl" and not args.no_graph_collectives))

    def try_capture(tag):
        """ONE hipGraph of one_step, or None; N > 1: only when every rank captured."""
        cap = None
        if can_capture:
            from d3ga_amd.graph import CapturedStep
            try:
                cap = CapturedStep(one_step, params=leaves)
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: capture of the {tag} failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
            if world > 1:
                ok = torch.tensor([1.0 if cap is not None else 0.0], device=dev)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                if float(ok) == 0.0:
                    cap = None
        return cap

    # N = 1 reference on every rank, in the SAME launch mode as the N-rank step (identical replicas afterwards: the parameters
    # are restored and the optimiser is new)
    state = [q.detach().clone() for q in leaves]
    n1_ms = None
    if world > 1:
        active = red._active
        red._active = lambda: False                        # the N = 1 step of the same frame: hooks count, nothing goes on the wire
        for _ in range(3):
            one_step()
        cap1 = try_capture("N = 1 reference step")
        run1 = (lambda: cap1.replay()) if cap1 is not None else one_step
        for _ in range(3):
            run1()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(5, args.steps // 5)):
            run1()
        torch.cuda.synchronize()
        n1 = torch.tensor([1e3 * (time.perf_counter() - t0) / max(5, args.steps // 5)], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(n1, op=torch.distributed.ReduceOp.MAX)
        n1_ms = float(n1.item())
        del cap1
        red._active = active
        with torch.no_grad():
            for q, q0 in zip(leaves, state):
                q.copy_(q0)
        opt = make_opt()
    for _ in range(args.warmup):
        one_step()
    # The eager step is bound by the host at this size (~150 launches: 3.6-4.6 ms against ~2 ms of GPU work at C4): capture the WHOLE
    # training step -- forward, backward with the per-bucket collectives issued from its hooks, clip, Adam -- in ONE hipGraph.
    # N > 1: over RCCL only (as for the frame's step: --no-graph-collectives keeps it eager); every rank must have captured.
    cap, launch_mode = try_capture("training step"), "eager"
    if cap is not None:
        launch_mode = "ONE hipGraph of the whole training step (forward, backward with the bucket collectives, clip_grad_norm_, Adam)"

    def timed_step():
        if cap is not None:
            return cap.replay()
        return one_step()
    for _ in range(3):
        timed_step()
    red.exchange_ms()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(timed_step().detach().clone())
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    ones = torch.ones(1, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(ones)
    ex_ms = red.exchange_ms()
    # replicas must stay bit-identical: every rank applied the same averaged gradients to the same parameters
    chk = torch.stack([q.detach().double().sum() for q in leaves] + [q.detach().double().abs().sum() for q in leaves])
    same = True
    if world > 1:
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    ms = 1e3 * float(tmax.item()) / args.steps
    ls = torch.stack(losses).float().cpu()
    if rank == 0:
        wl = frame.wl
        out = {"metric": "training steps (views)/sec, actor02-shaped step: 2 renders + Deformation/Canonical/ColorField + L1/SSIM/silhouette + Adam",
               "value": round(world * 1e3 / ms, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "lib_sha256": lib_sha256(),
               "config": {"workload": f"{wl_name}: {wl.n_gaussians} Gaussians, {frame.batch['width']}x{frame.batch['height']}, use_shs false "
                                      f"(configs/actorshq_actor02.yml), one view per rank", "views_per_step": world,
                          "parallelism": f"camera-sharded dp{world}", "grad_exchange": "per-bucket asynchronous all-reduce of ALL parameter "
                          "gradients from post-accumulate hooks (dist.BucketedGradReducer): the rasterizer's inputs are view-dependent",
                          "grad_exchange_bytes_per_rank": red.nbytes(), "buckets": len(red.buckets)},
               "roofline": None, "cpu_baseline": None,
               "loss_first_last": [round(float(ls[:5].mean()), 5), round(float(ls[-5:].mean()), 5)], "loss_finite": bool(torch.isfinite(ls).all()),
               "replicas_identical": same, "launch_mode": launch_mode,
               "distributed": {"nranks_seen": int(ones.item()), "n1_ms_per_step": None if n1_ms is None else round(n1_ms, 4),
                               "efficiency": None if n1_ms is None else round(n1_ms / ms, 4),
                               "bucket_ms": None if ex_ms is None else round(ex_ms, 4),
                               "note": "bucket_ms: mean HIP-event time from a bucket's launch to its averaged gradients (upper bound of the "
                                       "exposed wire time: early buckets complete behind the rest of the backward)"}}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if int(ones.item()) != world or not same:
            sys.exit(4)


def batched_views_bench(frame, k, steps, burst):
    """The frame step with k cameras of the pose rasterised as ONE batch (include/d3ga.h: d3ga_raster_params::n_views): upstream
    (LBS + cage deform) once, k views through one grid per rasterizer stage, loss = mean L1 over the k images (train.py:218-221
    averages the losses of a batch of frames), the whole backward.  One captured hipGraph, K replays; then an eager pass with
    HIP events per stage.  Returns the `batched_views` object of the bench line: frames/s counts VIEWS (k per step)."""
    from d3ga_amd import rasterizer as R
    from d3ga_amd.graph import CapturedStep
    from d3ga_amd.raster_views import CameraBatch
    from d3ga_amd.renderer import render_views
    dev, wl = frame.dev, frame.wl
    nv = max(8, k)
    batches = [frame.syn.make_batch(wl.width, wl.height, azimuth=2 * math.pi * v / nv, camera_id=v, fill=frame.fill) for v in range(k)]
    W, H = int(batches[0]["width"]), int(batches[0]["height"])      # (the raster: an odd image width is padded to the symmetric frustum, lib/batch.py:186-198)
    cams = CameraBatch(k, W, H, device=dev).set(batches)
    targets = torch.stack([torch.rand(3, H, W, generator=torch.Generator().manual_seed(100 + v)) for v in range(k)]).to(dev)
    one = torch.ones((), device=dev)
    params = list(frame.params.values())

    def step():
        loss = render_views(None, frame.upstream(), frame.bg, targets=targets, cameras=cams)["l1"]
        loss.backward(one)
        return loss

    def zero():
        for p in params:
            p.grad = None
    R.set_capacity_policy("auto")
    for _ in range(3):
        zero(); step()
    cnt = R.last_counters()
    R.set_capacity_policy("static", int(cnt["D"] * 1.25) + 4096)
    zero(); step()
    torch.cuda.synchronize()
    graph = CapturedStep(step, params=params)
    for _ in range(30):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    end = graph.check_overflow()
    del graph
    R.stage_timer.enabled = True
    R.stage_timer.reset()
    R.stage_timer.burst = {"composite_fwd": burst, "composite_bwd": burst}
    for _ in range(steps):
        zero(); step()
    torch.cuda.synchronize()
    R.stage_timer.enabled = False
    stages = R.stage_timer.summary()
    R.stage_timer.reset()
    P, M, D = wl.n_gaussians, frame.params["features"].shape[1], cnt["D"]
    tiles = math.ceil(W / 16) * math.ceil(H / 16)
    alg = {"preprocess": k * P * (88 + 12 * M), "bin_sort": 36 * D + 8 * tiles * k, "composite_fwd": 40 * D + 20 * W * H * k,
           "composite_bwd": 80 * D + 20 * W * H * k, "preprocess_bwd": k * P * (140 + 40) + P * 12 * M}
    kernels = {n: {"ms": round(ms, 4), "ms_per_view": round(ms / k, 4), "launches": c, "alg_bytes": alg[n],
                   "achieved_GBs": round(alg[n] / (ms * 1e-3) / 1e9, 1), "frac_hbm_peak": round(alg[n] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
               for n, (c, ms) in stages.items() if n in alg}
    worst = max((n for n in ("composite_bwd", "composite_fwd") if n in kernels), key=lambda n: kernels[n]["ms"], default=None)
    return {"views": k, "value": round(k * steps / dt, 3), "unit": "frames/s (views)", "ms_per_step": round(1e3 * dt / steps, 4),
            "ms_per_view": round(1e3 * dt / steps / k, 4), "steps": steps, "duplicates_D": D, "max_tile_list": cnt["max_tile"],
            "capacity_check": None if end is None else {"D": end["D"], "capacity": end["capacity"]},
            "step": "LBS + cage deform once, k views in one grid per rasterizer stage, mean L1 over the k images, the whole backward; "
                    "one hipGraph replay per step",
            "roofline": None if worst is None else {"kernel": worst, "bound": "hbm", "achieved": kernels[worst]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                                                    "unit": "GB/s", "frac": kernels[worst]["frac_hbm_peak"], "traffic": None,
                                                    "alg_bytes_per_launch": alg[worst], "avg_ms": kernels[worst]["ms"]},
            "kernels": kernels,
            "stage_events": f"separate eager pass, same K steps; compositing kernels launched {burst}x back to back per event pair"}


def batched_frames_bench(frame, k, steps):
    """The reference-faithful training step over a BATCH of k frames -- k poses, k cameras, RGB + silhouette render per frame, losses as
    train.py:190-193 averaged over the batch (train.py:218-221) -- as ONE view-batched rasterizer pass with per-frame geometry
    (d3ga_raster_params::per_view_geometry; LBS + cage deform per frame, one grid per rasterizer stage, the SSIM / L1 kernels over the
    3k channels) against the same k frames stepped one after the other (`Frame.train_step(pair=True)`, gradients accumulated).  One
    captured hipGraph each; ms per FRAME."""
    import numpy as np
    from d3ga_amd import rasterizer as R
    from d3ga_amd.cage_deform import lbs_cage_deform
    from d3ga_amd.graph import CapturedStep
    from d3ga_amd.losses import l1_loss, l1_ssim
    from d3ga_amd.raster_views import CameraBatch
    from d3ga_amd.renderer import render_pair, render_views
    dev, wl, p = frame.dev, frame.wl, frame.params
    joint_pos = frame.joint_pos
    poses = [torch.from_numpy(frame.syn.pose_matrices(joint_pos, np.random.default_rng(1000 + v))).to(dev) for v in range(k)]
    batches = [frame.syn.make_batch(wl.width, wl.height, azimuth=2 * math.pi * v / max(8, k), camera_id=v, fill=frame.fill) for v in range(k)]
    W, H = int(batches[0]["width"]), int(batches[0]["height"])
    cams = CameraBatch(k, W, H, device=dev).set(batches)
    targets = torch.stack([torch.rand(3, H, W, generator=torch.Generator().manual_seed(100 + v)) for v in range(k)]).to(dev)
    sil_t = (targets.mean(1, keepdim=True) > 0.5).float().expand(-1, 3, -1, -1).contiguous()
    P = frame.barys0.shape[0]
    sil_rgb, bg0, lam = torch.ones(P, 3, device=dev), torch.zeros(3, device=dev), 0.2
    params = list(p.values())

    def package(A):
        means, cov6, _ = lbs_cage_deform(frame.canon, p["delta_node"], A, frame.skin_idx, frame.skin_w, frame.tetras, frame.tetra_id,
                                         frame.barys0, frame.canon_grad, p["scaling"], p["rotation"], delta_barys=p["delta_bary"],
                                         scale_activation="exp", gradient_per_tet=frame.canon_grad_mode == "per-tet")
        return {"means3D": means, "cov3D_precomp": cov6, "opacity_logits": p["opacity"], "shs": p["features"], "rgb": None,
                "sh_degree": frame.sh_degree}

    def losses(img, sil, tgt, sil_tgt):
        rgb_l1, rgb_ssim = l1_ssim(img, tgt)
        return (1.0 - lam) * rgb_l1 + lam * (1.0 - rgb_ssim) + l1_loss(sil, sil_tgt)

    def step_batched():
        out = render_views(None, [package(A) for A in poses], frame.bg, cameras=cams, colors2=sil_rgb, bg_color2=bg0)
        loss = losses(out["render"].view(3 * k, H, W), out["render2"].view(3 * k, H, W), targets.view(3 * k, H, W), sil_t.view(3 * k, H, W))
        loss.backward()
        return loss

    def step_sequential():
        tot = None
        for v in range(k):
            both = render_pair(batches[v], package(poses[v]), frame.bg, sil_rgb, bg0)
            loss = losses(both["render"], both["render2"], targets[v], sil_t[v]) / k
            loss.backward()
            tot = loss.detach() if tot is None else tot + loss.detach()
        return tot

    def zero():
        for q in params:
            q.grad = None
    res = {}
    for name, fn in (("batched", step_batched), ("sequential", step_sequential)):
        R.set_capacity_policy("auto")
        for _ in range(3):
            zero(); fn()
        torch.cuda.synchronize()
        cap = int(R.last_counters()["D"] * 1.3) + 4096
        R.set_capacity_policy("static", cap)
        zero(); fn()
        torch.cuda.synchronize()
        graph = CapturedStep(fn, params=params)
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        res[name] = (1e3 * (time.perf_counter() - t0) / steps, float(graph.result.detach()), graph.check_overflow())
        grads = {n: q.grad.detach().clone() for n, q in p.items() if q.grad is not None}
        res[name + "_grads"] = grads
        del graph
    ga, gb = res["batched_grads"], res["sequential_grads"]
    agree = max(float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-30)) for n in gb)
    mb, ms_ = res["batched"][0], res["sequential"][0]
    return {"frames": k, "step": "k poses x k cameras: LBS + cage deform per frame, RGB + silhouette of all frames in one view-batched pass "
                                 "(per-frame geometry), 0.8 L1 + 0.2 (1 - SSIM) on RGB + L1 on the silhouette averaged over the batch, the whole "
                                 "backward; one hipGraph replay per step",
            "ms_per_step": round(mb, 4), "ms_per_frame": round(mb / k, 4), "frames_per_s": round(k * 1e3 / mb, 1),
            "sequential_ms_per_step": round(ms_, 4), "sequential_ms_per_frame": round(ms_ / k, 4),
            "speedup_over_sequential": round(ms_ / mb, 3), "loss_batched": round(res["batched"][1], 6),
            "loss_sequential": round(res["sequential"][1], 6), "max_rel_gradient_difference": float(f"{agree:.2e}"), "steps": steps}


def _self_launch(n):
    """`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset): re-exec this command line under
    `python -m torch.distributed.run`, one rank per GPU, exactly as the driver launches N > 1 -- a plain invocation must
    never measure one rank and call it N (VERDICT r3 missing #1).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), D3GA_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher (WORLD_SIZE unset): starting {n} ranks via torch.distributed.run on port {port}",
          file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.field_mlp:
        return field_mlp_bench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_cut:
        sys.exit(_self_launch(args.gpus))
    if args.train_step == "color":
        return color_train_bench(args)
    from d3ga_amd import dist as ddist
    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = ddist.init_process_group(args.backend)
    if world != max(args.gpus, 1):
        # a launcher that started another number of ranks than --gpus names: the line would be labelled with the wrong N
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line under the wrong N", file=sys.stderr)
        sys.exit(3)
    if world > 1 and not args.single_device and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"[bench] {world} ranks but {torch.cuda.device_count()} visible GPU(s): one GPU per rank is required "
                  f"(--single-device shares cuda:0 for functional checks only)", file=sys.stderr)
        sys.exit(3)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import d3ga_amd
    d3ga_amd.lib()
    from d3ga_amd import rasterizer as R

    if args.pmc and world == 1:
        collect_pmc(args)
    frame = Frame(args.workload, dev, view_index=rank % 8, scale_mult=args.scale_mult, fill=args.fill, order=args.gaussian_order,
                  canon_grad=args.canon_grad)
    frame.fused_l1 = not args.no_fused_l1
    frame.fused_lbs = not args.no_fused_lbs
    kv = max(int(args.views_per_rank), 1)
    if kv > 1:
        if world == 1 and not args.force_cut:
            raise SystemExit("--views-per-rank applies to the camera-sharded step (N > 1, or --force-cut on one GPU)")
        nv = max(8, world * kv)
        frame.batch_my_views = not args.sequential_views
        frame.my_views = []
        for j in range(kv):
            v = (rank * kv + j) % nv
            b = frame.syn.make_batch(frame.wl.width, frame.wl.height, azimuth=2 * math.pi * v / nv, camera_id=v, fill=args.fill)
            th, tw = (int(b["height"]), int(b["width"])) if frame.batch_my_views else (frame.wl.height, frame.wl.width)      # (batched: the raster; an odd width is padded, lib/batch.py:186-198)
            frame.my_views.append((b, torch.rand(3, th, tw, generator=torch.Generator().manual_seed(100 + v)).to(dev)))
    if not args.fresh_scratch:
        R.set_accumulator_policy("persistent")         # one training stream: the accumulator cleans itself
    flat = ddist.GradReducer(list(frame.params.values()))
    cut = world > 1 and args.reduce == "cut"
    if args.force_cut and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
        cut, args.fixed_camera = True, True
    cut_eager = [None]        # views-per-rank > 1: the eager step goes through the cut-step flow as well (ONE exchange per step)
    if cut:
        frame.grad_sync = ddist.ViewShardedGrads(timing=True)   # gradients leave the rasterizer already averaged over the ranks
        frame.grad_sync.always = args.force_cut
        if kv > 1:
            from d3ga_amd.graph import CapturedCutStep
            eager_sync = ddist.ViewShardedGrads(timing=True)   # its own instance: the captured variant freezes the buffers of the other
            eager_sync.always = args.force_cut

            def _cut_eager():                                  # upstream once | k renders to the cut | one exchange | the rest
                keep, frame.grad_sync = frame.grad_sync, eager_sync
                eager_sync.deferred = True
                try:
                    CapturedCutStep._eager(_CutShim(eager_sync), frame.upstream, frame.loss_from)
                finally:
                    eager_sync.deferred = False
                    frame.grad_sync = keep
            cut_eager[0] = _cut_eager
            cut_eager.append(eager_sync)
    # N = 1: the step follows the trainer -- a NEW camera and target every step (datasets/actorshq_dataset.py:229), written
    # into the static slots of ONE captured hipGraph (d3ga_amd/graph.py).  N > 1: every rank keeps its own view (camera
    # sharding: the views of one pose are spread over the ranks).
    cycle = world == 1 and not args.fixed_camera
    views = frame.camera_cycle() if cycle else None

    def set_view(i):
        if cycle:
            b, t = views[i % len(views)]
            frame.target.copy_(t, non_blocking=True)
            frame.target_slot.set(t)
            frame.slot.set(b)                         # one H2D copy: camera + the target's address

    def reduce_params():
        if not cut and world > 1:
            flat.all_reduce_mean()

    step_no = [0]

    def one_step():
        set_view(step_no[0]); step_no[0] += 1
        flat.zero()
        if cut_eager[0] is not None:
            cut_eager[0]()
        else:
            frame.step()
        reduce_params()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up (auto capacity: learns the duplicate count of every view), then freeze the capacity: no host sync inside a frame
    exchange_note = None
    if cut:
        try:                                   # first exchange: if the backend rejects a collective of the cut exchange,
            one_step()                         # fall back to the per-parameter all-reduce instead of losing the run
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            exchange_note = f"cut exchange failed ({type(e).__name__}: {e}); fell back to per-parameter all-reduce"
            print(f"[bench] rank {rank}: {exchange_note}", file=sys.stderr)
            cut = False
            frame.grad_sync = None
    d_max = 0
    for _ in range(max(args.warmup, 2, len(views) if cycle else 0)):
        one_step()
        d_max = max(d_max, R.last_counters()["D"])
    cnt = R.last_counters()
    R.set_capacity_policy("static", int(d_max * 1.25) + 4096)
    one_step()
    torch.cuda.synchronize()

    # The step is launch-bound on the host (~45 small launches): capture ONE whole step -- LBS, deform, rasterizer forward,
    # loss, the full backward -- into a hipGraph and replay it with this step's camera / target in its slots.  N > 1 runs
    # eagerly by default: the cut exchange sits INSIDE the backward, so a captured step contains the collectives;
    # --graph-collectives captures them too (RCCL supports capture; verified here only on a one-rank group,
    # tests/test_gpu_view_sharded.py -- the 8-GPU run keeps the conservative default).
    graph = None
    graph_one = None            # candidate: ONE hipGraph of the whole N > 1 step, the RCCL collectives captured (probed below)
    if not args.no_graph and cut and not args.graph_collectives and not args.single_device:
        # (--single-device puts several ranks on ONE GPU for functional checks: their graph launches are time-sliced by the
        # driver in ~100 ms quanta and every later launch of the process suffers -- eager there)
        # N > 1 with the cut exchange: TWO graphs, the collectives issued eagerly between them (d3ga_amd/graph.py:
        # CapturedCutStep) -- no collective is captured, and the step is no longer bound by the host
        from d3ga_amd.graph import CapturedCutStep, CapturedStep
        if not args.no_graph_collectives and torch.distributed.get_backend() == "nccl":
            try:          # (k views per rank: the eager cut flow -- upstream once, k renders, ONE exchange, the rest -- in one graph)
                graph_one = CapturedStep(cut_eager[0] if cut_eager[0] is not None else frame.step, params=list(frame.params.values()))
            except Exception as e:  # noqa: BLE001
                exchange_note = (exchange_note or "") + f" one-graph capture incl. the collectives failed ({type(e).__name__}: {e});"
                graph_one = None
        try:
            graph = CapturedCutStep(frame.upstream, frame.loss_from, frame.grad_sync, params=list(frame.params.values()))
            exchange_note = (exchange_note or "") + " step captured as two hipGraphs around the eager exchange"
        except Exception as e:  # noqa: BLE001
            exchange_note = (exchange_note or "") + f" two-graph capture of the N>1 step failed ({type(e).__name__}: {e}); eager"
            print(f"[bench] rank {rank}: {exchange_note}", file=sys.stderr)
            graph = None
    elif not args.no_graph and (world == 1 or args.graph_collectives):
        from d3ga_amd.graph import CapturedStep
        try:
            graph = CapturedStep(frame.step, params=list(frame.params.values()),
                                 slots={"target": frame.target_slot} if cycle else None, camera=frame.slot if cycle else None)
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            exchange_note = (exchange_note or "") + f" graph capture of the N>1 step failed ({type(e).__name__}); eager"
            graph = None

    def timed_step():
        i = step_no[0]; step_no[0] += 1
        if graph is not None:
            if cycle:
                b, t = views[i % len(views)]
                graph.replay(camera=b, target=t)      # gradients land in the graph's static .grad tensors
            else:
                graph.replay()
        else:
            set_view(i)
            flat.zero()
            if cut_eager[0] is not None:
                cut_eager[0]()
            else:
                frame.step()
        reduce_params()
        if watchdog is not None:
            watchdog.arm(("step", i))

    # a rank whose device work never completes (a collective a peer never joined) ends with exit code 124 instead of hanging
    watchdog = None
    if world > 1 and args.watchdog > 0 and dev.type == "cuda":
        from d3ga_amd.graph import ReplayWatchdog
        watchdog = ReplayWatchdog(timeout_s=args.watchdog)

    # N > 1: the N = 1 step of THIS run, on every rank, before the group step is timed -- the same frame without the exchange,
    # ONE captured hipGraph, K replays with this rank's camera (slowest rank counts).  `distributed.efficiency` = n1 / N-step
    # time is then a statement of the run itself (weak scaling: every rank renders one view either way).
    n1_ms = None
    if world > 1 or args.force_cut:
        from d3ga_amd.graph import CapturedStep
        keep_sync, frame.grad_sync = frame.grad_sync, None
        try:
            for _ in range(2):
                flat.zero(); frame.step()
            torch.cuda.synchronize()
            cap1 = CapturedStep(frame.step, params=list(frame.params.values()))
            for _ in range(3):
                cap1.replay()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                cap1.replay()
            barrier()
            t_n1 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if world > 1:
                torch.distributed.all_reduce(t_n1, op=torch.distributed.ReduceOp.MAX)
            n1_ms = 1e3 * float(t_n1) / args.steps
            cap1.check_overflow()
            del cap1
        except Exception as e:  # noqa: BLE001  (the reference number must never take the N > 1 line down)
            exchange_note = (exchange_note or "") + f" in-run N=1 reference failed ({type(e).__name__}: {e})"
        finally:
            frame.grad_sync = keep_sync
            flat.zero()

    if world > 1 and cut and not args.no_graph and not args.single_device:
        # the candidates must be the same on every rank (a rank whose capture failed would skip collectives the others issue)
        have = torch.tensor([1.0 if graph_one is not None else 0.0, 1.0 if (graph is not None and hasattr(graph, "graph_b")) else 0.0], device=dev)
        torch.distributed.all_reduce(have, op=torch.distributed.ReduceOp.MIN)
        if float(have[0]) == 0.0:
            graph_one = None
        if float(have[1]) == 0.0 and graph is not None and hasattr(graph, "graph_b"):
            graph = None
    if (graph is not None and hasattr(graph, "graph_b")) or graph_one is not None:
        # Which launch mode for the N > 1 step?  (a) ONE hipGraph incl. the collectives (0.436 vs 0.419 ms for the N = 1 step at
        # C3 over a one-rank RCCL group: 0.96 per rank), (b) two graphs around the eager exchange (0.55), (c) eager (0.56).
        # (a) is accepted only when 3 replays reproduce the EAGER step's gradients on EVERY rank (checksums, agreed with an
        # all-reduce); a captured collective that hangs instead ends the rank through the watchdog (exit code 124).  All three are
        # then timed (ranks in step, slowest rank counts) and the fastest is kept.
        def grad_fingerprint():
            gs = [p.grad for p in frame.params.values() if p.grad is not None]
            return torch.stack([g.double().abs().sum() for g in gs] + [g.double().sum() for g in gs])

        def probe(g, n=12):
            nonlocal graph
            keep, graph = graph, g
            for _ in range(3):
                timed_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                timed_step()
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], device=dev)
            if world > 1:
                torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
            graph = keep
            return float(dt) / n
        cands = {}
        if graph_one is not None:
            keep, graph = graph, None
            for _ in range(2):
                timed_step()                                # the eager step of this rank's view: the reference gradients
            torch.cuda.synchronize()
            want = grad_fingerprint()
            graph = graph_one
            for _ in range(3):
                timed_step()
            torch.cuda.synchronize()
            got = grad_fingerprint()
            graph = keep
            agree = torch.tensor([1.0 if bool(((got - want).abs() <= 1e-4 * want.abs().max() + 1e-12).all()) else 0.0], device=dev)
            if world > 1:
                torch.distributed.all_reduce(agree, op=torch.distributed.ReduceOp.MIN)
            if float(agree) == 1.0:
                cands["one hipGraph incl. the collectives"] = graph_one
            else:
                exchange_note = (exchange_note or "") + " one-graph step incl. the collectives did NOT reproduce the eager gradients on every rank: dropped;"
        if graph is not None and hasattr(graph, "graph_b"):
            cands["two hipGraphs around the eager exchange"] = graph
        cands["eager"] = None
        times = {k: probe(g) for k, g in cands.items()}
        best = min(times, key=times.get)
        exchange_note = (exchange_note or "") + " (probe: " + ", ".join(f"{k} {1e3 * v:.3f} ms/step" for k, v in times.items()) + f": kept {best})"
        graph = cands[best]

    # Python's cyclic GC: the first full collection of the (large) post-import heap costs ~40 ms and lands wherever it likes --
    # seen as one 3 ms/step outlier in an eager loop of 1.4 ms steps.  Collect now and freeze the survivors.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(3):
        timed_step()
    # Steady state: the first few hundred steps of a process run ~3 % slower than the rest (clocks and caches settling: the
    # per-step HIP-event times below show it as well as the wall clock), and with the driver's K = 20 the whole measurement would
    # sit inside that transient.  `--settle` more untimed steps of exactly the timed kind first (default 200, ~0.1 s at C3).
    for _ in range(max(0, args.settle)):
        timed_step()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    # Per-step distribution (HIP events around the op sequence, median / p10 / p90), outside (just before) the
    # timed region: one event pair per step on the launch stream, K more steps of the same kind (graph replays at N = 1).
    step_dist = None
    if world == 1:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for e0, e1 in evs:
            e0.record()
            timed_step()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        pick = lambda q: round(ts[min(len(ts) - 1, int(q * len(ts)))], 4)
        step_dist = {"median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9), "n": len(ts),
                     "frames_per_s_at_median": round(1e3 / pick(0.5), 1), "method": "HIP events per step, before the timed region"}

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed_step()
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the K steps (diagnostic: host- vs GPU-bound)
    barrier()
    dt = time.perf_counter() - t0
    cnt_end = R.last_counters()
    assert not cnt_end["overflow"], f"binning capacity overflowed inside the timed region: result invalid ({cnt} -> {cnt_end})"

    # Per-stage kernel times: HIP events on the launch stream around every rasterizer stage, over the same K steps.
    # Events cannot be recorded inside a graph replay, so with --graph this is a second, eager pass over the same
    # workload (the kernels and their inputs are identical); with --no-graph it IS the timed region's stream.
    if not args.no_stage_events:
        R.stage_timer.enabled = True
        R.stage_timer.reset()
        # the two compositing kernels (the roofline kernels) are launched 4x back to back between their event pair: the quotient
        # is the kernel's duration without the dispatch latency of a lone launch on an idle queue (rasterizer.StageTimer)
        R.stage_timer.burst = {"composite_fwd": args.stage_burst, "composite_bwd": args.stage_burst}
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        stages_burst = R.stage_timer.summary()
        # ... and the same pass with ONE launch per event pair (launches 2..R of a burst find the lists warm in L2 / the
        # Infinity Cache, so the burst figure is the kernel's duration on warm caches; the lone launch also carries the dispatch
        # latency of a packet on an idle queue -- the truth for the captured step lies between them, rocprofv3's trace of the
        # same command is the third opinion under profiles/)
        R.stage_timer.reset()
        R.stage_timer.burst = {}
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        stages_single = R.stage_timer.summary()
        R.stage_timer.enabled = False
        R.stage_timer.reset()

    # the reference-faithful training step (RGB + silhouette render), reported beside the headline frame
    train = None
    if world == 1 and not args.no_train_step:
        for _ in range(3):
            flat.zero()
            frame.train_step()
        assert not R.last_counters()["overflow"]
        torch.cuda.synchronize()
        n_ts = max(20, args.steps)      # (eager steps are host-bound: a 10-step window is mostly pipeline start-up)

        def timed_train(captured=False, **kw):
            """ms per training step (a new camera + target every step): median of three runs of n_ts steps (an occasional host
            stall of tens of ms -- Python GC / allocator -- otherwise lands in one run's mean).  captured: ONE hipGraph of
            the step, replayed with the step's camera / target in its slots (d3ga_amd/graph.py)."""
            cap = None
            if captured:
                from d3ga_amd.graph import CapturedStep
                leaves = list(frame.params.values()) + list(getattr(frame, "field_params", []))
                leaves += [t for t in (getattr(frame, "color_feat", None), getattr(frame, "frame_enc", None)) if t is not None]
                for _ in range(2):
                    flat.zero(); frame.train_step(**kw)
                leaves = list(frame.params.values()) + list(getattr(frame, "field_params", []))
                leaves += [t for t in (getattr(frame, "color_feat", None), getattr(frame, "frame_enc", None)) if t is not None]
                cap = CapturedStep(lambda: frame.train_step(**kw), params=leaves, slots={"target": frame.target} if cycle else None,
                                   camera=frame.slot if cycle else None)

            def one(i):
                if cap is not None:
                    if cycle:
                        cap.replay(camera=views[i % len(views)][0], target=views[i % len(views)][1])
                    else:
                        cap.replay()
                else:
                    set_view(i)
                    flat.zero()
                    frame.train_step(**kw)
            for i in range(3):
                one(i)
            runs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n_ts):
                    one(i)
                torch.cuda.synchronize()
                runs.append(1e3 * (time.perf_counter() - t1) / n_ts)
            return round(sorted(runs)[1], 4)

        train = {"renders_per_step": 2, "losses": "0.8 L1 + 0.2 (1 - SSIM) on RGB, L1 on the silhouette", "steps": n_ts,
                 "cameras": "a new camera and target image every step (8 views)" if cycle else "one fixed camera",
                 "launch_mode": "eager", "timing": "median of 3 runs of `steps` steps", "ms_per_step": timed_train()}
        train["steps_per_s"] = round(1e3 / train["ms_per_step"], 2)
        # the same steps as ONE captured hipGraph each, replayed with the step's camera / target
        try:
            train["captured_ms_per_step"] = timed_train(captured=True)
            train["captured_render_pair_ms_per_step"] = timed_train(captured=True, pair=True)
        except Exception as e:  # noqa: BLE001
            train["captured_error"] = repr(e)

        # the same step with both images from one compositing pass (render_pair: an extension, same results)
        train["render_pair_ms_per_step"] = timed_train(pair=True)
        train["render_pair_with_field_and_color_networks_ms_per_step"] = timed_train(pair=True, with_fields="color")
        # the same step with the field networks in front of the deform (models/cage_net.py:197-215)
        train["with_field_networks_ms_per_step"] = timed_train(with_fields=True)
        # the reference's main configuration (configs/actorshq_actor02.yml: use_shs false): ColorField supplies colour
        # and opacity as well
        train["with_field_and_color_networks_ms_per_step"] = timed_train(with_fields="color")
        try:          # the same two steps as ONE captured hipGraph each (a captured figure for the real configuration)
            train["with_field_and_color_networks_captured_ms_per_step"] = timed_train(captured=True, with_fields="color")
            train["render_pair_with_field_and_color_networks_captured_ms_per_step"] = timed_train(captured=True, pair=True, with_fields="color")
        except Exception as e:  # noqa: BLE001
            train["captured_networks_error"] = repr(e)

    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist_info = None
    if world == 1 and args.force_cut and cut:
        tsync = cut_eager[1] if len(cut_eager) > 1 else frame.grad_sync
        tsync.exchange_ms()
        for _ in range(5):
            one_step()
        ex_ms = tsync.exchange_ms()
        step_ms_all = 1e3 * dt / args.steps
        dist_info = {"nranks_seen": 1, "n1_ms_per_step": None if n1_ms is None else round(n1_ms, 4),
                     "efficiency": None if n1_ms is None else round(n1_ms / step_ms_all, 4),
                     "exchange_ms": None if ex_ms is None else round(ex_ms, 4),
                     "compute_ms": None if ex_ms is None else round(step_ms_all - ex_ms, 4),
                     "note": "--force-cut on ONE GPU: the collectives are identities; efficiency here = what the camera-sharded step "
                             "costs per rank apart from the wire time (launch mode, SH rebuild, parked buffers)"}
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        # compute / exchange split of the N > 1 step: HIP events around every exchange (dist.ViewShardedGrads timing) resp.
        # around the parameter all-reduce, over a few more steps; nranks_seen = what an all-reduce of ones returns
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ex_ms = None
        if cut:
            tsync = cut_eager[1] if len(cut_eager) > 1 else frame.grad_sync
            tsync.exchange_ms()                                # drop what the timed region recorded
            for _ in range(5):
                one_step()
            ex_ms = tsync.exchange_ms()
        else:
            evs = []
            for _ in range(5):
                flat.zero(); frame.step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); flat.all_reduce_mean(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ex_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        step_ms_all = 1e3 * float(tmax.item()) / args.steps
        dist_info = {"nranks_seen": int(ones.item()),
                     "n1_ms_per_step": None if n1_ms is None else round(n1_ms, 4),
                     "efficiency": None if n1_ms is None else round(n1_ms / step_ms_all, 4),
                     "efficiency_note": "weak scaling of this run: (N = 1 captured step of the same frame, timed on every rank just before, "
                                        "slowest rank) / (N-rank step); 1.0 = the exchange and the N > 1 launch mode cost nothing",
                     "exchange_ms": None if ex_ms is None else round(ex_ms, 4),
                     "compute_ms": None if ex_ms is None else round(step_ms_all - ex_ms, 4),
                     "note": "exchange_ms: HIP events around the collectives of one step (all-reduce + all-gather in flight "
                             "together); compute_ms = ms_per_step - exchange_ms (the exchange is not overlapped with compute)",
                     "watchdog_s": args.watchdog if watchdog is not None else None}
    dt = float(tmax.item())

    if rank == 0:
        wl = frame.wl
        P, W, H, D = wl.n_gaussians, frame.batch["width"], frame.batch["height"], cnt_end["D"]
        M = frame.params["features"].shape[1]
        stages = stages_burst if not args.no_stage_events else {}
        alg = {   # algorithmic bytes per launch
            "preprocess": P * (88 + 12 * M),
            "bin_sort": 36 * D + 8 * (math.ceil(W / 16) * math.ceil(H / 16)),
            "composite_fwd": 40 * D + 20 * W * H,
            "composite_bwd": 80 * D + 20 * W * H,
            "preprocess_bwd": P * (140 + 40 + 12 * M),
        }
        kernels = {k: {"ms": round(ms, 4), "launches": n, "alg_bytes": alg[k],
                       "achieved_GBs": round(alg[k] / (ms * 1e-3) / 1e9, 1),
                       "frac_hbm_peak": round(alg[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                   for k, (n, ms) in stages.items() if k in alg}
        if not args.no_stage_events:
            for k in kernels:
                if k in stages_single:
                    kernels[k]["ms_single_launch"] = round(stages_single[k][1], 4)
        # HBM traffic of the compositing kernels from the committed rocprofv3 --pmc passes of this same command
        # (bench.py --pmc -> profiles/pmc_<workload>.json): (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH doubled per the gfx950
        # correction of MI355X_MICROARCH.md.  None when no PMC summary for this workload is present.
        pmc_traffic, pmc_lds, pmc_valu, pmc_all, pmc_src, pmc_stale = {}, {}, {}, {}, None, None
        try:
            suffix = "" if (args.scale_mult == 1.0 and args.fill == 0.85) else f"_s{args.scale_mult:g}_f{args.fill:g}"
            raw, pmc_src, pmc_stale = pmc_summary(args.workload, suffix)
            for kname, c in raw.items():
                short = re.sub(r"_(rows\d*|scan|tile|q)(<.*)?$", "", kname.split("::")[-1].split("_kernel")[0])
                pmc_all[short] = c
                if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    pmc_traffic[short] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
                    if c.get("SQ_LDS_IDX_ACTIVE"):     # LDS bank-conflict cycles / LDS-active cycles
                        pmc_lds[short] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
                    if c.get("SQ_INSTS_VALU"):
                        pmc_valu[short] = int(c["SQ_INSTS_VALU"])
        except Exception:
            pmc_traffic, pmc_all = {}, {}
        roof = None
        comp = [k for k in ("composite_bwd", "composite_fwd") if k in kernels]
        if comp:
            k = max(comp, key=lambda n: kernels[n]["ms"])
            roof = {"kernel": k, "bound": "hbm", "achieved": kernels[k]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kernels[k]["frac_hbm_peak"], "traffic": pmc_traffic.get(k),
                    "lds_bank_conflict_per_lds_active": pmc_lds.get(k), "valu_wave_instructions": pmc_valu.get(k),
                    "alg_bytes_per_launch": alg[k], "avg_ms": kernels[k]["ms"],
                    "traffic_source": ("%s (bench.py --pmc: separate rocprofv3 --pmc passes of this command; "
                                       "(2 x FETCH_SIZE + WRITE_SIZE) KiB)" % pmc_src) if pmc_traffic.get(k) else None,
                    # True: the counters were collected on another build of libd3ga_hip.so than the one timed here
                    "traffic_stale": pmc_stale if pmc_traffic.get(k) else None}
            # VALU issue share, calibrated: wave-instructions x the kernel's average issue cost (static mix
            # of its loop x the measured cycles of each instruction class) / (1024 SIMDs x launch cycles at the measured clock)
            c = pmc_all.get(k, {})
            avg_cyc, model = valu_issue_model()
            if k == "composite_bwd" and avg_cyc and c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
                launch_cycles = c["GRBM_GUI_ACTIVE"] / 8.0               # summed over the 8 XCDs
                roof["valu_frac"] = round(c["SQ_INSTS_VALU"] * avg_cyc / (1024.0 * launch_cycles), 4)
                # the issue floor of the launch: every VALU wave-instruction at its calibrated cost on 1024 SIMDs at the peak clock --
                # what the kernel would take if nothing but instruction issue limited it (the distance to the 0.40 target as a number)
                floor_us = c["SQ_INSTS_VALU"] * avg_cyc / (1024.0 * 2.4e9) * 1e6
                roof["valu_floor_us"] = round(floor_us, 1)
                roof["frac_at_valu_floor"] = round(alg[k] / (floor_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                roof["valu_model"] = {"avg_cycles_per_valu_instruction": round(avg_cyc, 3), **model,
                                      "source": "issue cost per instruction class: tools/micro/valu_issue.hip (profiles/r02_valu_issue_pmc.json -- a "
                                                "property of the chip, not of a build) x the static mix of THIS round's kernel "
                                                "(tools/isa_mix.py -> the newest profiles/r*_composite_bwd_mix.json)"}
            le = lane_efficiency(args.workload)
            if le:
                roof["lane_efficiency"] = le
        out = {
            "metric": "fwd+bwd frames/sec @500k Gaussians 1920x1080" if args.workload == "C3" else f"fwd+bwd frames/sec ({wl.name})",
            "value": round(world * kv * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": max(0, args.settle), "lib_sha256": lib_sha256(),
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name + ("" if (args.scale_mult == 1.0 and args.fill == 0.85) else
                                              f" [sensitivity: scales x{args.scale_mult:g}, body fills {args.fill:g} of the image height]")
                                   + {"tet": "", "random": " [sensitivity: Gaussians in random index order]",
                                      "morton": " [sensitivity: Gaussians shuffled, then numbered by tetra.spatial_order]"}[args.gaussian_order],
                       "gaussians": P, "width": W, "height": H, "sh_degree": wl.sh_degree,
                       "canon_grad": frame.canon_grad_mode, "duplicates_D": D, "max_tile_list": cnt_end["max_tile"], "visible": cnt_end["visible"],
                       "views_per_step": world * kv, "views_per_rank": kv, "parallelism": f"camera-sharded dp{world}",
                       "grad_exchange": ("none" if world == 1 else "cut: all-reduce of the rasterizer-input gradients + "
                                         "all-gather of the factored SH gradient" if cut else "per-parameter all-reduce"),
                       "grad_exchange_bytes_per_rank": (0 if world == 1 else frame.grad_sync.bytes_last if cut
                                                        else flat.nbytes())},
            "roofline": roof, "kernels": kernels,
            "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 4),
            "step_ms": step_dist,
            "launch_mode": ("ONE hipGraph of the N > 1 step, the RCCL collectives captured (probed against the eager step on every rank)"
                            if graph is not None and graph is graph_one else "hipGraph replay of ONE captured step; before every replay a new camera (matrices + FoV) is written into "
                            "its static slot and the loss is pointed at that camera's resident target image (8-byte cell, "
                            "d3ga_amd/graph.py: TensorSlot)" if graph is not None and cycle else
                            "two hipGraphs per step (up to the rasterizer's backward | the rest of the backward) with the gradient "
                            "exchange issued eagerly between them" if graph is not None and hasattr(graph, "graph_b") else
                            "hipGraph replay of one captured step" if graph is not None else "eager"),
            **({"distributed": dict(dist_info, launch_mode=("two hipGraphs around the eager exchange" if graph is not None and hasattr(graph, "graph_b")
                                                             else "one hipGraph incl. the collectives" if graph is not None else "eager"))}
               if dist_info else {}),
            **({"grad_exchange_note": exchange_note} if exchange_note else {}),
            "stage_events": ("none" if args.no_stage_events else ("separate eager pass, same K steps" if graph is not None else "eager pass")
                             + f"; compositing kernels launched {args.stage_burst}x back to back per event pair"),
        }
        if train is not None:
            out["training_step"] = train
        if world == 1 and not args.no_train_step:
            try:
                out["field_mlp"] = field_mlp_summary(P, dev, steps=args.steps)
            except Exception as e:  # noqa: BLE001
                out["field_mlp"] = {"error": repr(e)}
        if world == 1 and args.batch_views > 1 and not args.force_cut:
            try:
                out["batched_views"] = batched_views_bench(frame, int(args.batch_views), args.steps, args.stage_burst)
            except Exception as e:  # noqa: BLE001  (an extra regime must never take the headline down)
                out["batched_views"] = {"error": repr(e)}
            if not args.no_train_step:
                try:
                    out["batched_frames"] = batched_frames_bench(frame, int(args.batch_views), max(10, args.steps // 2))
                except Exception as e:  # noqa: BLE001
                    out["batched_frames"] = {"error": repr(e)}
        if args.init_timing:
            try:
                out["init"] = init_timing(frame)
            except Exception as e:  # noqa: BLE001
                out["init"] = {"error": repr(e)}
        if roof is not None:
            try:
                roof["measured_copy_GBs"] = round(measured_copy_gbs(dev), 1)
            except Exception as e:
                roof["measured_copy_GBs"] = None
        if not args.no_cpu_baseline and world == 1:
            try:
                same_gpu = deform_gpu_comparison(frame)          # before the CPU leg: the GPU is still at full clocks
                out["cpu_baseline"] = cpu_baseline(args.workload)
                out["cpu_baseline"]["deform_same_gpu"] = same_gpu
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if dist_info and dist_info.get("nranks_seen") != world:      # an all-reduce of ones over the group must count every rank
            print(f"[bench] nranks_seen {dist_info.get('nranks_seen')} != WORLD_SIZE {world}", file=sys.stderr)
            sys.exit(4)


if __name__ == "__main__":
    main()
