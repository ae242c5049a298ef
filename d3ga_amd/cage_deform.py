"""Autograd ops of the cage side of the hot path: LBS of cage vertices, fused tet-cage deformation, FEM energy.

Host-side mirror of the reference's per-frame tensor program:
    lib/smplman.py:155-171        -> lbs_cage
    models/cage_net.py:218-230    -> cage_deform      (replaces tetpoints[tetra_faces], compute_def_grad,
                                                        J S J^T, strip_symmetric and the bary einsum)
    lib/cage.py:349-361           -> fem_energy
All three call hand-written gfx950 kernels through the C ABI (include/d3ga.h); GPU tensors only.
"""
import os

import torch

from . import _lib
from ._lib import check, dptr, require_cuda, stream_handle


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_i32_cache = {}


def _i32c(t):
    """int32 contiguous view of an index buffer.  The reference registers int64 buffers (lib/cage.py:331-337); their
    int32 copies are cached (keyed on storage + version, holding the source alive) so that per-frame calls neither
    convert again nor defeat the adjacency cache below."""
    if t.dtype == torch.int32 and t.is_contiguous():
        return t
    key = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
    hit = _i32_cache.get(key)
    if hit is None:
        if len(_i32_cache) > 64:
            _i32_cache.clear()
        hit = (t, t.to(torch.int32).contiguous())
        _i32_cache[key] = hit
    return hit[1]


_adjacency_cache = {}


def vertex_adjacency(tetras, tetra_id, n_vertices):
    """Static CSR adjacency vertex -> items (4*gaussian + corner) of a cage binding; built once and cached.
    (tetra_id / tetras are buffers fixed at initialisation, lib/cage.py:331-337.)"""
    key = (tetras.data_ptr(), tetra_id.data_ptr(), tetras._version, tetra_id._version, tetra_id.shape[0], n_vertices)
    hit = _adjacency_cache.get(key)
    if hit is None:
        with torch.no_grad():
            vid = tetras.long()[tetra_id.long()].reshape(-1)                    # (4P,) vertex of item 4*i + corner
            order = torch.sort(vid, stable=True)[1]
            counts = torch.bincount(vid, minlength=n_vertices)
            start = torch.zeros(n_vertices + 1, dtype=torch.int64, device=vid.device)
            start[1:] = torch.cumsum(counts, 0)
            hit = (start.to(torch.int32).contiguous(), order.to(torch.int32).contiguous(), tetras, tetra_id)
        if len(_adjacency_cache) > 64:
            _adjacency_cache.clear()
        _adjacency_cache[key] = hit          # holds tetras / tetra_id alive: their addresses cannot be recycled
    return hit[0], hit[1]


_plan_cache = {}
_MERGE_BLOCK = 256        # = kBlock of csrc/deform.hip: the Gaussians of one workgroup
# D3GA_DEFORM_MERGE=0: per-corner gradient records + a gather over all 4P items (rounds 1-3; kept for A/B runs)
_merge_policy = {"enabled": os.environ.get("D3GA_DEFORM_MERGE", "1") != "0"}


def merge_plan(tetras, tetra_id, n_vertices):
    """Static plan of the block-merged backward (d3ga_cage_deform_bwd_merged), built once per binding and cached:
    -> dict(item_pos (P,4) int16-as-uint16, seg_ptr (blocks+1) int32, seg_begin (segments) uint16 stored as int16,
            vert_start (V+1) int32, vert_parts (segments) int32, n_segments).
    Items are (Gaussian, corner) pairs, 1024 per workgroup of 256 consecutive Gaussians; inside a workgroup they are ordered
    by cage vertex (stable), a SEGMENT is a run of equal vertex, and every segment's sum becomes one partial."""
    key = (tetras.data_ptr(), tetra_id.data_ptr(), tetras._version, tetra_id._version, tetra_id.shape[0], n_vertices)
    hit = _plan_cache.get(key)
    if hit is None:
        with torch.no_grad():
            dev = tetra_id.device
            P = tetra_id.shape[0]
            per = 4 * _MERGE_BLOCK
            vid = tetras.long()[tetra_id.long()].reshape(-1)                    # (4P,) vertex of item 4 i + corner
            item = torch.arange(4 * P, device=dev)
            blk = item // per
            order = torch.sort(blk * n_vertices + vid, stable=True)[1]          # items by (workgroup, vertex), stable
            inv = torch.empty_like(order)
            inv[order] = item
            item_pos = (inv - blk * per).to(torch.int16).reshape(P, 4).contiguous()      # < 1024: the bit pattern of a uint16
            key_sorted = (blk * n_vertices + vid)[order]
            first = torch.ones(4 * P, dtype=torch.bool, device=dev)
            first[1:] = key_sorted[1:] != key_sorted[:-1]
            seg_first = torch.nonzero(first).reshape(-1)                        # sorted index of every segment's first item
            seg_key = key_sorted[seg_first]
            seg_blk, seg_vid = seg_key // n_vertices, seg_key % n_vertices
            nb = (P + _MERGE_BLOCK - 1) // _MERGE_BLOCK
            seg_ptr = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
            seg_ptr[1:] = torch.cumsum(torch.bincount(seg_blk, minlength=nb), 0)
            seg_begin = (seg_first - seg_blk * per).to(torch.int16).contiguous()
            nseg = int(seg_first.numel())
            order2 = torch.sort(seg_vid, stable=True)[1]
            vstart = torch.zeros(n_vertices + 1, dtype=torch.int64, device=dev)
            vstart[1:] = torch.cumsum(torch.bincount(seg_vid, minlength=n_vertices), 0)
            hit = dict(item_pos=item_pos, seg_ptr=seg_ptr.to(torch.int32).contiguous(), seg_begin=seg_begin,
                       vert_start=vstart.to(torch.int32).contiguous(), vert_parts=order2.to(torch.int32).contiguous(),
                       n_segments=nseg, pins=(tetras, tetra_id))
        if len(_plan_cache) > 64:
            _plan_cache.clear()
        _plan_cache[key] = hit               # holds tetras / tetra_id alive: their addresses cannot be recycled
    return hit


class _CageDeform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations, delta_barys, flags):
        require_cuda(tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations)
        tetpoints, barys, canon_grad, scales, rotations, delta_barys = map(
            _f32c, (tetpoints, barys, canon_grad, scales, rotations, delta_barys))
        P = barys.shape[0]
        means = torch.empty((P, 3), dtype=torch.float32, device=barys.device)
        cov6 = torch.empty((P, 6), dtype=torch.float32, device=barys.device)
        check(_lib.lib().d3ga_cage_deform_fwd_ex(P, dptr(tetpoints), dptr(tetras), dptr(tetra_id), dptr(barys),
                                                 dptr(canon_grad), dptr(scales), dptr(rotations), dptr(delta_barys),
                                                 flags, dptr(means), dptr(cov6), stream_handle()),
              "d3ga_cage_deform_fwd_ex")
        ctx.flags = flags
        ctx.has_delta = delta_barys is not None
        ctx.set_materialize_grads(False)                 # backward() fills in the gradient of an unused output itself
        ctx.save_for_backward(tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations,
                              delta_barys if delta_barys is not None else torch.empty(0, device=barys.device))
        return means, cov6

    @staticmethod
    def backward(ctx, g_means, g_cov6):
        tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations, delta_barys = ctx.saved_tensors
        if not ctx.has_delta:
            delta_barys = None
        P, V = barys.shape[0], tetpoints.shape[0]
        dev = barys.device
        g_means = torch.zeros((P, 3), device=dev) if g_means is None else _f32c(g_means)
        g_cov6 = torch.zeros((P, 6), device=dev) if g_cov6 is None else _f32c(g_cov6)
        need = ctx.needs_input_grad
        g_tp = torch.empty((V, 3), dtype=torch.float32, device=dev) if need[0] else None
        g_b = torch.empty((P, 4), dtype=torch.float32, device=dev) if (need[3] or need[7]) else None
        g_s = torch.empty((P, 3), dtype=torch.float32, device=dev) if need[5] else None
        g_r = torch.empty((P, 4), dtype=torch.float32, device=dev) if need[6] else None
        vstart = vitems = corner = None
        if need[0] and P > 0 and _merge_policy["enabled"]:
            plan = merge_plan(tetras, tetra_id, V)
            partials = torch.empty((plan["n_segments"], 3), dtype=torch.float32, device=dev)
            check(_lib.lib().d3ga_cage_deform_bwd_merged(
                P, V, dptr(tetpoints), dptr(tetras), dptr(tetra_id), dptr(barys), dptr(canon_grad), dptr(scales),
                dptr(rotations), dptr(delta_barys), ctx.flags, dptr(g_means), dptr(g_cov6), dptr(g_tp), dptr(g_b), dptr(g_s),
                dptr(g_r), dptr(plan["item_pos"]), dptr(plan["seg_ptr"]), dptr(plan["seg_begin"]), plan["n_segments"],
                dptr(plan["vert_start"]), dptr(plan["vert_parts"]), dptr(partials), stream_handle()), "d3ga_cage_deform_bwd_merged")
            return (g_tp, None, None, g_b if need[3] else None, None, g_s, g_r, g_b if need[7] else None, None)
        if need[0] and P > 0:
            vstart, vitems = vertex_adjacency(tetras, tetra_id, V)
            corner = torch.empty((P, 4, 3), dtype=torch.float32, device=dev)
        check(_lib.lib().d3ga_cage_deform_bwd_ex(P, V, dptr(tetpoints), dptr(tetras), dptr(tetra_id), dptr(barys),
                                                 dptr(canon_grad), dptr(scales), dptr(rotations), dptr(delta_barys),
                                                 ctx.flags, dptr(g_means), dptr(g_cov6), dptr(g_tp), dptr(g_b),
                                                 dptr(g_s), dptr(g_r), dptr(vstart), dptr(vitems), dptr(corner),
                                                 stream_handle()), "d3ga_cage_deform_bwd_ex")
        return (g_tp, None, None, g_b if need[3] else None, None, g_s, g_r, g_b if need[7] else None, None)


def cage_deform(tetpoints, tetras, tetra_id, barys, canonical_gradient, scales, rotations, delta_barys=None,
                scale_activation=None, gradient_per_tet=None):
    """(tetpoints (V,3), tetras (T,4), tetra_id (P), barys (P,4), canonical_gradient (P,3,3), scales (P,3),
    rotations (P,4) wxyz) -> (means3D (P,3), cov3D_precomp (P,6)); differentiable in tetpoints, barys, scales,
    rotations.  Drop-in for models/cage_net.py:218-230.  Index tensors may be int64
    (as the reference registers them, lib/cage.py:331-337); they are converted to int32 once per call --
    pass int32 to avoid the copy.

    Optional fusion of the two activations in front of the op (models/cage_net.py:213-214):
    `delta_barys` (P,4) is added to `barys` inside the kernel (differentiable), and `scale_activation="exp"` makes
    `scales` the raw log-scales (`scaling + delta`), with exp applied on load and the chain rule in the backward.

    `canonical_gradient` may also be ONE matrix per tetrahedron, (T,3,3) = `canonical_gradient_per_tet(...)` (round 4): the
    reference gathers the same matrices per Gaussian at init (lib/cage.py:329), which makes the op stream 36 B x P per pass;
    the per-tet table is read through `tetra_id` and stays in L2."""
    if scale_activation not in (None, "exp"):
        raise ValueError(f"scale_activation must be None or 'exp', got {scale_activation!r}")
    P, T = barys.shape[0], tetras.shape[0]
    if gradient_per_tet is None:             # (T,3,3) is recognised by its length unless T == P (then say which)
        if canonical_gradient.shape[0] == T and T == P:
            import warnings
            warnings.warn("cage_deform: as many tetrahedra as Gaussians -- canonical_gradient is read per GAUSSIAN (the reference's "
                          "layout, lib/cage.py:329); pass gradient_per_tet=True if it is the per-tetrahedron table", stacklevel=2)
        gradient_per_tet = canonical_gradient.shape[0] == T and T != P
    if canonical_gradient.shape[0] != (T if gradient_per_tet else P):
        raise ValueError(f"canonical_gradient has {canonical_gradient.shape[0]} matrices, expected "
                         f"{T if gradient_per_tet else P} ({'one per tetrahedron' if gradient_per_tet else 'one per Gaussian'})")
    flags = (1 if scale_activation == "exp" else 0) | (2 if gradient_per_tet else 0)
    return _CageDeform.apply(tetpoints, _i32c(tetras), _i32c(tetra_id), barys, canonical_gradient, scales, rotations,
                             delta_barys, flags)


class _LbsCage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, template, delta, joint_mats, skin_idx, skin_w, Rh, Th):
        require_cuda(template, delta, joint_mats, skin_idx, skin_w, Rh, Th)
        template, delta, joint_mats, skin_w, Rh, Th = map(_f32c, (template, delta, joint_mats, skin_w, Rh, Th))
        V, K = skin_w.shape
        out = torch.empty((V, 3), dtype=torch.float32, device=template.device)
        check(_lib.lib().d3ga_lbs_cage_fwd(V, K, dptr(template), dptr(delta), dptr(joint_mats), dptr(skin_idx),
                                           dptr(skin_w), dptr(Rh), dptr(Th), dptr(out), stream_handle()),
              "d3ga_lbs_cage_fwd")
        ctx.save_for_backward(joint_mats, skin_idx, skin_w, Rh if Rh is not None else torch.empty(0, device=out.device))
        ctx.has_Rh = Rh is not None
        ctx.has_delta = delta is not None
        return out

    @staticmethod
    def backward(ctx, g):
        joint_mats, skin_idx, skin_w, Rh = ctx.saved_tensors
        V, K = skin_w.shape
        gd = torch.empty((V, 3), dtype=torch.float32, device=g.device)
        check(_lib.lib().d3ga_lbs_cage_bwd(V, K, dptr(joint_mats), dptr(skin_idx), dptr(skin_w),
                                           dptr(Rh if ctx.has_Rh else None), dptr(_f32c(g)), dptr(gd),
                                           stream_handle()), "d3ga_lbs_cage_bwd")
        g_t = gd if ctx.needs_input_grad[0] else None
        g_d = gd if (ctx.has_delta and ctx.needs_input_grad[1]) else None
        return g_t, g_d, None, None, None, None, None


def skeleton_matrices(bind_state, target_states):
    """(B,J,4,4) joint matrices for `lbs_cage` from Goliath-style skeleton states (translation 3 | quaternion xyzw 4 | scale 1):
    M_j = T_target,j . T_bind,j^-1 -- what lbsmodel/body_model.py:350-387 (states_to_matrix) hands to LinearBlendSkinning.skinning
    (:208-234), whose (V,8) skin_indices / skin_weights buffers are `lbs_cage`'s skin_idx / skin_w as they are.  A few dozen joints
    per pose: plain torch on whatever device the states live on.  bind_state (1,J,8) | (J,8), target_states (B,J,8) | (J,8)."""
    def split(s):
        return s[..., 0:3], s[..., 3:7] / s[..., 3:7].norm(dim=-1, keepdim=True), s[..., 7:8]

    def qmul(a, b):                                        # Hamilton product, xyzw
        av, aw, bv, bw = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
        return torch.cat([aw * bv + bw * av + torch.cross(av, bv, dim=-1), aw * bw - (av * bv).sum(-1, keepdim=True)], -1)

    def qrot(q, v):
        qv, qw = q[..., :3], q[..., 3:]
        c = torch.cross(qv, v, dim=-1)
        return v + 2.0 * (qw * c + torch.cross(qv, c, dim=-1))
    bt, bq, bs = split(bind_state)
    tt, tq, ts = split(target_states)
    bq_inv = bq * bq.new_tensor([-1.0, -1.0, -1.0, 1.0])
    q = qmul(tq, bq_inv)                                   # rotation of the composed map
    sc = ts / bs                                           # its scale
    t = tt - sc * qrot(q, bt.expand_as(tt) if bt.dim() == tt.dim() else bt)      # x -> sc R(q) (x - t_bind) + t_target
    x, y, z, w = q.unbind(-1)
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
    M = torch.zeros(q.shape[:-1] + (4, 4), dtype=q.dtype, device=q.device)
    M[..., :3, :3] = R * sc[..., None]
    M[..., :3, 3] = t
    M[..., 3, 3] = 1.0
    return M


def lbs_cage(template, delta, joint_mats, skin_idx, skin_w, Rh=None, Th=None):
    """K-sparse linear blend skinning of cage vertices: (sum_k w_k A[idx_k]) [v+delta;1], then .Rh^T + Th
    (lib/smplman.py:155-171).  Differentiable in template and delta (the deformation_field output when
    train.tet_offset_pre_lbs is on, models/cage_net.py:207-208); joint transforms are treated as constants."""
    return _LbsCage.apply(template, delta, joint_mats, _i32c(skin_idx), skin_w, Rh, Th)


class _LbsCageDeform(torch.autograd.Function):
    """lbs_cage + cage_deform as ONE autograd node (round 5): the forward is the two launches of the separate operators; the
    backward merges the corner gradients per workgroup (merge_plan) and forms dL/d(delta) in the vertex-gather launch
    (d3ga_cage_deform_bwd_merged_lbs) -- one launch instead of the gather + d3ga_lbs_cage_bwd."""

    @staticmethod
    def forward(ctx, template, delta, joint_mats, skin_idx, skin_w, Rh, Th, tetras, tetra_id, barys, canon_grad, scales, rotations,
                delta_barys, flags):
        require_cuda(template, delta, joint_mats, skin_idx, skin_w, Rh, Th, tetras, tetra_id, barys, canon_grad, scales, rotations)
        template, delta, joint_mats, skin_w, Rh, Th = map(_f32c, (template, delta, joint_mats, skin_w, Rh, Th))
        barys, canon_grad, scales, rotations, delta_barys = map(_f32c, (barys, canon_grad, scales, rotations, delta_barys))
        V, K = skin_w.shape
        P, dev = barys.shape[0], barys.device
        L = _lib.lib()
        tetpoints = torch.empty((V, 3), dtype=torch.float32, device=dev)
        check(L.d3ga_lbs_cage_fwd(V, K, dptr(template), dptr(delta), dptr(joint_mats), dptr(skin_idx), dptr(skin_w), dptr(Rh),
                                  dptr(Th), dptr(tetpoints), stream_handle()), "d3ga_lbs_cage_fwd")
        means = torch.empty((P, 3), dtype=torch.float32, device=dev)
        cov6 = torch.empty((P, 6), dtype=torch.float32, device=dev)
        check(L.d3ga_cage_deform_fwd_ex(P, dptr(tetpoints), dptr(tetras), dptr(tetra_id), dptr(barys), dptr(canon_grad),
                                        dptr(scales), dptr(rotations), dptr(delta_barys), flags, dptr(means), dptr(cov6),
                                        stream_handle()), "d3ga_cage_deform_fwd_ex")
        ctx.flags, ctx.has_dbary, ctx.has_Rh, ctx.has_delta = flags, delta_barys is not None, Rh is not None, delta is not None
        ctx.set_materialize_grads(False)
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations,
                              delta_barys if delta_barys is not None else none, joint_mats, skin_idx, skin_w,
                              Rh if Rh is not None else none)
        return means, cov6, tetpoints

    @staticmethod
    def backward(ctx, g_means, g_cov6, g_tp_extra):
        (tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations, delta_barys, joint_mats, skin_idx, skin_w,
         Rh) = ctx.saved_tensors
        P, V, K, dev = barys.shape[0], tetpoints.shape[0], skin_w.shape[1], barys.device
        g_means = torch.zeros((P, 3), device=dev) if g_means is None else _f32c(g_means)
        g_cov6 = torch.zeros((P, 6), device=dev) if g_cov6 is None else _f32c(g_cov6)
        need = ctx.needs_input_grad
        g_b = torch.empty((P, 4), dtype=torch.float32, device=dev) if (need[9] or need[13]) else None
        g_s = torch.empty((P, 3), dtype=torch.float32, device=dev) if need[11] else None
        g_r = torch.empty((P, 4), dtype=torch.float32, device=dev) if need[12] else None
        g_d = torch.empty((V, 3), dtype=torch.float32, device=dev)
        plan = merge_plan(tetras, tetra_id, V)
        partials = torch.empty((max(plan["n_segments"], 1), 3), dtype=torch.float32, device=dev)
        check(_lib.lib().d3ga_cage_deform_bwd_merged_lbs(
            P, V, dptr(tetpoints), dptr(tetras), dptr(tetra_id), dptr(barys), dptr(canon_grad), dptr(scales), dptr(rotations),
            dptr(delta_barys if ctx.has_dbary else None), ctx.flags, dptr(g_means), dptr(g_cov6), None, dptr(g_b), dptr(g_s),
            dptr(g_r), dptr(plan["item_pos"]), dptr(plan["seg_ptr"]), dptr(plan["seg_begin"]), plan["n_segments"],
            dptr(plan["vert_start"]), dptr(plan["vert_parts"]), dptr(partials), K, dptr(joint_mats), dptr(skin_idx), dptr(skin_w),
            dptr(Rh if ctx.has_Rh else None), dptr(None if g_tp_extra is None else _f32c(g_tp_extra)), dptr(g_d),
            stream_handle()), "d3ga_cage_deform_bwd_merged_lbs")
        return (g_d if need[0] else None, g_d if (ctx.has_delta and need[1]) else None, None, None, None, None, None, None, None,
                g_b if need[9] else None, None, g_s, g_r, g_b if need[13] else None, None)


def lbs_cage_deform(template, delta, joint_mats, skin_idx, skin_w, tetras, tetra_id, barys, canonical_gradient, scales, rotations,
                    delta_barys=None, scale_activation=None, gradient_per_tet=None, Rh=None, Th=None):
    """`cage_deform(lbs_cage(template, delta, joint_mats, skin_idx, skin_w, Rh, Th), tetras, ...)` as one operator (an extension
    like `renderer.render_l1`; lib/smplman.py:155-171 feeding models/cage_net.py:218-230): same outputs, same gradients, one
    launch less in the backward.  -> (means3D (P,3), cov3D_precomp (P,6), tetpoints (V,3)); the posed cage vertices are returned
    for the terms that read them directly (the FEM regulariser, lib/cage.py:349-361) -- their gradient joins the skinning backward."""
    if scale_activation not in (None, "exp"):
        raise ValueError(f"scale_activation must be None or 'exp', got {scale_activation!r}")
    P, T = barys.shape[0], tetras.shape[0]
    if gradient_per_tet is None:
        if canonical_gradient.shape[0] == T and T == P:
            import warnings
            warnings.warn("lbs_cage_deform: as many tetrahedra as Gaussians -- canonical_gradient is read per GAUSSIAN (the reference's "
                          "layout, lib/cage.py:329); pass gradient_per_tet=True if it is the per-tetrahedron table", stacklevel=2)
        gradient_per_tet = canonical_gradient.shape[0] == T and T != P
    if canonical_gradient.shape[0] != (T if gradient_per_tet else P):
        raise ValueError(f"canonical_gradient has {canonical_gradient.shape[0]} matrices, expected {T if gradient_per_tet else P}")
    flags = (1 if scale_activation == "exp" else 0) | (2 if gradient_per_tet else 0)
    return _LbsCageDeform.apply(template, delta, joint_mats, _i32c(skin_idx), skin_w, Rh, Th, _i32c(tetras), _i32c(tetra_id), barys,
                                canonical_gradient, scales, rotations, delta_barys, flags)


class _FemEnergy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tetpoints, tetras, Dn_inv):
        require_cuda(tetpoints, tetras, Dn_inv)
        tetpoints, Dn_inv = _f32c(tetpoints), _f32c(Dn_inv)
        T = tetras.shape[0]
        e = torch.empty((T,), dtype=torch.float32, device=tetpoints.device)
        check(_lib.lib().d3ga_fem_energy_fwd(T, dptr(tetpoints), dptr(tetras), dptr(Dn_inv), dptr(e), stream_handle()),
              "d3ga_fem_energy_fwd")
        ctx.save_for_backward(tetpoints, tetras, Dn_inv)
        return e

    @staticmethod
    def backward(ctx, g):
        tetpoints, tetras, Dn_inv = ctx.saved_tensors
        T, V = tetras.shape[0], tetpoints.shape[0]
        gt = torch.empty((V, 3), dtype=torch.float32, device=g.device)
        check(_lib.lib().d3ga_fem_energy_bwd(T, V, dptr(tetpoints), dptr(tetras), dptr(Dn_inv), dptr(_f32c(g)),
                                             dptr(gt), stream_handle()), "d3ga_fem_energy_bwd")
        return gt, None, None


def fem_energy(tetpoints, tetras, Dn_inv):
    """Per-tet 0.5 (det F - 1)^2 + 0.5 (tr F^T F - 3), F = Ds Dn^-1 (lib/cage.py:349-361).  Returns (T,)."""
    return _FemEnergy.apply(tetpoints, _i32c(tetras), Dn_inv)


def canonical_gradient(canon_points, tetras, tetra_id):
    """inv(Dm) per Gaussian (lib/cage.py:329); init-time, plain torch on whatever device the inputs live."""
    c = canon_points[tetras.long()][tetra_id.long()]
    Dm = torch.stack([c[:, 3] - c[:, 0], c[:, 2] - c[:, 0], c[:, 1] - c[:, 0]], dim=2)
    return torch.linalg.inv(Dm)


def canonical_gradient_per_tet(canon_points, tetras):
    """inv(Dm) per TETRAHEDRON, (T,3,3): `canonical_gradient(...)` == this[tetra_id]; accepted by cage_deform directly."""
    c = canon_points[tetras.long()]
    Dm = torch.stack([c[:, 3] - c[:, 0], c[:, 2] - c[:, 0], c[:, 1] - c[:, 0]], dim=2)
    return torch.linalg.inv(Dm)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """Axis-angle vectors (B,3) -> rotation matrices (B,3,3): R = I + sin(t) K + (1 - cos(t)) K^2 with t = |r + eps|,
    K = [r / t]x.  This is the helper `tetra_sampler.lbs.batch_rodrigues` the reference imports at lib/smplman.py:16 and
    calls on the global rotation `Rh` (lib/smplman.py:167,203); tetra-sampler is un-vendored, so the published SMPL(-X)
    convention is restated here (parity unpinned; pinned by its group properties in tests/test_abi_and_host.py).  Needs no
    asset, runs on whatever device `rot_vecs` lives (init / per-pose glue, not a hot kernel)."""
    if rot_vecs.dim() != 2 or rot_vecs.shape[1] != 3:
        raise ValueError(f"batch_rodrigues expects (B,3) axis-angle vectors, got {tuple(rot_vecs.shape)}")
    angle = torch.linalg.norm(rot_vecs + epsilon, dim=1, keepdim=True)        # (B,1); eps keeps the zero rotation finite
    k = rot_vecs / angle
    s, c = torch.sin(angle)[:, :, None], torch.cos(angle)[:, :, None]
    z = torch.zeros_like(k[:, 0])
    K = torch.stack([z, -k[:, 2], k[:, 1], k[:, 2], z, -k[:, 0], -k[:, 1], k[:, 0], z], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1.0 - c) * torch.bmm(K, K)


class SMPLlayer(torch.nn.Module):
    """Import-compatible placeholder for `tetra_sampler.body_model.SMPLlayer` (lib/smplman.py:9,68-74).

    The SMPL-X body model needs the licensed SMPL-X asset files (`config.data.smplx_model`, the joint regressor) and is
    OUT OF SCOPE here: the import succeeds so that the reference's modules load unchanged, constructing
    the layer fails with an error that says what is missing.  Everything downstream of the body model -- the per-vertex
    blend `Smplman.deform` (lib/smplman.py:155-171) -- is `d3ga_amd.cage_deform.lbs_cage` and takes the joint transforms
    any SMPL-X implementation produces."""

    def __init__(self, model_path=None, model_type="smplx", gender="neutral", use_joints=True, regressor_path=None, **kw):
        super().__init__()
        raise NotImplementedError(
            "tetra_sampler.body_model.SMPLlayer is not provided by d3ga_amd: it requires the licensed SMPL-X model files "
            f"(model_path={model_path!r}, regressor_path={regressor_path!r}) and the un-vendored tetra-sampler package "
            "(github.com/Zielon/sampler).  Install that package for the body model; d3ga_amd replaces the deform / "
            "rasterize path only (lbs_cage takes the joint transforms A and blend offsets any SMPL-X layer returns).")
