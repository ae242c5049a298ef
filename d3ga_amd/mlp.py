"""Field networks of the reference (models/mlp.py) on the MI355X matrix cores.

Every field of the reference is the same trunk -- ``z -> [Linear(128) + leaky_relu(0.1)] x (1 + n_layers) -> Linear`` --
with ``z = [pose.expand(P, -1) | per-row features]`` (models/mlp.py:58-69, 94-105).  Here
  * the broadcast part of the first layer is folded into its bias once per call (``W0[:, :n_pose] @ pose + b0``: a
    128-vector instead of 98 of the 109 input columns for every one of the P rows),
  * every dense layer is one launch of ``d3ga_mlp_linear`` (f32-equivalent arithmetic on the bf16 matrix cores -- an exact
    3-way split of every operand, six products, f32 accumulate: DESIGN.md sec. 4.5 -- with bias + leaky_relu fused),
  * the whole trunk is ONE autograd node (``_Chain``): the forward keeps one sign bit per activation, the input-gradient
    GEMM of a layer applies the leaky_relu derivative of the layer below from those bits in its epilogue, so every
    pre-activation gradient is written once and no activation is re-read for masking,
  * weight and bias gradients (``dPre^T @ X``, a reduction over all rows) come from ``d3ga_mlp_wgrad_acc`` into one
    zero-filled buffer per chain.
Modules keep the reference's parameter names (``network.{i}.weight/bias``, ``output.weight/bias``): state dicts
interchange.  GPU tensors only.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import f32c16, check, dptr, require_cuda, stream_handle

_panels = {}


def _panel(weight, transpose):
    """The layer's weights split into bf16 planes in the kernel's operand order (d3ga_mlp_pack_weights).  transpose=True:
    forward (weight is (N,K) as in nn.Linear: weight of input k for output n = weight[n][k]); False: the input-gradient
    GEMM, which contracts over the layer's outputs (input n_layer, output k_layer: weight[n_layer][k_layer]).  Cached per
    (storage, version): an optimizer step bumps the version and the next call re-packs (one tiny launch)."""
    key = (_lib.replay_epoch[0], weight.data_ptr(), weight._version, tuple(weight.shape), tuple(weight.stride()), transpose)
    # Under stream capture the pack itself must be IN the graph: a cache hit would bake today's panel into a graph that is
    # replayed after an optimizer has changed the weights in place (no version check can run at replay time).
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = None if capturing else _panels.get(key)
    if hit is None:
        if len(_panels) > 64 and not capturing:
            _panels.clear()
        w = weight.detach()
        if w.dtype != torch.float32:
            w = w.float()
        n_layer, k_layer = w.shape
        s_n, s_k = w.stride()
        K, N, ld_k, ld_n = (k_layer, n_layer, s_k, s_n) if transpose else (n_layer, k_layer, s_n, s_k)
        nbytes = _lib.lib().d3ga_mlp_panel_bytes(K, N)
        if nbytes < 0:
            raise ValueError(f"linear_act: layer {n_layer}x{k_layer} is outside the kernel's range (K, N <= 128)")
        p = torch.empty(nbytes // 4, dtype=torch.int32, device=w.device)
        check(_lib.lib().d3ga_mlp_pack_weights(K, N, dptr(w), ld_k, ld_n, dptr(p), stream_handle()), "d3ga_mlp_pack_weights")
        # holds the weight's STORAGE alive (its address cannot be recycled) through detached aliases: a cached view that
        # still carried the grad_fn of an earlier forward made PyTorch-ROCm 2.10 crash in capture_end() of a later capture
        hit = (p, weight.detach(), w)
        if not capturing:
            _panels[key] = hit
    return hit[0]


_chain_panels = {}
_FUSED = {"enabled": __import__("os").environ.get("D3GA_MLP_FUSED", "1") != "0"}


def set_fused_forward(enabled):
    """A/B switch: one launch per trunk (d3ga_mlp_chain_fwd, the default) or one per layer (d3ga_mlp_linear)."""
    _FUSED["enabled"] = bool(enabled)


def _chain_panel(weight, transposed=False):
    """The layer's weights in the operand order of the fused trunk kernel (d3ga_mlp_pack_chain: the k order carries the
    feature permutation that lets a layer's output registers be the next layer's operand as they are) + 512 bytes of room for
    the bias, which every call writes itself.  transposed: the panel of W^T (the backward's input-gradient chain).  Cached
    per weight version like `_panel`."""
    key = (_lib.replay_epoch[0], weight.data_ptr(), weight._version, tuple(weight.shape), tuple(weight.stride()), bool(transposed))
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()      # (see _panel: the pack goes into the graph)
    hit = None if capturing else _chain_panels.get(key)
    if hit is None:
        if len(_chain_panels) > 64 and not capturing:
            _chain_panels.clear()
        w = weight.detach()
        if w.dtype != torch.float32:
            w = w.float()
        N, K = w.shape
        s_n, s_k = w.stride()
        if transposed:                                     # y = x W: inputs run over W's rows, outputs over its columns
            N, K, s_n, s_k = K, N, s_k, s_n
        nbytes = _lib.lib().d3ga_mlp_chain_panel_bytes(K, N)
        p = torch.empty(nbytes // 4, dtype=torch.int32, device=w.device)
        check(_lib.lib().d3ga_mlp_pack_chain(K, N, dptr(w), s_k, s_n, dptr(p), stream_handle()), "d3ga_mlp_pack_chain")
        hit = (p, weight.detach(), w)              # (detached aliases keep the storage alive: see _panel)
        if not capturing:
            _chain_panels[key] = hit
    return hit[0]


def _chain_shapes_ok(P, k0, widths):
    """What d3ga_mlp_chain_fwd is built for: >= 2 layers, all but the last 128 wide, k0 <= 128, last <= 128."""
    return (_FUSED["enabled"] and 2 <= len(widths) <= 8 and 0 < P < (1 << 23) and 1 <= k0 <= 128
            and all(n == 128 for n in widths[:-1]) and 1 <= widths[-1] <= 128)


def _chain_run(h, panels, dims, biases, slopes, want_signs, masks=None, mask_slopes=None):
    """ONE launch for a chain of layers: dims = [(K, N), ...] -> (outputs of every layer, sign words per layer or None)."""
    L, P, dev = len(panels), h.shape[0], h.device
    Ks = (ctypes.c_int32 * L)(*[k for k, _ in dims])
    Ns = (ctypes.c_int32 * L)(*[n for _, n in dims])
    bs = [None if b is None else b.detach().float().contiguous() for b in biases]
    outs = [torch.empty((P, n), dtype=torch.float32, device=dev) for _, n in dims]
    signs = [torch.empty((P, (n + 31) // 32), dtype=torch.int32, device=dev) if ws else None for (_, n), ws in zip(dims, want_signs)]
    vp = ctypes.c_void_p
    arr = lambda ts: (vp * L)(*[dptr(t) for t in ts])
    fl = lambda vs: (ctypes.c_float * L)(*[float(v) for v in vs])
    check(_lib.lib().d3ga_mlp_chain_fwd(P, h.shape[1], dptr(h), L, Ks, Ns, arr(panels), arr(bs), fl(slopes), arr(outs), arr(signs),
                                        None if masks is None else arr(masks), None if mask_slopes is None else fl(mask_slopes),
                                        stream_handle()), "d3ga_mlp_chain_fwd")
    return outs, signs


def _chain_forward(h, weights, biases, slopes, track):
    """All layers of a trunk in ONE launch: -> (outputs of every layer, sign words of every layer or None)."""
    return _chain_run(h, [_chain_panel(w) for w in weights], [(w.shape[1], w.shape[0]) for w in weights], biases, slopes,
                      [track and sl != 1.0 for sl in slopes])


def _linear(x, panel, bias, slope, n_out, want_sign=False, mask_bits=None, mask_slope=1.0):
    P, K = x.shape
    y = torch.empty((P, n_out), dtype=torch.float32, device=x.device)
    sign = torch.empty((P, (n_out + 31) // 32), dtype=torch.int32, device=x.device) if want_sign else None
    check(_lib.lib().d3ga_mlp_linear(P, K, n_out, dptr(x), dptr(panel), dptr(bias), float(slope), dptr(sign), dptr(mask_bits),
                                     float(mask_slope), dptr(y), stream_handle()), "d3ga_mlp_linear")
    return y, sign


def _chain_fwd_impl(x, slopes, weights, biases, track):
    """The forward of a trunk (no autograd): -> (h_L, [h_0 .. h_L], sign words per layer or None).  One launch when the shapes
    allow (d3ga_mlp_chain_fwd), else one per layer."""
    require_cuda(x, *weights)
    h = f32c16(x)
    acts, signs = [h], []
    fused = (_chain_shapes_ok(h.shape[0], h.shape[1], [w.shape[0] for w in weights])
             and all(weights[i].shape[1] == (h.shape[1] if i == 0 else weights[i - 1].shape[0]) for i in range(len(weights))))
    if fused:                                              # the whole trunk in one launch: activations stay in registers
        outs, signs = _chain_forward(h, weights, biases, slopes, track)
        acts += outs
        h = outs[-1]
        weights_loop = ()
    else:
        weights_loop = zip(weights, biases, slopes)
    for w, b, slope in weights_loop:
        N, K = w.shape
        if h.shape[1] != K or K > 128 or N > 128:
            raise ValueError(f"linear_act: x (P,{h.shape[1]}) weight {tuple(w.shape)}: need matching K <= 128 and N <= 128")
        h, sign = _linear(h, _panel(w, True), None if b is None else b.float().contiguous(), slope, N,
                          want_sign=track and slope != 1.0)
        acts.append(h)
        signs.append(sign)
    return h, acts, signs


def _chain_bwd_impl(dy, acts, weights, signs, slopes, need, want_dx):
    """The backward of a trunk: need[i] = (weight gradient wanted, bias gradient wanted) -> (dx or None, [dW_0, db_0, dW_1, ...]).
    The chain is walked once: the input-gradient GEMM of layer i multiplies by the leaky_relu derivative of layer i-1 in its
    epilogue, so each layer's pre-activation gradient is written exactly once and feeds both its weight gradient and the next GEMM."""
    L = len(weights)
    dpre = f32c16(dy)
    if slopes[-1] != 1.0:                                 # a chain that ENDS in an activation (not the fields' case)
        dpre = dpre * torch.where(acts[L] > 0, 1.0, slopes[-1])
    grads = [None] * (2 * L)
    dx = None
    # all weight / bias gradients of the chain live in ONE zero-filled buffer (one fill instead of two memsets a layer)
    sizes = [(w.numel() if nw or nbias else 0, w.shape[0] if nbias else 0) for w, (nw, nbias) in zip(weights, need)]
    flat = torch.zeros(sum(a + b for a, b in sizes), dtype=torch.float32, device=dpre.device)
    offs, o = [], 0
    for a, b in sizes:
        offs.append((o, o + a))
        o += a + b
    # the input-gradient chain dPre_{L-1} = dy -> dPre_{L-2} -> ... -> dPre_0 [-> dx]: one launch when the shapes allow
    # (the same kernel as the forward: transposed weights, the sign words of the forward as masks, no bias)
    dpres = [None] * L
    dpres[L - 1] = dpre
    chain_layers = list(range(L - 1, 0, -1)) + ([0] if want_dx else [])       # layer i maps dPre_i -> dPre_{i-1} (or dx)
    widths = [weights[i].shape[1] for i in chain_layers]
    if L >= 2 and _chain_shapes_ok(dpre.shape[0], dpre.shape[1], widths):
        masks = [signs[i - 1] if (i > 0 and slopes[i - 1] != 1.0) else None for i in chain_layers]
        mslopes = [slopes[i - 1] if i > 0 else 1.0 for i in chain_layers]
        outs, _ = _chain_run(dpre, [_chain_panel(weights[i], True) for i in chain_layers],
                             [(weights[i].shape[0], weights[i].shape[1]) for i in chain_layers], [None] * len(chain_layers),
                             [1.0] * len(chain_layers), [False] * len(chain_layers), masks, mslopes)
        for i, o in zip(chain_layers, outs):
            if i > 0:
                dpres[i - 1] = o
            else:
                dx = o
        fused_bwd = True
    else:
        fused_bwd = False
    for i in range(L - 1, -1, -1):
        w, x_in = weights[i], acts[i]
        N, K = w.shape
        need_w, need_b = need[i]
        dpre = dpres[i]
        if need_w or need_b:                              # dW += dPre^T X (a reduction over all rows), db += column sums
            dw = flat[offs[i][0]:offs[i][1]].view(N, K)
            db = flat[offs[i][1]:offs[i][1] + N] if need_b else None
            check(_lib.lib().d3ga_mlp_wgrad_acc(dpre.shape[0], N, K, dptr(dpre), dptr(x_in), dptr(dw), dptr(db),
                                                stream_handle()), "d3ga_mlp_wgrad_acc")
            grads[2 * i], grads[2 * i + 1] = (dw if need_w else None), db
        if fused_bwd:
            continue
        if i > 0:                                         # dPre of the layer below: (dPre W) (.) act'_{i-1}(h_i)
            below = slopes[i - 1]
            dpres[i - 1] = _linear(dpre, _panel(w, False), None, 1.0, K, mask_bits=signs[i - 1] if below != 1.0 else None,
                                   mask_slope=below)[0]
        elif want_dx:
            dx = _linear(dpre, _panel(w, False), None, 1.0, K)[0]
    return dx, grads


class _Chain(torch.autograd.Function):
    """h_{i+1} = act_i(h_i @ W_i.T + b_i), act(y) = y if y > 0 else slope_i * y (slope 1: none), i = 0..L-1, h_0 = x.
    apply(x, slopes, W_0, b_0, W_1, b_1, ...) -> h_L.  ONE autograd node for the whole trunk (_chain_fwd_impl / _chain_bwd_impl)."""

    @staticmethod
    def forward(ctx, x, slopes, *wb):
        weights, biases = wb[0::2], wb[1::2]
        track = any(ctx.needs_input_grad)                      # (grad mode is off inside forward: ask the node instead)
        h, acts, signs = _chain_fwd_impl(x, slopes, weights, biases, track)
        ctx.slopes = tuple(float(v) for v in slopes)
        ctx.n_layers = len(weights)
        ctx.has_bias = tuple(b is not None for b in biases)
        ctx.signs = signs                                      # int32 bit words: not differentiable, kept on the ctx
        ctx.save_for_backward(*acts, *weights)
        return h

    @staticmethod
    def backward(ctx, dy):
        L = ctx.n_layers
        acts, weights = ctx.saved_tensors[:L + 1], ctx.saved_tensors[L + 1:]
        need = [(ctx.needs_input_grad[2 + 2 * i], ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]) for i in range(L)]
        dx, grads = _chain_bwd_impl(dy, acts, weights, ctx.signs, ctx.slopes, need, bool(ctx.needs_input_grad[0]))
        return (dx, None, *grads)


def _merge_ranges(ranges):
    out = []
    for a, b in ranges:
        if out and out[-1][1] == a:
            out[-1] = (out[-1][0], b)
        else:
            out.append((a, b))
    return tuple(out)


def _take_columns(w, ranges):
    """Columns of w in `ranges` as one (N, sum of widths) tensor: a view for one range (the packed-panel cache then follows the
    parameter's version), one small cat otherwise."""
    return w[:, ranges[0][0]:ranges[0][1]] if len(ranges) == 1 else torch.cat([w[:, a:b] for a, b in ranges], dim=1)


class _FieldTrunk(torch.autograd.Function):
    """A field's trunk with the reference's input layout z = [column groups, some per row, some broadcast] as ONE autograd node.
    apply(x_rows (P, per-row columns), bc (broadcast columns, 1-D, or None), layout, slopes, W_0, b_0, W_1, b_1, ...) with
    layout = (row_ranges, bc_ranges): the column ranges of W_0 that meet x_rows / bc, each in z's order.
    The broadcast columns are folded into the first bias (b_0 + W_0[:, bc] . bc: one addmv) and the first layer's weight gradient
    is assembled in ONE buffer here -- its per-row columns from the trunk's weight gradient, its broadcast columns as the outer
    product of the bias gradient and bc.  (Round 5: as separate autograd ops -- two column slices of W_0, F.linear, the trunk -- the
    same arithmetic cost two zero fills, two strided copies, an add and three hipBLASLt launches of 8-14 us on a 128 x 98
    matrix per field and step: ~150 us of ~30 tiny launches in the colour step.)"""

    @staticmethod
    def forward(ctx, x_rows, bc, layout, slopes, *wb):
        row_ranges, bc_ranges = layout
        first_w, first_b = wb[0], wb[1]
        w_row = _take_columns(first_w, row_ranges)
        w_bc = _take_columns(first_w, bc_ranges) if bc_ranges else None
        if w_bc is not None:
            bcv = bc.detach().reshape(-1).float()
            bias0 = torch.addmv(first_b, w_bc, bcv) if first_b is not None else torch.mv(w_bc, bcv)
        else:
            bcv, bias0 = None, first_b
        weights, biases = (w_row,) + tuple(wb[2::2]), (bias0,) + tuple(wb[3::2])
        track = any(ctx.needs_input_grad)
        h, acts, signs = _chain_fwd_impl(x_rows, slopes, weights, biases, track)
        ctx.slopes = tuple(float(v) for v in slopes)
        ctx.layout = layout
        ctx.n_layers = len(weights)
        ctx.has_bias = tuple(b is not None for b in wb[1::2])
        ctx.signs = signs                                      # int32 bit words: not differentiable, kept on the ctx
        ctx.has_bc = w_bc is not None
        # everything else through save_for_backward -- the output h among the activations: kept as a plain attribute it would close
        # a reference cycle (ctx -> h -> grad_fn -> ctx) that keeps a capture's tensors alive past capture_end()
        extras = (w_row, w_bc, bcv) if ctx.has_bc else (w_row,)
        ctx.save_for_backward(*acts, *wb[0::2], *extras)
        return h

    @staticmethod
    def backward(ctx, dy):
        L = ctx.n_layers
        saved = ctx.saved_tensors
        acts, params, extras = saved[:L + 1], saved[L + 1:2 * L + 1], saved[2 * L + 1:]
        first_w, w_row = params[0], extras[0]
        w_bc, bcv = (extras[1], extras[2]) if ctx.has_bc else (None, None)
        weights = (w_row,) + tuple(params[1:])
        nig = ctx.needs_input_grad
        need_x, need_bc = bool(nig[0]), bool(nig[1]) and ctx.has_bc
        need_w0, need_b0 = bool(nig[4]), ctx.has_bias[0] and bool(nig[5])
        # the folded bias' gradient also feeds W_0's broadcast columns and d(bc)
        want_db0 = need_b0 or need_bc or (need_w0 and ctx.has_bc)
        need = [(need_w0, want_db0)] + [(bool(nig[4 + 2 * i]), ctx.has_bias[i] and bool(nig[5 + 2 * i])) for i in range(1, L)]
        dx, grads = _chain_bwd_impl(dy, acts, weights, ctx.signs, ctx.slopes, need, need_x)
        dw_row, db0 = grads[0], grads[1]
        g_w0 = None
        if need_w0:
            row_ranges, bc_ranges = ctx.layout
            g_w0 = torch.empty_like(first_w, dtype=torch.float32, memory_format=torch.contiguous_format)
            o = 0
            for a, b in row_ranges:
                g_w0[:, a:b].copy_(dw_row[:, o:o + b - a])
                o += b - a
            o = 0
            for a, b in bc_ranges:
                torch.mul(db0[:, None], bcv[None, o:o + b - a], out=g_w0[:, a:b])
                o += b - a
        g_bc = torch.mv(w_bc.t(), db0) if need_bc else None
        grads[0], grads[1] = g_w0, (db0 if need_b0 else None)
        return (dx, g_bc, None, None, *grads)


def linear_act(x, weight, bias=None, negative_slope=1.0):
    """``leaky_relu(F.linear(x, weight, bias), negative_slope)`` in one launch (negative_slope=1: no activation)."""
    return _Chain.apply(x, (negative_slope,), weight, bias)


def mlp_chain(x, layers, slopes):
    """The whole trunk as ONE autograd node: layers = [(weight, bias), ...], slopes[i] the leaky_relu slope after layer i."""
    flat = [t for wb in layers for t in wb]
    return _Chain.apply(x, tuple(slopes), *flat)


_ACT = {"none": 0, "tanh": 1, "sigmoid": 2}


class _Heads(torch.autograd.Function):
    """pred (P,N) -> its column groups, each contiguous and through its activation, in one launch each way
    (d3ga_field_heads_*).  spec = ((width, act, param), ...): act "none", "tanh" (param * tanh(x)) or "sigmoid"
    (sigmoid(x + param))."""

    @staticmethod
    def forward(ctx, pred, spec):
        require_cuda(pred)
        pred = pred.float().contiguous()
        P, N = pred.shape
        n = len(spec)
        ctx.c_spec = ((ctypes.c_int32 * n)(*[w for w, _, _ in spec]), (ctypes.c_int32 * n)(*[_ACT[a] for _, a, _ in spec]),
                      (ctypes.c_float * n)(*[float(v) for _, _, v in spec]))
        out = torch.empty(P * N, dtype=torch.float32, device=pred.device)
        check(_lib.lib().d3ga_field_heads_fwd(P, N, n, *ctx.c_spec, dptr(pred), dptr(out), stream_handle()), "d3ga_field_heads_fwd")
        heads, off = [], 0
        for w, _, _ in spec:
            heads.append(out[P * off:P * (off + w)].view(P, w))
            off += w
        ctx.shape, ctx.n = (P, N), n
        ctx.save_for_backward(out)
        return tuple(heads)

    @staticmethod
    def backward(ctx, *grads):
        (out,) = ctx.saved_tensors
        P, N = ctx.shape
        gs = [None if g is None else g.float().contiguous() for g in grads] + [None] * (4 - ctx.n)
        d_pred = torch.empty((P, N), dtype=torch.float32, device=out.device)
        check(_lib.lib().d3ga_field_heads_bwd(P, N, ctx.n, *ctx.c_spec, dptr(out), dptr(gs[0]), dptr(gs[1]), dptr(gs[2]),
                                              dptr(gs[3]), dptr(d_pred), stream_handle()), "d3ga_field_heads_bwd")
        return d_pred, None


def field_heads(pred, spec):
    return _Heads.apply(pred, tuple(spec))


class FieldMLP(nn.Module):
    """The trunk shared by the reference's fields (models/mlp.py:50-69): ``n_layers + 1`` hidden layers of ``n_nodes`` with
    leaky_relu(0.1) and a linear head; kaiming-leaky init, head weights scaled by 0.33 (models/mlp.py:17-20,55-57)."""

    def __init__(self, n_input, n_output, n_nodes=128, n_layers=3):
        super().__init__()
        self.network = nn.ModuleList([nn.Linear(n_input, n_nodes)] + [nn.Linear(n_nodes, n_nodes) for _ in range(n_layers)])
        self.output = nn.Linear(n_nodes, n_output)
        self._col_index = {}             # forward_parts: column layout -> (per-row columns, broadcast columns) index tensors
        with torch.no_grad():
            self.output.weight *= 0.33
            for layer in self.network:
                nn.init.kaiming_normal_(layer.weight, a=0.1, mode="fan_in", nonlinearity="leaky_relu")

    def forward_parts(self, parts):
        """General column layout: `parts` lists z's column groups in the reference's order, each a (P, w) per-row tensor or a
        1-D broadcast vector (ColorField mixes both kinds: models/mlp.py:208-226).  Broadcast groups are folded into the
        first layer's bias, per-row groups are concatenated and meet the matching columns of the first weight (_FieldTrunk)."""
        rows = [t for t in parts if t.dim() != 1]
        bcs = [t for t in parts if t.dim() == 1]
        sig = tuple((t.dim() == 1, t.shape[-1]) for t in parts)
        return self.forward_layout(torch.cat(rows, dim=1) if len(rows) > 1 else rows[0], bcs, sig)

    def forward_layout(self, x, bcs, sig):
        """x = the per-row column groups already side by side (P, sum of their widths), bcs = the broadcast vectors, sig = z's
        column groups in order as (is_broadcast, width)."""
        first = self.network[0]
        layout = self._col_index.get(sig)
        if layout is None:               # column ranges of the two kinds, adjacent ranges merged, built once per layout
            row_r, bc_r, c = [], [], 0
            for is_bc, w in sig:
                if w:
                    (bc_r if is_bc else row_r).append((c, c + w))
                c += w
            if c != first.weight.shape[1]:
                raise ValueError(f"field input has {c} columns, the first layer expects {first.weight.shape[1]}")
            if not row_r:
                raise ValueError("a field needs at least one per-row column group")
            layout = (_merge_ranges(row_r), _merge_ranges(bc_r))
            self._col_index[sig] = layout
        bcs = [t for t in bcs if t.numel()]
        bc = (torch.cat(bcs) if len(bcs) > 1 else bcs[0]) if bcs else None
        return self._trunk(x, bc, layout)

    def _trunk(self, x, bc, layout):
        layers = [(l.weight, l.bias) for l in self.network] + [(self.output.weight, self.output.bias)]
        flat = [t for wb in layers for t in wb]
        return _FieldTrunk.apply(x, bc, layout, tuple([0.1] * len(self.network) + [1.0]), *flat)

    def forward(self, row_feats, broadcast):
        """z = [broadcast.expand(P, -1) | row_feats] (the reference's column order) -> (P, n_output)."""
        nb, K = broadcast.numel(), self.network[0].weight.shape[1]
        if nb + row_feats.shape[1] != K:
            raise ValueError(f"field input has {nb + row_feats.shape[1]} columns, the first layer expects {K}")
        return self._trunk(row_feats, broadcast if nb else None, (((nb, K),), (((0, nb),) if nb else ())))


class CanonicalField(FieldMLP):
    """models/mlp.py:74-110.  forward(barys, rots, scales, pose) -> (tanh(.)*scale_bary (P,4), (P,4), (P,3)) with
    z = [pose | rots | scales | barys]."""

    def __init__(self, n_cond=98, n_nodes=128, n_layers=3, scale_bary=0.25, bary_size=4):
        super().__init__(n_cond + 4 + 3 + bary_size, 4 + 3 + bary_size, n_nodes, n_layers)
        self.scale_bary, self.bary_size = scale_bary, bary_size

    def forward(self, barys, rots, scales, pose):
        pred = super().forward(torch.cat([rots, scales, barys], dim=1), pose)
        s = self.bary_size
        return field_heads(pred, ((s, "tanh", self.scale_bary), (4, "none", 0.0), (pred.shape[1] - s - 4, "none", 0.0)))


def embed(x, multires=7):
    """utils/pos_encoder.py get_embedder(7): [x, sin(2^i x), cos(2^i x)] -> 45 columns."""
    out = [x]
    for i in range(multires):
        out += [torch.sin(x * float(2 ** i)), torch.cos(x * float(2 ** i))]
    return torch.cat(out, -1)


_embed_cache = {}


def embed_const(x, multires=7):
    """embed() of an input that does not require a gradient (the canonical cage vertices, models/cage_net.py:197: the same tensor
    every step): evaluated once per (storage, version) -- 22 element-wise launches per step otherwise, each at the ~2.6 us floor
    of a launch.  Not used under stream capture (a cached result would be baked into the graph)."""
    if x.requires_grad or (x.is_cuda and torch.cuda.is_current_stream_capturing()):
        return embed(x, multires)
    key = (_lib.replay_epoch[0], x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), x.dtype, x.device, multires)
    hit = _embed_cache.get(key)
    if hit is None:
        if len(_embed_cache) > 16:
            _embed_cache.clear()
        hit = (embed(x, multires), x)              # (x kept alive: its address cannot be recycled under the same key)
        _embed_cache[key] = hit
    return hit[0]


class DeformationField(FieldMLP):
    """models/mlp.py:39-71.  forward(canonical (V,3), pose) -> tanh(.) * scaling with z = [pose | embed_7(canonical)]."""

    def __init__(self, n_cond=98, n_nodes=128, n_layers=3, scaling=0.2):
        super().__init__(n_cond + 45, 3, n_nodes, n_layers)
        self.scaling = scaling

    def set_constant_input(self, canonical):
        """The caller's promise that `canonical` (the canonical cage vertices: a buffer of the model, models/cage_net.py:197) never
        changes: its embedding is evaluated once, here, and a forward that is handed this very tensor uses it -- also under stream
        capture, where embed_const() must not answer from its cache (the reference's ShadowDecoder keeps its embedded template the
        same way).  None: forget it."""
        self._const_src = canonical
        self._const_ver = None if canonical is None else canonical._version
        self._const_emb = None if canonical is None else embed(canonical.detach())

    def forward(self, canonical, pose):
        if canonical is getattr(self, "_const_src", None) and not canonical.requires_grad:
            if canonical._version != self._const_ver:      # written in place since (load_state_dict, buffer.copy_): same object, new values
                if canonical.is_cuda and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("DeformationField: the tensor given to set_constant_input() was modified in place; call "
                                       "set_constant_input() again before capturing")
                self.set_constant_input(canonical)
            z = self._const_emb
        else:
            z = embed_const(canonical)
        return field_heads(super().forward(z, pose), ((3, "tanh", self.scaling),))[0]


class ShadowDecoder(FieldMLP):
    """models/mlp.py:262-297.  forward(pose) -> sigmoid(.) (V,1) with z = [pose[6:] | embed_7(template)]; the embedded
    template is a constant of the module, as in the reference."""

    def __init__(self, template, n_cond=98, n_nodes=128, n_layers=3):
        super().__init__(n_cond + 45, 1, n_nodes, n_layers)
        self.register_buffer("embedded_template", embed(template), persistent=False)

    def forward(self, pose):
        return torch.sigmoid(super().forward(self.embedded_template, pose[6:]))


class FaceDecoder(FieldMLP):
    """models/mlp.py:235-259.  forward(kpt (n,3)) -> (n_output,): the flattened keypoints through the trunk (one row)."""

    def __init__(self, n_valid_kpts, n_output=128, n_nodes=128, n_layers=3):
        super().__init__(n_valid_kpts * 3, n_output, n_nodes, n_layers)

    def forward(self, kpt):
        z = kpt.reshape(1, -1)
        return super().forward(z, z.new_zeros(0))[0]


class _Sh4Encoding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d):
        require_cuda(d)
        d = f32c16(d)
        P = d.shape[0]
        enc = torch.empty((P, 16), dtype=torch.float32, device=d.device)
        check(_lib.lib().d3ga_sh4_encoding_fwd(P, dptr(d), dptr(enc), stream_handle()), "d3ga_sh4_encoding_fwd")
        ctx.save_for_backward(d)
        return enc

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        gd = torch.empty_like(d)
        check(_lib.lib().d3ga_sh4_encoding_bwd(d.shape[0], dptr(d), dptr(f32c16(g)), dptr(gd), stream_handle()),
              "d3ga_sh4_encoding_bwd")
        return gd


class _ColorRows(torch.autograd.Function):
    """x (P, 16 + F) = [sh4_direction_encoding(view_dir) | feats]: ColorField's per-row input columns written in ONE pass
    (d3ga_color_rows_fwd) instead of the encoding call + torch.cat (the 64-byte encoding rows and 2 x 320 B per Gaussian moved
    again), and its input gradient split in one pass (d3ga_color_rows_bwd) instead of two strided copies."""

    @staticmethod
    def forward(ctx, d, feats):
        require_cuda(d, feats)
        d, feats = f32c16(d), f32c16(feats)
        P, F_ = feats.shape
        x = torch.empty((P, 16 + F_), dtype=torch.float32, device=d.device)
        check(_lib.lib().d3ga_color_rows_fwd(P, F_, dptr(d), dptr(feats), dptr(x), stream_handle()), "d3ga_color_rows_fwd")
        ctx.save_for_backward(d)
        ctx.n_feat = F_
        return x

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        P, F_ = d.shape[0], ctx.n_feat
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gf = torch.empty((P, F_), dtype=torch.float32, device=d.device) if ctx.needs_input_grad[1] else None
        check(_lib.lib().d3ga_color_rows_bwd(P, F_, dptr(d), dptr(f32c16(g)), dptr(gd), dptr(gf), stream_handle()), "d3ga_color_rows_bwd")
        return gd, gf


def sh4_direction_encoding(d):
    """Stand-in for tiny-cuda-nn's degree-4 `SphericalHarmonics` direction encoding (16 outputs, models/mlp.py:166-179):
    x = 2 d - 1, then the real SH polynomials of degree < 4 (constants of utils/sh_utils.py:7-24); one HIP kernel each way
    (csrc/encoding.hip).  tiny-cuda-nn is un-vendored: parity of THIS function is unpinned; everything
    around it is pinned."""
    if d.dim() != 2 or d.shape[1] != 3:
        raise ValueError("sh4_direction_encoding: (P,3) directions")
    return _Sh4Encoding.apply(d)


class _ViewDirs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, campos):
        require_cuda(means3D, campos)
        means3D, campos = means3D.float().contiguous(), campos.float().contiguous()
        v = torch.empty_like(means3D)
        check(_lib.lib().d3ga_view_dirs_fwd(means3D.shape[0], dptr(means3D), dptr(campos), dptr(v), stream_handle()),
              "d3ga_view_dirs_fwd")
        ctx.save_for_backward(means3D, campos)
        return v

    @staticmethod
    def backward(ctx, g):
        means3D, campos = ctx.saved_tensors
        gm = torch.empty_like(means3D)
        check(_lib.lib().d3ga_view_dirs_bwd(means3D.shape[0], dptr(means3D), dptr(campos), dptr(g.float().contiguous()),
                                            dptr(gm), stream_handle()), "d3ga_view_dirs_bwd")
        return gm, None


def view_directions(means3D, camera_center):
    """models/cage_net.py:233-235: unit vectors from the (detached) camera centre to every Gaussian, one kernel each way."""
    return _ViewDirs.apply(means3D, camera_center.detach().reshape(3))


class ColorField(FieldMLP):
    """models/mlp.py:152-232 with use_pose and use_view_enc (configs/actorshq_actor02.yml:100-105): the per-Gaussian colour
    network of the reference's main configuration (`use_shs: false`).
    forward(shs (P,n_features), pose, view_dir (P,3), frame_encoding=None, camera_encoding=None, shadow=None)
      -> (sigmoid(pred[:, :3]), sigmoid(0.1 + pred[:, 3:4])),  z = [enc(view_dir) | pose | shadow | camera | frame | shs]."""

    def __init__(self, n_features=64, n_cond=98, frame_dims=32, camera_dims=0, n_nodes=128, n_layers=4,
                 direction_encoding=sh4_direction_encoding, n_view_enc=16, shadow_dims=0):
        # (the reference sizes its first layer without the shadow column, models/mlp.py:181-193, so a forward WITH shadow
        # fails there; shadow_dims = 1 makes room for it)
        super().__init__(n_cond + n_features + n_view_enc + frame_dims + camera_dims + shadow_dims, 3 + 1, n_nodes, n_layers)
        self.direction_encoding = direction_encoding

    def forward(self, shs, pose, view_dir, frame_encoding=None, camera_encoding=None, shadow=None):
        if (shadow is None and self.direction_encoding is sh4_direction_encoding and shs.dim() == 2 and shs.shape[1] % 4 == 0
                and shs.shape[0] > 0 and view_dir.dim() == 2 and view_dir.shape[1] == 3):
            # the two per-row groups (first and last columns of z) as one buffer, written in one pass; same layout as below
            bcs = [t.reshape(-1) for t in (pose, camera_encoding, frame_encoding) if t is not None]
            sig = [(False, 16)] + [(True, t.numel()) for t in bcs] + [(False, shs.shape[1])]
            x = _ColorRows.apply(view_dir, shs)
            return field_heads(self.forward_layout(x, bcs, tuple(sig)), ((3, "sigmoid", 0.0), (1, "sigmoid", 0.1)))
        parts = [self.direction_encoding(view_dir), pose.reshape(-1)]
        if shadow is not None:
            parts.append(shadow)
        if camera_encoding is not None:
            parts.append(camera_encoding.reshape(-1))
        if frame_encoding is not None:
            parts.append(frame_encoding.reshape(-1))
        parts.append(shs)
        return field_heads(self.forward_parts(parts), ((3, "sigmoid", 0.0), (1, "sigmoid", 0.1)))
