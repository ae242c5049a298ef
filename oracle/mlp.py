"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's field networks (models/mlp.py, utils/pos_encoder.py).

Pinned: tests/golden/field_cases.npz holds weights, inputs, outputs and autograd gradients of the reference's own
`CanonicalField`, `DeformationField`, `ShadowDecoder`, `FaceDecoder` and `ColorField` (tools/gen_golden.py imports
/root/reference/models/mlp.py in the build container).  PARITY UNPINNED for one function: `sh4_direction_encoding`, the
stand-in for tiny-cuda-nn's degree-4 SphericalHarmonics encoding inside ColorField (tiny-cuda-nn is un-vendored,
install.sh:14; restated from its published definition) -- the ColorField golden was generated with this stand-in plugged
into the reference module, so everything around the encoding is pinned.
"""
import torch
import torch.nn.functional as F


def embed(x, multires=7):
    """utils/pos_encoder.py:13-66 with get_embedder(7): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^6 x), cos(2^6 x)] -> 3 + 42."""
    out = [x]
    for freq in 2.0 ** torch.linspace(0.0, multires - 1, steps=multires):
        out += [torch.sin(x * freq), torch.cos(x * freq)]
    return torch.cat(out, -1)


def field_mlp(z, hidden, out_w, out_b):
    """The trunk every field shares (models/mlp.py:64-69,100-105): h = leaky_relu(W h + b, 0.1) for every layer of
    `hidden` = [(W, b), ...], then the linear head."""
    h = z
    for w, b in hidden:
        h = F.leaky_relu(F.linear(h, w, b), negative_slope=0.1)
    return F.linear(h, out_w, out_b)


def canonical_field(barys, rots, scales, pose, hidden, out_w, out_b, scale_bary=0.25, bary_size=4):
    """models/mlp.py:94-110: z = [pose | rots | scales | barys] -> (tanh(pred[:, :4]) * scale_bary, pred[:, 4:8], pred[:, 8:])."""
    P = barys.shape[0]
    z = torch.cat([pose.expand(P, -1), rots, scales, barys], dim=1)
    pred = field_mlp(z, hidden, out_w, out_b)
    return torch.tanh(pred[:, :bary_size]) * scale_bary, pred[:, bary_size:bary_size + 4], pred[:, bary_size + 4:]


def deformation_field(canonical, pose, hidden, out_w, out_b, scaling):
    """models/mlp.py:58-71: z = [pose | embed_7(canonical)] -> tanh(pred) * scaling."""
    P = canonical.shape[0]
    z = torch.cat([pose.expand(P, -1), embed(canonical)], dim=1)
    return torch.tanh(field_mlp(z, hidden, out_w, out_b)) * scaling


def shadow_decoder(template, pose, hidden, out_w, out_b):
    """models/mlp.py:262-297: z = [pose[6:] | embed_7(template)] -> sigmoid(pred)  (per template vertex)."""
    P = template.shape[0]
    z = torch.cat([pose[6:].expand(P, -1), embed(template)], dim=1)
    return torch.sigmoid(field_mlp(z, hidden, out_w, out_b))


def face_decoder(kpt, hidden, out_w, out_b):
    """models/mlp.py:235-259: the flattened keypoints through the trunk (a single row)."""
    return field_mlp(kpt.reshape(-1), hidden, out_w, out_b)


def sh4_direction_encoding(d):
    """Stand-in for tiny-cuda-nn's `SphericalHarmonics` encoding of degree 4 (16 outputs) that models/mlp.py:166-179 builds
    (un-vendored third-party code: PARITY UNPINNED for this one function).  tiny-cuda-nn maps its input from [0,1] to
    [-1,1] (x = 2 d - 1) and evaluates the real SH polynomials of degree < 4 -- the same basis and constants as
    utils/sh_utils.py:7-24."""
    x, y, z = (2.0 * d - 1.0).unbind(-1)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * zz - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * xx - 0.54627421529603959 * yy,
        0.59004358992664352 * y * (-3.0 * xx + yy), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * zz), 0.3731763325901154 * z * (5.0 * zz - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * zz), 1.4453057213202769 * z * (xx - yy),
        0.59004358992664352 * x * (-xx + 3.0 * yy)], dim=-1)


def color_field(shs, pose, view_dir, frame_encoding, camera_encoding, shadow, hidden, out_w, out_b, encode=sh4_direction_encoding):
    """models/mlp.py:208-232 (use_pose, use_view_enc): z = [enc(view_dir) | pose | shadow | camera | frame | shs] built by the
    reference's successive concatenations; returns (sigmoid(pred[:, :3]), sigmoid(0.1 + pred[:, 3:4]))."""
    P = shs.shape[0]
    z = shs
    if frame_encoding is not None:
        z = torch.cat([frame_encoding.expand(P, -1), z], dim=1)
    if camera_encoding is not None:
        z = torch.cat([camera_encoding.expand(P, -1), z], dim=1)
    if shadow is not None:
        z = torch.cat([shadow, z], dim=1)
    z = torch.cat([pose.expand(P, -1), z], dim=1)
    z = torch.cat([encode(view_dir), z], dim=1)
    pred = field_mlp(z, hidden, out_w, out_b)
    return torch.sigmoid(pred[:, 0:3]), torch.sigmoid(0.1 + pred[:, 3:4])
