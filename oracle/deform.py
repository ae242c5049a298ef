"""Oracle: tetrahedral-cage deformation of Gaussians (CPU, torch, any float dtype).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates, in this repo's own words, the tensor program of the reference:
  * models/cage_net.py:213-230   (bary means, J, cov3D = J S J^T, 6-vector pack)
  * lib/cage.py:339-342          (compute_def_grad: J = Ds(tetpoints[tetras][tetra_id]) @ canonical_gradient)
  * lib/cage.py:349-361          (fem_energy)
  * utils/general_utils.py:24-35,58-90 (strip_symmetric order, build_rotation wxyz, L = R diag(s))
  * lib/tet_mesh.py:88-94        (edge order v3-v0, v2-v0, v1-v0, stacked as COLUMNS -- the in-tree
                                  analogue of the un-vendored tetra_sampler.Tetra.gradient)
  * lib/smplman.py:155-171       (LBS of cage vertices; here with K-sparse weights)
Pinned by tests/golden/deform_*.npz (tools/gen_golden.py runs the reference's CageNet.forward).
"""
import torch


def quat_to_rotmat(q):
    """(N,4) quaternion, order (w,x,y,z), normalised here -> (N,3,3).  general_utils.py:58-79."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    w, x, y, z = q.unbind(dim=1)
    rows = [
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], dim=1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], dim=1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1),
    ]
    return torch.stack(rows, dim=1)


def covariance_from_scale_rot(scales, rotations):
    """Sigma = (R diag s)(R diag s)^T.  general_utils.py:81-90 + cage_net.py:161-164."""
    R = quat_to_rotmat(rotations)
    L = R * scales[:, None, :]
    return L @ L.transpose(1, 2)


def pack_sym6(S):
    """(N,3,3) symmetric -> (N,6) in order xx,xy,xz,yy,yz,zz.  general_utils.py:24-35."""
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def unpack_sym6(c):
    """(N,6) -> (N,3,3) symmetric."""
    xx, xy, xz, yy, yz, zz = c.unbind(dim=1)
    return torch.stack(
        [torch.stack([xx, xy, xz], 1), torch.stack([xy, yy, yz], 1), torch.stack([xz, yz, zz], 1)], dim=1
    )


def tet_edge_matrix(corners):
    """(N,4,3) tet corners -> (N,3,3) with COLUMNS (v3-v0, v2-v0, v1-v0).  lib/tet_mesh.py:88-94."""
    v0, v1, v2, v3 = corners[:, 0], corners[:, 1], corners[:, 2], corners[:, 3]
    return torch.stack([v3 - v0, v2 - v0, v1 - v0], dim=2)


def canonical_gradient(canon_points, tetras, tetra_id):
    """inv(Dm) per Gaussian.  lib/cage.py:329."""
    return torch.linalg.inv(tet_edge_matrix(canon_points[tetras][tetra_id]))


def cage_deform(tetpoints, tetras, tetra_id, barys, canon_grad, scales, rotations):
    """Posed means and deformed, packed covariances.

    tetpoints (V,3) posed cage vertices; tetras (T,4) int; tetra_id (P,) int; barys (P,4)
    (already barys+delta_bary, NOT renormalised -- cage_net.py:213); canon_grad (P,3,3);
    scales (P,3) activated; rotations (P,4) wxyz.
    Returns means3D (P,3), cov3D_precomp (P,6).   cage_net.py:218-230.
    """
    corners = tetpoints[tetras.long()][tetra_id.long()]                 # (P,4,3)
    J = tet_edge_matrix(corners) @ canon_grad                            # (P,3,3)
    Sigma = covariance_from_scale_rot(scales, rotations)
    cov = J @ Sigma @ J.transpose(1, 2)
    means = (corners * barys[:, :, None]).sum(dim=1)
    return means, pack_sym6(cov)


def fem_energy(tetpoints, tetras, Dn_inv):
    """Per-tet 0.5 (det F - 1)^2 + 0.5 (tr F^T F - 3).  lib/cage.py:349-361."""
    F = tet_edge_matrix(tetpoints[tetras.long()]) @ Dn_inv
    det = torch.linalg.det(F)
    return 0.5 * (det - 1) ** 2 + 0.5 * ((F * F).sum(dim=(1, 2)) - 3)


def lbs_cage(template, delta, joint_mats, skin_idx, skin_w, Rh=None, Th=None):
    """Linear blend skinning of cage vertices with K-sparse weights.

    template (V,3), delta (V,3) or None, joint_mats (J,4,4), skin_idx (V,K) int, skin_w (V,K).
    v' = (sum_k w_k A[idx_k]) [v+delta; 1];  then v' Rh^T + Th.     lib/smplman.py:155-171
    (the reference uses a dense (V,J) weight matrix; K-sparse is the same sum with zeros dropped).
    """
    v = template if delta is None else template + delta
    T = (joint_mats[skin_idx.long()] * skin_w[:, :, None, None]).sum(dim=1)   # (V,4,4)
    out = (T[:, :3, :3] @ v[:, :, None])[:, :, 0] + T[:, :3, 3]
    if Rh is not None:
        out = out @ Rh.transpose(0, 1)
    if Th is not None:
        out = out + Th
    return out


def skeleton_matrices(bind_state, target_states):
    """Per-joint skinning matrices of Goliath's body model: target transform composed with the inverse bind transform.

    Follows lbsmodel/body_model.py:350-387 (states_to_matrix).  A skeleton state is (translation 3 | unit quaternion xyzw 4 |
    scale 1) per joint, i.e. x -> s R(q) x + t.  bind_state (1,J,8), target_states (B,J,8) -> (B,J,4,4) homogeneous matrices
    M = T_target . T_bind^-1 (the reference returns the top 3x4 block).  Written with rotation matrices instead of the
    reference's quaternion algebra -- the same map."""
    def rot(q):                                            # xyzw unit quaternion -> (..., 3, 3)
        x, y, z, w = q.unbind(-1)
        return torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
            torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
            torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)

    def homo(state):
        t, q, sc = state[..., 0:3], state[..., 3:7], state[..., 7:8]
        q = q / q.norm(dim=-1, keepdim=True)
        M = torch.zeros(state.shape[:-1] + (4, 4), dtype=state.dtype)
        M[..., :3, :3] = rot(q) * sc[..., None]
        M[..., :3, 3] = t
        M[..., 3, 3] = 1.0
        return M
    return homo(target_states) @ torch.linalg.inv(homo(bind_state))
