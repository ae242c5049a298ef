"""Oracle: camera matrices of the render boundary (numpy, float64 -> float32).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates lib/cameras.py:68-75 and utils/graphics_utils.py:41-75 of the reference:
row-vector convention (matrices are stored transposed), znear=0.01, zfar=100, z in [0,1],
P[3,2]=1.  Pinned by tests/golden/camera_*.npz.
"""
import math
import numpy as np


def world_to_view(R, T):
    """4x4 world->camera with rotation R^T and translation T (graphics_utils.py:41-53, trans=0, scale=1)."""
    M = np.zeros((4, 4), dtype=np.float64)
    M[:3, :3] = np.asarray(R, dtype=np.float64).T
    M[:3, 3] = np.asarray(T, dtype=np.float64)
    M[3, 3] = 1.0
    # the reference inverts, shifts the centre by `translate`=0, scales by 1 and inverts back
    M = np.linalg.inv(np.linalg.inv(M))
    return M.astype(np.float32)


def projection(znear, zfar, fovx, fovy):
    """graphics_utils.py:55-75 (float32 arithmetic as torch.zeros(4,4) is float32)."""
    ty = math.tan(fovy / 2)
    tx = math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera(R, T, fovx, fovy, znear=0.01, zfar=100.0):
    """Returns dict(world_view_transform, projection_matrix, full_proj_transform, camera_center,
    tanfovx, tanfovy) as float32 arrays, all in the reference's transposed layout."""
    wv = world_to_view(R, T).T.copy()
    pr = projection(znear, zfar, fovx, fovy).T.copy()
    full = (wv.astype(np.float32) @ pr.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(wv.astype(np.float32))[3, :3].astype(np.float32)
    return {
        "world_view_transform": wv,
        "projection_matrix": pr,
        "full_proj_transform": full,
        "camera_center": center,
        "tanfovx": math.tan(fovx * 0.5),
        "tanfovy": math.tan(fovy * 0.5),
    }


def paste_shape(crop, h, w):
    """Output (H', W') of renderer.paste for a (3,h,w) image.  renderer.py:36-47."""
    W, H = int(crop[4]), int(crop[5])
    return min(H, h), min(W, w)


def paste(img, crop):
    """renderer.py:36-47 on a numpy/torch (3,h,w) array."""
    left_w, right_w, top_h, bottom_h, W, H = crop[0], crop[1], crop[2], crop[3], int(crop[4]), int(crop[5])
    img = img[:, :, :W] if left_w > right_w else img[:, :, -W:]
    img = img[:, :H, :] if top_h > bottom_h else img[:, -H:, :]
    return img
