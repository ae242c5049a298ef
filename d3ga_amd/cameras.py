"""Host-side camera set-up of the render boundary (mirrors lib/cameras.py:14-75 and
utils/graphics_utils.py:41-75 of the reference).

The reference rebuilds a `Camera` (two numpy 4x4 inversions, three small H2D copies) on every `render()` call,
twice per training step.  Here the matrices of a (R, T, FoV) triple are computed once on the host in float64,
rounded to float32 exactly where the reference rounds, uploaded once, and cached per device.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

ZNEAR, ZFAR = 0.01, 100.0          # lib/cameras.py:62-63


def _view_matrix(R, T):
    """float32 world->view matrix (graphics_utils.getWorld2View2 with translate=0, scale=1)."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(T, dtype=np.float64).reshape(3)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)           # the reference inverts twice (camera-centre shift of zero in between)
    return np.linalg.inv(c2w).astype(np.float32)


def _projection_matrix(fovx, fovy, znear=ZNEAR, zfar=ZFAR):
    """float32 perspective matrix with z in [0,1] and w = z_view (graphics_utils.getProjectionMatrix)."""
    top = math.tan(fovy / 2) * znear
    right = math.tan(fovx / 2) * znear
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (2.0 * right)
    P[1, 1] = 2.0 * znear / (2.0 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera:
    """Same attribute names as the reference's Camera (world_view_transform, projection_matrix,
    full_proj_transform, camera_center are float32 tensors in the transposed, row-vector layout)."""

    def __init__(self, colmap_id, R, T, FoVx, FoVy, image_name=None, uid=None, width=None, height=None,
                 data_device="cuda", _matrices=None):
        self.uid, self.colmap_id, self.image_name = uid, colmap_id, image_name
        self.R, self.T, self.FoVx, self.FoVy = R, T, float(FoVx), float(FoVy)
        self.image_width, self.image_height = width, height
        self.znear, self.zfar = ZNEAR, ZFAR
        buf = _matrices if _matrices is not None else self.pack_matrices(R, T, self.FoVx, self.FoVy, data_device)
        self.matrices = buf                                    # (51,) float32 on the device: view | proj | full | centre
        self.world_view_transform = buf[0:16].view(4, 4)
        self.projection_matrix = buf[16:32].view(4, 4)
        self.full_proj_transform = buf[32:48].view(4, 4)
        self.camera_center = buf[48:51]
        self.tanfovx = math.tan(self.FoVx * 0.5)
        self.tanfovy = math.tan(self.FoVy * 0.5)

    @staticmethod
    def pack_host_cached(batch):
        """pack_host of a batch's camera (+ the two tangents: 53 floats), cached on the numbers that define it: a capture rig has a
        few hundred cameras and a trainer draws them again and again (datasets/actorshq_dataset.py:229) -- three 4x4 inversions in
        numpy cost ~60 us of host time per step otherwise (tools/prof_host.py, round 6)."""
        R = np.ascontiguousarray(np.asarray(batch["R"], dtype=np.float64))
        T = np.ascontiguousarray(np.asarray(batch["T"], dtype=np.float64))
        key = (R.tobytes(), T.tobytes(), float(batch["FoVx"]), float(batch["FoVy"]))
        hit = _host_cache.get(key)
        if hit is None:
            hit = np.empty(53, dtype=np.float32)
            hit[:51] = Camera.pack_host(R, T, batch["FoVx"], batch["FoVy"])
            hit[51] = math.tan(float(batch["FoVx"]) * 0.5)
            hit[52] = math.tan(float(batch["FoVy"]) * 0.5)
            _host_cache[key] = hit
            if len(_host_cache) > _CACHE_MAX:
                _host_cache.popitem(last=False)
        else:
            _host_cache.move_to_end(key)
        return hit

    @staticmethod
    def pack_host(R, T, FoVx, FoVy):
        """The 51 float32 numbers of a camera on the HOST: view (16) | projection (16) | full projection (16) | centre (3)."""
        wv = _view_matrix(R, T).T.copy()
        pr = _projection_matrix(float(FoVx), float(FoVy)).T.copy()
        full = wv @ pr
        center = np.linalg.inv(wv)[3, :3].astype(np.float32)
        return np.concatenate([wv.ravel(), pr.ravel(), full.ravel(), center]).astype(np.float32)

    @staticmethod
    def pack_matrices(R, T, FoVx, FoVy, device):
        return torch.from_numpy(Camera.pack_host(R, T, FoVx, FoVy)).to(torch.device(device))     # one H2D copy


class CameraSlot:
    """A camera whose numbers live in ONE static device buffer, so that a step captured in a hipGraph can be replayed with a
    different camera every time (the reference trains on one random camera per step, datasets/actorshq_dataset.py:229,
    models/trainer.py:91-110).  Layout (53 float32): view (16) | projection (16) | full projection (16) | centre (3) |
    tan(FoVx/2) | tan(FoVy/2).  The two tangents are read by the kernels FROM THE BUFFER (the settings carry tanfovx = 0 as
    the marker, include/d3ga.h), so only the raster size is fixed per slot -- use one slot / one captured step per
    resolution.  `set(batch)` stages the numbers in pinned memory and enqueues one asynchronous H2D copy on the current
    stream; put `batch["camera_slot"] = slot` into the batch handed to `render()`."""

    _RING = 16
    _CAM_BYTES = 224                                          # 53 floats, padded to a multiple of 8 bytes for the cells behind

    def __init__(self, width, height, device="cuda", cells=0):
        """cells: number of 8-byte ADDRESS CELLS kept behind the camera in the same device buffer (`cell(i)`): a
        `graph.TensorSlot(first, arena=slot, index=i)` then lives there, and `set()` moves the camera AND the cells with the ONE
        host-to-device copy it makes anyway (a step that repoints its target image needs one copy per replay instead of two)."""
        self.image_width, self.image_height = int(width), int(height)
        dev = torch.device(device)
        self.n_cells = int(cells)
        nbytes = self._CAM_BYTES + 8 * self.n_cells
        self.buffer = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.matrices = self.buffer[:212].view(torch.float32)
        self.cells = self.buffer[self._CAM_BYTES:].view(torch.int64) if self.n_cells else None
        # pinned staging ring: the host runs many steps ahead of the GPU, so a staging buffer may only be rewritten once the
        # asynchronous copy that reads it has executed (an event per slot of the ring)
        self._ring = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(self._RING)]
        self._events = [None] * self._RING
        self._next = 0
        if dev.type == "cuda":
            self._ring = [t.pin_memory() for t in self._ring]
        self._host_cam = torch.zeros(53, dtype=torch.float32)   # what the device holds / will hold: re-sent with every copy
        self._host_cam_np = self._host_cam.numpy()              # (numpy views of the host tensors: an element write through torch costs ~2 us)
        self._ring_np = [t.numpy() for t in self._ring]
        self._host_cells = torch.zeros(max(self.n_cells, 1), dtype=torch.int64)
        self._host_cells_np = self._host_cells.numpy()
        self._cells_dirty = False
        self.world_view_transform = self.matrices[0:16].view(4, 4)
        self.projection_matrix = self.matrices[16:32].view(4, 4)
        self.full_proj_transform = self.matrices[32:48].view(4, 4)
        self.camera_center = self.matrices[48:53]              # 3 used as the centre; [3], [4] = the tangents
        self.tanfovx = self.tanfovy = 0.0                      # marker: read them from the buffer
        self.znear, self.zfar = ZNEAR, ZFAR

    def cell(self, i):
        """(1,) int64 device view of address cell i (for graph.TensorSlot)."""
        if not 0 <= i < self.n_cells:
            raise IndexError(f"CameraSlot has {self.n_cells} address cells")
        return self.cells[i:i + 1]

    def stage_cell(self, i, address):
        """Host side of a TensorSlot living in this slot: the address goes out with the next `set()` / `flush()`."""
        self._host_cells_np[i] = int(address)
        self._cells_dirty = True

    def flush(self):
        """One asynchronous H2D copy of camera + cells from a pinned staging buffer, on the current stream."""
        i = self._next
        self._next = (i + 1) % self._RING
        if self._events[i] is not None:
            self._events[i].synchronize()                      # the copy that last read this staging buffer has run
        stage = self._ring[i]
        snp = self._ring_np[i]
        snp[:212].view(np.float32)[:] = self._host_cam_np
        if self.n_cells:
            snp[self._CAM_BYTES:].view(np.int64)[:] = self._host_cells_np[:self.n_cells]
        self.buffer.copy_(stage, non_blocking=True)
        self._cells_dirty = False
        if self.buffer.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._events[i] = ev
        return self

    def set(self, batch):
        if int(batch["width"]) != self.image_width or int(batch["height"]) != self.image_height:
            raise ValueError(f"CameraSlot is {self.image_width}x{self.image_height}; the batch is "
                             f"{batch['width']}x{batch['height']} (one slot / captured step per raster size)")
        self._host_cam_np[:] = Camera.pack_host_cached(batch)
        return self.flush()


_cache = OrderedDict()
_host_cache = OrderedDict()
_CACHE_MAX = 512


def batch_to_camera(batch, device="cuda"):
    """lib/cameras.py:14-26.  Only the DEVICE MATRICES are cached (keyed on the numbers that define them: R, T, FoV, device);
    the Camera object itself is built fresh from the current batch, so uid / colmap_id / image_name / image size always
    belong to this frame."""
    R = np.ascontiguousarray(np.asarray(batch["R"], dtype=np.float64))
    T = np.ascontiguousarray(np.asarray(batch["T"], dtype=np.float64))
    key = (R.tobytes(), T.tobytes(), float(batch["FoVx"]), float(batch["FoVy"]), str(device))
    mats = _cache.get(key)
    if mats is None:
        mats = Camera.pack_matrices(R, T, batch["FoVx"], batch["FoVy"], device)
        _cache[key] = mats
        if len(_cache) > _CACHE_MAX:
            _cache.popitem(last=False)
    else:
        _cache.move_to_end(key)
    return Camera(colmap_id=batch.get("camera_id"), R=R, T=T, FoVx=batch["FoVx"], FoVy=batch["FoVy"],
                  image_name=f'{batch.get("frame_id")}_{batch.get("camera_id")}', uid=batch.get("frame_id"),
                  width=batch.get("width"), height=batch.get("height"), data_device=device, _matrices=mats)
