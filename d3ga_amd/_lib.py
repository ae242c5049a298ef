"""ctypes binding of libd3ga_hip.so (include/d3ga.h).  Fails loudly when the library is absent."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# D3GA_LIB_PATH: a diagnostic build (D3GA_DIAG=... python d3ga_amd/csrc/build.py -> libd3ga_hip_diag.so); never set in production
_PATH = os.environ.get("D3GA_LIB_PATH") or os.path.join(_HERE, "libd3ga_hip.so")
_lib = None
ACC_STRIDE = 16          # D3GA_ACC_STRIDE (include/d3ga.h): floats per Gaussian in the screen-space gradient accumulator
# Bumped by every graph replay (graph.CapturedStep / CapturedCutStep): a replayed optimizer updates weights IN PLACE without
# touching their `_version`, so host-side caches keyed by (data_ptr, _version) -- the packed weight panels of mlp.py --
# carry this epoch in their key as well (eager forward -> replays with in-graph Adam -> eager forward used stale panels).
replay_epoch = [0]
ABI_VERSION = 110       # D3GA_VERSION (include/d3ga.h)
# D3GA_KNOB_* (include/d3ga.h), in key order
KNOBS = ("composite_variant", "merge_slots", "tile_assign", "bwd_split", "sort_merge", "ssim_impl", "wgrad_ws", "chain_abl", "chain_grid")
LOSS_PARTIALS = 2048     # D3GA_LOSS_PARTIALS (include/d3ga.h): floats of scratch behind a two-stage loss reduction


class D3GAError(RuntimeError):
    pass


def library_path():
    return _PATH


class RasterParams(ctypes.Structure):
    """struct d3ga_raster_params (include/d3ga.h)."""
    _fields_ = [("P", ctypes.c_int32), ("M", ctypes.c_int32), ("sh_degree", ctypes.c_int32),
                ("W", ctypes.c_int32), ("H", ctypes.c_int32),
                ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
                ("antialiasing", ctypes.c_int32), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
                ("opacity_activation", ctypes.c_int32), ("forward_only", ctypes.c_int32),
                ("acc_self_clearing", ctypes.c_int32), ("n_views", ctypes.c_int32), ("per_view_geometry", ctypes.c_int32),
                ("factor_rows", ctypes.c_int32)]


_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_prm = ctypes.POINTER(RasterParams)

# name -> (argtypes, restype); the trailing _vp of every launch entry is the hipStream_t
_SIGNATURES = {
    "d3ga_version": ([], _i),
    "d3ga_status_string": ([_i], ctypes.c_char_p),
    "d3ga_debug_set": ([ctypes.c_int32, ctypes.c_int32], _i),
    "d3ga_debug_defaults": ([ctypes.POINTER(ctypes.c_int32), ctypes.c_int32], _i),
    "d3ga_lbs_cage_fwd": ([_i, _i] + [_vp] * 8 + [_vp], _i),
    "d3ga_lbs_cage_bwd": ([_i, _i] + [_vp] * 6 + [_vp], _i),
    "d3ga_cage_deform_fwd": ([_i] + [_vp] * 9 + [_vp], _i),
    "d3ga_cage_deform_bwd": ([_i, _i] + [_vp] * 16 + [_vp], _i),
    "d3ga_cage_deform_fwd_ex": ([_i] + [_vp] * 8 + [ctypes.c_int32] + [_vp] * 2 + [_vp], _i),
    "d3ga_cage_deform_bwd_ex": ([_i, _i] + [_vp] * 8 + [ctypes.c_int32] + [_vp] * 9 + [_vp], _i),
    "d3ga_cage_deform_bwd_merged": ([_i, _i] + [_vp] * 8 + [ctypes.c_int32] + [_vp] * 9 + [ctypes.c_int32] + [_vp] * 3 + [_vp], _i),
    "d3ga_cage_deform_bwd_merged_lbs": ([_i, _i] + [_vp] * 8 + [ctypes.c_int32] + [_vp] * 9 + [ctypes.c_int32] + [_vp] * 3 + [_i] + [_vp] * 6 + [_vp], _i),
    "d3ga_fem_energy_fwd": ([_i] + [_vp] * 4 + [_vp], _i),
    "d3ga_fem_energy_bwd": ([_i, _i] + [_vp] * 5 + [_vp], _i),
    "d3ga_raster_scratch_bytes": ([ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _i64, ctypes.POINTER(_i64)], _i),
    "d3ga_raster_scratch_bytes_views": ([ctypes.c_int32] * 4 + [_i64, ctypes.c_int32, ctypes.POINTER(_i64)], _i),
    "d3ga_raster_binning_layout": ([ctypes.c_int32, ctypes.c_int32, _i64, ctypes.POINTER(_i64)], _i),
    "d3ga_raster_img_layout": ([ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_i64)], _i),
    "d3ga_raster_img_layout_blocks": ([ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_i64)], _i),
    "d3ga_raster_img_bytes": ([ctypes.c_int32, ctypes.c_int32, _i64, ctypes.c_int32], _i64),
    "d3ga_raster_preprocess": ([_prm] + [_vp] * 12 + [_i64, _vp, _vp], _i),
    "d3ga_raster_bin_sort": ([_prm, _vp, _vp, _i64, _vp], _i),
    "d3ga_raster_composite_fwd": ([_prm, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_composite_bwd": ([_prm, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_composite_bwd_depth": ([_prm, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_composite_fwd2": ([_prm, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_composite_bwd2": ([_prm, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_preprocess_bwd": ([_prm] + [_vp] * 18 + [_vp], _i),
    "d3ga_raster_forward": ([_prm] + [_vp] * 14 + [_i64, _vp, _vp, _vp, _vp], _i),
    "d3ga_raster_backward": ([_prm] + [_vp] * 11 + [_i64] + [_vp] * 11 + [_vp], _i),
    "d3ga_raster_composite_fwd_l1": ([_prm, _vp, _vp, _vp, _i64] + [_vp] * 7 + [_vp], _i),
    "d3ga_raster_composite_bwd_l1": ([_prm, _vp, _vp, _vp, _i64] + [_vp] * 7 + [_vp], _i),
    "d3ga_raster_backward_l1": ([_prm] + [_vp] * 11 + [_i64] + [_vp] * 15 + [_vp], _i),
    "d3ga_raster_recolor": ([_prm] + [_vp] * 6 + [_vp], _i),
    "d3ga_raster_mark_visible": ([ctypes.c_int32, _vp, _vp, _vp, _vp], _i),
    "d3ga_sh_grad_from_views": ([ctypes.c_int32] * 4 + [_vp, _vp, _i64, _vp, _i64, ctypes.c_float, _vp, _vp], _i),
    "d3ga_compute_bary": ([_i, _i] + [_vp] * 5 + [_vp], _i),
    "d3ga_selftest_row_scan": ([_i, _vp, _vp, _vp], _i),
    "d3ga_selftest_alpha": ([ctypes.c_int32, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_knn3_mean_dist2": ([_i, _vp, _vp, _vp], _i),
    "d3ga_compute_bary_grid": ([_i] + [_vp] * 9 + [_vp], _i),
    "d3ga_knn3_mean_dist2_grid": ([_i] + [_vp] * 6 + [_vp], _i),
    "d3ga_l1_mean_fwd": ([_i64, _vp, _vp, _vp, _vp], _i),
    "d3ga_l1_mean_fwd_ws": ([_i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_l1_mean_bwd": ([_i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_l1_mean_fwd_ws_cell": ([_i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_l1_mean_bwd_cell": ([_i64, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_mlp_panel_bytes": ([ctypes.c_int32, ctypes.c_int32], _i64),
    "d3ga_mlp_pack_weights": ([ctypes.c_int32, ctypes.c_int32, _vp, _i64, _i64, _vp, _vp], _i),
    "d3ga_mlp_linear": ([ctypes.c_int32] * 3 + [_vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_float, _vp, _vp], _i),
    "d3ga_mlp_chain_panel_bytes": ([ctypes.c_int32, ctypes.c_int32], _i64),
    "d3ga_mlp_pack_chain": ([ctypes.c_int32, ctypes.c_int32, _vp, _i64, _i64, _vp, _vp], _i),
    "d3ga_mlp_chain_fwd": ([ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32] + [_vp] * 9 + [_vp], _i),
    "d3ga_mlp_wgrad": ([ctypes.c_int32] * 3 + [_vp] * 4 + [_vp], _i),
    "d3ga_mlp_wgrad_acc": ([ctypes.c_int32] * 3 + [_vp] * 4 + [_vp], _i),
    "d3ga_field_heads_fwd": ([ctypes.c_int32] * 3 + [_vp, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_field_heads_bwd": ([ctypes.c_int32] * 3 + [_vp, _vp, _vp] + [_vp] * 6 + [_vp], _i),
    "d3ga_view_dirs_fwd": ([ctypes.c_int32, _vp, _vp, _vp, _vp], _i),
    "d3ga_view_dirs_bwd": ([ctypes.c_int32, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_sh4_encoding_fwd": ([ctypes.c_int32, _vp, _vp, _vp], _i),
    "d3ga_sh4_encoding_bwd": ([ctypes.c_int32, _vp, _vp, _vp, _vp], _i),
    "d3ga_color_rows_fwd": ([ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp], _i),
    "d3ga_color_rows_bwd": ([ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp], _i),
    "d3ga_ssim_fwd": ([ctypes.c_int32] * 3 + [_vp] * 6 + [_vp], _i),
    "d3ga_ssim_bwd": ([ctypes.c_int32] * 3 + [_vp] * 7 + [_vp], _i),
    "d3ga_ssim_l1_fwd": ([ctypes.c_int32] * 3 + [_vp] * 7 + [_vp], _i),
    "d3ga_ssim_l1_bwd": ([ctypes.c_int32] * 3 + [_vp] * 8 + [_vp], _i),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """The loaded library.  Raises D3GAError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise D3GAError(
                f"{_PATH} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or python d3ga_amd/csrc/build.py). "
                "There is no CPU fallback for this path.")
        import torch  # noqa: F401  -- loads PyTorch-ROCm's HIP runtime first so the process holds one runtime
        L = ctypes.CDLL(_PATH)
        for name, (args, res) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        if L.d3ga_version() != ABI_VERSION:
            raise D3GAError(f"libd3ga_hip.so version {L.d3ga_version()} does not match the Python layer ({ABI_VERSION})")
        info = (ctypes.c_int32 * (2 + 2 * len(KNOBS)))()
        L.d3ga_debug_defaults(info, len(info))
        if info[0] != 0 and os.environ.get("D3GA_ALLOW_ABLATION") != "1":
            raise D3GAError(f"{_PATH} is a timing-ablation build (D3GA_SCAN_ABL={info[0]}): its results are wrong by design. "
                            "Set D3GA_ALLOW_ABLATION=1 to load it for a timing run.")
        # The library reads no environment.  Tests and A/B runs set its debug knobs through THIS layer:
        # D3GA_KNOBS="merge_slots=256,tile_assign=1" (names: KNOBS) -> d3ga_debug_set at load.  Product runs leave it unset.
        spec = os.environ.get("D3GA_KNOBS", "")
        for item in filter(None, (x.strip() for x in spec.split(","))):
            name, _, val = item.partition("=")
            if name not in KNOBS:
                raise D3GAError(f"D3GA_KNOBS: unknown knob {name!r} (known: {', '.join(KNOBS)})")
            st = L.d3ga_debug_set(KNOBS.index(name), int(val))
            if st != 0:
                raise D3GAError(f"d3ga_debug_set({name}, {val}) failed with status {st}")
        _lib = L
    return _lib


def debug_defaults():
    """d3ga_debug_defaults() as a dict: scan_abl, diag, then knob name -> (compiled default, value in effect)."""
    info = (ctypes.c_int32 * (2 + 2 * len(KNOBS)))()
    check(lib().d3ga_debug_defaults(info, len(info)), "d3ga_debug_defaults")
    out = {"scan_abl": info[0], "diag": info[1]}
    for k, name in enumerate(KNOBS):
        out[name] = (info[2 + 2 * k], info[3 + 2 * k])
    return out


def debug_set(name, value=None):
    """Set a debug knob of the library (tests / A/B runs only); value None restores the compiled default."""
    check(lib().d3ga_debug_set(KNOBS.index(name), -2**31 if value is None else int(value)), "d3ga_debug_set")


def check(status, what):
    if status != 0:
        msg = lib().d3ga_status_string(status)
        raise D3GAError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")


def stream_handle():
    """Raw hipStream_t of torch's CURRENT stream on the current device (the stream every launch of this library goes to).
    `torch.cuda.current_stream()` costs ~40 us of host time per call on this build (an os.environ lookup inside
    `_get_device_index`); the raw accessor is a plain C call."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def f32c16(t):
    """float32, contiguous AND 16-byte aligned (the vectorised kernels load 16 bytes per lane).  `.contiguous()` keeps a
    contiguous VIEW with a storage offset as it is -- e.g. `imgs[1]` of an (N,3,H,W) batch with odd H*W, or `x[1:]` -- so such
    a tensor is copied here instead of being rejected by the C ABI (D3GA_E_CONFIG)."""
    if t is None:
        return None
    t = t.float().contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=__import__("torch").contiguous_format)
    return t


def require_cuda(*tensors):
    """Every launch of this library goes to the CURRENT device's current stream (stream_handle): a tensor that lives on
    another GPU would be touched from the wrong device / an unordered stream, so that is refused here."""
    import torch
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise D3GAError("d3ga_amd ops run on the GPU only (tensor on %s); there is no CPU fallback" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise D3GAError(f"tensor on {t.device} but the current device is cuda:{cur}: wrap the call in "
                            f"`with torch.cuda.device({t.device.index}):` (launches go to the current device's stream)")
