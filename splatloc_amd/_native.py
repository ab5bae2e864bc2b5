"""ctypes binding of libsplatraster.so (C ABI: include/splatraster.h).

Fails loudly when the HIP library is missing and cannot be built: the product path never
falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_LIB = None
ABI_VERSION = 13   # == SPLATRASTER_ABI_VERSION of include/splatraster.h

OK = 0
WARN_LOOKBACK_STALL = 5   # splatraster_poll_errors() only; not an error of any frame
_ERR_NAMES = {1: "bad argument", 2: "HIP runtime error", 3: "unsupported configuration",
              4: "tile instance count overflow", 5: "look-back stall (results late, never wrong)"}


class Settings(C.Structure):
    """struct splatraster_settings"""
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32),
        ("sh_coeffs", C.c_int32),
        ("channels", C.c_int32),
        ("bg_channels", C.c_int32),
        ("prefiltered", C.c_int32),
        ("debug", C.c_int32),
    ]


MAX_WINDOW_VIEWS = 8   # SPLATRASTER_MAX_WINDOW_VIEWS


class WindowView(C.Structure):
    """struct splatraster_window_view"""
    _fields_ = [("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("radii", C.c_void_p), ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("out_alpha", C.c_void_p),
                ("dL_dout_color", C.c_void_p), ("dL_dout_depth", C.c_void_p), ("dL_dout_alpha", C.c_void_p),
                ("dL_dmeans2D", C.c_void_p), ("dL_dout_last", C.c_void_p), ("color_grad_channels", C.c_int32)]


class LossView(C.Structure):
    """struct splatraster_loss_view"""
    _fields_ = [(n, C.c_void_p) for n in ("image", "depth", "marker", "gt_image", "gt_depth", "kp", "exposure", "g_image", "g_depth",
                                          "g_marker")]


class RawParams(C.Structure):
    """struct splatraster_raw_params"""
    _fields_ = [("scaling", C.c_void_p), ("rotation", C.c_void_p), ("opacity", C.c_void_p), ("f_dc", C.c_void_p),
                ("extra_channels", C.c_int32), ("dL_dscaling", C.c_void_p), ("dL_drotation", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_df_dc", C.c_void_p), ("dL_dextra", C.c_void_p), ("reg_row_grad", C.c_void_p), ("reg_out", C.c_void_p),
                ("reg_weight", C.c_float)]


class RawForward(C.Structure):
    """struct splatraster_raw_forward"""
    _fields_ = [("scaling", C.c_void_p), ("rotation", C.c_void_p), ("opacity", C.c_void_p), ("f_dc", C.c_void_p), ("extra", C.c_void_p),
                ("extra_channels", C.c_int32), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("opacities", C.c_void_p),
                ("colors", C.c_void_p)]


class GeometryLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ("rec0", "rec1", "tiles_touched", "depth_order", "offsets", "rgb", "clamped", "total")]


class BinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("point_list", "tile_list", "ranges", "total")]


class ImageLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("final_T", "n_contrib", "total")]


class Model(C.Structure):
    """struct splatraster_model: the 8 raw parameter tensors of GaussianModel (or their Adam moments)"""
    _fields_ = [("P", C.c_int32), ("f_rest_width", C.c_int32), ("marker_width", C.c_int32), ("kp_width", C.c_int32),
                ("scaling_width", C.c_int32)] + [(n, C.c_void_p) for n in
                                                 ("xyz", "f_dc", "f_rest", "opacity", "marker", "kp_score", "scaling",
                                                  "rotation")]


class AdamGroup(C.Structure):
    """struct splatraster_adam_group"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("row_gate", C.c_void_p), ("numel", C.c_int64), ("row_width", C.c_int32), ("lr", C.c_float),
                ("step", C.c_double)]


# every symbol include/splatraster.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
SYMBOLS = {
    "splatraster_geometry_bytes": (_sz, [_i32]),
    "splatraster_binning_bytes": (_sz, [_i32, _i64, _i32, _i32, _i32]),
    "splatraster_image_bytes": (_sz, [_i32, _i32]),
    "splatraster_forward_geometry": (C.c_int, [C.POINTER(Settings), _i32] + [_vp] * 9 + [_vp, _vp, C.POINTER(_i64), _vp]),
    "splatraster_forward_render": (C.c_int, [C.POINTER(Settings), _i32, _i64] + [_vp] * 9),
    "splatraster_backward": (C.c_int, [C.POINTER(Settings), _i32, _i64] + [_vp] * 33),
    "splatraster_window_geometry_bytes": (_sz, [_i32, _i32]),
    "splatraster_window_binning_bytes": (_sz, [_i32, _i32, _i64, _i32, _i32, _i32]),
    "splatraster_window_image_bytes": (_sz, [_i32, _i32, _i32]),
    "splatraster_forward_window_geometry": (C.c_int, [C.POINTER(Settings), _i32, C.POINTER(WindowView), _i32] + [_vp] * 6
                                            + [C.POINTER(_i64), _vp]),
    "splatraster_forward_window_render": (C.c_int, [C.POINTER(Settings), _i32, C.POINTER(WindowView), _i32, C.POINTER(_i64)]
                                          + [_vp] * 6),
    "splatraster_backward_window": (C.c_int, [C.POINTER(Settings), _i32, C.POINTER(WindowView), _i32, C.POINTER(_i64)]
                                    + [_vp] * 16),
    "splatraster_forward_window_geometry_raw": (C.c_int, [C.POINTER(Settings), _i32, C.POINTER(WindowView), _i32, _vp, C.POINTER(RawForward),
                                                          _vp, C.POINTER(_i64), _vp]),
    "splatraster_backward_window_raw": (C.c_int, [C.POINTER(Settings), _i32, C.POINTER(WindowView), _i32, C.POINTER(_i64)]
                                        + [_vp] * 8 + [C.POINTER(RawParams)] + [_vp] * 2),
    "splatraster_get_window_geometry_layout": (C.c_int, [_i32, _i32, C.POINTER(GeometryLayout)]),
    "splatraster_get_window_binning_layout": (C.c_int, [_i32, _i32, _i64, _i32, _i32, _i32, C.POINTER(BinningLayout)]),
    "splatraster_get_window_image_layout": (C.c_int, [_i32, _i32, _i32, C.POINTER(ImageLayout)]),
    "splatraster_debug_set_small_panel_max_waves": (C.c_int, [C.c_int]),
    "splatraster_debug_set_split_max_waves": (C.c_int, [C.c_int]),
    "splatraster_debug_set_fwd_team": (C.c_int, [C.c_int]),
    "splatraster_debug_set_front_end": (C.c_int, [C.c_int]),
    "splatraster_debug_set_tile_sort_cap": (C.c_int, [C.c_int]),
    "splatraster_debug_set_sort_fork": (C.c_int, [C.c_int]),
    "splatraster_debug_set_payload_stream_min": (C.c_int, [C.c_int64]),
    "splatraster_mark_visible": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp]),
    "splatraster_get_geometry_layout": (C.c_int, [_i32, C.POINTER(GeometryLayout)]),
    "splatraster_get_binning_layout": (C.c_int, [_i32, _i64, _i32, _i32, _i32, C.POINTER(BinningLayout)]),
    "splatraster_get_image_layout": (C.c_int, [_i32, _i32, C.POINTER(ImageLayout)]),
    "splatraster_sort_tmp_bytes": (_sz, [_i64]),
    "splatraster_sort_pairs_u32": (C.c_int, [_i64, _vp, _vp, _i32, _vp, _vp]),
    "splatraster_timing_enable": (C.c_int, [C.c_int]),
    "splatraster_timing_select": (C.c_int, [C.c_uint32]),
    "splatraster_timing_collect": (C.c_int, [_vp, _vp]),
    "splatknn_workspace_bytes": (_sz, [_i32]),
    "splatknn_dist2": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "splatknn_debug_set_grid_min": (C.c_int, [_i32]),
    "splatraster_activate_forward": (C.c_int, [_i32] * 5 + [_vp] * 13),
    "splatraster_activate_backward": (C.c_int, [_i32] * 5 + [_vp] * 19),
    "splatraster_densification_stats": (C.c_int, [_i32] + [_vp] * 6),
    "splatraster_densification_stats_window": (C.c_int, [_i32, _i32] + [_vp] * 6),
    "splatraster_densify_workspace_bytes": (_sz, [_i32]),
    "splatraster_densify_plan": (C.c_int, [C.POINTER(Model), _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _i32,
                                           _i32, _vp, C.POINTER(_i32), _vp]),
    "splatraster_densify_apply": (C.c_int, [C.POINTER(Model), C.POINTER(Model), C.POINTER(Model), _vp, C.c_uint64,
                                            C.c_uint64, _vp, _i32, C.POINTER(Model), C.POINTER(Model), C.POINTER(Model),
                                            _vp, _vp, _vp]),
    "splatraster_model_append": (C.c_int, [C.POINTER(Model)] * 7 + [_vp]),
    "splatraster_adam_step": (C.c_int, [_i32, C.POINTER(AdamGroup), C.c_double, C.c_double, C.c_double, C.c_float, _vp]),
    "splatraster_adam_step_radii": (C.c_int, [_i32, C.POINTER(AdamGroup), C.c_double, C.c_double, C.c_double, C.c_float, _i32, _vp, _vp, _vp]),
    "splatraster_isotropic_loss_workspace_bytes": (_sz, [_i32]),
    "splatraster_isotropic_loss": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "splatraster_mapping_loss_workspace_bytes": (_sz, [_i32]),
    "splatraster_mapping_loss": (C.c_int, [_i32] + [_vp] * 6 + [C.c_float] + [_vp] * 7),
    "splatraster_mapping_loss_window": (C.c_int, [_i32, _i32, _vp, C.c_float, _vp, _vp, _vp]),
    "splatraster_refinement_loss_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "splatraster_refinement_loss": (C.c_int, [_i32, _i32, _i32, C.c_float] + [_vp] * 6),
    "splatraster_eval_metrics_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "splatraster_eval_metrics": (C.c_int, [_i32, _i32, _i32] + [_vp] * 5),
    "splatraster_l1_rgbd_loss": (C.c_int, [_i64, _vp, _vp, _i64, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "splatraster_pose_step": (C.c_int, [_vp] * 5 + [C.c_float] * 5 + [C.c_int] + [_vp] * 5),
    "splatraster_error_string": (C.c_char_p, [C.c_int]),
    "splatraster_last_hip_error": (C.c_char_p, []),
    "splatraster_abi_version": (C.c_int, []),
    "splatraster_poll_errors": (C.c_int, []),
    "splatraster_debug_set_spin_limit": (C.c_int, [C.c_uint32]),
    "splatraster_debug_set_deterministic": (C.c_int, [C.c_int]),
    "splatraster_debug_exp2": (C.c_int, [_i64, _vp, _vp, _vp]),
    "splatraster_debug_poison_lds": (C.c_int, [C.c_uint32, _vp]),
}


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load (building first when possible) the HIP library; raise if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("SPLATRASTER_LIB", _build.LIB_PATH)  # override: perf experiments only
    if path == _build.LIB_PATH and build_if_missing and _build.have_hipcc():
        # mtime-cached: a no-op when nothing changed, a rebuild when a source / header is newer than
        # the in-tree .so (which is git-ignored and survives checkouts)
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(
                "splatloc_amd: the HIP extension libsplatraster.so could not be built "
                f"({e}). There is no CPU fallback.") from e
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing and hipcc is not available: run `python -m splatloc_amd.build` "
                           "on a machine with ROCm. There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.splatraster_abi_version.restype = C.c_int
    got = int(lib.splatraster_abi_version())
    if got != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {got}, this binding needs {ABI_VERSION}: stale library, "
                           "rebuild with `python -m splatloc_amd.build --force`")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    # A/B knobs for perf experiments (process-wide debug switches of the library; never set in production)
    if os.environ.get("SPLATRASTER_SPLIT_MAX_WAVES"):
        lib.splatraster_debug_set_split_max_waves(int(os.environ["SPLATRASTER_SPLIT_MAX_WAVES"]))
    if os.environ.get("SPLATRASTER_FWD_TEAM"):     # -1 auto, 0 one wave per quadrant, 1 teams wherever a launch order exists
        lib.splatraster_debug_set_fwd_team(int(os.environ["SPLATRASTER_FWD_TEAM"]))
    if os.environ.get("SPLATRASTER_FRONT_END"):   # -1 auto, 0 radix sorts, 1 binned whenever the shape allows
        lib.splatraster_debug_set_front_end(int(os.environ["SPLATRASTER_FRONT_END"]))
    if os.environ.get("SPLATRASTER_TILE_SORT_CAP"):
        lib.splatraster_debug_set_tile_sort_cap(int(os.environ["SPLATRASTER_TILE_SORT_CAP"]))
    if os.environ.get("SPLATRASTER_SORT_FORK"):
        lib.splatraster_debug_set_sort_fork(int(os.environ["SPLATRASTER_SORT_FORK"]))
    if os.environ.get("SPLATRASTER_SMALL_PANEL_MAX_WAVES"):
        lib.splatraster_debug_set_small_panel_max_waves(int(os.environ["SPLATRASTER_SMALL_PANEL_MAX_WAVES"]))
    return lib


def check(status: int, what: str) -> None:
    if status != OK:
        lib = load()
        detail = lib.splatraster_last_hip_error().decode() if status in (2, 5) else ""
        raise RuntimeError(f"{what} failed: {_ERR_NAMES.get(status, status)} {detail}".strip())


STAGES = ("preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges", "composite_fwd", "composite_bwd",
          "preprocess_bwd", "payload")


def timing_enable(on: bool) -> None:
    check(load().splatraster_timing_enable(int(on)), "timing_enable")


def timing_select(stages) -> None:
    """Time only the named stages (an event record between kernels costs ~10 us of idle GPU)."""
    mask = 0
    for st in stages:
        mask |= 1 << STAGES.index(st)
    check(load().splatraster_timing_select(mask), "timing_select")


def timing_collect() -> dict:
    """{stage: (total_ms, launches)} since the last collect (HIP events on the launch stream)."""
    n = len(STAGES)
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(load().splatraster_timing_collect(ms, cnt), "timing_collect")
    return {s: (ms[i], cnt[i]) for i, s in enumerate(STAGES)}


def set_deterministic(on: bool) -> None:
    """Deterministic-sum debug mode of the backward (bit-reproducible gradients; see splatraster.h)."""
    check(load().splatraster_debug_set_deterministic(int(bool(on))), "set_deterministic")


def set_front_end(mode: int) -> None:
    """-1: default choice; 0: the two global radix sorts always; 1: the binned front end whenever the shape allows
    (splatraster_debug_set_front_end; bit-identical results either way)."""
    check(load().splatraster_debug_set_front_end(int(mode)), "set_front_end")


def poll_stall() -> bool:
    """True once per soft look-back stall raised since the last poll (monitoring; the frames are still correct)."""
    st = load().splatraster_poll_errors()
    if st == WARN_LOOKBACK_STALL:
        return True
    check(st, "poll_errors")
    return False
