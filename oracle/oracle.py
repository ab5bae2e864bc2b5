"""ctypes front-end of the CPU oracle (oracle/splat_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of splat_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: raster arithmetic "parity unpinned" (reference sources un-vendored);
covariance / SH / camera / boundary pinned by tests/golden.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TILE = 16


class _Settings(C.Structure):
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
    ]


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (building the checker is not using it)."""
    out = os.path.join(_HERE, "_build")
    src = os.path.join(_HERE, "splat_oracle.c")
    libs = [os.path.join(out, "liborc.so"), os.path.join(out, "liborc_omp.so")]
    fresh = all(os.path.exists(l) and os.path.getmtime(l) >= os.path.getmtime(src) for l in libs)
    if fresh and not force:
        return
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


_LIBS: dict = {}


def _lib(omp: bool):
    key = "omp" if omp else "st"
    if key not in _LIBS:
        build()
        path = os.path.join(_HERE, "_build", "liborc_omp.so" if omp else "liborc.so")
        lib = C.CDLL(path)
        lib.orc_bin.restype = C.c_int64
        lib.orc_num_threads.restype = C.c_int
        _LIBS[key] = lib
    return _LIBS[key]


def num_threads(omp: bool = True) -> int:
    return int(_lib(omp).orc_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class Settings:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    scale_modifier: float = 1.0
    sh_degree: int = 0

    def c(self, channels: int, bg_channels: int, sh_coeffs: int) -> _Settings:
        return _Settings(self.image_height, self.image_width, self.tanfovx, self.tanfovy,
                         self.scale_modifier, self.sh_degree, sh_coeffs, channels, bg_channels)


def forward(st: Settings, bg, means3D, opacities, viewmatrix, projmatrix, campos=None,
            colors_precomp=None, shs=None, scales=None, rotations=None, cov3D_precomp=None,
            omp: bool = False) -> dict:
    """Full forward. Returns every intermediate the spec names."""
    lib = _lib(omp)
    means3D = _f32(means3D, (-1, 3))
    P = means3D.shape[0]
    opacities = _f32(opacities, (-1,))
    scales = _f32(scales, (-1, 3)) if scales is not None else None
    rotations = _f32(rotations, (-1, 4)) if rotations is not None else None
    cov3D_precomp = _f32(cov3D_precomp, (-1, 6)) if cov3D_precomp is not None else None
    V = _f32(viewmatrix, (16,))
    PM = _f32(projmatrix, (16,))
    campos = _f32(campos if campos is not None else np.zeros(3), (3,))
    bg = _f32(bg, (-1,))
    if (shs is None) == (colors_precomp is None):
        raise ValueError("provide exactly one of shs / colors_precomp")
    if ((scales is None) or (rotations is None)) == (cov3D_precomp is None):
        raise ValueError("provide exactly one of (scales, rotations) / cov3D_precomp")
    if shs is not None:
        shs = _f32(shs)
        M = shs.shape[1]
        Cn = 3
    else:
        colors_precomp = _f32(colors_precomp, (P, -1))
        M = 0
        Cn = colors_precomp.shape[1]
    cs = st.c(Cn, bg.shape[0], M)
    H, W = st.image_height, st.image_width
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), np.float32)
    depth = np.zeros(P, np.float32)
    cov3D = np.zeros((P, 6), np.float32)
    conic_opacity = np.zeros((P, 4), np.float32)
    tiles_touched = np.zeros(P, np.uint32)
    rgb = np.zeros((P, 3), np.float32)
    clamped = np.zeros((P, 3), np.uint8)
    lib.orc_preprocess(C.byref(cs), C.c_int32(P), _p(means3D), _p(shs), _p(opacities), _p(scales),
                       _p(rotations), _p(cov3D_precomp), _p(V), _p(PM), _p(campos), _p(radii), _p(xy),
                       _p(depth), _p(cov3D), _p(conic_opacity), _p(tiles_touched), _p(rgb), _p(clamped))
    R = int(tiles_touched.sum(dtype=np.int64))
    keys = np.zeros(max(R, 1), np.uint64)
    vals = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    R2 = lib.orc_bin(C.byref(cs), C.c_int32(P), _p(xy), _p(depth), _p(radii), _p(keys), _p(vals), _p(ranges))
    assert R2 == R
    keys, vals = keys[:R], vals[:R]
    feat = rgb if shs is not None else colors_precomp
    out_color = np.zeros((Cn, H, W), np.float32)
    out_depth = np.zeros((1, H, W), np.float32)
    out_alpha = np.zeros((1, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    lib.orc_composite_fwd(C.byref(cs), _p(ranges), _p(vals), _p(xy), _p(depth), _p(conic_opacity),
                          _p(feat), _p(bg), _p(out_color), _p(out_depth), _p(out_alpha), _p(final_T),
                          _p(n_contrib))
    return dict(color=out_color, depth=out_depth, alpha=out_alpha, radii=radii, xy=xy, view_depth=depth,
                cov3D=cov3D, conic_opacity=conic_opacity, tiles_touched=tiles_touched, rgb=rgb,
                clamped=clamped, keys=keys, point_list=vals, ranges=ranges, final_T=final_T,
                n_contrib=n_contrib, num_rendered=R,
                _inputs=dict(st=st, cs=cs, bg=bg, means3D=means3D, opacities=opacities, V=V, PM=PM,
                             campos=campos, colors_precomp=colors_precomp, shs=shs, scales=scales,
                             rotations=rotations, cov3D_precomp=cov3D_precomp, feat=feat))


def backward(fwd: dict, dL_dcolor, dL_ddepth=None, dL_dalpha=None, omp: bool = False) -> dict:
    """Full backward from a forward() result (double accumulation, see splat_oracle.c)."""
    lib = _lib(omp)
    inp = fwd["_inputs"]
    cs = inp["cs"]
    P = inp["means3D"].shape[0]
    Cn = cs.channels
    H, W = cs.image_height, cs.image_width
    dL_dcolor = _f32(dL_dcolor, (Cn, H, W))
    dL_ddepth = _f32(dL_ddepth, (H, W)) if dL_ddepth is not None else None
    dL_dalpha = _f32(dL_dalpha, (H, W)) if dL_dalpha is not None else None
    dmean2D = np.zeros((P, 2), np.float64)
    dconic = np.zeros((P, 3), np.float64)
    dopacity = np.zeros(P, np.float64)
    dcolors = np.zeros((P, Cn), np.float64)
    ddepth = np.zeros(P, np.float64)
    lib.orc_composite_bwd(C.byref(cs), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(fwd["xy"]),
                          _p(fwd["view_depth"]), _p(fwd["conic_opacity"]), _p(inp["feat"]), _p(inp["bg"]),
                          _p(dL_dcolor), _p(dL_ddepth), _p(dL_dalpha), _p(dmean2D), _p(dconic), _p(dopacity),
                          _p(dcolors), _p(ddepth))
    shs = inp["shs"]
    M = cs.sh_coeffs
    dL_dmeans3D = np.zeros((P, 3), np.float32)
    dL_dmeans2D = np.zeros((P, 3), np.float32)
    have_sr = inp["scales"] is not None
    dL_dscales = np.zeros((P, 3), np.float32) if have_sr else None
    dL_drot = np.zeros((P, 4), np.float32) if have_sr else None
    dL_dcov3D = np.zeros((P, 6), np.float32) if not have_sr else None
    dL_dshs = np.zeros((P, M, 3), np.float32) if shs is not None else None
    dV, dPM, dcam = np.zeros(16, np.float64), np.zeros(16, np.float64), np.zeros(3, np.float64)
    lib.orc_preprocess_bwd(C.byref(cs), C.c_int32(P), _p(inp["means3D"]), _p(shs), _p(inp["scales"]),
                           _p(inp["rotations"]), _p(inp["cov3D_precomp"]), _p(inp["V"]), _p(inp["PM"]),
                           _p(inp["campos"]), _p(fwd["radii"]), _p(fwd["cov3D"]), _p(fwd["clamped"]),
                           _p(dmean2D), _p(dconic), _p(ddepth), _p(dcolors if shs is not None else None),
                           _p(dL_dmeans3D), _p(dL_dmeans2D), _p(dL_dscales), _p(dL_drot), _p(dL_dcov3D),
                           _p(dL_dshs), _p(dV), _p(dPM), _p(dcam))
    return dict(dL_dmeans3D=dL_dmeans3D, dL_dmeans2D=dL_dmeans2D,
                dL_dcolors=None if shs is not None else dcolors.astype(np.float32),
                dL_dopacities=dopacity.astype(np.float32).reshape(P, 1), dL_dscales=dL_dscales,
                dL_drotations=dL_drot, dL_dcov3D=dL_dcov3D, dL_dshs=dL_dshs,
                dL_dviewmatrix=dV.reshape(4, 4), dL_dprojmatrix=dPM.reshape(4, 4), dL_dcampos=dcam,
                _dconic=dconic, _dmean2D=dmean2D, _ddepth=ddepth, _dcolors=dcolors)


def mark_visible(means3D, viewmatrix) -> np.ndarray:
    means3D = _f32(means3D, (-1, 3))
    out = np.zeros(means3D.shape[0], np.uint8)
    _lib(False).orc_mark_visible(C.c_int32(means3D.shape[0]), _p(means3D), _p(_f32(viewmatrix, (16,))), _p(out))
    return out.astype(bool)


def dist2(points, omp: bool = True) -> np.ndarray:
    points = _f32(points, (-1, 3))
    out = np.zeros(points.shape[0], np.float32)
    _lib(omp).orc_dist2(C.c_int32(points.shape[0]), _p(points), _p(out))
    return out


def exp2(x) -> np.ndarray:
    """The oracle's 2^x (orc_exp2: the alpha arithmetic shared bit-for-bit with the HIP path)."""
    lib = _lib(False)
    lib.orc_exp2_array.restype = None
    x = _f32(x).ravel()
    out = np.zeros_like(x)
    lib.orc_exp2_array(C.c_int64(x.size), _p(x), _p(out))
    return out


def set_alpha_mode(mode: int) -> None:
    """0 = shared arithmetic contract (default), 1 = the lineage's literal expf form (both builds)."""
    for omp in (False, True):
        _lib(omp).orc_set_alpha_mode(C.c_int(mode))


def set_row_band(y0: int = 0, y1: int = 0x7FFFFFFF) -> None:
    """Restrict the compositing passes to image rows [y0, y1) (bench.py's single-core sample)."""
    for omp in (False, True):
        _lib(omp).orc_set_row_band(C.c_int(y0), C.c_int(y1))
