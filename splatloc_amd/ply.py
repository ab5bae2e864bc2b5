"""The on-disk artefact exchanged between train_gaussians.py and test.py: `point_cloud.ply`
(SURVEY.md §8f-4).  Drop-ins for `GaussianModel.save_ply` / `GaussianModel.load_ply`
(gaussian_model.py:345-377 / 394-475; consumer test.py:141-144) that need no `plyfile`:

  vertex element, one float32 property per column, in the reference's order
  (construct_list_of_attributes, gaussian_model.py:327-343):
      x y z  nx ny nz  f_dc_0..2  f_rest_0..(3 K - 4)  opacity  scale_0..  rot_0..3  marker  kp_score
  f_dc / f_rest are stored channel-major (`transpose(1, 2).flatten(1)`), normals are zeros, all values
  are the RAW (pre-activation) parameters.

The encoding is what `plyfile` 0.8.1 (environment.yml:8) emits for `PlyData([el]).write(path)`: header
`ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... / end_header`,
then N packed little-endian records.  Host-side byte shuffling: there is no GPU work here.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn


def construct_list_of_attributes(n_dc: int, n_rest: int, n_scale: int, n_rot: int = 4):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    names += ["marker", "kp_score"]
    return names


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False)


def ply_table(gaussians):
    """(names, [P, len(names)] float32 matrix) exactly as save_ply assembles them."""
    xyz = _np(gaussians._xyz)
    f_dc = _np(gaussians._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = _np(gaussians._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    cols = (xyz, np.zeros_like(xyz), f_dc, f_rest, _np(gaussians._opacity), _np(gaussians._scaling),
            _np(gaussians._rotation), _np(gaussians._marker), _np(gaussians._kp_score))
    table = np.concatenate([c.reshape(xyz.shape[0], -1) for c in cols], axis=1)
    names = construct_list_of_attributes(f_dc.shape[1], f_rest.shape[1], gaussians._scaling.shape[1],
                                         gaussians._rotation.shape[1])
    if table.shape[1] != len(names):
        raise RuntimeError(f"save_ply: {table.shape[1]} columns for {len(names)} attributes "
                           "(marker and kp_score are one column each in the reference's layout)")
    return names, np.ascontiguousarray(table, dtype="<f4")


def header_bytes(names, count: int) -> bytes:
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {count}"]
    lines += [f"property float {n}" for n in names]
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def save_ply(gaussians, path: str) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)      # mkdir_p(os.path.dirname(path))
    names, table = ply_table(gaussians)
    with open(path, "wb") as f:
        f.write(header_bytes(names, table.shape[0]))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
              "int32": "i4", "uint": "u4", "uint32": "u4"}


def read_vertex_table(path: str):
    """{property name: 1-D array} of the first element of a PLY file (binary little/big endian or ascii;
    scalar properties only — what save_ply writes)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise RuntimeError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise RuntimeError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise RuntimeError(f"{path}: list properties are not part of the Gaussian map layout")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise RuntimeError(f"{path}: incomplete PLY header")
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, k].astype(t) for k, (n, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        raw = f.read(dt.itemsize * count)
        if len(raw) != dt.itemsize * count:
            raise RuntimeError(f"{path}: truncated payload")
        rec = np.frombuffer(raw, dtype=dt, count=count)
        return {n: rec[n] for n, _ in props}


def load_ply(gaussians, path: str, device="cuda") -> None:
    """GaussianModel.load_ply (gaussian_model.py:394-475): fills the 8 parameter tensors (as
    nn.Parameter on `device`), `active_sh_degree = max_sh_degree` and a zero `max_radii2D`."""
    v = read_vertex_table(path)
    col = lambda n: np.asarray(v[n], dtype=np.float64)  # noqa: E731
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    P = xyz.shape[0]
    features_dc = np.zeros((P, 3, 1))
    for k in range(3):
        features_dc[:, k, 0] = col(f"f_dc_{k}")
    by_index = lambda pre: sorted((n for n in v if n.startswith(pre)), key=lambda x: int(x.split("_")[-1]))  # noqa: E731
    extra = by_index("f_rest_")
    want = 3 * (gaussians.max_sh_degree + 1) ** 2 - 3
    assert len(extra) == want, "{} not eq {}".format(len(extra), want)
    features_extra = np.zeros((P, len(extra)))
    for k, n in enumerate(extra):
        features_extra[:, k] = col(n)
    features_extra = features_extra.reshape((P, 3, (gaussians.max_sh_degree + 1) ** 2 - 1))
    scales = np.stack([col(n) for n in by_index("scale_")], axis=1)
    rots = np.stack([col(n) for n in by_index("rot")], axis=1)
    par = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).requires_grad_(True))  # noqa: E731
    gaussians._xyz = par(xyz)
    gaussians._features_dc = nn.Parameter(torch.tensor(features_dc, dtype=torch.float, device=device)
                                          .transpose(1, 2).contiguous().requires_grad_(True))
    gaussians._features_rest = nn.Parameter(torch.tensor(features_extra, dtype=torch.float, device=device)
                                            .transpose(1, 2).contiguous().requires_grad_(True))
    gaussians._opacity = par(col("opacity")[..., np.newaxis])
    gaussians._scaling = par(scales)
    gaussians._rotation = par(rots)
    gaussians._marker = par(col("marker")[..., np.newaxis])
    gaussians._kp_score = par(col("kp_score")[..., np.newaxis])
    gaussians.active_sh_degree = gaussians.max_sh_degree
    gaussians.max_radii2D = torch.zeros((P,), device=device)
