"""splatloc_amd.ply against tests/golden/ply.npz: the property list, column order and values the reference's
GaussianModel.save_ply assembles, and the tensors its load_ply produces (tests/golden/make_golden_ply.py);
the bytes on disk are plyfile 0.8.1's binary_little_endian encoding.  Host-side: runs without a GPU."""
import os
import types

import numpy as np
import pytest
import torch

from splatloc_amd import ply

KEYS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_marker", "_kp_score")


def _model(d, pre, which):
    return types.SimpleNamespace(**{k: torch.from_numpy(d[f"{pre}{which}{k}"]) for k in KEYS})


@pytest.mark.parametrize("deg", [0, 1])
def test_save_and_load_match_reference(golden_dir, tmp_path, deg):
    d = np.load(os.path.join(golden_dir, "ply.npz"))
    pre = f"deg{deg}_"
    gm = _model(d, pre, "in")
    names, table = ply.ply_table(gm)
    assert names == list(d[pre + "names"]) and all(t == "<f4" for t in d[pre + "dtypes"])
    assert np.array_equal(table, d[pre + "table"])                      # same columns, same values, bit for bit
    path = str(tmp_path / "point_cloud" / "final" / "point_cloud.ply")  # save_ply creates the directories
    ply.save_ply(gm, path)
    raw = open(path, "rb").read()
    P = table.shape[0]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P \
        + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    assert raw.startswith(header.encode()) and len(raw) == len(header) + 4 * P * len(names)
    assert raw[len(header):] == table.astype("<f4").tobytes()
    # load: the reference's tensors (shapes [P,1,3] / [P,K-1,3], transposes), on the CPU here
    out = types.SimpleNamespace(max_sh_degree=deg)
    ply.load_ply(out, path, device="cpu")
    for k in KEYS:
        got = getattr(out, k)
        assert isinstance(got, torch.nn.Parameter) and got.requires_grad and got.dtype == torch.float32
        assert np.array_equal(got.detach().numpy(), d[f"{pre}out{k}"]), k
    assert out.active_sh_degree == int(d[pre + "active_sh_degree"]) == deg
    assert np.array_equal(out.max_radii2D.numpy(), d[pre + "max_radii2D"])
    # and it round-trips
    for k in KEYS:
        assert np.array_equal(getattr(out, k).detach().numpy(), d[f"{pre}in{k}"]), k


def test_reader_handles_ascii_big_endian_and_errors(tmp_path):
    names = ply.construct_list_of_attributes(3, 0, 3)
    tab = np.arange(2 * len(names), dtype=np.float32).reshape(2, -1)
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n"
                 + "".join(f"property float {n}\n" for n in names) + "end_header\n"
                 + "\n".join(" ".join(str(float(v)) for v in row) for row in tab) + "\n")
    v = ply.read_vertex_table(str(a))
    assert np.array_equal(np.stack([v[n] for n in names], 1), tab)
    b = tmp_path / "b.ply"
    b.write_bytes(ply.header_bytes(names, 2).replace(b"little", b"big") + tab.astype(">f4").tobytes())
    v = ply.read_vertex_table(str(b))
    assert np.array_equal(np.stack([v[n] for n in names], 1), tab)
    c = tmp_path / "c.ply"
    c.write_bytes(ply.header_bytes(names, 3) + tab.astype("<f4").tobytes())     # one record short
    with pytest.raises(RuntimeError):
        ply.read_vertex_table(str(c))
    with pytest.raises(AssertionError):                                         # wrong SH degree for the file
        ply.load_ply(types.SimpleNamespace(max_sh_degree=1), str(b), device="cpu")
    (tmp_path / "d.ply").write_text("plx\n")
    with pytest.raises(RuntimeError):
        ply.read_vertex_table(str(tmp_path / "d.ply"))
