"""PLY interchange format of the model: layout restated from scene/gaussian_model.py:279-358.
Parity unpinned: plyfile is not installed and the reference ships no PLY."""
import numpy as np
import torch

from mvs_gaussian_splatting_amd.ply_io import attribute_names, load_ply, read_ply_vertices, save_ply
from mvs_gaussian_splatting_amd.synthetic import SyntheticGaussianModel


def test_save_load_roundtrip_and_layout(tmp_path):
    model = SyntheticGaussianModel(257, 3, seed=3)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    save_ply(model, path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    names = [l.split()[2] for l in lines if l.startswith("property")]
    assert all(l.split()[1] == "float" for l in lines if l.startswith("property"))
    assert names == attribute_names(3, 45) and len(names) == 62
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] and names[54:] == \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == 257 * 62 * 4
    rows = np.frombuffer(body, dtype="<f4").reshape(257, 62)
    assert np.array_equal(rows[:, 0:3], model._xyz.numpy()) and not rows[:, 3:6].any()
    # f_rest is channel-major: property f_rest_j holds coefficient j % 15 + 1 of channel j // 15
    assert np.array_equal(rows[:, 9 + 15 * 1 + 4], model._features_rest[:, 4, 1].numpy())
    got = load_ply(path, max_sh_degree=3)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert got[k].shape == getattr(model, k).shape and torch.equal(got[k], getattr(model, k)), k
    v, props = read_ply_vertices(path)
    assert props == names and v.shape == (257,)


def test_reads_ascii_and_degree_zero(tmp_path):
    model = SyntheticGaussianModel(5, 0, seed=1)
    names = attribute_names(3, 0)
    p = tmp_path / "a.ply"
    cols = np.concatenate([model._xyz.numpy(), np.zeros((5, 3)), model._features_dc.reshape(5, 3).numpy(),
                           model._opacity.numpy(), model._scaling.numpy(), model._rotation.numpy()], axis=1)
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment test\nelement vertex 5\n" + "".join(f"property float {n}\n" for n in names)
                + "end_header\n")
        for r in cols:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    got = load_ply(str(p), max_sh_degree=0)
    assert torch.allclose(got["_xyz"], model._xyz) and got["_features_rest"].shape == (5, 0, 3)
    assert torch.allclose(got["_rotation"], model._rotation)
