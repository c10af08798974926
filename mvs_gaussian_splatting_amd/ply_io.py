"""The on-disk interchange format of the model that feeds the rasterizer: the PLY written by
``GaussianModel.save_ply`` / read by ``load_ply`` (``scene/gaussian_model.py:279-358``; ``render.py`` loads it through
``scene/__init__.py:77-81``).  The reference uses the ``plyfile`` package (absent here); this is a dependency-free
numpy reader / writer of the same file: one ``vertex`` element, binary little-endian, every property ``float``,
in the order

    x y z nx ny nz f_dc_0..2 f_rest_0..(3*(deg+1)^2-4) opacity scale_0..2 rot_0..3

with RAW (pre-activation) values, SH coefficients channel-major (``_features_*`` transposed to ``[P, 3, k]`` and
flattened), normals all zero.  Pinned to the reference by ``tests/golden/ply_layout.npz``: the structured array the
reference's ``save_ply`` hands to ``plyfile`` (field names, order, formats, raw bytes) and the tensors its ``load_ply``
builds from it, captured with a recording stub (``tests/golden/make_golden_ply.py``); only the header text that
``plyfile`` itself writes in front of that array is restated from that package's published behaviour.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """``construct_list_of_attributes`` (``scene/gaussian_model.py:279-291``)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(model, path: str) -> None:
    """Write ``model`` (anything with ``_xyz, _features_dc [P,1,3], _features_rest [P,k,3], _opacity, _scaling,
    _rotation``) exactly as ``GaussianModel.save_ply`` does."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    xyz = model._xyz.detach().cpu().numpy().astype(np.float32)
    f_dc = model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    opac = model._opacity.detach().cpu().numpy().reshape(xyz.shape[0], -1)
    scale = model._scaling.detach().cpu().numpy()
    rot = model._rotation.detach().cpu().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opac, scale, rot), axis=1).astype("<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], scale.shape[1], rot.shape[1])
    assert cols.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {cols.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())


def read_ply_vertices(path: str) -> Tuple[np.ndarray, List[str]]:
    """Structured array of the ``vertex`` element of a binary-little-endian or ascii PLY, and its property names."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif props:
                    raise ValueError(f"{path}: elements after 'vertex' are not supported")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(), dtype=np.dtype([(n, "<" + t) for n, t in props]), count=count)
        elif fmt == "ascii":
            flat = np.loadtxt(f, dtype=np.float64, max_rows=count).reshape(count, len(props))
            data = np.zeros(count, dtype=[(n, "<" + t) for n, t in props])
            for i, (n, _) in enumerate(props):
                data[n] = flat[:, i]
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
    return data, [n for n, _ in props]


def load_ply(path: str, max_sh_degree: int = 3, device="cpu") -> Dict[str, torch.Tensor]:
    """``GaussianModel.load_ply``: returns the raw parameter tensors in the model's layout
    (``_features_dc [P,1,3]``, ``_features_rest [P,(deg+1)^2-1,3]``), float32, on ``device``."""
    v, names = read_ply_vertices(path)
    P = v.shape[0]
    col = lambda n: np.asarray(v[n], dtype=np.float32)  # noqa: E731
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1).reshape(P, 3, 1)
    rest = sorted((n for n in names if n.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    if len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest_* properties do not match SH degree {max_sh_degree}")
    f_rest = np.stack([col(n) for n in rest], axis=1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) if rest \
        else np.zeros((P, 3, 0), np.float32)
    scales = np.stack([col(n) for n in sorted((n for n in names if n.startswith("scale_")),
                                              key=lambda s: int(s.split("_")[-1]))], axis=1)
    rots = np.stack([col(n) for n in sorted((n for n in names if n.startswith("rot")),
                                            key=lambda s: int(s.split("_")[-1]))], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)  # noqa: E731
    return {"_xyz": t(xyz), "_features_dc": t(f_dc).transpose(1, 2).contiguous(),
            "_features_rest": t(f_rest).transpose(1, 2).contiguous(), "_opacity": t(col("opacity")[:, None]),
            "_scaling": t(scales), "_rotation": t(rots), "active_sh_degree": max_sh_degree}
