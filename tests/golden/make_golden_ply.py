"""Generates tests/golden/ply_layout.npz from the reference's own ``GaussianModel.save_ply`` / ``load_ply``
(scene/gaussian_model.py:279-310 and :312-358), run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden_ply.py

The reference hands its arrays to the third-party ``plyfile`` package (absent here).  Everything up to that hand-over
is the reference's own numpy code: ``save_ply`` builds ONE structured array -- field names from
``construct_list_of_attributes``, all 'f4', rows = concatenate(xyz, normals, f_dc, f_rest, opacity, scale, rotation) --
and passes it to ``PlyElement.describe(elements, 'vertex')``; ``load_ply`` reads named columns back from
``plydata.elements[0]`` and its ``.properties``.  A RECORDING stub for ``plyfile`` (the technique of
make_golden_model.py) captures that array (field names, formats, raw bytes, element name) and serves it back to
``load_ply``; the fixture stores the model's raw parameters, the captured array and the tensors ``load_ply`` produced.
What plyfile itself adds -- the textual header ``ply / format binary_little_endian 1.0 / element vertex N /
property float <name> ... / end_header`` in front of the array's bytes (``PlyData([el]).write(path)``: binary, native
byte order) -- is plyfile's published behaviour and stays restated in mvs_gaussian_splatting_amd/ply_io.py.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_model as gm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
RECORDED = {}


class _Prop:
    def __init__(self, name):
        self.name = name


class RecordingPlyElement:
    def __init__(self, data, name):
        self.data, self.name = data, name
        self.properties = [_Prop(n) for n in data.dtype.names]

    @staticmethod
    def describe(data, name, *a, **kw):
        assert not a and not kw, "save_ply passes (elements, 'vertex') only"
        RECORDED["describe"] = (np.array(data, copy=True), name)
        return RecordingPlyElement(data, name)

    def __getitem__(self, key):
        return self.data[key]


class RecordingPlyData:
    def __init__(self, elements, *a, **kw):
        assert not a and not kw, "save_ply constructs PlyData([el]) with defaults (binary, native byte order)"
        self.elements = list(elements)

    def write(self, path):
        RECORDED["write"] = (path, [e.name for e in self.elements])

    @staticmethod
    def read(path):
        data, name = RECORDED["describe"]
        return RecordingPlyData([RecordingPlyElement(data, name)])


def main():
    mod = gm.load_reference_model_module()
    mod.PlyData, mod.PlyElement = RecordingPlyData, RecordingPlyElement       # the names gaussian_model.py imported
    out = {}
    for tag, P, deg, seed in (("deg3", 257, 3, 51), ("deg1", 64, 1, 52), ("deg0", 33, 0, 53)):
        m, _ = gm.build_model(mod, P, deg, seed)
        for k, a in gm.ATTR.items():
            out[f"{tag}/param/{k}"] = getattr(m, a).detach().numpy().copy()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "point_cloud", "iteration_7", "point_cloud.ply")
            m.save_ply(path)                                                  # :293-310
            assert os.path.isdir(os.path.dirname(path))                       # mkdir_p ran; nothing was written (stub)
            assert RECORDED["write"] == (path, ["vertex"])
            arr, name = RECORDED["describe"]
            out[f"{tag}/element_name"] = np.array(name)
            out[f"{tag}/field_names"] = np.array(list(arr.dtype.names))
            out[f"{tag}/field_formats"] = np.array([arr.dtype[n].str for n in arr.dtype.names])
            out[f"{tag}/itemsize_count"] = np.array([arr.dtype.itemsize, arr.shape[0]])
            out[f"{tag}/raw_bytes"] = np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()
            m2 = mod.GaussianModel(deg, modelcg=types.SimpleNamespace(learn_split_distance=False, learn_split_scale=False,
                                                                      symmetric_split=False, split_notreinit=False))
            with gm._Patched():                                               # device="cuda" dropped
                m2.load_ply(path)                                             # :312-358
            for k, a in gm.ATTR.items():
                t = getattr(m2, a)
                assert t.requires_grad and t.dtype == torch.float32
                out[f"{tag}/loaded/{k}"] = t.detach().numpy().copy()
            out[f"{tag}/loaded/active_sh_degree"] = np.array(m2.active_sh_degree)
        print(tag, "fields", len(arr.dtype.names), "itemsize", arr.dtype.itemsize, "rows", arr.shape[0],
              {k: out[f"{tag}/loaded/{k}"].shape for k in ("f_dc", "f_rest")})
    np.savez_compressed(os.path.join(OUT, "ply_layout.npz"), **out)
    print("wrote ply_layout.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
