"""The host-side mirror (acgpathtracing_amd/host: our own OBJ/MTL parser, Camera, Trackball) against
golden vectors produced by the reference's own TinyObjWrapper / sutil code (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import acgpathtracing_amd as pt
from acgpathtracing_amd import _native

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))


def mats_bits(obj):
    m = obj.getMaterials()
    if len(m) == 0:
        return np.zeros(0, np.uint32)
    import ctypes
    return np.frombuffer(ctypes.string_at(ctypes.addressof(m), ctypes.sizeof(m)), np.uint32)


@pytest.mark.parametrize("k", range(len(G["obj_names"])))
def test_obj_ingest_matches_reference_wrapper(built, k):
    rel = str(G["obj_names"][k])
    obj = pt.TinyObjWrapper(os.path.join(ROOT, rel))
    assert obj.dataLoaded
    assert np.array_equal(obj.getVerticesFloat().view(np.uint32), G["obj%d_verts" % k].view(np.uint32)), "vertices (bit-exact, tinyobj's decimal parser)"
    assert np.array_equal(obj.getIndexBuffer(), G["obj%d_idx" % k]), "index buffer (quad split / ear clipping order)"
    assert np.array_equal(obj.getMaterialIndices(), G["obj%d_mat_ids" % k]), "per-triangle material ids (0xFFFFFFFF = none)"
    assert np.array_equal(mats_bits(obj), G["obj%d_mats" % k]), "Material records incl. BSDF-by-name"


def test_obj_semantics(built, tmp_path):
    obj = pt.TinyObjWrapper(os.path.join(ROOT, "tests/golden/obj/quads_ngons.obj"))
    ids = obj.getMaterialIndices()
    assert ids[-1] == 0xFFFFFFFF and ids[-2] == 0xFFFFFFFF          # unknown usemtl -> tinyobj's -1
    kinds = [m.bsdfType for m in obj.getMaterials()]
    assert kinds == [pt.BSDF_DIFFUSE, pt.BSDF_METALLIC, pt.BSDF_REFRACTION]   # "Refractive" is tested before "Metallic"
    assert obj.getVerticesFloat().reshape(-1, 4)[:, 3].tolist() == [1.0] * 19
    # a zero vertex index is a parse error, like tinyobj (load fails, nothing is produced)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")
    o = pt.TinyObjWrapper(str(bad))
    assert not o.dataLoaded and o.getIndexBuffer().size == 0 and "Failed to parse" in o.error
    assert not pt.TinyObjWrapper(str(tmp_path / "missing.obj")).dataLoaded
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\n")
    o = pt.TinyObjWrapper(str(empty))
    assert o.dataLoaded and o.getIndexBuffer().size == 0 and o.getNumMaterials() == 0


TRICKY_OBJ = """# chunk boundaries may fall anywhere in here
mtllib quads_ngons.mtl
v 0 0 0
v 1 0 0\r
v 1 1 0
vt 0 0
vn 0 0 1
v 0 1 0
usemtl red
f 1 2 3
f -4 -3 -2 -1
f 1/1 2/1 3/1
f 1//1 2//1 4//1
f 1/1/1 2/1/1 3/1/1 4/1/1
   f 2 3 4

g second
v 2 0 0
v 2 1 0.5
v 3 1 0
v 3 0 -0.5
v 2.5 -1 0
f 5 6 7 8 9
usemtl does_not_exist
f -1 -2 -3
o third
f 5 7 10 11
v 4 0 0
v 4 1 0
usemtl red
f 9 8 7
usemtlX
f 1 3 4
s off
f 1 2"""


def _load_with(monkeypatch, path, threads, chunk_bytes):
    monkeypatch.setenv("ACGPT_OBJ_THREADS", str(threads))
    monkeypatch.setenv("ACGPT_OBJ_CHUNK_BYTES", str(chunk_bytes))
    return pt.TinyObjWrapper(path)


@pytest.mark.parametrize("chunk_bytes,threads", [(1, 4), (5, 3), (16, 2), (37, 4), (200, 8)])
def test_obj_chunked_ingest_equals_single_pass(built, tmp_path, monkeypatch, chunk_bytes, threads):
    """The loader cuts the file into chunks for its parallel passes; wherever the cuts fall, vertices, indices,
    material ids, warnings and the first error (with its line number) are those of one sequential pass."""
    import shutil
    shutil.copy(os.path.join(ROOT, "tests/golden/obj/quads_ngons.mtl"), tmp_path / "quads_ngons.mtl")
    good = tmp_path / "tricky.obj"
    good.write_bytes(TRICKY_OBJ.encode().replace(b"\\r", b"\r"))
    crlf = tmp_path / "tricky_crlf.obj"
    crlf.write_bytes(TRICKY_OBJ.encode().replace(b"\\r", b"").replace(b"\n", b"\r\n") + b"\r\n")
    bad = tmp_path / "bad.obj"
    bad.write_bytes(("\n".join(TRICKY_OBJ.replace("\\r", "").split("\n")[:24]) + "\nf 1 2 -40\nf 1 2 3\n").encode())
    bad_vt = tmp_path / "bad_vt.obj"
    bad_vt.write_bytes(b"v 0 0 0\nv 1 0 0\nv 0 1 0\n\n# c\nf 1/-1 2/-1 3/-1\n")      # relative vt index with no vt defined
    for path in (good, crlf, bad, bad_vt, os.path.join(pt.SCENES, "cornell_box.obj"), os.path.join(ROOT, "tests/golden/obj/quads_ngons.obj")):
        one = _load_with(monkeypatch, str(path), 1, 1 << 30)
        many = _load_with(monkeypatch, str(path), threads, chunk_bytes)
        assert one.dataLoaded == many.dataLoaded, path
        assert one.error == many.error and one.warning == many.warning, path
        assert np.array_equal(one.getVerticesFloat().view(np.uint32), many.getVerticesFloat().view(np.uint32)), path
        assert np.array_equal(one.getIndexBuffer(), many.getIndexBuffer()), path
        assert np.array_equal(one.getMaterialIndices(), many.getMaterialIndices()), path
        assert np.array_equal(mats_bits(one), mats_bits(many)), path
    one = _load_with(monkeypatch, str(good), 1, 1 << 30)
    assert one.dataLoaded and one.getIndexBuffer().size // 3 == 16 and "Degenerated face" in one.warning
    assert one.getMaterialIndices().tolist().count(0xFFFFFFFF) == 4
    assert "Line 25" in _load_with(monkeypatch, str(bad), 1, 1 << 30).error
    assert "Line 6" in _load_with(monkeypatch, str(bad_vt), 3, 4).error


def test_camera_matches_reference(built):
    for c, want in zip(G["camera_in"], G["camera_uvw"]):
        cam = pt.Camera(c[0:3], c[3:6], c[6:9], float(c[9]), float(c[10]))
        got = np.concatenate(cam.UVWFrame())
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    cam = pt.initCamera()
    assert cam.eye() == (278.0, 273.0, -900.0) and cam.lookat() == (278.0, 273.0, 330.0) and cam.fovY() == 35.0


def test_trackball_matches_reference(built):
    L = _native.host()
    ev = G["trackball_events"]; off = 0
    for cfg, n, want in zip(G["trackball_cfg"], G["trackball_event_counts"], G["trackball_out"]):
        c = G["camera_in"][cfg[0]]
        e = np.ascontiguousarray(ev[off:off + n], np.int32); off += n
        out = np.zeros(9, np.float32)
        eye = np.ascontiguousarray(c[0:3]); look = np.ascontiguousarray(c[3:6]); up = np.ascontiguousarray(c[6:9])
        L.pth_trackball_script(eye.ctypes.data, look.ctypes.data, up.ctypes.data, float(c[9]), float(c[10]), int(cfg[1]), 10.0, int(cfg[2]),
                               512, 512, e.ctypes.data, int(n), out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_key_toggle_state_machine():
    """PathTracerMain.cpp:100-141: every toggle resets accumulation; maxDepth clamps to [1, 28]."""
    s = pt.PathTracerState()
    s.params.maxDepth = 27
    for key, field, want in (("0", "useDirectLighting", 1), ("0", "useDirectLighting", 0), ("1", "useImportanceSampling", 1)):
        s.refreshAccumulationBuffer = False
        pt.keyCallback(s, key)
        assert getattr(s.params, field) == want and s.refreshAccumulationBuffer
    pt.keyCallback(s, "UP"); pt.keyCallback(s, "UP"); pt.keyCallback(s, "UP")
    assert s.params.maxDepth == 28
    for _ in range(40):
        pt.keyCallback(s, "DOWN")
    assert s.params.maxDepth == 1
    s.refreshAccumulationBuffer = False
    pt.keyCallback(s, "R")
    assert s.refreshAccumulationBuffer


def test_image_writers(built, tmp_path):
    """PPM / PNG writers: bottom-left origin in memory, top-down in the file, alpha dropped
    (sutil::saveImage, sutil/sutil.cpp:542-655)."""
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    img[0, :, :3] = (255, 0, 0)                 # bottom row red -> last row of the file
    for name in ("a.ppm", "a.png"):
        path = str(tmp_path / name)
        pt.saveImage(path, img)
        got = np.asarray(Image.open(path).convert("RGB"))
        assert got.shape == (37, 53, 3)
        assert np.array_equal(got, img[::-1, :, :3])
        assert tuple(got[-1, 0]) == (255, 0, 0)
    with pytest.raises(pt.PathTracerError):
        pt.saveImage(str(tmp_path / "a.bmp"), img)
