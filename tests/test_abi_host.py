"""CPU-side checks of the C ABI: the library loads, exports every symbol include/lmi_hip.h
declares, and its host-only entry points agree with the oracle.  No GPU compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import cv2_resize as OC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from object_detectors_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lmi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lmi_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_typed(lib):
    from object_detectors_amd import _lib
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in lmi_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_status_strings(lib):
    assert lib.lmi_version() == 102
    assert lib.lmi_status_string(0) == b"ok"
    assert b"workspace" in lib.lmi_status_string(-3)


def test_frame_struct_layout():
    from object_detectors_amd._lib import lmi_frame
    assert C.sizeof(lmi_frame) == 40 and lmi_frame.h.offset == 8 and lmi_frame.pad_t.offset == 32


def test_letterbox_geometry_matches_oracle(lib):
    rng = np.random.default_rng(0)
    sizes = [(1080, 1920, 640, 640), (2160, 3840, 640, 640), (1080, 1920, 1280, 1280), (375, 500, 640, 640), (641, 333, 640, 640),
             (100, 1000, 640, 640), (7, 5, 32, 32), (333, 777, 512, 384)]
    sizes += [tuple(int(v) for v in rng.integers(1, 3000, 2)) + tuple(int(v) * 32 for v in rng.integers(1, 41, 2)) for _ in range(400)]
    out = (C.c_int32 * 6)()
    for h, w, sh, sw in sizes:
        assert lib.lmi_letterbox_geometry(h, w, sh, sw, 0, out) == 0
        assert tuple(out) == OC.letterbox_geometry(h, w, sh, sw), (h, w, sh, sw)
        assert lib.lmi_letterbox_geometry(h, w, sh, sw, 1, out) == 0
        assert tuple(out) == (sw, sh, 0, 0, 0, 0)
    assert lib.lmi_letterbox_geometry(0, 5, 32, 32, 0, out) == -1
    assert b"non-positive" in lib.lmi_last_error()


def test_argument_validation_needs_no_gpu(lib):
    # every check below fails before any HIP call is made
    assert lib.lmi_nms_workspace_bytes(64, 8400) > 64 * 8400 * 28
    # logits (capacity x mh x mw fp32) + one 16-byte box record per slot, each rounded up to 256 bytes
    assert lib.lmi_mask_workspace_bytes(10, 160, 160) == 10 * 160 * 160 * 4 + 256
    assert lib.lmi_preprocess_u8(None, 4, 4, 3, 12, None, 0, None) == -1
    assert lib.lmi_nms_v8(None, 1, 80, 0, 8400, 0.25, 0.45, 0, 300, 30000, 7680.0, None, None, None, None, 0, None) == -1
    assert lib.lmi_process_mask(None, 1, 32, 160, 160, None, 300, 38, None, None, 1080, 1920, None, 2, 10, None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without the built extension the loader raises at once, and so does everything that binds it
    (hip_ops imports the loader at module import)."""
    import importlib
    import object_detectors_amd._lib as L
    import object_detectors_amd.hip_ops as H   # with the real library, before the path is bent
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "liblmi_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        L.load()
    with pytest.raises(ImportError, match="no CPU fallback"):
        importlib.reload(H)
    monkeypatch.undo()
    importlib.reload(H)   # back to the real library for the tests that follow
    assert H._L is L.load()


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/lmi_hip.h through `gcc -std=c99 -pedantic -Werror`, linked against liblmi_hip.so and run: the boundary
    is a C ABI a cgo / JNI / ctypes binding can sit on (host-only entry points, no GPU needed)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "object_detectors_amd")
    exe = str(tmp_path / "abi_c_client")
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_c_client.c"), "-o", exe, "-L", libdir, "-llmi_hip", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "640 360 0 0 140 140" and lines[-1] == "bad argument"
