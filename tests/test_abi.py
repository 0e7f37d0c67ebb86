"""The drop-in boundary: both libraries export every symbol include/svr.h declares, the ctypes
structs have the reference's layouts, and the product fails loudly instead of falling back.
No compute calls here (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import __graft_entry__ as g

pkg = g.load_package()
A = pkg.abi
HEADER = os.path.join(g.ROOT, "include", "svr.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svr_[a-z0-9_]+)\s*\(", text)))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, stdout=subprocess.PIPE, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_header_symbols_match_binding():
    assert declared_symbols() == sorted(A.SYMBOLS)


def test_product_library_exports_the_abi():
    g.build()  # hipcc cross-compiles gfx950 here; also builds the oracle
    assert os.path.exists(pkg.PRODUCT_LIBRARY)
    missing = set(declared_symbols()) - exported(pkg.PRODUCT_LIBRARY)
    assert not missing, missing
    lib = pkg.load_product_library()
    assert lib.backend == "hip-gfx950"


def test_oracle_exports_the_same_abi(oracle):
    missing = set(declared_symbols()) - exported(oracle.path)
    assert not missing, missing
    assert oracle.backend == "cpu-oracle"


def test_product_never_links_the_oracle():
    out = subprocess.run(["ldd", pkg.PRODUCT_LIBRARY], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(g.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "libsvr_oracle" not in text and "svr_testlib" not in text, f"{f} references the oracle"


def test_struct_layouts():
    # src/vk_types.h:97-125, src/vk_loader.h:11-15, src/vk_engine.h:29-38
    assert C.sizeof(A.SvrVertex) == 48
    assert [getattr(A.SvrVertex, n).offset for n in ("position", "uv_x", "normal", "uv_y", "color")] == [0, 12, 16, 28, 32]
    assert C.sizeof(A.SvrSceneData) == 240
    assert [getattr(A.SvrSceneData, n).offset for n in ("view", "proj", "viewproj", "ambient_color",
                                                          "sunlight_direction", "sunlight_color")] == [0, 64, 128, 192, 208, 224]
    assert C.sizeof(A.SvrBounds) == 28
    assert C.sizeof(A.SvrRenderObject) == 108 and A.SvrRenderObject.transform.offset == 44
    assert A.VERTEX_DTYPE.itemsize == 48 and A.RENDER_OBJECT_DTYPE.itemsize == 108
    assert C.sizeof(A.SvrStats) == 80


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(pkg, "PRODUCT_LIBRARY", str(tmp_path / "libsvr_hip.so"))
    monkeypatch.setattr(pkg, "_product", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.load_product_library()


def test_no_gpu_is_an_error_not_a_fallback():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = pkg.load_product_library()
    with pytest.raises(pkg.SvrError) as ei:
        lib.create(64, 64)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)
