"""host/svr_image.h (with svr_png.h and svr_jpeg.h) against the reference's own decoder.

tests/golden/images.npz holds image files and the RGBA8 pixels the reference's vendored stb_image
(stbi_load(..., 4), the call behind load_image, src/vk_loader.cpp:94) makes of them; it was produced by
tests/make_golden_images.py with oracle/_ref/stb_decode (the reference header compiled in place).  The
C++ host's decoders must reproduce every image byte for byte: PNG and JPEG in every coding the reference
takes, and the minor formats stb_image also accepts (BMP, TGA, PGM/PPM, GIF, PSD, Softimage PIC, Radiance
HDR; the files come from tests/golden/minor_image_cases.py).  Files the reference's decoder refuses must
be refused as well (load_image then falls back to the error checkerboard, src/vk_loader.cpp:151).  CPU only."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as g

HOST_DIR = os.path.join(g.PKG_DIR, "host")
GOLD = os.path.join(g.ROOT, "tests", "golden", "images.npz")


def cases():
    z = np.load(GOLD)
    names, dims, sizes = [str(n) for n in z["names"]], z["dims"], z["file_sizes"]
    files, pixels = z["files"], z["pixels"]
    fo = po = 0
    out = []
    for name, (w, h), n in zip(names, dims, sizes):
        out.append((name, files[fo:fo + n].tobytes(), int(w), int(h), pixels[po:po + int(w) * int(h) * 4].reshape(int(h), int(w), 4)))
        fo += int(n)
        po += int(w) * int(h) * 4
    return out


CASES = cases()


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_decoder_matches_stb_image(tmp_path, name):
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    _, data, w, h, expect = next(c for c in CASES if c[0] == name)
    f = tmp_path / name
    f.write_bytes(data)
    r = subprocess.run([os.path.join(HOST_DIR, "svr_demo"), "--png", str(f), "--dump", str(tmp_path / "out")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.split()[:3] == ["png", str(w), str(h)], r.stdout
    got = np.fromfile(str(tmp_path / "out.rgba"), dtype=np.uint8).reshape(h, w, 4)
    if not np.array_equal(got, expect):
        d = np.abs(got.astype(int) - expect.astype(int))
        raise AssertionError(f"{name}: {int((d > 0).sum())} of {d.size} bytes differ, max |diff| {int(d.max())}, "
                             f"first at {np.argwhere(d > 0)[:3].tolist()}")


def refused():
    z = np.load(GOLD)
    out, at = [], 0
    for name, n in zip(z["refused_names"], z["refused_sizes"]):
        out.append((str(name), z["refused_files"][at:at + int(n)].tobytes()))
        at += int(n)
    return out


REFUSED = refused()


@pytest.mark.parametrize("name", [c[0] for c in REFUSED])
def test_decoder_refuses_what_stb_image_refuses(tmp_path, name):
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    data = next(c[1] for c in REFUSED if c[0] == name)
    f = tmp_path / name
    f.write_bytes(data)
    r = subprocess.run([os.path.join(HOST_DIR, "svr_demo"), "--png", str(f), "--dump", str(tmp_path / "out")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1, r.stdout


def test_every_minor_format_is_exercised(tmp_path):
    """The probe order: each file is taken by the decoder its name says."""
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    seen = set()
    for name, data, *_ in CASES:
        kind = name.replace("pillow_", "").split("_")[0]
        want = {"jpg": "jpeg", "ppm": "pnm", "pgm": "pnm"}.get(kind, kind)
        f = tmp_path / "image"
        f.write_bytes(data)
        r = subprocess.run([os.path.join(HOST_DIR, "svr_demo"), "--png", str(f)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and r.stdout.split()[3] == want, (name, r.stdout)
        seen.add(want)
    assert seen == {"png", "jpeg", "bmp", "tga", "pnm", "gif", "psd", "pic", "hdr"}
