"""Generate tests/golden/images.npz: image files and what the REFERENCE's decoder makes of them.

The expected pixels come from the reference's vendored stb_image (thirdparty/stb_image/stb_image.h,
the decoder behind load_image, src/vk_loader.cpp:81-160) compiled and run in this container by
`make -C oracle ref` (oracle/_ref/stb_decode).  The files themselves are made here: hand-assembled PNGs
(every colour type / bit depth / filter / interlace), Pillow-encoded PNGs and JPEGs (baseline and
progressive, 4:4:4 / 4:2:2 / 4:2:0 / greyscale, odd extents, restart markers), and the minor formats
stb_image also takes (BMP, TGA, PGM/PPM, GIF, PSD, Softimage PIC, Radiance HDR:
tests/golden/minor_image_cases.py).  Run where
/root/reference exists; the .npz travels with the repository, the reference does not.

    python tests/make_golden_images.py
"""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gltf_loader as TL  # noqa: E402  (the hand-assembled PNGs)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import minor_image_cases  # noqa: E402  (BMP, TGA, PNM, GIF, PSD, PIC, HDR)

STB = os.path.join(ROOT, "oracle", "_ref", "stb_decode")


def pillow_cases(rng):
    cases = []

    def noise(w, h, c):
        # smooth-ish content so that JPEG has something to quantise, plus sharp edges
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        base = np.stack([128 + 100 * np.sin(x / (3 + k) + k) * np.cos(y / (4 + k)) for k in range(c)], axis=2)
        base += rng.normal(0, 12, base.shape)
        base[h // 3: h // 2, w // 4: w // 2] = 255 - base[h // 3: h // 2, w // 4: w // 2]
        return np.clip(base, 0, 255).astype(np.uint8)

    def save(img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        return b.getvalue()

    rgb = Image.fromarray(noise(37, 23, 3), "RGB")
    cases.append(("png_pillow_rgb", save(rgb, "PNG")))
    cases.append(("png_pillow_rgba", save(Image.fromarray(noise(16, 19, 4), "RGBA"), "PNG")))
    cases.append(("png_pillow_l", save(Image.fromarray(noise(21, 8, 1)[..., 0], "L"), "PNG")))
    cases.append(("png_pillow_la", save(Image.fromarray(noise(9, 14, 2), "LA"), "PNG")))
    cases.append(("png_pillow_p", save(rgb.quantize(17), "PNG")))
    cases.append(("png_pillow_1bit", save(Image.fromarray((noise(19, 7, 1)[..., 0] > 128)), "PNG")))
    cases.append(("png_pillow_16", save(Image.fromarray((noise(12, 10, 1)[..., 0].astype(np.uint16) * 257 + 31)), "PNG")))
    for name, (w, h) in (("a", (64, 48)), ("b", (33, 17)), ("c", (8, 8)), ("d", (1, 1)), ("e", (50, 70))):
        im = Image.fromarray(noise(w, h, 3), "RGB")
        cases.append((f"jpg_444_{name}", save(im, "JPEG", quality=90, subsampling=0)))
        cases.append((f"jpg_422_{name}", save(im, "JPEG", quality=75, subsampling=1)))
        cases.append((f"jpg_420_{name}", save(im, "JPEG", quality=60, subsampling=2)))
        cases.append((f"jpg_prog_{name}", save(im, "JPEG", quality=80, subsampling=2, progressive=True)))
        cases.append((f"jpg_prog444_{name}", save(im, "JPEG", quality=95, subsampling=0, progressive=True)))
    g = Image.fromarray(noise(45, 29, 1)[..., 0], "L")
    cases.append(("jpg_grey", save(g, "JPEG", quality=85)))
    cases.append(("jpg_grey_prog", save(g, "JPEG", quality=85, progressive=True)))
    cases.append(("jpg_opt", save(Image.fromarray(noise(40, 40, 3), "RGB"), "JPEG", quality=50, optimize=True)))
    cases.append(("jpg_q100", save(Image.fromarray(noise(24, 24, 3), "RGB"), "JPEG", quality=100, subsampling=0)))
    cases.append(("jpg_q5", save(Image.fromarray(noise(32, 24, 3), "RGB"), "JPEG", quality=5)))
    big = Image.fromarray(noise(70, 50, 3), "RGB")
    cases.append(("jpg_rst_blocks", save(big, "JPEG", quality=80, subsampling=2, restart_marker_blocks=3)))
    cases.append(("jpg_rst_rows", save(big, "JPEG", quality=80, subsampling=0, restart_marker_rows=1)))
    cases.append(("jpg_rst_prog", save(big, "JPEG", quality=70, subsampling=1, progressive=True, restart_marker_blocks=5)))
    cases.append(("jpg_411", save(big, "JPEG", quality=85, subsampling="4:1:1") if hasattr(Image, "core") else b""))
    return cases


def main():
    if not os.path.exists(STB):
        raise SystemExit("oracle/_ref/stb_decode is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(11)
    cases = [(f"png_hand_{k}", png) for k, (png, _exp) in enumerate(TL.make_test_images(np.random.default_rng(7)))]
    cases += pillow_cases(rng)
    cases += minor_image_cases.all_cases(np.random.default_rng(23))
    names, blobs, dims, pixels = [], [], [], []
    refused = minor_image_cases.refused_cases(np.random.default_rng(29))
    with tempfile.TemporaryDirectory() as tmp:
        def reference(name, data):
            src, dst = os.path.join(tmp, name), os.path.join(tmp, name + ".bin")
            with open(src, "wb") as f:
                f.write(data)
            r = subprocess.run([STB, src, dst], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            return (r.returncode == 0), r.stdout.strip(), dst

        for name, data in cases:
            ok, said, dst = reference(name, data)
            if not ok:
                print(f"{name}: the reference's decoder refuses it ({said}): skipped")
                continue
            raw = np.fromfile(dst, dtype=np.uint8)
            w, h = np.frombuffer(raw[:8].tobytes(), dtype=np.uint32)
            names.append(name)
            blobs.append(np.frombuffer(data, dtype=np.uint8))
            dims.append((int(w), int(h)))
            pixels.append(raw[8:])
        for name, data in refused:
            ok, said, _ = reference(name, data)
            if ok:
                raise SystemExit(f"{name}: meant to be refused, but the reference's decoder takes it")
            print(f"{name}: refused as intended ({said})")
    out = os.path.join(ROOT, "tests", "golden", "images.npz")
    np.savez_compressed(out, names=np.array(names), dims=np.array(dims, dtype=np.uint32),
                        file_sizes=np.array([b.size for b in blobs], dtype=np.uint32), files=np.concatenate(blobs),
                        pixels=np.concatenate(pixels), refused_names=np.array([n for n, _ in refused]),
                        refused_sizes=np.array([len(d) for _, d in refused], dtype=np.uint32),
                        refused_files=np.frombuffer(b"".join(d for _, d in refused), dtype=np.uint8))
    print(f"{len(names)} images + {len(refused)} refused files -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
