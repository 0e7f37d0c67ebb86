"""Test files in the minor formats stb_image takes besides PNG and JPEG: BMP, TGA, binary PGM/PPM, GIF, PSD,
Softimage PIC and Radiance HDR.  Hand-assembled variant by variant (header versions, bit depths, run-length
forms, interlace, palettes, the corners where stb_image departs from the formats' specifications), plus the
same formats out of Pillow's encoders.  Used by tests/make_golden_images.py, which runs the reference's
decoder over them to produce tests/golden/images.npz.
"""
import io
import struct

import numpy as np


def blocky(rng, h, w, c, levels=256):
    """Random content with horizontal runs in it (for the run-length coders)."""
    a = rng.integers(0, levels, (h, w, c))
    for _ in range(max(1, h * w // 12)):
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        a[y, x:x + int(rng.integers(2, 9))] = a[y, x]
    return a.astype(np.uint8)


def pad4(row):
    return row + bytes((-len(row)) & 3)


def pack_bits(values, bits):
    out, acc, n = bytearray(), 0, 0
    for v in values:
        acc = (acc << bits) | int(v)
        n += bits
        while n >= 8:
            out.append((acc >> (n - 8)) & 0xff)
            n -= 8
        acc &= (1 << n) - 1
    if n:
        out.append((acc << (8 - n)) & 0xff)
    return bytes(out)


# ---- BMP ------------------------------------------------------------------------------------------------
def bmp_file(w, h, bpp, body, header=40, compress=0, masks=(), palette=None, top_down=False, gap=0):
    """body: the pixel rows as stored (already padded); palette: list of (r, g, b)."""
    pal = b""
    if palette is not None:
        pal = b"".join(bytes((b_, g_, r_)) + (b"" if header == 12 else b"\0") for r_, g_, b_ in palette)
    if header == 12:
        info = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        info = struct.pack("<IiiHHIIiiII", header, w, -h if top_down else h, 1, bpp, compress, len(body), 2835, 2835, 0, 0)
        four = (tuple(masks) + (0, 0, 0, 0))[:4]
        if header == 40:
            info += b"".join(struct.pack("<I", m) for m in masks[:3])  # BI_BITFIELDS: the masks follow the header
        elif header == 56:
            info += b"".join(struct.pack("<I", m) for m in four)
        else:
            info += b"".join(struct.pack("<I", m) for m in four) + b"BGRs" + bytes(48) + (bytes(16) if header == 124 else b"")
    offset = 14 + len(info) + len(pal) + gap
    return b"BM" + struct.pack("<IHHI", offset + len(body), 0, 0, offset) + info + pal + bytes(gap) + body


def bmp_cases(rng):
    cases = []
    pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(256)]
    for bpp, (w, h) in ((1, (13, 5)), (1, (32, 3)), (4, (7, 6)), (4, (10, 2)), (8, (9, 7)), (8, (4, 4))):
        idx = rng.integers(0, 1 << bpp, (h, w))
        body = b"".join(pad4(pack_bits(r, bpp)) for r in idx)
        cases.append((f"bmp_pal{bpp}_{w}x{h}", bmp_file(w, h, bpp, body, palette=pal[:1 << bpp])))
    idx = rng.integers(0, 5, (6, 11))
    body = b"".join(pad4(bytes(r.tolist())) for r in idx)
    cases.append(("bmp_pal8_short_palette_gap", bmp_file(11, 6, 8, body, palette=pal[:5], gap=12)))
    cases.append(("bmp_pal8_topdown", bmp_file(11, 6, 8, body, palette=pal[:5], top_down=True)))
    # a 12-byte header: stb_image sizes the palette 4 entries short ((offset - 14 - 24) / 3), so only indices below 12 are defined
    cases.append(("bmp_os2_pal4", bmp_file(5, 3, 4, b"".join(pad4(pack_bits(r, 4)) for r in rng.integers(0, 12, (3, 5))), header=12, palette=pal[:16])))
    cases.append(("bmp_os2_24", bmp_file(3, 2, 24, b"".join(pad4(rng.integers(0, 256, 9, dtype=np.uint8).tobytes()) for _ in range(2)), header=12)))
    for w, h in ((5, 4), (8, 3), (1, 1)):
        body = b"".join(pad4(rng.integers(0, 256, 3 * w, dtype=np.uint8).tobytes()) for _ in range(h))
        cases.append((f"bmp_24_{w}x{h}", bmp_file(w, h, 24, body)))
    px16 = rng.integers(0, 65536, (5, 7), dtype=np.uint16)
    body16 = b"".join(pad4(r.astype("<u2").tobytes()) for r in px16)
    cases.append(("bmp_16_555", bmp_file(7, 5, 16, body16)))
    cases.append(("bmp_16_565", bmp_file(7, 5, 16, body16, compress=3, masks=(0xf800, 0x07e0, 0x001f))))
    cases.append(("bmp_16_4444_v4", bmp_file(7, 5, 16, body16, header=108, compress=3, masks=(0x0f00, 0x00f0, 0x000f, 0xf000))))
    cases.append(("bmp_16_1555_v5", bmp_file(7, 5, 16, body16, header=124, compress=3, masks=(0x7c00, 0x03e0, 0x001f, 0x8000))))
    cases.append(("bmp_16_v4_rgb", bmp_file(7, 5, 16, body16, header=108)))
    cases.append(("bmp_16_332", bmp_file(7, 5, 16, body16, compress=3, masks=(0x00e0, 0x001c, 0x0003))))
    px32 = rng.integers(0, 1 << 32, (4, 6), dtype=np.uint32)
    body32 = b"".join(r.astype("<u4").tobytes() for r in px32)
    cases.append(("bmp_32_rgb", bmp_file(6, 4, 32, body32)))
    cases.append(("bmp_32_rgb_alpha0", bmp_file(6, 4, 32, b"".join((r & 0x00ffffff).astype("<u4").tobytes() for r in px32))))
    cases.append(("bmp_32_bitfields", bmp_file(6, 4, 32, body32, compress=3, masks=(0x0ff00000, 0x0000ff00, 0x000000ff))))
    cases.append(("bmp_32_bitfields_rgba_v4", bmp_file(6, 4, 32, body32, header=108, compress=3, masks=(0xff000000, 0x00ff0000, 0x0000ff00, 0x000000ff))))
    cases.append(("bmp_32_v5_standard", bmp_file(6, 4, 32, body32, header=124, compress=3, masks=(0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000))))
    cases.append(("bmp_32_v5_rgb_topdown", bmp_file(6, 4, 32, body32, header=124, top_down=True)))
    cases.append(("bmp_32_56", bmp_file(6, 4, 32, body32, header=56)))
    cases.append(("bmp_32_high_fields", bmp_file(6, 4, 32, body32, compress=3, masks=(0xfe000000, 0x01f80000, 0x0007c000))))
    return cases


# ---- TGA ------------------------------------------------------------------------------------------------
def tga_file(w, h, image_type, bpp, body, cmap=None, cmap_bits=0, cmap_first=0, descriptor=0, ident=b""):
    n_map = 0 if not cmap else len(cmap) // ((cmap_bits + 7) // 8)
    head = struct.pack("<BBBHHBHHHHBB", len(ident), 1 if cmap else 0, image_type, cmap_first, n_map, cmap_bits, 0, 0, w, h, bpp, descriptor)
    return head + ident + bytes(cmap_first) + (cmap or b"") + body


def tga_rle(pixels, size, rng):
    """pixels: bytes, `size` bytes per pixel; packets may cross rows."""
    px = [pixels[i:i + size] for i in range(0, len(pixels), size)]
    out, i = bytearray(), 0
    while i < len(px):
        run = 1
        while i + run < len(px) and px[i + run] == px[i] and run < 128:
            run += 1
        if run >= 2 or rng.random() < 0.2:
            out += bytes((0x80 | (run - 1),)) + px[i]
            i += run
        else:
            lit = min(int(rng.integers(1, 9)), len(px) - i)
            out += bytes((lit - 1,)) + b"".join(px[i:i + lit])
            i += lit
    return bytes(out)


def tga_cases(rng):
    cases = []
    for comp, bpp, typ in ((1, 8, 3), (2, 16, 3), (3, 24, 2), (4, 32, 2)):
        a = blocky(rng, 6, 9, comp)
        cases.append((f"tga_raw_{bpp}_t{typ}", tga_file(9, 6, typ, bpp, a.tobytes(), descriptor=8 if comp == 4 else 0)))
        cases.append((f"tga_raw_{bpp}_t{typ}_topdown", tga_file(9, 6, typ, bpp, a.tobytes(), descriptor=0x20, ident=b"hello")))
        cases.append((f"tga_rle_{bpp}_t{typ}", tga_file(9, 6, typ + 8, bpp, tga_rle(a.tobytes(), comp, rng))))
        cases.append((f"tga_rle_{bpp}_t{typ}_topdown", tga_file(9, 6, typ + 8, bpp, tga_rle(a.tobytes(), comp, rng), descriptor=0x20)))
    a16 = blocky(rng, 5, 8, 2)
    for bpp in (15, 16):
        cases.append((f"tga_raw_{bpp}_555", tga_file(8, 5, 2, bpp, a16.tobytes())))
        cases.append((f"tga_rle_{bpp}_555", tga_file(8, 5, 10, bpp, tga_rle(a16.tobytes(), 2, rng), descriptor=0x20)))
    idx = blocky(rng, 7, 10, 1, levels=12)
    for bits in (8, 15, 16, 24, 32):
        size = (bits + 7) // 8
        cmap = rng.integers(0, 256, 12 * size, dtype=np.uint8).tobytes()
        cases.append((f"tga_map{bits}_raw", tga_file(10, 7, 1, 8, idx.tobytes(), cmap=cmap, cmap_bits=bits)))
        cases.append((f"tga_map{bits}_rle", tga_file(10, 7, 9, 8, tga_rle(idx.tobytes(), 1, rng), cmap=cmap, cmap_bits=bits, descriptor=0x20)))
    cmap = rng.integers(0, 256, 12 * 3, dtype=np.uint8).tobytes()
    cases.append(("tga_map24_index16", tga_file(10, 7, 1, 16, idx.astype("<u2").tobytes(), cmap=cmap, cmap_bits=24)))
    cases.append(("tga_map24_first3", tga_file(10, 7, 1, 8, idx.tobytes(), cmap=cmap, cmap_bits=24, cmap_first=3)))
    wild = idx.copy()
    wild[2, 3:6] = 200  # beyond the 12-entry map
    cases.append(("tga_map24_index_out_of_range", tga_file(10, 7, 1, 8, wild.tobytes(), cmap=cmap, cmap_bits=24)))
    return cases


# ---- PGM / PPM ------------------------------------------------------------------------------------------
def pnm_cases(rng):
    cases = []
    g = rng.integers(0, 256, (5, 7), dtype=np.uint8)
    c = rng.integers(0, 256, (4, 6, 3), dtype=np.uint8)
    cases.append(("pnm_p5", b"P5\n7 5\n255\n" + g.tobytes()))
    cases.append(("pnm_p6", b"P6 6 4 255 " + c.tobytes()))
    cases.append(("pnm_p6_comments", b"P6\n# made by hand\n6 # width\n4\n#max\n255\n" + c.tobytes()))
    cases.append(("pnm_p5_maxval15", b"P5\n7 5\n15\n" + (g & 15).tobytes()))
    g16 = rng.integers(0, 65536, (3, 5), dtype=np.uint16)
    cases.append(("pnm_p5_16", b"P5\n5 3\n65535\n" + g16.astype(">u2").tobytes()))
    c16 = rng.integers(0, 1024, (2, 3, 3), dtype=np.uint16)
    cases.append(("pnm_p6_16", b"P6\n3 2\n1023\n" + c16.astype(">u2").tobytes()))
    cases.append(("pnm_p6_crlf", b"P6\r\n6\t4\r\n255\r" + c.tobytes()))
    cases.append(("pnm_p6_trailing", b"P6 6 4 255\n" + c.tobytes() + b"trailing bytes"))
    return cases


# ---- GIF ------------------------------------------------------------------------------------------------
def gif_lzw(indices, min_size, early_clear=0):
    clear, stop = 1 << min_size, (1 << min_size) + 1
    out, acc, nbits = bytearray(), 0, 0

    def put(code, size):
        nonlocal acc, nbits
        acc |= code << nbits
        nbits += size
        while nbits >= 8:
            out.append(acc & 0xff)
            acc >>= 8
            nbits -= 8

    def fresh():
        return {(i,): i for i in range(clear)}, min_size + 1, clear + 2

    table, size, nxt = fresh()
    put(clear, size)
    cur, emitted = (), 0
    for v in indices:
        v = int(v)
        if cur + (v,) in table:
            cur = cur + (v,)
            continue
        put(table[cur], size)
        emitted += 1
        if nxt < 4096:
            table[cur + (v,)] = nxt
            nxt += 1
            if nxt > (1 << size) and size < 12:
                size += 1
        if nxt >= 4096 or (early_clear and emitted % early_clear == 0):
            put(clear, size)
            table, size, nxt = fresh()
        cur = (v,)
    if cur:
        put(table[cur], size)
        if nxt < 4096:  # the decoder adds an entry (and may widen its codes) after this one too
            nxt += 1
            if nxt > (1 << size) and size < 12:
                size += 1
    put(stop, size)
    if nbits:
        out.append(acc & 0xff)
    return bytes(out)


def gif_table(entries):
    k = max(1, int(np.ceil(np.log2(len(entries)))))
    return k, b"".join(bytes(e) for e in list(entries) + [(0, 0, 0)] * ((1 << k) - len(entries)))


def gif_file(W, H, frames, global_table=None, bg=0, version=b"89a", tail=b"\x3b"):
    """frames: dicts x, y, w, h, indices, min_size [, local_table, interlace, transparent, gce, pre, block, early_clear]."""
    flags, body = 0, b""
    if global_table is not None:
        k, body = gif_table(global_table)
        flags = 0x80 | (k - 1) | 0x70
    out = b"GIF" + version + struct.pack("<HHBBB", W, H, flags, bg, 0) + body
    for f in frames:
        out += f.get("pre", b"")
        t = f.get("transparent")
        if t is not None or f.get("gce"):
            out += b"\x21\xf9\x04" + struct.pack("<BHB", (1 if t is not None else 0) | (f.get("dispose", 0) << 2), 7, t or 0) + b"\0"
        lflags, ltb = (0x40 if f.get("interlace") else 0), b""
        if f.get("local_table") is not None:
            k, ltb = gif_table(f["local_table"])
            lflags |= 0x80 | (k - 1)
        out += b"\x2c" + struct.pack("<HHHHB", f["x"], f["y"], f["w"], f["h"], lflags) + ltb
        idx = np.asarray(f["indices"]).reshape(f["h"], f["w"])
        if f.get("interlace"):
            idx = idx[[r for first, step in ((0, 8), (4, 8), (2, 4), (1, 2)) for r in range(first, f["h"], step)]]
        data = gif_lzw(idx.reshape(-1), f["min_size"], early_clear=f.get("early_clear", 0))
        out += bytes((f["min_size"],))
        block = f.get("block", 255)
        for i in range(0, len(data), block):
            out += bytes((len(data[i:i + block]),)) + data[i:i + block]
        out += b"\0"
    return out + tail


def gif_cases(rng):
    cases = []
    pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(256)]

    def frame(w, h, levels, **kw):
        return dict(x=0, y=0, w=w, h=h, indices=blocky(rng, h, w, 1, levels=levels), **kw)

    cases.append(("gif_2colour", gif_file(9, 5, [frame(9, 5, 2, min_size=2)], global_table=pal[:2])))
    cases.append(("gif_16colour", gif_file(21, 13, [frame(21, 13, 16, min_size=4)], global_table=pal[:16], version=b"87a")))
    cases.append(("gif_256colour", gif_file(40, 33, [frame(40, 33, 256, min_size=8)], global_table=pal)))
    cases.append(("gif_256_noise_tablefull", gif_file(90, 80, [dict(x=0, y=0, w=90, h=80, min_size=8, indices=rng.integers(0, 256, 7200))], global_table=pal)))
    cases.append(("gif_interlaced", gif_file(17, 19, [frame(17, 19, 8, min_size=3, interlace=True)], global_table=pal[:8])))
    cases.append(("gif_interlaced_short", gif_file(6, 3, [frame(6, 3, 4, min_size=2, interlace=True)], global_table=pal[:4])))
    cases.append(("gif_transparent", gif_file(12, 9, [frame(12, 9, 8, min_size=3, transparent=3)], global_table=pal[:8])))
    cases.append(("gif_local_table", gif_file(12, 9, [frame(12, 9, 8, min_size=3, local_table=pal[40:48])], global_table=pal[:4])))
    cases.append(("gif_local_table_only_transparent", gif_file(12, 9, [frame(12, 9, 8, min_size=3, local_table=pal[40:48], transparent=5)])))
    sub = dict(x=3, y=2, w=6, h=4, indices=blocky(rng, 4, 6, 1, levels=8), min_size=3)
    cases.append(("gif_subimage_bg0", gif_file(12, 9, [sub], global_table=pal[:8], bg=0)))
    cases.append(("gif_subimage_bg5", gif_file(12, 9, [sub], global_table=pal[:8], bg=5)))
    cases.append(("gif_subimage_bg5_transparent5", gif_file(12, 9, [dict(sub, transparent=5)], global_table=pal[:8], bg=5)))
    cases.append(("gif_subimage_interlaced_bg2", gif_file(12, 14, [dict(x=2, y=1, w=7, h=11, indices=blocky(rng, 11, 7, 1, levels=8), min_size=3, interlace=True)], global_table=pal[:8], bg=2)))
    cases.append(("gif_two_frames", gif_file(12, 9, [frame(12, 9, 8, min_size=3, gce=True), frame(12, 9, 8, min_size=3, gce=True, dispose=2)], global_table=pal[:8])))
    comment = b"\x21\xfe\x05hello\x03abc\x00" + b"\x21\xff\x0bNETSCAPE2.0\x03\x01\x00\x00\x00"
    cases.append(("gif_extensions_first", gif_file(12, 9, [frame(12, 9, 8, min_size=3, pre=comment, transparent=1)], global_table=pal[:8])))
    cases.append(("gif_small_blocks_early_clear", gif_file(30, 20, [frame(30, 20, 32, min_size=5, block=7, early_clear=40)], global_table=pal[:32])))
    cases.append(("gif_unused_table_tail", gif_file(10, 6, [frame(10, 6, 8, min_size=3)], global_table=pal[:5])))
    whole = gif_file(14, 11, [frame(14, 11, 16, min_size=4)], global_table=pal[:16], bg=2)
    cases.append(("gif_truncated", whole[:len(whole) * 2 // 3]))
    return cases


# ---- PSD ------------------------------------------------------------------------------------------------
def packbits(row, rng):
    out, i = bytearray(), 0
    while i < len(row):
        run = 1
        while i + run < len(row) and row[i + run] == row[i] and run < 128:
            run += 1
        if run >= 2:
            out += bytes((257 - run, row[i]))
            i += run
        else:
            lit = min(int(rng.integers(1, 6)), len(row) - i)
            out += bytes((lit - 1,)) + bytes(row[i:i + lit])
            i += lit
        if rng.random() < 0.1:
            out.append(128)  # no-op
    return bytes(out)


def psd_file(planes, depth=8, rle=False, rng=None, resources=b""):
    """planes: (channels, h, w) uint8 or uint16."""
    ch, h, w = planes.shape
    head = b"8BPS" + struct.pack(">H6xHIIHH", 1, ch, h, w, depth, 3)
    head += struct.pack(">I", 0) + struct.pack(">I", len(resources)) + resources + struct.pack(">I", 0)
    if not rle:
        return head + struct.pack(">H", 0) + planes.astype(">u2" if depth == 16 else "u1").tobytes()
    rows = [packbits(planes[c, y].tolist(), rng) for c in range(ch) for y in range(h)]
    return head + struct.pack(">H", 1) + b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)


def psd_cases(rng):
    cases = []

    def planes(c, h, w):
        return np.moveaxis(blocky(rng, h, w, c), 2, 0).copy()

    for c in (1, 3, 4, 6):
        cases.append((f"psd_raw_{c}ch", psd_file(planes(c, 5, 7))))
        cases.append((f"psd_rle_{c}ch", psd_file(planes(c, 6, 11), rle=True, rng=rng, resources=b"8BIM" + bytes(10))))
    cases.append(("psd_raw_16bit", psd_file(rng.integers(0, 65536, (4, 3, 5), dtype=np.uint16), depth=16)))
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    a = planes(4, 16, 16)
    a[3] = ramp  # every alpha against assorted colours
    cases.append(("psd_matte_all_alphas", psd_file(a)))
    a2 = np.zeros((4, 16, 16), dtype=np.uint8)
    a2[0], a2[1], a2[2], a2[3] = 255, 0, ramp.T, ramp
    cases.append(("psd_matte_extremes", psd_file(a2)))
    return cases


# ---- Softimage PIC --------------------------------------------------------------------------------------
def pic_file(w, h, packets, rows):
    """packets: (type, channel mask); rows[y][k] = bytes of packet k on row y."""
    head = b"\x53\x80\xf6\x34" + struct.pack(">f", 0.0) + b"made by tests/golden/minor_image_cases.py".ljust(80, b"\0") + b"PICT"
    head += struct.pack(">HHfHH", w, h, 1.0, 3, 0)
    for k, (typ, mask) in enumerate(packets):
        head += bytes((1 if k + 1 < len(packets) else 0, 8, typ, mask))
    return head + b"".join(b"".join(r) for r in rows)


def pic_cases(rng):
    cases = []
    w, h = 11, 6
    rgb, alpha = blocky(rng, h, w, 3), blocky(rng, h, w, 1)

    def pixels(px, size):
        return [bytes(p) for p in np.ascontiguousarray(px).reshape(-1, size).tolist()]

    def mixed(px, size):
        out, i, px = bytearray(), 0, pixels(px, size)
        while i < len(px):
            run = 1
            while i + run < len(px) and px[i + run] == px[i] and run < 100:
                run += 1
            if run >= 2:
                out += (bytes((127 + run,)) if rng.random() < 0.7 else b"\x80" + struct.pack(">H", run)) + px[i]
                i += run
            else:
                lit = min(int(rng.integers(1, 5)), len(px) - i)
                out += bytes((lit - 1,)) + b"".join(px[i:i + lit])
                i += lit
        return bytes(out)

    def pure(px, size):
        out, i, px = bytearray(), 0, pixels(px, size)
        while i < len(px):
            run = 1
            while i + run < len(px) and px[i + run] == px[i] and run < 255:
                run += 1
            out += bytes((run,)) + px[i]
            i += run
        return bytes(out)

    cases.append(("pic_raw_rgb", pic_file(w, h, [(0, 0xe0)], [[rgb[y].tobytes()] for y in range(h)])))
    cases.append(("pic_raw_rgb_alpha", pic_file(w, h, [(0, 0xe0), (0, 0x10)], [[rgb[y].tobytes(), alpha[y].tobytes()] for y in range(h)])))
    cases.append(("pic_mixed_rgb_alpha", pic_file(w, h, [(2, 0xe0), (2, 0x10)], [[mixed(rgb[y], 3), mixed(alpha[y], 1)] for y in range(h)])))
    cases.append(("pic_pure_rgb", pic_file(w, h, [(1, 0xe0)], [[pure(rgb[y], 3)] for y in range(h)])))
    cases.append(("pic_split_channels", pic_file(w, h, [(2, 0x80), (1, 0x40), (0, 0x20)],
                                                 [[mixed(rgb[y][:, 0:1], 1), pure(rgb[y][:, 1:2], 1), rgb[y][:, 2].tobytes()] for y in range(h)])))
    cases.append(("pic_red_only", pic_file(w, h, [(2, 0x80)], [[mixed(rgb[y][:, 0:1], 1)] for y in range(h)])))
    return cases


# ---- Radiance HDR ---------------------------------------------------------------------------------------
def hdr_rgbe(rng, h, w):
    a = blocky(rng, h, w, 4)
    a[..., 3] = rng.integers(118, 140, (h, w))  # exponents around 1.0
    a[0, 0, 3] = 0
    a[h - 1, w - 1] = (255, 255, 255, 160)
    return a


def hdr_rle_row(row):
    w = row.shape[0]
    out = bytearray((2, 2, w >> 8, w & 255))
    for k in range(4):
        v, i = row[:, k].tolist(), 0
        while i < w:
            run = 1
            while i + run < w and v[i + run] == v[i] and run < 127:
                run += 1
            if run >= 3:
                out += bytes((128 + run, v[i]))
                i += run
            else:
                j = i
                while j < w and j - i < 128 and not (j + 2 < w and v[j] == v[j + 1] == v[j + 2]):
                    j += 1
                j = max(j, i + 1)
                out += bytes((j - i,)) + bytes(v[i:j])
                i = j
    return bytes(out)


def hdr_cases(rng):
    cases = []
    head = b"#?RADIANCE\n# made by hand\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n"
    a = hdr_rgbe(rng, 4, 5)
    cases.append(("hdr_flat_narrow", head + b"-Y 4 +X 5\n" + a.tobytes()))
    b = hdr_rgbe(rng, 6, 23)
    cases.append(("hdr_rle", head + b"-Y 6 +X 23\n" + b"".join(hdr_rle_row(b[y]) for y in range(6))))
    cases.append(("hdr_rgbe_signature", b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 6   +X 23\n" + b"".join(hdr_rle_row(b[y]) for y in range(6))))
    c = hdr_rgbe(rng, 3, 9)
    c[0, 0] = (200, 100, 50, 128)  # a flat file must not begin with 2, 2, <128
    cases.append(("hdr_flat_wide", head + b"-Y 3 +X 9\n" + c.tobytes()))
    every = np.zeros((32, 256, 4), dtype=np.uint8)  # every mantissa against the exponents that matter for 8 bits
    every[..., 0] = np.arange(256)[None, :]
    every[..., 1] = 255 - np.arange(256)[None, :]
    every[..., 2] = (np.arange(256)[None, :] * 7 + np.arange(32)[:, None]) & 255
    every[..., 3] = 108 + np.arange(32)[:, None]
    cases.append(("hdr_every_mantissa", head + b"-Y 32 +X 256\n" + b"".join(hdr_rle_row(every[y]) for y in range(32))))
    return cases


# ---- the same formats out of an independent encoder -----------------------------------------------------
def pillow_cases(rng):
    from PIL import Image
    cases = []

    def save(img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        return b.getvalue()

    rgb = Image.fromarray(blocky(rng, 14, 19, 3), "RGB")
    rgba = Image.fromarray(blocky(rng, 10, 13, 4), "RGBA")
    grey = Image.fromarray(blocky(rng, 9, 15, 1)[..., 0], "L")
    cases.append(("pillow_bmp_rgb", save(rgb, "BMP")))
    cases.append(("pillow_bmp_rgba", save(rgba, "BMP")))
    cases.append(("pillow_bmp_l", save(grey, "BMP")))
    cases.append(("pillow_bmp_p", save(rgb.quantize(13), "BMP")))
    cases.append(("pillow_bmp_1", save(grey.point(lambda v: 255 * (v > 128)).convert("1"), "BMP")))
    cases.append(("pillow_tga_rgb", save(rgb, "TGA")))
    cases.append(("pillow_tga_rgba_rle", save(rgba, "TGA", compression="tga_rle")))
    cases.append(("pillow_tga_l_rle", save(grey, "TGA", compression="tga_rle")))
    cases.append(("pillow_tga_la", save(Image.fromarray(blocky(rng, 6, 7, 2), "LA"), "TGA")))
    cases.append(("pillow_tga_p", save(rgb.quantize(20), "TGA")))
    cases.append(("pillow_tga_rgb_topdown", save(rgb, "TGA", orientation=1)))
    cases.append(("pillow_ppm", save(rgb, "PPM")))
    cases.append(("pillow_pgm", save(grey, "PPM")))
    cases.append(("pillow_gif", save(rgb.quantize(31), "GIF")))
    cases.append(("pillow_gif_grey", save(grey, "GIF")))
    cases.append(("pillow_gif_transparency", save(rgb.quantize(9), "GIF", transparency=4)))
    cases.append(("pillow_gif_interlace", save(rgb.quantize(17), "GIF", interlace=True)))
    return cases


def all_cases(rng):
    return (bmp_cases(rng) + tga_cases(rng) + pnm_cases(rng) + gif_cases(rng) + psd_cases(rng) + pic_cases(rng)
            + hdr_cases(rng) + pillow_cases(rng))


# ---- files the reference's decoder refuses: the host's must refuse them too ---------------------------------
def refused_cases(rng):
    cases = []
    pal = [(i, 2 * i, 3 * i) for i in range(16)]
    body8 = b"".join(pad4(bytes(r.tolist())) for r in rng.integers(0, 16, (4, 6)))
    body16 = bytes(4 * 6 * 2)
    cases.append(("bad_bmp_rle8", bmp_file(6, 4, 8, body8, compress=1, palette=pal)))
    cases.append(("bad_bmp_png_inside", bmp_file(6, 4, 24, bytes(96), compress=5)))
    cases.append(("bad_bmp_bitfields_24", bmp_file(6, 4, 24, bytes(96), compress=3, masks=(0xff0000, 0xff00, 0xff))))
    cases.append(("bad_bmp_equal_masks", bmp_file(6, 4, 16, body16, compress=3, masks=(0x1f, 0x1f, 0x1f))))
    cases.append(("bad_bmp_wide_mask", bmp_file(6, 4, 32, bytes(96), compress=3, masks=(0x3ff00000, 0x000ffc00, 0x000003ff))))
    cases.append(("bad_bmp_os2_16", bmp_file(6, 4, 16, body16, header=12)))
    cases.append(("bad_bmp_2bpp", bmp_file(6, 4, 2, b"".join(pad4(bytes(2)) for _ in range(4)), palette=pal[:4])))
    cases.append(("bad_bmp_far_offset", bmp_file(6, 4, 24, bytes(96), gap=1100)))
    cases.append(("bad_bmp_no_palette", bmp_file(6, 4, 8, body8)))
    one = dict(x=0, y=0, w=5, h=4, indices=rng.integers(0, 4, 20), min_size=2)
    cases.append(("bad_gif_no_table", gif_file(5, 4, [one])))
    cases.append(("bad_gif_outside", gif_file(5, 4, [dict(one, x=1)], global_table=pal[:4])))
    cases.append(("bad_gif_no_image", gif_file(5, 4, [], global_table=pal[:4])))
    ok = gif_file(5, 4, [one], global_table=pal[:4])
    at = ok.index(b"\x2c") + 10  # the LZW minimum code size
    cases.append(("bad_gif_code_size", ok[:at] + b"\x0d" + ok[at + 1:]))
    cases.append(("bad_gif_no_clear_code", ok[:at + 2] + bytes((ok[at + 2] & 0xf8 | 1,)) + ok[at + 3:]))
    cases.append(("bad_gif_unknown_block", ok[:ok.index(b"\x2c")] + b"\x55" + ok[ok.index(b"\x2c"):]))
    planes = np.moveaxis(blocky(rng, 4, 5, 3), 2, 0).copy()
    good = psd_file(planes)
    cases.append(("bad_psd_version2", good[:4] + b"\0\2" + good[6:]))
    cases.append(("bad_psd_cmyk", good[:24] + b"\0\4" + good[26:]))
    cases.append(("bad_psd_depth32", good[:22] + b"\0\x20" + good[24:]))
    cases.append(("bad_psd_zip", good[:38] + b"\0\2" + good[40:]))
    cases.append(("bad_psd_17_channels", good[:12] + b"\0\x11" + good[14:]))
    rle = psd_file(planes, rle=True, rng=rng)
    first = 40 + 2 * 3 * 4
    cases.append(("bad_psd_run_overflow", rle[:first] + bytes((0x81, 7)) * 40))
    rgb = blocky(rng, 3, 5, 3)
    cases.append(("bad_pic_4bit_packet", pic_file(5, 3, [(0, 0xe0)], [[rgb[y].tobytes()] for y in range(3)]).replace(b"\0\x08\0\xe0", b"\0\x04\0\xe0", 1)))
    cases.append(("bad_pic_packet_type", pic_file(5, 3, [(3, 0xe0)], [[rgb[y].tobytes()] for y in range(3)])))
    cases.append(("bad_pic_truncated", pic_file(5, 3, [(0, 0xe0)], [[rgb[y].tobytes()] for y in range(3)])[:-7]))
    cases.append(("bad_pic_overrun", pic_file(5, 3, [(2, 0x80)], [[bytes((127 + 9, 1))] for y in range(3)])))
    b = hdr_rgbe(rng, 2, 9)
    rows = b"".join(hdr_rle_row(b[y]) for y in range(2))
    cases.append(("bad_hdr_xyze", b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 9\n" + rows))
    cases.append(("bad_hdr_orientation", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 2 +X 9\n" + rows))
    cases.append(("bad_hdr_no_format", b"#?RADIANCE\nEXPOSURE=2\n\n-Y 2 +X 9\n" + rows))
    cases.append(("bad_hdr_scanline_length", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 10\n" + rows))
    cases.append(("bad_hdr_run", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 9\n" + bytes((2, 2, 0, 9, 128 + 12, 5)) + rows))
    cases.append(("bad_pnm_zero_width", b"P5 0 5 255 " + bytes(25)))
    cases.append(("bad_pnm_maxval", b"P5 5 5 70000 " + bytes(50)))
    cases.append(("bad_pnm_truncated", b"P6 5 5 255 " + bytes(70)))
    cases.append(("bad_pnm_ascii", b"P3 1 1 255 1 2 3\n"))
    cases.append(("bad_tga_empty_map", tga_file(4, 4, 1, 8, bytes(16), cmap=None, cmap_bits=24).replace(b"\0\0\1", b"\0\1\1", 1)))
    cases.append(("bad_tga_type", tga_file(4, 4, 4, 24, bytes(48))))
    cases.append(("bad_tga_bpp", tga_file(4, 4, 2, 12, bytes(48))))
    cases.append(("bad_unknown", bytes(rng.integers(2, 256, 200, dtype=np.uint8).tolist())))
    cases.append(("bad_empty", b""))
    return cases
