// svr_image.h — every image format the reference's load_image can take, decoded to RGBA8.
//
// The reference hands each glTF image to stb_image (stbi_load(..., 4) / stbi_load_from_memory(..., 4),
// src/vk_loader.cpp:94, 108, 131) with no STBI_ONLY_* restriction, so besides PNG and JPEG (svr_png.h,
// svr_jpeg.h) a texture may be a BMP, GIF, PSD, Softimage PIC, binary PGM/PPM, Radiance HDR or TGA file.
// decode() below probes the formats in stb_image's order (stbi__load_main,
// thirdparty/stb_image/stb_image.h:1136-1187) and reproduces its pixels byte for byte, including the
// places where it departs from the formats' own specifications (noted at each decoder);
// tests/golden/images.npz pins all of them against the reference's decoder run in place.
//
// Each decoder writes four components directly (the reference always asks for 4), so stb_image's
// separate n-to-4 conversion pass has no counterpart here: grey g -> (g, g, g, 255), grey+alpha ->
// (g, g, g, a), RGB -> (r, g, b, 255)  (stbi__convert_format, :1754).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "svr_jpeg.h"
#include "svr_png.h"

namespace svrimg {

struct Image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;  // w*h*4
};

constexpr uint32_t MAX_DIMENSION = 1u << 24;  // STBI_MAX_DIMENSIONS

// stb_image's view of a file: reading past the end yields zeros, and the position may be moved beyond
// the end (stbi__get8 / stbi__skip / stbi__at_eof / stbi__getn, memory form).
class Cursor {
 public:
  Cursor(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  uint32_t u8() { return at_ < n_ ? p_[at_++] : 0u; }
  uint32_t le16() { uint32_t a = u8(); return a | (u8() << 8); }
  uint32_t le32() { uint32_t a = le16(); return a | (le16() << 16); }
  uint32_t be16() { uint32_t a = u8(); return (a << 8) | u8(); }
  uint32_t be32() { uint32_t a = be16(); return (a << 16) | be16(); }
  void skip(long long k) {
    if (k < 0) at_ = n_;  // a negative distance jumps to the end of the file
    else at_ += (size_t)k;
  }
  bool eof() const { return at_ >= n_; }
  size_t tell() const { return at_; }
  // all k bytes or nothing useful: what is there is copied, the return value says whether it was all
  bool take(uint8_t* dst, size_t k) {
    size_t have = at_ < n_ ? n_ - at_ : 0;
    if (have < k) {
      if (have) std::memcpy(dst, p_ + at_, have);
      at_ = n_;
      return false;
    }
    if (k) std::memcpy(dst, p_ + at_, k);
    at_ += k;
    return true;
  }

 private:
  const uint8_t* p_;
  size_t n_, at_ = 0;
};

inline bool extent_ok(long long w, long long h) { return w > 0 && h > 0 && w * h * 4 <= 0x7fffffffLL; }

inline bool start(Image& img, long long w, long long h, uint8_t fill) {
  if (!extent_ok(w, h)) return false;
  img.w = (uint32_t)w;
  img.h = (uint32_t)h;
  img.rgba.assign((size_t)w * h * 4, fill);
  return true;
}

inline void flip_rows(Image& img) {
  size_t row = (size_t)img.w * 4;
  std::vector<uint8_t> tmp(row);
  for (uint32_t y = 0; y < img.h / 2; y++) {
    uint8_t* a = &img.rgba[y * row];
    uint8_t* b = &img.rgba[(img.h - 1 - y) * row];
    std::memcpy(tmp.data(), a, row);
    std::memcpy(a, b, row);
    std::memcpy(b, tmp.data(), row);
  }
}

// ---------------------------------------------------------------------------------------------------
// BMP (stbi__bmp_load, stb_image.h:5528).  Uncompressed only: 1/4/8-bit palettes, 16/32-bit bit
// fields, 24-bit; 12-, 40-, 56-, 108- and 124-byte headers.  A 32-bit BI_RGB file whose alpha bytes
// are all zero comes out opaque.
inline bool bmp_probe(const uint8_t* p, size_t n) {
  Cursor c(p, n);
  if (c.u8() != 'B' || c.u8() != 'M') return false;
  c.skip(12);
  uint32_t hsz = c.le32();
  return hsz == 12 || hsz == 40 || hsz == 56 || hsz == 108 || hsz == 124;
}

// a masked field of `bits` bits, moved down to bit 0 and widened to 8 bits by replication
inline uint32_t bmp_field(uint32_t v, int shift, int bits) {
  static const uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
  static const uint32_t down[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
  v = shift < 0 ? v << -shift : v >> shift;
  v >>= 8 - bits;
  return (v * mul[bits]) >> down[bits];
}

inline int top_bit(uint32_t v) { return v ? 31 - __builtin_clz(v) : -1; }

inline bool bmp_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  c.skip(10);
  int offset = (int)c.le32(), hsz = (int)c.le32();
  uint32_t mr = 0, mg = 0, mb = 0, ma = 0, any_alpha = 255;
  int header_read = 14;
  if (offset < 0) return fail("bmp: bad data offset");
  int32_t w, h;
  if (hsz == 12) {
    w = (int32_t)c.le16();
    h = (int32_t)c.le16();
  } else {
    w = (int32_t)c.le32();
    h = (int32_t)c.le32();
  }
  if (c.le16() != 1) return fail("bmp: planes != 1");
  int bpp = (int)c.le16();
  auto default_masks = [&]() {  // BI_RGB
    if (bpp == 16) {
      mr = 31u << 10, mg = 31u << 5, mb = 31u;
    } else if (bpp == 32) {
      mr = 0xffu << 16, mg = 0xffu << 8, mb = 0xffu, ma = 0xffu << 24;
      any_alpha = 0;  // alpha is read, and dropped at the end if it was zero everywhere
    } else {
      mr = mg = mb = ma = 0;
    }
  };
  if (hsz != 12) {
    int compress = (int)c.le32();
    if (compress == 1 || compress == 2) return fail("bmp: RLE is not supported");
    if (compress >= 4) return fail("bmp: unsupported compression");
    if (compress == 3 && bpp != 16 && bpp != 32) return fail("bmp: bit fields need 16 or 32 bits");
    c.skip(20);
    if (hsz == 40 || hsz == 56) {
      if (hsz == 56) c.skip(16);
      if (bpp == 16 || bpp == 32) {
        if (compress == 0) {
          default_masks();
        } else if (compress == 3) {
          mr = c.le32(), mg = c.le32(), mb = c.le32();
          header_read += 12;
          if (mr == mg && mg == mb) return fail("bmp: bad masks");
        } else {
          return fail("bmp: bad compression");
        }
      }
    } else {  // V4 / V5
      mr = c.le32(), mg = c.le32(), mb = c.le32(), ma = c.le32();
      if (compress == 0) default_masks();
      c.skip(hsz == 124 ? 68 : 52);
    }
  }
  bool bottom_up = h > 0;
  if (h < 0) h = h == INT32_MIN ? h : -h;
  if (h < 0 || (uint32_t)h > MAX_DIMENSION || (uint32_t)w > MAX_DIMENSION) return fail("bmp: too large");

  int psize = 0;
  if (hsz == 12) {
    if (bpp < 24) psize = (offset - header_read - 24) / 3;
  } else if (bpp < 16) {
    psize = (offset - header_read - hsz) >> 2;
  }
  if (psize == 0) {
    long long pos = (long long)c.tell();
    if (pos <= 0 || pos > 1024) return fail("bmp: bad header");
    if (offset < pos || offset - pos > 1024) return fail("bmp: bad offset");
    c.skip(offset - pos);
  }
  if (!start(img, w, h, 255)) return fail("bmp: bad extent");
  uint8_t* out = img.rgba.data();
  size_t z = 0;
  if (bpp < 16) {
    if (psize <= 0 || psize > 256) return fail("bmp: bad palette");
    uint8_t pal[256][3] = {};
    for (int i = 0; i < psize; i++) {
      pal[i][2] = (uint8_t)c.u8();
      pal[i][1] = (uint8_t)c.u8();
      pal[i][0] = (uint8_t)c.u8();
      if (hsz != 12) c.u8();
    }
    c.skip(offset - header_read - hsz - psize * (hsz == 12 ? 3 : 4));
    int row_bytes;
    if (bpp == 1) row_bytes = (w + 7) >> 3;
    else if (bpp == 4) row_bytes = (w + 1) >> 1;
    else if (bpp == 8) row_bytes = w;
    else return fail("bmp: bad bits per pixel");
    int pad = (-row_bytes) & 3;
    auto put = [&](uint32_t index) {
      out[z] = pal[index][0], out[z + 1] = pal[index][1], out[z + 2] = pal[index][2], out[z + 3] = 255;
      z += 4;
    };
    for (int y = 0; y < h; y++) {
      if (bpp == 1) {
        uint32_t v = 0;
        for (int x = 0; x < w; x++) {
          if ((x & 7) == 0) v = c.u8();
          put((v >> (7 - (x & 7))) & 1);
        }
      } else if (bpp == 4) {
        uint32_t v = 0;
        for (int x = 0; x < w; x++) {
          if ((x & 1) == 0) v = c.u8();
          put(x & 1 ? v & 15 : v >> 4);
        }
      } else {
        for (int x = 0; x < w; x++) put(c.u8());
      }
      c.skip(pad);
    }
  } else {
    c.skip(offset - header_read - hsz);
    int row_bytes = bpp == 24 ? 3 * w : bpp == 16 ? 2 * w : 0;
    int pad = (-row_bytes) & 3;
    int easy = 0;
    if (bpp == 24) easy = 1;
    else if (bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000) easy = 2;
    int rs = 0, gs = 0, bs = 0, as = 0, rn = 0, gn = 0, bn = 0, an = 0;
    if (!easy) {
      if (!mr || !mg || !mb) return fail("bmp: bad masks");
      rs = top_bit(mr) - 7, rn = __builtin_popcount(mr);
      gs = top_bit(mg) - 7, gn = __builtin_popcount(mg);
      bs = top_bit(mb) - 7, bn = __builtin_popcount(mb);
      as = top_bit(ma) - 7, an = __builtin_popcount(ma);
      if (rn > 8 || gn > 8 || bn > 8 || an > 8) return fail("bmp: bad masks");
    }
    for (int y = 0; y < h; y++) {
      for (int x = 0; x < w; x++, z += 4) {
        uint32_t a;
        if (easy) {
          out[z + 2] = (uint8_t)c.u8();
          out[z + 1] = (uint8_t)c.u8();
          out[z + 0] = (uint8_t)c.u8();
          a = easy == 2 ? c.u8() : 255;
        } else {
          uint32_t v = bpp == 16 ? c.le16() : c.le32();
          out[z + 0] = (uint8_t)bmp_field(v & mr, rs, rn);
          out[z + 1] = (uint8_t)bmp_field(v & mg, gs, gn);
          out[z + 2] = (uint8_t)bmp_field(v & mb, bs, bn);
          a = ma ? bmp_field(v & ma, as, an) : 255;
        }
        any_alpha |= a;
        out[z + 3] = (uint8_t)a;
      }
      c.skip(pad);
    }
  }
  if (any_alpha == 0)
    for (size_t i = 3; i < img.rgba.size(); i += 4) out[i] = 255;
  if (bottom_up) flip_rows(img);
  return true;
}

// ---------------------------------------------------------------------------------------------------
// TGA (stbi__tga_load, stb_image.h:5868).  Types 1, 2, 3 and their run-length forms 9, 10, 11; 8-bit
// grey, 16-bit grey+alpha, 15/16-bit 5-5-5 (the top bit is not alpha), 24, 32 bits; colour maps with 8-
// or 16-bit indices.  Runs continue across rows.  The colour map's "first entry index" is taken as a
// number of BYTES to skip before the map, and an index beyond the map reads entry 0.
inline bool tga_probe(const uint8_t* p, size_t n) {
  Cursor c(p, n);
  c.u8();
  uint32_t mapped = c.u8();
  if (mapped > 1) return false;
  uint32_t type = c.u8();
  if (mapped) {
    if (type != 1 && type != 9) return false;
    c.skip(4);
    uint32_t bits = c.u8();
    if (bits != 8 && bits != 15 && bits != 16 && bits != 24 && bits != 32) return false;
    c.skip(4);
  } else {
    if (type != 2 && type != 3 && type != 10 && type != 11) return false;
    c.skip(9);
  }
  if (c.le16() < 1 || c.le16() < 1) return false;
  uint32_t bpp = c.u8();
  if (mapped && bpp != 8 && bpp != 16) return false;
  return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

inline bool tga_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  uint32_t id_len = c.u8(), mapped = c.u8(), type = c.u8();
  uint32_t map_first = c.le16(), map_len = c.le16(), map_bits = c.u8();
  c.skip(4);
  uint32_t w = c.le16(), h = c.le16(), bpp = c.u8(), descriptor = c.u8();
  bool rle = type >= 8;
  if (rle) type -= 8;
  bool bottom_up = ((descriptor >> 5) & 1) == 0;
  // components per pixel and whether they are packed 5-5-5
  bool packed = false;
  auto components = [&](uint32_t bits, bool grey) -> uint32_t {
    switch (bits) {
      case 8: return 1;
      case 16: if (grey) return 2;  // fall through
      case 15: packed = true; return 3;
      case 24: return 3;
      case 32: return 4;
      default: return 0;
    }
  };
  uint32_t comp = mapped ? components(map_bits, false) : components(bpp, type == 3);
  if (!comp) return fail("tga: unsupported pixel format");
  if (!start(img, w, h, 0)) return fail("tga: bad extent");
  auto unpack555 = [&](uint8_t* dst) {  // stored R, G, B
    uint32_t px = c.le16();
    dst[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31);
    dst[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31);
    dst[2] = (uint8_t)(((px & 31) * 255) / 31);
  };
  size_t count = (size_t)w * h;
  std::vector<uint8_t> raw(count * comp);  // file order of rows, stored component order
  c.skip(id_len);
  if (!mapped && !rle && !packed) {
    c.take(raw.data(), raw.size());
  } else {
    std::vector<uint8_t> map;
    if (mapped) {
      if (map_len == 0) return fail("tga: empty colour map");
      c.skip(map_first);
      map.resize((size_t)map_len * comp);
      if (packed) {
        for (uint32_t i = 0; i < map_len; i++) unpack555(&map[(size_t)i * comp]);
      } else if (!c.take(map.data(), map.size())) {
        return fail("tga: truncated colour map");
      }
    }
    uint8_t px[4] = {0, 0, 0, 0};
    uint32_t run = 0;
    bool repeat = false;
    for (size_t i = 0; i < count; i++) {
      bool fetch = true;
      if (rle) {
        if (run == 0) {
          uint32_t cmd = c.u8();
          run = 1 + (cmd & 127);
          repeat = (cmd >> 7) != 0;
        } else if (repeat) {
          fetch = false;
        }
      }
      if (fetch) {
        if (mapped) {
          uint32_t index = bpp == 8 ? c.u8() : c.le16();
          if (index >= map_len) index = 0;
          for (uint32_t j = 0; j < comp; j++) px[j] = map[(size_t)index * comp + j];
        } else if (packed) {
          unpack555(px);
        } else {
          for (uint32_t j = 0; j < comp; j++) px[j] = (uint8_t)c.u8();
        }
      }
      for (uint32_t j = 0; j < comp; j++) raw[i * comp + j] = px[j];
      run--;
    }
  }
  bool bgr = comp >= 3 && !packed;
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t* src = &raw[(size_t)(bottom_up ? h - 1 - y : y) * w * comp];
    uint8_t* dst = &img.rgba[(size_t)y * w * 4];
    for (uint32_t x = 0; x < w; x++, src += comp, dst += 4) {
      if (comp <= 2) {
        dst[0] = dst[1] = dst[2] = src[0];
        dst[3] = comp == 2 ? src[1] : 255;
      } else {
        dst[0] = src[bgr ? 2 : 0], dst[1] = src[1], dst[2] = src[bgr ? 0 : 2];
        dst[3] = comp == 4 ? src[3] : 255;
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// Binary PGM / PPM (stbi__pnm_load, stb_image.h:7503).  One whitespace byte after maxval, then the
// samples.  16-bit files (maxval > 255) are stored big-endian but narrowed as if little-endian, so what
// comes out is the LOW byte of every sample.
inline bool pnm_probe(const uint8_t* p, size_t n) { return n >= 2 && p[0] == 'P' && (p[1] == '5' || p[1] == '6'); }

inline bool pnm_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  c.u8();
  uint32_t comp = c.u8() == '6' ? 3 : 1;
  char ch = (char)c.u8();
  auto space = [](char k) { return k == ' ' || k == '\t' || k == '\n' || k == '\v' || k == '\f' || k == '\r'; };
  auto skip_blank = [&]() {
    for (;;) {
      while (!c.eof() && space(ch)) ch = (char)c.u8();
      if (c.eof() || ch != '#') break;
      while (!c.eof() && ch != '\n' && ch != '\r') ch = (char)c.u8();
    }
  };
  auto number = [&]() -> int {  // 0 on overflow
    int v = 0;
    while (!c.eof() && ch >= '0' && ch <= '9') {
      v = v * 10 + (ch - '0');
      ch = (char)c.u8();
      if (v > 214748364 || (v == 214748364 && ch > '7')) return 0;
    }
    return v;
  };
  skip_blank();
  int w = number();
  if (w == 0) return fail("pnm: bad width");
  skip_blank();
  int h = number();
  if (h == 0) return fail("pnm: bad height");
  skip_blank();
  int maxv = number();
  if (maxv > 65535) return fail("pnm: maxval above 65535");
  uint32_t bytes = maxv > 255 ? 2 : 1;
  if ((uint32_t)w > MAX_DIMENSION || (uint32_t)h > MAX_DIMENSION) return fail("pnm: too large");
  if ((long long)w * h * comp * bytes > 0x7fffffffLL || !start(img, w, h, 255)) return fail("pnm: too large");
  std::vector<uint8_t> raw((size_t)w * h * comp * bytes);
  if (!c.take(raw.data(), raw.size())) return fail("pnm: truncated");
  const uint8_t* src = raw.data() + (bytes - 1);
  uint8_t* dst = img.rgba.data();
  for (size_t i = 0; i < (size_t)w * h; i++, dst += 4, src += comp * bytes) {
    dst[0] = src[0];
    dst[1] = src[comp == 3 ? bytes : 0];
    dst[2] = src[comp == 3 ? 2 * bytes : 0];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// PSD (stbi__psd_load, stb_image.h:6123).  The composite image of an RGB-mode version-1 file, raw or
// PackBits, 8 bits per channel or the high byte of 16; missing channels are 0 (alpha 255).  With four or
// more channels the colour is un-blended from white in float arithmetic.
inline bool psd_probe(const uint8_t* p, size_t n) { return n >= 4 && std::memcmp(p, "8BPS", 4) == 0; }

inline bool psd_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  c.skip(4);
  if (c.be16() != 1) return fail("psd: unsupported version");
  c.skip(6);
  int channels = (int)c.be16();
  if (channels > 16) return fail("psd: too many channels");
  int32_t h = (int32_t)c.be32(), w = (int32_t)c.be32();
  if (h > (int32_t)MAX_DIMENSION || w > (int32_t)MAX_DIMENSION) return fail("psd: too large");
  uint32_t depth = c.be16();
  if (depth != 8 && depth != 16) return fail("psd: bit depth is not 8 or 16");
  if (c.be16() != 3) return fail("psd: not RGB");
  for (int section = 0; section < 3; section++) c.skip((long long)(int32_t)c.be32());  // mode data, resources, layers
  uint32_t compression = c.be16();
  if (compression > 1) return fail("psd: unknown compression");
  if (!start(img, w, h, 0)) return fail("psd: bad extent");
  size_t count = (size_t)w * h;
  uint8_t* out = img.rgba.data();
  if (compression) c.skip((long long)h * channels * 2);  // the per-row byte counts
  for (int ch = 0; ch < 4; ch++) {
    uint8_t* q = out + ch;
    if (ch >= channels) {
      for (size_t i = 0; i < count; i++) q[i * 4] = ch == 3 ? 255 : 0;
    } else if (compression) {
      size_t done = 0;
      while (done < count) {
        uint32_t len = c.u8();
        if (len == 128) continue;
        if (len < 128) {
          len++;
          if (len > count - done) return fail("psd: bad run-length data");
          for (; len; len--) q[4 * done++] = (uint8_t)c.u8();
        } else {
          len = 257 - len;
          if (len > count - done) return fail("psd: bad run-length data");
          uint8_t v = (uint8_t)c.u8();
          for (; len; len--) q[4 * done++] = v;
        }
      }
    } else if (depth == 16) {
      for (size_t i = 0; i < count; i++) q[i * 4] = (uint8_t)(c.be16() >> 8);
    } else {
      for (size_t i = 0; i < count; i++) q[i * 4] = (uint8_t)c.u8();
    }
  }
  if (channels >= 4) {
    for (size_t i = 0; i < count; i++) {
      uint8_t* px = out + 4 * i;
      if (px[3] != 0 && px[3] != 255) {
        float a = px[3] / 255.0f;
        float ra = 1.0f / a;
        float inv_a = 255.0f * (1 - ra);
        for (int k = 0; k < 3; k++) px[k] = (uint8_t)(int32_t)(px[k] * ra + inv_a);  // wraps below zero, as on x86
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// Softimage PIC (stbi__pic_load, stb_image.h:6497).  8-bit channel packets, uncompressed / pure runs /
// mixed runs; channels a file does not carry stay 255.
inline bool pic_probe(const uint8_t* p, size_t n) {
  return n >= 92 && std::memcmp(p, "\x53\x80\xF6\x34", 4) == 0 && std::memcmp(p + 88, "PICT", 4) == 0;
}

inline bool pic_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  c.skip(92);
  uint32_t w = c.be16(), h = c.be16();
  if (c.eof()) return fail("pic: truncated header");
  c.skip(8);
  if (!start(img, w, h, 255)) return fail("pic: bad extent");
  struct Packet {
    uint32_t size, type, channels;
  } packets[10];
  int n_packets = 0;
  uint32_t chained;
  do {
    if (n_packets == 10) return fail("pic: too many packets");
    Packet& k = packets[n_packets++];
    chained = c.u8();
    k.size = c.u8(), k.type = c.u8(), k.channels = c.u8();
    if (c.eof()) return fail("pic: truncated packets");
    if (k.size != 8) return fail("pic: packet is not 8 bits per channel");
  } while (chained);
  // the channels bit 0x80 >> i says component i is present
  auto read_value = [&](uint32_t channels, uint8_t* dst) {
    for (int i = 0; i < 4; i++)
      if (channels & (0x80u >> i)) {
        if (c.eof()) return false;
        dst[i] = (uint8_t)c.u8();
      }
    return true;
  };
  auto copy_value = [](uint32_t channels, uint8_t* dst, const uint8_t* src) {
    for (int i = 0; i < 4; i++)
      if (channels & (0x80u >> i)) dst[i] = src[i];
  };
  for (uint32_t y = 0; y < h; y++) {
    for (int k = 0; k < n_packets; k++) {
      const Packet& pk = packets[k];
      uint8_t* dst = &img.rgba[(size_t)y * w * 4];
      if (pk.type == 0) {
        for (uint32_t x = 0; x < w; x++, dst += 4)
          if (!read_value(pk.channels, dst)) return fail("pic: truncated");
      } else if (pk.type == 1) {
        int left = (int)w;
        while (left > 0) {
          uint32_t count = c.u8();
          if (c.eof()) return fail("pic: truncated");
          if ((int)count > left) count = (uint32_t)left & 0xff;
          uint8_t v[4];
          if (!read_value(pk.channels, v)) return fail("pic: truncated");
          for (uint32_t i = 0; i < count; i++, dst += 4) copy_value(pk.channels, dst, v);
          left -= (int)count;
        }
      } else if (pk.type == 2) {
        int left = (int)w;
        while (left > 0) {
          int count = (int)c.u8();
          if (c.eof()) return fail("pic: truncated");
          if (count >= 128) {
            count = count == 128 ? (int)c.be16() : count - 127;
            if (count > left) return fail("pic: scanline overrun");
            uint8_t v[4];
            if (!read_value(pk.channels, v)) return fail("pic: truncated");
            for (int i = 0; i < count; i++, dst += 4) copy_value(pk.channels, dst, v);
          } else {
            count++;
            if (count > left) return fail("pic: scanline overrun");
            for (int i = 0; i < count; i++, dst += 4)
              if (!read_value(pk.channels, dst)) return fail("pic: truncated");
          }
          left -= count;
        }
      } else {
        return fail("pic: bad packet type");
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// GIF (stbi__gif_load, stb_image.h:7045): the first image of the file on a transparent-black canvas.
// Pixels of the transparent index are left untouched; when the background index is not 0, pixels the
// image did not cover take the background entry with its red and blue bytes exchanged (the colour
// tables are kept B, G, R and that entry is copied without swapping back).
inline bool gif_probe(const uint8_t* p, size_t n) {
  return n >= 6 && std::memcmp(p, "GIF8", 4) == 0 && (p[4] == '7' || p[4] == '9') && p[5] == 'a';
}

inline bool gif_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  c.skip(6);
  uint32_t W = c.le16(), H = c.le16(), flags = c.u8(), bg_index = c.u8();
  c.u8();  // aspect ratio
  struct Entry {
    uint8_t b, g, r, a;
  };
  std::vector<Entry> global(256, Entry{0, 0, 0, 0}), local(256, Entry{0, 0, 0, 0});
  auto read_table = [&](std::vector<Entry>& t, uint32_t entries, int transparent) {
    for (uint32_t i = 0; i < entries; i++) {
      t[i].r = (uint8_t)c.u8(), t[i].g = (uint8_t)c.u8(), t[i].b = (uint8_t)c.u8();
      t[i].a = (int)i == transparent ? 0 : 255;
    }
  };
  if (flags & 0x80) read_table(global, 2u << (flags & 7), -1);
  if (!start(img, W, H, 0)) return fail("gif: bad extent");
  std::vector<uint8_t> touched((size_t)W * H, 0);
  uint32_t gce_flags = 0;
  int transparent = -1;
  for (;;) {
    uint32_t tag = c.u8();
    if (tag == 0x21) {
      uint32_t label = c.u8(), len;
      if (label == 0xF9) {
        len = c.u8();
        if (len != 4) {
          c.skip(len);
          continue;
        }
        gce_flags = c.u8();
        c.le16();  // delay
        if (transparent >= 0) global[transparent].a = 255;
        if (gce_flags & 1) {
          transparent = (int)c.u8();
          global[transparent].a = 0;
        } else {
          c.skip(1);
          transparent = -1;
        }
      }
      while ((len = c.u8()) != 0) c.skip(len);
      continue;
    }
    if (tag == 0x3B) return fail("gif: no image");
    if (tag != 0x2C) return fail("gif: unknown block");
    break;
  }
  uint32_t x0 = c.le16(), y0 = c.le16(), w = c.le16(), h = c.le16();
  if (x0 + w > W || y0 + h > H) return fail("gif: image outside the screen");
  uint32_t lflags = c.u8();
  const std::vector<Entry>* table;
  if (lflags & 0x80) {
    read_table(local, 2u << (lflags & 7), gce_flags & 1 ? transparent : -1);
    table = &local;
  } else if (flags & 0x80) {
    table = &global;
  } else {
    return fail("gif: no colour table");
  }
  // where the next pixel goes: row order 0, 8, 16.. / 4, 12.. / 2, 6.. / 1, 3.. when interlaced
  uint32_t col = 0, row = 0, step = lflags & 0x40 ? 8 : 1;
  int passes_left = lflags & 0x40 ? 3 : 0;
  bool full = w == 0 || h == 0;
  auto emit = [&](uint8_t index) {
    if (full) return;
    size_t at = (size_t)(y0 + row) * W + x0 + col;
    touched[at] = 1;
    const Entry& e = (*table)[index];
    if (e.a > 128) {
      uint8_t* px = &img.rgba[at * 4];
      px[0] = e.r, px[1] = e.g, px[2] = e.b, px[3] = e.a;
    }
    if (++col >= w) {
      col = 0;
      row += step;
      while (row >= h && passes_left > 0) {
        step = 1u << passes_left;
        row = step >> 1;
        passes_left--;
      }
      if (row >= h) full = true;
    }
  };

  // LZW, at most 12-bit codes; a stream must begin with a clear code
  uint32_t min_size = c.u8();
  if (min_size > 12) return fail("gif: bad code size");
  const int clear = 1 << min_size;
  std::vector<int16_t> prefix(8192, -1);
  std::vector<uint8_t> head(8192, 0), tail(8192, 0), chain(8192);
  for (int i = 0; i < clear; i++) head[i] = tail[i] = (uint8_t)i;
  int code_size = (int)min_size + 1, code_mask = (1 << code_size) - 1, next = clear + 2, previous = -1;
  bool seen_clear = false;
  int32_t bits = 0;
  int have = 0;
  uint32_t block_left = 0;
  for (;;) {
    if (have < code_size) {
      if (block_left == 0) {
        block_left = c.u8();
        if (block_left == 0) break;  // also the end of a truncated file
      }
      block_left--;
      bits |= (int32_t)c.u8() << have;
      have += 8;
      continue;
    }
    int code = bits & code_mask;
    bits >>= code_size;
    have -= code_size;
    if (code == clear) {
      code_size = (int)min_size + 1;
      code_mask = (1 << code_size) - 1;
      next = clear + 2;
      previous = -1;
      seen_clear = true;
    } else if (code == clear + 1) {
      break;  // end of information; what follows in the file is not looked at
    } else if (code <= next) {
      if (!seen_clear) return fail("gif: no clear code");
      if (previous >= 0) {
        int slot = next++;
        if (next > 8192) return fail("gif: too many codes");
        prefix[slot] = (int16_t)previous;
        head[slot] = head[previous];
        tail[slot] = head[code];  // code == slot reads the head just written
      } else if (code == next) {
        return fail("gif: illegal code");
      }
      int depth = 0;
      for (int k = code; k >= 0 && depth < 8192; k = prefix[k]) chain[depth++] = tail[k];
      while (depth) emit(chain[--depth]);
      if ((next & code_mask) == 0 && next <= 0x0fff) {
        code_size++;
        code_mask = (1 << code_size) - 1;
      }
      previous = code;
    } else {
      return fail("gif: illegal code");
    }
  }
  if (bg_index > 0) {
    const Entry& e = global[bg_index];
    for (size_t i = 0; i < touched.size(); i++)
      if (!touched[i]) {
        uint8_t* px = &img.rgba[i * 4];
        px[0] = e.b, px[1] = e.g, px[2] = e.r, px[3] = 255;
      }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// Radiance HDR (stbi__hdr_load + stbi__hdr_to_ldr, stb_image.h:7155, :1883): RGBE, flat or the new
// run-length rows, "-Y h +X w" orientation only, tone-mapped to 8 bits as pow(v, 1/2.2) * 255 + 0.5
// in float arithmetic.
inline bool hdr_probe(const uint8_t* p, size_t n) {
  return (n >= 11 && std::memcmp(p, "#?RADIANCE\n", 11) == 0) || (n >= 7 && std::memcmp(p, "#?RGBE\n", 7) == 0);
}

inline bool hdr_decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  Cursor c(p, n);
  auto line = [&]() {  // up to 1022 characters; a final character at the very end of the file is dropped
    std::string s;
    char ch = (char)c.u8();
    while (!c.eof() && ch != '\n') {
      s.push_back(ch);
      if (s.size() == 1023) {
        while (!c.eof() && c.u8() != '\n') {
        }
        break;
      }
      ch = (char)c.u8();
    }
    return s;
  };
  std::string first = line();
  if (std::strcmp(first.c_str(), "#?RADIANCE") != 0 && std::strcmp(first.c_str(), "#?RGBE") != 0) return fail("hdr: bad signature");
  bool rgbe = false;
  for (;;) {
    std::string s = line();
    if (s.empty() || s[0] == 0) break;
    if (std::strcmp(s.c_str(), "FORMAT=32-bit_rle_rgbe") == 0) rgbe = true;
  }
  if (!rgbe) return fail("hdr: unsupported format");
  std::string res = line();
  const char* t = res.c_str();
  if (std::strncmp(t, "-Y ", 3)) return fail("hdr: unsupported orientation");
  char* end = nullptr;
  long hl = std::strtol(t + 3, &end, 10);
  while (*end == ' ') end++;
  if (std::strncmp(end, "+X ", 3)) return fail("hdr: unsupported orientation");
  long wl = std::strtol(end + 3, nullptr, 10);
  int width = (int)wl, height = (int)hl;
  if (height > (int)MAX_DIMENSION || width > (int)MAX_DIMENSION) return fail("hdr: too large");
  if (width <= 0 || height <= 0 || (long long)width * height * 16 > 0x7fffffffLL || !start(img, width, height, 255))
    return fail("hdr: bad extent");
  const float gamma = 1.0f / 2.2f;
  auto store = [&](size_t pixel, const uint8_t* q) {
    uint8_t* dst = &img.rgba[pixel * 4];
    float scale = q[3] ? std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
    for (int k = 0; k < 3; k++) {
      float v = q[3] ? q[k] * scale : 0.0f;
      float z = std::pow(v * 1.0f, gamma) * 255 + 0.5f;
      if (z < 0) z = 0;
      if (z > 255) z = 255;
      dst[k] = (uint8_t)(int)z;
    }
    dst[3] = 255;
  };
  size_t count = (size_t)width * height;
  auto flat_from = [&](size_t firstpixel) {
    for (size_t i = firstpixel; i < count; i++) {
      uint8_t q[4] = {0, 0, 0, 0};
      c.take(q, 4);
      store(i, q);
    }
  };
  if (width < 8 || width >= 32768) {
    flat_from(0);
    return true;
  }
  std::vector<uint8_t> scan((size_t)width * 4);
  for (int y = 0; y < height; y++) {
    uint32_t c1 = c.u8(), c2 = c.u8(), len = c.u8();
    if (c1 != 2 || c2 != 2 || (len & 0x80)) {
      // not a run-length row: these bytes are pixel 0 of a flat file, whichever row this is
      uint8_t q[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)c.u8()};
      store(0, q);
      flat_from(1);
      return true;
    }
    len = (len << 8) | c.u8();
    if ((int)len != width) return fail("hdr: bad scanline length");
    for (int k = 0; k < 4; k++) {
      int x = 0;
      while (x < width) {
        uint32_t run = c.u8();
        if (run > 128) {
          uint8_t v = (uint8_t)c.u8();
          run -= 128;
          if (run == 0 || (int)run > width - x) return fail("hdr: bad run-length data");
          for (; run; run--) scan[(size_t)x++ * 4 + k] = v;
        } else {
          if (run == 0 || (int)run > width - x) return fail("hdr: bad run-length data");
          for (; run; run--) scan[(size_t)x++ * 4 + k] = (uint8_t)c.u8();
        }
      }
    }
    for (int x = 0; x < width; x++) store((size_t)y * width + x, &scan[(size_t)x * 4]);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// The probe order of stbi__load_main (stb_image.h:1136): formats with a real signature first, TGA last.
// `format` (may be null) names the decoder that took the file.
inline bool decode(const uint8_t* p, size_t n, Image& img, std::string* err, const char** format = nullptr) {
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  const char* dummy;
  if (!format) format = &dummy;
  if (n >= 8 && std::memcmp(p, png_sig, 8) == 0) {
    *format = "png";
    svrpng::Image t;
    if (!svrpng::decode(p, n, t, err)) return false;
    img.w = t.w, img.h = t.h, img.rgba.swap(t.rgba);
    return true;
  }
  if (bmp_probe(p, n)) return *format = "bmp", bmp_decode(p, n, img, err);
  if (gif_probe(p, n)) return *format = "gif", gif_decode(p, n, img, err);
  if (psd_probe(p, n)) return *format = "psd", psd_decode(p, n, img, err);
  if (pic_probe(p, n)) return *format = "pic", pic_decode(p, n, img, err);
  if (n >= 2 && p[0] == 0xff && p[1] == 0xd8) {
    *format = "jpeg";
    svrjpeg::Image t;
    if (!svrjpeg::decode(p, n, t, err)) return false;
    img.w = t.w, img.h = t.h, img.rgba.swap(t.rgba);
    return true;
  }
  if (pnm_probe(p, n)) return *format = "pnm", pnm_decode(p, n, img, err);
  if (hdr_probe(p, n)) return *format = "hdr", hdr_decode(p, n, img, err);
  if (tga_probe(p, n)) return *format = "tga", tga_decode(p, n, img, err);
  *format = "unknown";
  if (err) *err = "image: not of any known type";
  return false;
}

}  // namespace svrimg
