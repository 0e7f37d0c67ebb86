// svr_png.h — PNG -> tightly packed RGBA8, the conversion stbi_load(..., 4) performs for the reference's
// load_image (src/vk_loader.cpp:81-160; stb_image v2.29 is vendored there, not here: this is an
// independent decoder of the PNG specification, inflate included).
//
// Covered: colour types 0/2/3/4/6, bit depths 1-16 (16-bit samples keep their high byte, as stb does),
// PLTE + tRNS (palette alpha, and the grey / RGB colour key), Adam7 interlace.  Chunk CRCs and the
// zlib Adler-32 are not verified (stb_image does not verify them either).  Pinned byte for byte against
// the reference's decoder by tests/golden/images.npz (tests/test_image_decoders.py).  JPEG: svr_jpeg.h;
// the other formats stb_image reads are not covered: load fails and the loader falls back to the error
// checkerboard, which is what the reference does for an image it cannot decode (src/vk_loader.cpp:226-231).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace svrpng {

// ---------------------------------------------------------------- inflate (RFC 1951) inside zlib (RFC 1950)
class Inflater {
 public:
  Inflater(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  bool run(std::vector<uint8_t>& out, std::string* err) {
    if (n_ < 2 || (p_[0] & 0x0f) != 8 || ((p_[0] << 8) | p_[1]) % 31 != 0 || (p_[1] & 0x20)) return fail(err, "bad zlib header");
    pos_ = 2;
    for (;;) {
      uint32_t final = bits(1), type = bits(2);
      if (bad_) return fail(err, "truncated deflate stream");
      if (type == 0) {
        nbits_ = 0;  // to the byte boundary
        acc_ = 0;
        if (pos_ + 4 > n_) return fail(err, "truncated stored block");
        uint32_t len = p_[pos_] | (p_[pos_ + 1] << 8), nlen = p_[pos_ + 2] | (p_[pos_ + 3] << 8);
        pos_ += 4;
        if ((len ^ 0xffffu) != nlen || pos_ + len > n_) return fail(err, "bad stored block");
        out.insert(out.end(), p_ + pos_, p_ + pos_ + len);
        pos_ += len;
      } else if (type == 1 || type == 2) {
        Table lit, dist;
        if (type == 1) {
          uint8_t l[288];
          for (int i = 0; i < 144; i++) l[i] = 8;
          for (int i = 144; i < 256; i++) l[i] = 9;
          for (int i = 256; i < 280; i++) l[i] = 7;
          for (int i = 280; i < 288; i++) l[i] = 8;
          uint8_t d[30];
          for (int i = 0; i < 30; i++) d[i] = 5;
          build(lit, l, 288);
          build(dist, d, 30);
        } else if (!dynamic_tables(lit, dist)) {
          return fail(err, "bad dynamic Huffman tables");
        }
        if (!block(lit, dist, out)) return fail(err, "bad compressed block");
      } else {
        return fail(err, "reserved block type");
      }
      if (final) return true;
    }
  }

 private:
  struct Table {
    uint16_t count[16];
    uint16_t symbol[288];
  };
  const uint8_t* p_;
  size_t n_, pos_ = 0;
  uint32_t acc_ = 0;
  int nbits_ = 0;
  bool bad_ = false;

  static bool fail(std::string* err, const char* m) {
    if (err) *err = m;
    return false;
  }
  uint32_t bits(int need) {
    while (nbits_ < need) {
      if (pos_ >= n_) {
        bad_ = true;
        return 0;
      }
      acc_ |= (uint32_t)p_[pos_++] << nbits_;
      nbits_ += 8;
    }
    uint32_t v = acc_ & ((1u << need) - 1u);
    acc_ >>= need;
    nbits_ -= need;
    return v;
  }
  static void build(Table& t, const uint8_t* len, int n) {
    std::memset(t.count, 0, sizeof(t.count));
    for (int i = 0; i < n; i++) t.count[len[i]]++;
    t.count[0] = 0;
    uint16_t offs[16];
    offs[1] = 0;
    for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + t.count[i]);
    for (int i = 0; i < n; i++)
      if (len[i]) t.symbol[offs[len[i]]++] = (uint16_t)i;
  }
  int decode(const Table& t) {  // canonical Huffman, bit by bit
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
      code |= (int)bits(1);
      if (bad_) return -1;
      int count = t.count[len];
      if (code - count < first) return t.symbol[index + (code - first)];
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
  bool dynamic_tables(Table& lit, Table& dist) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
    if (bad_ || nlen > 286 || ndist > 30) return false;
    uint8_t lengths[320] = {0};
    for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)bits(3);
    Table cl;
    build(cl, lengths, 19);
    std::memset(lengths, 0, sizeof(lengths));
    int i = 0;
    while (i < nlen + ndist) {
      int sym = decode(cl);
      if (sym < 0) return false;
      if (sym < 16) {
        lengths[i++] = (uint8_t)sym;
      } else {
        int prev = 0, rep;
        if (sym == 16) {
          if (i == 0) return false;
          prev = lengths[i - 1];
          rep = 3 + (int)bits(2);
        } else if (sym == 17) {
          rep = 3 + (int)bits(3);
        } else {
          rep = 11 + (int)bits(7);
        }
        if (bad_ || i + rep > nlen + ndist) return false;
        while (rep--) lengths[i++] = (uint8_t)prev;
      }
    }
    if (lengths[256] == 0) return false;
    build(lit, lengths, nlen);
    build(dist, lengths + nlen, ndist);
    return true;
  }
  bool block(const Table& lit, const Table& dist, std::vector<uint8_t>& out) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(lit);
      if (sym < 0) return false;
      if (sym < 256) {
        out.push_back((uint8_t)sym);
      } else if (sym == 256) {
        return true;
      } else {
        sym -= 257;
        if (sym >= 29) return false;
        size_t len = lbase[sym] + bits(lext[sym]);
        int ds = decode(dist);
        if (ds < 0 || ds >= 30) return false;
        size_t d = dbase[ds] + bits(dext[ds]);
        if (bad_ || d > out.size()) return false;
        size_t from = out.size() - d;
        for (size_t k = 0; k < len; k++) out.push_back(out[from + k]);
      }
    }
  }
};

// ---------------------------------------------------------------- PNG
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline uint8_t paeth(int a, int b, int c) {
  int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// undo the per-scanline filters of one (sub)image in place; rows are `stride` bytes + 1 filter byte
inline bool unfilter(uint8_t* data, size_t avail, uint32_t rows, size_t stride, uint32_t bpp) {
  if ((size_t)rows * (stride + 1) > avail) return false;
  for (uint32_t y = 0; y < rows; y++) {
    uint8_t* row = data + (size_t)y * (stride + 1);
    uint8_t ft = row[0];
    uint8_t* cur = row + 1;
    const uint8_t* up = y ? row - stride : nullptr;  // previous row's pixel bytes (starts at row - (stride+1) + 1)
    for (size_t i = 0; i < stride; i++) {
      int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      switch (ft) {
        case 0: break;
        case 1: cur[i] = (uint8_t)(cur[i] + a); break;
        case 2: cur[i] = (uint8_t)(cur[i] + b); break;
        case 3: cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1)); break;
        case 4: cur[i] = (uint8_t)(cur[i] + paeth(a, b, c)); break;
        default: return false;
      }
    }
  }
  return true;
}

struct Image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;  // w*h*4
};

inline bool decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return false;
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (n < 8 || std::memcmp(p, sig, 8) != 0) return fail("not a PNG");
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat;
  uint8_t pal[256][4];
  int pal_n = 0;
  bool has_key = false;
  uint16_t key[3] = {0, 0, 0};
  for (int i = 0; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0, pal[i][3] = 255;
  bool end = false;
  while (!end) {
    if (pos + 8 > n) return fail("truncated PNG");
    uint32_t len = be32(p + pos);
    const uint8_t* type = p + pos + 4;
    const uint8_t* body = p + pos + 8;
    if (pos + 12 + (size_t)len > n) return fail("truncated PNG chunk");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail("bad IHDR");
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
      if (w == 0 || h == 0 || w > 32768 || h > 32768) return fail("unsupported PNG extent");
      if (body[10] != 0 || body[11] != 0 || interlace > 1) return fail("unsupported PNG method");
      bool ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
      if (!ok) return fail("unsupported PNG colour type / bit depth");
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len % 3 || len > 768) return fail("bad PLTE");
      pal_n = (int)(len / 3);
      for (int i = 0; i < pal_n; i++) pal[i][0] = body[3 * i], pal[i][1] = body[3 * i + 1], pal[i][2] = body[3 * i + 2];
    } else if (!std::memcmp(type, "tRNS", 4)) {
      if (ctype == 3) {
        for (uint32_t i = 0; i < len && i < 256; i++) pal[i][3] = body[i];
      } else if (ctype == 0 && len >= 2) {
        has_key = true;
        key[0] = (uint16_t)((body[0] << 8) | body[1]);
      } else if (ctype == 2 && len >= 6) {
        has_key = true;
        for (int k = 0; k < 3; k++) key[k] = (uint16_t)((body[2 * k] << 8) | body[2 * k + 1]);
      }
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || idat.empty()) return fail("PNG without IHDR or IDAT");
  std::vector<uint8_t> raw;
  {
    Inflater inf(idat.data(), idat.size());
    if (!inf.run(raw, err)) return false;
  }
  const int channels = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : 4)));
  const uint32_t bits_pp = (uint32_t)(channels * depth);
  const uint32_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter unit
  img.w = w;
  img.h = h;
  img.rgba.assign((size_t)w * h * 4, 0);
  // one pixel of a defiltered row -> RGBA8 at (x, y)
  auto emit = [&](const uint8_t* row, uint32_t xi, uint32_t x, uint32_t y) {
    uint16_t s[4] = {0, 0, 0, 0};
    if (depth == 16) {
      for (int c = 0; c < channels; c++) s[c] = (uint16_t)((row[(size_t)(xi * channels + c) * 2] << 8) | row[(size_t)(xi * channels + c) * 2 + 1]);
    } else if (depth == 8) {
      for (int c = 0; c < channels; c++) s[c] = row[(size_t)xi * channels + c];
    } else {
      uint32_t bit = xi * (uint32_t)depth;
      s[0] = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u));
    }
    uint8_t* o = &img.rgba[((size_t)y * w + x) * 4];
    auto to8 = [&](uint16_t v) -> uint8_t {
      if (depth == 16) return (uint8_t)(v >> 8);
      if (depth == 8) return (uint8_t)v;
      return (uint8_t)(v * (255u / ((1u << depth) - 1u)));
    };
    switch (ctype) {
      case 0:
        o[0] = o[1] = o[2] = to8(s[0]);
        o[3] = (has_key && s[0] == key[0]) ? 0 : 255;
        break;
      case 2:
        o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
        o[3] = (has_key && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : 255;
        break;
      case 3: {
        const uint8_t* c = pal[s[0] & 255];
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
        break;
      }
      case 4:
        o[0] = o[1] = o[2] = to8(s[0]);
        o[3] = to8(s[1]);
        break;
      default:
        o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]); o[3] = to8(s[3]);
    }
  };
  if (!interlace) {
    size_t stride = ((size_t)w * bits_pp + 7) / 8;
    if (!unfilter(raw.data(), raw.size(), h, stride, bpp)) return fail("PNG data too short or bad filter");
    for (uint32_t y = 0; y < h; y++) {
      const uint8_t* row = raw.data() + (size_t)y * (stride + 1) + 1;
      for (uint32_t x = 0; x < w; x++) emit(row, x, x, y);
    }
  } else {
    static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t off = 0;
    for (int pass = 0; pass < 7; pass++) {
      uint32_t pw = (w - (uint32_t)xo[pass] + (uint32_t)xs[pass] - 1) / (uint32_t)xs[pass];
      uint32_t ph = (h - (uint32_t)yo[pass] + (uint32_t)ys[pass] - 1) / (uint32_t)ys[pass];
      if ((uint32_t)xo[pass] >= w || (uint32_t)yo[pass] >= h) pw = ph = 0;
      if (pw == 0 || ph == 0) continue;
      size_t stride = ((size_t)pw * bits_pp + 7) / 8;
      if (off > raw.size() || !unfilter(raw.data() + off, raw.size() - off, ph, stride, bpp)) return fail("PNG data too short or bad filter");
      for (uint32_t y = 0; y < ph; y++) {
        const uint8_t* row = raw.data() + off + (size_t)y * (stride + 1) + 1;
        for (uint32_t x = 0; x < pw; x++) emit(row, x, (uint32_t)xo[pass] + x * (uint32_t)xs[pass], (uint32_t)yo[pass] + y * (uint32_t)ys[pass]);
      }
      off += (size_t)ph * (stride + 1);
    }
  }
  return true;
}

}  // namespace svrpng
