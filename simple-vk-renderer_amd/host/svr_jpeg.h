// svr_jpeg.h — JPEG (ITU T.81 baseline and progressive Huffman, 8-bit) -> tightly packed RGBA8, the
// conversion stbi_load(..., 4) performs for the reference's load_image (src/vk_loader.cpp:81-160).
//
// An independent decoder of the standard whose reconstruction arithmetic is the one the reference's
// stb_image v2.29 documents for itself, because bytes must match:
//   IDCT            the integer "islow" transform (Loeffler-Ligtenberg-Moschytz as in jidctint) with 12-bit
//                   constants; column pass keeps 2 extra bits (+512 >> 10), row pass rounds, re-centres
//                   (+65536 + (128 << 17) >> 17) and clamps
//   chroma          "h2v1": 3:1 horizontal taps; "h1v2": 3:1 vertical; "h2v2": 3:1 vertical then 3:1
//                   horizontal in one 16th-precision step; other factors: nearest
//   YCbCr -> RGB    20-bit fixed point with the 4096-scaled BT.601 constants, the Cb term of green masked
//                   to its upper 16 bits, +0.5 rounding folded into Y
// Entropy decoding, progressive refinement, restart intervals and multi-scan files follow the standard
// (any conforming decoder yields the same coefficients).  Not covered, as rarely met in glTF assets:
// arithmetic coding, 12-bit precision, CMYK/YCCK (four components).  tests/golden/images.npz pins every
// path here against the reference's decoder run in place (tests/test_image_decoders.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace svrjpeg {

struct Image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;
};

namespace detail {

static const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {  // canonical code: per length the first code, the last code + 1 and the first symbol index
  bool present = false;
  int first_code[17], limit[17], first_index[17];
  uint8_t symbols[256];
  bool build(const uint8_t counts[16], const uint8_t* syms, int n) {
    int code = 0, index = 0;
    for (int len = 1; len <= 16; len++) {
      first_code[len] = code;
      first_index[len] = index;
      code += counts[len - 1];
      index += counts[len - 1];
      limit[len] = code;
      if (code > (1 << len)) return false;
      code <<= 1;
    }
    if (index != n || n > 256) return false;
    std::memcpy(symbols, syms, (size_t)n);
    present = true;
    return true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int td = 0, ta = 0;       // Huffman table selectors of the current scan
  int bw = 0, bh = 0;       // blocks per row / column, padded to whole MCUs
  int x = 0, y = 0;         // samples of the unpadded component
  int dc_pred = 0;
  std::vector<int16_t> coef;  // bw*bh*64, natural (de-zigzagged) order
  std::vector<uint8_t> plane; // (bw*8) * (bh*8)
};

inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one 8-point pass of the transform; s[] in, the four even-part sums x[] and odd-part terms t[] out
inline void idct_1d(const int s[8], int x[4], int t[4]) {
  auto f = [](double c) { return (int)(c * 4096 + 0.5); };
  static const int c0541 = f(0.5411961), cn1847 = f(-1.847759065), c0765 = f(0.765366865), c1175 = f(1.175875602),
                   c0298 = f(0.298631336), c2053 = f(2.053119869), c3072 = f(3.072711026), c1501 = f(1.501321110),
                   cn0899 = f(-0.899976223), cn2562 = f(-2.562915447), cn1961 = f(-1.961570560), cn0390 = f(-0.390180644);
  int p1 = (s[2] + s[6]) * c0541;
  int e2 = p1 + s[6] * cn1847, e3 = p1 + s[2] * c0765;
  int e0 = (s[0] + s[4]) * 4096, e1 = (s[0] - s[4]) * 4096;
  x[0] = e0 + e3;
  x[3] = e0 - e3;
  x[1] = e1 + e2;
  x[2] = e1 - e2;
  int o0 = s[7], o1 = s[5], o2 = s[3], o3 = s[1];
  int q3 = o0 + o2, q4 = o1 + o3, q1 = o0 + o3, q2 = o1 + o2;
  int q5 = (q3 + q4) * c1175;
  o0 *= c0298;
  o1 *= c2053;
  o2 *= c3072;
  o3 *= c1501;
  q1 = q5 + q1 * cn0899;
  q2 = q5 + q2 * cn2562;
  q3 *= cn1961;
  q4 *= cn0390;
  t[3] = o3 + q1 + q4;
  t[2] = o2 + q2 + q3;
  t[1] = o1 + q2 + q4;
  t[0] = o0 + q1 + q3;
}

inline void idct_block(uint8_t* out, int stride, const int16_t d[64]) {
  int v[64];
  for (int c = 0; c < 8; c++) {
    int s[8], x[4], t[4];
    for (int k = 0; k < 8; k++) s[k] = d[c + 8 * k];
    idct_1d(s, x, t);
    for (int k = 0; k < 4; k++) x[k] += 512;
    v[c + 0] = (x[0] + t[3]) >> 10;
    v[c + 56] = (x[0] - t[3]) >> 10;
    v[c + 8] = (x[1] + t[2]) >> 10;
    v[c + 48] = (x[1] - t[2]) >> 10;
    v[c + 16] = (x[2] + t[1]) >> 10;
    v[c + 40] = (x[2] - t[1]) >> 10;
    v[c + 24] = (x[3] + t[0]) >> 10;
    v[c + 32] = (x[3] - t[0]) >> 10;
  }
  for (int r = 0; r < 8; r++) {
    int x[4], t[4];
    idct_1d(v + 8 * r, x, t);
    for (int k = 0; k < 4; k++) x[k] += 65536 + (128 << 17);
    uint8_t* o = out + (size_t)r * stride;
    o[0] = (uint8_t)clamp255((x[0] + t[3]) >> 17);
    o[7] = (uint8_t)clamp255((x[0] - t[3]) >> 17);
    o[1] = (uint8_t)clamp255((x[1] + t[2]) >> 17);
    o[6] = (uint8_t)clamp255((x[1] - t[2]) >> 17);
    o[2] = (uint8_t)clamp255((x[2] + t[1]) >> 17);
    o[5] = (uint8_t)clamp255((x[2] - t[1]) >> 17);
    o[3] = (uint8_t)clamp255((x[3] + t[0]) >> 17);
    o[4] = (uint8_t)clamp255((x[3] - t[0]) >> 17);
  }
}

class Decoder {
 public:
  Decoder(const uint8_t* p, size_t n) : p_(p), n_(n) {}

  bool run(Image& img, std::string* err) {
    if (!parse()) {
      if (err) *err = err_.empty() ? "corrupt JPEG" : err_;
      return false;
    }
    reconstruct(img);
    return true;
  }

 private:
  const uint8_t* p_;
  size_t n_, pos_ = 0;
  std::string err_;
  // frame
  int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0;
  bool progressive_ = false, have_frame_ = false, jfif_ = false;
  int adobe_transform_ = -1, rgb_ids_ = 0;
  Component comp_[4];
  uint16_t quant_[4][64];
  Huffman dc_[4], ac_[4];
  int restart_interval_ = 0;
  // scan
  int scan_n_ = 0, order_[4] = {0, 0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eob_run_ = 0;
  // entropy-coded segment bit reader
  uint32_t acc_ = 0;
  int nbits_ = 0;
  bool hit_marker_ = false;

  bool fail(const char* m) {
    if (err_.empty()) err_ = m;
    return false;
  }
  int byte() { return pos_ < n_ ? p_[pos_++] : 0; }
  int be16() {
    int a = byte();
    return (a << 8) | byte();
  }

  // ---- bit reader: bytes after FF 00 stuffing; a marker ends the data (further bits read as 0)
  void fill() {
    while (nbits_ <= 24) {
      int b = 0;
      if (!hit_marker_ && pos_ < n_) {
        b = p_[pos_];
        if (b == 0xff) {
          int nx = pos_ + 1 < n_ ? p_[pos_ + 1] : 0xd9;
          if (nx == 0) {
            pos_ += 2;
          } else {
            hit_marker_ = true;  // leave the marker in place
            b = 0;
          }
        } else {
          pos_++;
        }
      }
      acc_ |= (uint32_t)b << (24 - nbits_);
      nbits_ += 8;
    }
  }
  int get_bits(int n) {
    if (n == 0) return 0;
    if (nbits_ < n) fill();
    int v = (int)(acc_ >> (32 - n));
    acc_ <<= n;
    nbits_ -= n;
    return v;
  }
  int get_bit() { return get_bits(1); }
  int receive_extend(int n) {  // T.81 F.2.2.1: n magnitude bits, sign by the leading bit
    if (n == 0) return 0;
    int v = get_bits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  int decode_symbol(const Huffman& h) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
      code = (code << 1) | get_bit();
      if (code < h.limit[len] && code >= h.first_code[len]) return h.symbols[h.first_index[len] + code - h.first_code[len]];
    }
    fail("bad Huffman code");
    return -1;
  }
  void reset_entropy() {
    acc_ = 0;
    nbits_ = 0;
    hit_marker_ = false;
    eob_run_ = 0;
    for (int i = 0; i < 4; i++) comp_[i].dc_pred = 0;
  }

  // ---- marker segments
  bool parse() {
    if (n_ < 4 || p_[0] != 0xff || p_[1] != 0xd8) return fail("not a JPEG");
    pos_ = 2;
    for (;;) {
      int m = next_marker();
      if (m < 0) return fail("JPEG ends before EOI");
      if (m == 0xd9) break;
      if (m == 0xda) {
        if (!scan_header() || !entropy_data()) return false;
        continue;
      }
      if (!segment(m)) return false;
    }
    return have_frame_ ? true : fail("JPEG without a frame");
  }
  int next_marker() {
    while (pos_ < n_) {
      if (p_[pos_++] != 0xff) continue;
      while (pos_ < n_ && p_[pos_] == 0xff) pos_++;
      if (pos_ >= n_) return -1;
      int m = p_[pos_++];
      if (m != 0) return m;
    }
    return -1;
  }
  bool segment(int m) {
    if (m >= 0xd0 && m <= 0xd7) return true;  // stray restart marker
    if (m == 0x01) return true;
    int len = be16();
    if (len < 2 || pos_ + (size_t)(len - 2) > n_) return fail("bad segment length");
    size_t end = pos_ + (size_t)(len - 2);
    switch (m) {
      case 0xc0: case 0xc1: case 0xc2:
        if (!frame_header(m == 0xc2)) return false;
        break;
      case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
        return fail("unsupported JPEG process (lossless / hierarchical / arithmetic)");
      case 0xc4:  // DHT
        while (pos_ < end) {
          int tc_th = byte();
          uint8_t counts[16];
          int total = 0;
          for (int i = 0; i < 16; i++) total += counts[i] = (uint8_t)byte();
          if ((tc_th >> 4) > 1 || (tc_th & 15) > 3 || total > 256 || pos_ + (size_t)total > end) return fail("bad DHT");
          Huffman& h = (tc_th >> 4) ? ac_[tc_th & 15] : dc_[tc_th & 15];
          if (!h.build(counts, p_ + pos_, total)) return fail("bad Huffman table");
          pos_ += (size_t)total;
        }
        break;
      case 0xdb:  // DQT
        while (pos_ < end) {
          int pq_tq = byte();
          if ((pq_tq >> 4) > 1 || (pq_tq & 15) > 3) return fail("bad DQT");
          for (int i = 0; i < 64; i++) quant_[pq_tq & 15][kZigZag[i]] = (uint16_t)((pq_tq >> 4) ? be16() : byte());
        }
        break;
      case 0xdd:  // DRI
        restart_interval_ = be16();
        break;
      case 0xe0:  // APP0: JFIF?
        if (len >= 7 && !std::memcmp(p_ + pos_, "JFIF\0", 5)) jfif_ = true;
        break;
      case 0xee:  // APP14: Adobe colour transform flag
        if (len >= 14 && !std::memcmp(p_ + pos_, "Adobe\0", 6)) adobe_transform_ = p_[pos_ + 11];
        break;
      default:
        break;
    }
    pos_ = end;
    return true;
  }
  bool frame_header(bool progressive) {
    if (have_frame_) return fail("second frame header");
    if (byte() != 8) return fail("only 8-bit JPEG is supported");
    height_ = be16();
    width_ = be16();
    ncomp_ = byte();
    if (width_ <= 0 || height_ <= 0 || width_ > 32768 || height_ > 32768) return fail("unsupported JPEG extent");
    if (ncomp_ != 1 && ncomp_ != 3) return fail("unsupported JPEG component count");
    progressive_ = progressive;
    static const char rgb[3] = {'R', 'G', 'B'};
    for (int i = 0; i < ncomp_; i++) {
      Component& c = comp_[i];
      c.id = byte();
      if (ncomp_ == 3 && c.id == rgb[i]) rgb_ids_++;
      int hv = byte();
      c.h = hv >> 4;
      c.v = hv & 15;
      c.tq = byte();
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component");
      if (c.h > hmax_) hmax_ = c.h;
      if (c.v > vmax_) vmax_ = c.v;
    }
    for (int i = 0; i < ncomp_; i++)
      if (hmax_ % comp_[i].h || vmax_ % comp_[i].v) return fail("unsupported sampling factors");
    mcux_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_);
    mcuy_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
    for (int i = 0; i < ncomp_; i++) {
      Component& c = comp_[i];
      c.x = (width_ * c.h + hmax_ - 1) / hmax_;
      c.y = (height_ * c.v + vmax_ - 1) / vmax_;
      c.bw = mcux_ * c.h;
      c.bh = mcuy_ * c.v;
      c.coef.assign((size_t)c.bw * c.bh * 64, 0);
    }
    have_frame_ = true;
    return true;
  }
  bool scan_header() {
    if (!have_frame_) return fail("scan before the frame header");
    int len = be16();
    scan_n_ = byte();
    if (scan_n_ < 1 || scan_n_ > ncomp_ || len != 6 + 2 * scan_n_) return fail("bad scan header");
    for (int i = 0; i < scan_n_; i++) {
      int id = byte(), tables = byte(), which = -1;
      for (int k = 0; k < ncomp_; k++)
        if (comp_[k].id == id) which = k;
      if (which < 0) return fail("scan names an unknown component");
      order_[i] = which;
      comp_[which].td = tables >> 4;
      comp_[which].ta = tables & 15;
      if (comp_[which].td > 3 || comp_[which].ta > 3) return fail("bad table selector");
    }
    ss_ = byte();
    se_ = byte();
    int a = byte();
    ah_ = a >> 4;
    al_ = a & 15;
    if (progressive_) {
      if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) return fail("bad progressive scan parameters");
      if (ss_ > 0 && scan_n_ != 1) return fail("interleaved AC scan");
    } else {
      ss_ = 0;
      se_ = 63;
      ah_ = al_ = 0;
    }
    return true;
  }

  // ---- entropy-coded data of one scan
  bool block_baseline(Component& c, int16_t* d) {
    const Huffman &hd = dc_[c.td], &ha = ac_[c.ta];
    if (!hd.present || !ha.present) return fail("scan uses a missing Huffman table");
    const uint16_t* q = quant_[c.tq];
    int t = decode_symbol(hd);
    if (t < 0 || t > 15) return fail("bad DC code");
    c.dc_pred += receive_extend(t);
    d[0] = (int16_t)(c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
      int rs = decode_symbol(ha);
      if (rs < 0) return false;
      int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (rs != 0xf0) break;
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return fail("AC run past the block");
      int z = kZigZag[k++];
      d[z] = (int16_t)(receive_extend(s) * q[z]);
    }
    return true;
  }
  bool block_dc_progressive(Component& c, int16_t* d) {
    if (ah_ == 0) {
      const Huffman& hd = dc_[c.td];
      if (!hd.present) return fail("scan uses a missing Huffman table");
      int t = decode_symbol(hd);
      if (t < 0 || t > 15) return fail("bad DC code");
      c.dc_pred += receive_extend(t);
      d[0] = (int16_t)(c.dc_pred * (1 << al_));
    } else if (get_bit()) {
      d[0] = (int16_t)(d[0] + (1 << al_));
    }
    return true;
  }
  bool block_ac_progressive(Component& c, int16_t* d) {
    const Huffman& ha = ac_[c.ta];
    if (!ha.present) return fail("scan uses a missing Huffman table");
    if (ah_ == 0) {  // first pass over this band
      if (eob_run_) {
        eob_run_--;
        return true;
      }
      for (int k = ss_; k <= se_;) {
        int rs = decode_symbol(ha);
        if (rs < 0) return false;
        int r = rs >> 4, s = rs & 15;
        if (s == 0) {
          if (r < 15) {
            eob_run_ = (1 << r) - 1;
            if (r) eob_run_ += get_bits(r);
            break;
          }
          k += 16;
        } else {
          k += r;
          if (k > 63) return fail("AC run past the block");
          d[kZigZag[k++]] = (int16_t)(receive_extend(s) * (1 << al_));
        }
      }
      return true;
    }
    // refinement (T.81 G.1.2.3): one more bit for every coefficient already non-zero, new +-1 values between
    const int16_t bit = (int16_t)(1 << al_);
    auto refine = [&](int16_t& v) {
      if (get_bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    };
    int k = ss_;
    if (eob_run_ == 0) {
      while (k <= se_) {
        int rs = decode_symbol(ha);
        if (rs < 0) return false;
        int r = rs >> 4, s = rs & 15, value = 0;
        if (s == 0) {
          if (r < 15) {
            eob_run_ = (1 << r);  // includes this block, decremented below
            if (r) eob_run_ += get_bits(r);
            break;
          }
          // r == 15: skip 16 zero coefficients (refining the non-zero ones on the way)
        } else {
          if (s != 1) return fail("bad refinement code");
          value = get_bit() ? bit : -bit;
        }
        while (k <= se_) {
          int16_t& v = d[kZigZag[k++]];
          if (v != 0) {
            refine(v);
          } else {
            if (r == 0) {
              if (value) v = (int16_t)value;
              break;
            }
            r--;
          }
        }
      }
    }
    if (eob_run_) {
      for (; k <= se_; k++) {
        int16_t& v = d[kZigZag[k]];
        if (v != 0) refine(v);
      }
      eob_run_--;
    }
    return true;
  }
  bool one_block(Component& c, int bx, int by) {
    int16_t* d = &c.coef[((size_t)by * c.bw + bx) * 64];
    if (!progressive_) return block_baseline(c, d);
    return ss_ == 0 ? block_dc_progressive(c, d) : block_ac_progressive(c, d);
  }
  bool restart_if_due(int& todo) {
    if (restart_interval_ == 0 || --todo > 0) return true;
    // the interval is over: a restart marker must follow (RSTn), entropy state starts afresh
    if (!hit_marker_) {  // byte-align: drop what is left of the accumulator and look at the stream
      nbits_ = 0;
      acc_ = 0;
    }
    size_t save = pos_;
    int m = next_marker();
    if (m >= 0xd0 && m <= 0xd7) {
      reset_entropy();
      todo = restart_interval_;
      return true;
    }
    pos_ = save;  // no restart marker: the scan is over (or the file is damaged); stop cleanly
    todo = 0x7fffffff;
    hit_marker_ = true;
    return true;
  }
  bool entropy_data() {
    reset_entropy();
    int todo = restart_interval_ ? restart_interval_ : 0x7fffffff;
    if (scan_n_ == 1) {  // non-interleaved: the component's own blocks, row by row (unpadded extent)
      Component& c = comp_[order_[0]];
      int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
      for (int by = 0; by < h; by++)
        for (int bx = 0; bx < w; bx++) {
          if (!one_block(c, bx, by)) return false;
          if (!restart_if_due(todo)) return false;
        }
    } else {
      for (int my = 0; my < mcuy_; my++)
        for (int mx = 0; mx < mcux_; mx++) {
          for (int i = 0; i < scan_n_; i++) {
            Component& c = comp_[order_[i]];
            for (int y = 0; y < c.v; y++)
              for (int x = 0; x < c.h; x++)
                if (!one_block(c, mx * c.h + x, my * c.v + y)) return false;
          }
          if (!restart_if_due(todo)) return false;
        }
    }
    return true;
  }

  // ---- coefficients -> pixels
  void reconstruct(Image& img) {
    for (int i = 0; i < ncomp_; i++) {
      Component& c = comp_[i];
      const int stride = c.bw * 8;
      c.plane.assign((size_t)stride * c.bh * 8, 0);
      const uint16_t* q = quant_[c.tq];
      for (int by = 0; by < c.bh; by++)
        for (int bx = 0; bx < c.bw; bx++) {
          int16_t* d = &c.coef[((size_t)by * c.bw + bx) * 64];
          if (progressive_)
            for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * q[k]);  // baseline blocks were scaled as they were read
          idct_block(&c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride, d);
        }
    }
    img.w = (uint32_t)width_;
    img.h = (uint32_t)height_;
    img.rgba.assign((size_t)width_ * height_ * 4, 255);
    const bool is_rgb = ncomp_ == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
    std::vector<uint8_t> line[3];
    struct Up {
      int hs, vs, ystep, ypos, w_lores;
      const uint8_t *line0, *line1;
    } up[3];
    for (int i = 0; i < ncomp_; i++) {
      line[i].assign((size_t)width_ + 8, 0);
      up[i].hs = hmax_ / comp_[i].h;
      up[i].vs = vmax_ / comp_[i].v;
      up[i].ystep = up[i].vs >> 1;
      up[i].ypos = 0;
      up[i].w_lores = (width_ + up[i].hs - 1) / up[i].hs;
      up[i].line0 = up[i].line1 = comp_[i].plane.data();
    }
    for (int j = 0; j < height_; j++) {
      const uint8_t* row[3] = {nullptr, nullptr, nullptr};
      for (int i = 0; i < ncomp_; i++) {
        Up& u = up[i];
        const bool bottom = u.ystep >= (u.vs >> 1);
        const uint8_t *near = bottom ? u.line1 : u.line0, *far = bottom ? u.line0 : u.line1;
        row[i] = upsample(line[i].data(), near, far, u.w_lores, u.hs, u.vs);
        if (++u.ystep >= u.vs) {
          u.ystep = 0;
          u.line0 = u.line1;
          if (++u.ypos < comp_[i].y) u.line1 += (size_t)comp_[i].bw * 8;
        }
      }
      uint8_t* out = &img.rgba[(size_t)j * width_ * 4];
      if (ncomp_ == 1) {
        for (int x = 0; x < width_; x++) out[4 * x] = out[4 * x + 1] = out[4 * x + 2] = row[0][x];
      } else if (is_rgb) {
        for (int x = 0; x < width_; x++) {
          out[4 * x] = row[0][x];
          out[4 * x + 1] = row[1][x];
          out[4 * x + 2] = row[2][x];
        }
      } else {
        auto fx = [](float c) { return ((int)(c * 4096.0f + 0.5f)) << 8; };
        static const int cr_r = fx(1.40200f), cr_g = fx(0.71414f), cb_g = fx(0.34414f), cb_b = fx(1.77200f);
        for (int x = 0; x < width_; x++) {
          int yy = (row[0][x] << 20) + (1 << 19), cb = row[1][x] - 128, cr = row[2][x] - 128;
          int r = yy + cr * cr_r;
          int g = yy + cr * -cr_g + (int)((uint32_t)(cb * -cb_g) & 0xffff0000u);
          int b = yy + cb * cb_b;
          out[4 * x] = (uint8_t)clamp255(r >> 20);
          out[4 * x + 1] = (uint8_t)clamp255(g >> 20);
          out[4 * x + 2] = (uint8_t)clamp255(b >> 20);
        }
      }
    }
  }
  // one output row of a component from its near (3/4) and far (1/4) sample rows
  static const uint8_t* upsample(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
      for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
      return out;
    }
    if (hs == 2 && vs == 1) {
      if (w == 1) {
        out[0] = out[1] = near[0];
        return out;
      }
      out[0] = near[0];
      out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
      int i;
      for (i = 1; i < w - 1; i++) {
        int n = 3 * near[i] + 2;
        out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
        out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
      }
      out[2 * i] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
      out[2 * i + 1] = near[w - 1];
      return out;
    }
    if (hs == 2 && vs == 2) {
      if (w == 1) {
        out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
        return out;
      }
      int t1 = 3 * near[0] + far[0];
      out[0] = (uint8_t)((t1 + 2) >> 2);
      for (int i = 1; i < w; i++) {
        int t0 = t1;
        t1 = 3 * near[i] + far[i];
        out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
      }
      out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
      return out;
    }
    for (int i = 0; i < w; i++)
      for (int k = 0; k < hs; k++) out[i * hs + k] = near[i];
    return out;
  }
};

}  // namespace detail

inline bool decode(const uint8_t* p, size_t n, Image& img, std::string* err) {
  detail::Decoder d(p, n);
  return d.run(img, err);
}

}  // namespace svrjpeg
