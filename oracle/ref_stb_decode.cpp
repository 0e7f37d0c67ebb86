// ref_stb_decode.cpp — TEST INFRASTRUCTURE.  Runs the reference's own image decoder on a file.
//
// The reference decodes glTF textures with stb_image (vendored at thirdparty/stb_image/stb_image.h,
// called as stbi_load(..., 4) / stbi_load_from_memory(..., 4) in src/vk_loader.cpp:94, 108, 131).  That
// single header compiles with g++ alone, so it is the one piece of the reference that can be RUN in
// this container: oracle/Makefile builds this driver against the header where it lies under
// /root/reference (nothing is copied), output oracle/_ref/stb_decode.  tests/make_golden_images.py
// uses it to produce tests/golden/images.npz, the fixtures that pin host/svr_png.h and
// host/svr_jpeg.h; the binary itself never runs on the GPU box (no /root/reference there).
//
//   stb_decode <image file> <out.bin>     out.bin = u32 width, u32 height, width*height*4 RGBA bytes
#include <cstdint>
#include <cstdio>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: stb_decode <image> <out.bin>\n");
    return 2;
  }
  int w = 0, h = 0, n = 0;
  unsigned char* data = stbi_load(argv[1], &w, &h, &n, 4);
  if (!data) {
    std::fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason());
    return 1;
  }
  FILE* f = std::fopen(argv[2], "wb");
  if (!f) return 1;
  uint32_t dims[2] = {(uint32_t)w, (uint32_t)h};
  std::fwrite(dims, 4, 2, f);
  std::fwrite(data, 1, (size_t)w * h * 4, f);
  std::fclose(f);
  stbi_image_free(data);
  return 0;
}
