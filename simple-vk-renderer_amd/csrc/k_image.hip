// k_image.hip — image kernels either side of the geometry pass.
//   fill_color      result of draw_background (src/vk_engine.cpp:1341-1355, gradient_color.comp with
//                   data1 == data2): one constant RGBA over the whole colour target
//   downsample      one level of vkutil::generate_mipmaps (src/vk_images.cpp:95-128): 2:1 LINEAR blit
//                   in exact integer arithmetic (contract C13)
//   rgba16f_to_rgba8  identity-extent vkutil::copy_image (src/vk_images.cpp:33-64): clamp, *255, RNE
// All three are pure streaming kernels: 16 bytes per lane, grid-stride, HBM-bound.
#include <hip/hip_fp16.h>

#include "svr_launch.h"

namespace svr {

__global__ __launch_bounds__(256) void fill16f_kernel(uint4* dst, uint32_t n_vec, uint32_t n_pixels, uint2 px, const uint32_t* poison) {
  if (*poison) return;  // an earlier pass is waiting for its replay: the host replays this clear after it
  // two RGBA16F pixels per 16-byte store
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += gridDim.x * blockDim.x)
    dst[i] = make_uint4(px.x, px.y, px.x, px.y);
  if (blockIdx.x == 0 && threadIdx.x == 0 && (n_pixels & 1u))
    reinterpret_cast<uint2*>(dst)[n_pixels - 1] = px;
}
__global__ __launch_bounds__(256) void fill8_kernel(uint32_t* dst, uint32_t n_pixels, uint32_t px, const uint32_t* poison) {
  if (*poison) return;
  uint32_t n_vec = n_pixels >> 2;
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += gridDim.x * blockDim.x)
    d4[i] = make_uint4(px, px, px, px);
  if (blockIdx.x == 0 && threadIdx.x < (n_pixels & 3u)) dst[(n_vec << 2) + threadIdx.x] = px;
}

static inline uint32_t stream_grid(uint32_t n_items) {
  uint32_t g = (n_items + 255u) / 256u;
  return g < 1u ? 1u : (g > 2048u ? 2048u : g);
}

void launch_fill_color(void* color, uint32_t n_pixels, int color_format, uint64_t packed_pixel, const uint32_t* poison,
                       hipStream_t s) {
  if (color_format == SVR_COLOR_RGBA16F) {
    uint2 px = make_uint2((uint32_t)packed_pixel, (uint32_t)(packed_pixel >> 32));
    uint32_t n_vec = n_pixels >> 1;
    hipLaunchKernelGGL(fill16f_kernel, dim3(stream_grid(n_vec)), dim3(256), 0, s, (uint4*)color, n_vec, n_pixels, px, poison);
  } else {
    hipLaunchKernelGGL(fill8_kernel, dim3(stream_grid(n_pixels >> 2)), dim3(256), 0, s, (uint32_t*)color, n_pixels,
                       (uint32_t)packed_pixel, poison);
  }
}

// one destination texel per lane; weights are exact rationals (see oracle downsample_level)
__global__ __launch_bounds__(256) void downsample_kernel(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst,
                                                         uint32_t dw, uint32_t dh) {
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= dw * dh) return;
  uint32_t i = idx % dw, j = idx / dw;
  long long nx = (long long)(2u * i + 1u) * sw - dw, dx = 2ll * dw;
  long long ny = (long long)(2u * j + 1u) * sh - dh, dy = 2ll * dh;
  long long i0 = nx >= 0 ? nx / dx : -((-nx + dx - 1) / dx);
  long long j0 = ny >= 0 ? ny / dy : -((-ny + dy - 1) / dy);
  long long wx1 = nx - i0 * dx, wx0 = dx - wx1;
  long long wy1 = ny - j0 * dy, wy0 = dy - wy1;
  long long i1 = i0 + 1, j1 = j0 + 1;
  i0 = min(max(i0, 0ll), (long long)sw - 1);
  i1 = min(max(i1, 0ll), (long long)sw - 1);
  j0 = min(max(j0, 0ll), (long long)sh - 1);
  j1 = min(max(j1, 0ll), (long long)sh - 1);
  uint32_t t00 = src[j0 * sw + i0], t10 = src[j0 * sw + i1], t01 = src[j1 * sw + i0], t11 = src[j1 * sw + i1];
  long long den = dx * dy;
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    long long a = (t00 >> (8 * c)) & 0xffu, b = (t10 >> (8 * c)) & 0xffu;
    long long cc = (t01 >> (8 * c)) & 0xffu, d = (t11 >> (8 * c)) & 0xffu;
    long long num = wy0 * (wx0 * a + wx1 * b) + wy1 * (wx0 * cc + wx1 * d);
    long long q = num / den, r = num - q * den;
    if (2 * r > den || (2 * r == den && (q & 1))) q++;
    out |= (uint32_t)q << (8 * c);
  }
  dst[idx] = out;
}

void launch_downsample(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh,
                       hipStream_t s) {
  uint32_t n = dw * dh;
  hipLaunchKernelGGL(downsample_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, (const uint32_t*)src, sw, sh,
                     (uint32_t*)dst, dw, dh);
}

__device__ __forceinline__ uint32_t h2un8(uint32_t hbits) {
  float f = __half2float(__ushort_as_half((unsigned short)hbits));
  return (uint32_t)__float2int_rn(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f);
}
// two pixels (16 bytes in, 8 bytes out) per lane
__global__ __launch_bounds__(256) void cvt16f_to_8_kernel(const uint2* src, uint32_t* dst, uint32_t n_pixels) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x) {
    uint2 p = src[i];
    dst[i] = h2un8(p.x & 0xffffu) | (h2un8(p.x >> 16) << 8) | (h2un8(p.y & 0xffffu) << 16) | (h2un8(p.y >> 16) << 24);
  }
}
void launch_rgba16f_to_rgba8(const void* src, void* dst, uint32_t n_pixels, hipStream_t s) {
  hipLaunchKernelGGL(cvt16f_to_8_kernel, dim3(stream_grid(n_pixels)), dim3(256), 0, s, (const uint2*)src,
                     (uint32_t*)dst, n_pixels);
}

}  // namespace svr
