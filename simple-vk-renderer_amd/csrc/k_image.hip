// k_image.hip — image kernels either side of the geometry pass.
//   fill_color      result of draw_background (src/vk_engine.cpp:1341-1355, gradient_color.comp with
//                   data1 == data2): one constant RGBA over the whole colour target
//   downsample      one level of vkutil::generate_mipmaps (src/vk_images.cpp:95-128): 2:1 LINEAR blit
//                   in exact integer arithmetic (contract C13)
//   rgba16f_to_rgba8  identity-extent vkutil::copy_image (src/vk_images.cpp:33-64): clamp, *255, RNE
//   background      draw_background's two compute effects (gradient_color.comp, sky.comp)
//   blit            vkutil::copy_image in general: LINEAR-filter scaling blit to the swapchain format
// All are pure streaming kernels, grid-stride, HBM-bound.
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
// ------------------------------------------------------------------------------------------------
// draw_background (src/vk_engine.cpp:1341-1355): gradient_color.comp and sky.comp  (contract C14, C15).
// One lane per pixel of the scissor's rows, 8 (RGBA16F) or 4 (RGBA8) bytes written per lane.
__device__ __forceinline__ float pin32(float x) {  // keep "fp32 result, then the store's rounding" apart (see Codec)
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void store_target(void* color, int fmt, size_t p, float r, float g, float b, float a) {
  if (fmt == SVR_COLOR_RGBA16F) {
    r = pin32(r); g = pin32(g); b = pin32(b); a = pin32(a);
    uint32_t hr = __half_as_ushort(__float2half_rn(r)), hg = __half_as_ushort(__float2half_rn(g));
    uint32_t hb = __half_as_ushort(__float2half_rn(b)), ha = __half_as_ushort(__float2half_rn(a));
    reinterpret_cast<uint2*>(color)[p] = make_uint2(hr | (hg << 16), hb | (ha << 16));
  } else {
    auto un8 = [](float f) { return (uint32_t)__float2int_rn(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f); };
    reinterpret_cast<uint32_t*>(color)[p] = un8(r) | (un8(g) << 8) | (un8(b) << 16) | (un8(a) << 24);
  }
}

// cos for sky.comp's hash, operation by operation as the oracle's sky_cos (three-term Cody-Waite by
// pi/2 with fma, minimax sin/cos on the reduced argument)
__device__ __forceinline__ float sky_cos(float x) {
  const float kTwoOverPi = 0x1.45f306p-1f, kPio2Hi = 0x1.921fb6p+0f, kPio2Mid = -0x1.777a5cp-25f, kPio2Lo = -0x1.ee59dap-50f;
  float k = rintf(x * kTwoOverPi);
  float r = fmaf(-k, kPio2Hi, x);
  r = fmaf(-k, kPio2Mid, r);
  r = fmaf(-k, kPio2Lo, r);
  float z = r * r;
  float sn = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  float cs = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                  fmaf(-0.5f, z, 1.0f));
  int q = (int)k & 3;
  float v = (q & 1) ? sn : cs;
  return (q == 1 || q == 2) ? -v : v;
}
__device__ __forceinline__ float sky_fract(float a) { return a - floorf(a); }
__device__ __forceinline__ float sky_noisy_star(float x, float y, float thr) {
  float s = sky_fract(415.92653f * (sky_cos(x * 37.0f) + sky_cos(y * 57.0f)));
  if (!(s >= thr)) return 0.0f;
  float t = (s - thr) / (1.0f - thr);
  float t2 = t * t, t4 = t2 * t2;
  return t4 * t2;
}

struct BackgroundData {
  float d[16];
};
__global__ __launch_bounds__(256) void background_kernel(void* color, int fmt, uint32_t W, uint32_t H, uint32_t y_first,
                                                         uint32_t n_rows, int effect, BackgroundData pc, const uint32_t* poison) {
  if (*poison) return;
  const float fh = (float)H;
  const uint32_t n = W * n_rows;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t x = i % W, y = y_first + i / W;
    float r, g, b, a;
    if (effect == SVR_BACKGROUND_GRADIENT) {
      float blend = (float)y / fh, inv = 1.0f - blend;
      r = fmaf(pc.d[4], blend, pc.d[0] * inv);
      g = fmaf(pc.d[5], blend, pc.d[1] * inv);
      b = fmaf(pc.d[6], blend, pc.d[2] * inv);
      a = fmaf(pc.d[7], blend, pc.d[3] * inv);
    } else {
      float fx = (float)x, fy = (float)y, thr = pc.d[3];
      float sx = fx + 0.2f, sy = fy + -0.06f;
      float frx = sky_fract(sx), fry = sky_fract(sy), gx = floorf(sx), gy = floorf(sy);
      float v1 = sky_noisy_star(gx, gy, thr), v2 = sky_noisy_star(gx, gy + 1.0f, thr);
      float v3 = sky_noisy_star(gx + 1.0f, gy, thr), v4 = sky_noisy_star(gx + 1.0f, gy + 1.0f, thr);
      float star = (v1 * (1.0f - frx)) * (1.0f - fry);
      star = fmaf(v2 * (1.0f - frx), fry, star);
      star = fmaf(v3 * frx, 1.0f - fry, star);
      star = fmaf(v4 * frx, fry, star);
      r = (pc.d[0] * fy) / fh + star;
      g = (pc.d[1] * fy) / fh + star;
      b = (pc.d[2] * fy) / fh + star;
      a = 1.0f;
    }
    store_target(color, fmt, (size_t)y * W + x, r, g, b, a);
  }
}
void launch_background(void* color, int color_format, uint32_t W, uint32_t H, uint32_t y_first, uint32_t n_rows, int effect,
                       const float data[16], const uint32_t* poison, hipStream_t s) {
  BackgroundData pc;
  for (int i = 0; i < 16; i++) pc.d[i] = data[i];
  hipLaunchKernelGGL(background_kernel, dim3(stream_grid(W * n_rows)), dim3(256), 0, s, color, color_format, W, H, y_first, n_rows,
                     effect, pc, poison);
}

// ------------------------------------------------------------------------------------------------
// vkutil::copy_image (src/vk_images.cpp:33-64): LINEAR blit of the colour target to the swapchain's
// format and extent (contract C16).  One lane per destination pixel; identity extent degenerates to
// weights 0 and is exactly the plain format conversion.
__device__ __forceinline__ float4 load_target(const void* color, int fmt, size_t p) {
  if (fmt == SVR_COLOR_RGBA16F) {
    uint2 e = reinterpret_cast<const uint2*>(color)[p];
    return make_float4(__half2float(__ushort_as_half((unsigned short)(e.x & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(e.x >> 16))),
                       __half2float(__ushort_as_half((unsigned short)(e.y & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(e.y >> 16))));
  }
  uint32_t t = reinterpret_cast<const uint32_t*>(color)[p];
  const float k = 0x1.010102p-8f;
  return make_float4((float)(t & 0xffu) * k, (float)((t >> 8) & 0xffu) * k, (float)((t >> 16) & 0xffu) * k, (float)(t >> 24) * k);
}
// rstride > 1 (identity extent only): of the 32-row tile rows counted from row_first, those with index % rstride == roff
// (svr_set_row_interleave) — n_rows is then 32 x the number of such tile rows and row_end the first row not to write.
// status (may be null): receives 1 when the blit is void (an earlier pass overflowed: the replay will run it again), else
// status_ok — 0 from the blit as first enqueued, 2 from the replay's (whoever read the rows in between cannot trust them)
__global__ __launch_bounds__(256) void blit_kernel(const void* color, int fmt, uint32_t W, uint32_t H, uint32_t* dst, uint32_t dw,
                                                   uint32_t dh, uint32_t row_first, uint32_t n_rows, int dst_format,
                                                   const uint32_t* poison, uint32_t rstride, uint32_t roff, uint32_t row_end,
                                                   uint32_t* status, uint32_t status_ok) {
  const uint32_t void_op = *poison;
  if (status && blockIdx.x == 0 && threadIdx.x == 0) *status = void_op ? 1u : status_ok;
  if (void_op) return;
  const float su = (float)W / (float)dw, sv = (float)H / (float)dh;
  const uint32_t n = dw * n_rows;  // destination rows [row_first, row_first + n_rows)
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    uint32_t i = k % dw, j = row_first + k / dw;
    if (rstride > 1u) {
      const uint32_t lr = k / dw;  // row among the owned ones
      j = row_first + ((lr >> 5) * rstride + roff) * 32u + (lr & 31u);
      if (j >= row_end) continue;
    }
    const uint32_t idx = j * dw + i;
    float u = ((float)i + 0.5f) * su - 0.5f, v = ((float)j + 0.5f) * sv - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv, i1 = i0 + 1, j1 = j0 + 1;
    i0 = min(max(i0, 0), (int)W - 1); i1 = min(max(i1, 0), (int)W - 1);
    j0 = min(max(j0, 0), (int)H - 1); j1 = min(max(j1, 0), (int)H - 1);
    float4 t00 = load_target(color, fmt, (size_t)j0 * W + i0), t10 = load_target(color, fmt, (size_t)j0 * W + i1);
    float4 t01 = load_target(color, fmt, (size_t)j1 * W + i0), t11 = load_target(color, fmt, (size_t)j1 * W + i1);
    auto filt = [&](float c00, float c10, float c01, float c11) {
      float top = fmaf(a, c10 - c00, c00), bot = fmaf(a, c11 - c01, c01);
      return fmaf(b, bot - top, top);
    };
    auto un8 = [](float f) { return (uint32_t)__float2int_rn(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f); };
    uint32_t r = un8(filt(t00.x, t10.x, t01.x, t11.x)), g = un8(filt(t00.y, t10.y, t01.y, t11.y));
    uint32_t bl = un8(filt(t00.z, t10.z, t01.z, t11.z)), al = un8(filt(t00.w, t10.w, t01.w, t11.w));
    dst[idx] = dst_format == SVR_SWAPCHAIN_B8G8R8A8 ? (bl | (g << 8) | (r << 16) | (al << 24)) : (r | (g << 8) | (bl << 16) | (al << 24));
  }
}
void launch_blit(const void* color, int color_format, uint32_t W, uint32_t H, void* dst, uint32_t dw, uint32_t dh, uint32_t row_first,
                 uint32_t n_rows, int dst_format, const uint32_t* poison, uint32_t rstride, uint32_t roff, uint32_t row_end, uint32_t* status,
                 uint32_t status_ok, hipStream_t s) {
  hipLaunchKernelGGL(blit_kernel, dim3(stream_grid(std::max(dw * n_rows, 1u))), dim3(256), 0, s, color, color_format, W, H, (uint32_t*)dst, dw,
                     dh, row_first, n_rows, dst_format, poison, rstride, roff, row_end, status, status_ok);
}

// Test hook: the contract's "IEEE 1/x" as the kernels compute it (rcp_ieee / its candidate refinements)
// against the compiler's correctly rounded division, over a range of fp32 bit patterns.  out[0] = mismatches,
// out[1] = inputs that took the refined path, out[2..17] = the first mismatching patterns.
template <int VARIANT>
__global__ __launch_bounds__(256) void rcp_sweep_kernel(unsigned long long first, unsigned long long count, unsigned long long* out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0, fast = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const uint32_t bits = (uint32_t)(first + i);
    const float x = u2f(bits);
    float want = 1.0f / x, got;
    if (VARIANT == 0) {
      got = rcp_ieee(x);
      fast += rcp_fast_ok(x) ? 1u : 0u;
    } else {
      got = rcp_fast_ok(x) ? rcp_refined<VARIANT>(x) : want;
      fast += rcp_fast_ok(x) ? 1u : 0u;
    }
    const bool same = f2u(want) == f2u(got) || (want != want && got != got);
    if (!same) {
      unsigned long long k = bad++;
      (void)k;
      unsigned long long slot = atomicAdd(&out[18], 1ull);
      if (slot < 16) out[2 + slot] = bits;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_down(bad, off);
    fast += __shfl_down(fast, off);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], fast);
  }
}
void launch_rcp_sweep(int variant, unsigned long long first, unsigned long long count, unsigned long long* out, hipStream_t s) {
  dim3 grid(4096), block(256);
  if (variant == 1) hipLaunchKernelGGL(rcp_sweep_kernel<1>, grid, block, 0, s, first, count, out);
  else if (variant == 2) hipLaunchKernelGGL(rcp_sweep_kernel<2>, grid, block, 0, s, first, count, out);
  else hipLaunchKernelGGL(rcp_sweep_kernel<0>, grid, block, 0, s, first, count, out);
}

void launch_rgba16f_to_rgba8(const void* src, void* dst, uint32_t n_pixels, hipStream_t s) {
  hipLaunchKernelGGL(cvt16f_to_8_kernel, dim3(stream_grid(n_pixels)), dim3(256), 0, s, (const uint2*)src,
                     (uint32_t*)dst, n_pixels);
}

}  // namespace svr
