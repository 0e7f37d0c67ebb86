// svr_launch.h — host-callable launchers of the HIP kernels (one per kernel file).
#pragma once
#include "svr_device.h"

namespace svr {

// k_geometry.hip
// copy_bytes / zero_bytes are rounded up to 16: both buffers must be padded accordingly
// n_draws: DrawDesc records at the head of the copy whose mvp the kernel fills in (viewproj * mat)
void launch_prologue(const void* host_src, void* dst, size_t copy_bytes, void* zero, size_t zero_bytes, uint32_t n_draws,
                     const SvrSceneData& scene, hipStream_t s);
void launch_setup(const FrameParams& P, hipStream_t s);
void launch_mesh_vert(const SvrVertex* vtx, uint32_t first, uint32_t n, const float* world16,
                      const float* viewproj16, const float* color_factors4, float* out_clip,
                      float* out_varyings, hipStream_t s);
// vtx == nullptr: colored_triangle.vert (index only); else colored_triangle_mesh.vert with matrix16 (device memory)
void launch_vertex_shader(const SvrVertex* vtx, uint32_t first, uint32_t n, const float* matrix16, float* out_clip,
                          float* out_varyings, hipStream_t s);
// k_flatten.hip
void launch_flatten(const FlattenParams& F, hipStream_t s);
// k_bin.hip
void launch_bin_count(const FrameParams& P, hipStream_t s);
void launch_bin_scan(const FrameParams& P, hipStream_t s);
void launch_bin_fill(const FrameParams& P, hipStream_t s, hipEvent_t done);  // done: signalled with the kernel (may be null)
// k_tile.hip
void launch_tiles(const FrameParams& P, int color_format, bool count_fragments, hipStream_t s, hipEvent_t start, hipEvent_t done);
// k_image.hip
// packed_pixel: the already encoded texel (RGBA16F: 4 halves, RGBA8: low 32 bits)
// poison: the context's sticky overflow flag (the clear is void while it is raised)
void launch_fill_color(void* color, uint32_t n_pixels, int color_format, uint64_t packed_pixel, const uint32_t* poison,
                       hipStream_t s);
void launch_background(void* color, int color_format, uint32_t W, uint32_t H, uint32_t y_first, uint32_t n_rows, int effect,
                       const float data[16], const uint32_t* poison, hipStream_t s);
// rstride / roff / row_end: the interleaved tile rows of svr_set_row_interleave (1, 0, row_first + n_rows: all rows);
// status: device word that receives 1 when the blit was void (poison), else status_ok (may be null)
void launch_blit(const void* color, int color_format, uint32_t W, uint32_t H, void* dst, uint32_t dw, uint32_t dh, uint32_t row_first,
                 uint32_t n_rows, int dst_format, const uint32_t* poison, uint32_t rstride, uint32_t roff, uint32_t row_end, uint32_t* status,
                 uint32_t status_ok, hipStream_t s);
void launch_downsample(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh,
                       hipStream_t s);
void launch_rgba16f_to_rgba8(const void* src, void* dst, uint32_t n_pixels, hipStream_t s);
void launch_rcp_sweep(int variant, unsigned long long first, unsigned long long count, unsigned long long* out19, hipStream_t s);

}  // namespace svr
