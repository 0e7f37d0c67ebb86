/*
 * svr.h — C ABI of the MI355X-native replacement for the Vulkan geometry pass of
 * imalexlee/simple-vk-renderer.
 *
 * Drop-in boundary: the reference has no FFI layer; the replaced path sits behind one C++ member,
 *     void VulkanEngine::draw_geometry(VkCommandBuffer cmd);        (src/vk_engine.h:189)
 * called once per frame from VulkanEngine::draw() (src/vk_engine.cpp:1265).  Every input of that
 * function is an implicit member of the engine singleton, so the ABI below makes them explicit:
 * POD mirrors of the reference structs + opaque u32 handles where the reference holds VkBuffer /
 * VkImage / VkSampler / pointers.  Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every call returns 0 on success or a negative SvrError; svr_last_error() gives the text
 *     (the reference aborts in VK_CHECK, src/vk_types.h:23-30; nothing here aborts or throws).
 *   - matrices are column-major float[16], exactly glm::mat4 memory layout.
 *   - a context is single-threaded (one context <-> one host thread), like the reference engine.
 *   - work is enqueued on the context's HIP stream and is asynchronous like a recorded
 *     VkCommandBuffer; svr_sync / svr_read_* are the fence waits.
 *   - handles are 1-based; 0 is never valid.
 *
 * Two libraries export exactly these symbols:
 *   simple-vk-renderer_amd/csrc/libsvr_hip.so   the product (HIP, gfx950)
 *   oracle/libsvr_oracle.so                     the CPU scalar oracle (test infrastructure only)
 */
#ifndef SVR_H
#define SVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SvrContext SvrContext;
typedef uint32_t SvrMesh;
typedef uint32_t SvrImage;
typedef uint32_t SvrSampler;
typedef uint32_t SvrMaterial;

enum SvrError {
  SVR_OK = 0,
  SVR_ERR_INVALID_ARGUMENT = -1,
  SVR_ERR_OUT_OF_MEMORY = -2,
  SVR_ERR_DEVICE = -3,      /* a HIP call failed (text in svr_last_error) */
  SVR_ERR_BAD_HANDLE = -4,
  SVR_ERR_UNSUPPORTED = -5,
  SVR_ERR_OVERFLOW = -6     /* an internal per-frame buffer overflowed twice in a row */
};

/* Colour target formats.  The reference draws into VK_FORMAT_R16G16B16A16_SFLOAT
 * (src/vk_engine.cpp:749); RGBA8 is what its blit hands the swapchain (src/vk_engine.cpp:1276). */
enum SvrColorFormat { SVR_COLOR_RGBA16F = 0, SVR_COLOR_RGBA8 = 1 };

/* MaterialPass, src/vk_types.h:127-131 */
enum SvrMaterialPass { SVR_PASS_MAIN_COLOR = 0, SVR_PASS_TRANSPARENT = 1, SVR_PASS_OTHER = 2 };

/* VkFilter / VkSamplerMipmapMode values used by src/vk_loader.cpp:26-54 */
enum SvrFilter { SVR_FILTER_NEAREST = 0, SVR_FILTER_LINEAR = 1 };
enum SvrMipmapMode { SVR_MIPMAP_NEAREST = 0, SVR_MIPMAP_LINEAR = 1 };

/* struct Vertex, src/vk_types.h:97-103 — 48 bytes, offsets 0/12/16/28/32 */
typedef struct SvrVertex {
  float position[3];
  float uv_x;
  float normal[3];
  float uv_y;
  float color[4];
} SvrVertex;

/* struct GPUSceneData, src/vk_types.h:118-125 — 240 bytes */
typedef struct SvrSceneData {
  float view[16];
  float proj[16];
  float viewproj[16];
  float ambient_color[4];
  float sunlight_direction[4];
  float sunlight_color[4];
} SvrSceneData;

/* struct Bounds, src/vk_loader.h:11-15 — 28 bytes */
typedef struct SvrBounds {
  float origin[3];
  float sphere_radius;
  float extents[3];
} SvrBounds;

/* struct RenderObject, src/vk_engine.h:29-38.  `mesh` stands for {index_buffer, vertex_buf_addr}
 * (both belong to one GPUMeshBuffers), `material` for MaterialInstance*. */
typedef struct SvrRenderObject {
  uint32_t index_count;
  uint32_t first_index;
  SvrMesh mesh;
  SvrMaterial material;
  SvrBounds bounds;
  float transform[16];
} SvrRenderObject;

/* VkSamplerCreateInfo fields the reference sets (src/vk_loader.cpp:197-211,
 * src/vk_engine.cpp:252-261); address mode is always REPEAT (zero-initialised). */
typedef struct SvrSamplerDesc {
  int32_t mag_filter;   /* SvrFilter */
  int32_t min_filter;   /* SvrFilter */
  int32_t mipmap_mode;  /* SvrMipmapMode */
  float min_lod;
  float max_lod;        /* VK_LOD_CLAMP_NONE = 1000.0f */
} SvrSamplerDesc;

/* struct EngineStats, src/vk_engine.h:16-22, plus device-side counters of this implementation. */
typedef struct SvrStats {
  float frame_time;         /* not produced by this path (run() fills it) */
  int32_t triangle_count;   /* sum index_count/3 over submitted draws, src/vk_engine.cpp:1456 */
  int32_t drawcall_count;   /* src/vk_engine.cpp:1455 */
  float scene_update_time;  /* not produced by this path */
  float mesh_draw_time;     /* host ms spent inside svr_draw_geometry (CPU record time) */
  /* extensions */
  float gpu_time_ms;             /* geometry_ms + binning_ms + tile_ms (0 unless SVR_OPT_KERNEL_TIMING is on) */
  uint32_t culled_draws;         /* opaque draws rejected by is_visible */
  uint32_t timed_passes;         /* passes averaged into the three *_ms fields below */
  uint64_t shaded_fragments;     /* fragment-shader evaluations (no helper lanes) */
  uint64_t rasterized_fragments; /* covered pixel-centre samples that reached the depth test (of every triangle: what a forward
                                  * rasteriser walks; uninstrumented passes may skip hidden triangles, SVR_OPT_TUNING bits 5/6) */
  uint64_t binned_triangles;     /* triangles that survived clip/cull/zero-area */
  uint64_t bin_entries;          /* (triangle, tile) pairs; like the three counters above: of the last
                                  * pass run with SVR_OPT_COUNT_FRAGMENTS (uninstrumented passes report nothing) */
  /* mean device ms per pass since SVR_OPT_KERNEL_TIMING was last set, from hipEvents recorded on
   * the pass's own stream between its kernels */
  float geometry_ms;             /* vertex + clip + setup kernels */
  float binning_ms;              /* bin count + scan + fill kernels */
  float tile_ms;                 /* the tile raster/shade kernel */
  uint32_t replayed_passes;      /* passes re-run after a queue overflow since svr_create (SVR_OPT_QUEUE_CAPS) */
} SvrStats;

typedef struct SvrConfig {
  int32_t device;         /* HIP device ordinal (ignored by the oracle) */
  uint32_t width;         /* _draw_extent, src/vk_engine.cpp:1251-1254 */
  uint32_t height;
  int32_t color_format;   /* SvrColorFormat */
  uint32_t reserved[4];
} SvrConfig;

/* engine bootstrap for this path: device + _draw_image/_depth_image (src/vk_engine.cpp:743-789) */
int svr_create(const SvrConfig* cfg, SvrContext** out);
void svr_destroy(SvrContext* ctx);

/* The HIP stream all work is enqueued on (a hipStream_t; NULL = the legacy default stream).
 * Stands for the graphics queue + command buffer of src/vk_engine.cpp:1243-1321. */
int svr_set_stream(SvrContext* ctx, void* hip_stream);

/* Render into caller-owned device memory instead of the context's own targets (pass NULL to
 * restore).  color: width*height*(8|4) bytes, depth: width*height*4 bytes, row-major, no padding.
 * Stands for _draw_image / _depth_image being engine-owned VkImages (src/vk_engine.h:118-119). */
int svr_bind_targets(SvrContext* ctx, void* color_dev, void* depth_dev);
int svr_get_targets(SvrContext* ctx, void** color_dev, void** depth_dev);

/* VulkanEngine::upload_mesh(span<u32>, span<Vertex>) -> GPUMeshBuffers  (src/vk_engine.cpp:340-390) */
int svr_upload_mesh(SvrContext* ctx, const uint32_t* indices, size_t n_indices,
                    const SvrVertex* vertices, size_t n_vertices, SvrMesh* out);
int svr_destroy_mesh(SvrContext* ctx, SvrMesh mesh);

/* VulkanEngine::create_image(void* data, extent, R8G8B8A8_UNORM, usage, mipmapped)
 * (src/vk_engine.cpp:1571-1612) incl. vkutil::generate_mipmaps (src/vk_images.cpp:66-133) */
int svr_create_image(SvrContext* ctx, const void* rgba8, uint32_t width, uint32_t height,
                     int mipmapped, SvrImage* out);
int svr_destroy_image(SvrContext* ctx, SvrImage image);
/* read one mip level back (tightly packed RGBA8); w and h receive its size.  Test hook. */
int svr_read_image_level(SvrContext* ctx, SvrImage image, uint32_t level, void* dst, size_t bytes,
                         uint32_t* w, uint32_t* h);

/* vkCreateSampler call sites: src/vk_loader.cpp:197-211, src/vk_engine.cpp:252-261 */
int svr_create_sampler(SvrContext* ctx, const SvrSamplerDesc* desc, SvrSampler* out);

/* GLTFMettallicRoughness::write_material (src/vk_engine.cpp:1690-1714) with the
 * MaterialConstants it points at (src/vk_engine.h:52-57).  Only the colour texture is sampled by
 * mesh.frag; metal_rough_factors are carried but unused, as in the reference shaders. */
int svr_write_material(SvrContext* ctx, int pass, const float color_factors[4],
                       const float metal_rough_factors[4], SvrImage color_image,
                       SvrSampler color_sampler, SvrMaterial* out);

/* Result of draw_background (src/vk_engine.cpp:1341-1355, gradient_color.comp with
 * data1 == data2): fill the colour target with one RGBA value.  The fill covers the rows of the
 * current scissor (all rows unless the multi-GPU path narrowed it: a rank only owns its band).
 * The fill may be deferred into the pass that follows (which then writes the clear value to the pixels
 * it does not cover: loadOp CLEAR instead of a clear command); every other svr_* call that reads, writes,
 * exposes or re-targets the colour target runs it first, so the library's own results never differ.
 * Only foreign work on a caller-bound target between the clear and the next svr_* call would see the
 * old contents: call svr_sync first, or set SVR_OPT_TUNING bit 2. */
int svr_clear_color(SvrContext* ctx, const float rgba[4]);

/* VulkanEngine::draw_background (src/vk_engine.cpp:1341-1355): run one of the two ComputeEffects of
 * init_background_pipelines (src/vk_engine.cpp:920-1000) over the colour target; `data` is the
 * effect's ComputePushConstants (4 x vec4, src/vk_engine.h ComputePushConstants).
 *   SVR_BACKGROUND_GRADIENT  shaders/gradient_color.comp:14-27: mix(data1, data2, float(y) / height)
 *                            (engine default data1 = data2 = 1: what svr_clear_color(1,1,1,1) gives)
 *   SVR_BACKGROUND_SKY       shaders/sky.comp: data1.xyz * y / height + star field, threshold data1.w
 *                            (engine default data1 = (0.1, 0.2, 0.4, 0.97))
 * Like svr_clear_color it writes the rows of the current scissor; `height` is the full target's.
 * cos / fract / pow(.,6) of sky.comp are defined operation by operation in DESIGN.md (C15). */
enum SvrBackground { SVR_BACKGROUND_GRADIENT = 0, SVR_BACKGROUND_SKY = 1 };
int svr_draw_background(SvrContext* ctx, int effect, const float data[16]);

/* vkutil::copy_image (src/vk_images.cpp:33-64) as VulkanEngine::draw() uses it (src/vk_engine.cpp:
 * 1268-1280): LINEAR-filter blit of the whole colour target to a dst_width x dst_height image of the
 * swapchain's format (B8G8R8A8_UNORM, src/vk_engine.cpp:553) — identity-sized unless the window was
 * resized.  dst_dev: device memory, dst_width*dst_height*4 bytes, row-major (stands for the
 * swapchain image); stream-ordered like a pass.  With the identity extent only the rows of the current
 * scissor are written (a rank of the multi-GPU path presents its band; full scissor = everything); a
 * scaled blit always writes the whole image.  svr_read_swapchain blits the whole image into an
 * internal buffer and copies it to the host (tests, screenshots). */
enum SvrSwapchainFormat { SVR_SWAPCHAIN_B8G8R8A8 = 0, SVR_SWAPCHAIN_R8G8B8A8 = 1 };
int svr_copy_to_swapchain(SvrContext* ctx, void* dst_dev, uint32_t dst_width, uint32_t dst_height, int dst_format);
int svr_read_swapchain(SvrContext* ctx, uint32_t dst_width, uint32_t dst_height, int dst_format, void* dst_host,
                       size_t bytes);

/* vkCmdSetScissor (src/vk_engine.cpp:1431-1437).  The reference always passes the full extent and
 * so does svr_create; the multi-GPU path gives each rank its band of rows.  The viewport stays
 * (0,0,width,height).  Pixels outside the scissor are left untouched (colour AND depth). */
int svr_set_scissor(SvrContext* ctx, uint32_t x, uint32_t y, uint32_t w, uint32_t h);

/* The interleaved multi-GPU partition ("row_tile % G == r"): of the 32-row tile rows of the
 * scissor, counted from its first row (tile row t = rows y + 32 t .. y + 32 t + 31), this context renders those with
 * t % stride == offset; stride 1 (the default) = all of them.  Everything that is said to write "the rows of the
 * scissor" then writes the context's own tile rows — passes (colour and depth), svr_copy_to_swapchain at the identity
 * extent (in place: row y of the target is row y of the image) — or at least those (svr_clear_color,
 * svr_draw_background).  Triangles that meet none of the rows are dropped behind the vertex stage.  svr_get_row_costs
 * numbers the context's own tile rows 0, 1, ...: its row i is tile row i * stride + offset.  A context that owns no tile
 * row draws nothing (svr_draw_geometry returns SVR_OK).  1 <= stride <= 64. */
int svr_set_row_interleave(SvrContext* ctx, uint32_t stride, uint32_t offset);

/* Where a present reports whether it was carried out: every svr_copy_to_swapchain from now on also stores, on the
 * stream, 0 to *status_dev when it wrote its rows and 1 when it was void — a pass in front of it overflowed an
 * internal queue and everything since awaits the replay (SVR_OPT_QUEUE_CAPS).  The replay runs the present again
 * and stores 2.  For callers that hand the image on by their own stream work (the exchange of include/svr_dist.h),
 * which the replay cannot see: they read the word behind their own work; 0 = the rows they handed on were final;
 * anything else = they were stale (1) or may have been rewritten under the reader (2): fence (svr_sync), clear the
 * word and hand the image on again.  NULL (the default): no report.  The oracle takes a host pointer and always
 * stores 0. */
int svr_set_present_status(SvrContext* ctx, uint32_t* status_dev);

/* VulkanEngine::draw_geometry (src/vk_engine.cpp:1357-1477): cull opaque with is_visible, sort,
 * begin rendering (colour LOAD, depth CLEAR 0.0), draw opaque then transparent with
 * mesh.vert/mesh.frag, end.  The arrays are borrowed for the duration of the call. */
int svr_draw_geometry(SvrContext* ctx, const SvrSceneData* scene,
                      const SvrRenderObject* opaque, size_t n_opaque,
                      const SvrRenderObject* transparent, size_t n_transparent,
                      SvrStats* out_stats);

/* BASELINE config 1: one render pass drawing 3 vertices with colored_triangle.vert/.frag
 * (shaders/colored_triangle.vert:6-25, .frag:9-12) and the opaque pipeline state. */
int svr_draw_colored_triangle(SvrContext* ctx, SvrStats* out_stats);

/* BASELINE config 2: one render pass with _mesh_pipeline (src/vk_engine.cpp:1006-1051):
 * colored_triangle_mesh.vert (gl_Position = render_matrix * pos) + tex_image.frag
 * (outColor = texture(displayTexture, uv), alpha included). */
int svr_draw_tex_image(SvrContext* ctx, SvrMesh mesh, uint32_t first_index, uint32_t index_count,
                       const float render_matrix[16], SvrImage image, SvrSampler sampler,
                       SvrStats* out_stats);

/* mesh.vert as a stand-alone operator over a vertex range (shaders/mesh.vert:29-38): per vertex
 * writes gl_Position (4 floats) to out_clip and normal.xyz,color.xyz,uv.xy (8 floats) to
 * out_varyings, both host pointers.  Per-stage parity hook. */
int svr_run_mesh_vert(SvrContext* ctx, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices,
                      const float world[16], const SvrSceneData* scene, SvrMaterial material,
                      float* out_clip, float* out_varyings);

/* The other two vertex programs as stand-alone operators (same purpose as svr_run_mesh_vert):
 *   SVR_VS_COLORED_TRIANGLE       shaders/colored_triangle.vert:6-25: gl_VertexIndex = first_vertex + i (at most 3
 *                                 vertices; mesh and render_matrix are ignored and may be 0 / NULL)
 *   SVR_VS_COLORED_TRIANGLE_MESH  shaders/colored_triangle_mesh.vert:28-38: gl_Position = render_matrix * vec4(position, 1)
 * out_clip: 4 floats per vertex; out_varyings: 8 floats per vertex laid out like mesh.vert's (the three normal slots
 * are 0: these programs have no such output; then color.xyz, uv.xy). */
enum SvrVertexShader { SVR_VS_COLORED_TRIANGLE = 1, SVR_VS_COLORED_TRIANGLE_MESH = 2 };
int svr_run_vertex_shader(SvrContext* ctx, int shader, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices,
                          const float render_matrix[16], float* out_clip, float* out_varyings);

/* Implementation switches (the reference has compile-time flags only).
 * SVR_OPT_COUNT_FRAGMENTS: 1 = count rasterized/shaded fragments and binned triangles with device
 * atomics (instrumented kernels; keep 0 for timed runs).
 * SVR_OPT_KERNEL_TIMING: 1 = time the tile kernel of every pass with the start/stop events of its own
 * dispatch (no extra packets in the stream), averaged into SvrStats.tile_ms as passes are validated;
 * 2 = hipEvent records around geometry, binning and tiles (five more packets per pass, a few
 * microseconds of stream bubbles each: for profiling, not for timed runs) -> SvrStats.{geometry,
 * binning,tile}_ms; setting the option (to 0, 1 or 2) resets the averages.
 * SVR_OPT_TILE_CYCLES: 1 = every tile workgroup records the shader-clock cycles of its phases
 * (svr_debug_read_tile_cycles); five s_memtime reads per tile, off by default.
 * SVR_OPT_TUNING: bit mask that switches individual optimisations OFF (A/B timing inside one
 * process; results are identical either way).  bit0: tile kernel walks tiles row-major instead of
 * heaviest-first.  bit1: geometry + binning run on the caller's stream instead of overlapping the
 * previous pass's tile stage on an internal stream.  bit2: svr_clear_color runs at once instead of riding in
 * the next pass.  bit3: heavy tiles are rendered by one workgroup instead of four row quarters.  bit4: finished passes are
 * looked at by fences only (svr_sync, read-backs, svr_get_stats), not in passing by other calls: a queue overflow
 * (SVR_OPT_QUEUE_CAPS) is then always found late — what the tests of the replay behind an exchange need.
 * bit5: the tile kernel's visibility phase walks every triangle; bit6: it runs its hierarchical depth test (triangles
 * that cannot win in any 8x8 block they reach are dropped before they are walked) in every pass.  With neither bit the
 * library chooses per pass: the test runs where bins are deep (96 entries per tile on average).  Passes run with
 * SVR_OPT_COUNT_FRAGMENTS drop nothing — rasterized_fragments stays every covered sample, as the forward rasteriser
 * counts them — and check the test instead: a dropped triangle's fragment that wins fails the pass (SVR_ERR_DEVICE).
 * SVR_OPT_DEVICE_FLATTEN: where svr_draw_geometry's host half runs — is_visible, the sort and the
 * per-object draw records (src/vk_engine.cpp:1361-1378, 1412-1457).  0 (default): on the device from
 * 2048 objects up, on the host below; 1: always on the device; 2: always on the host.  Same frames
 * either way.  Device-side, drawcall_count / triangle_count / culled_draws are not known when
 * svr_draw_geometry returns (its out_stats holds 0 there): svr_get_stats has them after the pass.
 * SVR_OPT_QUEUE_CAPS: initial capacity (entries) of the pass-internal queues — clip queue, clipper
 * output records, bin/pair lists — instead of the generous defaults; 0 restores the defaults.  A pass
 * that overflows a queue writes nothing, and is replayed with grown queues before its results can be
 * observed, so results never depend on this (tests set it tiny to exercise the replay). */
enum SvrOption {
  SVR_OPT_COUNT_FRAGMENTS = 1,
  SVR_OPT_KERNEL_TIMING = 2,
  SVR_OPT_TILE_CYCLES = 3,
  SVR_OPT_TUNING = 4,
  SVR_OPT_QUEUE_CAPS = 5,
  SVR_OPT_DEVICE_FLATTEN = 6
};
int svr_set_option(SvrContext* ctx, int option, int64_t value);

/* Parity test hook: ask the next instrumented pass (SVR_OPT_COUNT_FRAGMENTS = 1) to record the
 * intermediates of the fragment shader invocation that produced pixel (x,y) — barycentrics, 1/w,
 * uv, derivatives, texel, normal, colour, light, output — and read the 64 floats back.  x < 0
 * switches tracing off.  Both libraries fill the same slots (see k_tile.hip / svr_oracle.cpp). */
int svr_debug_trace_pixel(SvrContext* ctx, int x, int y);
int svr_debug_read_trace(SvrContext* ctx, float out[64]);
/* Profiling hook: per-tile triangle counts of the last pass (32x32-pixel tiles of the scissor,
 * row-major; first the opaque bins, then the transparent bins).  *n_tiles receives the tile count;
 * counts may be NULL to query it, else must hold 2 * n_tiles entries.  HIP library only. */
int svr_debug_read_bins(SvrContext* ctx, uint32_t* counts, size_t capacity, uint32_t* n_tiles);
/* Profiling hook: shader-clock cycles each tile's workgroup spent in its four phases (visibility,
 * shading, transparent layers, write-back) during the last pass run with SVR_OPT_TILE_CYCLES: 4 * n_tiles entries.
 * HIP library only. */
int svr_debug_read_tile_cycles(SvrContext* ctx, uint32_t* cycles, size_t capacity);

/* Load-balancing hook of the multi-GPU form (rank r renders a band of rows): the cost estimate
 * of the most recently VALIDATED pass per 32-pixel tile row of its scissor — sum over the row's tiles of
 * 40 + opaque bin entries / 8 + 3/4 transparent bin entries (thousands of cycles, the model the tile kernel's own
 * split rule is fitted to).  Does not wait for passes in flight: it reports the last one whose completion the
 * context has already seen (none yet: *n_tile_rows = 0).  *first_row / *n_rows: that pass's scissor rows; tile row t
 * covers rows first_row + 32 t .. + 31 (clipped to the scissor) — of the context's OWN tile rows when they are
 * interleaved (svr_set_row_interleave: its row t is tile row t * stride + offset).  costs may be NULL to query the
 * count.  HIP library: measured; oracle: 1 per tile row (it has no bins). */
int svr_get_row_costs(SvrContext* ctx, uint32_t* costs, size_t capacity, uint32_t* n_tile_rows, uint32_t* first_row,
                      uint32_t* n_rows);

/* Test hook for the arithmetic contract's "IEEE 1/x" (perspective divide, 1/area, 1/q per fragment): compares the
 * reciprocal exactly as the kernels compute it with the compiler's correctly rounded 1.0f / x for the fp32 bit
 * patterns first .. first + count - 1 (all 2^32 in one call is a few milliseconds).  variant 0 = the production
 * form, 1 / 2 = its candidate refinements without the fallback.  *mismatches = inputs whose result bits differ
 * (NaN == NaN), *refined = inputs that took the refined (division-free) path, first_bad = up to 16 of the
 * mismatching patterns; the last two may be NULL.  HIP library only. */
int svr_debug_rcp_sweep(SvrContext* ctx, int variant, uint64_t first, uint64_t count, uint64_t* mismatches,
                        uint64_t* refined, uint32_t first_bad[16]);

/* fence wait (vkWaitForFences, src/vk_engine.cpp:1226) */
int svr_sync(SvrContext* ctx);

/* Read the colour target back to host memory.  as_rgba8 != 0 converts RGBA16F texels with the
 * identity-extent blit of src/vk_images.cpp:33-64: clamp to [0,1], *255, round-to-nearest-even,
 * R,G,B,A byte order. */
int svr_read_color(SvrContext* ctx, void* dst, size_t bytes, int as_rgba8);
int svr_read_depth(SvrContext* ctx, float* dst, size_t bytes);

/* counters of the last pass; waits for it to finish */
int svr_get_stats(SvrContext* ctx, SvrStats* out);

const char* svr_last_error(void);
/* "hip-gfx950" or "cpu-oracle" */
const char* svr_backend_name(void);

#ifdef __cplusplus
}
#endif
#endif /* SVR_H */
