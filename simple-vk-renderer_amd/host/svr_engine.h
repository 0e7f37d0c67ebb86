// svr_engine.h — a VulkanEngine-shaped C++ host above the C ABI of include/svr.h.
//
// Mirrors, member for member, the parts of the reference engine that sit either side of the draw
// path, with the Vulkan plumbing replaced by svr_* calls:
//   init / cleanup / run-less frame:   src/vk_engine.cpp:171-201, 1218-1339
//   init_default_data                  src/vk_engine.cpp:226-306   (white/grey/black/checker, samplers, default material)
//   init_camera, Camera                src/vk_engine.cpp:203-210, src/camera.cpp:8-11, 54-66
//   upload_mesh / create_image         src/vk_engine.cpp:340-390, 1571-1612
//   update_scene                       src/vk_engine.cpp:1479-1512
//   draw_geometry                      src/vk_engine.cpp:1357-1477  -> one svr_draw_geometry call
//   Node / MeshNode / LoadedGLTF::Draw src/vk_types.h:146-170, src/vk_engine.cpp:1716-1736, src/vk_loader.cpp:56-60
// The library is opened with dlopen so the same harness drives the HIP product or, in tests, the oracle.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/svr.h"
#include "svr_math.h"

namespace svrhost {

using svrm::mat4;
using svrm::vec3;

// every svr.h entry point the harness uses, resolved from one shared library
struct SvrApi {
  void* handle = nullptr;
#define SVR_FN(name) decltype(&::name) name = nullptr;
  SVR_FN(svr_create) SVR_FN(svr_destroy) SVR_FN(svr_upload_mesh) SVR_FN(svr_create_image) SVR_FN(svr_create_sampler)
  SVR_FN(svr_write_material) SVR_FN(svr_clear_color) SVR_FN(svr_draw_geometry) SVR_FN(svr_sync) SVR_FN(svr_read_color)
  SVR_FN(svr_read_depth) SVR_FN(svr_get_stats) SVR_FN(svr_last_error) SVR_FN(svr_backend_name)
  SVR_FN(svr_draw_background) SVR_FN(svr_read_swapchain) SVR_FN(svr_copy_to_swapchain) SVR_FN(svr_set_option)
#undef SVR_FN
  bool load(const std::string& path, std::string* err);
  void unload();
};

struct EngineStats {  // src/vk_engine.h:16-22
  float frame_time = 0;
  int triangle_count = 0;
  int drawcall_count = 0;
  float scene_update_time = 0;
  float mesh_draw_time = 0;
};

struct Camera {  // src/camera.h:9-31 (static state there; one instance here)
  vec3 velocity, position;
  float pitch = 0.f, yaw = 0.f;
  mat4 get_view_matrix() const;
  mat4 get_rotation_matrix() const;
  void update();
};

struct MaterialInstance {  // src/vk_types.h:138-142: pipeline/descriptor set -> one handle
  SvrMaterial handle = 0;
  int pass_type = SVR_PASS_MAIN_COLOR;
};
struct GeoSurface {  // src/vk_loader.h:17-22
  uint32_t startIndex = 0, count = 0;
  SvrBounds bounds{};
  std::shared_ptr<MaterialInstance> material;
};
struct MeshAsset {  // src/vk_loader.h:24-28
  std::string name;
  std::vector<GeoSurface> surfaces;
  SvrMesh meshBuffers = 0;
};
struct DrawContext {  // src/vk_engine.h:40-43
  std::vector<SvrRenderObject> opaque_surfaces, transparent_surfaces;
};

struct Node {  // src/vk_types.h:150-170
  std::weak_ptr<Node> parent;
  std::vector<std::shared_ptr<Node>> children;
  mat4 local_transform = svrm::identity();
  mat4 world_transform = svrm::identity();
  virtual ~Node() = default;
  void refresh_transform(const mat4& parent_matrix);  // passes parent_matrix on unchanged
  virtual void Draw(const mat4& top_matrix, DrawContext& ctx);
};
struct MeshNode : Node {  // src/vk_engine.h:24-27
  std::shared_ptr<MeshAsset> mesh;
  void Draw(const mat4& top_matrix, DrawContext& ctx) override;
};
struct LoadedScene {  // LoadedGLTF, src/vk_loader.h:33-57
  std::vector<std::shared_ptr<MeshAsset>> meshes;
  std::vector<std::shared_ptr<Node>> nodes, top_nodes;
  std::vector<std::shared_ptr<MaterialInstance>> materials;
  std::vector<SvrImage> images;      // file-owned images (not the engine defaults), LoadedGLTF::images
  std::vector<SvrSampler> samplers;  // LoadedGLTF::samplers
  void Draw(const mat4& top_matrix, DrawContext& ctx);
};

struct ComputeEffect {  // src/vk_engine.h ComputeEffect: name + ComputePushConstants (4 x vec4)
  const char* name = "";
  int effect = SVR_BACKGROUND_GRADIENT;
  float data[16] = {0};
};

// GLTF loader's bounds rule (src/vk_loader.cpp:366-375): min/max start at the primitive's first vertex
// but run over every vertex accumulated in the mesh so far
SvrBounds loader_bounds(const std::vector<SvrVertex>& mesh_vertices_so_far, size_t initial_vtx);

struct SvrEngine;
// load_gltf_meshes (src/vk_loader.cpp:162-437): .glb or .gltf -> uploaded scene; nullptr + engine->error on failure
std::shared_ptr<LoadedScene> load_gltf_meshes(SvrEngine* engine, const std::string& file_path);

struct SvrEngine {
  SvrApi api;
  SvrContext* ctx = nullptr;
  uint32_t width = 1700, height = 900;  // _window_extent, src/vk_engine.h:219
  int device = 0;                        // SvrConfig.device: the rank's GPU in the sharded form (svr_dist.h)
  int frame_number = 0;
  EngineStats stats;
  DrawContext main_draw_context;
  SvrSceneData scene_data{};
  Camera main_camera;
  std::unordered_map<std::string, std::shared_ptr<LoadedScene>> loaded_scenes;
  // init_default_data
  SvrImage white_image = 0, grey_image = 0, black_image = 0, error_checkerboard_image = 0;
  SvrSampler default_sampler_nearest = 0, default_sampler_linear = 0;
  MaterialInstance default_data;
  std::string error;

  bool init(const std::string& library_path, uint32_t w, uint32_t h);
  void cleanup();
  SvrMesh upload_mesh(const std::vector<uint32_t>& indices, const std::vector<SvrVertex>& vertices);
  SvrImage create_image(const void* rgba8, uint32_t w, uint32_t h, bool mipmapped);
  std::shared_ptr<MaterialInstance> write_material(int pass, const float color_factors[4], SvrImage image, SvrSampler sampler);
  void init_camera();
  void update_scene();
  // init_background_pipelines (src/vk_engine.cpp:920-1000): gradient (white, white) and sky (0.1,0.2,0.4,0.97)
  std::vector<ComputeEffect> background_effects;
  int current_background_effect = 0;
  uint32_t swapchain_width = 0, swapchain_height = 0;  // _swap_chain_extent; 0 = the draw extent
  bool draw_background();
  bool draw_geometry();
  bool draw();  // update_scene -> draw_background -> draw_geometry (ImGui/present have no counterpart)
  // the swapchain image of the frame just drawn: vkutil::copy_image at src/vk_engine.cpp:1277 (B8G8R8A8)
  bool read_swapchain(std::vector<uint8_t>& out);
  bool read_color_rgba16f(std::vector<uint16_t>& out);
  bool read_depth(std::vector<float>& out);
};

}  // namespace svrhost
