// svr_engine.cpp — see svr_engine.h.  Every function names the reference code it stands for.
#include "svr_engine.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace svrhost {

bool SvrApi::load(const std::string& path, std::string* err) {
  handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!handle) {
    if (err) *err = dlerror();
    return false;
  }
  bool ok = true;
#define SVR_LOAD(name)                                                   \
  name = reinterpret_cast<decltype(name)>(dlsym(handle, #name));          \
  if (!name) {                                                           \
    ok = false;                                                          \
    if (err) *err = std::string("missing symbol ") + #name;              \
  }
  SVR_LOAD(svr_create) SVR_LOAD(svr_destroy) SVR_LOAD(svr_upload_mesh) SVR_LOAD(svr_create_image)
  SVR_LOAD(svr_create_sampler) SVR_LOAD(svr_write_material) SVR_LOAD(svr_clear_color) SVR_LOAD(svr_draw_geometry)
  SVR_LOAD(svr_sync) SVR_LOAD(svr_read_color) SVR_LOAD(svr_read_depth) SVR_LOAD(svr_get_stats) SVR_LOAD(svr_last_error)
  SVR_LOAD(svr_backend_name) SVR_LOAD(svr_draw_background) SVR_LOAD(svr_read_swapchain) SVR_LOAD(svr_copy_to_swapchain)
  SVR_LOAD(svr_set_option)
#undef SVR_LOAD
  return ok;
}
void SvrApi::unload() {
  if (handle) dlclose(handle);
  handle = nullptr;
}

// ---------------------------------------------------------------- Camera (src/camera.cpp)
mat4 Camera::get_rotation_matrix() const {  // :61-66
  svrm::quat pitch_rotation = svrm::angle_axis(pitch, vec3{1, 0, 0});
  svrm::quat yaw_rotation = svrm::angle_axis(yaw, vec3{0, -1, 0});
  return svrm::mul(svrm::to_mat4(yaw_rotation), svrm::to_mat4(pitch_rotation));
}
mat4 Camera::get_view_matrix() const {  // :54-59
  mat4 camera_translation = svrm::translate(svrm::identity(), position);
  return svrm::inverse(svrm::mul(camera_translation, get_rotation_matrix()));
}
void Camera::update() {  // :8-11
  mat4 r = get_rotation_matrix();
  svrm::vec4 d = svrm::mul(r, svrm::vec4{velocity.x * 0.5f, velocity.y * 0.5f, velocity.z * 0.5f, 0.f});
  position.x += d.x;
  position.y += d.y;
  position.z += d.z;
}

// ---------------------------------------------------------------- scene graph
void Node::refresh_transform(const mat4& parent_matrix) {  // src/vk_types.h:157-163
  world_transform = svrm::mul(parent_matrix, local_transform);
  for (auto& c : children) c->refresh_transform(parent_matrix);  // NOT world_transform: the reference's quirk (D8)
}
void Node::Draw(const mat4& top_matrix, DrawContext& ctx) {  // src/vk_types.h:165-169
  for (auto& c : children) c->Draw(top_matrix, ctx);
}
void MeshNode::Draw(const mat4& top_matrix, DrawContext& ctx) {  // src/vk_engine.cpp:1716-1736
  mat4 node_matrix = svrm::mul(world_transform, top_matrix);     // world * top, as written there
  for (auto& s : mesh->surfaces) {
    SvrRenderObject obj{};
    obj.material = s.material->handle;
    obj.index_count = s.count;
    obj.first_index = s.startIndex;
    obj.mesh = mesh->meshBuffers;
    obj.bounds = s.bounds;
    std::memcpy(obj.transform, node_matrix.data(), 64);
    if (s.material->pass_type == SVR_PASS_TRANSPARENT)
      ctx.transparent_surfaces.push_back(obj);
    else
      ctx.opaque_surfaces.push_back(obj);
  }
  Node::Draw(top_matrix, ctx);
}
void LoadedScene::Draw(const mat4& top_matrix, DrawContext& ctx) {  // src/vk_loader.cpp:56-60
  for (auto& n : top_nodes) n->Draw(top_matrix, ctx);
}

SvrBounds loader_bounds(const std::vector<SvrVertex>& v, size_t initial_vtx) {  // src/vk_loader.cpp:366-375
  float mn[3], mx[3];
  for (int k = 0; k < 3; k++) mn[k] = mx[k] = v[initial_vtx].position[k];
  for (const SvrVertex& vert : v)
    for (int k = 0; k < 3; k++) {
      mn[k] = std::min(mn[k], vert.position[k]);
      mx[k] = std::max(mx[k], vert.position[k]);
    }
  SvrBounds b{};
  for (int k = 0; k < 3; k++) {
    b.origin[k] = (mx[k] + mn[k]) / 2.f;
    b.extents[k] = (mx[k] - mn[k]) / 2.f;
  }
  b.sphere_radius = std::sqrt(b.extents[0] * b.extents[0] + b.extents[1] * b.extents[1] + b.extents[2] * b.extents[2]);
  return b;
}

// ---------------------------------------------------------------- engine
bool SvrEngine::init(const std::string& library_path, uint32_t w, uint32_t h) {
  width = w;
  height = h;
  if (!api.load(library_path, &error)) return false;
  SvrConfig cfg{};
  cfg.width = w;
  cfg.height = h;
  cfg.color_format = SVR_COLOR_RGBA16F;  // _draw_image format, src/vk_engine.cpp:749
  cfg.device = device;
  if (api.svr_create(&cfg, &ctx)) {
    error = api.svr_last_error();
    return false;
  }
  // init_default_data, src/vk_engine.cpp:226-283: bytes are R,G,B,A in memory (__builtin_bswap32)
  const uint8_t white[4] = {0xFF, 0xFF, 0xFF, 0xFF}, grey[4] = {0xAA, 0xAA, 0xAA, 0xFF}, black[4] = {0, 0, 0, 0xFF};
  white_image = create_image(white, 1, 1, false);
  grey_image = create_image(grey, 1, 1, false);
  black_image = create_image(black, 1, 1, false);
  std::vector<uint8_t> pixels(32 * 32 * 4);
  for (int x = 0; x < 32; x++)
    for (int y = 0; y < 32; y++) {
      bool magenta = ((x % 2) ^ (y % 2)) != 0;
      uint8_t* p = &pixels[(y * 32 + x) * 4];
      p[0] = magenta ? 0xFF : 0;
      p[1] = 0;
      p[2] = magenta ? 0xFF : 0;
      p[3] = 0xFF;
    }
  error_checkerboard_image = create_image(pixels.data(), 32, 32, false);
  SvrSamplerDesc sampl{};  // zero-initialised VkSamplerCreateInfo: mip NEAREST, lods 0
  sampl.mag_filter = sampl.min_filter = SVR_FILTER_NEAREST;
  api.svr_create_sampler(ctx, &sampl, &default_sampler_nearest);
  sampl.mag_filter = sampl.min_filter = SVR_FILTER_LINEAR;
  api.svr_create_sampler(ctx, &sampl, &default_sampler_linear);
  const float ones[4] = {1, 1, 1, 1};
  auto m = write_material(SVR_PASS_MAIN_COLOR, ones, white_image, default_sampler_linear);
  if (!m) return false;
  default_data = *m;
  init_camera();
  return white_image && error_checkerboard_image;
}

void SvrEngine::cleanup() {
  if (ctx) api.svr_destroy(ctx);
  ctx = nullptr;
  api.unload();
}

void SvrEngine::init_camera() {  // src/vk_engine.cpp:203-210
  main_camera.velocity = vec3{0, 0, 0};
  main_camera.position = vec3{30.f, 0.f, -85.f};
  main_camera.pitch = 0.f;
  main_camera.yaw = 0.f;
}

SvrMesh SvrEngine::upload_mesh(const std::vector<uint32_t>& indices, const std::vector<SvrVertex>& vertices) {
  SvrMesh h = 0;
  if (api.svr_upload_mesh(ctx, indices.data(), indices.size(), vertices.data(), vertices.size(), &h)) error = api.svr_last_error();
  return h;
}
SvrImage SvrEngine::create_image(const void* rgba8, uint32_t w, uint32_t h, bool mipmapped) {
  SvrImage img = 0;
  if (api.svr_create_image(ctx, rgba8, w, h, mipmapped ? 1 : 0, &img)) error = api.svr_last_error();
  return img;
}
std::shared_ptr<MaterialInstance> SvrEngine::write_material(int pass, const float color_factors[4], SvrImage image,
                                                            SvrSampler sampler) {
  const float mr[4] = {1.f, 0.5f, 0.f, 0.f};  // src/vk_engine.cpp:275
  auto m = std::make_shared<MaterialInstance>();
  m->pass_type = pass;
  if (api.svr_write_material(ctx, pass, color_factors, mr, image, sampler, &m->handle)) {
    error = api.svr_last_error();
    return nullptr;
  }
  return m;
}

void SvrEngine::update_scene() {  // src/vk_engine.cpp:1479-1512
  auto t0 = std::chrono::system_clock::now();
  main_draw_context.opaque_surfaces.clear();
  main_camera.update();
  mat4 view = main_camera.get_view_matrix();
  for (auto& kv : loaded_scenes) kv.second->Draw(svrm::identity(), main_draw_context);
  mat4 proj = svrm::perspective(svrm::radians(70.f), (float)width / (float)height, 10000.f, 0.1f);
  proj.m[1][1] *= -1;
  mat4 viewproj = svrm::mul(proj, view);
  std::memcpy(scene_data.view, view.data(), 64);
  std::memcpy(scene_data.proj, proj.data(), 64);
  std::memcpy(scene_data.viewproj, viewproj.data(), 64);
  for (int k = 0; k < 4; k++) {
    scene_data.ambient_color[k] = 0.1f;
    scene_data.sunlight_color[k] = 1.f;
  }
  const float dir[4] = {0, 1, 0.5f, 1.f};
  std::memcpy(scene_data.sunlight_direction, dir, 16);
  auto t1 = std::chrono::system_clock::now();
  stats.scene_update_time = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() / 1000.f;
}

bool SvrEngine::draw_background() {  // src/vk_engine.cpp:1341-1355: dispatch the current ComputeEffect
  if (background_effects.empty()) {  // init_background_pipelines, src/vk_engine.cpp:977-989
    ComputeEffect gradient, sky;
    gradient.name = "gradient";
    gradient.effect = SVR_BACKGROUND_GRADIENT;
    for (int k = 0; k < 8; k++) gradient.data[k] = 1.0f;
    sky.name = "sky";
    sky.effect = SVR_BACKGROUND_SKY;
    sky.data[0] = 0.1f; sky.data[1] = 0.2f; sky.data[2] = 0.4f; sky.data[3] = 0.97f;
    background_effects.push_back(gradient);
    background_effects.push_back(sky);
  }
  const ComputeEffect& effect = background_effects[(size_t)current_background_effect % background_effects.size()];
  if (api.svr_draw_background(ctx, effect.effect, effect.data)) {
    error = api.svr_last_error();
    return false;
  }
  return true;
}

bool SvrEngine::read_swapchain(std::vector<uint8_t>& out) {
  uint32_t sw = swapchain_width ? swapchain_width : width, sh = swapchain_height ? swapchain_height : height;
  out.resize((size_t)sw * sh * 4);
  if (api.svr_read_swapchain(ctx, sw, sh, SVR_SWAPCHAIN_B8G8R8A8, out.data(), out.size())) {
    error = api.svr_last_error();
    return false;
  }
  return true;
}

bool SvrEngine::draw_geometry() {  // src/vk_engine.cpp:1357-1477: the whole body is one call
  SvrStats st{};
  int rc = api.svr_draw_geometry(ctx, &scene_data, main_draw_context.opaque_surfaces.data(),
                                 main_draw_context.opaque_surfaces.size(), main_draw_context.transparent_surfaces.data(),
                                 main_draw_context.transparent_surfaces.size(), &st);
  if (rc) {
    error = api.svr_last_error();
    return false;
  }
  stats.drawcall_count = st.drawcall_count;
  stats.triangle_count = st.triangle_count;
  stats.mesh_draw_time = st.mesh_draw_time;
  main_draw_context.opaque_surfaces.clear();
  main_draw_context.transparent_surfaces.clear();
  return true;
}

bool SvrEngine::draw() {  // src/vk_engine.cpp:1218-1339 minus acquire/blit/ImGui/present
  update_scene();
  if (!draw_background()) return false;
  if (!draw_geometry()) return false;
  frame_number++;
  return true;
}

bool SvrEngine::read_color_rgba16f(std::vector<uint16_t>& out) {
  out.resize((size_t)width * height * 4);
  if (api.svr_read_color(ctx, out.data(), out.size() * 2, 0)) {
    error = api.svr_last_error();
    return false;
  }
  return true;
}
bool SvrEngine::read_depth(std::vector<float>& out) {
  out.resize((size_t)width * height);
  if (api.svr_read_depth(ctx, out.data(), out.size() * 4)) {
    error = api.svr_last_error();
    return false;
  }
  return true;
}

}  // namespace svrhost
