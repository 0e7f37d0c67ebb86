// svr_demo.cpp — drives the VulkanEngine-shaped harness (svr_engine.h) for a few frames and dumps what it
// submitted and what came back.  Used by tests/test_host_cpp.py (against the oracle on CPU, against the HIP
// library on the GPU box) and as the smallest example of the call sequence init -> load -> draw().
//
//   svr_demo --lib <libsvr_*.so> --width 160 --height 90 --frames 2 --dump /tmp/prefix
//   svr_demo --lib libsvr_hip.so --dist libsvr_dist.so --ranks 2 [--transport shm|rccl] [--bounds 0,13,90] [--rebalance 1]
//       the sharded frame (include/svr_dist.h): one process per rank (forked before anything touches the GPU;
//       rccl: rank r on device r; shm: every rank on device 0), every rank dumps the exchanged image as
//       <prefix>.rank<r>.swapchain — it must be the single-process <prefix>.swapchain
//
// Scene: a three-level node hierarchy of cubes (exercises Node::refresh_transform's parent_matrix quirk and
// MeshNode::Draw's world*top order) with an opaque default material and a Transparent checker one.
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/svr_dist.h"
#include "svr_engine.h"
#include "svr_image.h"

using namespace svrhost;

static void cube(std::vector<uint32_t>& idx, std::vector<SvrVertex>& vtx) {
  // 24 vertices / 36 indices, per-face uv in [0,1]^2, axial normals, colour 1 (same table as scenes.cube_mesh)
  const float faces[6][3][3] = {
      {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}},  {{0, 0, -1}, {-1, 0, 0}, {0, 1, 0}}, {{1, 0, 0}, {0, 0, -1}, {0, 1, 0}},
      {{-1, 0, 0}, {0, 0, 1}, {0, 1, 0}}, {{0, 1, 0}, {1, 0, 0}, {0, 0, -1}},  {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}}};
  const int corner[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  for (int f = 0; f < 6; f++) {
    uint32_t base = (uint32_t)vtx.size();
    for (int c = 0; c < 4; c++) {
      SvrVertex v{};
      for (int k = 0; k < 3; k++) {
        v.position[k] = faces[f][0][k] * 0.5f + faces[f][1][k] * ((float)corner[c][0] - 0.5f) + faces[f][2][k] * ((float)corner[c][1] - 0.5f);
        v.normal[k] = faces[f][0][k];
      }
      v.uv_x = (float)corner[c][0];
      v.uv_y = (float)corner[c][1];
      v.color[0] = v.color[1] = v.color[2] = v.color[3] = 1.f;
      vtx.push_back(v);
    }
    const uint32_t q[6] = {0, 1, 2, 0, 2, 3};
    for (uint32_t i : q) idx.push_back(base + i);
  }
}

template <class T>
static void dump(const std::string& path, const T* p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

static std::vector<pid_t> g_kids;  // rank 0 of --ranks N: the other ranks' processes
static bool g_exit_ok = false;

int main(int argc, char** argv) {
  std::string lib, prefix, gltf, png, dist_lib, transport = "shm", bounds_arg;
  int ranks = 1, rebalance = 0, queue_caps = 0, partition = 0, pick_partition = 0;
  uint32_t w = 160, h = 90;
  int frames = 2, background = 0;
  float cam[5] = {0, 0, 0, 0, 0};  // position, pitch, yaw
  uint32_t sw = 0, sh = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string a = argv[i];
    if (a == "--lib") lib = argv[i + 1];
    else if (a == "--gltf") gltf = argv[i + 1];
    else if (a == "--dist") dist_lib = argv[i + 1];
    else if (a == "--ranks") ranks = atoi(argv[i + 1]);
    else if (a == "--transport") transport = argv[i + 1];
    else if (a == "--bounds") bounds_arg = argv[i + 1];
    else if (a == "--rebalance") rebalance = atoi(argv[i + 1]);
    else if (a == "--queue-caps") queue_caps = atoi(argv[i + 1]);   // SVR_OPT_QUEUE_CAPS: tiny queues, passes overflow and are replayed
    else if (a == "--partition") partition = std::string(argv[i + 1]) == "interleaved" ? 1 : 0;  // bands | interleaved
    else if (a == "--pick") pick_partition = atoi(argv[i + 1]);     // after N frames of each partition: keep the faster one
    else if (a == "--png") png = argv[i + 1];
    else if (a == "--background") background = atoi(argv[i + 1]);
    else if (a == "--swapchain" && sscanf(argv[i + 1], "%ux%u", &sw, &sh) == 2) {}
    else if (a == "--camera" && sscanf(argv[i + 1], "%f,%f,%f,%f,%f", &cam[0], &cam[1], &cam[2], &cam[3], &cam[4]) == 5) {}
    else if (a == "--width") w = (uint32_t)atoi(argv[i + 1]);
    else if (a == "--height") h = (uint32_t)atoi(argv[i + 1]);
    else if (a == "--frames") frames = atoi(argv[i + 1]);
    else if (a == "--dump") prefix = argv[i + 1];
  }
  if (!png.empty()) {  // decoder check: PNG file -> <prefix>.rgba (tests/test_gltf_loader.py)
    std::ifstream f(png, std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::string err;
    svrimg::Image img;
    const char* format = "";
    if (!svrimg::decode(bytes.data(), bytes.size(), img, &err, &format)) {
      fprintf(stderr, "%s: %s\n", format, err.c_str());
      return 1;
    }
    uint32_t iw = img.w, ih = img.h;
    std::vector<uint8_t> rgba;
    rgba.swap(img.rgba);
    printf("png %u %u %s\n", iw, ih, format);
    if (!prefix.empty()) dump(prefix + ".rgba", rgba.data(), rgba.size());
    return 0;
  }
  if (lib.empty()) {
    fprintf(stderr, "usage: svr_demo --lib <shared library exporting svr.h> [--width W --height H --frames N --dump prefix]\n"
                    "                [--gltf file.glb|file.gltf --camera x,y,z,pitch,yaw] [--background 0|1] [--swapchain WxH]\n");
    return 2;
  }
  // the sharded frame: one process per rank, forked before anything touches the GPU; rank 0 makes the id
  int rank = 0;
  std::vector<int> id_pipe_r(ranks > 1 ? ranks : 0, -1), id_pipe_w(ranks > 1 ? ranks : 0, -1);
  if (ranks > 1) {
    if (dist_lib.empty()) {
      fprintf(stderr, "--ranks needs --dist <libsvr_dist.so>\n");
      return 2;
    }
    for (int r = 1; r < ranks; r++) {
      int fd[2];
      if (pipe(fd) != 0) return 1;
      id_pipe_r[r] = fd[0];
      id_pipe_w[r] = fd[1];
    }
    std::vector<pid_t> kids;
    for (int r = 1; r < ranks; r++) {
      pid_t pid = fork();
      if (pid < 0) return 1;
      if (pid == 0) {
        rank = r;
        kids.clear();
        break;
      }
      kids.push_back(pid);
    }
    if (rank == 0 && !kids.empty()) {  // rank 0 is the parent itself; it reaps the others at the end (below)
      g_kids = kids;
      // a rank that fails must not leave the others waiting for it (a test would sit out its timeout): when a child
      // exits with an error the parent kills the rest and goes; when the parent itself fails, it takes the children along
      struct sigaction sa {};
      sa.sa_handler = [](int) {
        int st = 0;
        pid_t p;
        while ((p = waitpid(-1, &st, WNOHANG)) > 0) {
          for (pid_t& k : g_kids)
            if (k == p) k = -1;
          if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
            for (pid_t k : g_kids)
              if (k > 0) kill(k, SIGKILL);
            _exit(3);
          }
        }
      };
      sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
      sigaction(SIGCHLD, &sa, nullptr);
      atexit([] {
        signal(SIGCHLD, SIG_DFL);
        for (pid_t p : g_kids) {
          if (p <= 0) continue;
          if (!g_exit_ok) kill(p, SIGKILL);
          int st = 0;
          waitpid(p, &st, 0);
        }
      });
    }
  }
  SvrEngine eng;
  eng.device = transport == "rccl" ? rank : 0;
  if (!eng.init(lib, w, h)) {
    fprintf(stderr, "init failed: %s\n", eng.error.c_str());
    return 1;
  }
  printf("backend %s, %ux%u\n", eng.api.svr_backend_name(), w, h);

  eng.current_background_effect = background;
  eng.swapchain_width = sw;
  eng.swapchain_height = sh;
  if (!gltf.empty()) {  // VulkanEngine::init: load_gltf_meshes(this, path) -> loaded_scenes["structure"] (src/vk_engine.cpp:192-198)
    auto loaded = load_gltf_meshes(&eng, gltf);
    if (!loaded) {
      fprintf(stderr, "%s\n", eng.error.c_str());
      return 1;
    }
    eng.loaded_scenes["structure"] = loaded;
    eng.main_camera.position = {cam[0], cam[1], cam[2]};
    eng.main_camera.pitch = cam[3];
    eng.main_camera.yaw = cam[4];
    size_t surfaces = 0;
    for (auto& m : loaded->meshes) surfaces += m->surfaces.size();
    printf("gltf %s: %zu meshes %zu surfaces %zu nodes %zu top nodes %zu materials %zu images %zu samplers\n", gltf.c_str(),
           loaded->meshes.size(), surfaces, loaded->nodes.size(), loaded->top_nodes.size(), loaded->materials.size(),
           loaded->images.size(), loaded->samplers.size());
  } else {
  // "load_gltf_meshes" by hand: one mesh with two primitives (two cubes' worth of geometry in one buffer)
  auto mesh = std::make_shared<MeshAsset>();
  mesh->name = "cubes";
  std::vector<uint32_t> idx;
  std::vector<SvrVertex> vtx;
  const float tint[4] = {0.4f, 0.3f, 0.2f, 1.f};
  auto transparent = eng.write_material(SVR_PASS_TRANSPARENT, tint, eng.error_checkerboard_image, eng.default_sampler_nearest);
  auto opaque = std::make_shared<MaterialInstance>(eng.default_data);
  for (int prim = 0; prim < 2; prim++) {
    size_t initial_vtx = vtx.size();
    std::vector<uint32_t> ci;
    std::vector<SvrVertex> cv;
    cube(ci, cv);
    for (auto& v : cv) v.position[0] += 1.25f * (float)prim;  // second primitive sits beside the first
    GeoSurface s;
    s.startIndex = (uint32_t)idx.size();
    s.count = (uint32_t)ci.size();
    for (uint32_t i : ci) idx.push_back(i + (uint32_t)initial_vtx);  // indices rebased, src/vk_loader.cpp:311-312
    vtx.insert(vtx.end(), cv.begin(), cv.end());
    s.bounds = loader_bounds(vtx, initial_vtx);
    s.material = prim == 0 ? opaque : transparent;
    mesh->surfaces.push_back(s);
  }
  mesh->meshBuffers = eng.upload_mesh(idx, vtx);

  auto scene = std::make_shared<LoadedScene>();
  scene->meshes.push_back(mesh);
  auto trs = [](svrm::vec3 t, svrm::quat q, svrm::vec3 s) {  // src/vk_loader.cpp:400-410: tm * rm * sm
    return svrm::mul(svrm::mul(svrm::translate(svrm::identity(), t), svrm::to_mat4(q)), svrm::scale(svrm::identity(), s));
  };
  auto root = std::make_shared<Node>();
  root->local_transform = trs({0, 0, -12}, {1, 0, 0, 0}, {1, 1, 1});
  auto child = std::make_shared<MeshNode>();
  child->mesh = mesh;
  child->local_transform = trs({-3, 0, -9}, {0.70710677f, 0, 0.70710677f, 0}, {2, 2, 2});
  auto grandchild = std::make_shared<MeshNode>();
  grandchild->mesh = mesh;
  grandchild->local_transform = trs({2, 1.5f, -7}, {1, 0, 0, 0}, {1, 1.5f, 1});
  root->children.push_back(child);
  child->parent = root;
  child->children.push_back(grandchild);
  grandchild->parent = child;
  scene->nodes = {root, child, grandchild};
  scene->top_nodes = {root};
  root->refresh_transform(svrm::identity());  // src/vk_loader.cpp:430-435
  eng.loaded_scenes["structure"] = scene;
  eng.main_camera.position = {0, 0, 0};
  }

  if (!dist_lib.empty()) {
    void* dh = dlopen(dist_lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dh) {
      fprintf(stderr, "dlopen %s: %s\n", dist_lib.c_str(), dlerror());
      return 1;
    }
#define DIST_FN(name) auto name##_ = reinterpret_cast<decltype(&::name)>(dlsym(dh, #name)); if (!name##_) { fprintf(stderr, "missing %s\n", #name); return 1; }
    DIST_FN(svr_dist_get_unique_id) DIST_FN(svr_dist_create) DIST_FN(svr_dist_destroy) DIST_FN(svr_dist_set_bounds) DIST_FN(svr_dist_get_bounds)
    DIST_FN(svr_dist_rebalance) DIST_FN(svr_dist_band) DIST_FN(svr_dist_begin_frame) DIST_FN(svr_dist_end_frame) DIST_FN(svr_dist_wait_frame)
    DIST_FN(svr_dist_read_frame) DIST_FN(svr_dist_last_error) DIST_FN(svr_dist_set_partition) DIST_FN(svr_dist_get_partition) DIST_FN(svr_dist_pick_partition)
    DIST_FN(svr_dist_replays)
#undef DIST_FN
    const int tr = transport == "rccl" ? SVR_DIST_RCCL : SVR_DIST_SHM;
    uint8_t id[SVR_DIST_ID_BYTES];
    if (rank == 0) {
      if (svr_dist_get_unique_id_(tr, id) != SVR_OK) {
        fprintf(stderr, "unique id: %s\n", svr_dist_last_error_());
        return 1;
      }
      for (int r = 1; r < ranks; r++)
        if (write(id_pipe_w[r], id, sizeof(id)) != (ssize_t)sizeof(id)) return 1;
    } else if (read(id_pipe_r[rank], id, sizeof(id)) != (ssize_t)sizeof(id)) {
      return 1;
    }
    SvrDist* dist = nullptr;
    if (svr_dist_create_(eng.ctx, tr, id, rank, ranks, w, h, SVR_SWAPCHAIN_B8G8R8A8, &dist) != SVR_OK) {
      fprintf(stderr, "rank %d: svr_dist_create: %s\n", rank, svr_dist_last_error_());
      return 1;
    }
    if (!bounds_arg.empty()) {
      std::vector<uint32_t> b;
      for (size_t p = 0; p < bounds_arg.size();) {
        b.push_back((uint32_t)strtoul(bounds_arg.c_str() + p, nullptr, 10));
        size_t c = bounds_arg.find(',', p);
        p = c == std::string::npos ? bounds_arg.size() : c + 1;
      }
      if (svr_dist_set_bounds_(dist, b.data(), b.size()) != SVR_OK) {
        fprintf(stderr, "rank %d: svr_dist_set_bounds: %s\n", rank, svr_dist_last_error_());
        return 1;
      }
    }
    // (SVR_OPT_TUNING bit 4 with it: the overflow is found at a fence, not in passing, so the present behind the
    // void pass is void too and the exchange carries stale rows — the case svr_dist_wait_frame repairs)
    if (queue_caps > 0 && (eng.api.svr_set_option(eng.ctx, SVR_OPT_QUEUE_CAPS, queue_caps) != SVR_OK ||
                           eng.api.svr_set_option(eng.ctx, SVR_OPT_TUNING, 16) != SVR_OK)) {
      fprintf(stderr, "rank %d: SVR_OPT_QUEUE_CAPS: %s\n", rank, eng.api.svr_last_error());
      return 1;
    }
    if (svr_dist_set_partition_(dist, partition) != SVR_OK) return 1;
    int in_flight = 0;
    std::vector<uint8_t> image((size_t)w * h * 4);
    double pick_ms[2] = {0, 0};
    for (int f = 0; f < frames; f++) {
      if (pick_partition > 0 && f <= 2 * pick_partition && f % pick_partition == 0) {
        // --pick N: N frames in bands, N interleaved, each timed by this rank; then the collective choice.  The switch
        // happens between two frames on every rank alike.
        while (in_flight > 0) {
          if (svr_dist_wait_frame_(dist, nullptr) != SVR_OK) return 1;
          in_flight--;
        }
        eng.api.svr_sync(eng.ctx);
        SvrStats st{};
        eng.api.svr_get_stats(eng.ctx, &st);
        if (f > 0) pick_ms[f / pick_partition - 1] = st.gpu_time_ms;
        if (f == 2 * pick_partition) {
          int picked = -1;
          if (svr_dist_pick_partition_(dist, (float)pick_ms[0], (float)pick_ms[1], &picked) != SVR_OK) {
            fprintf(stderr, "rank %d: pick_partition: %s\n", rank, svr_dist_last_error_());
            return 1;
          }
          if (rank == 0) printf("frame %d: bands %.4f ms, interleaved %.4f ms on this rank -> %s\n", f, pick_ms[0], pick_ms[1], picked ? "interleaved" : "bands");
        } else {
          if (svr_dist_set_partition_(dist, f / pick_partition) != SVR_OK) return 1;
          eng.api.svr_set_option(eng.ctx, SVR_OPT_KERNEL_TIMING, 2);  // resets the running means
        }
      }
      if (rebalance && f > 0 && f % rebalance == 0) {  // a collective between two frames: every rank, same frame
        eng.api.svr_sync(eng.ctx);
        SvrStats st{};
        eng.api.svr_get_stats(eng.ctx, &st);
        int changed = 0;
        if (svr_dist_rebalance_(dist, st.tile_ms, &changed) != SVR_OK) {
          fprintf(stderr, "rank %d: rebalance: %s\n", rank, svr_dist_last_error_());
          return 1;
        }
        std::vector<uint32_t> b((size_t)ranks + 1);
        svr_dist_get_bounds_(dist, b.data(), b.size());
        if (rank == 0) {
          printf("frame %d: rows", f);
          for (uint32_t v : b) printf(" %u", v);
          printf("%s\n", changed ? " (re-cut)" : "");
        }
      }
      if (in_flight == 2) {  // both slots busy: take the older frame first (this is where present was)
        if (svr_dist_wait_frame_(dist, nullptr) != SVR_OK) return 1;
        in_flight--;
      }
      uint32_t y0 = 0, rows = 0;
      svr_dist_band_(dist, &y0, &rows);  // (interleaved: every rank draws; one without a tile row of its own draws nothing)
      if (svr_dist_begin_frame_(dist) != SVR_OK) {
        fprintf(stderr, "rank %d: begin_frame: %s\n", rank, svr_dist_last_error_());
        return 1;
      }
      eng.update_scene();
      int part_now = 0;
      svr_dist_get_partition_(dist, &part_now);
      if ((rows || part_now == SVR_DIST_INTERLEAVED) && (!eng.draw_background() || !eng.draw_geometry())) {
        fprintf(stderr, "rank %d: draw failed: %s\n", rank, eng.error.c_str());
        return 1;
      }
      if (svr_dist_end_frame_(dist) != SVR_OK) {
        fprintf(stderr, "rank %d: end_frame: %s\n", rank, svr_dist_last_error_());
        return 1;
      }
      in_flight++;
    }
    while (in_flight > 1) {
      if (svr_dist_wait_frame_(dist, nullptr) != SVR_OK) return 1;
      in_flight--;
    }
    if (svr_dist_read_frame_(dist, image.data(), image.size()) != SVR_OK) {
      fprintf(stderr, "rank %d: read_frame: %s\n", rank, svr_dist_last_error_());
      return 1;
    }
    if (!prefix.empty()) dump(prefix + ".rank" + std::to_string(rank) + ".swapchain", image.data(), image.size());
    uint32_t again = 0;
    svr_dist_replays_(dist, &again);
    SvrStats st{};
    eng.api.svr_get_stats(eng.ctx, &st);
    printf("rank %d of %d (%s): %d frames, last one exchanged; %u passes replayed, %u frames exchanged again\n", rank, ranks,
           transport.c_str(), frames, st.replayed_passes, again);
    svr_dist_destroy_(dist);
    eng.cleanup();
    g_exit_ok = true;
    return 0;
  }
  for (int f = 0; f < frames; f++) {
    eng.update_scene();
    if (f == frames - 1 && !prefix.empty()) {
      dump(prefix + ".scene", &eng.scene_data, 1);
      dump(prefix + ".opaque", eng.main_draw_context.opaque_surfaces.data(), eng.main_draw_context.opaque_surfaces.size());
      dump(prefix + ".transparent", eng.main_draw_context.transparent_surfaces.data(), eng.main_draw_context.transparent_surfaces.size());
    }
    if (!eng.draw_background() || !eng.draw_geometry()) {
      fprintf(stderr, "draw failed: %s\n", eng.error.c_str());
      return 1;
    }
  }
  eng.api.svr_sync(eng.ctx);
  printf("draws %d triangles %d update %.3f ms record %.3f ms\n", eng.stats.drawcall_count, eng.stats.triangle_count,
         eng.stats.scene_update_time, eng.stats.mesh_draw_time);
  if (!prefix.empty()) {
    std::vector<uint16_t> color;
    std::vector<float> depth;
    if (!eng.read_color_rgba16f(color) || !eng.read_depth(depth)) {
      fprintf(stderr, "readback failed: %s\n", eng.error.c_str());
      return 1;
    }
    dump(prefix + ".color", color.data(), color.size());
    dump(prefix + ".depth", depth.data(), depth.size());
    std::vector<uint8_t> swap;  // what copy_image leaves in the swapchain image (src/vk_engine.cpp:1277)
    if (!eng.read_swapchain(swap)) {
      fprintf(stderr, "swapchain readback failed: %s\n", eng.error.c_str());
      return 1;
    }
    dump(prefix + ".swapchain", swap.data(), swap.size());
  }
  eng.cleanup();
  return 0;
}
