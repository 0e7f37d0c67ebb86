// svr_gltf.cpp — load_gltf_meshes: a .glb / .gltf file -> LoadedScene uploaded through the engine.
//
// Follows the reference loader step for step (src/vk_loader.cpp:162-437) with its own container/JSON/
// accessor reader in place of fastgltf and svr_image.h (svr_png.h, svr_jpeg.h and the minor formats) in place of stb_image:
//   samplers   :197-211  mag/min filter default NEAREST when absent, mipmap mode from the min filter
//                        (default LINEAR), minLod 0, maxLod = VK_LOD_CLAMP_NONE, address modes REPEAT
//   images     :218-231  decode to RGBA8, create_image(..., mipmapped); failure -> error checkerboard
//   materials  :244-289  baseColorFactor, metallic/roughness factors, alphaMode BLEND -> Transparent,
//                        baseColorTexture -> (image, sampler) else white + default linear sampler
//   meshes     :294-380  one vertex + one index buffer per mesh, indices rebased, defaults normal (1,0,0)
//                        colour 1 uv 0, missing material -> materials[0], bounds over ALL vertices so far
//   nodes      :383-434  matrix or T*R*S, hierarchy, top nodes refresh_transform(identity)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "svr_engine.h"
#include "svr_json.h"
#include "svr_image.h"

namespace svrhost {

namespace {

using svrjson::Value;

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  std::streamoff n = f.tellg();
  f.seekg(0);
  out.resize((size_t)n);
  f.read(reinterpret_cast<char*>(out.data()), n);
  return (bool)f;
}

bool base64(const std::string& s, size_t from, std::vector<uint8_t>& out) {
  auto val = [](char c) -> int {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+' || c == '-') return 62;
    if (c == '/' || c == '_') return 63;
    return -1;
  };
  uint32_t acc = 0;
  int nb = 0;
  for (size_t i = from; i < s.size(); i++) {
    if (s[i] == '=') break;
    int v = val(s[i]);
    if (v < 0) return false;
    acc = (acc << 6) | (uint32_t)v;
    nb += 6;
    if (nb >= 8) {
      nb -= 8;
      out.push_back((uint8_t)(acc >> nb));
    }
  }
  return true;
}

std::string percent_decode(const std::string& s) {
  std::string o;
  for (size_t i = 0; i < s.size(); i++) {
    if (s[i] == '%' && i + 2 < s.size() + 0 && std::isxdigit((unsigned char)s[i + 1]) && std::isxdigit((unsigned char)s[i + 2])) {
      o += (char)std::strtol(s.substr(i + 1, 2).c_str(), nullptr, 16);
      i += 2;
    } else {
      o += s[i];
    }
  }
  return o;
}

struct Asset {
  Value json;
  std::string dir;
  std::vector<std::vector<uint8_t>> buffers;
  std::string error;
};

// a data: URI or a path relative to the file
bool load_uri(const Asset& a, const std::string& uri, std::vector<uint8_t>& out) {
  if (uri.compare(0, 5, "data:") == 0) {
    size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.find(";base64") == std::string::npos) return false;
    return base64(uri, comma + 1, out);
  }
  return read_file(a.dir + percent_decode(uri), out);
}

struct View {
  const uint8_t* data = nullptr;
  size_t length = 0, stride = 0;
};
bool buffer_view(const Asset& a, long long index, View& v) {
  const Value& bv = a.json["bufferViews"][(size_t)index];
  if (!bv.is_object()) return false;
  long long buf = bv["buffer"].int_or(-1);
  if (buf < 0 || (size_t)buf >= a.buffers.size()) return false;
  size_t off = (size_t)bv["byteOffset"].int_or(0), len = (size_t)bv["byteLength"].int_or(0);
  if (off + len > a.buffers[(size_t)buf].size()) return false;
  v.data = a.buffers[(size_t)buf].data() + off;
  v.length = len;
  v.stride = (size_t)bv["byteStride"].int_or(0);
  return true;
}

int component_size(long long ct) {
  switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: return 0;
  }
}
int type_components(const std::string& t) {
  if (t == "SCALAR") return 1;
  if (t == "VEC2") return 2;
  if (t == "VEC3") return 3;
  if (t == "VEC4") return 4;
  return 0;
}

// Element i, component c of an accessor as float: integers convert like fastgltf's iterateAccessor
// does (normalized ones by the KHR_mesh_quantization rules, plain ones by value).
struct AccessorReader {
  View view;
  size_t offset = 0, count = 0, stride = 0;
  long long ctype = 0;
  int ncomp = 0, csize = 0;
  bool normalized = false;
  bool init(const Asset& a, long long index, std::string* err) {
    const Value& acc = a.json["accessors"][(size_t)index];
    if (!acc.is_object()) return fail(err, "accessor index out of range");
    ctype = acc["componentType"].int_or(0);
    csize = component_size(ctype);
    ncomp = type_components(acc["type"].string_or(""));
    const long long count_ll = acc["count"].int_or(0);
    if (count_ll < 0 || count_ll > (1ll << 31)) return fail(err, "accessor count out of range");  // (a hostile file: negative -> huge)
    count = (size_t)count_ll;
    normalized = acc["normalized"].kind == Value::Bool && acc["normalized"].b;
    if (!csize || !ncomp) return fail(err, "unsupported accessor type");
    const size_t elem = (size_t)(csize * ncomp);
    const bool sparse = acc.has("sparse");
    if (!acc.has("bufferView") && !sparse) return fail(err, "accessor without a bufferView");
    if (acc.has("bufferView")) {
      if (!buffer_view(a, acc["bufferView"].int_or(-1), view)) return fail(err, "bad bufferView");
      const long long off_ll = acc["byteOffset"].int_or(0);
      if (off_ll < 0) return fail(err, "negative accessor byteOffset");
      offset = (size_t)off_ll;
      stride = view.stride ? view.stride : elem;
      // by division: offset + (count - 1) * stride + elem must not wrap on its way past the length check
      if (count && (offset > view.length || elem > view.length - offset || (count - 1) > (view.length - offset - elem) / stride))
        return fail(err, "accessor runs past its bufferView");
    }
    if (!sparse) return true;
    // Sparse accessor (glTF 2.0 section 3.6.2.3; fastgltf's iterateAccessor, src/vk_loader.cpp:311-357, reads them):
    // the base elements — zeros without a bufferView — with `count` of them replaced; made dense here once.
    if (count > (size_t(1) << 28) / elem) return fail(err, "sparse accessor too large");  // 256 MiB of elements at most
    try {
      dense.assign(count * elem, 0);
    } catch (const std::exception&) {
      return fail(err, "out of memory for a sparse accessor");
    }
    if (acc.has("bufferView"))
      for (size_t i = 0; i < count; i++) std::memcpy(dense.data() + i * elem, view.data + offset + i * stride, elem);
    const Value& sp = acc["sparse"];
    const long long n_ll = sp["count"].int_or(0), ioff_ll = sp["indices"]["byteOffset"].int_or(0), voff_ll = sp["values"]["byteOffset"].int_or(0);
    if (n_ll < 0 || (size_t)n_ll > count || ioff_ll < 0 || voff_ll < 0) return fail(err, "sparse count / offsets out of range");
    const size_t n = (size_t)n_ll;
    const Value& si = sp["indices"];
    const Value& sv = sp["values"];
    View iv, vv;
    if (!buffer_view(a, si["bufferView"].int_or(-1), iv) || !buffer_view(a, sv["bufferView"].int_or(-1), vv)) return fail(err, "bad sparse bufferView");
    const long long ict = si["componentType"].int_or(0);
    const size_t isz = (size_t)component_size(ict), ioff = (size_t)si["byteOffset"].int_or(0), voff = (size_t)sv["byteOffset"].int_or(0);
    if (ict != 5121 && ict != 5123 && ict != 5125) return fail(err, "bad sparse index type");
    if (ioff > iv.length || n > (iv.length - ioff) / isz || voff > vv.length || n > (vv.length - voff) / elem)
      return fail(err, "sparse data runs past its bufferView");
    for (size_t k = 0; k < n; k++) {
      uint32_t idx = 0;
      std::memcpy(&idx, iv.data + ioff + k * isz, isz);  // little-endian u8 / u16 / u32
      if (idx >= count) return fail(err, "sparse index out of range");
      std::memcpy(dense.data() + (size_t)idx * elem, vv.data + voff + k * elem, elem);
    }
    view.data = dense.data();
    view.length = dense.size();
    view.stride = 0;
    offset = 0;
    stride = elem;
    return true;
  }
  std::vector<uint8_t> dense;  // a sparse accessor's elements, materialised
  static bool fail(std::string* err, const char* m) {
    if (err) *err = m;
    return false;
  }
  const uint8_t* at(size_t i, int c) const { return view.data + offset + i * stride + (size_t)(c * csize); }
  uint32_t as_index(size_t i) const {
    const uint8_t* p = at(i, 0);
    switch (ctype) {
      case 5121: return p[0];
      case 5123: { uint16_t v; std::memcpy(&v, p, 2); return v; }
      case 5125: { uint32_t v; std::memcpy(&v, p, 4); return v; }
      case 5120: return (uint32_t)(int8_t)p[0];
      case 5122: { int16_t v; std::memcpy(&v, p, 2); return (uint32_t)v; }
      default: { float v; std::memcpy(&v, p, 4); return (uint32_t)v; }
    }
  }
  float as_float(size_t i, int c) const {
    if (c >= ncomp) return 0.0f;
    const uint8_t* p = at(i, c);
    switch (ctype) {
      case 5126: { float v; std::memcpy(&v, p, 4); return v; }
      case 5121: return normalized ? (float)p[0] / 255.0f : (float)p[0];
      case 5123: { uint16_t v; std::memcpy(&v, p, 2); return normalized ? (float)v / 65535.0f : (float)v; }
      case 5120: { int8_t v = (int8_t)p[0]; return normalized ? std::fmax((float)v / 127.0f, -1.0f) : (float)v; }
      case 5122: { int16_t v; std::memcpy(&v, p, 2); return normalized ? std::fmax((float)v / 32767.0f, -1.0f) : (float)v; }
      default: { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    }
  }
};

// glTF sampler filter codes (fastgltf::Filter has the same values)
enum { F_NEAREST = 9728, F_LINEAR = 9729, F_NMN = 9984, F_LMN = 9985, F_NML = 9986, F_LML = 9987 };
int extract_filter(long long f) {  // src/vk_loader.cpp:26-41
  return (f == F_NEAREST || f == F_NMN || f == F_NML) ? SVR_FILTER_NEAREST : SVR_FILTER_LINEAR;
}
int extract_mipmap_mode(long long f) {  // src/vk_loader.cpp:43-54
  return (f == F_NMN || f == F_LMN) ? SVR_MIPMAP_NEAREST : SVR_MIPMAP_LINEAR;
}

bool parse_container(const std::string& path, Asset& a) {
  std::vector<uint8_t> file;
  if (!read_file(path, file)) {
    a.error = "cannot read " + path;
    return false;
  }
  size_t slash = path.find_last_of('/');
  a.dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
  std::string json_text;
  std::vector<uint8_t> glb_bin;
  bool have_bin = false;
  if (file.size() >= 12 && !std::memcmp(file.data(), "glTF", 4)) {  // binary container: 12-byte header + chunks
    uint32_t total;
    std::memcpy(&total, file.data() + 8, 4);
    size_t pos = 12, end = std::min<size_t>(total, file.size());
    while (pos + 8 <= end) {
      uint32_t len, type;
      std::memcpy(&len, file.data() + pos, 4);
      std::memcpy(&type, file.data() + pos + 4, 4);
      pos += 8;
      if (pos + len > end) {
        a.error = "GLB chunk runs past the file";
        return false;
      }
      if (type == 0x4E4F534Au) json_text.assign(reinterpret_cast<const char*>(file.data() + pos), len);  // "JSON"
      else if (type == 0x004E4942u && !have_bin) {                                                        // "BIN\0"
        glb_bin.assign(file.begin() + (long)pos, file.begin() + (long)(pos + len));
        have_bin = true;
      }
      pos += (len + 3u) & ~3u;
    }
  } else {
    json_text.assign(reinterpret_cast<const char*>(file.data()), file.size());
  }
  std::string jerr;
  if (!svrjson::parse(json_text, a.json, &jerr) || !a.json.is_object()) {
    a.error = "glTF JSON: " + jerr;
    return false;
  }
  // LoadGLBBuffers | LoadExternalBuffers: every buffer ends up in memory
  const Value& bufs = a.json["buffers"];
  for (size_t i = 0; i < bufs.size(); i++) {
    std::vector<uint8_t> data;
    if (bufs[i].has("uri")) {
      if (!load_uri(a, bufs[i]["uri"].str, data)) {
        a.error = "cannot load buffer " + std::to_string(i);
        return false;
      }
    } else if (i == 0 && have_bin) {
      data.swap(glb_bin);
    } else {
      a.error = "buffer " + std::to_string(i) + " has no data";
      return false;
    }
    a.buffers.push_back(std::move(data));
  }
  return true;
}

// load_image, src/vk_loader.cpp:81-160
SvrImage load_image(SvrEngine* engine, const Asset& a, const Value& image) {
  std::vector<uint8_t> bytes;
  const uint8_t* p = nullptr;
  size_t n = 0;
  if (image.has("uri")) {
    if (!load_uri(a, image["uri"].str, bytes)) return 0;
    p = bytes.data();
    n = bytes.size();
  } else if (image.has("bufferView")) {
    View v;
    if (!buffer_view(a, image["bufferView"].int_or(-1), v)) return 0;
    p = v.data;
    n = v.length;
  } else {
    return 0;
  }
  std::string err;
  svrimg::Image img;  // PNG, BMP, GIF, PSD, PIC, JPEG, PNM, HDR, TGA: whatever stbi_load_from_memory takes
  if (!svrimg::decode(p, n, img, &err)) return 0;
  return engine->create_image(img.rgba.data(), img.w, img.h, true);  // MIPMAP_ENABLED, :24
}

}  // namespace

std::shared_ptr<LoadedScene> load_gltf_meshes(SvrEngine* engine, const std::string& file_path) {
  Asset a;
  if (!parse_container(file_path, a)) {
    engine->error = "load_gltf_meshes: " + a.error;
    return nullptr;
  }
  const Value& gltf = a.json;
  auto scene = std::make_shared<LoadedScene>();
  auto fail = [&](const std::string& m) {
    engine->error = "load_gltf_meshes: " + m;
    return std::shared_ptr<LoadedScene>();
  };

  // The reference's parser is built for three extensions (src/vk_loader.cpp:169-173); fastgltf refuses a file that
  // REQUIRES any other one (Draco, meshopt, basisu ...: data this loader could not read either).
  for (size_t i = 0; i < gltf["extensionsRequired"].size(); i++) {
    const std::string& e = gltf["extensionsRequired"][i].str;
    if (e != "KHR_mesh_quantization" && e != "KHR_texture_transform" && e != "KHR_materials_variants")
      return fail("required extension " + e + " is not supported");
  }

  // samplers
  for (size_t i = 0; i < gltf["samplers"].size(); i++) {
    const Value& s = gltf["samplers"][i];
    long long mag = s["magFilter"].int_or(F_NEAREST), minf = s["minFilter"].int_or(F_NEAREST);
    SvrSamplerDesc d{};
    d.mag_filter = extract_filter(mag);
    d.min_filter = extract_filter(minf);
    d.mipmap_mode = extract_mipmap_mode(minf);
    d.min_lod = 0.0f;
    d.max_lod = 1000.0f;  // VK_LOD_CLAMP_NONE
    SvrSampler h = 0;
    if (engine->api.svr_create_sampler(engine->ctx, &d, &h)) return fail(engine->api.svr_last_error());
    scene->samplers.push_back(h);
  }

  // images (failure -> the error checkerboard, :226-231)
  std::vector<SvrImage> images;
  for (size_t i = 0; i < gltf["images"].size(); i++) {
    SvrImage h = load_image(engine, a, gltf["images"][i]);
    if (h) {
      scene->images.push_back(h);
    } else {
      h = engine->error_checkerboard_image;
      std::printf("gltf failed to load texture %s\n", gltf["images"][i]["name"].string_or("").c_str());
    }
    images.push_back(h);
  }

  // materials
  for (size_t i = 0; i < gltf["materials"].size(); i++) {
    const Value& m = gltf["materials"][i];
    const Value& pbr = m["pbrMetallicRoughness"];
    float cf[4] = {1, 1, 1, 1};
    for (int k = 0; k < 4; k++)
      if (pbr["baseColorFactor"].is_array()) cf[k] = (float)pbr["baseColorFactor"][(size_t)k].number_or(1.0);
    int pass = m["alphaMode"].string_or("OPAQUE") == "BLEND" ? SVR_PASS_TRANSPARENT : SVR_PASS_MAIN_COLOR;
    SvrImage img = engine->white_image;
    SvrSampler smp = engine->default_sampler_linear;
    if (pbr.has("baseColorTexture")) {
      const Value& tex = gltf["textures"][(size_t)pbr["baseColorTexture"]["index"].int_or(-1)];
      long long src = tex["source"].int_or(-1), si = tex["sampler"].int_or(-1);
      // the reference dereferences both optionals unchecked (:274-275); a texture without them is rejected here
      if (src < 0 || (size_t)src >= images.size() || si < 0 || (size_t)si >= scene->samplers.size())
        return fail("material " + std::to_string(i) + ": baseColorTexture needs both a source image and a sampler");
      img = images[(size_t)src];
      smp = scene->samplers[(size_t)si];
    }
    auto mat = engine->write_material(pass, cf, img, smp);
    if (!mat) return fail(engine->error);
    scene->materials.push_back(mat);
  }

  // meshes
  std::vector<uint32_t> indices;
  std::vector<SvrVertex> vertices;
  for (size_t mi = 0; mi < gltf["meshes"].size(); mi++) {
    const Value& mesh = gltf["meshes"][mi];
    auto newmesh = std::make_shared<MeshAsset>();
    newmesh->name = mesh["name"].string_or("");
    indices.clear();
    vertices.clear();
    for (size_t pi = 0; pi < mesh["primitives"].size(); pi++) {
      const Value& p = mesh["primitives"][pi];
      const Value& attrs = p["attributes"];
      std::string err;
      if (!attrs.has("POSITION")) return fail("primitive without POSITION");
      AccessorReader pos;
      if (!pos.init(a, attrs["POSITION"].int_or(-1), &err)) return fail("POSITION: " + err);
      GeoSurface surf;
      surf.startIndex = (uint32_t)indices.size();
      const size_t initial_vtx = vertices.size();
      if (p.has("indices")) {
        AccessorReader idx;
        if (!idx.init(a, p["indices"].int_or(-1), &err)) return fail("indices: " + err);
        surf.count = (uint32_t)idx.count;
        for (size_t k = 0; k < idx.count; k++) indices.push_back(idx.as_index(k) + (uint32_t)initial_vtx);
      } else {  // Options::GenerateMeshIndices (:178): 0..n-1
        surf.count = (uint32_t)pos.count;
        for (size_t k = 0; k < pos.count; k++) indices.push_back((uint32_t)(k + initial_vtx));
      }
      vertices.resize(initial_vtx + pos.count);
      for (size_t k = 0; k < pos.count; k++) {
        SvrVertex v{};
        for (int c = 0; c < 3; c++) v.position[c] = pos.as_float(k, c);
        v.normal[0] = 1.0f;
        v.color[0] = v.color[1] = v.color[2] = v.color[3] = 1.0f;
        vertices[initial_vtx + k] = v;
      }
      auto optional_attr = [&](const char* name, AccessorReader& r) -> int {  // 0 absent, 1 ok, -1 error
        if (!attrs.has(name)) return 0;
        if (!r.init(a, attrs[name].int_or(-1), &err) || r.count > pos.count) return -1;
        return 1;
      };
      AccessorReader nr, uv, col;
      int has = optional_attr("NORMAL", nr);
      if (has < 0) return fail("NORMAL: " + err);
      if (has)
        for (size_t k = 0; k < nr.count; k++)
          for (int c = 0; c < 3; c++) vertices[initial_vtx + k].normal[c] = nr.as_float(k, c);
      has = optional_attr("TEXCOORD_0", uv);
      if (has < 0) return fail("TEXCOORD_0: " + err);
      if (has)
        for (size_t k = 0; k < uv.count; k++) {
          vertices[initial_vtx + k].uv_x = uv.as_float(k, 0);
          vertices[initial_vtx + k].uv_y = uv.as_float(k, 1);
        }
      has = optional_attr("COLOR_0", col);
      if (has < 0) return fail("COLOR_0: " + err);
      if (has)
        for (size_t k = 0; k < col.count; k++)
          for (int c = 0; c < 4; c++) vertices[initial_vtx + k].color[c] = (c < col.ncomp) ? col.as_float(k, c) : 1.0f;
      long long mat = p["material"].int_or(0);  // missing -> materials[0] (:361-365)
      if (scene->materials.empty() || mat < 0 || (size_t)mat >= scene->materials.size()) return fail("primitive refers to a material that does not exist");
      surf.material = scene->materials[(size_t)mat];
      if (pos.count == 0) return fail("primitive without vertices");
      surf.bounds = loader_bounds(vertices, initial_vtx);
      newmesh->surfaces.push_back(surf);
    }
    newmesh->meshBuffers = engine->upload_mesh(indices, vertices);
    if (!newmesh->meshBuffers) return fail(engine->error);
    scene->meshes.push_back(newmesh);
  }

  // nodes
  for (size_t i = 0; i < gltf["nodes"].size(); i++) {
    const Value& n = gltf["nodes"][i];
    std::shared_ptr<Node> node;
    if (n.has("mesh")) {
      long long m = n["mesh"].int_or(-1);
      if (m < 0 || (size_t)m >= scene->meshes.size()) return fail("node refers to a mesh that does not exist");
      auto mn = std::make_shared<MeshNode>();
      mn->mesh = scene->meshes[(size_t)m];
      node = mn;
    } else {
      node = std::make_shared<Node>();
    }
    if (n["matrix"].is_array() && n["matrix"].size() == 16) {
      for (int k = 0; k < 16; k++) node->local_transform.data()[k] = (float)n["matrix"][(size_t)k].number_or(0.0);  // glm::make_mat4x4: column-major
    } else {
      float t[3] = {0, 0, 0}, r[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
      for (int k = 0; k < 3; k++) {
        if (n["translation"].is_array()) t[k] = (float)n["translation"][(size_t)k].number_or(0.0);
        if (n["scale"].is_array()) s[k] = (float)n["scale"][(size_t)k].number_or(1.0);
      }
      for (int k = 0; k < 4; k++)
        if (n["rotation"].is_array()) r[k] = (float)n["rotation"][(size_t)k].number_or(k == 3 ? 1.0 : 0.0);
      // tm * rm * sm with glm::translate / toMat4(quat(w,x,y,z)) / glm::scale (:405-415)
      node->local_transform = svrm::mul(svrm::mul(svrm::translate(svrm::identity(), vec3{t[0], t[1], t[2]}), svrm::to_mat4(svrm::quat{r[3], r[0], r[1], r[2]})),
                                        svrm::scale(svrm::identity(), vec3{s[0], s[1], s[2]}));
    }
    scene->nodes.push_back(node);
  }
  for (size_t i = 0; i < gltf["nodes"].size(); i++) {
    const Value& ch = gltf["nodes"][i]["children"];
    for (size_t k = 0; k < ch.size(); k++) {
      long long c = ch[k].int_or(-1);
      if (c < 0 || (size_t)c >= scene->nodes.size()) return fail("node child index out of range");
      scene->nodes[i]->children.push_back(scene->nodes[(size_t)c]);
      scene->nodes[(size_t)c]->parent = scene->nodes[i];
    }
  }
  for (auto& node : scene->nodes)
    if (node->parent.lock() == nullptr) {
      scene->top_nodes.push_back(node);
      node->refresh_transform(svrm::identity());
    }
  return scene;
}

}  // namespace svrhost
