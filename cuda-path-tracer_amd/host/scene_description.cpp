// scene_description.cpp -- see scene_description.hpp.  Follows the reference's semantics: materials indexed in
// name-sorted order; only the first mesh (by name) is uploaded and every mesh object refers to it
// (scene_description.cpp:42,59-66,95); transform arrays applied left to right as elem * mat
// (json_parser.cpp:85-88); the camera "transform" is decomposed into position + orientation
// (json_parser.cpp:190-203).
#include "scene_description.hpp"
#include "json.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace hip_pt {

// ---------------------------------------------------------------- matrices
Mat4 identity()
{
  Mat4 m{};
  m[0] = m[5] = m[10] = m[15] = 1.0f;
  return m;
}
Mat4 translate(float x, float y, float z)
{
  Mat4 m = identity();
  m[12] = x;
  m[13] = y;
  m[14] = z;
  return m;
}
Mat4 scale(float x, float y, float z)
{
  Mat4 m = identity();
  m[0] = x;
  m[5] = y;
  m[10] = z;
  return m;
}
Mat4 rotate(float a, float ax, float ay, float az)
{
  const float c = std::cos(a), s = std::sin(a);
  const float len = std::sqrt(ax * ax + ay * ay + az * az);
  const float axis[3] = {ax / len, ay / len, az / len};
  const float temp[3] = {(1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2]};
  Mat4 r = identity();
  r[0] = c + temp[0] * axis[0];
  r[1] = temp[0] * axis[1] + s * axis[2];
  r[2] = temp[0] * axis[2] - s * axis[1];
  r[4] = temp[1] * axis[0] - s * axis[2];
  r[5] = c + temp[1] * axis[1];
  r[6] = temp[1] * axis[2] + s * axis[0];
  r[8] = temp[2] * axis[0] + s * axis[1];
  r[9] = temp[2] * axis[1] - s * axis[0];
  r[10] = c + temp[2] * axis[2];
  return r;
}
static void normalize3(float v[3])
{
  const float inv = 1.0f / std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  v[0] *= inv;
  v[1] *= inv;
  v[2] *= inv;
}
static void cross3(const float a[3], const float b[3], float out[3])
{
  out[0] = a[1] * b[2] - b[1] * a[2];
  out[1] = a[2] * b[0] - b[2] * a[0];
  out[2] = a[0] * b[1] - b[0] * a[1];
}
Mat4 look_at(const float from[3], const float at[3], const float up[3])
{
  float dir[3] = {from[0] - at[0], from[1] - at[1], from[2] - at[2]};
  normalize3(dir);
  float left[3], new_up[3];
  cross3(up, dir, left);
  normalize3(left);
  cross3(dir, left, new_up);
  normalize3(new_up);
  Mat4 m = identity();
  for (int i = 0; i < 3; ++i) {
    m[i] = left[i];
    m[4 + i] = new_up[i];
    m[8 + i] = dir[i];
    m[12 + i] = from[i];
  }
  return m;
}
Mat4 multiply(const Mat4& a, const Mat4& b)
{
  Mat4 r{};
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
      r[4 * j + i] = ((a[i] * b[4 * j] + a[4 + i] * b[4 * j + 1]) + a[8 + i] * b[4 * j + 2]) + a[12 + i] * b[4 * j + 3];
  return r;
}

// rotation part of glm::decompose for a matrix without scale / skew: (w, x, y, z)
static void quat_from_matrix(const Mat4& m, float q_wxyz[4])
{
  const double row[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
  double q[4] = {0, 0, 0, 0};  // x y z w
  const double trace = row[0][0] + row[1][1] + row[2][2];
  if (trace > 0) {
    double root = std::sqrt(trace + 1.0);
    q[3] = 0.5 * root;
    root = 0.5 / root;
    q[0] = root * (row[1][2] - row[2][1]);
    q[1] = root * (row[2][0] - row[0][2]);
    q[2] = root * (row[0][1] - row[1][0]);
  } else {
    const int next[3] = {1, 2, 0};
    int i = 0;
    if (row[1][1] > row[0][0]) i = 1;
    if (row[2][2] > row[i][i]) i = 2;
    const int j = next[i], k = next[j];
    double root = std::sqrt(row[i][i] - row[j][j] - row[k][k] + 1.0);
    q[i] = 0.5 * root;
    root = 0.5 / root;
    q[j] = root * (row[i][j] + row[j][i]);
    q[k] = root * (row[i][k] + row[k][i]);
    q[3] = root * (row[j][k] - row[k][j]);
  }
  q_wxyz[0] = (float)q[3];
  q_wxyz[1] = (float)q[0];
  q_wxyz[2] = (float)q[1];
  q_wxyz[3] = (float)q[2];
}

// ---------------------------------------------------------------- SceneDescription
void SceneDescription::add_material(const std::string& name, Material material) { material_map_.try_emplace(name, material); }

void SceneDescription::add_object(Sphere sphere, const Mat4& transform, const std::string& material_name)
{
  if (material_map_.find(material_name) == material_map_.end()) throw std::runtime_error("Cannot find material " + material_name);
  objects_.push_back(Object{false, sphere, transform});
  objects_material_mapping_.push_back(material_name);
}
void SceneDescription::add_object(const Mesh&, const Mat4& transform, const std::string& material_name)
{
  if (material_map_.find(material_name) == material_map_.end()) throw std::runtime_error("Cannot find material " + material_name);
  objects_.push_back(Object{true, Sphere{}, transform});
  objects_material_mapping_.push_back(material_name);
}
const Mesh* SceneDescription::get_mesh(const std::string& name) const
{
  const auto it = mesh_map_.find(name);
  return it == mesh_map_.end() ? nullptr : &it->second;
}
const Mesh& SceneDescription::add_mesh(std::string name, Mesh&& mesh)
{
  auto [it, inserted] = mesh_map_.try_emplace(std::move(name), std::move(mesh));
  if (!inserted) throw std::runtime_error("Cannot add the same mesh twice!");
  return it->second;
}

FlatScene SceneDescription::build_scene() const
{
  FlatScene flat;
  const Mesh* mesh = mesh_map_.empty() ? nullptr : &mesh_map_.begin()->second;
  for (const Object& obj : objects_) {
    ptc_object out{};
    int rc;
    if (obj.is_mesh) {
      const float box[6] = {mesh->aabb_min[0], mesh->aabb_min[1], mesh->aabb_min[2],
                            mesh->aabb_max[0], mesh->aabb_max[1], mesh->aabb_max[2]};
      rc = ptc_make_object(1, 0, obj.transform.data(), nullptr, box, &out);
    } else {
      const ptc_sphere sp{{obj.sphere.center[0], obj.sphere.center[1], obj.sphere.center[2]}, obj.sphere.radius};
      rc = ptc_make_object(0, (uint32_t)flat.spheres.size(), obj.transform.data(), &sp, nullptr, &out);
      flat.spheres.push_back(sp);
    }
    if (rc < 0) throw std::runtime_error("ptc_make_object failed");
    flat.objects.push_back(out);
  }
  std::map<std::string, uint32_t, std::less<>> index_of;
  for (const auto& [name, material] : material_map_) {
    ptc_material m{};
    if (const auto* d = std::get_if<DiffuseMateral>(&material)) {
      m.type = 0;
      std::memcpy(m.p, d->albedo, sizeof d->albedo);
    } else if (const auto* mt = std::get_if<MetalMaterial>(&material)) {
      m.type = 1;
      std::memcpy(m.p, mt->albedo, sizeof mt->albedo);
      m.p[3] = mt->fuzz;
    } else {
      m.type = 2;
      m.p[0] = std::get<DielectricMaterial>(material).refraction_index;
    }
    index_of.emplace(name, (uint32_t)flat.materials.size());
    flat.materials.push_back(m);
  }
  for (const std::string& name : objects_material_mapping_) flat.object_material_indices.push_back(index_of.at(name));
  if (mesh) {
    flat.positions = mesh->positions;
    flat.indices = mesh->indices;
  }
  return flat;
}

// ---------------------------------------------------------------- JSON (just enough for the scene grammar)
namespace {

void vec3_from(const Json& j, float out[3])
{
  if (j.kind != Json::Array || j.arr.size() != 3) throw std::runtime_error("Json Parser: vec3 need to be 3d");
  for (int i = 0; i < 3; ++i) out[i] = j.arr[(size_t)i]->f();
}

// one transform command, json_parser.cpp:40-75
Mat4 command_from(const Json& j)
{
  float v[3];
  if (const Json* t = j.find("translate")) {
    vec3_from(*t, v);
    return translate(v[0], v[1], v[2]);
  }
  if (const Json* s = j.find("scale")) {
    if (s->kind == Json::Number) return scale(s->f(), s->f(), s->f());
    vec3_from(*s, v);
    return scale(v[0], v[1], v[2]);
  }
  if (const Json* r = j.find("rotate")) {
    vec3_from(j.at("axis"), v);
    return rotate(r->f() * 0.01745329251994329576923690768489f, v[0], v[1], v[2]);  // glm::radians
  }
  if (j.find("from") && j.find("at") && j.find("up")) {
    float from[3], at[3], up[3];
    vec3_from(j.at("from"), from);
    vec3_from(j.at("at"), at);
    vec3_from(j.at("up"), up);
    return look_at(from, at, up);
  }
  throw std::runtime_error("Json parser: Unrecognized transform command");
}

// json_parser.cpp:78-95
Mat4 transform_from(const Json& j)
{
  if (j.kind == Json::Object) return command_from(j);
  if (j.kind != Json::Array) throw std::runtime_error("Json Parser: Transform must be either an object or an array!");
  Mat4 mat = identity();
  for (const auto& elem : j.arr) mat = multiply(command_from(*elem), mat);
  return mat;
}

std::string dirname_of(const std::string& path)
{
  const size_t slash = path.find_last_of('/');
  return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

}  // namespace

SceneDescription scene_from_json(const std::string& filename)
{
  std::ifstream file(filename);
  if (!file.is_open()) throw std::runtime_error("Json Parser: Cannot open file " + filename);
  std::stringstream buffer;
  buffer << file.rdbuf();
  const JsonPtr root = JsonReader(buffer.str()).parse();
  const std::string dir = dirname_of(filename);

  SceneDescription scene;
  scene.filename = filename;
  for (const auto& m : root->at("materials").arr) {  // read_materials, json_parser.cpp:101-122
    const std::string name = m->at("name").str, type = m->at("type").str;
    if (type == "lambertian") {
      DiffuseMateral d{};
      vec3_from(m->at("albedo"), d.albedo);
      scene.add_material(name, d);
    } else if (type == "dielectric") {
      scene.add_material(name, DielectricMaterial{m->at("refraction_index").f()});
    } else if (type == "metal") {
      MetalMaterial mt{};
      vec3_from(m->at("albedo"), mt.albedo);
      mt.fuzz = m->at("fuzz").f();
      scene.add_material(name, mt);
    } else {
      throw std::runtime_error("Json Parser: Unsupported material type " + type);
    }
  }
  for (const auto& s : root->at("surfaces").arr) {  // read_surfaces, json_parser.cpp:133-159
    const std::string type = s->at("type").str, material = s->at("material").str;
    const Mat4 transform = transform_from(s->at("transform"));
    if (type == "sphere") {
      Sphere sp;
      sp.radius = s->at("radius").f();
      scene.add_object(sp, transform, material);
    } else if (type == "mesh") {
      const std::string path = dir + "/" + s->at("filename").str;
      const Mesh* mesh = scene.get_mesh(path);
      if (!mesh) mesh = &scene.add_mesh(path, load_obj(path));
      scene.add_object(*mesh, transform, material);
    } else {
      throw std::runtime_error("Json Parser: Not supported surface type " + type);
    }
  }
  const Json& camera = root->at("camera");
  if (const Json* t = camera.find("transform")) {
    const Mat4 m = transform_from(*t);
    scene.camera.position[0] = m[12];
    scene.camera.position[1] = m[13];
    scene.camera.position[2] = m[14];
    quat_from_matrix(m, scene.camera.rotation_wxyz);
  }
  scene.camera.vfov = camera.at("vfov").f() * 0.01745329251994329576923690768489f;
  if (const Json* r = camera.find("resolution")) {
    if (r->kind != Json::Array || r->arr.size() != 2) throw std::runtime_error("Json Parser: resolution need to be 2d");
    scene.resolution[0] = (int)r->arr[0]->num;
    scene.resolution[1] = (int)r->arr[1]->num;
  }
  if (const Json* sampler = root->find("sampler")) scene.spp = (int)sampler->at("samples").num;
  return scene;
}

// ---------------------------------------------------------------- OBJ (the file's first mesh, fan triangulation)
// The reference keeps assimp's mMeshes[0] only (model_loader.cpp:22).  assimp's OBJ importer opens a new mesh at every
// `o` / `g` statement that names another object or group and at every `usemtl` that names another material, and drops
// meshes without faces: mMeshes[0] is the first such chunk that holds a face.  Kept here: the faces of that chunk, and of
// the file's vertices the ones those faces use (assimp's mesh carries its own vertices, and GenBoundingBoxes bounds those),
// in file order.  (assimp does not join identical vertices under the reference's flags, so ITS vertex array has three
// entries per triangle; the positions a triangle's indices lead to are the same, and nothing on the path depends on more.)
Mesh load_obj(const std::string& filename)
{
  std::ifstream file(filename);
  if (!file.is_open()) throw std::runtime_error("Unable to load " + filename);
  Mesh mesh;
  std::vector<float> all;          // every `v` of the file
  std::vector<uint32_t> faces;     // the first mesh's triangles, indices into `all`
  std::string line, cur_o, cur_g, cur_m;
  bool closed = false;             // the first mesh is complete: a later chunk has begun
  auto name_of = [](const std::string& l, size_t from) {
    size_t a = l.find_first_not_of(" \t\r", from), b = l.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : l.substr(a, b - a + 1);
  };
  while (std::getline(file, line)) {
    if (line.size() < 2) continue;
    if (line[0] == 'v' && line[1] == ' ') {
      float x, y, z;
      if (std::sscanf(line.c_str() + 2, "%f %f %f", &x, &y, &z) == 3) {
        all.push_back(x);
        all.push_back(y);
        all.push_back(z);
      }
    } else if ((line[0] == 'o' || line[0] == 'g') && (line[1] == ' ' || line[1] == '\t')) {
      std::string& cur = line[0] == 'o' ? cur_o : cur_g;
      const std::string name = name_of(line, 1);
      if (name != cur && !faces.empty()) closed = true;
      cur = name;
    } else if (line.compare(0, 7, "usemtl ") == 0) {
      const std::string name = name_of(line, 6);
      if (name != cur_m && !faces.empty()) closed = true;
      cur_m = name;
    } else if (line[0] == 'f' && line[1] == ' ' && !closed) {
      std::vector<uint32_t> face;
      std::istringstream ss(line.substr(2));
      std::string tok;
      while (ss >> tok) {
        const long idx = std::strtol(tok.c_str(), nullptr, 10);  // "a", "a/b", "a/b/c", "a//c"
        const long count = (long)(all.size() / 3);
        face.push_back((uint32_t)(idx < 0 ? count + idx : idx - 1));
      }
      for (size_t k = 1; k + 1 < face.size(); ++k) {
        faces.push_back(face[0]);
        faces.push_back(face[k]);
        faces.push_back(face[k + 1]);
      }
    }
  }
  if (all.empty() || faces.empty()) throw std::runtime_error("Unable to load " + filename);
  // the vertices the mesh uses, in file order
  const size_t nv = all.size() / 3;
  std::vector<uint32_t> remap(nv, 0xffffffffu);
  for (uint32_t v : faces) {
    if (v >= nv) throw std::runtime_error("Unable to load " + filename + ": face index out of range");
    remap[v] = 0u;
  }
  uint32_t next = 0;
  for (size_t v = 0; v < nv; ++v)
    if (remap[v] == 0u) {
      remap[v] = next++;
      mesh.positions.insert(mesh.positions.end(), all.begin() + 3 * (long)v, all.begin() + 3 * (long)v + 3);
    }
  mesh.indices.reserve(faces.size());
  for (uint32_t v : faces) mesh.indices.push_back(remap[v]);
  for (int a = 0; a < 3; ++a) mesh.aabb_min[a] = mesh.aabb_max[a] = mesh.positions[(size_t)a];
  for (size_t v = 0; v < mesh.positions.size() / 3; ++v)
    for (int a = 0; a < 3; ++a) {
      mesh.aabb_min[a] = std::min(mesh.aabb_min[a], mesh.positions[3 * v + (size_t)a]);
      mesh.aabb_max[a] = std::max(mesh.aabb_max[a], mesh.positions[3 * v + (size_t)a]);
    }
  return mesh;
}

// ---------------------------------------------------------------- PNG (zlib "stored" blocks: no compression library needed)
namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char* data, size_t n)
{
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}
void put_be32(std::vector<unsigned char>& out, uint32_t v)
{
  for (int s = 24; s >= 0; s -= 8) out.push_back((unsigned char)(v >> s));
}
void chunk(std::vector<unsigned char>& png, const char type[4], const std::vector<unsigned char>& data)
{
  put_be32(png, (uint32_t)data.size());
  std::vector<unsigned char> body(type, type + 4);
  body.insert(body.end(), data.begin(), data.end());
  png.insert(png.end(), body.begin(), body.end());
  put_be32(png, crc32_update(0xffffffffu, body.data(), body.size()) ^ 0xffffffffu);
}
}  // namespace

bool write_png(const std::string& filename, int width, int height, const void* rgba)
{
  const auto* src = static_cast<const unsigned char*>(rgba);
  std::vector<unsigned char> raw;
  raw.reserve((size_t)height * ((size_t)width * 4 + 1));
  for (int y = 0; y < height; ++y) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), src + (size_t)y * width * 4, src + (size_t)(y + 1) * width * 4);
  }
  std::vector<unsigned char> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (unsigned char c : raw) {
    a = (a + c) % 65521u;
    b = (b + a) % 65521u;
  }
  for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
    const size_t n = std::min<size_t>(65535, raw.size() - off);
    z.push_back(off + n >= raw.size() ? 1 : 0);
    z.push_back((unsigned char)(n & 0xff));
    z.push_back((unsigned char)(n >> 8));
    z.push_back((unsigned char)(~n & 0xff));
    z.push_back((unsigned char)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
    if (raw.empty()) break;
  }
  put_be32(z, (b << 16) | a);
  std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<unsigned char> ihdr;
  put_be32(ihdr, (uint32_t)width);
  put_be32(ihdr, (uint32_t)height);
  const unsigned char tail[5] = {8, 6, 0, 0, 0};  // 8 bit, RGBA
  ihdr.insert(ihdr.end(), tail, tail + 5);
  chunk(png, "IHDR", ihdr);
  chunk(png, "IDAT", z);
  chunk(png, "IEND", {});
  std::ofstream out(filename, std::ios::binary);
  if (!out) return false;
  out.write(reinterpret_cast<const char*>(png.data()), (std::streamsize)png.size());
  return (bool)out;
}

}  // namespace hip_pt
