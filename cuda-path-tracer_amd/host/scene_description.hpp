// scene_description.hpp -- C++ host mirror of the reference's SceneDescription and asset front-end
// (/root/reference/src/lib/scene_description.{hpp,cpp}, camera.hpp, material.hpp, mesh.hpp, sphere.hpp,
// assets/json_parser.cpp, assets/model_loader.cpp), reduced to what feeds the render core.  No glm, assimp or
// nlohmann: the few matrix functions, a small JSON reader and an OBJ reader are written out here.
#pragma once

#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <variant>
#include <vector>

#include "../../include/ptcore.h"

namespace hip_pt {

using Mat4 = std::array<float, 16>;  // column-major like glm: m[4*col + row]

Mat4 identity();
Mat4 translate(float x, float y, float z);                         // glm::translate(vec3)
Mat4 scale(float x, float y, float z);                             // glm::scale(vec3)
Mat4 rotate(float angle_rad, float ax, float ay, float az);        // glm::rotate(angle, axis)
Mat4 look_at(const float from[3], const float at[3], const float up[3]);  // json_parser.cpp:57-70
Mat4 multiply(const Mat4& a, const Mat4& b);                       // glm mat4 * mat4

struct DiffuseMateral { float albedo[3]; };                 // (sic) material.hpp:6-8
struct MetalMaterial { float albedo[3]; float fuzz; };      // material.hpp:10-13
struct DielectricMaterial { float refraction_index; };      // material.hpp:15-17
using Material = std::variant<DiffuseMateral, MetalMaterial, DielectricMaterial>;

struct Sphere { float center[3] = {0, 0, 0}; float radius = 0; };  // sphere.hpp:8-11

struct Mesh {  // mesh.hpp:9-18
  std::vector<float> positions;    // xyz per vertex
  std::vector<uint32_t> indices;
  float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};
  [[nodiscard]] size_t triangle_count() const { return indices.size() / 3; }
};

struct Camera {  // camera.hpp:17-23
  float position[3] = {0, 0, 0};
  float rotation_wxyz[4] = {1, 0, 0, 0};
  float vfov = 1.57079632679f;
  [[nodiscard]] ptc_camera to_c() const
  {
    return ptc_camera{{position[0], position[1], position[2]},
                      {rotation_wxyz[0], rotation_wxyz[1], rotation_wxyz[2], rotation_wxyz[3]}, vfov};
  }
};

// what build_scene() hands to ptc_upload_scene: the reference's six cudaMemcpy sources
struct FlatScene {
  std::vector<ptc_object> objects;
  std::vector<uint32_t> object_material_indices;
  std::vector<ptc_sphere> spheres;
  std::vector<ptc_material> materials;
  std::vector<float> positions;
  std::vector<uint32_t> indices;
};

class SceneDescription {  // scene_description.hpp:27-49
public:
  std::string filename;
  Camera camera;
  int resolution[2] = {0, 0};
  int spp = 1;

  void add_material(const std::string& name, Material material);
  void add_object(Sphere sphere, const Mat4& transform, const std::string& material_name);
  void add_object(const Mesh& mesh, const Mat4& transform, const std::string& material_name);
  [[nodiscard]] const Mesh* get_mesh(const std::string& name) const;
  const Mesh& add_mesh(std::string name, Mesh&& mesh);
  [[nodiscard]] FlatScene build_scene() const;  // scene_description.cpp:12-117

private:
  struct Object {
    bool is_mesh;
    Sphere sphere;
    Mat4 transform;
  };
  std::vector<Object> objects_;
  std::map<std::string, Material, std::less<>> material_map_;
  std::vector<std::string> objects_material_mapping_;
  std::map<std::string, Mesh, std::less<>> mesh_map_;
};

// assets/json_parser.cpp:174-224 (scene_from_json) and assets/model_loader.cpp:11-44 (load_obj)
SceneDescription scene_from_json(const std::string& filename);
Mesh load_obj(const std::string& filename);

// image.cpp:9-22: RGBA8 PNG (the reference calls stbi_write_png)
bool write_png(const std::string& filename, int width, int height, const void* rgba);

}  // namespace hip_pt
