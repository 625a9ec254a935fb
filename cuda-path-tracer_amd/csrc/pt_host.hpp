// pt_host.hpp -- host-side pieces of the render core that need no GPU.
#pragma once

#include "../../include/ptcore.h"
#include "pt_math.hpp"

#include <vector>

namespace pt {

// bvh_from_mesh (reference accelerators/bvh.cpp:211-253); returns node count or a negative ptc_status
int build_bvh(const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
              ptc_bvh_node* out, uint32_t* max_depth);

// glm::inverse(mat4) as Transform's constructor applies it (reference transform.hpp:16-19)
m4 inverse(const m4& m);

// translate(identity, position) * mat4_cast(rotation)   (reference camera.cpp:5-13)
m4 camera_matrix(const float position[3], const float rotation_wxyz[4]);

// Per-object part of SceneDescription::build_scene (reference scene_description.cpp:17-52)
int make_object(uint32_t type, uint32_t index, const float* m16, const ptc_sphere* sphere, const float* mesh_aabb6,
                ptc_object* out);

// The reference tree re-laid for the fast traversal (see DScene in pt_device.hpp).
struct WideAccel {
  std::vector<float4> wide;         // 4 per inner node, breadth-first order of the inner nodes
  std::vector<uint32_t> tri_order;  // depth-first leaf rank -> triangle number (index offset / 3)
  uint32_t root_ref = 0;
  float root_min[3] = {0, 0, 0};
  float root_max[3] = {0, 0, 0};
};
int build_wide(const ptc_bvh_node* nodes, uint32_t count, WideAccel& out);

// Four-wide collapse of the reference tree for the persistent traversal (see DScene::bvh4 in pt_device.hpp).
struct Wide4Accel {
  // Nodes of 64 bytes (16 dwords each), depth-first preorder: origin xyz (f32) | exponents of the three
  // power-of-two grid steps | child planes as 8-bit grid coordinates, SoA (lo_x[4] lo_y[4] lo_z[4] hi_x[4] hi_y[4]
  // hi_z[4]), rounded outwards so that every quantised child box contains the exact one | 2 pad | the four child refs
  std::vector<uint32_t> nodes_q;
  std::vector<float4> leaf_parent;  // 2 per triangle (depth-first leaf order): box of the leaf's parent node; + the dummy's
  uint32_t dummy_ref = 0;           // kLeafBit | triangle count: what the unused child slots of a node point at, see quantise()
  uint32_t root_ref = 0;
  uint32_t depth = 0;               // levels of four-wide nodes above the deepest leaf
  uint32_t node_count = 0;
};
// leaf_rank comes from build_wide (same depth-first leaf order as WideAccel::tri_order)
int build_wide4(const ptc_bvh_node* nodes, uint32_t count, Wide4Accel& out);

// World-space triangles of one instance in depth-first leaf order: 3 float4 per triangle.
void build_instance_triangles(const m4& m, const float* positions, const uint32_t* indices,
                              const std::vector<uint32_t>& tri_order, float4* out);

}  // namespace pt
