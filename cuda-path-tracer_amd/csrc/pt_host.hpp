// pt_host.hpp -- host-side pieces of the render core that need no GPU.
#pragma once

#include "../../include/ptcore.h"
#include "pt_math.hpp"

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

}  // namespace pt
