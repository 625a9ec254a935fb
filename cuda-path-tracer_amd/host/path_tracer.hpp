// path_tracer.hpp -- C++ host mirror of the reference's `class PathTracer`
// (/root/reference/src/lib/path_tracer.hpp:60-99) over the C ABI of libptcore.so (include/ptcore.h).
// Same public methods and mutable fields, so a caller written against the reference class (its CLI,
// cli/cli.cpp:86-105, or its GUI) compiles against this one.  Header-only; errors become exceptions
// (the reference exits the process: cuda_utils/cuda_check.cpp:7-23).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "scene_description.hpp"

namespace hip_pt {

struct UResolution {  // resolution.hpp:13-22
  unsigned int width = 0;
  unsigned int height = 0;
};

enum class DisplayBufferType { final, color, normal, depth };  // path_tracer.hpp:19
enum class GPUMethod { megakernel, streaming };                // path_tracer.hpp:57

struct EdgeAvoidingATrousDenoiser {  // denoising/edge_avoiding_a_trous_denoiser.hpp:7-12
  int filter_size = 10;
  float color_weight = 0.45f;
  float normal_weight = 0.30f;
  float position_weight = 0.25f;
};

struct uchar4 {
  unsigned char x, y, z, w;
};

class PathTracer {
public:
  int max_iterations = 1;
  GPUMethod current_gpu_method = GPUMethod::streaming;
  EdgeAvoidingATrousDenoiser atrous_denoiser{};
  int max_bounces = 50;  // reference: compile-time constant, path_tracer.cu:27

  explicit PathTracer(int device = 0)
  {
    ptc_config cfg{device, max_bounces, PTC_METHOD_STREAMING, 0};
    if (ptc_create(&cfg, &ctx_) < 0) throw std::runtime_error(std::string("ptc_create: ") + ptc_last_error(nullptr));
  }
  ~PathTracer() { ptc_destroy(ctx_); }
  PathTracer(const PathTracer&) = delete;
  PathTracer& operator=(const PathTracer&) = delete;

  void restart() { check(ptc_restart(ctx_), "restart"); }
  [[nodiscard]] int iteration() const noexcept { return ptc_iteration(ctx_); }
  void resize_image(UResolution r) { check(ptc_resize(ctx_, r.width, r.height), "resize_image"); }

  void create_buffers(UResolution r, const SceneDescription& scene)
  {
    const FlatScene flat = scene.build_scene();
    ptc_scene_desc d{};
    d.objects = flat.objects.data();
    d.object_count = (uint32_t)flat.objects.size();
    d.object_material_indices = flat.object_material_indices.data();
    d.spheres = flat.spheres.data();
    d.sphere_count = (uint32_t)flat.spheres.size();
    d.materials = flat.materials.data();
    d.material_count = (uint32_t)flat.materials.size();
    d.positions = flat.positions.data();
    d.vertex_count = (uint32_t)(flat.positions.size() / 3);
    d.indices = flat.indices.data();
    d.index_count = (uint32_t)flat.indices.size();
    check(ptc_upload_scene(ctx_, &d), "create_buffers");
    resize_image(r);
  }

  void path_trace(const Camera& camera, UResolution)
  {
    push_fields();
    const ptc_camera c = camera.to_c();
    check(ptc_trace(ctx_, &c), "path_trace");
  }
  void denoise(UResolution)
  {
    push_fields();
    check(ptc_denoise(ctx_), "denoise");
  }
  // dst: device (or managed) pointer like the reference's PBO when dst_is_device, else host memory
  void send_to_preview(uchar4* dst, UResolution, DisplayBufferType type = DisplayBufferType::final,
                       bool dst_is_device = false) const
  {
    check(ptc_present_rgba8(ctx_, dst, dst_is_device ? 1 : 0, static_cast<int>(type)), "send_to_preview");
  }
  void synchronize() { check(ptc_synchronize(ctx_), "synchronize"); }
  [[nodiscard]] ptc_stats stats() const
  {
    ptc_stats s{};
    check(ptc_get_stats(ctx_, &s), "stats");
    return s;
  }
  ptc_ctx* handle() { return ctx_; }

private:
  void push_fields()
  {
    check(ptc_set_max_iterations(ctx_, max_iterations), "max_iterations");
    check(ptc_set_method(ctx_, current_gpu_method == GPUMethod::megakernel ? PTC_METHOD_MEGAKERNEL : PTC_METHOD_STREAMING), "method");
    check(ptc_set_max_bounces(ctx_, max_bounces), "max_bounces");
    const ptc_denoiser_params p{atrous_denoiser.filter_size, atrous_denoiser.color_weight, atrous_denoiser.normal_weight,
                                atrous_denoiser.position_weight};
    check(ptc_set_denoiser_params(ctx_, &p), "denoiser");
  }
  void check(int rc, const char* what) const
  {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ptc_last_error(ctx_));
  }
  ptc_ctx* ctx_ = nullptr;
};

}  // namespace hip_pt
