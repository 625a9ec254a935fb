// hip_pt -- headless command-line front-end with the reference's flags
//   cuda_pt [--spp N] [-o out.png] <scene.json>          (reference: src/main.cpp:9-25, src/lib/configurations.cpp:7-45)
// following the call sequence of execute_cli_version (src/cli/cli.cpp:62-116): read scene, create_buffers,
// max_iterations = spp, spp x path_trace, synchronize, send_to_preview, write PNG, stage timings.
// Extra flags: --max-bounces N (the reference's cap is a compile-time 50), --denoise, --method, --gpu,
// --dump-scene FILE (flattened scene arrays, for tests; needs no GPU).  Without -o the reference opens its
// GLFW viewer; this build is headless and says so.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <optional>
#include <string>
#include <vector>

#include "path_tracer.hpp"

namespace fs = std::filesystem;
using namespace hip_pt;

namespace {

struct CliConfigurations {  // configurations.hpp:11-15
  std::string filename;
  std::optional<int> spp;
  std::optional<std::string> output_filename;
  int max_bounces = 50;
  bool denoise = false;
  bool megakernel = false;
  int gpu = 0;
  std::optional<std::string> dump_scene;
};

void usage()
{
  std::fprintf(stderr,
               "A Path Tracer for AMD MI355X (hip_pt; command line of cuda_pt)\nUsage:\n  hip_pt [OPTION...] <filename>\n\n"
               "  -o, --output arg     Output path tracing result to a file\n"
               "  -h, --help           Print this message\n"
               "      --spp arg        Sample per pixel (overrides the setting in the scene file)\n"
               "      --max-bounces N  bounce cap (default 50)\n"
               "      --denoise        run the A-Trous denoiser before writing the image\n"
               "      --method M       streaming (default) | megakernel\n"
               "      --gpu N          HIP device ordinal\n"
               "      --dump-scene F   write the flattened scene to F and exit (no GPU needed)\n");
}

CliConfigurations parse_cli_args(int argc, char** argv)
{
  CliConfigurations c;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto need = [&](const char* name) -> std::string {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "Option '%s' is missing an argument\n", name);
        std::exit(1);
      }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") {
      usage();
      std::exit(0);
    } else if (a == "-o" || a == "--output") c.output_filename = need("output");
    else if (a.rfind("--output=", 0) == 0) c.output_filename = a.substr(9);
    else if (a == "--spp") c.spp = std::stoi(need("spp"));
    else if (a.rfind("--spp=", 0) == 0) c.spp = std::stoi(a.substr(6));
    else if (a == "--max-bounces") c.max_bounces = std::stoi(need("max-bounces"));
    else if (a == "--denoise") c.denoise = true;
    else if (a == "--method") c.megakernel = need("method") == "megakernel";
    else if (a == "--gpu") c.gpu = std::stoi(need("gpu"));
    else if (a == "--dump-scene") c.dump_scene = need("dump-scene");
    else if (!a.empty() && a[0] == '-') {
      std::fprintf(stderr, "Option '%s' does not exist\n", a.c_str());
      std::exit(1);
    } else c.filename = a;
  }
  if (c.filename.empty()) {
    std::fprintf(stderr, "Usage: hip_pt [options] <filename>\nRun 'hip_pt --help' for more information");
    std::exit(1);
  }
  return c;
}

// assets/assets.cpp:6-22: the OUTERMOST ancestor of the working directory that contains "assets/"
std::optional<fs::path> locate_asset_path(const fs::path& current_path)
{
  std::optional<fs::path> result;
  for (auto path = current_path; path != current_path.root_path(); path = path.parent_path()) {
    const auto assets_path = path / "assets";
    if (fs::exists(assets_path) && fs::is_directory(assets_path)) result = assets_path;
  }
  return result;
}

class Stopwatch {  // cli.cpp:27-60
  using clock = std::chrono::steady_clock;
  clock::time_point start_ = clock::now(), last_ = clock::now();
  std::vector<std::pair<std::string, double>> entries_;

public:
  void end_stage(std::string name)
  {
    const auto now = clock::now();
    entries_.emplace_back(std::move(name), std::chrono::duration<double>(now - last_).count());
    last_ = now;
  }
  void report() const
  {
    std::printf("Elapsed time\n===========\n");
    for (const auto& [name, s] : entries_) std::printf("%s: %gs\n", name.c_str(), s);
    std::printf("Total: %gs\n", std::chrono::duration<double>(last_ - start_).count());
  }
};

template <typename T>
void dump_vec(std::ofstream& out, const std::vector<T>& v)
{
  const uint64_t n = v.size();
  out.write(reinterpret_cast<const char*>(&n), 8);
  out.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(n * sizeof(T)));
}

}  // namespace

int main(int argc, char** argv)
try {
  const CliConfigurations configs = parse_cli_args(argc, argv);
  Stopwatch stopwatch;

  // read_scene, assets/scene_parser.cpp:6-25: asset_path / filename (a path that exists as given is used as is)
  fs::path path = configs.filename;
  if (!fs::exists(path)) {
    const auto assets = locate_asset_path(fs::current_path());
    if (!assets) {
      std::fprintf(stderr, "Panic: Cannot find assets directory\n");
      return 1;
    }
    path = *assets / configs.filename;
  }
  path = fs::canonical(path);
  if (path.extension() != ".json") {
    std::fprintf(stderr, "Panic: Unsupported file extension %s!\n", path.extension().string().c_str());
    return 1;
  }
  SceneDescription scene_desc = scene_from_json(path.string());
  scene_desc.filename = configs.filename;
  if (configs.spp) scene_desc.spp = *configs.spp;
  stopwatch.end_stage("Scene loading");

  if (configs.dump_scene) {
    const FlatScene flat = scene_desc.build_scene();
    std::ofstream out(*configs.dump_scene, std::ios::binary);
    dump_vec(out, flat.objects);
    dump_vec(out, flat.object_material_indices);
    dump_vec(out, flat.spheres);
    dump_vec(out, flat.materials);
    dump_vec(out, flat.positions);
    dump_vec(out, flat.indices);
    const ptc_camera cam = scene_desc.camera.to_c();
    out.write(reinterpret_cast<const char*>(&cam), sizeof cam);
    const int32_t tail[3] = {scene_desc.resolution[0], scene_desc.resolution[1], scene_desc.spp};
    out.write(reinterpret_cast<const char*>(tail), sizeof tail);
    return 0;
  }
  if (!configs.output_filename) {
    std::fprintf(stderr, "hip_pt: this build is headless (no GLFW viewer); give -o <file.png>\n");
    return 1;
  }

  const UResolution resolution{(unsigned)scene_desc.resolution[0], (unsigned)scene_desc.resolution[1]};
  const int spp = scene_desc.spp;
  PathTracer path_tracer{configs.gpu};
  path_tracer.max_bounces = configs.max_bounces;
  if (configs.megakernel) path_tracer.current_gpu_method = GPUMethod::megakernel;
  path_tracer.create_buffers(resolution, scene_desc);
  path_tracer.synchronize();
  std::printf("Start path tracing\nspp: %d\nwidth: %u, height: %u\n", spp, resolution.width, resolution.height);
  stopwatch.end_stage("Initialization");

  path_tracer.max_iterations = spp;
  for (int i = 0; i < spp; ++i) path_tracer.path_trace(scene_desc.camera, resolution);
  if (configs.denoise) path_tracer.denoise(resolution);
  path_tracer.synchronize();
  stopwatch.end_stage("Path Tracing");

  std::vector<uchar4> buffer((size_t)resolution.width * resolution.height);
  path_tracer.send_to_preview(buffer.data(), resolution);
  const fs::path output_path{*configs.output_filename};
  if (output_path.extension() == ".png") {
    if (!write_png(output_path.string(), (int)resolution.width, (int)resolution.height, buffer.data()))
      std::fprintf(stderr, "Failed to write to image file %s\n", output_path.string().c_str());
  } else {
    std::fprintf(stderr, "%s has an unrecognized extension\n", output_path.string().c_str());
  }
  stopwatch.end_stage("Write image file");

  const ptc_stats st = path_tracer.stats();
  std::printf("Done path tracing %s!\n\n", scene_desc.filename.c_str());
  stopwatch.report();
  std::printf("rays: %llu  triangles: %u  bvh depth: %u\n", (unsigned long long)st.rays_total, st.triangle_count, st.bvh_max_depth);
  return 0;
} catch (const std::exception& e) {
  std::fprintf(stderr, "hip_pt fatal error: %s\n", e.what());
  return 1;
}
