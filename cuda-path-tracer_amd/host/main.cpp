// hip_pt -- headless command-line front-end with the reference's flags
//   cuda_pt [--spp N] [-o out.png] <scene.json>          (reference: src/main.cpp:9-25, src/lib/configurations.cpp:7-45)
// following the call sequence of execute_cli_version (src/cli/cli.cpp:62-116): read scene, create_buffers,
// max_iterations = spp, spp x path_trace, synchronize, send_to_preview, write PNG, stage timings.
// Extra flags: --max-bounces N (the reference's cap is a compile-time 50), --denoise, --method, --gpu,
// --dump-scene FILE (flattened scene arrays, for tests; needs no GPU), --gpus N, --display final|color|normal|depth
// (DisplayBufferType of send_to_preview, path_tracer.cu:487-520: which buffer the PNG shows), --dump-raw FILE (the
// accumulated float framebuffers as they are: "PTRF", width, height, 7 as uint32, then colour rgb, normal xyz and
// depth planes as float32).  Without -o the reference opens its GLFW viewer; this build is headless and says so.
// --replay SCRIPT.json: the viewer WITHOUT a window -- App::main_loop / run_cuda (interactive-app/app.cpp:141-170) driven by
// a script of the events the reference takes from GLFW and ImGui (movement keys, right-drag, Space, the GUI's fields),
// one PNG per displayed frame (see run_replay below).
//
// --gpus N: one PROCESS per GPU (forked before anything touches HIP), rank r on device r % device_count.  The frame's
// rows are dealt to the ranks in blocks of 8 (ptc_set_interleave: sky rows are cheap, terrain rows expensive), the
// scene is replicated, nobody talks while tracing, and at the end every rank's rows go straight to rank 0 through the
// library's inter-process gather (ptc_band_export / import / publish, ptc_gather_present_rgba8: device-to-device
// copies, xGMI between GPUs).  Handles and the two barriers travel through a shared-memory page.
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <algorithm>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

#include "first_person_camera_controller.hpp"
#include "json.hpp"
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
  int gpus = 1;
  std::optional<std::string> dump_scene;
  std::optional<std::string> dump_raw;
  DisplayBufferType display = DisplayBufferType::final;
  std::optional<std::string> replay;
  bool dry_run = false;
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
               "      --gpus N         split the frame's rows over N processes / GPUs (rank r on device r %% device count)\n"
               "      --dump-scene F   write the flattened scene to F and exit (no GPU needed)\n"
               "      --display D      buffer the image shows: final (default) | color | normal | depth\n"
               "      --dump-raw F     also write the accumulated float buffers (colour, normal, depth) to F\n"
               "      --replay S.json  headless viewer: replay a script of viewer events, -o PREFIX -> PREFIX_0000.png ...\n"
               "      --dry-run        with --replay: no GPU, no images; print the camera after every event\n");
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
    else if (a == "--gpus") c.gpus = std::stoi(need("gpus"));
    else if (a == "--dump-scene") c.dump_scene = need("dump-scene");
    else if (a == "--dump-raw") c.dump_raw = need("dump-raw");
    else if (a == "--replay") c.replay = need("replay");
    else if (a == "--dry-run") c.dry_run = true;
    else if (a == "--display") {
      const std::string d = need("display");
      if (d == "final") c.display = DisplayBufferType::final;
      else if (d == "color") c.display = DisplayBufferType::color;
      else if (d == "normal") c.display = DisplayBufferType::normal;
      else if (d == "depth") c.display = DisplayBufferType::depth;
      else {
        std::fprintf(stderr, "Option '--display': unknown buffer '%s' (final, color, normal, depth)\n", d.c_str());
        std::exit(1);
      }
    }
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

// --dump-raw: the float framebuffers as they are.  fetch(which, dst) fills width*height*{3,3,1} floats.
template <typename Fetch>
bool write_raw(const std::string& path, uint32_t width, uint32_t height, Fetch fetch)
{
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  const uint32_t head[3] = {width, height, 7u};
  out.write("PTRF", 4);
  out.write(reinterpret_cast<const char*>(head), sizeof head);
  std::vector<float> plane((size_t)width * height * 3u);
  const int which[3] = {PTC_BUF_COLOR, PTC_BUF_NORMAL, PTC_BUF_DEPTH};
  for (int k = 0; k < 3; ++k) {
    const size_t floats = (size_t)width * height * (which[k] == PTC_BUF_DEPTH ? 1u : 3u);
    fetch(which[k], plane.data());
    out.write(reinterpret_cast<const char*>(plane.data()), (std::streamsize)(floats * sizeof(float)));
  }
  return (bool)out;
}

// ---- --replay: the interactive front-end without a window ---------------------------------------------------
// What App does (interactive-app/app.cpp): own a PathTracer and a FirstPersonCameraController on the scene's camera
// (:16-19), max_iterations = the scene's spp and buffers of the WINDOW's size, 800 x 800 (:36,127-131); then per loop
// turn (:162-170) poll the events -- a movement key or a right-drag moves the camera and restarts the accumulation
// (:64-67,108-113), Space restarts (:56), a resize reallocates (:45) -- run_cuda (:141-160: path_trace [+ denoise] +
// synchronise, repeated for 16 ms, then send_to_preview of the chosen DisplayBufferType), and let the GUI edit
// max_iterations, the method, the denoiser and the display (gui.cpp:84-108).  The script is those events:
//   {"window": [w, h], "iterations_per_frame": 2, "events": [
//      {"frames": 3},                      loop turns: run_cuda + one PNG each
//      {"key": "W", "count": 5},           GLFW_REPEAT of W/A/S/D/R/F            {"mouse": [dx, dy]}   right-drag, pixels
//      {"space": true}  {"resize": [w, h]}  {"denoise": true}  {"display": "normal"}  {"method": "megakernel"}
//      {"max_iterations": 64}  {"filter_size": 10}  {"speed": 0.5}  {"position": [x, y, z]}  {"reset": true} ]}
// A turn runs "iterations_per_frame" iterations (default 1) instead of as many as fit into 16 ms, so that a replay
// gives the same frames every time; "budget_ms": 16 restores the reference's clock.
int run_replay(const CliConfigurations& configs, SceneDescription& scene_desc)
{
  std::ifstream file(*configs.replay);
  if (!file) throw std::runtime_error("Cannot open replay script " + *configs.replay);
  std::stringstream text;
  text << file.rdbuf();
  const JsonPtr root = JsonReader(text.str()).parse();
  UResolution resolution{800u, 800u};  // app.cpp:36
  if (const Json* w = root->find("window")) resolution = {(unsigned)w->arr.at(0)->f(), (unsigned)w->arr.at(1)->f()};
  const int per_frame = root->find("iterations_per_frame") ? (int)root->at("iterations_per_frame").f() : 1;
  const double budget_ms = root->find("budget_ms") ? root->at("budget_ms").f() : 0.0;

  Camera& camera = scene_desc.camera;
  FirstPersonCameraController controller{camera};  // app.cpp:18 (reset(): the camera is re-expressed as yaw + pitch)
  if (configs.dry_run) {  // the controller alone (no GPU): one line per event
    std::printf("start: position %.9g %.9g %.9g yaw %.9g pitch %.9g rotation %.9g %.9g %.9g %.9g\n", camera.position[0], camera.position[1],
                camera.position[2], controller.yaw(), controller.pitch(), camera.rotation_wxyz[0], camera.rotation_wxyz[1],
                camera.rotation_wxyz[2], camera.rotation_wxyz[3]);
    int restarts = 0;
    for (const JsonPtr& ev : root->at("events").arr) {
      bool moved = false;
      if (const Json* j = ev->find("key")) {
        const int count = ev->find("count") ? (int)ev->at("count").f() : 1;
        for (int i = 0; i < count; ++i) moved = (!j->str.empty() && controller.on_key_press(j->str[0])) || moved;
      } else if (const Json* j = ev->find("mouse")) {
        const float rad = 0.01745329251994329576923690768489f;
        moved = controller.on_mouse_move(rad * j->arr.at(0)->f(), rad * j->arr.at(1)->f());
      } else if (const Json* j = ev->find("speed")) {
        controller.speed = j->f();
      } else if (const Json* j = ev->find("position")) {
        controller.set_position(j->arr.at(0)->f(), j->arr.at(1)->f(), j->arr.at(2)->f());
        controller.update_camera();
        moved = true;
      } else if (ev->find("reset")) {
        controller.reset();
        moved = true;
      } else {
        continue;
      }
      restarts += moved ? 1 : 0;
      std::printf("event: position %.9g %.9g %.9g yaw %.9g pitch %.9g rotation %.9g %.9g %.9g %.9g restarts %d\n", camera.position[0],
                  camera.position[1], camera.position[2], controller.yaw(), controller.pitch(), camera.rotation_wxyz[0],
                  camera.rotation_wxyz[1], camera.rotation_wxyz[2], camera.rotation_wxyz[3], restarts);
    }
    return 0;
  }
  PathTracer path_tracer{configs.gpu};
  path_tracer.max_bounces = configs.max_bounces;
  path_tracer.max_iterations = scene_desc.spp;      // app.cpp:130
  path_tracer.create_buffers(resolution, scene_desc);
  bool enable_denoising = configs.denoise;
  DisplayBufferType display = configs.display;
  if (configs.megakernel) path_tracer.current_gpu_method = GPUMethod::megakernel;

  int shown = 0;
  std::vector<uchar4> buffer;
  auto turn = [&]() {  // run_cuda, app.cpp:141-160
    const auto start = std::chrono::steady_clock::now();
    int done = 0;
    do {
      path_tracer.path_trace(camera, resolution);
      if (enable_denoising) path_tracer.denoise(resolution);
      path_tracer.synchronize();
      ++done;
    } while (budget_ms > 0.0 ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count() < budget_ms
                             : done < per_frame);
    buffer.resize((size_t)resolution.width * resolution.height);
    path_tracer.send_to_preview(buffer.data(), resolution, display);
    char name[32];
    std::snprintf(name, sizeof name, "_%04d.png", shown++);
    if (!write_png(*configs.output_filename + name, (int)resolution.width, (int)resolution.height, buffer.data()))
      throw std::runtime_error("Failed to write " + *configs.output_filename + name);
  };
  for (const JsonPtr& ev : root->at("events").arr) {
    if (const Json* j = ev->find("frames")) {
      for (int i = 0, n = (int)j->f(); i < n; ++i) turn();
    } else if (const Json* j = ev->find("key")) {
      const int count = ev->find("count") ? (int)ev->at("count").f() : 1;
      for (int i = 0; i < count; ++i)
        if (!j->str.empty() && controller.on_key_press(j->str[0])) path_tracer.restart();  // app.cpp:64-67
    } else if (const Json* j = ev->find("mouse")) {
      const float rad = 0.01745329251994329576923690768489f;  // glm::radians, app.cpp:110-111
      if (controller.on_mouse_move(rad * j->arr.at(0)->f(), rad * j->arr.at(1)->f())) path_tracer.restart();
    } else if (ev->find("space")) {
      path_tracer.restart();  // app.cpp:56
    } else if (const Json* j = ev->find("resize")) {
      resolution = {(unsigned)j->arr.at(0)->f(), (unsigned)j->arr.at(1)->f()};
      path_tracer.resize_image(resolution);  // app.cpp:45
    } else if (const Json* j = ev->find("denoise")) {
      enable_denoising = j->b;
    } else if (const Json* j = ev->find("display")) {
      display = j->str == "color" ? DisplayBufferType::color : j->str == "normal" ? DisplayBufferType::normal
                : j->str == "depth" ? DisplayBufferType::depth : DisplayBufferType::final;
    } else if (const Json* j = ev->find("method")) {
      path_tracer.current_gpu_method = j->str == "megakernel" ? GPUMethod::megakernel : GPUMethod::streaming;
    } else if (const Json* j = ev->find("max_iterations")) {
      path_tracer.max_iterations = std::max(1, (int)j->f());  // gui.cpp:103-104
    } else if (const Json* j = ev->find("filter_size")) {
      path_tracer.atrous_denoiser.filter_size = (int)j->f();
    } else if (const Json* j = ev->find("speed")) {
      controller.speed = j->f();
    } else if (const Json* j = ev->find("position")) {  // the camera panel's "Translation" field, first_person_camera_controller.cpp:121-126
      controller.set_position(j->arr.at(0)->f(), j->arr.at(1)->f(), j->arr.at(2)->f());
      controller.update_camera();
      path_tracer.restart();
    } else if (ev->find("reset")) {
      controller.reset();
      path_tracer.restart();
    } else {
      throw std::runtime_error("replay: unknown event");
    }
  }
  const ptc_stats st = path_tracer.stats();
  std::printf("replayed %s: %d frames shown, %llu rays, camera at (%g, %g, %g) yaw %g pitch %g\n", configs.replay->c_str(), shown,
              (unsigned long long)st.rays_total, camera.position[0], camera.position[1], camera.position[2], controller.yaw(),
              controller.pitch());
  return 0;
}

// ---- --gpus N -------------------------------------------------------------------------------------------------
constexpr int kMaxRanks = 64;
constexpr uint32_t kBlockRows = 8;
struct Shared {  // one MAP_SHARED page set up before the fork
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<int> failed;
  ptc_band_handle handles[kMaxRanks];
  unsigned long long rays[kMaxRanks];
};

// all ranks meet; false when a rank has failed or nothing happens for two minutes
bool barrier(Shared* sh, int n)
{
  const uint32_t gen = sh->generation.load();
  if (sh->arrived.fetch_add(1u) + 1u == (uint32_t)n) {
    sh->arrived.store(0u);
    sh->generation.fetch_add(1u);
    return sh->failed.load() == 0;
  }
  for (int waited_ms = 0; sh->generation.load() == gen; ++waited_ms) {
    if (sh->failed.load() != 0 || waited_ms > 120000) return false;
    usleep(1000);
  }
  return sh->failed.load() == 0;
}

int run_rank(const CliConfigurations& configs, const SceneDescription& scene_desc, Shared* sh, int rank, int world)
try {
  int devices = 0;
  if (ptc_device_count(&devices) < 0 || devices <= 0) throw std::runtime_error("no HIP device");
  const UResolution resolution{(unsigned)scene_desc.resolution[0], (unsigned)scene_desc.resolution[1]};
  const int spp = scene_desc.spp;
  Stopwatch stopwatch;
  PathTracer path_tracer{rank % devices};
  path_tracer.max_bounces = configs.max_bounces;
  path_tracer.create_buffers(resolution, scene_desc);
  auto check = [&](int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ptc_last_error(path_tracer.handle()));
  };
  check(ptc_set_interleave(path_tracer.handle(), (uint32_t)rank, (uint32_t)world, kBlockRows), "set_interleave");
  // every rank numbers its paths from 0: an offset keeps the ranks' random streams apart (path_tracer.cu:300)
  check(ptc_set_param(path_tracer.handle(), "slot_offset", (int)((unsigned)rank * resolution.width * resolution.height)), "slot_offset");
  if (rank != 0) check(ptc_band_export(path_tracer.handle(), &sh->handles[rank]), "band_export");
  path_tracer.synchronize();
  if (!barrier(sh, world)) throw std::runtime_error("another rank failed during set-up");
  if (rank == 0) {
    for (int r = 1; r < world; ++r) check(ptc_band_import(path_tracer.handle(), (uint32_t)r, &sh->handles[r]), "band_import");
    std::printf("Start path tracing\nspp: %d\nwidth: %u, height: %u\nranks: %d on %d device(s)\n", spp, resolution.width,
                resolution.height, world, devices);
  }
  stopwatch.end_stage("Initialization");
  path_tracer.max_iterations = spp;
  for (int i = 0; i < spp; ++i) path_tracer.path_trace(scene_desc.camera, resolution);
  // the rows the image will show (ptc_gather_present_rgba8 gathers the buffer of its display type)
  const int shown = configs.display == DisplayBufferType::normal ? PTC_BUF_NORMAL
                    : configs.display == DisplayBufferType::depth ? PTC_BUF_DEPTH : PTC_BUF_COLOR;
  if (rank != 0) check(ptc_band_publish(path_tracer.handle(), shown), "band_publish");
  else path_tracer.synchronize();
  sh->rays[rank] = path_tracer.stats().rays_total;
  if (!barrier(sh, world)) throw std::runtime_error("another rank failed while tracing");
  stopwatch.end_stage("Path Tracing");
  if (rank == 0) {
    std::vector<uchar4> buffer((size_t)resolution.width * resolution.height);
    check(ptc_gather_present_rgba8(path_tracer.handle(), buffer.data(), 0, static_cast<int>(configs.display)), "gather_present");
    const fs::path output_path{*configs.output_filename};
    if (output_path.extension() == ".png") {
      if (!write_png(output_path.string(), (int)resolution.width, (int)resolution.height, buffer.data()))
        std::fprintf(stderr, "Failed to write to image file %s\n", output_path.string().c_str());
    } else {
      std::fprintf(stderr, "%s has an unrecognized extension\n", output_path.string().c_str());
    }
    stopwatch.end_stage("Gather + write image file");
  }
  if (configs.dump_raw) {  // every buffer is one more publish / gather round
    const int which[3] = {PTC_BUF_COLOR, PTC_BUF_NORMAL, PTC_BUF_DEPTH};
    std::vector<std::vector<float>> planes(3);
    for (int k = 0; k < 3; ++k) {
      if (!barrier(sh, world)) throw std::runtime_error("another rank failed");  // the root has pulled the previous rows
      if (rank != 0) check(ptc_band_publish(path_tracer.handle(), which[k]), "band_publish");
      if (!barrier(sh, world)) throw std::runtime_error("another rank failed while publishing");
      if (rank == 0) {
        planes[k].resize((size_t)resolution.width * resolution.height * (which[k] == PTC_BUF_DEPTH ? 1u : 3u));
        check(ptc_gather_frame(path_tracer.handle(), which[k], planes[k].data(), 0), "gather_frame");
      }
    }
    if (rank == 0) {
      int k = 0;
      if (!write_raw(*configs.dump_raw, resolution.width, resolution.height,
                     [&](int, float* dst) { std::memcpy(dst, planes[k].data(), planes[k].size() * sizeof(float)); ++k; }))
        std::fprintf(stderr, "Failed to write %s\n", configs.dump_raw->c_str());
    }
  }
  if (!barrier(sh, world)) throw std::runtime_error("another rank failed at the end");  // peers keep their rows until here
  if (rank == 0) {
    unsigned long long rays = 0;
    for (int r = 0; r < world; ++r) rays += sh->rays[r];
    std::printf("Done path tracing %s!\n\n", scene_desc.filename.c_str());
    stopwatch.report();
    std::printf("rays: %llu over %d ranks\n", rays, world);
  }
  return 0;
} catch (const std::exception& e) {
  sh->failed.store(1);
  std::fprintf(stderr, "hip_pt rank %d fatal error: %s\n", rank, e.what());
  return 1;
}

int run_ranks(const CliConfigurations& configs, const SceneDescription& scene_desc)
{
  const int world = configs.gpus;
  if (world > kMaxRanks) {
    std::fprintf(stderr, "hip_pt: at most %d ranks\n", kMaxRanks);
    return 1;
  }
  if (configs.denoise || configs.megakernel) {
    std::fprintf(stderr, "hip_pt: --gpus N renders in streaming mode without the denoiser (it needs the whole frame in one context)\n");
    return 1;
  }
  void* page = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (page == MAP_FAILED) {
    std::perror("mmap");
    return 1;
  }
  Shared* sh = new (page) Shared{};
  std::fflush(stdout);
  std::fflush(stderr);
  std::vector<pid_t> children;
  for (int r = 0; r < world; ++r) {
    const pid_t pid = fork();  // nothing in this process has touched HIP yet
    if (pid < 0) {
      std::perror("fork");
      sh->failed.store(1);
      break;
    }
    if (pid == 0) {
      const int rc = run_rank(configs, scene_desc, sh, r, world);
      std::fflush(stdout);
      std::fflush(stderr);
      _exit(rc);
    }
    children.push_back(pid);
  }
  int rc = sh->failed.load() ? 1 : 0;
  for (pid_t pid : children) {
    int status = 0;
    if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = 1;
  }
  munmap(page, sizeof(Shared));
  return rc;
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
  if (configs.replay && configs.dry_run) return run_replay(configs, scene_desc);
  if (!configs.output_filename) {
    std::fprintf(stderr, "hip_pt: this build is headless (no GLFW viewer); give -o <file.png>, or --replay <script.json> -o <prefix>\n");
    return 1;
  }

  if (configs.replay) return run_replay(configs, scene_desc);
  if (configs.gpus > 1) return run_ranks(configs, scene_desc);

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
  path_tracer.send_to_preview(buffer.data(), resolution, configs.display);
  const fs::path output_path{*configs.output_filename};
  if (output_path.extension() == ".png") {
    if (!write_png(output_path.string(), (int)resolution.width, (int)resolution.height, buffer.data()))
      std::fprintf(stderr, "Failed to write to image file %s\n", output_path.string().c_str());
  } else {
    std::fprintf(stderr, "%s has an unrecognized extension\n", output_path.string().c_str());
  }
  if (configs.dump_raw &&
      !write_raw(*configs.dump_raw, resolution.width, resolution.height, [&](int which, float* dst) {
        if (ptc_download(path_tracer.handle(), which, dst, 0) < 0) throw std::runtime_error(ptc_last_error(path_tracer.handle()));
      }))
    std::fprintf(stderr, "Failed to write %s\n", configs.dump_raw->c_str());
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
