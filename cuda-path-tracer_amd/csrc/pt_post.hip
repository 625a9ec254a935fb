// pt_post.hip -- what follows the path: k_preview*, k_pack, k_gather_bands (multi-GPU present), the A-Trous denoiser
// (k_denoise_positions, k_denoise, k_denoise_lds<step>), k_selftest.

#include "pt_device.hpp"
#include "pt_rng.hpp"
#include "pt_beam_rules.hpp"
#include "pt_feed_rules.hpp"
static_assert(pt::beam_rules::kLeaf == pt::kLeafBit, "pt_beam_rules.hpp restates the leaf bit of the four-wide node (pt_device.hpp)");
static_assert((uint32_t)pt::beam_rules::kEntries == pt::kBeamEntries, "pt_beam_rules.hpp restates the entries per tile (pt_device.hpp)");
static_assert(pt::feed_rules::kBatch == (uint32_t)pt::kWave, "a feed batch is one wavefront's worth of rays");
#include <float.h>

namespace pt {

#include "pt_kernels_common.inc"

// preview_kernel / preview_depth_kernel, path_tracer.cu:334-385.
// mode 0: rgb of buf ; 1: normal view (xyz*0.5+0.5) ; 2: depth view (1/w, alpha 1)
__global__ __launch_bounds__(256) void k_preview(const float4* buf, uint32_t pix_count, int mode, uint32_t* rgba)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  const float4 v = buf[i];
  f3 c = xyz(v);
  uint32_t alpha = 255u;
  if (mode == 2) {
    const float d = 1.0f / v.w;
    c = mk3(d, d, d);
    alpha = 1u;
  } else if (mode == 1) {
    c = c * 0.5f + mk3(0.5f, 0.5f, 0.5f);
  }
  const float g = 1.f / 2.2f;
  c = mk3(powf(c.x, g), powf(c.y, g), powf(c.z, g));
  auto to255 = [](float x) -> uint32_t { return (uint32_t)(unsigned char)(sel_min(sel_max(x, 0.f), 1.f) * 255.99f); };
  rgba[i] = to255(c.x) | (to255(c.y) << 8) | (to255(c.z) << 16) | (alpha << 24);
}

// Multi-GPU gather on the root, ONE launch for all ranks: blockIdx.y = source band.  A band is a rank's packed rows
// (channels floats per pixel); src is the root's own buffer or a peer's buffer mapped through HIP IPC, read where it
// lies -- over xGMI when the peer is another GPU, every peer -> root link busy at once, no staging copy.  Thread i of a
// band moves float i (consecutive threads read consecutive floats; a row of the band is a run of the frame).
__global__ __launch_bounds__(256) void k_gather_bands(DGatherBands bands, int channels, uint32_t frame_pixels, float* frame)
{
  const DGatherBands::Src& b = bands.src[blockIdx.y];
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= (uint64_t)b.pix_count * (uint32_t)channels) return;
  const uint32_t s = (uint32_t)(i / (uint32_t)channels), c = (uint32_t)(i % (uint32_t)channels);
  const uint32_t pixel = band_pixel(b.band, s);
  if (pixel >= frame_pixels) return;  // (ptc_band_import has checked the geometry; a stray handle must not write outside)
  frame[(size_t)pixel * (size_t)channels + c] = __builtin_nontemporal_load(&b.src[i]);
}

// preview_kernel / preview_depth_kernel on a packed frame (the gathered frame of a multi-GPU run)
__global__ __launch_bounds__(256) void k_preview_packed(const float* buf, uint32_t pix_count, int channels, int mode, uint32_t* rgba)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  f3 c;
  uint32_t alpha = 255u;
  if (mode == 2) {
    const float d = 1.0f / buf[(size_t)i * (size_t)channels];
    c = mk3(d, d, d);
    alpha = 1u;
  } else {
    c = mk3(buf[3u * (size_t)i], buf[3u * (size_t)i + 1u], buf[3u * (size_t)i + 2u]);
    if (mode == 1) c = c * 0.5f + mk3(0.5f, 0.5f, 0.5f);
  }
  const float g = 1.f / 2.2f;
  c = mk3(powf(c.x, g), powf(c.y, g), powf(c.z, g));
  auto to255 = [](float x) -> uint32_t { return (uint32_t)(unsigned char)(sel_min(sel_max(x, 0.f), 1.f) * 255.99f); };
  rgba[i] = to255(c.x) | (to255(c.y) << 8) | (to255(c.z) << 16) | (alpha << 24);
}

// float4 framebuffer -> packed vec3 (which 0) or the w channel (which 1)
__global__ __launch_bounds__(256) void k_pack(const float4* buf, uint32_t pix_count, int which, float* dst)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  const float4 v = buf[i];
  if (which == 0) {
    dst[3u * (size_t)i] = v.x;
    dst[3u * (size_t)i + 1u] = v.y;
    dst[3u * (size_t)i + 2u] = v.z;
  } else {
    dst[i] = v.w;
  }
}

// denoising_kernel, denoising/edge_avoiding_a_trous_denoiser.cu:24-86, in two kernels.
// The reference rebuilds the view ray of every one of the 25 taps in every pass (generate_ray: a normalise and
// a matrix product each); the tap positions depend only on (pixel, accumulated depth), so k_denoise_positions
// computes them once per denoise call and the four passes read them back (16 B per tap instead of ~40
// instructions).  The reference clamps taps to [0,W] x [0,H] INCLUSIVE (cu:39-42): column W aliases the next
// row's column 0 but keeps its own view ray, and row H is out of bounds; taps on column W / row H therefore
// rebuild their ray here (edge pixels only), and an index beyond the array reads element W*H-1.
__global__ __launch_bounds__(256) void k_denoise_positions(DCamera cam, uint32_t pix_count, const float4* nd, float4* pos)
{
  const uint32_t index = blockIdx.x * 256u + threadIdx.x;
  if (index >= pix_count) return;
  const int x = (int)(index % cam.width), y = (int)(index / cam.width);
  f3 ro, rd;
  generate_ray(cam, (float)x + 0.5f, (float)y + 0.5f, ro, rd);
  const f3 p = ro + rd * nd[index].w;
  pos[index] = make_float4(p.x, p.y, p.z, 0.0f);
}

// Arithmetic of the pass: the three edge-stopping weights min(exp(-d/phi), 1) of a tap (cu:63-77) multiply to
// exp(-(dc/c_phi + dn/(step^2 n_phi) + dp/p_phi)) -- every d is a sum of squares, so no factor exceeds 1 and the
// clamps are inert.  The kernel evaluates that single exponential with v_exp_f32 on a base-2 argument whose
// three reciprocal scale factors are computed once per pass (the reference: three divisions and three expf per
// tap, 75 of each per pixel per pass, which made this kernel VALU-bound).  This stage is outside the random-number
// feedback loop and is compared with the oracle under a tolerance (1e-5 absolute on the radiance,
// tests/test_gpu_parity.py), not bit for bit; contraction into FMAs is allowed here for the same reason.
// kInterior: every tap of the tile is inside the image (no clamp, no off-by-one column / row): the common case,
// decided per tile so that the wavefront does not branch per tap.
template <bool kInterior>
__device__ __forceinline__ void denoise_pixel(const DCamera& cam, const uint32_t pix_count, const float4* color,
                                              const float4* nd, const float4* pos, float4* out, const int step_width,
                                              const DDenoise& prm, const int x, const int y)
{
#pragma clang fp contract(fast)
  const uint32_t W = cam.width, H = cam.height;
  const uint32_t index = (uint32_t)x + (uint32_t)y * W;
  const float kernel[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
  const f3 cval = xyz(color[index]);
  const f3 nval = xyz(nd[index]);
  const f3 pval = xyz(pos[index]);
  f3 sum = mk3(0.f, 0.f, 0.f);
  float cum_w = 0.0f;
  const float step2 = (float)(step_width * step_width);
  constexpr float kLog2e = 1.4426950408889634f;
  const float kc = -kLog2e / prm.c_phi, kn = -kLog2e / (step2 * prm.n_phi), kp = -kLog2e / prm.p_phi;
#pragma unroll 1
  for (int dy = -2; dy <= 2; ++dy) {
    int v = y + dy * step_width;
    if (!kInterior) v = v < 0 ? 0 : (v > (int)H ? (int)H : v);
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      int u = x + dx * step_width;
      if (!kInterior) u = u < 0 ? 0 : (u > (int)W ? (int)W : u);
      uint32_t ti = (uint32_t)u + (uint32_t)v * W;
      if (!kInterior && ti >= pix_count) ti = pix_count - 1u;
      const f3 ctemp = xyz(color[ti]);
      const float4 ndt = nd[ti];
      f3 ptmp;
      if (!kInterior && (u == (int)W || v == (int)H)) {  // the reference's off-by-one taps keep their own view ray
        f3 to, td;
        generate_ray(cam, (float)u + 0.5f, (float)v + 0.5f, to, td);
        ptmp = to + td * ndt.w;
      } else {
        ptmp = xyz(pos[ti]);
      }
      const f3 tc = cval - ctemp, tn = nval - xyz(ndt), tp = pval - ptmp;
      const float arg = dot(tc, tc) * kc + dot(tn, tn) * kn + dot(tp, tp) * kp;
      const float weight = __builtin_amdgcn_exp2f(arg);
      const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
      const float wk = weight * kernel[adx < ady ? adx : ady];
      sum = sum + ctemp * wk;
      cum_w += wk;
    }
  }
  const float inv_w = 1.0f / cum_w;
  out[index] = make_float4(sum.x * inv_w, sum.y * inv_w, sum.z * inv_w, 0.0f);
}

__global__ __launch_bounds__(256) void k_denoise(DCamera cam, uint32_t pix_count, const float4* color, const float4* nd,
                                                 const float4* pos, float4* out, int step_width, DDenoise prm)
{
  // 16x16 pixel tiles: neighbouring threads share most of their (dilated) taps in L1/L2
  const int W = (int)cam.width, H = (int)cam.height;
  const uint32_t tiles_x = ((uint32_t)W + 15u) / 16u;
  const int x0 = (int)(blockIdx.x % tiles_x) * 16, y0 = (int)(blockIdx.x / tiles_x) * 16;
  const int x = x0 + (int)(threadIdx.x & 15u), y = y0 + (int)(threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const int reach = 2 * step_width;
  const bool interior = x0 - reach >= 0 && y0 - reach >= 0 && x0 + 15 + reach < W && y0 + 15 + reach < H;
  if (interior) denoise_pixel<true>(cam, pix_count, color, nd, pos, out, step_width, prm, x, y);
  else denoise_pixel<false>(cam, pix_count, color, nd, pos, out, step_width, prm, x, y);
}

// The same pass with its taps staged in LDS (the default).  The taps of a pixel sit `step` apart: vertically a
// workgroup works on ONE residue class of rows (y mod step): four lattice rows of outputs need eight lattice rows of
// taps, whatever the step; horizontally it is dense (64 consecutive pixels of outputs, 2 * step more on either side),
// so every global load is a coalesced run of pixels and every byte of a fetched cache line is used.  36 bytes per
// staged pixel (colour, normal, position): (64 + 4 step) x 8 of them, 28 KB at step 8.  Every tap then is three LDS
// reads instead of three 16-byte global loads through L1: the pass was bound by the L1 / texture-address rate of its 75
// loads per pixel (110 us at 1080p).  (First attempt, measured: sub-lattices in BOTH directions -- 16x16 outputs from
// 20x20 staged points at any step -- fetch one cache line per point and array at step 8, and the step x step
// workgroups that share those lines are dealt round-robin to the eight XCDs, each with its own L2: 171 us for that
// pass.)  The reference's clamp of a tap coordinate to [0, W] x [0, H] (inclusive: column W aliases the next row, row
// H is out of bounds; see k_denoise_positions) depends only on the tap's coordinate, not on which output uses it, so it
// is applied once, when the pixel is staged.
// kStep is a template parameter (the steps of a denoise call are 1, 2, 4, ...): tap offsets become immediates of the
// LDS reads -- with a run-time step every tap cost three address additions.
constexpr int kDenW = 64, kDenRows = 4, kDenHalo = 2;
template <int kStep>
__global__ __launch_bounds__(256) void k_denoise_lds(DCamera cam, uint32_t pix_count, const float4* color, const float4* nd,
                                                     const float4* pos, float4* out, DDenoise prm)
{
#pragma clang fp contract(fast)
  extern __shared__ float4 s_dyn[];
  constexpr int step = kStep;
  const int W = (int)cam.width, H = (int)cam.height;
  constexpr int row_len = kDenW + 2 * kDenHalo * step, rows = kDenRows + 2 * kDenHalo, points = row_len * rows;
  float4* s_a = s_dyn;                                       // colour.rgb, normal.x
  float4* s_b = s_dyn + points;                              // normal.yz, position.xy
  float* s_c = reinterpret_cast<float*>(s_dyn + 2 * points);  // position.z
  const uint32_t tiles_x = ((uint32_t)W + kDenW - 1u) / kDenW;
  const uint32_t lattice_rows = ((uint32_t)H + (uint32_t)step - 1u) / (uint32_t)step;
  const uint32_t tiles_y = (lattice_rows + kDenRows - 1u) / kDenRows;
  // Workgroups that share staged rows (vertical neighbours of one residue class) must share an L2: workgroup b runs
  // on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch), so every XCD gets one contiguous eighth of the tiles in
  // (residue, tile row, tile column) order -- a workgroup's vertical neighbour is 30 workgroups away on the same XCD,
  // about 1 MB of input apart, well inside its 4 MB L2.  Dealt round-robin instead, neighbours land on different
  // XCDs, every halo row is fetched from the Infinity Cache again, and the pass is bound there (2-3x the bytes).
  const uint32_t total = tiles_x * tiles_y * (uint32_t)step, per_xcd = (total + 7u) / 8u;
  uint32_t b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || b >= total) return;
  const int tx = (int)(b % tiles_x);
  b /= tiles_x;
  const int ty = (int)(b % tiles_y), ry = (int)(b / tiles_y);
  const int x0 = tx * kDenW - kDenHalo * step;  // first staged column
  // stage (coordinates may lie outside the image: the reference's clamp).  Three rounds of 256 pixels at a time with
  // all their global loads issued before the first is used: staged one round after the other, the dependent round
  // trips (about 2 us each) were most of a workgroup's life and the pass ran at 65-70 us whatever the step
  constexpr int kRounds = 3;
  for (int base = 0; base < points; base += 256 * kRounds) {
    float4 c[kRounds], g[kRounds], q[kRounds];
    int uu[kRounds], vv[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int k = min(base + r * 256 + (int)threadIdx.x, points - 1);
      const int li = k % row_len, lj = k / row_len;
      int u = x0 + li;
      int v = (ty * kDenRows + lj - kDenHalo) * step + ry;
      u = u < 0 ? 0 : (u > W ? W : u);
      v = v < 0 ? 0 : (v > H ? H : v);
      uint32_t ti = (uint32_t)u + (uint32_t)v * (uint32_t)W;
      if (ti >= pix_count) ti = pix_count - 1u;
      uu[r] = u;
      vv[r] = v;
      c[r] = color[ti];
      g[r] = nd[ti];
      q[r] = pos[ti];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int k = base + r * 256 + (int)threadIdx.x;
      f3 p = xyz(q[r]);
      if (uu[r] == W || vv[r] == H) {  // the reference's off-by-one taps keep their own view ray
        f3 to, td;
        generate_ray(cam, (float)uu[r] + 0.5f, (float)vv[r] + 0.5f, to, td);
        p = to + td * g[r].w;
      }
      if (k < points) {
        s_a[k] = make_float4(c[r].x, c[r].y, c[r].z, g[r].x);
        s_b[k] = make_float4(g[r].y, g[r].z, p.x, p.y);
        s_c[k] = p.z;
      }
    }
  }
  __syncthreads();
  const int i = (int)(threadIdx.x & 63u), j = (int)(threadIdx.x >> 6);
  const int x = tx * kDenW + i, y = (ty * kDenRows + j) * step + ry;
  if (x >= W || y >= H) return;
  const int centre = (j + kDenHalo) * row_len + i + kDenHalo * step;
  const float4 ca = s_a[centre], cb = s_b[centre];
  const f3 cval = mk3(ca.x, ca.y, ca.z), nval = mk3(ca.w, cb.x, cb.y), pval = mk3(cb.z, cb.w, s_c[centre]);
  const float kernel[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
  const float step2 = (float)(step * step);
  constexpr float kLog2e = 1.4426950408889634f;
  const float kc = -kLog2e / prm.c_phi, kn = -kLog2e / (step2 * prm.n_phi), kp = -kLog2e / prm.p_phi;
  f3 sum = mk3(0.f, 0.f, 0.f);
  float cum_w = 0.0f;
  // one tap row per iteration (not unrolled: the fully unrolled 5x5 keeps 130 registers alive -- three wavefronts per
  // SIMD); the tap weight kernel[min(|dx|, |dy|)] of a row depends on |dx| only through three row constants
#pragma unroll 1
  for (int dy = -2; dy <= 2; ++dy) {
    const int ady = dy < 0 ? -dy : dy;
    const float w_by_adx[3] = {kernel[0], kernel[ady < 1 ? ady : 1], kernel[ady]};
    const int row = centre + dy * row_len;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int t = row + dx * step;
      const float4 ta = s_a[t], tb = s_b[t];
      const float tz = s_c[t];
      // (the squared distances written out here, inside the contraction pragma's scope: dot() from pt_math.hpp is
      // compiled under the file's -ffp-contract=off and kept the pass at 37 instead of 27 instructions per tap)
      const float cx = cval.x - ta.x, cy = cval.y - ta.y, cz = cval.z - ta.z;
      const float nx = nval.x - ta.w, ny = nval.y - tb.x, nz = nval.z - tb.y;
      const float px = pval.x - tb.z, py = pval.y - tb.w, pz = pval.z - tz;
      const float dc = cx * cx + cy * cy + cz * cz, dn = nx * nx + ny * ny + nz * nz, dp = px * px + py * py + pz * pz;
      const float arg = dc * kc + dn * kn + dp * kp;
      const float weight = __builtin_amdgcn_exp2f(arg);
      const float wk = weight * w_by_adx[dx < 0 ? -dx : dx];
      sum.x += ta.x * wk;
      sum.y += ta.y * wk;
      sum.z += ta.z * wk;
      cum_w += wk;
    }
  }
  const float inv_w = 1.0f / cum_w;
  out[(uint32_t)x + (uint32_t)y * (uint32_t)W] = make_float4(sum.x * inv_w, sum.y * inv_w, sum.z * inv_w, 0.0f);
}

__global__ void k_selftest(const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt, float* out_sin,
                           float* out_cos)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_div[i] = a[i] / b[i];
  out_sqrt[i] = ieee_sqrt(a[i]);
  float s, c;
  det_sincos(a[i], s, c);
  out_sin[i] = s;
  out_cos[i] = c;
}


// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

void launch_preview(hipStream_t s, const float4* buf, uint32_t pix_count, int mode, uint32_t* rgba)
{
  hipLaunchKernelGGL(k_preview, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, mode, rgba);
}
void launch_gather_bands(hipStream_t s, const DGatherBands& bands, uint32_t count, uint32_t max_pix, int channels,
                         uint32_t frame_pixels, float* frame)
{
  const uint64_t floats = (uint64_t)max_pix * (uint32_t)channels;
  hipLaunchKernelGGL(k_gather_bands, dim3((uint32_t)((floats + 255u) / 256u), count), dim3(256), 0, s, bands, channels,
                     frame_pixels, frame);
}
void launch_preview_packed(hipStream_t s, const float* buf, uint32_t pix_count, int channels, int mode, uint32_t* rgba)
{
  hipLaunchKernelGGL(k_preview_packed, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, channels, mode, rgba);
}
void launch_pack(hipStream_t s, const float4* buf, uint32_t pix_count, int which, float* dst)
{
  hipLaunchKernelGGL(k_pack, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, which, dst);
}
void launch_denoise_positions(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* nd, float4* pos)
{
  hipLaunchKernelGGL(k_denoise_positions, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, cam, pix_count, nd, pos);
}
void launch_denoise_pass(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* color, const float4* nd,
                         const float4* pos, float4* out, int step_width, DDenoise params)
{
  // (beyond step 32 the staged tile outgrows 64 KB of LDS: such filter sizes take the L1 / L2 kernel, and so does a
  // step that is no power of two -- ptc_denoise only issues 1, 2, 4, ...)
  if (params.variant == 1 || step_width > 32 || (step_width & (step_width - 1)) != 0) {  // taps through L1 / L2 (cross-check of the default)
    const uint32_t tiles = div_up(cam.width, 16u) * div_up(cam.height, 16u);
    hipLaunchKernelGGL(k_denoise, dim3(tiles), dim3(256), 0, s, cam, pix_count, color, nd, pos, out, step_width, params);
    return;
  }
  const uint32_t st = (uint32_t)step_width;
  const uint32_t total = div_up(cam.width, (uint32_t)kDenW) * div_up(div_up(cam.height, st), (uint32_t)kDenRows) * st;
  const dim3 grid(div_up(total, 8u) * 8u), block(256);  // one contiguous eighth of the tiles per XCD (see the kernel)
  const size_t lds = (size_t)(kDenW + 2 * kDenHalo * step_width) * (size_t)(kDenRows + 2 * kDenHalo) * 36u;
  switch (step_width) {
  case 1: hipLaunchKernelGGL(k_denoise_lds<1>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 2: hipLaunchKernelGGL(k_denoise_lds<2>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 4: hipLaunchKernelGGL(k_denoise_lds<4>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 8: hipLaunchKernelGGL(k_denoise_lds<8>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 16: hipLaunchKernelGGL(k_denoise_lds<16>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  default: hipLaunchKernelGGL(k_denoise_lds<32>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  }
}
void launch_selftest(hipStream_t s, const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt,
                     float* out_sin, float* out_cos)
{
  hipLaunchKernelGGL(k_selftest, dim3(div_up(n, 256u)), dim3(256), 0, s, a, b, n, out_div, out_sqrt, out_sin, out_cos);
}

}  // namespace pt
