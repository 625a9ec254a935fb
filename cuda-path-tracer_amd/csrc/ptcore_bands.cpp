// ptcore_bands.cpp -- several GPUs: which rows a context renders (ptc_set_rows / ptc_set_interleave), the exported band buffer,
// the root's mapping of its peers' buffers and the one-kernel gather at present time.  Part of libptcore.so (ptcore_ctx.hpp).
#include "ptcore_ctx.hpp"

using namespace pt;
using namespace ptcd;

extern "C" {

int ptc_set_rows(ptc_ctx* ctx, uint32_t row_begin, uint32_t row_end)
{
  if (!ctx || !ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (row_begin >= row_end || row_end > ctx->height) return fail(ctx, PTC_ERR_INVALID, "bad row range");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  ctx->pix_begin = row_begin * ctx->width;
  ctx->pix_count = (row_end - row_begin) * ctx->width;
  ctx->band = DBand{ctx->pix_begin, ctx->width, 0u, 1u, 0u};
  return ptc_restart(ctx);
}

int ptc_set_interleave(ptc_ctx* ctx, uint32_t rank, uint32_t nranks, uint32_t block_rows)
{
  if (!ctx || !ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (nranks == 0 || rank >= nranks || block_rows == 0) return fail(ctx, PTC_ERR_INVALID, "bad interleave");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  uint32_t rows = 0;
  const uint32_t blocks = (ctx->height + block_rows - 1u) / block_rows;
  for (uint32_t gb = rank; gb < blocks; gb += nranks) rows += std::min(block_rows, ctx->height - gb * block_rows);
  if (rows == 0) return fail(ctx, PTC_ERR_INVALID, "this rank gets no rows");
  ctx->pix_begin = 0;
  ctx->pix_count = rows * ctx->width;
  ctx->band = DBand{0u, ctx->width, rank, nranks, block_rows};
  return ptc_restart(ctx);
}

// ---- several GPUs: bands over HIP inter-process memory (include/ptcore.h) -------------------------------------------
static int band_pack(ptc_ctx* ctx, int which, float* dst, size_t* floats)
{
  const float4* src = nullptr;
  int sel = 0;
  *floats = (size_t)ctx->pix_count * 3u;
  switch (which) {
  case PTC_BUF_COLOR: src = ctx->fb.color4; break;
  case PTC_BUF_NORMAL: src = ctx->fb.nd4; break;
  case PTC_BUF_DEPTH: src = ctx->fb.nd4; sel = 1; *floats = ctx->pix_count; break;
  case PTC_BUF_FINAL: src = ctx->result; break;
  default: return fail(ctx, PTC_ERR_INVALID, "unknown buffer");
  }
  if (int rc = sync_frames(ctx)) return rc;
  launch_pack(ctx->stream, src, ctx->pix_count, sel, dst);
  return check_last(ctx, "pack");
}

int ptc_band_export(ptc_ctx* ctx, ptc_band_handle* out)
{
  if (!ctx || !out) return PTC_ERR_INVALID;
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(ctx)) return rc;
  if (!ctx->band_buf) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->band_buf), (size_t)ctx->pix_capacity * 3u * sizeof(float)));
  std::memset(out, 0, sizeof *out);
  hipIpcMemHandle_t h;
  static_assert(sizeof h <= sizeof out->ipc_mem, "ipc handle size");
  HIP_TRY(ctx, hipIpcGetMemHandle(&h, ctx->band_buf));
  std::memcpy(out->ipc_mem, &h, sizeof h);
  out->pix_count = ctx->pix_count;
  out->pix_begin = ctx->band.pix_begin;
  out->width = ctx->width;
  out->rank = ctx->band.rank;
  out->nranks = ctx->band.nranks;
  out->block_rows = ctx->band.block_rows;
  return PTC_OK;
}

int ptc_band_import(ptc_ctx* root, uint32_t rank, const ptc_band_handle* handle)
{
  if (!root || !handle || rank > 0xffffu) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (handle->width != root->width) return fail(root, PTC_ERR_INVALID, "band of another frame width");
  // The handle arrives from another process: its geometry decides where k_scatter_band writes, so it must describe a
  // band of THIS frame (a peer that resized or re-partitioned after exporting sends a stale one).
  {
    const uint64_t P = (uint64_t)root->width * root->height;
    if (handle->pix_count == 0u || (uint64_t)handle->pix_count > P) return fail(root, PTC_ERR_INVALID, "band larger than the frame");
    if (handle->nranks <= 1u) {
      if ((uint64_t)handle->pix_begin + handle->pix_count > P) return fail(root, PTC_ERR_INVALID, "band reaches beyond the frame");
    } else {
      if (handle->block_rows == 0u || handle->rank >= handle->nranks) return fail(root, PTC_ERR_INVALID, "bad interleave in the band handle");
      uint64_t rows = 0;  // as ptc_set_interleave counts them
      const uint32_t blocks = (root->height + handle->block_rows - 1u) / handle->block_rows;
      for (uint32_t gb = handle->rank; gb < blocks; gb += handle->nranks)
        rows += std::min(handle->block_rows, root->height - gb * handle->block_rows);
      if (rows * root->width != handle->pix_count) return fail(root, PTC_ERR_INVALID, "band handle does not match this frame's interleave");
    }
  }
  if (int rc = bind_device(root)) return rc;
  if (root->peers.size() <= rank) root->peers.resize((size_t)rank + 1u);
  auto& peer = root->peers[rank];
  if (peer.opened && peer.mapped) (void)hipIpcCloseMemHandle(peer.mapped);
  peer = ptc_ctx::Peer{};
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle->ipc_mem, sizeof h);
  HIP_TRY(root, hipIpcOpenMemHandle(&peer.mapped, h, hipIpcMemLazyEnablePeerAccess));
  peer.opened = true;
  peer.h = *handle;
  return PTC_OK;
}

int ptc_band_publish(ptc_ctx* ctx, int which)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (!ctx->band_buf) return fail(ctx, PTC_ERR_INVALID, "ptc_band_export first");
  if (int rc = bind_device(ctx)) return rc;
  size_t floats = 0;
  if (int rc = band_pack(ctx, which, ctx->band_buf, &floats)) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the rows are in the exported buffer when this returns
  return PTC_OK;
}

static int gather_rows(ptc_ctx* root, int which, int channels)
{
  const size_t P = (size_t)root->width * root->height;
  if (!root->gather_frame) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->gather_frame), P * 3u * sizeof(float)));
  if (!root->band_buf) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->band_buf), (size_t)root->pix_capacity * 3u * sizeof(float)));
  if (!root->gather_ev[0]) {
    HIP_TRY(root, hipEventCreate(&root->gather_ev[0]));
    HIP_TRY(root, hipEventCreate(&root->gather_ev[1]));
  }
  // the root's own rows, packed like a peer's
  size_t floats = 0;
  if (int rc = band_pack(root, which, root->band_buf, &floats)) return rc;
  // One launch pulls every band -- the root's own and every imported rank's, straight out of the peers' mapped
  // buffers -- into row order: all peer -> root xGMI links carry their band at the same time, nothing is staged.
  HIP_TRY(root, hipEventRecord(root->gather_ev[0], root->stream));
  DGatherBands bands{};
  uint32_t n = 0, max_pix = 0;
  auto flush = [&]() {
    if (n) launch_gather_bands(root->stream, bands, n, max_pix, channels, (uint32_t)P, root->gather_frame);
    n = 0;
    max_pix = 0;
  };
  auto add = [&](const float* src, const DBand& band, uint32_t pix_count) {
    bands.src[n].src = src;
    bands.src[n].band = band;
    bands.src[n].pix_count = pix_count;
    max_pix = std::max(max_pix, pix_count);
    if (++n == (uint32_t)kGatherBands) flush();
  };
  add(root->band_buf, root->band, root->pix_count);
  for (size_t r = 0; r < root->peers.size(); ++r) {
    const auto& peer = root->peers[r];
    if (!peer.mapped) continue;
    add(static_cast<const float*>(peer.mapped), DBand{peer.h.pix_begin, peer.h.width, peer.h.rank, peer.h.nranks, peer.h.block_rows},
        peer.h.pix_count);
  }
  flush();
  HIP_TRY(root, hipEventRecord(root->gather_ev[1], root->stream));
  root->gather_timed = true;
  return check_last(root, "gather");
}

int ptc_gather_last_us(ptc_ctx* root, float* microseconds)
{
  if (!root || !microseconds) return PTC_ERR_INVALID;
  *microseconds = 0.0f;
  if (!root->gather_timed) return fail(root, PTC_ERR_INVALID, "no gather has run");
  if (int rc = bind_device(root)) return rc;
  float ms = 0.0f;
  HIP_TRY(root, hipEventSynchronize(root->gather_ev[1]));
  HIP_TRY(root, hipEventElapsedTime(&ms, root->gather_ev[0], root->gather_ev[1]));
  *microseconds = ms * 1e3f;
  return PTC_OK;
}

int ptc_gather_frame(ptc_ctx* root, int which, void* dst, int dst_is_device)
{
  if (!root || !dst) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (which < PTC_BUF_COLOR || which > PTC_BUF_FINAL) return fail(root, PTC_ERR_INVALID, "unknown buffer");
  if (int rc = bind_device(root)) return rc;
  const int channels = which == PTC_BUF_DEPTH ? 1 : 3;
  if (int rc = gather_rows(root, which, channels)) return rc;
  const size_t bytes = (size_t)root->width * root->height * (size_t)channels * sizeof(float);
  HIP_TRY(root, hipMemcpyAsync(dst, root->gather_frame, bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(root, hipStreamSynchronize(root->stream));
  return PTC_OK;
}

int ptc_gather_present_rgba8(ptc_ctx* root, void* dst, int dst_is_device, int display_type)
{
  if (!root || !dst) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(root)) return rc;
  int which = PTC_BUF_COLOR, mode = 0;
  switch (display_type) {
  case PTC_DISPLAY_FINAL:
  case PTC_DISPLAY_COLOR: break;
  case PTC_DISPLAY_NORMAL: which = PTC_BUF_NORMAL; mode = 1; break;
  case PTC_DISPLAY_DEPTH: which = PTC_BUF_DEPTH; mode = 2; break;
  default: return fail(root, PTC_ERR_INVALID, "unknown display type");
  }
  const int channels = which == PTC_BUF_DEPTH ? 1 : 3;
  if (int rc = gather_rows(root, which, channels)) return rc;
  const uint32_t P = root->width * root->height;
  if (!root->gather_rgba) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->gather_rgba), (size_t)P * 4u));
  uint32_t* out = dst_is_device ? static_cast<uint32_t*>(dst) : root->gather_rgba;
  launch_preview_packed(root->stream, root->gather_frame, P, channels, mode, out);
  if (int rc = check_last(root, "preview")) return rc;
  if (!dst_is_device) HIP_TRY(root, hipMemcpyAsync(dst, root->gather_rgba, (size_t)P * 4u, hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(root, hipStreamSynchronize(root->stream));
  return PTC_OK;
}

}  // extern "C"

