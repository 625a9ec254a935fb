"""Row-band partition of one frame over several GPUs (one process per GPU, torch.distributed / RCCL).

The path shards by pixels: the scene is replicated, each rank traces a contiguous band of image rows with its
own `PathTracer` (`set_rows`), and the only inter-GPU traffic is the gather of the bands' radiance at present
time.  Two slot-numbering modes:

  "local"   (default, used by bench.py) every band numbers its compacted paths from 0 at each bounce.  No
            collective while tracing.  The material RNG is keyed on the slot index (path_tracer.cu:297-301), so
            the noise pattern differs from the single-GPU image (same distribution).
  "global"  reproduces the single-GPU image bit for bit: before bounce b every rank learns the live-path
            counts of all ranks (one all_gather of one uint32 per rank) and offsets its slots by the sum over
            the lower bands -- exactly the global compacted index the reference would have used.
"""
import numpy as np


def split_rows(height, world):
    """Contiguous row ranges [(row_begin, row_end)] for `world` ranks; sizes differ by at most one row."""
    edges = [(height * r) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1]) for r in range(world)]


def slot_base_from_counts(counts, rank):
    """Global slot index of a band's first live path = live paths of all lower bands (stable order)."""
    return int(np.sum(np.asarray(counts[:rank], dtype=np.uint64)))


def interleaved_rows(height, world, block_rows):
    """For each rank the list of global row indices it renders under ptc_set_interleave, in local order."""
    blocks = (height + block_rows - 1) // block_rows
    return [[y for gb in range(r, blocks, world) for y in range(gb * block_rows, min(height, (gb + 1) * block_rows))]
            for r in range(world)]


def assemble_interleaved(bands, height, world, block_rows):
    """Scatter per-rank packed rows back into the full frame."""
    rows = interleaved_rows(height, world, block_rows)
    out = np.empty((height,) + tuple(bands[0].shape[1:]), dtype=bands[0].dtype)
    for r in range(world):
        out[rows[r]] = bands[r]
    return out


def assemble(bands):
    """Stack per-band arrays (rows first) into the full frame, in rank order."""
    return np.concatenate(list(bands), axis=0)


class BandRenderer:
    """One rank's band of the frame.  `dist` is torch.distributed (initialised) or None for a single rank.

    mode "global", exchange "device": the live count stays on the GPU -- ptc_copy_live_count into a torch tensor,
    all_gather over RCCL on torch's current stream, exclusive prefix, ptc_trace_bounce reads it from device memory; the
    library orders its frame stream against torch's stream with events (no host synchronisation; needs the context on
    torch's current stream, which this class arranges).  exchange "host": the count is read back and gathered as a
    Python object (any backend, e.g. gloo on a box without RCCL peers) -- the reference's own per-bounce host
    round trip (path_tracer.cu:457)."""

    def __init__(self, path_tracer, width, height, rank=0, world=1, dist=None, mode="local", exchange="device"):
        assert mode in ("local", "global") and exchange in ("device", "host")
        self.pt, self.width, self.height = path_tracer, width, height
        self.rank, self.world, self.dist, self.mode, self.exchange = rank, world, dist, mode, exchange
        self.rows = split_rows(height, world)[rank]
        if world > 1:
            self.pt.set_rows(*self.rows)
        self._torch = None
        if mode == "global" and world > 1:
            import torch
            self._torch = torch
            self.pt.set_stream(torch.cuda.current_stream().cuda_stream)
            self._mine = torch.zeros(1, dtype=torch.int32, device="cuda")
            self._all = torch.zeros(world, dtype=torch.int32, device="cuda")
            self._base = torch.zeros(1, dtype=torch.int32, device="cuda")

    def trace(self, camera):
        """One iteration of this band."""
        if self.mode == "local" or self.world == 1:
            self.pt.path_trace(camera)
            return
        torch = self._torch
        self.pt.trace_begin(camera)
        for b in range(self.pt.max_bounces):
            # live count of this band entering bounce b -> all ranks -> exclusive prefix = slot base
            if self.exchange == "device":
                self.pt.copy_live_count(b, self._mine.data_ptr())
                self.dist.all_gather_into_tensor(self._all, self._mine)
                self._base.copy_(self._all[: self.rank].sum().to(torch.int32).reshape(1) if self.rank
                                 else torch.zeros(1, dtype=torch.int32, device="cuda"))
            else:
                counts = [None] * self.world
                self.dist.all_gather_object(counts, self.pt.read_live_count(b))
                self._base.copy_(torch.tensor([slot_base_from_counts(counts, self.rank)], dtype=torch.int32))
            self.pt.trace_bounce(b, self._base.data_ptr())
        self.pt.trace_end()


class BandGather:
    """Present-time gather of the ranks' rows on rank 0 through the library (ptc_band_* / ptc_gather_frame: HIP
    inter-process memory, device-to-device copies into the root; xGMI between GPUs).  `dist` carries the handles
    (once) and the "rows are published" barrier (every present)."""

    def __init__(self, path_tracer, rank, world, dist, group=None):
        """group: the process group the handles and barriers travel over (None = the default one; bench.py passes its gloo
        control group so that this gather still works where RCCL does not)"""
        self.pt, self.rank, self.world, self.dist, self.group = path_tracer, rank, world, dist, group
        if world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, path_tracer.band_export() if rank else b"", group=group)
            if rank == 0:
                for r in range(1, world):
                    path_tracer.band_import(r, handles[r])

    def gather(self, which="color", dev_ptr=None):
        """collective; returns the frame on rank 0 (None elsewhere, or when dev_ptr is given)"""
        if self.world > 1:
            if self.rank:
                self.pt.band_publish(which)
            self.dist.barrier(group=self.group)
        out = self.pt.gather_frame(which, dev_ptr) if self.rank == 0 else None
        if self.world > 1:
            self.dist.barrier(group=self.group)   # nobody overwrites its exported rows before the root has pulled them
        return out
