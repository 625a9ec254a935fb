"""Worker of tests/test_gpu_multiprocess.py: one of N processes that share cuda:0 (torch.distributed over gloo, started
by torch.distributed.run).  Renders a row band with GLOBAL slot numbering (live counts exchanged per bounce) and sends
its rows to rank 0 through the library's inter-process gather; rank 0 compares the assembled frame with a
single-context render of the same scene, bit for bit.  Then the interleaved split with per-rank numbering (what bench.py
--gpus N runs): the frame the library gathers on rank 0 must be, bit for bit, the CPU oracle's rendering of every rank's
rows (orc_render_streaming_interleaved) put into frame order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo")
    pkg = graft.load_package()
    W, H, MB, ITERS = 160, 96, 6, 3
    glm = pkg.glmlite
    scene = pkg.scenes.cornell_bunny((W, H), n_lat=20, n_lon=40)
    mesh = list(scene.mesh_map_.values())[0]
    scene.add_object(mesh, glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.7, 0.4, 0.9)),
                                        glm.translate((0.1, 0.2, 0.5))]), "glass")
    flat = scene.build_scene()

    # --- contiguous bands, global slot numbering: the single-GPU image, bit for bit
    pt = pkg.PathTracer(device=0, max_bounces=MB)
    pt.set_param("frames_in_flight", 1)
    pt.create_buffers((W, H), flat)
    pt.max_iterations = ITERS
    band = pkg.bands.BandRenderer(pt, W, H, rank, world, dist, mode="global", exchange="host")
    for _ in range(ITERS):
        band.trace(scene.camera)
    gather = pkg.bands.BandGather(pt, rank, world, dist)
    frames = {k: gather.gather(k) for k in ("color", "normal", "depth")}
    rgba = None
    if world > 1:
        if rank:
            pt.band_publish("color")
        dist.barrier()
        if rank == 0:
            rgba = pt.gather_present()
        dist.barrier()
    rays = [None] * world
    dist.all_gather_object(rays, pt.stats()["rays_total"])
    ok = True
    if rank == 0:
        with pkg.PathTracer(device=0, max_bounces=MB) as ref:
            ref.set_param("frames_in_flight", 1)
            ref.create_buffers((W, H), flat)
            ref.max_iterations = ITERS
            for _ in range(ITERS):
                ref.path_trace(scene.camera)
            want = {k: ref.download(k) for k in ("color", "normal", "depth")}
            want_rgba = ref.send_to_preview()
            want_rays = ref.stats()["rays_total"]
        for k in want:
            if not np.array_equal(frames[k], want[k]):
                print(f"MISMATCH global {k}: {int(np.sum(frames[k] != want[k]))} values differ", flush=True)
                ok = False
        if rgba is not None and not np.array_equal(rgba, want_rgba):
            print("MISMATCH gathered rgba", flush=True)
            ok = False
        if sum(rays) != want_rays:
            print("MISMATCH rays", sum(rays), want_rays, flush=True)
            ok = False
    pt.close()

    # --- interleaved row blocks, local numbering with a per-rank slot offset (what bench.py and hip_pt --gpus run):
    #     every rank's rows bit-identical to the oracle's rendering of that rank, through the library's gather; the
    #     first-hit G-buffer is also the single-GPU one exactly
    pt = pkg.PathTracer(device=0, max_bounces=MB)
    pt.set_param("frames_in_flight", 4)
    pt.set_param("batch_frames", 2)
    pt.create_buffers((W, H), flat)
    pt.max_iterations = ITERS
    if world > 1:
        pt.set_interleave(rank, world, 8)
        pt.set_param("slot_offset", rank * W * H)
    for _ in range(ITERS):
        pt.path_trace(scene.camera)
    gather = pkg.bands.BandGather(pt, rank, world, dist)
    got = {k: gather.gather(k) for k in ("color", "normal", "depth")}
    if rank == 0:
        if not (np.array_equal(got["normal"], want["normal"]) and np.array_equal(got["depth"], want["depth"])):
            print("MISMATCH interleaved G-buffer", flush=True)
            ok = False
        orc = graft.load_oracle()
        sh = orc.SceneHandle(flat)
        parts = [orc.render_interleaved(flat, scene.camera, W, H, r, world, 8, r * W * H if world > 1 else 0, 0, ITERS, MB,
                                        scene_handle=sh) for r in range(world)]
        for k in ("color", "normal", "depth"):
            oracle_frame = pkg.bands.assemble_interleaved([p[k] for p in parts], H, world, 8)
            if not np.array_equal(got[k], oracle_frame):
                print(f"MISMATCH interleaved {k} vs oracle: {int(np.sum(got[k] != oracle_frame))} values differ", flush=True)
                ok = False
    pt.close()
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    if rank == 0:
        print("BAND_OK" if flag[0] else "BAND_FAILED", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
