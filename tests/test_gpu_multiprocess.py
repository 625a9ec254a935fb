"""The N > 1 path with real processes on the one GPU of the test box: two ranks share cuda:0 and talk over gloo
(RCCL refuses two ranks on one device), exchange live counts per bounce (global slot numbering: the assembled frame
must be the single-context frame bit for bit) and move their rows to rank 0 through the library's inter-process
gather (ptc_band_export / import / publish, ptc_gather_frame, ptc_gather_present_rgba8)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_band_processes_share_one_gpu(world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29620 + world), os.path.join(ROOT, "tests", "mp_band_worker.py")]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert proc.returncode == 0 and "BAND_OK" in proc.stdout, proc.stdout[-4000:]


def test_cli_with_ranks(pkg, tmp_path):
    """hip_pt --gpus 3 (three forked ranks on the one device, rows in blocks of 8, the library's gather at the end)
    against the Python path doing the same split in one process: the same image, byte for byte."""
    import numpy as np
    from PIL import Image
    hip_pt = os.path.join(ROOT, "cuda-path-tracer_amd", "host", "hip_pt")
    if not os.path.exists(hip_pt):
        subprocess.run(["make"], cwd=os.path.dirname(hip_pt), check=True, stdout=subprocess.DEVNULL)
    scene_file, world, spp, mb = "cornell_mesh.json", 3, 3, 6
    out = tmp_path / "out.png"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([hip_pt, "scenes/" + scene_file, "-o", str(out), "--spp", str(spp), "--max-bounces", str(mb), "--gpus", str(world)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"ranks: {world}" in r.stdout and "Gather + write image file" in r.stdout
    got = np.array(Image.open(out))
    scene = pkg.json_parser.scene_from_json(os.path.join(ROOT, "assets", "scenes", scene_file))
    w, h = scene.resolution
    bands = []
    for rank in range(world):
        with pkg.PathTracer(max_bounces=mb) as pt:
            pt.create_buffers(scene.resolution, scene)
            pt.set_interleave(rank, world, 8)
            pt.set_param("slot_offset", rank * w * h)
            pt.max_iterations = spp
            for _ in range(spp):
                pt.path_trace(scene.camera)
            bands.append(pt.send_to_preview())
    want = pkg.bands.assemble_interleaved(bands, h, world, 8)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_band_handle_geometry_is_checked_on_import(pkg):
    """A band handle arrives from another process and decides where the root's gather kernel writes: a stale or
    corrupt one (the peer resized or re-partitioned after exporting) must be refused before anything is mapped
    (round-2 advisor finding: only width and pix_count <= W*H were checked)."""
    import copy
    capi = pkg._capi
    scene = pkg.scenes.cornell_spheres((64, 48))
    flat = scene.build_scene()
    with pkg.PathTracer() as root, pkg.PathTracer() as peer:
        root.create_buffers((64, 48), flat)
        peer.create_buffers((64, 48), flat)
        peer.set_interleave(1, 2, 8)
        good = capi.ptc_band_handle.from_buffer_copy(peer.band_export())
        assert (good.rank, good.nranks, good.block_rows, good.pix_count) == (1, 2, 8, 24 * 64)

        def tampered(**kw):
            h = copy.copy(good)
            for k, v in kw.items():
                setattr(h, k, v)
            return bytes(h)

        for bad in (dict(block_rows=0), dict(rank=2), dict(rank=5, nranks=3), dict(pix_count=25 * 64),
                    dict(block_rows=16), dict(nranks=3), dict(width=32), dict(pix_count=0),
                    dict(nranks=1, pix_begin=64 * 48 - 10), dict(nranks=0, pix_count=64 * 48 + 1)):
            with pytest.raises(pkg.PtcError) as e:
                root.band_import(1, tampered(**bad))
            assert e.value.code == capi.PTC_ERR_INVALID, bad


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5` with no launcher around it: bench.py
    spawns its two ranks itself (a child torch.distributed.run, before the parent touches the GPU) and rank 0 prints
    ONE line with n_gpus 2, the gather time on its own, and both ranks' rays.  Without --rehearse-on-one-gpu the same
    request on this one-GPU box must fail loudly instead of measuring one GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "20", "--warmup", "5",
           "--no-extras"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0
    assert d["rccl_ranks"] == 0 and "gloo" in d["gather"]["backend"]          # a rehearsal says so
    assert d["gather"]["rccl_us"] > 0 and d["gather"]["ipc_us"] > 0, d["gather"]
    assert d["gather"]["transports_agree"] is True      # the RCCL-path frame and the library's gathered frame are the same bits
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["rays"] > 0 for r in d["ranks"])
    assert abs(sum(r["rays"] for r in d["ranks"]) - d["config"]["rays_per_step"] * 20) <= 20
    import torch
    if torch.cuda.device_count() < 2:
        proc = subprocess.run(cmd[:4] + cmd[5:], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert proc.returncode != 0 and '"metric"' not in proc.stdout and "--rehearse-on-one-gpu" in proc.stderr
