"""RCCL on the GPU box.  The gpurun boxes have ONE MI355X, so the clip-parallel path can only be driven at world size 1
here: the process group is created on the nccl (= RCCL) backend, the product's collective - one all_gather_into_tensor of the edited
frames, the C4 payload of 38 MB per rank - runs on the device through RCCL, and bench.py is launched the way the driver launches it for
N > 1 (torch.distributed.run, RANK / WORLD_SIZE from the environment).  World sizes 2 and 3 are covered on gloo (tests/test_cpu_host.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instruct-video-to-video_amd")
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from insv2v.clip_parallel import shard_units, gather_frames
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl")
rank, world = dist.get_rank(), dist.get_world_size()
assert dist.get_backend() == "nccl" and world == 1
frames = torch.randn(2, 32, 3, 256, 384, device="cuda").half()          # two C4 units of this rank: 38 MB
out = torch.empty_like(frames)
dist.all_gather_into_tensor(out, frames)                                   # the collective gather_frames issues, through RCCL
torch.cuda.synchronize()
assert torch.equal(out, frames)
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
assert t.tolist() == [1.0] * 4
assert shard_units(2, rank, world) == [0, 1] and gather_frames(frames, 2) is frames
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
'''


def _torchrun(args, port, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_rccl_collective_world1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = _torchrun([str(script), PKG], 29631, 600)
    assert r.returncode == 0 and "ok 0" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_under_torchrun_world1():
    """bench.py exactly as the driver launches it for N > 1, at N = 1: process group from the environment, barrier + max over ranks
    around the timed region, the gather inside it, one JSON line from rank 0."""
    r = _torchrun(["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ddim-steps", "2"], 29632, 1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
