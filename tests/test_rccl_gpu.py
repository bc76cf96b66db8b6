"""The collective path on hardware: RCCL (torch.distributed backend "nccl") with ONE rank.

It proves nothing about scaling - a sum over one rank is the identity - but it executes, on the
MI355X, everything the multi-GPU run goes through before an 8-GPU node is there to run it:
process-group creation on the device, parameter broadcast, the flat-gradient all-reduce after a
real backward, the device barrier and the max-over-ranks reduction
(reference: configs/trainer/ddp.yaml:8-13, one process per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from superpoint_transformer_amd import hotpath, parallel
from superpoint_transformer_amd.synthetic import make_nag
nag = make_nag("R", seed=5, device=dev, sizes=(20000, 600, 250, 9000, 7000, 2))
path = hotpath.SPTTrainStep(nag, dev, world=1)
assert path.bucket.always
loss = path.step()                       # forward + loss + backward + all-reduce + AdamW
torch.cuda.synchronize()
assert torch.isfinite(loss).item() and path.bucket.check_views()
before = path.bucket.flat.clone()
after = path.bucket.reduce().clone()     # one more collective on the filled buffer
torch.cuda.synchronize()
assert torch.equal(before, after) and before.abs().sum().item() > 0
parallel.broadcast_parameters(path.params, src=0)
t = parallel.max_over_ranks(1.25, dev)
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RCCL_OK", t)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", SPT_FORCE_COLLECTIVES="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.gpu
def test_train_step_and_flat_all_reduce_through_rccl():
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL_OK 1.25" in r.stdout


@pytest.mark.gpu
def test_bench_line_under_a_torchrun_style_environment_with_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup",
                        "1", "--settle", "0", "--scene", "R", "--no-cpu-baseline",
                        "--no-preprocess", "--no-f32-exact"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # (RCCL prints its version banner on stdout too: pick the JSON line)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
