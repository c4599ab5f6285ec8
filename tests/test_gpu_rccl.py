"""RCCL on the hardware at hand (BASELINE config 5's exchange step).

The production multi-GPU path is: az_propose_launch -> az_propose_stage_result_dev (device-to-device copy of the
search's result record into the send buffer, on the ctx stream) -> az_propose_fetch -> ONE
all_gather_into_tensor per batch over RCCL (aznet_hip.dist.DeviceGather).  A one-rank "nccl" process group still
builds an RCCL communicator and runs the collective on the GPU, so the whole path -- cross-stream ordering,
padding rows, the restaging after a fallback rerun (err bits 8 and 32) -- runs here on ONE MI355X; the two-rank
case runs when the box has two GPUs.  Workers are fresh processes (tests/rccl_worker.py); nothing is re-exec'ed
after GPU initialisation.  `python bench.py --gpus 1 --launcher` takes the same launcher plumbing
(torch.distributed.run, one rank) the driver uses for N > 1."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _run_world(world, n_images, rows, native=False, lanes=1):
    port = _free_port()
    env = _env()
    env["AZ_TEST_NATIVE_GATHER"] = "1" if native else "0"
    env["AZ_TEST_LANES"] = str(lanes)
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "rccl_worker.py"), str(r), str(world),
                               str(port), str(n_images), str(rows)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-4000:])
        assert "RCCL_WORKER_OK %d %d" % (r, n_images) in out, out[-2000:]


@pytest.mark.parametrize("n_images,rows", [(7, 3), (2, 4)])
def test_rccl_gather_world_1(n_images, rows):
    """ONE rank, backend "nccl": records staged device-to-device, the all-gather forced, padding rows (batches
    shorter than the send buffer), reruns with restaging; every gathered image equals a plain az_propose."""
    _run_world(1, n_images, rows)


@pytest.mark.parametrize("lanes", [1, 2])
def test_native_rccl_gather_world_1(lanes):
    """The same exchange with the library's own ncclAllGather on the ctx stream (az_rccl_init / az_gather_records, RCCL
    bound at run time to the librccl.so the process already holds) instead of torch.distributed's: a one-rank
    communicator, padding rows, reruns with restaging, records staged from both lanes; every gathered image equals a
    plain az_propose."""
    _run_world(1, 7, 3, native=True, lanes=lanes)


@pytest.mark.parametrize("native", [False, True])
def test_rccl_gather_world_2(native):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    _run_world(2, 9, 2, native=native, lanes=2)


def test_bench_through_the_launcher_one_rank():
    """bench.py under torch.distributed.run with ONE rank (--launcher): rendezvous on 127.0.0.1, "nccl" process
    group, the image shard + RCCL gather inside the timed loop, one JSON line relayed by the parent."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--launcher", "--steps", "16", "--warmup", "4",
           "--no-cpu-baseline", "--no-e2e", "--no-pipelined", "--no-fast", "--no-calibrated", "--no-extras"]
    p = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 16 and rec["value"] > 0
    assert rec["config"]["gather"].startswith("RCCL all_gather"), rec["config"]["gather"]
    assert rec["rccl"]["backend"] == "nccl" and rec["rccl"]["collectives"] >= 2
