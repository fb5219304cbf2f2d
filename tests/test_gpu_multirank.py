"""N > 1 plumbing on the REAL kernels (one MI355X is enough): bench.py launched with a plain `--gpus 2` must start two
ranks by itself and report n_gpus == 2, and fullsubnet_plus_amd.dist.forward_sharded over two gloo ranks that both drive
GPU 0 must reproduce the single-process HIP forward in both batch modes.  The RCCL ("nccl") backend needs one GPU per
rank, so here the collectives run on gloo; the data path (no collective in it) is the production one."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_plain_gpus2_self_launches_two_ranks():
    res = _run([sys.executable, "bench.py", "--gpus", "2", "--same-device", "--dist-backend", "gloo", "--steps", "2",
                "--warmup", "1", "--batch", "4", "--cpu-budget-s", "2", "--strong"])
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    # round 6: the N > 1 line carries the CPU baseline (rank 0), the box object with one probe per rank, the median-of-three runs and
    # the strong-scaling leg (one global batch of 8 on rank 0: scatter + forward + gather)
    assert r["cpu_baseline"]["value"] > 0 and r["cirm_rel_err"] < 1e-3
    assert len(r["ms_per_step_runs"]) == 3 and min(r["ms_per_step_runs"]) <= r["ms_per_step"] <= max(r["ms_per_step_runs"])
    assert len(r["box"]["per_rank"]) == 2 and r["box"]["mfma_peak_tflops"] > 50
    assert r["strong"]["global_batch"] == 8 and r["strong"]["value"] > 0
    assert r["n_gpus"] == 2 and r["dist"]["world_size_seen"] == 2
    assert r["config"]["global_batch"] == 8 and r["scaling"] == "weak"
    assert r["dist"]["per_rank_ms_per_step"]["max"] >= r["dist"]["per_rank_ms_per_step"]["min"] > 0
    assert abs(r["value"] - 8 * r["config"]["frames_per_clip"] * r["steps"] / (r["ms_per_step"] * r["steps"] * 1e-3)) < 1e-6 * r["value"]
    assert r["gather_ms"] > 0


def test_bench_eight_ranks_on_one_device():
    """eight processes loading the library, creating handles, packing weights and running at once had never happened.
    `bench.py --gpus 8 --same-device`: eight ranks share GPU 0 (gloo for the collectives: RCCL wants a GPU per rank; the exchange-free
    kernel, because eight processes' co-resident launches could hold parts of one chip against each other) - what remains is exactly
    the N = 8 plumbing: self-launch, rendezvous, per-rank inputs, barrier-bracketed timing, max over ranks, the mask gather, and the
    roofline block of the slowest rank."""
    res = _run([sys.executable, "bench.py", "--gpus", "8", "--same-device", "--dist-backend", "gloo", "--steps", "2",
                "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-alt", "--strong", "--repeats", "1", "--probe-ms", "5"], timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    # round 6: the same launch also times BASELINE.json configs[2] as worded - ONE global batch on rank 0, scatter + forward + gather - next
    # to the weak-scaling headline, and every rank reports what its box held (here: eight probes of the one device)
    st = r["strong"]
    assert st["global_batch"] == 16 and st["ms_per_step"] > 0 and st["value"] > 0 and st["via_host_memory"] is True
    assert st["scatter_ms"] > 0 and st["gather_ms"] > 0 and st["forward_ms_slowest_rank"] > 0
    assert st["bytes_scattered_per_step"] == 16 * r["config"]["frames_per_clip"] * 257 * 8
    assert len(r["box"]["per_rank"]) == 8 and all(p["mfma_tflops_after"] > 0 for p in r["box"]["per_rank"])
    assert r["n_gpus"] == 8 and r["dist"]["world_size_seen"] == 8 and r["config"]["global_batch"] == 16
    assert r["dist"]["gathered_shape"][0] == 16 and 0 <= r["dist"]["roofline_of_rank"] < 8
    assert r["dist"]["per_rank_ms_per_step"]["max"] >= r["dist"]["per_rank_ms_per_step"]["min"] > 0
    assert r["roofline"]["avg_launch_ms"] > 0 and r["roofline"]["subband_plan"][0]["kernel"].startswith("lstm2_fc_kernel")


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    res = _run([sys.executable, "bench.py", "--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert res.returncode != 0
    assert "visible" in (res.stderr + res.stdout)


@pytest.mark.parametrize("mode", ["full", "parity"])
def test_forward_sharded_two_ranks_equals_single_process(mode, tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = tmp_path / "sharded.npy"
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_worker.py"), mode, str(out)])
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    from fullsubnet_plus_amd import FullSubNet_Plus
    from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m.load_state_dict(make_state_dict(0, "default"), strict=True)
    m = m.to("cuda").eval()
    m.batch_mode = mode
    ins = [t.cuda() for t in make_inputs(5, 0.5, 77)]
    want = m(*ins).cpu().numpy()
    assert got.shape == want.shape == ((5, 2, 257, want.shape[-1]) if mode == "full" else (5, 2, 128, want.shape[-1]))
    # shards of 3 + 2 utterances run other sub-band kernel plans than the batch of 5: same rows, other summation order
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()


# ---- RCCL itself (SURVEY.md 8(e) caveat: with one GPU per box, validate the "nccl" path at world_size = 1)
def test_bench_under_torchrun_initialises_rccl_at_world_size_1():
    """The driver's multi-GPU launch line with N = 1: `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`.
    A real init_process_group("nccl", device_id=...), barrier, all_gather of the per-rank times and all_gather_into_tensor of
    the [32, 2, 257, 126] masks over RCCL; `bench.py --gpus 8` differs from this run only in N."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "1", "--dist-backend", "nccl", "--steps", "3", "--warmup", "1",
                "--no-cpu-baseline", "--no-alt"])
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["dist"]["world_size_seen"] == 1 and r["dist"]["backend"].startswith("nccl")
    assert r["dist"]["gathered_shape"] == [32, 2, 257, 126] and r["gather_ms"] > 0
    assert r["config"]["global_batch"] == 32 and r["value"] > 0


@pytest.mark.parametrize("mode", ["full", "parity"])
def test_forward_sharded_over_rccl_world_size_1(mode, tmp_path):
    """forward_sharded(gather=True) on the nccl (= RCCL) backend: all_gather_into_tensor in "full" mode, the all_gather of row
    blocks + index scatter in "parity" mode - bit-identical to the plain forward (one rank = the same plan)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = tmp_path / "rccl.npy"
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(ROOT, "tests", "multirank_worker.py"), mode, str(out), "nccl"])
    assert res.returncode == 0, res.stderr[-3000:]
    assert open(str(out) + ".info").read().split() == ["nccl", "1"]
    got = np.load(out)
    from fullsubnet_plus_amd import FullSubNet_Plus
    from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m.load_state_dict(make_state_dict(0, "default"), strict=True)
    m = m.to("cuda").eval()
    m.batch_mode = mode
    want = m(*[t.cuda() for t in make_inputs(5, 0.5, 77)]).cpu().numpy()
    assert np.array_equal(got, want)
