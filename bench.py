#!/usr/bin/env python3
"""bench.py - STFT frames/s of the FullSubNet+ forward on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one forward of the hot path over one batch of synthetic utterances per GPU
(BASELINE.json configs[1]: batch = 32 x 2 s clips, fp32, num_neighbors = 15, "full" mode = all 257
bins for every utterance).  Weak scaling: every rank processes its own 32 utterances (configs[2] =
256 utterances on 8 GPUs); utterances are independent, so there is NO collective in the data path.
Inputs (the STFT of seeded noise, exactly the tensors the reference inferencer hands to the model,
inferencer.py:142-147) are resident in HBM before the timed region; the STFT itself is outside it
on both sides.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     - dominant kernel = fused sub-band LSTM: algorithmic FLOPs per launch / its average
                 duration measured with hipEvents on the forward's own stream (inside libfsnp_hip).
  cpu_baseline - oracle/fsnp_torch.py (torch-CPU restatement of the reference, kind "port") timed on
                 this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)" (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--mode", default="full", choices=["full", "parity"])
    ap.add_argument("--norm", default="offline_laplace_norm")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16_ih"],
                    help="bf16_ih = BASELINE.json configs[4] (NOT the headline: reduced-precision ih-GEMM)")
    ap.add_argument("--model", default="plus", choices=["plus", "fullsubnet"],
                    help="plus = FullSubNet+ (the headline); fullsubnet = the original FullSubNet Model (SURVEY.md 8f-2)")
    ap.add_argument("--sequence-model", default="LSTM", choices=["LSTM", "GRU", "TCN"],
                    help="sub-band sequence model (SURVEY.md 8f-4 variants; the headline is LSTM; TCN: FullSubNet+ only)")
    ap.add_argument("--wave", action="store_true",
                    help="time waveform -> waveform (HIP STFT + forward + cIRM + iSTFT, model.enhance_wave) instead of the "
                         "headline forward; extra information, not BASELINE.json's metric")
    ap.add_argument("--dist-backend", default="nccl", help='"nccl" (= RCCL; the real launch) or "gloo" (plumbing tests)')
    ap.add_argument("--same-device", action="store_true",
                    help="testing only: every rank uses GPU 0 (checks the N>1 plumbing on a 1-GPU box with --dist-backend gloo)")
    ap.add_argument("--pipeline", type=int, default=1, choices=[0, 1],
                    help="1 (default) = the serving loop: fsnp_set_pipeline, the remainder chunk of forward i (the 32 of 8224 "
                         "sequences that do not fit the chip-filling launch) overlaps the full-band stages of forward i+1; all "
                         "K forwards are complete (fsnp_flush + synchronize) inside the timed region.  0 = every forward runs "
                         "strictly back to back.  The other mode is timed too and reported as `alt_ms_per_step`")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timing of the other loop mode (profiling runs)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="every timed loop (K steps, barrier + synchronise on both sides) is run this many times; the line reports the "
                         "MEDIAN run (`ms_per_step`, `value`, `roofline`) and all of them (`ms_per_step_runs`): one 0.6 s region "
                         "cannot tell a clock ramp or a slow box from a regression (VERDICT r05)")
    ap.add_argument("--probe-ms", type=float, default=50.0,
                    help="length of the fp32-MFMA peak probe (fsnp_debug_box_probe) run before and after the timed loops; 0 = off")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: ALSO time BASELINE.json configs[2] as worded - ONE global batch of batch x N utterances lives on rank 0; "
                         "a step = scatter the STFT shards + forward + gather the masks to rank 0 (reported as `strong`, next to the "
                         "weak-scaling number, which stays the headline)")
    ap.add_argument("--verify-sample", type=int, default=None,
                    help="fsnp_set_verify_sample for every loop (default: the module's policy - 16 in the drop-in loop's error_check='sync', off "
                         "in the deferred loops); 0 = off everywhere")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(sd, inputs, budget_s, norm, fullsubnet=False):
    """Time the torch-CPU port of the reference forward on host cores, one utterance per call
    (== "full" semantics), cycling through the batch until the time budget (15 s) is used.  torch's CPU LSTM
    collapses when oversubscribed (256 threads on the GPU box: 118 s per utterance), so the thread count
    is picked by a one-utterance sweep and reported as `cores`."""
    from oracle import fsnp_torch
    mag, real, imag = inputs
    T = mag.shape[-1]
    if fullsubnet:
        def fwd(m, r, i, norm_type):
            return fsnp_torch.forward_fullsubnet_full(sd, m, norm_type=norm_type)
    else:
        def fwd(m, r, i, norm_type):
            return fsnp_torch.forward_full(sd, m, r, i, norm_type=norm_type)
    ncpu = os.cpu_count() or 1
    best_threads, best_dt = 1, float("inf")
    for th in [c for c in (8, 16, 32) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        fwd(mag[:1], real[:1], imag[:1], norm)  # warm-up
        t0 = time.perf_counter()
        fwd(mag[:1], real[:1], imag[:1], norm)
        dt = time.perf_counter() - t0
        if dt < best_dt:
            best_threads, best_dt = th, dt
    torch.set_num_threads(best_threads)
    done, t0 = 0, time.perf_counter()
    outs = {}                                                 # every utterance the timed loop reaches is kept for the check
    while time.perf_counter() - t0 < budget_s:               # cycle through the batch until the budget is used
        i = done % mag.shape[0]
        o = fwd(mag[i:i + 1], real[i:i + 1], imag[i:i + 1], norm)
        outs.setdefault(i, o)
        done += 1
    dt = time.perf_counter() - t0
    last = mag.shape[0] - 1                                   # untimed: the LAST utterance too (B = 32: its upper bins are the
    if last not in outs:                                      # 32 sequences that run on the remainder kernel)
        outs[last] = fwd(mag[last:], real[last:], imag[last:], norm)
    return {"value": done * T / dt, "unit": "frames/s", "cores": best_threads, "kind": "port",
            "sample": f"{done} x 1-utterance forwards of the {T}-frame clips (oracle/fsnp_torch.py, "
                      f"torch {torch.__version__} CPU, {best_threads} of {ncpu} host threads), {dt:.1f} s"}, outs


def dominant_kernel_source_digest():
    """sha256 of the sources the one-tile-per-CU LSTM kernel is compiled from (profiles/lstm_pmc.json records the same digest)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("lstm.hip", "lstm_common.h", "fsnp_common.h"):
        with open(os.path.join(ROOT, "fullsubnet_plus_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pci_address(dev):
    """"0000:05:00.0" of a torch device (None when this torch build does not expose it): selects the GPU's sysfs directory."""
    try:
        p = torch.cuda.get_device_properties(dev)
        return "{:04x}:{:02x}:{:02x}.0".format(p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:                                            # noqa: BLE001 - best effort
        return None


def strong_scaling_loop(args, model, dist, dev, rank, world, B, sync_all):
    """BASELINE.json configs[2] as worded: ONE global batch of B x world utterances, resident on rank 0.  A step =
    scatter the complex STFT shards (what the reference inferencer holds after torch.stft, inferencer.py:142) -> forward on every
    rank (fsnp_forward_complex consumes the buffer in place) -> gather the masks on rank 0.  The batch split needs no other
    collective.  RCCL implements scatter / gather as grouped send / recv over xGMI; on gloo (tests) the shards go through host memory."""
    from fullsubnet_plus_amd.synthetic import make_wave
    n_fft, hop = 2 * (model.num_freqs - 1), model.num_freqs - 1
    via_host = args.dist_backend != "nccl"
    GB = B * world
    T = None
    chunks = None
    if rank == 0:
        wav = torch.from_numpy(make_wave(GB, args.seconds, 4242)).to(dev)
        spec = torch.stft(wav, n_fft, hop, n_fft, window=torch.hann_window(n_fft, device=dev), return_complex=True)   # [GB, F, T], memory [GB][T][F]
        T = spec.shape[-1]
        flat = torch.view_as_real(spec.transpose(1, 2).contiguous())       # [GB, T, F, 2] contiguous
        chunks = [c.cpu() if via_host else c.contiguous() for c in flat.chunk(world, dim=0)]
    tt = torch.tensor([T or 0], dtype=torch.int64, device="cpu" if via_host else dev)
    dist.broadcast(tt, 0)
    T = int(tt.item())
    cdev = "cpu" if via_host else dev
    mine = torch.empty((B, T, model.num_freqs, 2), dtype=torch.float32, device=cdev)
    masks = [torch.empty((B, model.output_size, model.num_freqs, T), dtype=torch.float32, device=cdev) for _ in range(world)] if rank == 0 else None
    model.set_pipeline(False, dev)
    t_sc = t_fw = t_ga = 0.0

    def step(timed):
        nonlocal t_sc, t_fw, t_ga
        a = time.perf_counter()
        dist.scatter(mine, chunks if rank == 0 else None, src=0)
        x = mine.to(dev) if via_host else mine
        torch.cuda.synchronize(dev)
        b = time.perf_counter()
        o = model.forward_complex(torch.view_as_complex(x).transpose(1, 2))           # [B, F, T] view with torch.stft's strides
        torch.cuda.synchronize(dev)
        c = time.perf_counter()
        oc = o.cpu() if via_host else o.contiguous()
        dist.gather(oc, masks, dst=0)
        torch.cuda.synchronize(dev)
        d = time.perf_counter()
        if timed:
            t_sc += b - a; t_fw += c - b; t_ga += d - c
        return o

    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            step(False)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        sync_all()
        dt = time.perf_counter() - t0
    model.poll_errors()
    every = [None] * world
    dist.all_gather_object(every, (dt, t_sc, t_fw, t_ga))
    slow = max(range(world), key=lambda r: every[r][0])
    dt = every[slow][0]
    return {"what": "configs[2] as worded: ONE global batch on rank 0; step = scatter STFT shards + forward + gather masks to rank 0",
            "global_batch": GB, "ms_per_step": dt / args.steps * 1e3, "value": GB * T * args.steps / dt, "unit": "frames/s",
            "scaling": "strong in the data placement (one source rank), weak in the work per GPU (batch x N)",
            "scatter_ms": every[0][1] / args.steps * 1e3, "forward_ms_slowest_rank": max(e[2] for e in every) / args.steps * 1e3,
            "gather_ms": every[0][3] / args.steps * 1e3, "via_host_memory": via_host,
            "bytes_scattered_per_step": GB * T * model.num_freqs * 8, "bytes_gathered_per_step": GB * model.output_size * model.num_freqs * T * 4}


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks ourselves (one process per
    GPU, rendezvous on 127.0.0.1) and exit with their status.  Fails loudly when fewer than N GPUs are visible."""
    visible = torch.cuda.device_count()
    if not args.same_device and visible < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / cross-process tensors)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
        # several ranks on ONE GPU (plumbing tests only): the column-split kernels need all their workgroups resident, so
        # two processes must not each plan launches that fill every CU twice, nor calibrate against each other
        os.environ.setdefault("FSNP_COOP_OCC", "1")
        os.environ.setdefault("FSNP_CALIBRATE", "0")
        if world > 2:
            # more than two processes interleaving co-resident launches on one GPU can each hold part of the chip and wait for the
            # rest (the kernels' 2 s time-out would end it, loudly): the plumbing test of many ranks runs the exchange-free kernel
            os.environ.setdefault("FSNP_LSTM_COOP", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):     # under torch.distributed.run: also at N = 1 (RCCL with one rank)
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU")

    from fullsubnet_plus_amd import FullSubNet, FullSubNet_Plus
    from fullsubnet_plus_amd.synthetic import (DEFAULT_MODEL_ARGS, FULLSUBNET_MODEL_ARGS, make_inputs, make_state_dict,
                                               make_state_dict_fullsubnet)

    fsn = args.model == "fullsubnet"
    if fsn:
        assert args.precision == "fp32", "the bf16 ih-GEMM variant is defined for FullSubNet+ only"
        sd = make_state_dict_fullsubnet(0, "default", sequence_model=args.sequence_model)
        model = FullSubNet(**{**FULLSUBNET_MODEL_ARGS, "norm_type": args.norm, "sequence_model": args.sequence_model})
    else:
        sd = make_state_dict(0, "default", sequence_model=args.sequence_model)
        model = FullSubNet_Plus(**{**DEFAULT_MODEL_ARGS, "norm_type": args.norm, "sequence_model": args.sequence_model})
    if args.sequence_model != "LSTM":
        assert args.precision == "fp32", "the bf16 ih-GEMM variant exists for the LSTM sub-band model only"
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.batch_mode = args.mode
    model.error_check = "deferred"       # no host wait per forward; poll_errors() below, after the final synchronisation
    model.verify_sample_every = args.verify_sample

    B = args.batch
    cpu_in = make_inputs(B, args.seconds, 1000 + rank)           # synthetic, per-rank seed
    gpu_in = []
    for t in cpu_in:                                             # keep the stft strides on the device
        g = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev)
        g.copy_(t)
        gpu_in.append(g)
    T = cpu_in[0].shape[-1]
    if fsn:
        gpu_in = gpu_in[:1]                                      # the original FullSubNet takes the magnitude only

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.wave:
        from fullsubnet_plus_amd.synthetic import make_wave
        wav = torch.from_numpy(make_wave(B, args.seconds, 1000 + rank)).to(dev)

    def run_once():
        return model.enhance_wave(wav) if args.wave else model(*gpu_in)

    with torch.no_grad():
        out = model.enhance_wave(wav[:1]) if args.wave else model(*[t[:1] for t in gpu_in])   # creates the handle
        model.set_precision(args.precision)

        def timed_loop(pipeline):
            model.set_pipeline(bool(pipeline), dev)
            o = None
            for _ in range(max(args.warmup, 1)):                 # also re-grows the (re-shaped) workspace outside the timing
                o = run_once()
            model.flush()
            sync_all()
            model.set_timing(True)
            model.get_timing(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                o = run_once()
            model.flush()                                        # deferred remainder chunks: ordered before the sync below
            sync_all()
            dt = time.perf_counter() - t0
            tm = model.get_timing(reset=True)
            model.set_timing(False)
            model.poll_errors()                                  # a column-split launch that gave up would have flagged the handle
            return dt, tm, o

        from fullsubnet_plus_amd import box as box_mod
        stream_ptr = torch.cuda.current_stream(dev).cuda_stream
        sysfs_dir = box_mod.sysfs_device(pci_address(dev))
        idle_state = box_mod.read_sysfs(sysfs_dir)
        # what this box's matrix pipes hold, right before the timed loops (the first forward above has woken the clocks)
        probe_before = box_mod.probe(args.probe_ms, stream_ptr) if args.probe_ms > 0 else None
        sampler = box_mod.Sampler(sysfs_dir) if sysfs_dir else None
        if sampler:
            sampler.start()

        def timed_runs(pipeline):
            """args.repeats x timed_loop -> (elapsed of the MEDIAN run, its timing, its last output, [ms per step of every run], [dominant-kernel ms of every run])"""
            runs = [timed_loop(pipeline) for _ in range(max(args.repeats, 1))]
            order = sorted(range(len(runs)), key=lambda i: runs[i][0])
            med = runs[order[len(order) // 2]]
            return (med[0], med[1], runs[-1][2], [r[0] / args.steps * 1e3 for r in runs],
                    [r[1]["lstm_first_chunk_ms"] / max(r[1]["count"], 1) for r in runs])

        pipelined = bool(args.pipeline) and not args.wave
        alt_elapsed, alt_timing, _, alt_runs, _ = timed_runs(not pipelined) if not (args.wave or args.no_alt) else (None, None, None, None, None)
        # the drop-in default: what a maintainer who edits the one TOML line gets - error_check="sync" (every forward waits for its
        # own launches and polls the error word before it returns), no pipelined loop
        dropin_elapsed, dropin_runs = None, None
        if not (args.wave or args.no_alt) and world == 1:
            model.error_check = "sync"
            dropin_elapsed, _, _, dropin_runs, _ = timed_runs(False)
            model.error_check = "deferred"
        elapsed, timing, out, main_runs, launch_runs = timed_runs(pipelined)
        launch_clock = model.launch_clock()                     # the clocks of the LAST dominant-kernel launch (None: no such launch)
        sampled = sampler.stop() if sampler else {"samples": 0, "note": "no readable /sys/class/drm/card*/device for this GPU"}
        probe_after = box_mod.probe(args.probe_ms, stream_ptr) if args.probe_ms > 0 else None

        strong = None
        if args.strong and dist is not None and not args.wave and not fsn:
            strong = strong_scaling_loop(args, model, dist, dev, rank, world, B, sync_all)

    rank_ms = None
    if dist is not None:
        cnt = max(timing["count"], 1)
        mine = torch.tensor([elapsed, timing["lstm_ms"] / cnt, timing["lstm_first_chunk_ms"] / cnt, timing["fullband_ms"] / cnt,
                             timing["forward_ms"] / cnt, probe_before["mfma_tflops"] if probe_before else 0.0,
                             probe_after["mfma_tflops"] if probe_after else 0.0, probe_after["clock_mhz"] if probe_after else 0.0],
                            dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t[0].item()) / args.steps * 1e3 for t in every]
        slow = max(range(world), key=lambda r: float(every[r][0].item()))
        elapsed = float(every[slow][0].item())                 # the job is as slow as its slowest rank ...
        # ... and the roofline block below describes THAT rank's kernels (the N > 1 line carries the slowest rank's roofline)
        timing = {"count": 1, "lstm_ms": float(every[slow][1]), "lstm_first_chunk_ms": float(every[slow][2]),
                  "fullband_ms": float(every[slow][3]), "forward_ms": float(every[slow][4])}
        rank_probe = [{"rank": r, "mfma_tflops_before": float(every[r][5]), "mfma_tflops_after": float(every[r][6]),
                       "clock_mhz_after": float(every[r][7]), "ms_per_step": rank_ms[r]} for r in range(world)]
        # batch-split plumbing: gather every rank's masks once (outside the timed region)
        g0 = time.perf_counter()
        gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, out.contiguous())
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - g0) * 1e3
    else:
        gather_ms, rank_probe = None, None

    frames_total = world * B * T * args.steps
    value = frames_total / elapsed
    rows = B * (model.num_freqs // 2 if (args.mode == "parity" and B > 1) else model.num_freqs)
    Tp = T + model.look_ahead
    from fullsubnet_plus_amd import _lib
    # roofline of the DOMINANT kernel = the first chunk of the sub-band plan (B=32: 8192 of the 8224 sequences on the
    # one-tile-per-CU kernel; the 32 left over run on the K-split kernel afterwards); stage_* = the whole sub-band model
    plan = model.describe_plan(B, parity=(args.mode == "parity" and B > 1))
    stage_flops = float(_lib.load().fsnp_lstm_flops(model._handle, rows, Tp))
    stage_ms = timing["lstm_ms"] / max(timing["count"], 1)
    lstm_flops = float(_lib.load().fsnp_lstm_flops(model._handle, plan[0]["sequences"], Tp))
    lstm_ms = timing["lstm_first_chunk_ms"] / max(timing["count"], 1)
    achieved = lstm_flops / (lstm_ms * 1e-3) / 1e12 if lstm_ms > 0 else 0.0
    stage_achieved = stage_flops / (stage_ms * 1e-3) / 1e12 if stage_ms > 0 else 0.0

    lstm_kernel_name = plan[0]["kernel"]
    probes = [p for p in (probe_before, probe_after) if p]
    box_peak = min(p["mfma_tflops"] for p in probes) if probes else None
    # HBM-side bytes per launch of the dominant kernel, from committed rocprofv3 PMC passes - reported only while the kernel's sources are
    # the ones those passes ran (the file carries their digest): a kernel edit without new counter passes turns the field to null
    traffic, traffic_note = None, None
    pmc_path = os.path.join(ROOT, "profiles", "lstm_pmc.json")
    if os.path.exists(pmc_path) and B == 32 and abs(args.seconds - 2.0) < 1e-9 and args.mode == "full":
        with open(pmc_path) as f:
            pmc = json.load(f)
        want = pmc.get("kernel_source_sha256")
        have = dominant_kernel_source_digest()
        if want is None or want == have:
            traffic = pmc.get("traffic_bytes_per_launch")
            traffic_note = ("profiles/lstm_pmc.json (rocprofv3 --pmc passes of this kernel, refreshed at the end of every round by tools/gpu_r06_final.sh and "
                            "committed; NOT re-measured by this run; kernel sources " + ("match the passes': " + have[:12] if want else "not recorded in the file") + ")")
        else:
            traffic_note = f"profiles/lstm_pmc.json was measured on other kernel sources ({want[:12]}, now {have[:12]}): not reported"

    result = {
        "metric": "STFT frames/sec (257-bin, 2 s clips), " + ("FullSubNet" if fsn else "FullSubNet+") +
                  (" waveform -> waveform (HIP STFT + forward + cIRM + iSTFT)" if args.wave else " forward"),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        # every timed loop runs --repeats times (each: K steps between barrier + synchronise); the headline is the MEDIAN run
        "ms_per_step_runs": main_runs, "alt_ms_per_step_runs": alt_runs, "dropin_ms_per_step_runs": dropin_runs,
        "alt_ms_per_step": None if alt_elapsed is None else alt_elapsed / args.steps * 1e3,
        # the drop-in default (error_check="sync", no pipelined loop): one TOML line edited, nothing else
        "dropin_ms_per_step": None if dropin_elapsed is None else dropin_elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak",
        # the arithmetic type per launch of the sub-band plan (the bf16 variants exist for the one-tile-per-CU LSTM kernel only)
        "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else
                       "; ".join(f"{c['precision']} on {c['sequences']} sequences" for c in plan) +
                       " (BASELINE configs[4])",
        "data": "synthetic",
        "config": {"workload": f"batch={B} x {args.seconds:g} s clips per GPU (T={T} frames, 257 bins), "
                               f"{args.mode} mode, num_neighbors=15, {args.norm}, random-init weights (seed 0)" +
                               ("" if args.sequence_model == "LSTM" else f", sequence_model={args.sequence_model}"),
                   "global_batch": world * B, "frames_per_clip": T, "parallelism": f"dp{world} (batch split, no data-path collective)",
                   "loop": ("pipelined serving loop: the column-split remainder chunk of forward i overlaps the full-band stages "
                            "of forward i+1 where the planner says that pays (fsnp_set_pipeline; roofline.subband_plan[].deferred_when_pipelined); "
                            "every forward is complete inside the timed region; alt_ms_per_step = the same K forwards strictly back to back; "
                            "dropin_ms_per_step = the same with the module's default error_check='sync' (each forward waits for its launches)") if pipelined else
                           "forwards strictly back to back (alt_ms_per_step = the pipelined serving loop)"},
        "roofline": {"bound": "mfma", "kernel": lstm_kernel_name, "achieved": achieved,
                     "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                     "traffic": traffic,
                     "traffic_source": traffic_note,
                     # the same rate against what THIS box's matrix pipes held in a pure-MFMA probe right before / after the loops
                     # (box.mfma_peak_tflops: the lower of the two) - a slow box shows as frac < frac_of_box_peak, a slow kernel in both
                     "frac_of_box_peak": (achieved / box_peak) if box_peak else None,
                     "flops_per_launch": lstm_flops, "avg_launch_ms": lstm_ms, "avg_launch_ms_runs": launch_runs,
                     # the last dominant launch from the inside: workgroup 0's wall time (100 MHz clock) and the shader clock it held
                     # (s_memtime = shader cycles), and the wall time of the launch's slowest / fastest workgroup - the launch lasts as
                     # long as its slowest workgroup, and the XCDs of a chip do not all hold the same clock
                     "last_launch_clock": launch_clock,
                     # ... and the rate against the peak AT THAT CLOCK (65,536 FLOP per shader cycle on 256 CUs): the kernel needs a
                     # fixed number of cycles per launch, its milliseconds follow the clock the box holds
                     "frac_at_held_clock": (lstm_flops / (launch_clock["wall_ms"] * 1e-3) / (65536.0 * launch_clock["s_memtime_mhz"] * 1e6)
                                            if launch_clock and plan[0]["kernel"].startswith("lstm2_fc_kernel") else None),
                     "subband_plan": plan, "subband_stage_ms": stage_ms, "subband_stage_tflops": stage_achieved,
                     "fullband_ms": timing["fullband_ms"] / max(timing["count"], 1),
                     # the same stage in the OTHER loop (in the serving loop it runs beside the previous forward's remainder
                     # chunk, which owns 48 CUs; back to back it has the chip to itself)
                     "alt_fullband_ms": None if alt_timing is None else alt_timing["fullband_ms"] / max(alt_timing["count"], 1),
                     "forward_ms": timing["forward_ms"] / max(timing["count"], 1)},
    }
    result["box"] = {
        "mfma_peak_tflops": box_peak, "spec_peak_tflops": PEAK_FP32_MFMA_TFLOPS,
        "probe": "fsnp_debug_box_probe: v_mfma_f32_32x32x2_f32 only, one wave per SIMD on every CU, %g ms, random operands; run right "
                 "before and right after the timed loops" % args.probe_ms,
        "probe_before": probe_before, "probe_after": probe_after,
        "idle_before": idle_state, "during_timed_loops": sampled,
        "hostname": socket.gethostname(), "gpu": torch.cuda.get_device_name(dev), "gpu_unique_id": box_mod.gpu_unique_id(sysfs_dir),
    }
    if rank_probe is not None:
        result["box"]["per_rank"] = rank_probe
    if strong is not None:
        result["strong"] = strong
    if gather_ms is not None:
        result["gather_ms"] = gather_ms
        result["dist"] = {"backend": args.dist_backend + (" (RCCL)" if args.dist_backend == "nccl" else ""),
                          "world_size_seen": dist.get_world_size(), "same_device": bool(args.same_device),
                          "gathered_shape": list(gathered.shape),
                          "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)}, "roofline_of_rank": slow}
    if rank == 0 and not args.no_cpu_baseline:           # (rank 0 only; at N > 1 the other ranks wait in destroy_process_group)
        base, ref_outs = cpu_baseline(sd, cpu_in, args.cpu_budget_s, args.norm, fsn)
        ref_path = next((p for p in (os.path.join(ROOT, "profiles", f"r0{n}_cli_e2e.json") for n in (6, 5, 4)) if os.path.exists(p)),
                        os.path.join(ROOT, "profiles", "r04_cli_e2e.json"))    # the REAL reference class timed on a GPU box's host cores
        if os.path.exists(ref_path) and not fsn:                        # (tools/cli_e2e.py; the reference is not on this box)
            with open(ref_path) as f:
                rm = json.load(f)["reference_cpu_forward"]
            base["reference_measured"] = {"value": rm["value"], "unit": "frames/s", "cores": rm["best_threads"], "kind": "reference",
                                          "source": os.path.relpath(ref_path, ROOT) + " (the unmodified reference class on a GPU box's host cores, tools/cli_e2e.py; committed measurement, not re-run here)"}
        result["cpu_baseline"] = base
        if args.mode == "full" and not args.wave:
            # every utterance of the timed batch the CPU leg computed (B = 32 x 2 s: all 32 - the 15 s budget cycles through
            # the batch about four times) against the HIP output of the last timed forward
            got = out.cpu()
            errs = {i: (float((got[i:i + 1] - r).abs().max()), float(r.abs().max())) for i, r in sorted(ref_outs.items())}
            result["cirm_max_abs_err"] = max(e for e, _ in errs.values())
            result["cirm_rel_err"] = max(e / m for e, m in errs.values())
            result["cirm_checked_utterances"] = sorted(errs)
            result["cirm_worst_utterance"] = max(errs, key=lambda i: errs[i][0] / errs[i][1])
    if rank == 0:
        from fullsubnet_plus_amd import box as box_mod2
        result["box"]["rocm_smi_after"] = box_mod2.smi_snapshot()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()                       # (rank 0 ran the CPU baseline meanwhile)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
