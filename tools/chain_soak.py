#!/usr/bin/env python3
"""Soak for the relaxed per-device chain of column-split launches (csrc/fsnp_abi.hip launch_coop_chained, round 6): the original FullSubNet's
full-band LSTM of forward i + 1 runs beside the deferred remainder chunk of forward i.  Pipelined serving loops of FullSubNet at the batch sizes
whose plans defer a remainder chunk, alone and next to a FullSubNet+ handle on a second stream (cross-handle launches stay chained); every
50th result is compared with the first forward's (<= 1e-5 of the peak: the statistics are fp64 atomics whose order moves with what else runs on
the chip - a lost hand-off would be 1e-1), the error word is polled.    python tools/chain_soak.py [forwards per case]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullsubnet_plus_amd import FullSubNet, FullSubNet_Plus  # noqa: E402
from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, FULLSUBNET_MODEL_ARGS, make_inputs, make_state_dict, make_state_dict_fullsubnet  # noqa: E402


def build(cls, args, sd):
    m = cls(**args)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.batch_mode = "full"
    m.error_check = "deferred"
    m.set_pipeline(True)
    return m


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    fsn = build(FullSubNet, FULLSUBNET_MODEL_ARGS, make_state_dict_fullsubnet(0, "default"))
    plus = build(FullSubNet_Plus, DEFAULT_MODEL_ARGS, make_state_dict(0, "default"))
    side = torch.cuda.Stream()
    for B in (32, 16, 8, 40):
        mag = make_inputs(B, 2.0, 7)[0].cuda()
        ins_plus = [t.cuda() for t in make_inputs(B, 2.0, 7)]
        for _ in range(2):                              # (flush BEFORE the result is read: the deferred chunk's rows are complete only then)
            want = fsn(mag); fsn.flush(); torch.cuda.synchronize()
        with torch.cuda.stream(side):
            want_plus = plus(*ins_plus); plus.flush()
        torch.cuda.synchronize()
        for both in (False, True):
            t0 = time.perf_counter()
            bad, worst = 0, 0.0
            def dev(a, b):
                return float((a - b).abs().max() / b.abs().max())
            for i in range(n):
                out = fsn(mag)
                if both:
                    with torch.cuda.stream(side):
                        out_plus = plus(*ins_plus)
                if i % 50 == 49:
                    fsn.flush()
                    if both:
                        with torch.cuda.stream(side):
                            plus.flush()
                    torch.cuda.synchronize()
                    fsn.poll_errors()
                    e = dev(out, want)
                    if both:
                        plus.poll_errors()
                        e = max(e, dev(out_plus, want_plus))
                    worst = max(worst, e)
                    bad += int(not e <= 1e-5)
            fsn.flush()
            with torch.cuda.stream(side):
                plus.flush()
            torch.cuda.synchronize()
            fsn.check_errors(); plus.check_errors()
            dt = (time.perf_counter() - t0) / n * 1e3
            print(f"FullSubNet B = {B}{' + FullSubNet+ on a second stream' if both else ''}: {n} forwards, {dt:.3f} ms each, worst deviation {worst:.1e}, {bad} checks beyond 1e-5", flush=True)
            assert bad == 0


if __name__ == "__main__":
    main()
