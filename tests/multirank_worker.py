"""Rank body of tests/test_gpu_multirank.py (started by torch.distributed.run): every rank drives GPU 0, computes its shard
of a 5-utterance batch with the HIP model through fullsubnet_plus_amd.dist.forward_sharded and gathers over gloo - or, with
a third argument "nccl", over RCCL (one rank per GPU: world size 1 on the one-GPU test box)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    if backend == "nccl":                      # RCCL: one GPU per rank (world size 1 on a one-GPU box)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    # two processes share GPU 0 here: keep every column-split launch at one workgroup per CU (see bench.py --same-device)
    os.environ.setdefault("FSNP_COOP_OCC", "1")
    os.environ.setdefault("FSNP_CALIBRATE", "0")
    torch.cuda.set_device(0)
    from fullsubnet_plus_amd import FullSubNet_Plus
    from fullsubnet_plus_amd.dist import forward_sharded
    from fullsubnet_plus_amd.synthetic import DEFAULT_MODEL_ARGS, make_inputs, make_state_dict
    m = FullSubNet_Plus(**DEFAULT_MODEL_ARGS)
    m.load_state_dict(make_state_dict(0, "default"), strict=True)
    m = m.to("cuda:0").eval()
    m.batch_mode = mode
    ins = [t.cuda() for t in make_inputs(5, 0.5, 77)]
    with torch.no_grad():
        out = forward_sharded(m, *ins, gather=True)
    m.check_errors()
    if dist.get_rank() == 0:
        np.save(out_path, out.cpu().numpy())
        with open(out_path + ".info", "w") as f:
            f.write(f"{dist.get_backend()} {dist.get_world_size()}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
