"""Batch sharding of the forward across the GPUs of one node (one process per GPU).

Utterances are independent (every statistic on the path is per utterance: base_model.py:221, the TSSE pool,
GroupNorm(1), the per-sequence LSTM), so the forward shards over the batch with NO collective in the data path.
The only traffic is the optional gather of the masks, `torch.distributed.all_gather_into_tensor` on the "nccl"
backend (= RCCL over xGMI on ROCm); on CPU-only boxes the same code runs on "gloo" for tests.

In the reference's literal B>1 semantics ("parity" mode = drop_band, acoustics/feature.py:254-285) the frequency
parity and the output row of a sample depend on its GLOBAL batch index, so a shard passes (batch_offset,
global_batch) down to fsnp_forward and receives its rows of the global [B,2,F//2,T] tensor.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world_size):
    """Contiguous, balanced split: rank r owns samples [lo, hi)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def parity_output_row(sample, global_batch, num_groups=2):
    """Row of the reference's drop_band output that holds global sample `sample`: the samples of group 0
    (sample % num_groups == 0) come first, then group 1's, ... (feature.py:270-283)."""
    g = sample % num_groups
    return sum(-(-(global_batch - q) // num_groups) for q in range(g)) + sample // num_groups


def forward_sharded(model, noisy_mag, noisy_real=None, noisy_imag=None, gather=True, group=None):
    """noisy_real / noisy_imag = None: the original FullSubNet (`model(noisy_mag)`, fullsubnet.py:68).
    Every rank passes the FULL-batch tensors' own shard (already on its device) or the full batch; here each
    rank receives the global batch on its device and computes only its shard.

    Returns the global mask tensor on every rank if `gather`, else this rank's rows:
      full mode   -> [hi-lo, 2, F, T]
      parity mode -> the global [B, 2, F//2, T] with only this rank's rows filled.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = noisy_mag.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    groups = getattr(model, "num_groups_in_drop_band", 2)
    parity = B > 1 and model.batch_mode == "parity" and groups > 1
    sl = slice(lo, hi)
    if hi > lo:
        ins = [noisy_mag[sl]] if noisy_real is None else [noisy_mag[sl], noisy_real[sl], noisy_imag[sl]]
        out = model(*ins, batch_offset=lo, global_batch=B)
    else:
        F = model.num_freqs // groups if parity else model.num_freqs
        out = torch.zeros((B if parity else 0, getattr(model, "output_size", 2), F, noisy_mag.shape[-1]), dtype=torch.float32, device=noisy_mag.device)
    if not gather or not dist.is_initialized():
        return out                              # (a process group of ONE rank still runs the collective: the RCCL path is
                                                #  the same call at every world size, tests/test_gpu_multirank.py exercises it)
    sizes = [shard_bounds(B, r, world) for r in range(world)]
    rows = max(h - l for l, h in sizes)
    if parity:
        # every rank holds the global [B, OC, F // G, T] tensor with only its own rows written (rows of the reference's drop_band
        # order, parity_output_row).  Gather the row BLOCKS and scatter them by index: world x fewer bytes on the wire than summing
        # the zero-filled global tensors, and exact (no 0 + x arithmetic).
        mine = [parity_output_row(sm, B, groups) for sm in range(lo, hi)]
        block = out[mine] if mine else out[:0]
    else:
        block = out
    if block.shape[0] < rows:                   # ragged / empty shards: pad to the largest, trim after the gather
        pad = torch.zeros((rows - block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        block = torch.cat([block, pad], dim=0)
    full = torch.empty((world * rows,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(full, block.contiguous(), group=group)
    if parity:
        res = torch.empty_like(out)
        for r, (l, h) in enumerate(sizes):
            if h > l:
                res[[parity_output_row(sm, B, groups) for sm in range(l, h)]] = full[r * rows: r * rows + (h - l)]
        return res
    if all(h - l == rows for l, h in sizes):
        return full
    return torch.cat([full[r * rows: r * rows + (h - l)] for r, (l, h) in enumerate(sizes)], dim=0)
