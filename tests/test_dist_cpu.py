"""CPU tests of the N>1 path: world_size-2 gloo processes exercise fullsubnet_plus_amd.dist (shard bounds,
parity-mode global row placement, the gather collectives) with a stand-in model whose forward is the torch-CPU
oracle restricted to a shard - the HIP forward itself needs a GPU, the sharding logic does not."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fullsubnet_plus_amd import dist as fdist
from oracle import fsnp_torch
from oracle.make_golden import make_spec
from oracle.weights import make_state_dict


def test_shard_bounds_cover_batch():
    for B in (1, 3, 7, 32, 256):
        for W in (1, 2, 3, 8):
            spans = [fdist.shard_bounds(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_parity_rows_match_drop_band_order():
    # acoustics/feature.py:276-285: even samples first, then odd
    for B in (3, 4, 5, 9):
        order = list(range(0, B, 2)) + list(range(1, B, 2))
        for s in range(B):
            assert order[fdist.parity_output_row(s, B)] == s
    for G in (3, 4):                                          # num_groups_in_drop_band != 2
        for B in (G + 1, 7, 10):
            order = [s for g in range(G) for s in range(g, B, G)]
            for s in range(B):
                assert order[fdist.parity_output_row(s, B, G)] == s


class _OracleShardModel:
    """Mimics FullSubNet_Plus.forward(batch_offset, global_batch) semantics with the CPU oracle."""

    def __init__(self, sd, mode):
        self.sd, self.batch_mode, self.num_freqs = sd, mode, 257

    def __call__(self, mag, real, imag, batch_offset=0, global_batch=None):
        gb = mag.shape[0] if global_batch is None else global_batch
        full = fsnp_torch.forward_full(self.sd, mag, real, imag)            # [b,2,257,T]
        if not (gb > 1 and self.batch_mode == "parity"):
            return full
        out = torch.zeros((gb, 2, 128, mag.shape[-1]))
        for i in range(mag.shape[0]):
            s = batch_offset + i
            out[fdist.parity_output_row(s, gb)] = full[i][:, (s % 2):256:2, :]
        return out


class _OracleShardFullSubNet(_OracleShardModel):
    """Same for the original FullSubNet: model(noisy_mag, batch_offset, global_batch)."""

    def __call__(self, mag, batch_offset=0, global_batch=None):
        gb = mag.shape[0] if global_batch is None else global_batch
        full = fsnp_torch.forward_fullsubnet_full(self.sd, mag)
        if not (gb > 1 and self.batch_mode == "parity"):
            return full
        out = torch.zeros((gb, 2, 128, mag.shape[-1]))
        for i in range(mag.shape[0]):
            s = batch_offset + i
            out[fdist.parity_output_row(s, gb)] = full[i][:, (s % 2):256:2, :]
        return out


def _worker_fsn(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle.weights import make_state_dict_fullsubnet
    model = _OracleShardFullSubNet(make_state_dict_fullsubnet(3), "parity")
    out = fdist.forward_sharded(model, make_spec(5, 12, 43)[0], gather=True)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_forward_sharded_fullsubnet_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fsn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle.weights import make_state_dict_fullsubnet
    want = fsnp_torch.forward_fullsubnet(make_state_dict_fullsubnet(3), make_spec(5, 12, 43)[0]).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()


def _worker(rank, world, port, mode, q, batch=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sd = make_state_dict(3)
    mag, real, imag = make_spec(batch, 12, 42)
    model = _OracleShardModel(sd, mode)
    out = fdist.forward_sharded(model, mag, real, imag, gather=True)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["full", "parity"])
def test_forward_sharded_world2_gloo(mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = make_state_dict(3)
    mag, real, imag = make_spec(5, 12, 42)
    if mode == "full":
        want = fsnp_torch.forward_full(sd, mag, real, imag).numpy()
    else:
        want = fsnp_torch.forward(sd, mag, real, imag).numpy()              # the reference's literal B>1 call
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("world,batch,mode", [(4, 3, "full"), (4, 3, "parity"), (8, 5, "full"), (8, 5, "parity")])
def test_forward_sharded_ragged_and_empty_shards_gloo(world, batch, mode):
    """More ranks than utterances (the 8-GPU node with a small batch): shards of one utterance and EMPTY shards, both batch
    modes - the parity-mode gather moves row blocks + an index scatter (no all_reduce of zero-filled global tensors)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, batch)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sd = make_state_dict(3)
    mag, real, imag = make_spec(batch, 12, 42)
    want = (fsnp_torch.forward_full if mode == "full" else fsnp_torch.forward)(sd, mag, real, imag).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()
    assert [fdist.shard_bounds(batch, r, world) for r in range(world)].count((batch, batch)) + sum(1 for r in range(world) if fdist.shard_bounds(batch, r, world)[0] == fdist.shard_bounds(batch, r, world)[1]) >= world - batch
