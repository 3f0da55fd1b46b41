"""Data-parallel host logic on CPU with gloo, world_size 2 (the N>1 path of bench.py / GDTrainer):
gradient slab reduction (mean, bucketed), the ITC all-gather with its slice-only backward (reference
efficient_models/xvlm.py:54-74), parameter broadcast, and rank-sharded synthetic batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _reduce_case(rank, world):
    from efficientvlm_amd.trainer import GradReducer
    torch.manual_seed(0)
    base = [torch.randn(1000), torch.randn(37), torch.randn(4096)]
    flats = [b.clone() * (rank + 1) for b in base]            # rank r holds (r+1) * base
    red = GradReducer(flats, bucket_bytes=1024)                # forces several buckets per slab
    assert red.world == world and len(red.buckets) > 3
    red.reduce()
    exp = [b * (sum(range(1, world + 1)) / world) for b in base]
    return all(torch.allclose(f, e, rtol=1e-6, atol=1e-6) for f, e in zip(flats, exp))


def test_grad_reducer_means_flat_slabs_over_ranks():
    assert run2(_reduce_case) == [True, True]


def _allgather_case(rank, world):
    from efficientvlm_amd.efficient_models.xvlm import allgather
    from oracle import xvlm_oracle as O
    g = torch.Generator().manual_seed(123)
    B, E = 3, 8
    img_all = torch.nn.functional.normalize(torch.randn(world * B, E, generator=g), dim=-1)
    txt_all = torch.nn.functional.normalize(torch.randn(world * B, E, generator=g), dim=-1)
    img = img_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    txt = txt_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    gi, gt = allgather(img), allgather(txt)
    ok = torch.equal(gi.detach(), img_all) and torch.equal(gt.detach(), txt_all)
    temp = torch.tensor(0.07)
    loss = O.contrastive_loss(gi, gt, temp)                    # oracle as the checker of the gathered ITC
    ref_i = img_all.clone().requires_grad_(True)
    ref_t = txt_all.clone().requires_grad_(True)
    ref = O.contrastive_loss(ref_i, ref_t, temp)
    ok = ok and torch.allclose(loss, ref)
    loss.backward()
    ref.backward()
    # the backward keeps ONLY this rank's slice of the gathered gradient (no cross-rank reduction)
    ok = ok and torch.allclose(img.grad, ref_i.grad[rank * B:(rank + 1) * B], atol=1e-7)
    ok = ok and torch.allclose(txt.grad, ref_t.grad[rank * B:(rank + 1) * B], atol=1e-7)
    # idx all-gather (soft ITC labels)
    idx = torch.arange(B).view(-1, 1) + rank * B
    ok = ok and torch.equal(allgather(idx).view(-1), torch.arange(world * B))
    return bool(ok)


def test_itc_allgather_forward_and_slice_only_backward():
    assert run2(_allgather_case) == [True, True]


def _shard_case(rank, world):
    from oracle import synth
    geom = synth.GEOMS["tiny"]
    b = synth.make_batch(geom, 4, seed=42 + rank)              # bench.py: weak scaling, seed 42 + rank
    t = b["image"].sum().reshape(1).clone()
    outs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(outs, t)
    # broadcast of a parameter slab from rank 0 (GDTrainer start-up, apex_ddp_accelerator.py:75-77)
    p = torch.full((10,), float(rank + 1))
    dist.broadcast(p, 0)
    return (float(outs[0]) != float(outs[1])) and bool((p == 1.0).all())


def test_rank_sharded_batches_and_param_broadcast():
    assert run2(_shard_case) == [True, True]


def test_single_process_allgather_is_identity():
    from efficientvlm_amd.efficient_models.xvlm import allgather
    x = torch.randn(4, 3)
    assert allgather(x) is x


def _segment_case(rank, world):
    """early / late segment reduction (overlap path) gives the same result as one full reduce"""
    from efficientvlm_amd.trainer import GradReducer
    torch.manual_seed(1)
    base = torch.randn(5000)
    flat = [base.clone() * (rank + 1)]
    red = GradReducer(flat, bucket_bytes=4096)
    red.reduce_async([flat[0][:1234]])       # "early" part, launched from the backward hook
    red.reduce_async([flat[0][1234:]])       # "late" part, after backward
    red.finish()
    return bool(torch.allclose(flat[0], base * (sum(range(1, world + 1)) / world), rtol=1e-6, atol=1e-6))


def test_segmented_overlap_reduction_equals_full_reduce():
    assert run2(_segment_case) == [True, True]
