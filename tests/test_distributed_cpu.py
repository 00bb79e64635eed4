"""World-size-2 gloo tests (CPU, two processes on 127.0.0.1) of the head-parallel exchange that C4 uses over RCCL:
token<->head all-to-all layouts, the text all-gather, head_parallel_attention end to end against a single-process
attention over the full sequence, and the query-group-sharded all-gather variant."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _full_inputs(b, s_img, s_txt, s_extra, a, d):
    g = torch.Generator().manual_seed(0)
    mk = lambda n: torch.randn(b, n, a, d, generator=g)
    return {k: (mk(s_img), mk(s_txt), mk(s_extra)) for k in "qkv"}


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chipmunk_amd import distributed as D
    D.setup_dist(dist.group.WORLD, rank, world)
    try:
        b, a, d, s_img, s_txt, s_extra = 1, 4, 16, 24, 6, 2
        full = _full_inputs(b, s_img, s_txt, s_extra, a, d)
        ls = s_img // world
        local = {}
        for key, (img, txt, extra) in full.items():
            local[key] = torch.cat([img[:, rank * ls:(rank + 1) * ls], txt, extra], dim=1)

        # ---- layout of the two all-to-alls
        qkv = torch.stack([local[k][:, :ls] for k in "qkv"])                  # [3, b, ls, a, d]
        got = D.all_to_all_collect_tokens(qkv)                                # [3, b, lh, s, d]
        lh = a // world
        want = torch.stack([full[k][0][:, :, rank * lh:(rank + 1) * lh].permute(0, 2, 1, 3) for k in "qkv"])
        assert torch.equal(got, want), "collect_tokens layout"
        back = D.all_to_all_collect_heads(got[0].contiguous())                # [b, ls, a*d]
        assert torch.equal(back, local["q"][:, :ls].reshape(b, ls, a * d)), "collect_heads inverts collect_tokens"

        # ---- head_parallel_attention == attention over the gathered sequence
        attn = lambda q, k, v: F.scaled_dot_product_attention(q, k, v)
        cu = [0, ls + s_txt]
        out = D.head_parallel_attention(attn, local["q"], local["k"], local["v"], ls, ls, cu, cu)
        qf = torch.cat([full["q"][0], full["q"][1]], dim=1).permute(0, 2, 1, 3)
        kf = torch.cat([full["k"][0], full["k"][1]], dim=1).permute(0, 2, 1, 3)
        vf = torch.cat([full["v"][0], full["v"][1]], dim=1).permute(0, 2, 1, 3)
        ref = F.scaled_dot_product_attention(qf, kf, vf).permute(0, 2, 1, 3)  # [b, s_img+s_txt, a, d]
        ref_img = ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d)
        ref_txt = ref[:, s_img:].reshape(b, s_txt, a * d)
        e = [full[k][2].permute(0, 2, 1, 3) for k in "qkv"]
        ref_extra = F.scaled_dot_product_attention(*e).permute(0, 2, 1, 3).reshape(b, s_extra, a * d)
        torch.testing.assert_close(out, torch.cat([ref_img, ref_txt, ref_extra], dim=1), rtol=1e-5, atol=1e-5)

        # ---- query-group sharding with K/V all-gather
        ql, kl, vl = [full[k][0][:, rank * ls:(rank + 1) * ls].permute(0, 2, 1, 3).contiguous() for k in "qkv"]
        o = D.group_parallel_attention(attn, ql, kl, vl)
        full_o = F.scaled_dot_product_attention(*[full[k][0].permute(0, 2, 1, 3) for k in "qkv"])
        torch.testing.assert_close(o, full_o[:, :, rank * ls:(rank + 1) * ls], rtol=1e-5, atol=1e-5)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_head_parallel_exchange_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_paths_without_group():
    from chipmunk_amd import distributed as D
    D.setup_dist(None, None, None)
    x = torch.randn(3, 1, 8, 4, 16)
    y = D.all_to_all_collect_tokens(x)
    assert y.shape == (3, 1, 4, 8, 16) and torch.equal(y[0, 0, 2, 5], x[0, 0, 5, 2])
    o = torch.randn(1, 4, 8, 16)
    assert torch.equal(D.all_to_all_collect_heads(o), o.permute(0, 2, 1, 3).reshape(1, 8, 64))
    assert D.all_gather(o) is o
