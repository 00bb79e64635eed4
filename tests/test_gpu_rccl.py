"""RCCL (torch.distributed backend "nccl" on ROCm) over xGMI: the head-parallel exchange on real GPUs, one process per
GPU.  Needs >= 2 GPUs on the node -- skipped on the single-GPU boxes of the test pool (the same code paths run under
gloo on CPU in tests/test_distributed_cpu.py)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import chipmunk_amd  # noqa: F401
        from chipmunk_amd import distributed as D
        D.setup_dist(dist.group.WORLD, rank, world)
        b, a, d, s_img, s_txt = 1, 8, 128, 192 * 4 * world, 64
        ls, lh = s_img // world, a // world
        g = torch.Generator().manual_seed(3)
        img = torch.randn(3, b, s_img, a, d, generator=g).to(torch.bfloat16)
        txt = torch.randn(3, b, s_txt, a, d, generator=g).to(torch.bfloat16)
        qf, kf, vf = [torch.cat([img[i], txt[i]], dim=1).permute(0, 2, 1, 3).contiguous().to(dev) for i in range(3)]
        ref, _ = torch.ops.chipmunk.dense_attn(qf, kf, vf)                      # all heads, whole sequence, one GPU
        ref = ref.permute(0, 2, 1, 3)
        for overlap in (True, False):
            pipe = D.HeadParallelPipeline(dist.group.WORLD, a, ls, s_txt, d, torch.bfloat16, dev, chunk_heads=1, overlap=overlap)
            attn = [lambda q, k, v: torch.ops.chipmunk.dense_attn(q.contiguous(), k.contiguous(), v.contiguous())[0]] * lh
            for _ in range(2):
                o_img, o_txt = pipe.run(img[:, :, rank * ls:(rank + 1) * ls].contiguous().to(dev), txt.to(dev), attn)
            torch.cuda.synchronize()
            assert torch.equal(o_img, ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d)), f"image rows (overlap={overlap})"
            assert torch.equal(o_txt, ref[:, s_img:].reshape(b, s_txt, a * d)), f"text rows (overlap={overlap})"
        # the reference-named serial exchange gives the same bits
        local = [torch.cat([img[i][:, rank * ls:(rank + 1) * ls], txt[i]], dim=1).to(dev) for i in range(3)]
        cu = [0, ls + s_txt]
        out = D.head_parallel_attention(lambda q, k, v: torch.ops.chipmunk.dense_attn(q.contiguous(), k.contiguous(), v.contiguous())[0],
                                        local[0], local[1], local[2], ls, ls, cu, cu)
        assert torch.equal(out[:, :ls], ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_head_parallel_pipeline_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on the node (RCCL over xGMI)")
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
