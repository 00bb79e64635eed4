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


def _same(a, b):
    """a sharded launch (fewer heads / rows per call) may take another split of the key range than the all-heads reference call
    and round differently in the last bf16 place; a layout or synchronisation error is off by O(1)"""
    return torch.allclose(a.float(), b.float(), atol=8e-3, rtol=8e-3)


def _worker(rank, world, port, ret, share_gpu=False):
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0 if share_gpu else rank)
    dev = torch.device("cuda", 0 if share_gpu else rank)
    if share_gpu:   # rehearsal on one device: gloo, device tensors staged through host memory (distributed._staged)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
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
            assert _same(o_img, ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d)), f"image rows (overlap={overlap})"
            assert _same(o_txt, ref[:, s_img:].reshape(b, s_txt, a * d)), f"text rows (overlap={overlap})"
        # the reference-named serial exchange gives the same bits
        local = [torch.cat([img[i][:, rank * ls:(rank + 1) * ls], txt[i]], dim=1).to(dev) for i in range(3)]
        cu = [0, ls + s_txt]
        out = D.head_parallel_attention(lambda q, k, v: torch.ops.chipmunk.dense_attn(q.contiguous(), k.contiguous(), v.contiguous())[0],
                                        local[0], local[1], local[2], ls, ls, cu, cu)
        assert _same(out[:, :ls], ref[:, rank * ls:(rank + 1) * ls].reshape(b, ls, a * d))
        # query-group sharding (K/V all-gather per head chunk), uneven rows: whole 192-row groups, the text rows on the last rank
        n_tok = s_img + s_txt
        rows = D.group_rows(n_tok, world)
        lo = sum(rows[:rank])
        gp = D.GroupParallelPipeline(dist.group.WORLD, a, rows, d, torch.bfloat16, dev, chunks=[3, 5])
        dense = lambda q, k, v: torch.ops.chipmunk.dense_attn(q.contiguous(), k.contiguous(), v.contiguous())[0]
        for _ in range(2):
            og = gp.run(qf[:, :, lo:lo + rows[rank]].contiguous(), kf[:, :, lo:lo + rows[rank]].contiguous(),
                        vf[:, :, lo:lo + rows[rank]].contiguous(), [dense, dense])
        torch.cuda.synchronize()
        assert _same(og, ref[:, lo:lo + rows[rank]].reshape(b, rows[rank], a * d)), "query-group sharding"
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


def test_sharded_pipelines_two_ranks_on_one_gpu():
    """The same two-rank program as the RCCL test with both ranks on cuda:0 and the collectives staged through host memory over
    gloo (RCCL refuses two ranks on one device): every line of the pipelines except the transport itself runs on a real GPU,
    with the HIP attention kernels, on the one-GPU boxes of the test pool -- bit-identical to the unsharded dense attention."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, True), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_bench_two_ranks_over_rccl_reports_transport_and_exposed_communication():
    """`bench.py --gpus 2` on the real transport (>= 2 GPUs): the launcher, RCCL initialisation with its recorded settings, the
    chunk planner fed by a measured exchange, and the exposed-communication probe of the sparse steps -- asserted, not just run
    (on the one-GPU boxes the same line is rehearsed over gloo by tests/test_gpu_bench_contract.py)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on the node (RCCL over xGMI)")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    first = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-only"], capture_output=True, text=True, timeout=600)
    assert first.returncode == 0, first.stderr[-2000:]
    lo = json.loads(first.stdout.strip().splitlines()[-1])
    assert lo["backend"].startswith("nccl") and lo["rank_sum"] == 1 and lo["rccl"]["init"].startswith("nccl")
    assert lo["collectives"]["all_to_all_single"]["payload_ok"] and lo["collectives"]["all_gather_into_tensor"]["payload_ok"]
    for mode in ("heads", "groups"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--sp-mode", mode, "--grid", "8,12,16", "--layers", "3",
                            "--steps", "3", "--warmup", "3"], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["config"]["sp_mode"] == mode
        assert d["config"]["rccl"]["init"].startswith("nccl") and d["config"]["rccl"]["rccl_version"]
        assert "rehearsal" not in d["config"] and d["value"] > 0
        assert 0.0 <= d["exposed_comm"]["exposed_fraction"] <= 1.0
