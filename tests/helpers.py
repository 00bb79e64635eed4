"""Shared helpers for the parity tests: seeded synthetic inputs in the reference's shapes, index generators."""
import torch


def randn_bf16(*shape, seed=0, device="cpu", scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(device)


def random_index_sets(B, H, G, n_keys, count, width, seed=0, multiple_of=1):
    """Per (b,h,g) a random sorted subset of `count` distinct columns of range(n_keys); rows padded to `width` with -1."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    inds = torch.full((B, H, G, width), -1, dtype=torch.int32)
    counts = torch.full((B, H, G), count, dtype=torch.int32)
    for b in range(B):
        for h in range(H):
            for q in range(G):
                perm = torch.randperm(n_keys, generator=g)[:count].sort().values
                inds[b, h, q, :count] = perm.to(torch.int32)
    assert count % multiple_of == 0
    return inds, counts


def assert_close_bf16(a, b, atol=2e-2, rtol=2e-2, what=""):
    a32, b32 = a.float().cpu(), b.float().cpu()
    diff = (a32 - b32).abs()
    tol = atol + rtol * b32.abs()
    bad = diff > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} elements off, max abs diff {diff.max().item():.4g}"
