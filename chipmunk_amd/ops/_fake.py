"""Shape/dtype-only implementations of the returning ops for FakeTensor / torch.compile tracing.

The reference registers none, so every op is a graph break under ``torch.compile`` ("adding fake impls is
a free win").  The in-place ops need nothing (they return ``()``)."""
import torch


def _o_like(q, token_major):
    B, H, N, D = q.shape
    return q.new_empty((B, N, H, D)).permute(0, 2, 1, 3) if token_major else q.new_empty((B, H, N, D))


def _register():
    lib = torch.library

    @lib.register_fake("chipmunk::csp_128_attn")
    def _(q, k, v, indices, indices_counts):
        return torch.empty_like(q)

    @lib.register_fake("chipmunk::csp_attn_out")
    def _(q, k, v, o_in, indices, indices_counts, o_scale):
        return torch.empty_like(o_in)

    @lib.register_fake("chipmunk::residual_ln_modulate")
    def _(x, y, gate, shift, scale, eps):
        return [torch.empty_like(x), torch.empty_like(x)] if y is not None else [torch.empty_like(x)]

    @lib.register_fake("chipmunk::csp_attn_out_ragged")
    def _(q, k, v, o_in, indices, offsets, indices_counts, o_scale):
        return torch.empty_like(o_in)

    @lib.register_fake("chipmunk::dense_attn")
    def _(q, k, v):
        return [_o_like(q, False), q.new_empty((q.shape[0], q.shape[1], q.shape[2], 1), dtype=torch.float32)]

    @lib.register_fake("chipmunk::dense_attn_layout")
    def _(q, k, v, token_major_o):
        return [_o_like(q, token_major_o), q.new_empty((q.shape[0], q.shape[1], q.shape[2], 1), dtype=torch.float32)]

    @lib.register_fake("chipmunk::dense_colsum_attn_layout")
    def _(q, k, v, p, token_major_o):
        groups = (q.shape[2] + 191) // 192
        return [_o_like(q, token_major_o),
                q.new_empty((q.shape[0], q.shape[1], groups, max(q.shape[2], k.shape[2]))),
                q.new_empty((q.shape[0], q.shape[1], q.shape[2], 1), dtype=torch.float32)]

    @lib.register_fake("chipmunk::dense_colsum_attn")
    def _(q, k, v, p):
        groups = (q.shape[2] + 191) // 192
        return [_o_like(q, False),
                q.new_empty((q.shape[0], q.shape[1], groups, max(q.shape[2], k.shape[2]))),
                q.new_empty((q.shape[0], q.shape[1], q.shape[2], 1), dtype=torch.float32)]

    @lib.register_fake("chipmunk::mask_to_indices")
    def _(mask, multiple_of, pad_to_multiple_of):
        b, h, m, n = mask.shape
        pad_n = (n + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        return [mask.new_empty((b, h, m, pad_n), dtype=torch.int32), mask.new_empty((b, h, m), dtype=torch.int32)]

    @lib.register_fake("chipmunk::packed_mask_to_indices")
    def _(packed, shape, multiple_of, pad_to_multiple_of):
        b, h, m, n = shape
        pad_n = (n + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        return [packed.new_empty((b, h, m, pad_n), dtype=torch.int32), packed.new_empty((b, h, m), dtype=torch.int32)]

    @lib.register_fake("chipmunk::topk_mask")
    def _(cs, k, random_amount, groups, static_mask):
        return cs.new_empty(cs.shape, dtype=torch.bool)

    @lib.register_fake("chipmunk::block_mean")
    def _(x, mbm):
        return x.new_empty((x.shape[0], x.shape[1] // mbm, x.shape[2]))

    @lib.register_fake("chipmunk::quantize_fp8")
    def _(x, scale, max_value):
        return x.new_empty(x.shape, dtype=torch.float8_e4m3fn)

    @lib.register_fake("chipmunk::bitpack")
    def _(mask):
        return mask.new_empty(((mask.numel() + 7) // 8,), dtype=torch.uint8)

    @lib.register_fake("chipmunk::bitunpack")
    def _(packed, shape):
        return packed.new_empty(tuple(shape), dtype=torch.bool)

    @lib.register_fake("chipmunk::gather_rows")
    def _(src, map):
        shape = list(src.shape)
        shape[-2] = map.shape[0]
        return src.new_empty(shape)

    @lib.register_fake("chipmunk::dense_colsum_topk_mask")
    def _(q, k, v, p, k_top, random_amount, groups, static_mask, token_major_o=False):
        B, H, Nq, D = q.shape
        G = (Nq + 191) // 192
        return [_o_like(q, token_major_o), q.new_empty((B, H, G, k.shape[2]), dtype=torch.bool),
                q.new_empty((B, H, Nq, 1), dtype=torch.float32)]

    @lib.register_fake("chipmunk::qkv_split_norm")
    def _(qkv, q_weight, k_weight, heads, eps, freqs_cos=None, freqs_sin=None):
        out = qkv.new_empty((3, 1, heads, qkv.shape[0], 128))
        return [out[0], out[1], out[2]]


_register()
