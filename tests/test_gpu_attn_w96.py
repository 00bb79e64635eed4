"""The one-wave-per-SIMD attention kernel (attn_w96_kernel: 2 waves x 96 query rows per workgroup, O^T accumulators and
four query blocks of Q^T in hand-named accumulator registers) forced on for the parity tests of tests/test_gpu_attn.py /
test_gpu_fullsize.py: same oracle comparisons and edge cases."""
import pytest
import torch

import test_gpu_attn as A
import test_gpu_fullsize as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev():
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd import _native
    _native.set_option("attn_w96", 1)
    yield torch.device("cuda:0")
    _native.set_option("attn_w96", 0)


@pytest.mark.parametrize("n", [384, 512, 1000, 1984])
def test_dense_vs_oracle(dev, n):
    A.test_dense_attn_vs_oracle_and_sdpa(dev, n)


def test_dense_strided(dev):
    A.test_dense_attn_strided_inputs(dev)


@pytest.mark.parametrize("n", [4480, 4592])
def test_identity_indices(dev, n):
    A.test_csp_attn_identity_indices_is_sdpa(dev, n)


@pytest.mark.parametrize("o_scale", [1, -1])
@pytest.mark.parametrize("n,count", [(960, 224), (1100, 336)])
def test_inplace(dev, n, count, o_scale):
    A.test_csp_attn_inplace_random_indices(dev, n, count, o_scale)


@pytest.mark.parametrize("o_scale", [1, -1])
def test_out_of_place(dev, o_scale):
    A.test_csp_attn_out_equals_clone_plus_inplace(dev, o_scale)


def test_key_split_forced(dev):
    A.test_csp_attn_key_split_forced(dev)


def test_strided_qkv(dev):
    A.test_csp_attn_strided_qkv(dev)


@pytest.mark.parametrize("n,nk,count", [(768, 768, 256), (1152, 1100, 384), (960, 960, 64), (960, 960, 32), (576, 576, 33)])
def test_csp_128(dev, n, nk, count):
    A.test_csp_128_attn_random_indices(dev, n, nk, count)


def test_roundtrip(dev):
    A.test_csp_attn_full_minus_sparse_roundtrip(dev)


def test_batched(dev):
    A.test_batched_inputs_all_attention_ops(dev)


def test_right_fill(dev):
    A.test_packed_positions_past_the_key_count_are_masked(dev)


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_running_max(dev, pattern):
    A.test_running_max_update_paths(dev, pattern)


def test_empty_and_ragged(dev):
    F.test_empty_and_ragged_groups(dev)


def test_duplicates(dev):
    F.test_duplicate_indices_are_not_deduplicated(dev)


def test_key_split_tail_at_scale(dev):
    F.test_key_split_tail_matches_unsplit_at_scale()


@pytest.mark.parametrize("H", [1, 3])
def test_sliced_heavy_items(dev, H):
    F.test_sliced_heavy_items_match_unsliced_and_oracle(H)


def test_c3_all_keys_equals_dense(dev):
    F.test_c3_sparse_attention_with_all_keys_equals_dense(dev)
