/*
 * chipmunk_hip.h -- C ABI of the MI355X (gfx950) column-sparse DiT hot path.
 *
 * One entry point per operator of the reference's PyTorch operator surface
 * (`TORCH_LIBRARY(chipmunk)`, reference csrc/chipmunk.cpp:45-60).  Plain pointers and sizes only:
 * no torch types cross this boundary.  The torch-side registration that turns these back into
 * `torch.ops.chipmunk.*` lives in chipmunk_amd/csrc/torch_registry.cpp; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated otherwise; bf16 tensors are raw 16-bit storage;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); calls only enqueue work;
 *   - return value 0 = success; non-zero = error, message available from chipmunk_last_error();
 *     nothing is ever written to stderr and the process is never exited (the reference exit()s in
 *     csrc/mlp/csp_mlp_mm2_and_scatter_add.cu:15-42; that behaviour is deliberately not reproduced);
 *   - strides are in ELEMENTS for dims (batch, head, token); the head dimension (128) is contiguous.
 */
#ifndef CHIPMUNK_HIP_H
#define CHIPMUNK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHIPMUNK_OK 0
#define CHIPMUNK_ERR_INVALID 1 /* bad shape / argument (the reference raises TORCH_CHECK / std::runtime_error) */
#define CHIPMUNK_ERR_LAUNCH 2  /* HIP launch failure */
#define CHIPMUNK_ERR_UNSUPPORTED 3 /* a fused entry point that does not apply to this launch: nothing was enqueued, use the unfused operators */

/* element types for the dtype-generic ops (reference dispatches bf16/fp16/fp32: topk_indices.cu:171-215) */
#define CHIPMUNK_DTYPE_BF16 0
#define CHIPMUNK_DTYPE_FP16 1
#define CHIPMUNK_DTYPE_FP32 2

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *chipmunk_last_error(void);
/* ABI version of this library (bumped on signature changes). */
int chipmunk_abi_version(void);
/* Tuning knob: selects a kernel variant ("mm1_variant", "mm2_variant", "attn_variant", ...); 0 = shipped default.
 * Results are identical across variants; only speed differs.  Used by tools/kbench.py for A/B measurement. */
int chipmunk_set_option(const char *name, int value);
/* Seed of the random-key hash of chipmunk_topk_indices / _topk_delta_indices / _topk_mask (`random_amount` > 0).
 * Every such launch draws a fresh set (per-launch salt = f(seed, launch counter), mixed on the device with the first
 * element of each row); setting the seed restarts the sequence, so seed + launch order reproduce a run.  Replaces the
 * reference's per-call cuRAND reseed (csrc/indexed_io/topk_indices.cu:46-49) and torch.randint (modules/attn.py:77). */
int chipmunk_set_random_seed(uint64_t seed);

/* ---------------------------------------------------------------- column-sparse attention
 * Replaces chipmunk::csp_attn (reference csrc/attn/csp_attn.cu:315-423; schema csrc/chipmunk.cpp:52).
 * For every (b, h, 192-query group g): J = indices[b,h,g,0:counts[b,h,g]];
 *   o[b,h,i,:] += o_scale * sum_{j in J} softmax_j(q_i.k_j / sqrt(128)) v_j     for i in group g
 * q,o: [B,H,Nq,128] bf16 (strided); k,v: [B,H,Nk,128] bf16 (strided); indices [B,H,G,idx_stride] int32
 * (G = ceil(Nq/192), row stride idx_stride >= max count); counts [B,H,G] int32; o_scale in {1,-1}.
 * The accumulate is bf16: o_new = bf16(o_old + bf16(o_scale*result)) (csp_attn.cu:294-300). */
int chipmunk_csp_attn(const void *q, const void *k, const void *v, void *o, const int64_t q_strides[3],
                      const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
                      const int32_t *indices, const int32_t *counts, int B, int H, int Nq, int Nk, int idx_stride,
                      int o_scale, void *stream);

/* Out-of-place form of chipmunk_csp_attn: o_out = o_in + bf16(o_scale * sparse_attention), o_in untouched; o_in and
 * o_out share `o_strides`.  It replaces the `o = out_cache.clone(); csp_attn(q, k, v, o, ...)` pair of the reference's
 * sparse step (src/chipmunk/modules/attn.py:186-188): same bytes, one kernel, no 2 x B*H*N*128*2-byte copy. */
int chipmunk_csp_attn_out(const void *q, const void *k, const void *v, const void *o_in, void *o_out,
                          const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                          const int64_t o_strides[3], const int32_t *indices, const int32_t *counts, int B, int H,
                          int Nq, int Nk, int idx_stride, int o_scale, void *stream);

/* chipmunk_csp_attn_out over RAGGED index rows (no reference counterpart; the reference's index tensor is [B,H,G,Nk] int32 --
 * 7 GB per HunyuanVideo layer for 0.56 GB of kept keys, which is why its modules store a bit-packed mask and rebuild the indices
 * in every step, modules/attn.py:95-100,161): row (b,h,g) = indices + idx_offsets[(b*H+h)*G+g], idx_offsets[item+1] -
 * idx_offsets[item] entries wide (B*H*G+1 int64 offsets, each a multiple of 4; indices 16-byte aligned); counts as before.
 * chipmunk_compact_indices builds that layout from a padded tensor: flat[offsets[r] + j] = indices[r*idx_stride + j] for
 * j < counts[r], 0 up to offsets[r+1]. */
int chipmunk_csp_attn_out_ragged(const void *q, const void *k, const void *v, const void *o_in, void *o_out,
                                 const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                                 const int64_t o_strides[3], const int32_t *indices, const int64_t *idx_offsets,
                                 const int32_t *counts, int B, int H, int Nq, int Nk, int o_scale, void *stream);
int chipmunk_compact_indices(const int32_t *indices, int64_t idx_stride, const int32_t *counts, const int64_t *offsets,
                             int32_t *flat, int64_t rows, void *stream);

/* Replaces chipmunk::csp_128_attn (reference csrc/attn/csp_128_attn.cu:355-461; schema csrc/chipmunk.cpp:53).
 * Same math, out of place into `o` (contiguous [B,H,Nq,128] bf16, fully overwritten), contiguous q/k/v. */
int chipmunk_csp_128_attn(const void *q, const void *k, const void *v, void *o, const int32_t *indices,
                          const int32_t *counts, int B, int H, int Nq, int Nk, int idx_stride, void *stream);

/* Replaces chipmunk::dense_attn (reference csrc/attn/dense_attn.cu:246-372; schema csrc/chipmunk.cpp:54).
 * o [B,H,Nq,128] bf16 contiguous; l [B,H,Nq] fp32 = 1 / sum_j exp(q_i.k_j / sqrt(128)) (dense_attn.cu:225-227). */
int chipmunk_dense_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                        const int64_t k_strides[3], const int64_t v_strides[3], void *o, float *l, int B, int H,
                        int Nq, int Nk, void *stream);

/* The dense operators with an output layout: o_strides = (batch, head, row) element strides of `o`, rows of 128 contiguous
 * elements (NULL = contiguous [B,H,Nq,128], i.e. the entries above).  Token-major storage [B,Nq,H,128] -- strides
 * {Nq*H*128, 128, H*128} -- is what the model's next GEMM reads: the reference's `rearrange(o, "b h s d -> b s (h d)")`
 * (a 2 x B*H*Nq*256-byte copy after every attention, e.g. hunyuan/modules/models.py:264) becomes a view.  The sparse step
 * keeps the layout by itself: chipmunk_csp_attn / chipmunk_csp_attn_out take o_strides already. */
int chipmunk_dense_attn_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                const int64_t k_strides[3], const int64_t v_strides[3], void *o, const int64_t *o_strides,
                                float *l, int B, int H, int Nq, int Nk, void *stream);
int chipmunk_dense_colsum_attn_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                       const int64_t k_strides[3], const int64_t v_strides[3], const float *p, void *o,
                                       const int64_t *o_strides, void *cs, float *l, int B, int H, int Nq, int Nk,
                                       int cs_stride, void *stream);
int chipmunk_dense_colsum_topk_mask_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                            const int64_t k_strides[3], const int64_t v_strides[3], const float *p, void *o,
                                            const int64_t *o_strides, float *l, int B, int H, int Nq, int Nk,
                                            const void *static_mask, int64_t static_stride, int static_rows,
                                            const void *group_flags, void *mask, int k_top, double random_amount, void *stream);

/* Replaces chipmunk::dense_colsum_attn (reference csrc/attn/dense_colsum_attn.cu:521-668; schema chipmunk.cpp:55).
 * p [B,H,Nq] fp32 = previous step's l.  cs [B,H,ceil(Nq/192),cs_stride] bf16:
 *   cs[b,h,g,j] = sum_{i in group g} bf16(exp(s_ij - m_i)) * bf16(exp(m_i) * p_i)   for j < Nk
 * (dense_colsum_attn.cu:267-277); columns j >= Nk are left untouched.
 * Launches that fill the CUs run as ONE pass (the column sums ride the dense kernel's softmax pipeline) and keep, per
 * (device, stream), a library-owned buffer of bf16 partial sums of B*H*ceil(Nq/256)*2*Nk*2 bytes (halved per chunk of
 * heads if that cannot be allocated; bounded, see chipmunk_release_scratch); the first call on a stream allocates it, so call
 * once before capturing a graph. */
int chipmunk_dense_colsum_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                               const int64_t k_strides[3], const int64_t v_strides[3], const float *p, void *o,
                               void *cs, float *l, int B, int H, int Nq, int Nk, int cs_stride, void *stream);

/* Row-wise pass of the block around the operators (torch ops in the reference's model code, e.g.
 * examples/hunyuan/hyvideo/modules/models.py:184-186, 262-275): with y, gate, x_out set
 *   x_out = bf16(x + gate * y);  xm = bf16(shift + bf16(LayerNorm(x_out)) * bf16(1 + scale))
 * and with y = gate = x_out = NULL the LayerNorm + modulate of x alone.  x, y, x_out, xm [rows, cols] bf16 contiguous (x_out may
 * be x); gate, shift, scale [cols] bf16; LayerNorm over cols without affine, statistics in fp32 (two passes over the row in
 * registers).  cols % 8 == 0, cols <= 8192, 16-byte aligned pointers.  One read of x and y, one write of x_out and xm. */
int chipmunk_residual_ln_modulate(const void *x, const void *y, const void *gate, const void *shift, const void *scale,
                                  void *x_out, void *xm, int64_t rows, int cols, double eps, void *stream);

/* ---------------------------------------------------------------- column-sparse MLP
 * Replaces chipmunk::csp_mlp_mm1 (reference csrc/mlp/csp_mlp_mm1.cu:625-702; schema csrc/chipmunk.cpp:47).
 * For 128-row group g and packed column j < counts[g]:
 *   c[m,j] = bf16( gelu_tanh(a[m,:].b[idx[g,j],:] + bias[idx[g,j]]) - pa_cache[idx[g,j], m] )
 * a [M,K], b [F,K] (= fc1.weight), c [M,F] (packed; columns >= counts[g] untouched), bias [F],
 * pa_cache [F,M] (column-major activation cache), indices [M/128,F] int32, counts [M/128] int32; all bf16.
 * Requires M % 128 == 0, K % 64 == 0 (csp_mlp_mm1.cu:215,230); counts need only be multiples of 16. */
int chipmunk_csp_mlp_mm1(const void *a, const void *b, void *c, const void *bias, const void *pa_cache,
                         const int32_t *indices, const int32_t *counts, int M, int K, int F, void *stream);

/* GEMM1 that also applies the scatter-add of its own output: after computing c as above,
 *   pa_cache[idx[g,j], m] = bf16(pa_cache[idx[g,j], m] + c[m,j])          (== chipmunk_csp_scatter_add(c, pa_cache, ...))
 * from the cache block the epilogue already holds in LDS.  Bit-identical to chipmunk_csp_mlp_mm1 followed by
 * chipmunk_csp_scatter_add; saves that kernel's launch and its re-read of c and the cache.  Each (group, column) block of
 * the cache is read and written by exactly one workgroup. */
int chipmunk_csp_mlp_mm1_scatter(const void *a, const void *b, void *c, const void *bias, void *pa_cache,
                                 const int32_t *indices, const int32_t *counts, int M, int K, int F, void *stream);

/* fp8 (OCP e4m3fn) GEMM1 for BASELINE config C5: native counterpart of the reference's Triton csp_mlp_mm1_fp8
 * (src/chipmunk/triton/csp_mlp_mm1.py:37-164).  a [M,K] fp8, b [F,K] fp8; scale_a / scale_b: device pointers to ONE
 * fp32 each (the reference passes 0-dim tensors holding the RECIPROCAL quantisation scales, modules/mlp.py:98-99):
 *   x = bf16(gelu_tanh(a.b[idx] * scale_a * scale_b + bias[idx]));  c[m,j] = bf16(x - pa_cache[idx, m])
 * update_cache = 1 additionally stores x into pa_cache like the Triton kernel (:140); 0 leaves the cache to the
 * scatter-add (the bf16 path's contract); 2 applies that scatter-add itself -- pa_cache[idx, m] += c[m, j] in bf16, bit for
 * bit what chipmunk_csp_scatter_add would do afterwards (as chipmunk_csp_mlp_mm1_scatter on the bf16 path).
 * Requires K % 128 == 0. */
int chipmunk_csp_mlp_mm1_fp8(const void *a, const void *b, void *c, const void *bias, void *pa_cache,
                             const int32_t *indices, const int32_t *counts, const float *scale_a,
                             const float *scale_b, int M, int K, int F, int update_cache, void *stream);

/* Replaces chipmunk::csp_mlp_mm2_and_scatter_add (reference csrc/mlp/csp_mlp_mm2_and_scatter_add.cu:96-259 plus the
 * Triton GEMM src/chipmunk/triton/csp_mlp_mm2.py:26-129; schema csrc/chipmunk.cpp:48).
 *  (i)  unpacked_colmajor[idx[g,c], g*128 + r] += packed[g*128 + r, c]           (bf16 add, c < counts[g])
 *  (ii) mma_c[m,:] = bf16(sum_{c<counts[g]} mma_a[m,c] * mma_b[idx[g,c],:]) + mma_c[m,:]
 * packed, mma_a [M,F]; unpacked_colmajor [F,M]; mma_b [F,N2] (= fc2.weight.T contiguous); mma_c [M,N2].
 * `num_sms_scatter_add` is the reference's SM-partition hint (kept for signature parity; ignored). */
int chipmunk_csp_mlp_mm2_and_scatter_add(const void *packed, void *unpacked_colmajor, const int32_t *indices,
                                         const int32_t *counts, const void *mma_a, const void *mma_b, void *mma_c,
                                         int M, int F, int N2, int num_sms_scatter_add, void *stream);

/* Replaces chipmunk::csp_scatter_add (reference csrc/indexed_io/scatter_add.cu:102-181; schema chipmunk.cpp:59): (i) only. */
int chipmunk_csp_scatter_add(const void *packed, void *unpacked_colmajor, const int32_t *indices,
                             const int32_t *counts, int M, int F, int num_sms, void *stream);

/* GEMM (ii) alone: the native counterpart of the reference's Triton csp_mlp_mm2 (triton/csp_mlp_mm2.py:104-129). */
int chipmunk_csp_mlp_mm2(const void *mma_a, const void *mma_b, void *mma_c, const int32_t *indices,
                         const int32_t *counts, int M, int F, int N2, void *stream);

/* ---------------------------------------------------------------- indexed IO
 * Replaces chipmunk::topk_indices (reference csrc/indexed_io/topk_indices.cu:145-218; schema chipmunk.cpp:58).
 * activation [B*R, C] of `dtype`; indices [B*R, C] int32; counts [B*R] int32.  Threshold = element
 * int(1024*sparsity) of the ascending-sorted first 1024 values of the row (:94-101); keep x >= threshold.
 * Deterministic canonical order (a refinement of the reference's atomics-dependent order): kept columns
 * ascending, then padding up to `multiple_of` with "last rejected column of residue t (mod 1024)" for ascending t.
 * random_amount > 0 (cuRAND in the reference) keeps extra columns chosen by a counter-based hash. */
int chipmunk_topk_indices(const void *activation, int dtype, int32_t *indices, int32_t *counts, int rows, int cols,
                          double sparsity_amount, int multiple_of, double random_amount, void *stream);

/* Fused |activation - cache| -> topk_indices -> copy_indices for the sparse-MLP step with bm == mbm (reference
 * src/chipmunk/modules/mlp.py:70-85): ranks the element-wise |delta| (rounded to the tensor dtype like the eager ops),
 * writes indices / counts exactly like chipmunk_topk_indices on that delta, and copies every selected column of
 * `activation` into `cache`.  One kernel instead of abs + sum + topk_indices + copy_indices. */
int chipmunk_topk_delta_indices(const void *activation, void *cache, int dtype, int32_t *indices, int32_t *counts,
                                int rows, int cols, double sparsity_amount, int multiple_of, double random_amount,
                                void *stream);

/* chipmunk_dense_colsum_attn + chipmunk_topk_mask without the cs tensor between them (reference modules/attn.py:131-141 calls
 * dense_colsum_attn, then random_and_topk on its 3.55 GB result at HunyuanVideo size): o, l as chipmunk_dense_colsum_attn,
 * mask [B*H*ceil(Nq/192), Nk] bool bytes as chipmunk_topk_mask would produce from that call's cs -- bit for bit.
 * Returns CHIPMUNK_ERR_UNSUPPORTED without enqueuing anything when the launch would not take the one-pass route in one piece
 * (small launches, Nk % 4 != 0, no room for the partial sums): run the two operators then. */
int chipmunk_dense_colsum_topk_mask(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                    const int64_t k_strides[3], const int64_t v_strides[3], const float *p, void *o, float *l,
                                    int B, int H, int Nq, int Nk, const void *static_mask, int64_t static_stride, int static_rows,
                                    const void *group_flags, void *mask, int k_top, double random_amount, void *stream);

/* Fused mask construction of the attention mask-building step (reference
 * src/chipmunk/modules/attn.py:76-82 `random_and_topk`):
 *   mask[r, c] = ((c in topk_k(cs[r, :n])) | (u(r, c) < random_amount)) & group_flags[r]  |  static_mask[r % static_rows, c]
 * cs [rows, cs_stride] bf16 column sums; static_mask (optional) bool bytes with row stride static_stride, broadcast over
 * rows modulo static_rows; group_flags (optional) one byte per row; mask [rows, n] bool bytes, fully overwritten.
 * Exactly k columns per active row come from the top-k part; ties at the k-th value are broken deterministically (the
 * reference's torch.topk leaves that choice unspecified); u is a counter-based hash (RNG streams cannot match torch's
 * randint), so results are comparable with the reference chain for random_amount = 0.  n <= 122 880. */
int chipmunk_topk_mask(const void *cs, int64_t cs_stride, const void *static_mask, int64_t static_stride,
                       int static_rows, const void *group_flags, void *mask, int rows, int n, int k,
                       double random_amount, void *stream);

/* Replaces chipmunk::mask_to_indices (reference csrc/indexed_io/mask_to_indices.cu:92-143; schema chipmunk.cpp:60).
 * mask [rows, n] bool bytes; indices [rows, pad_n] int32; counts [rows] int32.  Bit-exact order: True columns of
 * residue class t (mod 32) ascending for t = 0..31, then the first False columns ascending up to multiple_of. */
int chipmunk_mask_to_indices(const void *mask, int32_t *indices, int32_t *counts, int64_t rows, int n, int pad_n,
                             int multiple_of, void *stream);

/* Fused variant: same output from the bit-packed mask (reference ops/bitpack.py:4-40 layout,
 * little-endian, flat) without materialising the bool mask.  Requires n % 8 == 0. */
int chipmunk_packed_mask_to_indices(const void *packed, int32_t *indices, int32_t *counts, int64_t rows, int n,
                                    int pad_n, int multiple_of, void *stream);

/* Same kept set, counts and padding columns as chipmunk_mask_to_indices, but the kept columns come out ASCENDING
 * (packed != 0: input is the bit-packed mask).  Not the reference's order; for consumers that only need the set
 * (the attention kernels): ascending keys make the K/V gather walk DRAM pages in order. */
int chipmunk_mask_to_sorted_indices(const void *mask, int packed, int32_t *indices, int32_t *counts, int64_t rows,
                                    int n, int pad_n, int multiple_of, void *stream);

/* Replaces chipmunk::copy_indices (reference csrc/indexed_io/copy_indices.cu:82-154; schema chipmunk.cpp:57).
 * dst[b,row,idx] = src[b,row,idx] for the first counts[b,row/R] entries of inds[b,row/R,:]; elem_size 2 or 4. */
int chipmunk_copy_indices(const void *src, void *dst, const int32_t *inds, const int32_t *counts, int B, int M, int R,
                          int F, int elem_size, void *stream);

/* [B,R,C] -> [B,C,R] for 16-bit elements: builds the column-major activation cache `pa.transpose(-1,-2).contiguous()`
 * of the sparse MLP's full step (reference src/chipmunk/modules/mlp.py:56) at HBM rate. */
int chipmunk_transpose16(const void *src, void *dst, int B, int R, int C, void *stream);

/* [rows, C] bf16 -> [rows / mbm, C] bf16: mean over consecutive blocks of `mbm` rows, fp32 sums, one rounding (the caller's
 * `block_mean(x, mbm)` in front of the fc1 probe of every sparse MLP step, src/chipmunk/modules/mlp.py:11-16,62).  rows % mbm == 0,
 * mbm % 4 == 0, C % 8 == 0. */
int chipmunk_block_mean(const void *x, void *out, int64_t rows, int C, int mbm, void *stream);

/* bf16 [n] -> OCP fp8 e4m3 [n]: F8Linear.quantize_input's `(x * scale).clamp(-max, max).to(float8_e4m3fn)` (reference
 * src/chipmunk/modules/mlp_fp8.py, the input side of csp_mlp_mm1_fp8) as one pass with the same roundings (fp32 product -> bf16 -> clamp ->
 * e4m3, round-to-nearest-even).  scale: one float on the device.  n % 8 == 0. */
int chipmunk_quantize_fp8(const void *x, const float *scale, void *out, int64_t n, float max_value, void *stream);

/* bitpack / bitunpack (reference src/chipmunk/ops/bitpack.py:4-69): 8 bools -> 1 byte, little-endian, flat. */
int chipmunk_bitpack(const void *mask, void *packed, int64_t n, void *stream);
int chipmunk_bitunpack(const void *packed, void *mask, int64_t n, void *stream);

/* ---------------------------------------------------------------- token reorder
 * dst[o, i, :] = src[o, map[i], :] for o < outer, i < n_out; rows of `row_bytes` bytes, src has n_src rows per outer
 * index.  One gather for each of the reference's token reorders -- patchify / unpatchify / patchify_rope
 * (src/chipmunk/ops/patch.py:7-80) and voxel_chunk_no_padding / reverse_voxel_chunk_no_padding
 * (src/chipmunk/ops/voxel.py:9-99) -- whose permutations the host side computes once per shape. `map` int32 [n_out]. */
int chipmunk_gather_rows(const void *src, void *dst, const int32_t *map, int64_t outer, int64_t n_src, int64_t n_out,
                         int64_t row_bytes, void *stream);

/* ---------------------------------------------------------------- pinned host offload pool
 * Replaces the reference's per-layer pinned CPU buffers and its two module-level CUDA streams' copies
 * (src/chipmunk/util/storage/offloaded_tensor.py:12-13 the streams, :42-44,71 `torch.empty(..., pin_memory=True)`, :104-118 and
 * :140-160 `copy_(non_blocking=True)`): hipHostMalloc'd buffers owned by the library and hipMemcpyAsync on the caller's side stream.
 * chipmunk_host_alloc: page-locked host memory (hipHostMallocDefault); chipmunk_host_free gives it back (the caller makes sure no copy
 * on it is in flight).  chipmunk_copy_d2h_async / _h2d_async: one hipMemcpyAsync of `bytes` contiguous bytes on `stream` (a dense
 * permuted tensor travels as its storage); ordering against other streams is the caller's (events), as in the reference.
 * chipmunk_host_bytes: bytes currently allocated through chipmunk_host_alloc (diagnostic). */
int chipmunk_host_alloc(size_t bytes, void **host_ptr);
int chipmunk_host_free(void *host_ptr);
int chipmunk_copy_d2h_async(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int chipmunk_copy_h2d_async(void *dev_dst, const void *host_src, size_t bytes, void *stream);
size_t chipmunk_host_bytes(void);

/* Gives back the library's device scratch (work plans, split partials, the multi-GB partial column sums of the fused
 * dense_colsum_attn pass -- bounded by option "big_scratch_gb", default 24).  Synchronises the device. */
int chipmunk_release_scratch(void);
/* How many requests for the multi-GB column-sum scratch were refused since the last release (each one sent a dense_colsum_attn /
 * dense_colsum_topk_mask call down the slower two-pass or chunked route): 0 in a healthy run. */
int chipmunk_big_scratch_fallbacks(void);

/* ---------------------------------------------------------------- projection output -> attention operands
 * qkv [n rows of row_stride elements, the first 3*heads*128 of each = (q|k|v, head, 128)] bf16  ->  q, k, v [heads, n, 128] bf16
 * with RMSNorm over the 128 head elements applied to q and k: bf16(bf16(x * rsqrt(mean(x^2) + eps)) * weight); weights bf16
 * [128] or NULL (= ones).  Optionally the rotary embedding of q and k for the first rope_rows tokens (the image tokens;
 * apply_rotary_emb, posemb_layers.py:133-172, (cos, sin) form): bf16(x * cos + rotate_half(x) * sin) in fp32, freqs_cos /
 * freqs_sin fp32 [rope_rows, 128] or both NULL.  One pass instead of the caller's rearrange + two RMSNorm modules + rotary
 * embedding + three transposes (examples/hunyuan/hyvideo/modules/models.py:188-199,376-392; norm_layers.py:43-58). */
int chipmunk_qkv_split_norm(const void *qkv, int64_t row_stride, const void *q_weight, const void *k_weight, void *q, void *k,
                            void *v, int64_t n, int heads, float eps, const float *freqs_cos, const float *freqs_sin,
                            int64_t rope_rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CHIPMUNK_HIP_H */
