// Launch parameters shared by the attention kernels (attn.hip: the general kernel and the host side; attn64.hip: the
// one-wave-per-SIMD kernels).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct AttnParams {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    const uint16_t *o_in;  // INPLACE only: the accumulation base (o itself for the in-place op, the cache for csp_attn_out)
    int64_t qs[3], ks[3], vs[3], os[3];
    const int32_t *indices, *counts;
    // ragged index rows (nullptr: row `item` = indices + item * idx_stride, idx_stride entries wide): row `item` = indices +
    // idx_off[item], idx_off[item + 1] - idx_off[item] entries wide; B*H*G + 1 offsets, each a multiple of 4
    const int64_t *idx_off;
    float *l_out;
    const float *p_in;
    uint16_t *cs;
    int cs_stride;
    int B, H, Nq, Nk, G, idx_stride;
    float o_scale;
    // key-split tail (see launch_attn): items >= split_full are handed to `nsplit` workgroups, each over a slice of the
    // item's key tiles; partial (o, m, l) go through `ws`, the last arriver (ticket) merges and runs the epilogue
    // (row-split tail, attn.hip MIX: split_full = the first item that runs as three 64-row workgroups, nsplit = 0; work-balanced launch,
    // BAL: tickets + 4096.. hold its id counter and flags, ws its published states)
    int split_full, nsplit;
    float *ws;
    int32_t *tickets;
    // optional work plan (attn_plan_kernel): block i processes item plan[2i] (< 0: nothing), slice (m & 0xff) of ((m >> 8) & 0xff)
    // slices over the item's key tiles, m = plan[2i+1]; m >> 16 = scratch slot (and ticket) of the item's slice 0; slices of one item are adjacent
    const int32_t *plan;
    int xcd_chunks;  // 1: every XCD walks its own contiguous (head, group) range; 0: all XCDs sweep one head together
    // attn64.hip / attn96.hip: per (batch, head) the largest Euclidean norm of a K row (knorm_max_kernel), or nullptr.  With it
    // a wave can prove |s_ij| <= |q_i| * kmax for all its queries; if that bound is small enough for the exponent range, the
    // exponentials are taken against the FIXED reference point |q_i| * kmax and the running-maximum work disappears.
    const float *kmax;
    // attn64.hip MODE 3 (dense + fused column sums): bf16 partial column sums, one row of Nk per (batch*head, 64-row wave block)
    uint16_t *cs_part;
    int cs_pstride;  // elements between two of those rows (chipmunk_colsum_part_stride)
    int probe;  // timing probes (tools/kbench.py --variants): 1 = no gathers after the prologue, 2 = gathers only
};


// attn64.hip: gathered attention over a work plan (p.plan, p.tickets, p.ws set by launch_attn); inplace = 1 for the
// accumulate forms (o_out = o_in + o_scale * result); `grid` = plan entries
int chipmunk_csp64_launch(const AttnParams &p, int inplace, int grid, hipStream_t stream);
// attn64.hip: dense attention with the column sums of dense_colsum_attn folded into the same pass (p.p_in, p.cs, p.cs_stride
// set; `part` = scratch of chipmunk_colsum_part_bytes(...) bytes), followed -- when p.cs is set -- by the combine of the per-wave
// partial sums into cs; with p.cs == nullptr the partial rows are the result (chipmunk_topk_mask_parts reads them)
int chipmunk_dense64_colsum_launch(const AttnParams &p, uint16_t *part, hipStream_t stream);
// indexed_io.hip: the top-k mask straight from the partial rows (row r of the mask = sum of the one or two rows colsum_part_rows names in
// its (batch*head) block of `nrb` = 2 * workgroups rows, rounded to bf16 -- exactly what the combine would have written)
int chipmunk_topk_mask_parts(const uint16_t *part, int part_stride, int nrb, int groups_per_bh, int Nq, const void *static_mask, int64_t static_stride,
                             int static_rows, const void *group_flags, void *mask, int rows, int n, int k, double random_amount,
                             hipStream_t stream);
size_t chipmunk_colsum_part_bytes(int B, int H, int Nq, int Nk);
// elements between two partial rows: Nk rounded up to whole 128-byte lines, so that a wave's 64 sums of a key tile are ONE
// aligned line (a row of 119 056 keys is 1 860.25 lines: at stride Nk every store straddled two)
int chipmunk_colsum_part_stride(int Nk);
// attn64.hip: the column-sum pass of dense_colsum_attn for long launches (one wave per 192-row group)
int chipmunk_colsum64_launch(const AttnParams &p, hipStream_t stream);
// attn96.hip: gathered attention, two waves x 96 rows per 192-row group, two workgroups per CU (plan as for csp64)
int chipmunk_csp96_launch(const AttnParams &p, int inplace, int grid, hipStream_t stream);
// attn64.hip: max_j |k_j| per (batch, head) into library scratch (nullptr if switched off / unavailable); see AttnParams::kmax
const float *chipmunk_knorm_max(const uint16_t *k, const int64_t ks[3], int B, int H, int Nk, hipStream_t stream);

// Partial column-sum rows of attn64.hip MODE 3: per (batch*head) 2 rows per 256-row workgroup -- row 2a = the waves of workgroup a that
// belong to its FIRST 192-row group, row 2a + 1 = the rest (see the kernel).  Group j starts at wave block 3j = workgroup a, wave w0;
// it is a's first set when w0 == 0, else its second, and with w0 >= 2 it continues as the first set of workgroup a + 1.
// Returns the number of rows (1 or 2) and their indices inside the (batch*head) block of 2 * nwg rows.
__host__ __device__ inline int colsum_part_rows(int j, int nwg, int &r0, int &r1) {
    const int a = (3 * j) >> 2, w0 = 3 * j - 4 * a;
    r0 = 2 * a + (w0 ? 1 : 0);
    r1 = 2 * (a + 1);
    return (w0 >= 2 && a + 1 < nwg) ? 2 : 1;
}

struct IndexRow {
    const int32_t *ptr;
    int width;   // entries that may be read (positions past it are clamped / read as 0)
};
__device__ __forceinline__ IndexRow index_row(const AttnParams &p, int64_t item) {
    if (p.idx_off) {
        const int64_t o = p.idx_off[item];
        return {p.indices + o, (int)(p.idx_off[item + 1] - o)};
    }
    return {p.indices + item * p.idx_stride, p.idx_stride};
}
