// Indexed IO for gfx950: mask -> indices, top-k indices, indexed copy, bit packing.
// Replaces reference csrc/indexed_io/{mask_to_indices,topk_indices,copy_indices}.cu and ops/bitpack.py.
// HBM-bound integer/byte kernels: wide coalesced loads, wave64 ballots / popcounts instead of CUB scans,
// LDS for the per-row bit matrices.  Results are bit-exact against the reference's ordering rules.
#include "common.h"
#include "attn_params.h"

namespace {

// =====================================================================================  mask_to_indices
// Reference order (mask_to_indices.cu:49-86, one 32-lane warp per row): for residue class t = 0..31 all True columns
// c == t (mod 32) ascending, then padding with the first False columns ascending.
//
// Here one 256-thread workgroup per row:
//   A. each lane turns 32 mask bytes (one "i-row": columns 32i..32i+31) into a 32-bit word (bits[i] in LDS);
//   B. 32 wave64 ballots per 64 i-rows transpose the bit matrix: T[t][blk] = 64-bit vector of class t;
//   C. popcounts + a per-class scan over blocks + a scan over classes give every (class, block) cell its output offset;
//   D. each lane walks the set bits of its cell and writes the column numbers;
//   E. one lane appends the padding columns from the untransposed words.
// inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1/2/4/8 inside the 16-lane rows, row_bcast:15 and :31
// across them) -- the __shfl_up form is six ds_bpermute round trips through the LDS crossbar, each with its index
// arithmetic and select, and the sorted compaction below runs two such chains per 2 048 columns.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
struct M2IParams {
    const uint8_t *mask;  // bool bytes [rows, n]  (PACKED == false)  or packed bits [rows*n/8] (PACKED == true)
    int32_t *indices;
    int32_t *counts;
    int n, pad_n, multiple_of;
};

// SORTED = true emits the kept columns in ASCENDING order instead (same set, same counts, same padding columns): not
// the reference's order, but the attention result does not depend on it and ascending keys turn the K/V gather into
// near-sequential DRAM pages (measured 1.6x on the C3 sparse step).  Used by SparseDiffAttn's fused path only.
template <bool PACKED, bool SORTED = false>
__global__ __launch_bounds__(256) void mask_to_indices_kernel(const M2IParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = p.n;
    const int NI = (n + 31) >> 5;   // 32-column words per row
    const int NB = (NI + 63) >> 6;  // 64-word blocks per row
    uint32_t *bits = (uint32_t *)smem;                               // [NB*64]
    unsigned long long *T = (unsigned long long *)(bits + NB * 64);  // [32][NB]
    uint32_t *pre = (uint32_t *)(T + 32 * NB);                       // [32][NB] exclusive offsets
    uint32_t *cls = pre + 32 * NB;                                   // [33] class totals / offsets

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t row = blockIdx.x;
    int32_t *out = p.indices + row * p.pad_n;

    // ---- A: bit words
    if constexpr (PACKED) {
        const uint8_t *src = p.mask + ((row * n) >> 3);  // n % 8 == 0 guaranteed by the host
        const bool aligned4 = ((((uintptr_t)src) & 3) == 0);
        for (int i = tid; i < NB * 64; i += 256) {
            uint32_t word = 0;
            if (i < NI) {
                const int nbytes = min(4, (n - i * 32 + 7) >> 3);
                if (aligned4 && nbytes == 4) {
                    word = *(const uint32_t *)(src + i * 4);
                } else {
                    for (int j = 0; j < nbytes; ++j) word |= (uint32_t)src[i * 4 + j] << (8 * j);
                }
                const int rem = n - i * 32;
                if (rem < 32) word &= (1u << rem) - 1u;
            }
            bits[i] = word;
        }
    } else {
        const uint8_t *src = p.mask + row * n;
        const bool aligned16 = ((((uintptr_t)src) & 15) == 0);
        for (int i = tid; i < NB * 64; i += 256) {
            uint32_t word = 0;
            if (i < NI) {
                if (aligned16 && i * 32 + 32 <= n) {
                    const u32x4 lo = *(const u32x4 *)(src + i * 32);
                    const u32x4 hi = *(const u32x4 *)(src + i * 32 + 16);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        // bool bytes are 0/1 (any non-zero counts as true): gather bit 0 of the 4 bytes of each dword
                        uint32_t x = lo[d], y = hi[d];
                        x = (x | (x >> 1) | (x >> 2) | (x >> 3) | (x >> 4) | (x >> 5) | (x >> 6) | (x >> 7)) & 0x01010101u;
                        y = (y | (y >> 1) | (y >> 2) | (y >> 3) | (y >> 4) | (y >> 5) | (y >> 6) | (y >> 7)) & 0x01010101u;
                        const uint32_t nx = (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
                        const uint32_t ny = (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xfu;
                        word |= nx << (4 * d);
                        word |= ny << (16 + 4 * d);
                    }
                } else {
                    const int rem = min(32, n - i * 32);
                    for (int j = 0; j < rem; ++j) word |= (uint32_t)(src[i * 32 + j] != 0) << j;
                }
            }
            bits[i] = word;
        }
    }
    __syncthreads();

    if constexpr (SORTED) {
        // ordered compaction, wave-cooperative: a wave takes 64 consecutive words (2048 columns) at a time, lane = word.
        // Pass 1: popcount per word -> per-block totals in LDS; one wave scans the (<= 64 per step) block totals.
        // Pass 2: wave prefix of the lane popcounts + block base = every word's output offset; lanes walk their set bits.
        // Consecutive lanes write consecutive output runs, so a wave's store instruction covers a few cache lines (the
        // per-thread-chunk form wrote 256 separate streams per row: 0.95 ms per C3 layer).
        uint32_t *blktot = (uint32_t *)T;   // [NB + 1] (the transpose matrix is not used by this variant)
        for (int blk = w; blk < NB; blk += 4) {
            const uint32_t c = wave_incl_scan(__popc(bits[blk * 64 + lane]));
            if (lane == 63) blktot[blk] = c;
        }
        __syncthreads();
        if (w == 0) {   // exclusive scan of the block totals, 64 blocks per step
            uint32_t run = 0;
            for (int b0 = 0; b0 < NB; b0 += 64) {
                const uint32_t v = b0 + lane < NB ? blktot[b0 + lane] : 0u;
                const uint32_t incl = wave_incl_scan(v);
                if (b0 + lane < NB) blktot[b0 + lane] = run + incl - v;
                run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (lane == 0) blktot[NB] = run;
        }
        __syncthreads();
        const int total = (int)blktot[NB];
        for (int blk = w; blk < NB; blk += 4) {
            uint32_t v = bits[blk * 64 + lane];
            const uint32_t c = __popc(v);
            const uint32_t incl = wave_incl_scan(c);
            int pos = (int)(blktot[blk] + incl - c);
            const int col0 = (blk * 64 + lane) * 32;
            while (v) {
                const int bit = __builtin_ctz(v);
                v &= v - 1;
                out[pos++] = col0 + bit;
            }
        }
        if (tid == 0) {
            const int padded = ((total + p.multiple_of - 1) / p.multiple_of) * p.multiple_of;
            int pp = total;
            for (int i = 0; i < NI && pp < padded; ++i) {
                uint32_t z = ~bits[i];
                const int rem = n - i * 32;
                if (rem < 32) z &= (1u << rem) - 1u;
                while (z && pp < padded) {
                    const int bit = __builtin_ctz(z);
                    z &= z - 1;
                    out[pp++] = i * 32 + bit;
                }
            }
            p.counts[row] = padded;
        }
        return;
    }
    // ---- B: transpose by ballots (each wave takes blocks w, w+4, ...)
    for (int blk = w; blk < NB; blk += 4) {
        const uint32_t word = bits[blk * 64 + lane];
        unsigned long long mine = 0;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const unsigned long long bal = __ballot((word >> t) & 1u);
            if (lane == t) mine = bal;
        }
        if (lane < 32) T[lane * NB + blk] = mine;
    }
    __syncthreads();

    // ---- C: offsets.  One lane per class scans its blocks; then 32 classes are scanned by lane 0.
    if (tid < 32) {
        uint32_t run = 0;
        for (int blk = 0; blk < NB; ++blk) {
            pre[tid * NB + blk] = run;
            run += __popcll(T[tid * NB + blk]);
        }
        cls[tid + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 32; ++t) {
            const uint32_t c = cls[t + 1];
            cls[t] = run;  // exclusive class offset (cls[t+1] is consumed before it is overwritten next iteration)
            run += c;
            if (t == 31) cls[32] = run;
        }
    }
    __syncthreads();
    const int total = (int)cls[32];

    // ---- D: emit.  Consecutive lanes take consecutive blocks of one class => neighbouring output runs.
    for (int cell = tid; cell < 32 * NB; cell += 256) {
        const int t = cell / NB, blk = cell - t * NB;
        unsigned long long v = T[cell];
        int pos = (int)(cls[t] + pre[cell]);
        while (v) {
            const int bit = __builtin_ctzll(v);
            v &= v - 1;
            out[pos++] = ((blk * 64 + bit) << 5) + t;
        }
    }

    // ---- E: padding with the first False columns (lane 0 only, <= multiple_of-1 columns)
    if (tid == 0) {
        const int padded = ((total + p.multiple_of - 1) / p.multiple_of) * p.multiple_of;
        int pos = total;
        for (int i = 0; i < NI && pos < padded; ++i) {
            uint32_t z = ~bits[i];
            const int rem = n - i * 32;
            if (rem < 32) z &= (1u << rem) - 1u;
            while (z && pos < padded) {
                const int bit = __builtin_ctz(z);
                z &= z - 1;
                out[pos++] = i * 32 + bit;
            }
        }
        p.counts[row] = padded;
    }
}

// =====================================================================================  topk_indices
struct TopkParams {
    const void *act;
    void *cache;  // DELTA variant: the block-mean cache the activations are compared with and copied into
    int32_t *indices;
    int32_t *counts;
    int rows, cols, multiple_of;
    float quantile, random_amount;
    uint32_t salt;  // per-launch value (host counter mixed with the seed, see chipmunk_next_random_salt)
};

template <typename T>
__device__ __forceinline__ float load_as_float(const T *p, int i);
template <>
__device__ __forceinline__ float load_as_float<uint16_t>(const uint16_t *p, int i) { return bf16_bits_to_f32(p[i]); }
template <>
__device__ __forceinline__ float load_as_float<_Float16>(const _Float16 *p, int i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float load_as_float<float>(const float *p, int i) { return p[i]; }

template <typename T>
__device__ __forceinline__ float round_to(float x);
template <>
__device__ __forceinline__ float round_to<uint16_t>(float x) { return round_bf16(x); }
template <>
__device__ __forceinline__ float round_to<_Float16>(float x) { return (float)(_Float16)x; }
template <>
__device__ __forceinline__ float round_to<float>(float x) { return x; }

// counter-based uniform in [0,1): replaces cuRAND Philox of topk_indices.cu:46-49,108 (RNG streams cannot match).
// `salt` changes with every launch (host counter + seed) and with the row's data (its first element, as the reference
// seeds cuRAND from the first activation word, topk_indices.cu:47-49): the random keys exist to refresh stale cache
// columns over time, so the chosen set must differ between layers, steps and generations.
__device__ __forceinline__ float hash_uniform(uint32_t row, uint32_t col, uint32_t salt) {
    uint32_t x = (row * 0x9E3779B9u ^ (col + 0x7F4A7C15u) * 0x85EBCA6Bu) + salt * 0xC2B2AE35u;
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// Block-wide ordered compaction step for a 1024-thread workgroup (16 waves): returns this thread's output slot (or
// -1) and advances `base`.  One ballot + popcount per wave, the 16 wave totals are scanned through LDS.
__device__ __forceinline__ int compact_slot(bool keep, int lane, int w, uint32_t *wave_tot, int &base) {
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[w] = __popcll(bal);
    __syncthreads();
    int off = base, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = wave_tot[i];
        off += i < w ? c : 0;
        tot += c;
    }
    __syncthreads();
    base += tot;
    return keep ? off + before : -1;
}

// One 1024-thread workgroup per row (the reference's launch shape, topk_indices.cu:200-213).  The quantile of the
// 1024-column sample is found by rank counting (each thread ranks its own sample against all 1024 through LDS
// broadcasts) instead of the reference's CUB block merge sort: same element, ~1 us.
// DELTA = true fuses the sparse-MLP sequence |bmfc1 - cache| -> topk_indices -> copy_indices (reference
// modules/mlp.py:70-85 with bm == mbm): the ranked value is |T(b - cache)| computed on the fly and every selected
// column (padding included, like copy_indices) is copied into the cache by the thread that owns it.
template <typename T, bool DELTA>
__global__ __launch_bounds__(1024) void topk_indices_kernel(const TopkParams p) {
    __shared__ uint32_t wave_tot[16];
    __shared__ float thr_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x;
    const T *x = (const T *)p.act + (int64_t)row * p.cols;
    T *xc = DELTA ? (T *)p.cache + (int64_t)row * p.cols : nullptr;
    int32_t *out = p.indices + (int64_t)row * p.cols;
    const int cols = p.cols;
    auto value = [&](int c) {
        const float b = load_as_float<T>(x, c);
        if constexpr (DELTA) return fabsf(round_to<T>(b - load_as_float<T>(xc, c)));
        else return b;
    };

    const uint32_t salt = p.salt ^ (__float_as_uint(value(0)) * 0x27D4EB2Fu);  // the fused (delta) form sees what the unfused op sees
    if (p.quantile == 0.f) {  // keep everything (topk_indices.cu:51-59)
        if constexpr (DELTA)
            for (int c = tid; c < cols; c += 1024) xc[c] = x[c];
        for (int c = tid; c < cols; c += 1024) out[c] = c;
        if (tid == 0) p.counts[row] = cols;
        return;
    }
    if (p.quantile == 1.f) {  // keep nothing (topk_indices.cu:60-69)
        for (int c = tid; c < cols; c += 1024) out[c] = -1;
        if (tid == 0) p.counts[row] = 0;
        return;
    }
    // ---- threshold = element int(1024*q) of the ascending-sorted first 1024 values (topk_indices.cu:91-101).
    //      Order statistic by bitwise bisection inside ONE wave (16 sample keys per lane, no barriers, no sort):
    //      result = largest v with #{keys < v} <= k, built from the most significant bit down.
    // every thread first requests the 16 columns it will test in the first compaction pass (t, t + 1024, ...): column
    // t of that batch is also its share of the 1024-column quantile sample, handed to wave 0 through LDS -- one memory
    // round trip for sample and first pass instead of two (the rows are cold in the real loop: 31 -> 25 us)
    __shared__ float sample[1024];
    float v_first[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = j * 1024 + tid;
        v_first[j] = c < cols ? value(c) : 0.f;
    }
    sample[tid] = v_first[0];
    __syncthreads();
    if (w == 0) {
        uint32_t key[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t u = __float_as_uint(sample[lane + 64 * j]);
            key[j] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> unsigned map
        }
        const int k = (int)(1024 * p.quantile);
        uint32_t res = 0;
        // |delta| rounded to a 16-bit type: every key is a non-negative float whose LOWZ low bits are zero (bf16: 16, fp16: 13), so
        // the LOWZ low rounds all ask the same question (#{key <= res} <= k ?) and set all of their bits or none -- one round
        // decides them, same result bit for bit, 17 rounds instead of 32
        constexpr int LOWZ = !DELTA ? 0 : std::is_same<T, uint16_t>::value ? 16 : std::is_same<T, _Float16>::value ? 13 : 0;
        constexpr int LAST = LOWZ ? LOWZ - 1 : 0;
        for (int bit = 31; bit >= LAST; --bit) {
            const uint32_t cand = res | (LOWZ && bit == LAST ? (1u << LOWZ) - 1u : (1u << bit));
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) cnt += key[j] < cand ? 1 : 0;
            // wave total by DPP row rotations + two lane swaps (exact in fp32: <= 1024).  Six __shfl_xor steps are six ds_bpermute round
            // trips per round, 32 rounds in a row on one wave while the other 15 wait: ~10 us of this kernel's 19 (round 4)
            const int tot = (int)sum_across_rows(row16_sum((float)cnt));
            if (tot <= k) res = cand;
        }
        const uint32_t u = (res & 0x80000000u) ? (res & 0x7fffffffu) : ~res;
        if (lane == 0) thr_s = __uint_as_float(u);
    }
    __syncthreads();
    const float thr = thr_s;

    // ---- ordered compaction of kept columns, 16 x 1024 columns per pass: every thread tests up to 16 columns
    //      (t, t+1024, ...: all loads in flight at once), one ballot per (chunk, wave), ONE scan of the 256 wave totals.
    __shared__ uint32_t tot[16 * 16 + 1];
    int base = 0;
    int my_last_invalid = -1;
    for (int pass0 = 0; pass0 < cols; pass0 += 16 * 1024) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = pass0 + j * 1024 + tid;
            v[j] = pass0 == 0 ? v_first[j] : (c < cols ? value(c) : 0.f);
        }
        uint32_t keepbits = 0;
        int before[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = pass0 + j * 1024 + tid;
            bool keep = false;
            if (c < cols) {
                keep = v[j] >= thr;
                if (!keep && p.random_amount > 0.f) keep = hash_uniform(row, c, salt) < p.random_amount;
                if (!keep) my_last_invalid = c;  // ascending c per thread => ends as the last rejected column
            }
            const unsigned long long bal = __ballot(keep);
            before[j] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) tot[j * 16 + w] = __popcll(bal);
            keepbits |= keep ? (1u << j) : 0u;
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the 256 (chunk, wave) totals by one wave, 4 entries per lane
            uint32_t a0 = tot[tid * 4], a1 = tot[tid * 4 + 1], a2 = tot[tid * 4 + 2], a3 = tot[tid * 4 + 3];
            uint32_t sum = a0 + a1 + a2 + a3, incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const uint32_t excl = incl - sum;
            tot[tid * 4] = excl, tot[tid * 4 + 1] = excl + a0, tot[tid * 4 + 2] = excl + a0 + a1;
            tot[tid * 4 + 3] = excl + a0 + a1 + a2;
            if (tid == 63) tot[256] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (keepbits & (1u << j)) {
                const int c = pass0 + j * 1024 + tid;
                out[base + tot[j * 16 + w] + before[j]] = c;
                if constexpr (DELTA) xc[c] = x[c];
            }
        base += tot[256];
        __syncthreads();
    }
    const int kept = base;
    const int mod = kept % p.multiple_of;
    const int pad = mod == 0 ? 0 : p.multiple_of - mod;
    if (tid == 0) p.counts[row] = kept + pad;
    // ---- padding candidates in ascending residue order (a deterministic choice among the reference's outcomes)
    if (pad > 0) {
        int pbase = 0;
        const int slot = compact_slot(my_last_invalid != -1, lane, w, wave_tot, pbase);
        if (slot >= 0 && slot < pad) {
            out[kept + slot] = my_last_invalid;
            if constexpr (DELTA) xc[my_last_invalid] = x[my_last_invalid];
        }
    }
}

// =====================================================================================  copy_indices
template <typename T>
__global__ __launch_bounds__(256) void copy_indices_kernel(const T *src, T *dst, const int32_t *inds,
                                                           const int32_t *counts, int M, int R, int F) {
    const int64_t grow = blockIdx.x;                 // row over B*M*R
    const int64_t b = grow / ((int64_t)M * R);
    const int base_m = (int)((grow % ((int64_t)M * R)) / R);
    const int cnt = counts[b * M + base_m];
    const int32_t *ii = inds + (b * M + base_m) * (int64_t)F;
    for (int c = threadIdx.x; c < cnt; c += 256) {
        const int col = ii[c];
        dst[grow * F + col] = src[grow * F + col];
    }
}

// =====================================================================================  bitpack / bitunpack
__global__ __launch_bounds__(256) void bitpack_kernel(const uint8_t *mask, uint8_t *packed, int64_t n) {
    const int64_t nb = (n + 7) >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
        uint8_t byte = 0;
        if (i * 8 + 8 <= n && ((((uintptr_t)mask) & 7) == 0)) {
            unsigned long long v = *(const unsigned long long *)(mask + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) byte |= (uint8_t)((((v >> (8 * j)) & 0xffull) != 0) << j);
        } else {
            for (int j = 0; j < 8 && i * 8 + j < n; ++j) byte |= (uint8_t)((mask[i * 8 + j] != 0) << j);
        }
        packed[i] = byte;
    }
}
__global__ __launch_bounds__(256) void bitunpack_kernel(const uint8_t *packed, uint8_t *mask, int64_t n) {
    const int64_t nb = (n + 7) >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
        const uint8_t byte = packed[i];
        if (i * 8 + 8 <= n && ((((uintptr_t)mask) & 7) == 0)) {
            unsigned long long v = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v |= (unsigned long long)((byte >> j) & 1u) << (8 * j);
            *(unsigned long long *)(mask + i * 8) = v;
        } else {
            for (int j = 0; j < 8 && i * 8 + j < n; ++j) mask[i * 8 + j] = (byte >> j) & 1u;
        }
    }
}

// =====================================================================================  2-byte transpose
// [B, R, C] -> [B, C, R] for 16-bit elements (the column-major activation cache of the sparse MLP is act^T,
// reference modules/mlp.py:56).  64x64 tiles through LDS: 128-byte row segments on both the read and the write side.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t *src, uint16_t *dst, int R, int C) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];
    const int64_t boff = (int64_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = i * 256 + tid, r = item >> 3, ch = item & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c0 + ch * 8 + 8 <= C) v = *(const u32x4 *)(src + boff + (int64_t)(r0 + r) * C + c0 + ch * 8);
        else if (r0 + r < R)
            for (int e = 0; e < 8; ++e)
                if (c0 + ch * 8 + e < C) {
                    const uint32_t x = src[boff + (int64_t)(r0 + r) * C + c0 + ch * 8 + e];
                    v[e >> 1] |= x << (16 * (e & 1));
                }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[ch * 8 + 2 * e][r] = (uint16_t)(v[e] & 0xffffu);
            tile[ch * 8 + 2 * e + 1][r] = (uint16_t)(v[e] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = i * 256 + tid, c = item >> 3, ch = item & 7;
        if (c0 + c >= C) continue;
        uint16_t *d = dst + boff + (int64_t)(c0 + c) * R + r0 + ch * 8;
        if (r0 + ch * 8 + 8 <= R && ((R & 7) == 0)) *(u32x4 *)d = *(const u32x4 *)&tile[c][ch * 8];
        else
            for (int e = 0; e < 8; ++e)
                if (r0 + ch * 8 + e < R) d[e] = tile[c][ch * 8 + e];
    }
}

// ragged copy of index rows: row r of `flat` (at offsets[r]) = the first counts[r] entries of row r of `indices`, the rest of
// the row (up to offsets[r + 1]) zero.  One workgroup per row, 16 bytes per lane when both sides allow.
__global__ __launch_bounds__(256) void compact_indices_kernel(const int32_t *indices, int64_t idx_stride, const int32_t *counts,
                                                             const int64_t *offsets, int32_t *flat) {
    const int64_t r = blockIdx.x;
    const int32_t *src = indices + r * idx_stride;
    const int64_t o = offsets[r];
    int32_t *dst = flat + o;
    const int width = (int)(offsets[r + 1] - o);
    int c = counts[r];
    c = c < 0 ? 0 : (c < width ? c : width);
    c = c < idx_stride ? c : (int)idx_stride;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int c4 = c >> 2;
        for (int i = threadIdx.x; i < c4; i += 256) ((u32x4 *)dst)[i] = ((const u32x4 *)src)[i];
        for (int i = (c4 << 2) + threadIdx.x; i < c; i += 256) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < c; i += 256) dst[i] = src[i];
    }
    for (int i = c + threadIdx.x; i < width; i += 256) dst[i] = 0;
}

size_t m2i_lds_bytes(int n, bool sorted) {
    const int NI = (n + 31) >> 5, NB = (NI + 63) >> 6;
    if (sorted) return (size_t)NB * 64 * 4 + (size_t)(NB + 1) * 4 + 16;   // bit words + block totals: 8 workgroups per CU
    return (size_t)NB * 64 * 4 + (size_t)32 * NB * 8 + (size_t)32 * NB * 4 + 33 * 4 + 16;
}

template <bool PACKED, bool SORTED = false>
int launch_m2i(const void *mask, int32_t *indices, int32_t *counts, int64_t rows, int n, int pad_n, int multiple_of,
               void *stream) {
    CM_CHECK(mask && indices && counts, "mask_to_indices: null pointer");
    CM_CHECK(rows >= 0 && n > 0 && pad_n >= n && multiple_of > 0, "mask_to_indices: bad sizes rows=%lld n=%d pad_n=%d multiple_of=%d",
             (long long)rows, n, pad_n, multiple_of);
    CM_CHECK(rows < (1ll << 31), "mask_to_indices: too many rows");
    if (PACKED) CM_CHECK(n % 8 == 0, "packed_mask_to_indices: n must be a multiple of 8 (got %d)", n);
    const size_t lds = m2i_lds_bytes(n, SORTED);
    CM_CHECK(lds <= 160 * 1024, "mask_to_indices: row length %d needs %zu B of LDS (> 160 KiB)", n, lds);
    if (rows == 0) return CHIPMUNK_OK;
    auto kern = mask_to_indices_kernel<PACKED, SORTED>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    M2IParams p = {(const uint8_t *)mask, indices, counts, n, pad_n, multiple_of};
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}


// ------------------------------------------------------------------------------------------------ top-k -> mask
// SURVEY 8f rank 1, second half: the reference's `random_and_topk` (src/chipmunk/modules/attn.py:76-82)
//     mask = randint(0,100) == 0;  mask.scatter_(-1, cs.topk(k).indices, True);  mask = (mask * groups) | static
// as ONE pass over the column sums instead of randint + topk + scatter + two element-wise kernels on a [H, G, N] bool
// tensor (N = 119 056: 24.7 ms -> see DESIGN.md).
// One 1024-thread workgroup per (head, query group) row.  The row's bf16 column sums are loaded ONCE into registers as
// order-preserving 16-bit keys (4 consecutive columns per thread per step, KPT/2 packed VGPRs), the k-th largest key is
// found by 16 rounds of bitwise bisection on block-wide counts (no sort, no histogram: the keys of one row sit in two or
// three exponent bins, a histogram's atomics would serialise), ties at the threshold are handed out in thread order,
// and the final mask bytes are written straight from the registers, 4 per thread per store.
struct TopkMaskParams {
    const uint16_t *cs;      // [rows, cs_stride] bf16
    const uint8_t *stat;     // optional static mask, row r -> stat + (r % stat_rows) * stat_stride
    const uint8_t *groups;   // optional per-row flag (the random / top-k part only applies where it is set)
    uint8_t *mask;           // [rows, n] bool bytes, fully overwritten
    int64_t cs_stride, stat_stride;
    int rows, n, k, stat_rows;
    float random_amount;
    uint32_t salt;           // per-launch value, see chipmunk_next_random_salt
    // PARTS form: row r of "cs" = bf16(sum of rows 3g .. 3g+2 of the partial rows of its (batch*head) block), g = r % groups
    const uint16_t *parts;
    int64_t pstride;         // elements between two partial rows
    int nrb, ngroups, nq;    // partial rows per (batch*head); 192-row groups per (batch*head); query rows
};

__device__ __forceinline__ uint32_t bf16_key(uint32_t u) {  // monotone bf16 bits -> u16 (larger value = larger key)
    return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
}

template <int KPT, bool ALIGNED, bool PARTS = false>  // keys per thread (multiple of 4); the row must have n <= 1024 * KPT columns.
// PARTS (ALIGNED only): the row's column sums are not in memory; they are the sum of up to three bf16 partial rows (see above)
// ALIGNED: n % 4 == 0 and every row of cs / static mask / mask starts on an 8 / 4 / 4-byte boundary, so each thread
// step is one 8-byte load, one 4-byte load and one 4-byte store; the generic form goes element by element.
__global__ __launch_bounds__(1024) void topk_mask_kernel(const TopkMaskParams p) {
    constexpr int NV = KPT / 4;  // 4-column steps per thread
    __shared__ int wave_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = blockIdx.x;
    const int n = p.n;
    const uint16_t *x = nullptr;
    int prow = 1;
    int64_t row1 = 0;      // PARTS: elements from the group's first partial row to its second
    if constexpr (PARTS) {
        const int bh = row / p.ngroups, g = row - bh * p.ngroups;
        int r0, r1;
        prow = colsum_part_rows(g, p.nrb / 2, r0, r1);
        x = p.parts + ((int64_t)bh * p.nrb + r0) * p.pstride;
        row1 = (int64_t)(r1 - r0) * p.pstride;
    } else {
        x = p.cs + (int64_t)row * p.cs_stride;
    }
    const uint32_t lane_off8 = 8u * (uint32_t)tid;    // byte offset of this thread's 4 bf16 inside a 4096-column step
    const uint32_t lane_off4 = 4u * (uint32_t)tid;    // ... of its 4 mask bytes
    // columns 4*tid + 4096*j .. +3; columns past n read as bf16 bits 0xffff = key 0, below every real key
    uint32_t key[NV][2];
    constexpr int LB = NV % 10 == 0 ? 10 : (NV % 8 == 0 ? 8 : (NV % 6 == 0 ? 6 : 4));  // loads in flight per batch
    static_assert(NV % LB == 0, "NV must split into whole load batches");
#pragma unroll
    for (int j0 = 0; j0 < NV; j0 += LB) {
        // batch of LB loads, then their conversion: bounds the live raw values (all NV at once would spill)
        uint32_t raw[LB][2];
#pragma unroll
        for (int jj = 0; jj < LB; ++jj) {
            const int c = 4 * tid + 4096 * (j0 + jj);
            uint32_t v0 = 0xffffffffu, v1 = 0xffffffffu;
            if constexpr (ALIGNED) {
                if (c < n) {
                    // wave-uniform base + one shared 32-bit lane offset: the saddr form, no 64-bit address per step
                    const u32x2 v = *(const u32x2 *)((const unsigned char *)(x + 4096 * (j0 + jj)) + lane_off8);
                    v0 = v[0], v1 = v[1];
                    if constexpr (PARTS) {   // the combine, in its order: row 0 + row 1 in fp32, one rounding to bf16
                        if (prow == 2) {
                            float a0 = __uint_as_float(v0 << 16), a1 = __uint_as_float(v0 & 0xffff0000u);
                            float a2 = __uint_as_float(v1 << 16), a3 = __uint_as_float(v1 & 0xffff0000u);
                            const u32x2 y = *(const u32x2 *)((const unsigned char *)(x + row1 + 4096 * (j0 + jj)) + lane_off8);
                            a0 += __uint_as_float(y[0] << 16), a1 += __uint_as_float(y[0] & 0xffff0000u);
                            a2 += __uint_as_float(y[1] << 16), a3 += __uint_as_float(y[1] & 0xffff0000u);
                            v0 = pack_bf16x2(a0, a1), v1 = pack_bf16x2(a2, a3);
                        }
                    }
                }
            } else if (c < n) {
                v0 = v1 = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t u = c + e < n ? x[c + e] : 0xffffu;
                    if (e < 2) v0 |= u << (16 * e);
                    else v1 |= u << (16 * (e - 2));
                }
            }
            raw[jj][0] = v0, raw[jj][1] = v1;
        }
#pragma unroll
        for (int jj = 0; jj < LB; ++jj) {
            key[j0 + jj][0] = bf16_key(raw[jj][0] & 0xffffu) | (bf16_key(raw[jj][0] >> 16) << 16);
            key[j0 + jj][1] = bf16_key(raw[jj][1] & 0xffffu) | (bf16_key(raw[jj][1] >> 16) << 16);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    auto block_count = [&](int mine) {  // sum over the 1024 threads, broadcast
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
        if (lane == 0) wave_cnt[w] = mine;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += wave_cnt[i];
        __syncthreads();
        return t;
    };
    const bool active = p.groups == nullptr || p.groups[row] != 0;
    const int k = p.k < n ? p.k : n;
    uint32_t thr = 0x10000u;  // keep nothing
    int ties_to_take = 0;
    if (active && k > 0) {
        // ---- largest T with #{key >= T} >= k, most significant bit first
        uint32_t res = 0;
        for (int bit = 15; bit >= 0; --bit) {
            const uint32_t cand = res | (1u << bit);
            // both 16-bit keys of a word at once: sat(key - (cand - 1)) is non-zero iff key >= cand; min(., 1) makes it a count;
            // packed u16 counters (a thread holds at most 4 * NV <= 480 keys).  Three VALU operations per two keys (the 32-bit
            // form took eight): this loop is 16 rounds x NV * 2 words x 16 waves on one CU's four SIMDs, the kernel's longest phase
            const uint32_t cm1 = (cand - 1u) | ((cand - 1u) << 16);
            const uint32_t one2 = 0x00010001u;
            uint32_t cnt2 = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t d;
                    // (asm also keeps the optimiser from hoisting per-word work out of the 16 rounds)
                    asm volatile("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(d) : "v"(key[j][h]), "v"(cm1), "v"(one2));
                    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(cnt2) : "v"(d));
                }
            const int mine = (int)(cnt2 & 0xffffu) + (int)(cnt2 >> 16);
            if (block_count(mine) >= k) res = cand;
        }
        thr = res;
        int gt = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t kw = key[j][h];
                asm volatile("" : "+v"(kw));
                gt += ((kw & 0xffffu) > thr ? 1 : 0) + ((kw >> 16) > thr ? 1 : 0);
            }
        ties_to_take = k - block_count(gt);  // >= 1 by construction of thr
    }
    // ---- ties at the threshold go to the lowest thread ids (torch.topk leaves the choice among equals unspecified)
    int my_ties = 0;
    if (ties_to_take > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t kw = key[j][h];
                asm volatile("" : "+v"(kw));
                my_ties += ((kw & 0xffffu) == thr ? 1 : 0) + ((kw >> 16) == thr ? 1 : 0);
            }
    }
    int tie_budget = 0;
    {
        int incl = my_ties;  // inclusive scan over the block: wave scan, then the 16 wave totals
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_cnt[w] = incl;
        __syncthreads();
        int before = incl - my_ties;
#pragma unroll
        for (int i = 0; i < 16; ++i) before += i < w ? wave_cnt[i] : 0;
        __syncthreads();
        tie_budget = ties_to_take - before;  // how many of MY ties are taken (clamped below)
    }
    // ---- write the mask: (top-k | random) & group | static.  ROLLED loop that re-reads the row's column sums (the
    //      workgroup's own row: L2 / Infinity-Cache hits) instead of taking them from the key registers: an unrolled
    //      register-fed output loop needs ~60 more live registers than the selection and spilled at 30 steps per thread.
    uint8_t *out = p.mask + (int64_t)row * n;
    const uint8_t *st = p.stat ? p.stat + (int64_t)(row % p.stat_rows) * p.stat_stride : nullptr;
    const bool rnd = active && p.random_amount > 0.f;
    // (PARTS: the combined value at column c0..c0+3 again, same order of additions as in the selection pass)
    auto combined4 = [&](int c0, uint32_t &v0, uint32_t &v1) {
        const u32x2 v = *(const u32x2 *)(x + c0);
        v0 = v[0], v1 = v[1];
        if constexpr (PARTS) {
            float a0 = __uint_as_float(v0 << 16), a1 = __uint_as_float(v0 & 0xffff0000u);
            float a2 = __uint_as_float(v1 << 16), a3 = __uint_as_float(v1 & 0xffff0000u);
            if (prow == 2) {
                const u32x2 y = *(const u32x2 *)(x + row1 + c0);
                a0 += __uint_as_float(y[0] << 16), a1 += __uint_as_float(y[0] & 0xffff0000u);
                a2 += __uint_as_float(y[1] << 16), a3 += __uint_as_float(y[1] & 0xffff0000u);
            }
            v0 = pack_bf16x2(a0, a1), v1 = pack_bf16x2(a2, a3);
        }
    };
    uint32_t first = x[0];
    if constexpr (PARTS) {
        uint32_t f0, f1;
        combined4(0, f0, f1);
        first = f0 & 0xffffu;
    }
    const uint32_t salt = p.salt ^ (first * 0x27D4EB2Fu);
    const int nsteps = (n + 4095) / 4096;
    // keyed: v0 / v1 hold the order-preserving keys already (the selection's registers), else raw bf16 bits
    auto emit = [&](auto keyed, int c, uint32_t v0, uint32_t v1, uint32_t sb) {
        uint32_t bytes = 0;
        if constexpr (decltype(keyed)::value) asm volatile("" : "+v"(v0), "+v"(v1));   // opaque: no unpacking of all NV * 2 words ahead of time
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t half = ((e < 2 ? v0 : v1) >> (16 * (e & 1))) & 0xffffu;
            const uint32_t kv = decltype(keyed)::value ? half : bf16_key(half);
            bool keep = kv > thr;                      // thr = 0x10000 when the row is inactive or k = 0
            const bool tie = kv == thr && tie_budget > 0 && c + e < n;
            tie_budget -= tie ? 1 : 0;
            keep = keep || tie;
            if (rnd && !keep) keep = hash_uniform(row, c + e, salt) < p.random_amount;
            keep = keep || ((sb >> (8 * e)) & 0xffu) != 0;
            bytes |= (keep ? 1u : 0u) << (8 * e);
        }
        if constexpr (ALIGNED) {
            *(uint32_t *)(out + c) = bytes;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < n) out[c + e] = (uint8_t)(bytes >> (8 * e));
        }
    };
    if constexpr (ALIGNED) {
        // straight from the selection's key registers (no second pass over the column sums: with PARTS that was three rows again);
        // the static-mask words of OB steps are in flight together -- one workgroup per CU runs this kernel, so the loop's memory
        // parallelism is what a thread issues itself.  (Unrolled over all NV steps with the loads batched like this it stays under
        // the 128 registers of a 1024-thread workgroup; hoisting every load first is what spilled in round 2.)
        constexpr int OB = 5;
#pragma unroll
        for (int j0 = 0; j0 < NV; j0 += OB) {
            uint32_t sbv[OB];
#pragma unroll
            for (int jj = 0; jj < OB; ++jj) {
                const int c = 4 * tid + 4096 * (j0 + jj);
                sbv[jj] = (j0 + jj < NV && st && c < n) ? *(const uint32_t *)(st + c) : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < OB; ++jj) {
                if constexpr (true) {
                    const int c = 4 * tid + 4096 * (j0 + jj);
                    if (j0 + jj < NV && c < n) emit(std::integral_constant<int, 1>{}, c, key[(j0 + jj) < NV ? (j0 + jj) : 0][0], key[(j0 + jj) < NV ? (j0 + jj) : 0][1], sbv[jj]);
                    asm volatile("" ::: "memory");   // one step's work at a time (the scheduler otherwise interleaves all NV and spills)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int j = 0; j < nsteps; ++j) {
            const int c = 4 * tid + 4096 * j;
            if (c >= n) break;
            uint32_t v0 = 0, v1 = 0, sb = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t u = c + e < n ? x[c + e] : 0xffffu;
                if (e < 2) v0 |= u << (16 * e);
                else v1 |= u << (16 * (e - 2));
                if (st && c + e < n) sb |= (uint32_t)st[c + e] << (8 * e);
            }
            emit(std::integral_constant<int, 0>{}, c, v0, v1, sb);
        }
    }
}

}  // namespace

extern "C" int chipmunk_mask_to_indices(const void *mask, int32_t *indices, int32_t *counts, int64_t rows, int n,
                                        int pad_n, int multiple_of, void *stream) {
    return launch_m2i<false>(mask, indices, counts, rows, n, pad_n, multiple_of, stream);
}

extern "C" int chipmunk_packed_mask_to_indices(const void *packed, int32_t *indices, int32_t *counts, int64_t rows,
                                               int n, int pad_n, int multiple_of, void *stream) {
    return launch_m2i<true>(packed, indices, counts, rows, n, pad_n, multiple_of, stream);
}

static int launch_topk(const void *activation, void *cache, int dtype, int32_t *indices, int32_t *counts, int rows,
                       int cols, double sparsity_amount, int multiple_of, double random_amount, void *stream) {
    CM_CHECK(activation && indices && counts, "topk_indices: null pointer");
    CM_CHECK(rows >= 0 && cols >= 1024, "topk_indices: rows >= 0 and cols >= 1024 required (the quantile is taken over the first 1024 columns); got rows=%d cols=%d", rows, cols);
    CM_CHECK(multiple_of > 0, "topk_indices: multiple_of must be positive");
    CM_CHECK(sparsity_amount >= 0.0 && sparsity_amount <= 1.0, "topk_indices: sparsity_amount must be in [0,1]");
    if (rows == 0) return CHIPMUNK_OK;
    TopkParams p = {activation, cache, indices, counts, rows, cols, multiple_of, (float)sparsity_amount, (float)random_amount,
                    random_amount > 0.0 ? chipmunk_next_random_salt() : 0u};
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TOPK(T)                                                                                           \
    do {                                                                                                         \
        if (cache) hipLaunchKernelGGL((topk_indices_kernel<T, true>), dim3(rows), dim3(1024), 0, s, p);          \
        else hipLaunchKernelGGL((topk_indices_kernel<T, false>), dim3(rows), dim3(1024), 0, s, p);               \
    } while (0)
    switch (dtype) {
        case CHIPMUNK_DTYPE_BF16: LAUNCH_TOPK(uint16_t); break;
        case CHIPMUNK_DTYPE_FP16: LAUNCH_TOPK(_Float16); break;
        case CHIPMUNK_DTYPE_FP32: LAUNCH_TOPK(float); break;
        default: CM_CHECK(false, "topk_indices: unsupported dtype code %d", dtype);
    }
#undef LAUNCH_TOPK
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_mask_to_sorted_indices(const void *mask, int packed, int32_t *indices, int32_t *counts,
                                               int64_t rows, int n, int pad_n, int multiple_of, void *stream) {
    return packed ? launch_m2i<true, true>(mask, indices, counts, rows, n, pad_n, multiple_of, stream)
                  : launch_m2i<false, true>(mask, indices, counts, rows, n, pad_n, multiple_of, stream);
}

extern "C" int chipmunk_topk_indices(const void *activation, int dtype, int32_t *indices, int32_t *counts, int rows,
                                     int cols, double sparsity_amount, int multiple_of, double random_amount,
                                     void *stream) {
    return launch_topk(activation, nullptr, dtype, indices, counts, rows, cols, sparsity_amount, multiple_of,
                       random_amount, stream);
}

extern "C" int chipmunk_topk_delta_indices(const void *activation, void *cache, int dtype, int32_t *indices,
                                           int32_t *counts, int rows, int cols, double sparsity_amount,
                                           int multiple_of, double random_amount, void *stream) {
    CM_CHECK(cache != nullptr, "topk_delta_indices: cache missing");
    return launch_topk(activation, cache, dtype, indices, counts, rows, cols, sparsity_amount, multiple_of,
                       random_amount, stream);
}

extern "C" int chipmunk_copy_indices(const void *src, void *dst, const int32_t *inds, const int32_t *counts, int B,
                                     int M, int R, int F, int elem_size, void *stream) {
    CM_CHECK(src && dst && inds && counts, "copy_indices: null pointer");
    CM_CHECK(B >= 0 && M > 0 && R > 0 && F > 0, "copy_indices: bad sizes");
    CM_CHECK(elem_size == 2 || elem_size == 4, "copy_indices: element size must be 2 or 4 bytes (got %d)", elem_size);
    const int64_t rows = (int64_t)B * M * R;
    if (rows == 0) return CHIPMUNK_OK;
    CM_CHECK(rows < (1ll << 31), "copy_indices: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 2)
        hipLaunchKernelGGL(copy_indices_kernel<uint16_t>, dim3((unsigned)rows), dim3(256), 0, s, (const uint16_t *)src,
                           (uint16_t *)dst, inds, counts, M, R, F);
    else
        hipLaunchKernelGGL(copy_indices_kernel<uint32_t>, dim3((unsigned)rows), dim3(256), 0, s, (const uint32_t *)src,
                           (uint32_t *)dst, inds, counts, M, R, F);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_bitpack(const void *mask, void *packed, int64_t n, void *stream) {
    CM_CHECK(mask && packed && n >= 0, "bitpack: bad arguments");
    if (n == 0) return CHIPMUNK_OK;
    const int64_t nb = (n + 7) >> 3;
    const unsigned grid = (unsigned)((nb + 255) / 256 < 8192 ? (nb + 255) / 256 : 8192);
    hipLaunchKernelGGL(bitpack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)mask,
                       (uint8_t *)packed, n);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_bitunpack(const void *packed, void *mask, int64_t n, void *stream) {
    CM_CHECK(mask && packed && n >= 0, "bitunpack: bad arguments");
    if (n == 0) return CHIPMUNK_OK;
    const int64_t nb = (n + 7) >> 3;
    const unsigned grid = (unsigned)((nb + 255) / 256 < 8192 ? (nb + 255) / 256 : 8192);
    hipLaunchKernelGGL(bitunpack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)packed,
                       (uint8_t *)mask, n);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_transpose16(const void *src, void *dst, int B, int R, int C, void *stream) {
    CM_CHECK(src && dst && B > 0 && R > 0 && C > 0, "transpose16: bad arguments");
    CM_CHECK((C & 7) == 0 || true, "unreachable");
    hipLaunchKernelGGL(transpose16_kernel, dim3((C + 63) / 64, (R + 63) / 64, B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, R, C);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

// =====================================================================================  token reorder (row gather)
// The reference reorders tokens with einops / slicing chains under torch.compile (ops/patch.py:7-80: FLUX two-level
// patch order; ops/voxel.py:9-99: HunyuanVideo / Wan (4,6,8) voxel order with three tail regions, and their inverses).
// Every one of them is a fixed permutation of the token axis, so here it is ONE gather: dst[o, i, :] = src[o, map[i], :]
// with the permutation precomputed once per shape on the host side (chipmunk_amd/ops/_reorder.py) -- one pass at HBM
// rate instead of a reshape / permute / cat / index_put chain (HunyuanVideo: 118 800 rows of 6 KB per direction and step).
template <typename V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const V *src, V *dst, const int32_t *map, int64_t n_out,
                                                          int64_t n_src, int64_t cpr, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / cpr, c = idx - row * cpr;
        const int64_t o = row / n_out, r = row - o * n_out;
        dst[idx] = src[(o * n_src + map[r]) * cpr + c];
    }
}

extern "C" int chipmunk_gather_rows(const void *src, void *dst, const int32_t *map, int64_t outer, int64_t n_src,
                                    int64_t n_out, int64_t row_bytes, void *stream) {
    CM_CHECK(src && dst && map, "gather_rows: null pointer");
    CM_CHECK(outer >= 0 && n_src > 0 && n_out >= 0 && row_bytes > 0, "gather_rows: bad sizes");
    if (outer == 0 || n_out == 0) return CHIPMUNK_OK;
    hipStream_t s = (hipStream_t)stream;
    const uintptr_t align = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes;
#define LAUNCH_GATHER(V)                                                                                              \
    do {                                                                                                              \
        const int64_t cpr = row_bytes / (int64_t)sizeof(V), total = outer * n_out * cpr;                               \
        const int64_t blocks = (total + 255) / 256;                                                                    \
        hipLaunchKernelGGL((gather_rows_kernel<V>), dim3((unsigned)(blocks > 65536 * 16 ? 65536 * 16 : blocks)),       \
                           dim3(256), 0, s, (const V *)src, (V *)dst, map, n_out, n_src, cpr, total);                  \
    } while (0)
    if ((align & 15) == 0) LAUNCH_GATHER(u32x4);
    else if ((align & 3) == 0) LAUNCH_GATHER(uint32_t);
    else if ((align & 1) == 0) LAUNCH_GATHER(uint16_t);
    else LAUNCH_GATHER(uint8_t);
#undef LAUNCH_GATHER
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

int chipmunk_topk_mask_parts(const uint16_t *part, int part_stride, int nrb, int groups_per_bh, int Nq, const void *static_mask, int64_t static_stride,
                             int static_rows, const void *group_flags, void *mask, int rows, int n, int k, double random_amount,
                             hipStream_t s) {
    CM_CHECK(part && mask && rows >= 0 && n > 0 && (n & 3) == 0 && n <= 1024 * 120 && k >= 0 && part_stride >= n && (part_stride & 3) == 0, "topk_mask_parts: bad arguments");
    CM_CHECK(!static_mask || (static_rows > 0 && static_stride >= n && (static_stride & 3) == 0 && ((uintptr_t)static_mask & 3) == 0),
             "topk_mask_parts: bad static mask geometry");
    CM_CHECK(((uintptr_t)mask & 3) == 0 && ((uintptr_t)part & 7) == 0, "topk_mask_parts: unaligned buffers");
    if (rows == 0) return CHIPMUNK_OK;
    TopkMaskParams p = {nullptr, (const uint8_t *)static_mask, (const uint8_t *)group_flags, (uint8_t *)mask, 0, static_stride,
                        rows, n, k, static_mask ? static_rows : 1, (float)random_amount,
                        random_amount > 0.0 ? chipmunk_next_random_salt() : 0u};
    p.parts = part, p.pstride = part_stride, p.nrb = nrb, p.ngroups = groups_per_bh, p.nq = Nq;
    if (n <= 1024 * 16) hipLaunchKernelGGL((topk_mask_kernel<16, true, true>), dim3(rows), dim3(1024), 0, s, p);
    else if (n <= 1024 * 48) hipLaunchKernelGGL((topk_mask_kernel<48, true, true>), dim3(rows), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL((topk_mask_kernel<120, true, true>), dim3(rows), dim3(1024), 0, s, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

// ---- QKV projection output -> attention operands: split + q/k RMSNorm + head-major layout in ONE pass.
// The caller of the attention ops (reference examples/hunyuan/hyvideo/modules/models.py:188-193, 376-381) takes the projection's
// [n, 3*H*128] output apart with rearrange("B L (K H D) -> K B L H D"), applies RMSNorm over the head dimension to q and k
// (norm_layers.py:43-58: bf16(x_f32 * rsqrt(mean(x^2) + eps)) * weight, the product rounded to bf16 again) and transposes all
// three to the [B, H, n, 128] operands the kernels take -- two norm passes and three transposing copies in torch (8+ ms for the
// 2.2 GB of a HunyuanVideo layer).  Here: one 16-lane group per 256-byte (token, q|k|v, head) segment, 16 bytes per lane, the
// sum of squares by DPP inside the group; reads are contiguous over the projection's rows, writes are whole 256-byte rows.
// Optional rotary embedding of q and k (apply_rotary_emb, posemb_layers.py:133-172, the (cos, sin) form): for the first
// `rope_rows` tokens out = bf16(x_f32 * cos + rotate_half(x_f32) * sin), pairs (2i, 2i+1) -> (-x[2i+1], x[2i]); cos / sin fp32 [rope_rows, 128].
// A lane holds 8 consecutive elements = 4 whole pairs, so the rotation is lane-local.
__global__ __launch_bounds__(256) void qkv_split_norm_kernel(const uint16_t *qkv, int64_t row_stride, const uint16_t *qw,
                                                             const uint16_t *kw, uint16_t *q, uint16_t *k, uint16_t *v, int64_t n,
                                                             int H, float eps, int64_t segments, const float *fcos, const float *fsin,
                                                             int64_t rope_rows) {
    const int l15 = threadIdx.x & 15;
    u32x4 wq = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, wk = wq;   // bf16 1.0 pairs
    if (qw) wq = *(const u32x4 *)(qw + l15 * 8);
    if (kw) wk = *(const u32x4 *)(kw + l15 * 8);
    for (int64_t seg = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; seg < segments; seg += ((int64_t)gridDim.x * 256) >> 4) {
        const int64_t tok = seg / (3 * H);
        const int rem = (int)(seg - tok * 3 * H), which = rem / H, h = rem - which * H;
        u32x4 x = *(const u32x4 *)(qkv + tok * row_stride + (int64_t)rem * 128 + l15 * 8);
        uint16_t *dst = (which == 0 ? q : which == 1 ? k : v) + ((int64_t)h * n + tok) * 128 + l15 * 8;
        if (which < 2) {   // (uniform over the 16-lane group; the DPP row sum only mixes lanes of one group)
            float f[8], ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = __uint_as_float(x[e] << 16), f[2 * e + 1] = __uint_as_float(x[e] & 0xffff0000u);
                ss = __builtin_fmaf(f[2 * e], f[2 * e], ss);
                ss = __builtin_fmaf(f[2 * e + 1], f[2 * e + 1], ss);
            }
            ss = row16_sum(ss);
            const float r = 1.0f / __builtin_sqrtf(ss * (1.0f / 128.0f) + eps);   // (IEEE sqrt and divide, as torch.rsqrt on the host)
            const u32x4 w = which == 0 ? wq : wk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = f[2 * e] * r, b = f[2 * e + 1] * r;
                round_bf16_pair(a, b);                                  // .type_as(x)
                x[e] = pack_bf16x2(a * __uint_as_float(w[e] << 16), b * __uint_as_float(w[e] & 0xffff0000u));   // * weight, in bf16
            }
            if (fcos && tok < rope_rows) {
                const f32x4 c0 = *(const f32x4 *)(fcos + tok * 128 + l15 * 8), c1 = *(const f32x4 *)(fcos + tok * 128 + l15 * 8 + 4);
                const f32x4 s0 = *(const f32x4 *)(fsin + tok * 128 + l15 * 8), s1 = *(const f32x4 *)(fsin + tok * 128 + l15 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float re = __uint_as_float(x[e] << 16), im = __uint_as_float(x[e] & 0xffff0000u);
                    const float cr = e < 2 ? c0[2 * e] : c1[2 * e - 4], ci = e < 2 ? c0[2 * e + 1] : c1[2 * e - 3];
                    const float sr = e < 2 ? s0[2 * e] : s1[2 * e - 4], si = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
                    x[e] = pack_bf16x2(re * cr + (-im) * sr, im * ci + re * si);
                }
            }
        }
        *(u32x4 *)dst = x;
    }
}

extern "C" int chipmunk_qkv_split_norm(const void *qkv, int64_t row_stride, const void *q_weight, const void *k_weight, void *q,
                                       void *k, void *v, int64_t n, int heads, float eps, const float *freqs_cos,
                                       const float *freqs_sin, int64_t rope_rows, void *stream) {
    CM_CHECK(qkv && q && k && v, "qkv_split_norm: null pointer");
    CM_CHECK(n >= 0 && heads > 0 && row_stride >= (int64_t)3 * heads * 128, "qkv_split_norm: bad sizes");
    CM_CHECK((((uintptr_t)qkv | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q_weight | (uintptr_t)k_weight) & 15) == 0 &&
                 (row_stride & 7) == 0 && (((uintptr_t)freqs_cos | (uintptr_t)freqs_sin) & 15) == 0,
             "qkv_split_norm: pointers and the row stride must be 16-byte aligned");
    CM_CHECK((freqs_cos == nullptr) == (freqs_sin == nullptr) && rope_rows >= 0 && rope_rows <= n, "qkv_split_norm: bad rotary arguments");
    if (n == 0) return CHIPMUNK_OK;
    const int64_t segments = n * 3 * heads, blocks = (segments + 15) / 16;
    hipLaunchKernelGGL(qkv_split_norm_kernel, dim3((unsigned)(blocks > 256 * 64 ? 256 * 64 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)qkv, row_stride, (const uint16_t *)q_weight, (const uint16_t *)k_weight, (uint16_t *)q,
                       (uint16_t *)k, (uint16_t *)v, n, heads, eps, segments, freqs_cos, freqs_sin, freqs_cos ? rope_rows : 0);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_topk_mask(const void *cs, int64_t cs_stride, const void *static_mask, int64_t static_stride,
                                  int static_rows, const void *group_flags, void *mask, int rows, int n, int k,
                                  double random_amount, void *stream) {
    CM_CHECK(cs && mask, "topk_mask: null pointer");
    CM_CHECK(rows >= 0 && n > 0 && k >= 0, "topk_mask: bad sizes (rows=%d n=%d k=%d)", rows, n, k);
    CM_CHECK(cs_stride >= n, "topk_mask: cs row stride %lld < n %d", (long long)cs_stride, n);
    CM_CHECK(n <= 1024 * 120, "topk_mask: rows of more than 122880 columns are not supported (got %d)", n);
    CM_CHECK(!static_mask || (static_rows > 0 && static_stride >= n), "topk_mask: bad static mask geometry");
    CM_CHECK(random_amount >= 0.0 && random_amount <= 1.0, "topk_mask: random_amount must be in [0,1]");
    if (rows == 0) return CHIPMUNK_OK;
    TopkMaskParams p = {(const uint16_t *)cs, (const uint8_t *)static_mask, (const uint8_t *)group_flags, (uint8_t *)mask,
                        cs_stride, static_stride, rows, n, k, static_mask ? static_rows : 1, (float)random_amount,
                        random_amount > 0.0 ? chipmunk_next_random_salt() : 0u};
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = n % 4 == 0 && (((uintptr_t)cs) & 7) == 0 && (cs_stride * 2) % 8 == 0 && (((uintptr_t)mask) & 3) == 0 &&
                         (!static_mask || ((((uintptr_t)static_mask) & 3) == 0 && static_stride % 4 == 0));
#define LAUNCH_TM(KPT)                                                                                       \
    do {                                                                                                     \
        if (aligned) hipLaunchKernelGGL((topk_mask_kernel<KPT, true>), dim3(rows), dim3(1024), 0, s, p);     \
        else hipLaunchKernelGGL((topk_mask_kernel<KPT, false>), dim3(rows), dim3(1024), 0, s, p);            \
    } while (0)
    if (n <= 1024 * 16) LAUNCH_TM(16);
    else if (n <= 1024 * 48) LAUNCH_TM(48);
    else LAUNCH_TM(120);
#undef LAUNCH_TM
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_compact_indices(const int32_t *indices, int64_t idx_stride, const int32_t *counts, const int64_t *offsets,
                                        int32_t *flat, int64_t rows, void *stream) {
    CM_CHECK(indices && counts && offsets && flat, "compact_indices: null pointer");
    CM_CHECK(rows >= 0 && rows < (1ll << 31) && idx_stride > 0, "compact_indices: bad sizes rows=%lld idx_stride=%lld", (long long)rows,
             (long long)idx_stride);
    if (rows == 0) return CHIPMUNK_OK;
    hipLaunchKernelGGL(compact_indices_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, indices, idx_stride, counts,
                       offsets, flat);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}
