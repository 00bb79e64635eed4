// Column-sparse / dense attention for gfx950 (MI355X), one templated kernel behind five C-ABI entry points.
//
// Replaces the reference's four Hopper kernels (csrc/attn/{csp_attn,csp_128_attn,dense_attn,dense_colsum_attn}.cu).
// Nothing of their structure (TMA, WGMMA, producer/consumer warpgroups, 112/128-row KV tiles) is kept; the design is
// CDNA4-first:
//   * one workgroup = one (batch, head, 192-query group) = 4 waves x 48 query rows, one wave per SIMD,
//     two workgroups per CU (LDS 68 KiB each) so every SIMD holds two waves;
//   * "swapped" QK^T: S^T = K.Q^T with v_mfma_f32_16x16x32_bf16, so a lane owns ONE query column and its softmax
//     statistics (running max, partial sum, rescale factor) are lane-local scalars;
//   * P^T (bf16) is consumed straight from registers as the B operand of O^T += V^T.P^T -- the k-order permutation of
//     the accumulator layout is absorbed by the order in which V^T fragments are fetched (ds_read_b64_tr_b16);
//   * K/V rows (256 B each) are gathered L2 -> LDS by buffer-form LDS-DMA: the per-lane offset does the gather, the XOR
//     swizzle of the 16-byte chunk index is applied on the SOURCE side so the lane-linear LDS image is
//     bank-conflict-free for ds_read_b128 (K) and ds_read_b64_tr_b16 (V);
//   * 32-key tiles in a 4-slot LDS ring (3 tiles in flight), gather indices travelling further ahead through an LDS
//     key ring, one raw s_barrier + counted vmcnt per tile;
//   * scheduling across workgroups: longest-first order for ragged key counts, key-split tail with a last-arriver merge
//     for near-equal items (see launch_attn); no float atomics, every reduction has a fixed order.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int QG = 192;   // query rows per group == per workgroup (reference mbm = 192, modules/attn.py:95-96)
constexpr int QW = 48;    // query rows per wave
constexpr int KVT = 32;   // gathered keys per LDS tile
constexpr int NST = 4;    // LDS ring depth: data of 3 tiles in flight ahead of the one being consumed
constexpr int KRING = 8;  // key ring slots (keys travel NST-1+3 tiles ahead of their use)
constexpr int HD = 128;   // head dim (reference: "Head dimension must be 128", csp_attn.cu:381-383)
constexpr int TILE_BYTES = KVT * HD * 2;  // 8 KiB
constexpr float SCALE_LOG2E = 0.08838834764f * 1.44269504089f;  // csp_128_attn.cu:307
constexpr int NSTV = NST + 1;  // V ring: one slot deeper, tile t-1's V is read during iteration t (PV runs one tile behind QK^T)
constexpr int KEY_RING_OFF = (NST + NSTV) * TILE_BYTES;
constexpr int CS_OFF = KEY_RING_OFF + KRING * 256;
constexpr int ATTN_LDS_BYTES = CS_OFF + 2 * 2 * 4 * KVT * 4;  // column-sum partials [2 iterations][2 tiles][4 waves][KVT]

struct AttnParams {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    const uint16_t *o_in;  // INPLACE only: the accumulation base (o itself for the in-place op, the cache for csp_attn_out)
    int64_t qs[3], ks[3], vs[3], os[3];
    const int32_t *indices, *counts;
    float *l_out;
    const float *p_in;
    uint16_t *cs;
    int cs_stride;
    int B, H, Nq, Nk, G, idx_stride;
    float o_scale;
    // key-split tail (see launch_attn): items >= split_full are handed to `nsplit` workgroups, each over a slice of the
    // item's key tiles; partial (o, m, l) go through `ws`, the last arriver (ticket) merges and runs the epilogue
    int split_full, nsplit;
    float *ws;
    int32_t *tickets;
    // optional work plan for ragged key counts (attn_plan_kernel): block i processes item plan[2i] (< 0: nothing), slice
    // (plan[2i+1] & 0xffff) of (plan[2i+1] >> 16) slices over the item's key tiles; slices of one item are adjacent
    const int32_t *plan;
    int xcd_chunks;  // 1: every XCD walks its own contiguous (head, group) range; 0: all XCDs sweep one head together
    int probe;  // timing probes (tools/kbench.py --variants): 1 = no gathers after the prologue, 2 = gathers only
};

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

#ifdef ATTN_PROF
// Cycle anatomy of the main loop (tools/attn_prof.py builds a separate library with -DATTN_PROF): per wave of
// workgroup 0, s_memtime at the segment boundaries of every tile, summed per segment.  Not part of the product build.
__device__ unsigned long long g_attn_prof[8 * 8];
#define PROF_DECL unsigned long long pt_, pacc_[7] = {0, 0, 0, 0, 0, 0, 0}; const bool prof_on_ = blockIdx.x == (gridDim.x / 2)
#define PROF_START() do { if (prof_on_) pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define PROF_MARK(i) do { if (prof_on_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } } while (0)
#define PROF_END(w, ntiles) do { if (prof_on_ && lane == 0) { for (int i_ = 0; i_ < 7; ++i_) g_attn_prof[(w) * 8 + i_] = pacc_[i_]; g_attn_prof[(w) * 8 + 7] = (ntiles); } } while (0)
extern "C" int chipmunk_attn_prof_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_prof), sizeof(g_attn_prof)) == hipSuccess ? 0 : 2;
}
#else
#define PROF_DECL
#define PROF_START()
#define PROF_MARK(i)
#define PROF_END(w, ntiles)
#endif

// Pin a value to this point of the program: the optimiser may neither sink its computation below nor hoist its uses above
// (the IR passes move pure arithmetic across sched_barrier freely; a hand-placed slice has to materialise where it stands).
template <typename T>
__device__ __forceinline__ void pin(T &x) {
    asm volatile("" : "+v"(x));
}

// three-input maximum as ONE instruction (nested fmaxf on MFMA outputs makes hipcc insert canonicalising v_max first)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max over the four 16-lane rows of a wave (common.h: max_across_rows) without the canonicalising v_max pairs
__device__ __forceinline__ float max_rows(float x) {
    float a = x, b = x;
    lane_swap32(a, b);
    float c = max2(a, b), d = c;
    lane_swap16(c, d);
    return max2(c, d);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Pipeline (per workgroup, t = tile of 32 gathered keys):
//   iteration t:  wait until tile t has landed (counted vmcnt: the 2 younger tiles stay in flight) -> s_barrier
//                 -> issue the LDS-DMA of tile t+3 into the slot tile t-1 just vacated (+ wave 0: the gather keys of
//                 tile t+6 into the key ring) -> QK^T, online softmax, PV on tile t.
//   The gather keys reach the lanes through LDS as well (global_load_lds_dword by wave 0, ds_read_b32 by everybody):
//   an ordinary global load of the keys would make hipcc wait vmcnt(0) at its use and drain the DMA pipeline.
// CSONLY = second pass of dense_colsum_attn: only K is staged, S^T is recomputed and reduced to the 192-row column sums;
// no softmax state, no V, no O accumulators (so it runs at twice the occupancy).  The reference's summand
// exp2(s*c - m*c) * (exp2(m*c) * prev_l) does not depend on the max that centres it, so the pass evaluates it as
// exp2(s*c + log2(prev_l)) -- one fma, one exp2 and one add per score -- in fp32, without the reference's two
// intermediate bf16 roundings.  It has its own loop (two key tiles per barrier) right after the prologue.
template <bool GATHER, bool INPLACE, bool WRITE_L, bool CSONLY = false>
__global__ __launch_bounds__(256, CSONLY ? 4 : 2) void attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KEYOFF = (CSONLY ? NST : NST + NSTV) * TILE_BYTES;  // the column-sum pass has no V ring
    unsigned char *Kl = smem;                      // [NST][TILE_BYTES]
    unsigned char *Vl = smem + NST * TILE_BYTES;   // [NSTV][TILE_BYTES]
    int *key_ring = (int *)(smem + KEYOFF);        // [KRING][64]
    // [2][2][4][KVT]: per-wave column-sum partials of the CSONLY pass (iteration parity, tile parity, wave), summed in
    // wave order (the reference reduces with shared-memory atomics and is order-dependent in the last bits; this is
    // run-to-run deterministic)
    float *cs_acc = (float *)(smem + KEYOFF + KRING * 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    int wid0 = p.xcd_chunks ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    // sp / nsp: this workgroup's slice of the item's key tiles; slot0: scratch slot of the item's slice 0; tix: its ticket
    int sp = 0, nsp = 1, slot0 = 0, tix = 0, wid = wid0;
    if (!CSONLY && p.plan) {
        wid = p.plan[2 * wid0];
        if (wid < 0) return;
        const int meta = p.plan[2 * wid0 + 1];
        sp = meta & 0xffff, nsp = meta >> 16;
        slot0 = tix = wid0 - sp;
    } else if (!CSONLY && p.nsplit > 1 && wid0 >= p.split_full) {
        const int k = wid0 - p.split_full;
        tix = k / p.nsplit;
        sp = k - tix * p.nsplit;
        nsp = p.nsplit;
        slot0 = tix * p.nsplit;
        wid = p.split_full + tix;
    }
    const int bh = wid / p.G, g = wid - bh * p.G;
    const int b = bh / p.H, h = bh - b * p.H;

    const int count = GATHER ? p.counts[(int64_t)bh * p.G + g] : p.Nk;
    // packed positions >= Nk are masked out by the reference (right_fill, csp_128_attn.cu:314)
    const int valid = count < p.Nk ? count : p.Nk;
    const int ntiles = (valid + KVT - 1) / KVT;
    if (!CSONLY && nsp > 1 && !p.plan) {  // every workgroup of the item derives the same effective split: at least 4 key tiles per slice
        const int cap = ntiles / 4 > 1 ? ntiles / 4 : 1;
        nsp = nsp < cap ? nsp : cap;
        if (sp >= nsp) return;
    }
    const int tbeg = (int)((int64_t)ntiles * sp / nsp), tend = (int)((int64_t)ntiles * (sp + 1) / nsp);
    const int32_t *idx = GATHER ? p.indices + ((int64_t)bh * p.G + g) * p.idx_stride : nullptr;
    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t *vbase = p.v + b * p.vs[0] + h * p.vs[1];
    const int row0 = g * QG + w * QW;
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase), vrsrc = make_rsrc(vbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u, vstride_b = (uint32_t)p.vs[2] * 2u;  // row strides in bytes

    // ---- Q^T fragments (B operand): lane = query column li, k = lg*8..lg*8+7 of each 32-wide d step
    bf16x8 qf[3][4];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        const int qrow = row0 + qb * 16 + li;
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qrow * p.qs[2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 z = {};
            qf[qb][ks] = (qrow < p.Nq && !(p.probe & 8)) ? *(const bf16x8 *)(qp + ks * 32 + lg * 8) : z;
            if constexpr (CSONLY) {
                // column-sum pass: q * (log2e / sqrt(D)) rounded to bf16 once per item, so the scores leave the MFMA in
                // the exp2 domain and -- with log2(prev_l) as the accumulator's initial value -- need no per-score fma
                // (the sums only rank columns and are stored as bf16: one more bf16 rounding of q is far inside that)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[qb][ks][e] = (__bf16)((float)qf[qb][ks][e] * SCALE_LOG2E);
            }
        }
    }

    f32x4 cs_off[3];  // CSONLY: log2(prev_l) of query rows qb*16 + lg*4 + 0..3 (that pass computes S, not S^T)
    if constexpr (CSONLY) {
#pragma unroll
        for (int qb = 0; qb < 3; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = row0 + qb * 16 + lg * 4 + r;
                const float pl = qr < p.Nq ? p.p_in[(int64_t)bh * p.Nq + qr] : 0.f;
                // (rows past Nq and rows with prev_l == 0 contribute exp2(-huge) = 0; a finite value keeps 0 * -inf away)
                cs_off[qb][r] = pl > 0.f ? __builtin_amdgcn_logf(pl) : -1.0e30f;
            }
        for (int i = tid; i < 2 * 2 * 4 * KVT; i += 256) cs_acc[i] = 0.f;
    }

    // wave 0 streams 64 indices starting at tile T into key slot T % KRING (only the first 32 are tile T's)
    auto issue_keys = [&](int T) {
        if constexpr (GATHER) {
            if (w == 0) {
                int pos = T * KVT + lane;
                pos = pos < p.idx_stride ? pos : p.idx_stride - 1;
                __builtin_amdgcn_global_load_lds(GLB_PTR(idx + pos), LDS_PTR(key_ring + (T % KRING) * 64), 4, 0, 0);
            }
        }
    };
    // every wave stages rows (2w+i)*4 + lg, i = 0..1, of the K tile and of the V tile: 4 DMA instructions per wave
    auto issue_data = [&](int T) {
        const int slot = T % NST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (w * 2 + i) * 4 + lg;  // row inside the tile
            const int pos = T * KVT + r;
            int key = 0;
            if (pos < valid) {
                key = GATHER ? key_ring[(T % KRING) * 64 + r] : pos;
                key = key < 0 ? 0 : (key >= p.Nk ? p.Nk - 1 : key);  // memory safety for malformed indices
            }
            // buffer-form DMA: per-(b, h) SGPR resource + 32-bit lane offset (row * stride + swizzled 16-byte chunk)
            const uint32_t koff = ((uint32_t)key * kstride_b) + ((uint32_t)(li ^ (r & 15)) << 4);
            const uint32_t voff = ((uint32_t)key * vstride_b) + ((uint32_t)(li ^ ((r & 7) << 1)) << 4);
            blds16(krsrc, koff, 0, Kl + slot * TILE_BYTES + (w * 2 + i) * 1024);
            if constexpr (!CSONLY) blds16(vrsrc, voff, 0, Vl + (T % NSTV) * TILE_BYTES + (w * 2 + i) * 1024);
        }
    };


    f32x4 o[3][8];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb)
#pragma unroll
        for (int db = 0; db < 8; ++db) o[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m[3] = {-INFINITY, -INFINITY, -INFINITY};
    float lsum[3] = {0.f, 0.f, 0.f};

    // ---- prologue: keys of tiles 0..NST-2 synchronously, then the data of those tiles (+ keys NST-1 .. 2NST-3)
    if (tend > tbeg) {
#pragma unroll
        for (int T = 0; T < NST - 1; ++T) issue_keys(tbeg + T);
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int T = 0; T < (CSONLY ? 2 : NST - 1); ++T) {
            if (tbeg + T < tend) {
                issue_data(tbeg + T);
                issue_keys(tbeg + T + NST - 1);
            }
        }
    }

    if constexpr (CSONLY) {
        // ---- column-sum pass: TWO 32-key tiles per barrier (the pass has registers to spare and is bound by the
        //      per-tile synchronisation, not by a pipe: MFMA 48 % busy with one tile per barrier).  Ring use: tiles
        //      t, t+1 are consumed while t+2, t+3 land in the other two slots.
        // S = Q . K^T with the MFMA operands swapped relative to the main loop: s[qb][kt][r] = score(q = qb*16 + lg*4 + r,
        // kv = kt*16 + li), so a lane owns one key and the sum over the wave's 48 queries is mostly in-lane
        auto cs_tile = [&](int t, int it) {
            const unsigned char *Kb = Kl + (t % NST) * TILE_BYTES;
            f32x4 s[3][2];
            auto load_k = [&](int idx) {
                const int kt = idx >> 2, ks = idx & 3;
                const int pc = (ks * 4 + lg) ^ li;
                return *(const bf16x8 *)(Kb + (kt * 16 + li) * 256 + pc * 16);
            };
            bf16x8 kr[3];
            kr[0] = load_k(0), kr[1] = load_k(1);
#pragma unroll
            for (int idx = 0; idx < 8; ++idx) {
                if (idx + 2 < 8) kr[(idx + 2) % 3] = load_k(idx + 2);
                __builtin_amdgcn_sched_barrier(0);
                const int kt = idx >> 2, ks = idx & 3;
                // S*c + log2(prev_l) = (Q*c) . K^T accumulated on top of log2(prev_l)
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) s[qb][kt] = mfma16(qf[qb][ks], kr[idx % 3], ks == 0 ? cs_off[qb] : s[qb][kt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // exp2(s*c + log2 prev_l), summed over this wave's 48 queries: 12 in-lane terms, then the 4 lane rows
            float cacc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                float a = 0.f;
#pragma unroll
                for (int qb = 0; qb < 3; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        a += __builtin_amdgcn_exp2f(s[qb][kt][r]);
                cacc[kt] = t * KVT + kt * 16 + li < valid ? a : 0.f;
            }
            lane_swap32(cacc[0], cacc[1]);      // [0] = {kt0 rows 0-1, kt1 rows 0-1}, [1] = {kt0 rows 2-3, kt1 rows 2-3}
            float x = cacc[0] + cacc[1], y = x;
            lane_swap16(x, y);                  // x = {r0, r0, r2, r2}, y = {r1, r1, r3, r3}
            x += y;                             // lanes 0-15: key li of kt 0, lanes 32-47: key li of kt 1
            if ((lane & 16) == 0) cs_acc[(((it & 1) * 2 + (t & 1)) * 4 + w) * KVT + (lane >> 5) * 16 + li] = x;
        };
        auto flush = [&](int t0, int it) {  // column sums of tiles t0, t0+1 (iteration `it`) -> HBM, by threads 0..63
            if (tid < 2 * KVT) {
                const int pos = t0 * KVT + tid;
                const float *acc = cs_acc + (((it & 1) * 2 + (tid >> 5)) * 4) * KVT + (tid & (KVT - 1));
                const float tot = (acc[0] + acc[KVT]) + (acc[2 * KVT] + acc[3 * KVT]);
                if (pos < p.Nk && pos < ntiles * KVT) p.cs[((int64_t)bh * p.G + g) * p.cs_stride + pos] = f32_to_bf16_bits(tot);
            }
        };
        int it = 0;
        for (int t = 0; t < ntiles; t += 2, ++it) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (t > 0) flush(t - 2, it - 1);
            if (t + 2 < ntiles) issue_data(t + 2);
            if (t + 3 < ntiles) issue_data(t + 3);
            cs_tile(t, it);
            if (t + 1 < ntiles) cs_tile(t + 1, it);
        }
        __syncthreads();
        if (ntiles > 0) flush((ntiles - 1) & ~1, it - 1);
        return;
    }

    PROF_DECL;
    PROF_START();
    {
        // ---- pipelined loop: PV runs ONE TILE BEHIND QK^T.  Iteration t: S(t) = K(t).Q^T, then ONE scheduling region holding
        // the 24 MFMAs of O += V(t-1).P(t-1) and the softmax of S(t): the softmax VALU/exp instructions issue in the
        // shadow of MFMAs that do not depend on them (a lone wave spends ~2100 of its ~2900 cycles per tile outside the
        // matrix pipe, most of it exposed VALU / LDS latency; tools/attn_prof.py).  Costs 12 registers (P of two tiles)
        // and one more V slot in LDS.
        // The reference point m of the exponentials lags the true running maximum by at most MAX_LAG (in exp2 units): p <= 2^4,
        // the relative precision of the bf16 P and of the fp32 sums is unchanged, and the O / l rescale (72 VALU issues) runs
        // only when some query column of the wave outgrows the lag -- VALU issue slots, not MFMA time, bound this loop.
        constexpr float MAX_LAG = 4.0f;
        bf16x8 pq[3];   // P^T of the tile whose PV is pending
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) pq[qb] = (bf16x8){};
        f32x4 s[3][2];
        float alpha[3] = {1.f, 1.f, 1.f};
        auto tile_sync = [&](int t) {   // tile t has landed for everybody; the DMA of tile t+3 goes out
            if (t + NST - 1 <= tend) {
                constexpr int L = 4;
                if (GATHER && w == 0) wait_vmcnt<(NST - 2) * (L + 1)>();
                else wait_vmcnt<(NST - 2) * L>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            PROF_MARK(0);
            if (t + NST - 1 < tend) {
                issue_data(t + NST - 1);
                issue_keys(t + 2 * (NST - 1));
            }
            PROF_MARK(1);
        };
        auto qk_tile = [&](int t) {     // S^T(t) = K(t) . Q^T, dead keys of a ragged last tile masked
            const unsigned char *Kb = Kl + (t % NST) * TILE_BYTES;
            auto load_k = [&](int i) {
                const int kt = i >> 2, ks = i & 3;
                const int pc = (ks * 4 + lg) ^ li;
                return *(const bf16x8 *)(Kb + (kt * 16 + li) * 256 + pc * 16);
            };
            constexpr int FR = 3;
            bf16x8 kr[FR];
#pragma unroll
            for (int i = 0; i < FR - 1; ++i) kr[i] = load_k(i);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + FR - 1 < 8) kr[(i + FR - 1) % FR] = load_k(i + FR - 1);
                __builtin_amdgcn_sched_barrier(0);
                const int kt = i >> 2, ks = i & 3;
                // the first k step starts from the inline constant 0 (an explicit zero fill costs 24 VALU issues per tile)
#pragma unroll
                for (int qb = 0; qb < 3; ++qb)
                    s[qb][kt] = mfma16(kr[i % FR], qf[qb][ks], ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : s[qb][kt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t == ntiles - 1 && (valid & (KVT - 1)) != 0) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool dead = t * KVT + kt * 16 + lg * 4 + r >= valid;
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) s[qb][kt][r] = dead ? -INFINITY : s[qb][kt][r];
                    }
            }
            PROF_MARK(2);
        };
        auto tile_max = [&](int qb) {   // max over the 8 scores a lane holds for query column qb: 3 x v_max3 + 1 x v_max
            float x = max3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            x = max3(x, s[qb][0][3], s[qb][1][0]);
            x = max3(x, s[qb][1][1], s[qb][1][2]);
            return max2(x, s[qb][1][3]);
        };
        auto exp_block = [&](int qb, float msc) {   // p = exp2(s*c - m*c), in place
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], SCALE_LOG2E, -msc));
        };
        auto row_sum = [&](int qb) {    // l += sum of the 8 p a lane holds (in place of s)
            float x = (s[qb][0][0] + s[qb][0][1]) + (s[qb][0][2] + s[qb][0][3]);
            x += (s[qb][1][0] + s[qb][1][1]) + (s[qb][1][2] + s[qb][1][3]);
            lsum[qb] += x;
        };
        auto to_bf16 = [&](int qb) {
            bf16x8 pk;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[kt * 4 + r] = (__bf16)s[qb][kt][r];
            return pk;
        };
        auto pv_mfmas = [&](int tv) {   // O^T += V^T(tv) . P^T(pq): 16 transpose reads + 24 MFMAs
            const unsigned char *Vb = Vl + (tv % NSTV) * TILE_BYTES;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const int row_a = lg * 4 + (li >> 2);
                const int chunk = (db * 2 + ((li & 3) >> 1)) ^ ((row_a & 7) << 1);
                const unsigned char *va = Vb + row_a * 256 + chunk * 16 + (li & 1) * 8;
                const s16x4 lo = lds_read_tr16_b64(va);
                const s16x4 hi = lds_read_tr16_b64(va + 16 * 256);
                const bf16x8 vf = __builtin_bit_cast(
                    bf16x8, (__attribute__((ext_vector_type(8))) short){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) o[qb][db] = mfma16(vf, pq[qb], o[qb][db]);
            }
        };
        if (tend > tbeg) {
            // ---- first tile: the reference point is its own maximum (exact), nothing to rescale
            tile_sync(tbeg);
            qk_tile(tbeg);
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                m[qb] = max_rows(tile_max(qb));
                exp_block(qb, m[qb] * SCALE_LOG2E);
                row_sum(qb);
                pq[qb] = to_bf16(qb);
            }
            PROF_MARK(3);
            for (int t = tbeg + 1; t < tend; ++t) {
                tile_sync(t);
                qk_tile(t);
                // ---- PV of tile t-1 and the softmax of tile t, hand-interleaved: 8 chunks, each = 3 MFMAs (one 16-wide d
                // block) + the V^T fragment of the block two ahead + one slice of the softmax, fenced so that the slices
                // stay in the shadow of MFMAs that do not depend on them (left to itself hipcc runs the whole softmax first
                // and the MFMAs after it; sched_group_barrier requests did not move it).
                {
                    const unsigned char *Vb = Vl + ((t - 1) % NSTV) * TILE_BYTES;
                    auto load_v = [&](int db) {
                        const int row_a = lg * 4 + (li >> 2);
                        const int chunk = (db * 2 + ((li & 3) >> 1)) ^ ((row_a & 7) << 1);
                        const unsigned char *va = Vb + row_a * 256 + chunk * 16 + (li & 1) * 8;
                        const s16x4 lo = lds_read_tr16_b64(va);
                        const s16x4 hi = lds_read_tr16_b64(va + 16 * 256);
                        return __builtin_bit_cast(
                            bf16x8, (__attribute__((ext_vector_type(8))) short){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                    };
                    bf16x8 vr[3];
                    vr[0] = load_v(0), vr[1] = load_v(1);
                    float mx[3], msc[3];
                    __builtin_amdgcn_sched_barrier(0);
                    // chunks 0, 1: maxima; then the (rare) reference update in its own block; chunks 2..7: exp2, row sums
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        vr[(db + 2) % 3] = load_v(db + 2);
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) o[qb][db] = mfma16(vr[db % 3], pq[qb], o[qb][db]);
                        if (db == 0) {
#pragma unroll
                            for (int qb = 0; qb < 3; ++qb) {
                                mx[qb] = tile_max(qb);
                                pin(mx[qb]);
                            }
                        } else {
#pragma unroll
                            for (int qb = 0; qb < 3; ++qb) {
                                mx[qb] = max_rows(mx[qb]);
                                pin(mx[qb]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;   // the lag in units of the raw scores
                    if (!__all(mx[0] <= m[0] + LAG_RAW && mx[1] <= m[1] + LAG_RAW && mx[2] <= m[2] + LAG_RAW)) {
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) {
                            const float m_new = max2(m[qb], mx[qb]);
                            alpha[qb] = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
                            lsum[qb] *= alpha[qb];   // (o is rescaled once the pending PV has been accumulated)
                            m[qb] = m_new;
                        }
                    }
#pragma unroll
                    for (int qb = 0; qb < 3; ++qb) msc[qb] = m[qb] * SCALE_LOG2E;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int db = 2; db < 8; ++db) {
                        if (db + 2 < 8) vr[(db + 2) % 3] = load_v(db + 2);
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) o[qb][db] = mfma16(vr[db % 3], pq[qb], o[qb][db]);
                        if (db <= 4) {              // exp2 of one query block per chunk
                            const int qb = db - 2;
                            exp_block(qb, msc[qb]);
                            pin(s[qb][0]);
                            pin(s[qb][1]);
                        } else {                    // row sums
                            const int qb = db - 5;
                            row_sum(qb);
                            pin(lsum[qb]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // P^T of tile t as bf16: only once the last MFMA on tile t-1's P has been issued (a second set of P
                    // registers would push the kernel over 256 VGPRs)
#pragma unroll
                    for (int qb = 0; qb < 3; ++qb) pq[qb] = to_bf16(qb);
                }
                PROF_MARK(3);
                // rescale AFTER the pending PV has been accumulated: O_t = alpha_t (O_{t-1} + P_{t-1} V_{t-1}) + P_t V_t
                if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f && alpha[2] == 1.0f)) {
#pragma unroll
                    for (int qb = 0; qb < 3; ++qb) {
#pragma unroll
                        for (int db = 0; db < 8; ++db) o[qb][db] *= alpha[qb];
                        alpha[qb] = 1.f;
                    }
                }
                PROF_MARK(4);
            }
            pv_mfmas(tend - 1);
        }
    }
    PROF_END(w, tend - tbeg);

    if (nsp > 1) {
        // ---- key-split item: publish this slice's (o, m, l) lane-linear (26 float4 per lane), take a ticket; the last
        //      arriver folds the other slices in (the lane layout is the same in every slice, so the merge is the
        //      online-softmax rescale element by element) and alone runs the epilogue.
        //      Visibility: plain stores -> barrier -> one agent-scope release (L2 write-back) -> drained -> relaxed ticket;
        //      last arriver: ticket -> one agent-scope acquire -> barrier -> plain loads (MI355X_MICROARCH.md, hand-offs).
        //      (A fence-free variant on write-through stores and L1-bypassing loads issued through inline asm produced
        //      wrong elements at HunyuanVideo scale in one build and none in the next: loads the compiler cannot see
        //      are not worth ~1 % of a launch.)
        int *ticket_s = (int *)cs_acc;
        f32x4 *mine = (f32x4 *)p.ws + (int64_t)(slot0 + sp) * (26 * 256) + tid;
#pragma unroll
        for (int qb = 0; qb < 3; ++qb)
#pragma unroll
            for (int db = 0; db < 8; ++db) mine[(qb * 8 + db) * 256] = o[qb][db];
        mine[24 * 256] = (f32x4){m[0], m[1], m[2], 0.f};
        mine[25 * 256] = (f32x4){lsum[0], lsum[1], lsum[2], 0.f};
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the fence's own wait (see guide)
            *ticket_s = __hip_atomic_fetch_add(p.tickets + tix, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*ticket_s != nsp - 1) return;
        if (tid == 0) {
            __hip_atomic_store(p.tickets + tix, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // fold ALL slices (the own one included, from its published copy) in slice order: the result does not depend on
        // which workgroup happened to arrive last
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) {
            m[qb] = -INFINITY, lsum[qb] = 0.f;
#pragma unroll
            for (int db = 0; db < 8; ++db) o[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int s2 = 0; s2 < nsp; ++s2) {
            const f32x4 *oth = (const f32x4 *)p.ws + (int64_t)(slot0 + s2) * (26 * 256) + tid;
            const f32x4 ms = oth[24 * 256], ls = oth[25 * 256];
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                const float m_new = fmaxf(m[qb], ms[qb]);
                if (m_new == -INFINITY) continue;  // nothing so far and an empty slice
                const float a = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
                const float c = __builtin_amdgcn_exp2f((ms[qb] - m_new) * SCALE_LOG2E);
                m[qb] = m_new;
                lsum[qb] = lsum[qb] * a + ls[qb] * c;
#pragma unroll
                for (int db = 0; db < 8; ++db) o[qb][db] = o[qb][db] * a + oth[(qb * 8 + db) * 256] * c;
            }
        }
    }
    // ---- epilogue: O = O^T / l ; lane holds 4 consecutive d of one query row per (qb, db)
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        const float l = sum_across_rows(lsum[qb]);
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        const int qrow = row0 + qb * 16 + li;
        if (qrow >= p.Nq) continue;
        const int64_t ooff = b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + lg * 4;
        uint16_t *op = p.o + ooff;
        const uint16_t *oin = INPLACE ? p.o_in + ooff : nullptr;
        if (INPLACE && ntiles == 0) {  // nothing to add: in place leaves o alone, out of place copies the base
            if (oin != op) {
#pragma unroll
                for (int db = 0; db < 8; ++db) *(u32x2 *)(op + db * 16) = *(const u32x2 *)(oin + db * 16);
            }
            continue;
        }
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            float x0 = o[qb][db][0] * inv, x1 = o[qb][db][1] * inv, x2 = o[qb][db][2] * inv, x3 = o[qb][db][3] * inv;
            u32x2 out;
            if constexpr (INPLACE) {
                // bf16 store of o_scale*result, then bf16 reduce-add into o (csp_attn.cu:294-300)
                u32x2 old = {0u, 0u};
                if (!(p.probe & 4)) old = *(const u32x2 *)(oin + db * 16);
                const float a0 = round_bf16(x0 * p.o_scale), a1 = round_bf16(x1 * p.o_scale);
                const float a2 = round_bf16(x2 * p.o_scale), a3 = round_bf16(x3 * p.o_scale);
                out[0] = pack_bf16x2(__uint_as_float(old[0] << 16) + a0, __uint_as_float(old[0] & 0xffff0000u) + a1);
                out[1] = pack_bf16x2(__uint_as_float(old[1] << 16) + a2, __uint_as_float(old[1] & 0xffff0000u) + a3);
            } else {
                out[0] = pack_bf16x2(x0, x1);
                out[1] = pack_bf16x2(x2, x3);
            }
            if (!(p.probe & 16) || out[0] == 0x12345678u) *(u32x2 *)(op + db * 16) = out;
        }
        if constexpr (WRITE_L) {
            // l = 1 / (exp2(m*c) * norm) = 1 / sum_j exp(s_ij / sqrt(D))   (dense_attn.cu:225-227)
            if (lg == 0) {
                p.l_out[(int64_t)bh * p.Nq + qrow] = 1.0f / (__builtin_amdgcn_exp2f(m[qb] * SCALE_LOG2E) * l);
            }
        }
    }
}

// ======================================================================================================================
// One-wave-per-SIMD form of the main loop ("W96"): a workgroup is TWO waves of 96 query rows each (one 192-row group),
// two workgroups per CU, so every wave owns a SIMD and its whole 512-entry register file.
//
// Why (tools/attn_prof.py, DESIGN.md 4.1): at two waves per SIMD a tile costs ~3150 cycles per 96 query rows against
// ~1550 of matrix-pipe time -- exposed latencies, and a 256-register budget that blocks the remaining VALU savings.  With
// 96 rows per wave the K / V fragments are read from LDS once per 96 rows instead of once per 48, the 102 MFMAs of a tile
// carry the ~170 VALU instructions of the same wave in their shadow, and the register file holds what the 4-wave kernel
// cannot:  (a) Q pre-scaled by log2e/sqrt(D) -- the scores leave the
// matrix pipe as s*c and the softmax is max -> sub -> exp2 -> sum -> bf16 with no multiply per score;  (b) a
// second P^T buffer, so the bf16 conversion of tile t also sits in the shadow of the PV MFMAs of tile t-1.
// hipcc cannot allocate such a kernel (it shuttles values between the two halves of the register file with 600+
// v_accvgpr moves per tile), so the MFMA-only state lives in ACCUMULATOR REGISTERS NAMED BY HAND and is touched by inline
// asm only:   a[0:191]   O^T accumulators, block (qb, db) at a[(qb*8+db)*4 .. +3]
//             a[192:255] the pre-scaled Q^T fragments of query blocks 2..5 (B operands of their QK^T MFMAs)
// -- the WHOLE accumulator half, so the compiler has nowhere to spill but scratch (which the audit forbids): it uses
// accumulator registers as spill space on its own whenever it runs out of VGPRs, clobber lists notwithstanding.
// Everything VALU touches (scores, P, softmax state, Q^T of blocks 0-1, addresses) stays with the compiler in v0..v255.
// Hazards the compiler cannot see are padded inside the strings (MI355X guide 5.7: MFMA D -> VALU reader 12 states).
// The audit after every edit: no compiler v_accvgpr_* and no scratch in the .s of this kernel.
constexpr int W96_QB = 6;
constexpr int W96_LDS_BYTES = (NST + NSTV) * TILE_BYTES + KRING * 256 + 64;
constexpr int W96_Q0 = 192;

template <int V>
using ic = std::integral_constant<int, V>;
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}
// acc[BASE..BASE+3] += A . B   (accumulator-file C/D, VGPR A and B)
template <int BASE>
__device__ __forceinline__ void mfma_acc(bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(BASE), "i"(BASE + 3));
}
// d = A . acc[QBASE..+3] + c   (VGPR C/D, B operand from the accumulator file); FIRST: d is a fresh register set
template <int QBASE, bool FIRST, bool LAST>
__device__ __forceinline__ void mfma_qacc(f32x4 &d, bf16x8 a) {
    if constexpr (FIRST) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(QBASE), "i"(QBASE + 3));
    } else if constexpr (LAST) {   // the scores are read by VALU next: pad the MFMA D -> VALU hazard here
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0\n\ts_nop 7\n\ts_nop 4" : "+v"(d) : "v"(a), "i"(QBASE), "i"(QBASE + 3));
    } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(QBASE), "i"(QBASE + 3));
    }
}
// d = A . B (+ d), everything in VGPRs -- as an asm statement too: a builtin MFMA lets hipcc pick accumulator registers
// for its C/D (it did: a0..a15, on top of O), and this kernel owns all of them
template <bool FIRST>
__device__ __forceinline__ void mfma_vv(f32x4 &d, bf16x8 a, bf16x8 b) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
template <int BASE>
__device__ __forceinline__ f32x4 acc_read4() {
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
    return (f32x4){x0, x1, x2, x3};
}
template <int BASE>
__device__ __forceinline__ void acc_write4(f32x4 x) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3\n\ts_nop 1"
                 ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
}
template <int BASE>
__device__ __forceinline__ void acc_scale4(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0\n\t"
                 "v_accvgpr_read_b32 %0, a%c3\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c3, %0\n\t"
                 "v_accvgpr_read_b32 %0, a%c4\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c4, %0\n\t"
                 "v_accvgpr_read_b32 %0, a%c5\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c5, %0"
                 : "=&v"(t)
                 : "v"(f), "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
}

template <bool GATHER, bool INPLACE, bool WRITE_L>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_w96_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Kl = smem;                      // [NST][TILE_BYTES]
    unsigned char *Vl = smem + NST * TILE_BYTES;   // [NSTV][TILE_BYTES]
    int *key_ring = (int *)(smem + (NST + NSTV) * TILE_BYTES);  // [KRING][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // the accumulator registers this kernel owns: zero O and the row sums (the clobber list also makes the kernel
    // descriptor allocate the accumulator half of the register file)
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0\n\tv_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0\n\tv_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0\n\tv_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0\n\tv_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

    const int wid0 = (int)blockIdx.x;
    int sp = 0, nsp = 1, slot0 = 0, tix = 0, wid = wid0;
    if (p.plan) {
        wid = p.plan[2 * wid0];
        if (wid < 0) return;
        const int meta = p.plan[2 * wid0 + 1];
        sp = meta & 0xffff, nsp = meta >> 16;
        slot0 = tix = wid0 - sp;
    } else if (p.nsplit > 1 && wid0 >= p.split_full) {
        const int k = wid0 - p.split_full;
        tix = k / p.nsplit;
        sp = k - tix * p.nsplit;
        nsp = p.nsplit;
        slot0 = tix * p.nsplit;
        wid = p.split_full + tix;
    }
    const int bh = wid / p.G, g = wid - bh * p.G;
    const int b = bh / p.H, h = bh - b * p.H;
    const int count = GATHER ? p.counts[(int64_t)bh * p.G + g] : p.Nk;
    const int valid = count < p.Nk ? count : p.Nk;
    const int ntiles = (valid + KVT - 1) / KVT;
    if (nsp > 1 && !p.plan) {
        const int cap = ntiles / 4 > 1 ? ntiles / 4 : 1;
        nsp = nsp < cap ? nsp : cap;
        if (sp >= nsp) return;
    }
    const int tbeg = (int)((int64_t)ntiles * sp / nsp), tend = (int)((int64_t)ntiles * (sp + 1) / nsp);
    const int32_t *idx = GATHER ? p.indices + ((int64_t)bh * p.G + g) * p.idx_stride : nullptr;
    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t *vbase = p.v + b * p.vs[0] + h * p.vs[1];
    const int row0 = g * QG + w * (W96_QB * 16);
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase), vrsrc = make_rsrc(vbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u, vstride_b = (uint32_t)p.vs[2] * 2u;

    // ---- Q^T fragments; query blocks 2..5 -> a[192:255]
    auto load_q = [&](int qb, int ks) {
        const int qrow = row0 + qb * 16 + li;
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (int64_t)qrow * p.qs[2];
        bf16x8 z = {};
        return qrow < p.Nq ? *(const bf16x8 *)(qp + ks * 32 + lg * 8) : z;
    };
    bf16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = load_q(qb, ks);
    static_for<0, 16>([&](auto I) {
        constexpr int qb = 2 + I.value / 4, ks = I.value % 4;
        acc_write4<W96_Q0 + (qb - 2) * 16 + ks * 4>(__builtin_bit_cast(f32x4, load_q(qb, ks)));
    });

    auto issue_keys = [&](int T) {
        if constexpr (GATHER) {
            if (w == 0) {
                int pos = T * KVT + lane;
                pos = pos < p.idx_stride ? pos : p.idx_stride - 1;
                __builtin_amdgcn_global_load_lds(GLB_PTR(idx + pos), LDS_PTR(key_ring + (T % KRING) * 64), 4, 0, 0);
            }
        }
    };
    // every wave stages rows (4w+i)*4 + lg, i = 0..3, of the K tile and of the V tile: 8 DMA instructions per wave and tile
    auto issue_data = [&](int T) {
        const int slot = T % NST;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (w * 4 + i) * 4 + lg;
            const int pos = T * KVT + r;
            int key = 0;
            if (pos < valid) {
                key = GATHER ? key_ring[(T % KRING) * 64 + r] : pos;
                key = key < 0 ? 0 : (key >= p.Nk ? p.Nk - 1 : key);
            }
            const uint32_t koff = ((uint32_t)key * kstride_b) + ((uint32_t)(li ^ (r & 15)) << 4);
            const uint32_t voff = ((uint32_t)key * vstride_b) + ((uint32_t)(li ^ ((r & 7) << 1)) << 4);
            blds16(krsrc, koff, 0, Kl + slot * TILE_BYTES + (w * 4 + i) * 1024);
            blds16(vrsrc, voff, 0, Vl + (T % NSTV) * TILE_BYTES + (w * 4 + i) * 1024);
        }
    };

    f32x4 s[W96_QB][2];
    bf16x8 pq[W96_QB];
    float m[W96_QB], alpha[W96_QB], lsum[W96_QB];   // m: reference point of the exponentials, in exp2 units (scaled scores)
#pragma unroll
    for (int qb = 0; qb < W96_QB; ++qb) {
        pq[qb] = (bf16x8){};
        m[qb] = 0.f, alpha[qb] = 1.f, lsum[qb] = 0.f;
    }

    if (tend > tbeg) {
#pragma unroll
        for (int T = 0; T < NST - 1; ++T) issue_keys(tbeg + T);
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int T = 0; T < NST - 1; ++T) {
            if (tbeg + T < tend) {
                issue_data(tbeg + T);
                issue_keys(tbeg + T + NST - 1);
            }
        }
    }

    constexpr float MAX_LAG = 4.0f;                     // in exp2 units
    constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;    // ... and in units of the raw scores
    // m[]: reference point of the exponentials in RAW score units (p = exp2(s*c - m*c)), msc[] = m*c
    float msc[W96_QB];
#pragma unroll
    for (int qb = 0; qb < W96_QB; ++qb) msc[qb] = 0.f;

    auto wait_tile = [&](int t) {   // tile t has landed for everybody
        if (t + NST - 1 <= tend) {
            constexpr int L = 8;
            if (GATHER && w == 0) wait_vmcnt<(NST - 2) * (L + 1)>();
            else wait_vmcnt<(NST - 2) * L>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
    };
    // one LDS-DMA piece pair (K rows and V rows 4*(4w+i) + lg of tile T): rides in a QK^T MFMA gap
    auto issue_piece = [&](int T, int i) {
        const int r = (w * 4 + i) * 4 + lg;
        const int pos = T * KVT + r;
        int key = 0;
        if (pos < valid) {
            key = GATHER ? key_ring[(T % KRING) * 64 + r] : pos;
            key = key < 0 ? 0 : (key >= p.Nk ? p.Nk - 1 : key);
        }
        const uint32_t koff = ((uint32_t)key * kstride_b) + ((uint32_t)(li ^ (r & 15)) << 4);
        const uint32_t voff = ((uint32_t)key * vstride_b) + ((uint32_t)(li ^ ((r & 7) << 1)) << 4);
        blds16(krsrc, koff, 0, Kl + (T % NST) * TILE_BYTES + (w * 4 + i) * 1024);
        blds16(vrsrc, voff, 0, Vl + (T % NSTV) * TILE_BYTES + (w * 4 + i) * 1024);
    };
    auto load_k = [&](const unsigned char *Kb, int i) {
        const int kt = i >> 2, ks = i & 3;
        const int pc = (ks * 4 + lg) ^ li;
        return *(const bf16x8 *)(Kb + (kt * 16 + li) * 256 + pc * 16);
    };
    auto load_v = [&](const unsigned char *Vb, int db) {
        const int row_a = lg * 4 + (li >> 2);
        const int chunk = (db * 2 + ((li & 3) >> 1)) ^ ((row_a & 7) << 1);
        const unsigned char *va = Vb + row_a * 256 + chunk * 16 + (li & 1) * 8;
        const s16x4 lo = lds_read_tr16_b64(va);
        const s16x4 hi = lds_read_tr16_b64(va + 16 * 256);
        return __builtin_bit_cast(bf16x8, (__attribute__((ext_vector_type(8))) short){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };
    auto mask_tail = [&](f32x4 (&sx)[W96_QB][2], int t) {
        if (t == ntiles - 1 && (valid & (KVT - 1)) != 0) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = t * KVT + kt * 16 + lg * 4 + r >= valid;
#pragma unroll
                    for (int qb = 0; qb < W96_QB; ++qb) sx[qb][kt][r] = dead ? -INFINITY : sx[qb][kt][r];
                }
        }
    };
    auto raise_reference = [&](int qb, float m_new) {   // m_new >= m: move the reference point up
        alpha[qb] = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
        m[qb] = m_new;
        msc[qb] = m_new * SCALE_LOG2E;
        lsum[qb] *= alpha[qb];
    };

    // ---- phase 1 of iteration t: S(t) = K(t) . Q^T into `sn` (48 MFMAs) with, in the MFMA gaps, the tail of tile t-1's
    //      softmax on `sc` (row sums, bf16 -> pq) and the LDS-DMA pieces of tile t+3.  `tail` / `dma`: do those parts.
    auto qk_phase = [&](f32x4 (&sn)[W96_QB][2], f32x4 (&sc)[W96_QB][2], int t, bool tail, bool dma) {
        const unsigned char *Kb = Kl + (t % NST) * TILE_BYTES;
        bf16x8 kr[3];
        kr[0] = load_k(Kb, 0), kr[1] = load_k(Kb, 1);
        static_for<0, 8>([&](auto I) {
            constexpr int i = I.value, kt = i >> 2, ks = i & 3;
            if constexpr (i + 2 < 8) kr[(i + 2) % 3] = load_k(Kb, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, W96_QB>([&](auto Q) {
                constexpr int qb = Q.value;
                if constexpr (qb < 2) mfma_vv<ks == 0>(sn[qb][kt], kr[i % 3], qf[qb][ks]);
                else mfma_qacc<W96_Q0 + (qb - 2) * 16 + ks * 4, ks == 0, ks == 3 && qb == W96_QB - 1>(sn[qb][kt], kr[i % 3]);
                // gap work: group i < 6 finishes query block i of the previous tile: 8 adds + 4 converts over 6 gaps
                if constexpr (i < W96_QB) {
                    if (tail) {
                        if constexpr (qb == 0) {
                            pin(sc[i][0]);
                            pin(sc[i][1]);
                        }
                        if constexpr (qb == 0) lsum[i] += sc[i][0][0] + sc[i][0][1];
                        if constexpr (qb == 1) lsum[i] += sc[i][0][2] + sc[i][0][3];
                        if constexpr (qb == 2) lsum[i] += sc[i][1][0] + sc[i][1][1];
                        if constexpr (qb == 3) lsum[i] += sc[i][1][2] + sc[i][1][3];
                        if constexpr (qb == 4) {
                            bf16x8 pk = pq[i];
                            pk[0] = (__bf16)sc[i][0][0], pk[1] = (__bf16)sc[i][0][1], pk[2] = (__bf16)sc[i][0][2], pk[3] = (__bf16)sc[i][0][3];
                            pq[i] = pk;
                        }
                        if constexpr (qb == 5) {
                            bf16x8 pk = pq[i];
                            pk[4] = (__bf16)sc[i][1][0], pk[5] = (__bf16)sc[i][1][1], pk[6] = (__bf16)sc[i][1][2], pk[7] = (__bf16)sc[i][1][3];
                            pq[i] = pk;
                        }
                        pin(lsum[i]);
                        pin(pq[i]);
                    }
                }
            });
            if (dma) {
                if constexpr (i < 4) issue_piece(t + NST - 1, i);
                if constexpr (i == 4) issue_keys(t + 2 * (NST - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        mask_tail(sn, t);
    };
    // ---- phase 2 of iteration t: O^T += V^T(t-1) . P^T(pq) (48 MFMAs) with, in the gaps, the head of tile t's softmax
    //      on `sn`: in-lane max (d block 0), row max (1), [rare: reference update], exp2 of query block db-2 (2..7)
    auto pv_phase = [&](f32x4 (&sn)[W96_QB][2], int tv, bool head) {
        const unsigned char *Vb = Vl + (tv % NSTV) * TILE_BYTES;
        bf16x8 vr[3];
        float mx[W96_QB];
        vr[0] = load_v(Vb, 0), vr[1] = load_v(Vb, 1);
        static_for<0, 8>([&](auto DBc) {
            constexpr int db = decltype(DBc)::value;
            if constexpr (db + 2 < 8) vr[(db + 2) % 3] = load_v(Vb, db + 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (db == 2) {
                if (head) {   // every row maximum is known: does any query column of the wave outgrow the lag?
                    bool quiet = true;
#pragma unroll
                    for (int qb = 0; qb < W96_QB; ++qb) quiet = quiet && (mx[qb] <= m[qb] + LAG_RAW);
                    if (!__all(quiet)) {
#pragma unroll
                        for (int qb = 0; qb < W96_QB; ++qb) raise_reference(qb, fmaxf(m[qb], mx[qb]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            static_for<0, W96_QB>([&](auto Q) {
                constexpr int qb = Q.value;
                mfma_acc<(qb * 8 + db) * 4>(vr[db % 3], pq[qb]);
                if (head) {
                    if constexpr (db == 0) {            // in-lane max of query block qb
                        pin(sn[qb][0]);
                        pin(sn[qb][1]);
                        float x = max3(sn[qb][0][0], sn[qb][0][1], sn[qb][0][2]);
                        x = max3(x, sn[qb][0][3], sn[qb][1][0]);
                        x = max3(x, sn[qb][1][1], sn[qb][1][2]);
                        mx[qb] = max2(x, sn[qb][1][3]);
                        pin(mx[qb]);
                    } else if constexpr (db == 1) {     // max across the 4 lane rows
                        pin(mx[qb]);
                        mx[qb] = max_rows(mx[qb]);
                        pin(mx[qb]);
                    } else {                            // exp2 of query block db-2: 8 scores over 6 gaps
                        constexpr int e = db - 2;
                        constexpr int lo = qb < 2 ? qb * 2 : qb + 2, hi = qb < 2 ? qb * 2 + 2 : qb + 3;
                        if constexpr (qb == 0) {
                            pin(sn[e][0]);
                            pin(sn[e][1]);
                        }
#pragma unroll
                        for (int j = lo; j < hi; ++j)
                            sn[e][j >> 2][j & 3] = __builtin_amdgcn_exp2f(__builtin_fmaf(sn[e][j >> 2][j & 3], SCALE_LOG2E, -msc[e]));
                        pin(sn[e][0]);
                        pin(sn[e][1]);
                    }
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto rescale_if_needed = [&]() {   // AFTER the pending PV has been accumulated (its MFMAs must have written back)
        bool unit = true;
#pragma unroll
        for (int qb = 0; qb < W96_QB; ++qb) unit = unit && (alpha[qb] == 1.0f);
        if (!__all(unit)) {
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            static_for<0, W96_QB>([&](auto Q) {
                constexpr int qb = Q.value;
                static_for<0, 8>([&](auto DBc) { acc_scale4<(qb * 8 + decltype(DBc)::value) * 4>(alpha[qb]); });
                alpha[qb] = 1.f;
            });
            asm volatile("s_nop 1" ::: "memory");
        }
    };

    f32x4 sB[W96_QB][2];   // the second score buffer: S(t) is produced while P(t-1) is still being summed / converted
    if (tend > tbeg) {
        // ---- first tile: its own maximum becomes the reference point (exact), nothing to rescale
        wait_tile(tbeg);
        if (tbeg + NST - 1 < tend) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(tbeg + NST - 1, i);
            issue_keys(tbeg + 2 * (NST - 1));
        }
        qk_phase(s, sB, tbeg, false, false);
#pragma unroll
        for (int qb = 0; qb < W96_QB; ++qb) {
            float x = max3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
            x = max3(x, s[qb][0][3], s[qb][1][0]);
            x = max3(x, s[qb][1][1], s[qb][1][2]);
            m[qb] = max_rows(max2(x, s[qb][1][3]));
            m[qb] = m[qb] == -INFINITY ? 0.f : m[qb];
            msc[qb] = m[qb] * SCALE_LOG2E;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], SCALE_LOG2E, -msc[qb]));
        }
        // steady state, two tiles per trip so that the two score buffers keep their names
        int t = tbeg + 1;
        for (; t + 1 < tend; t += 2) {
            wait_tile(t);
            qk_phase(sB, s, t, true, t + NST - 1 < tend);
            pv_phase(sB, t - 1, true);
            rescale_if_needed();
            wait_tile(t + 1);
            qk_phase(s, sB, t + 1, true, t + NST < tend);
            pv_phase(s, t, true);
            rescale_if_needed();
        }
        if (t < tend) {   // one more tile: ends with P in sB
            wait_tile(t);
            qk_phase(sB, s, t, true, t + NST - 1 < tend);
            pv_phase(sB, t - 1, true);
            rescale_if_needed();
#pragma unroll
            for (int qb = 0; qb < W96_QB; ++qb) s[qb][0] = sB[qb][0], s[qb][1] = sB[qb][1];
        }
        // ---- drain: the last tile's row sums and bf16 P (in s), then its PV
#pragma unroll
        for (int qb = 0; qb < W96_QB; ++qb) {
            lsum[qb] += (s[qb][0][0] + s[qb][0][1]) + (s[qb][0][2] + s[qb][0][3]);
            lsum[qb] += (s[qb][1][0] + s[qb][1][1]) + (s[qb][1][2] + s[qb][1][3]);
            bf16x8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = (__bf16)s[qb][j >> 2][j & 3];
            pq[qb] = pk;
        }
        pv_phase(s, tend - 1, false);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs' results before the accumulator file is read

#pragma unroll
    for (int qb = 0; qb < W96_QB; ++qb) lsum[qb] = sum_across_rows(lsum[qb]);   // the 4 lane rows hold 8 keys of every tile each

    if (nsp > 1) {
        // key-split item: same hand-off as attn_kernel; 52 float4 per thread x 128 threads = the same 104 KB slot
        int *ticket_s = key_ring;
        f32x4 *mine = (f32x4 *)p.ws + (int64_t)(slot0 + sp) * (26 * 256) + tid;
        static_for<0, W96_QB * 8>([&](auto J) { mine[J.value * 128] = acc_read4<J.value * 4>(); });
        mine[48 * 128] = (f32x4){m[0], m[1], m[2], m[3]};
        mine[49 * 128] = (f32x4){m[4], m[5], 0.f, 0.f};
        mine[50 * 128] = (f32x4){lsum[0], lsum[1], lsum[2], lsum[3]};
        mine[51 * 128] = (f32x4){lsum[4], lsum[5], 0.f, 0.f};
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *ticket_s = __hip_atomic_fetch_add(p.tickets + tix, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*ticket_s != nsp - 1) return;
        if (tid == 0) {
            __hip_atomic_store(p.tickets + tix, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // ---- epilogue, one query block at a time (32 accumulator registers pass through VGPRs): merge the slices of a
    //      key-split item in slice order, then O = O^T / l
    static_for<0, W96_QB>([&](auto Q) {
        constexpr int qb = Q.value;
        f32x4 ob[8];
        float mq = m[qb], l = lsum[qb];
        if (nsp > 1) {
            mq = -INFINITY, l = 0.f;
#pragma unroll
            for (int db = 0; db < 8; ++db) ob[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int s2 = 0; s2 < nsp; ++s2) {
                const f32x4 *oth = (const f32x4 *)p.ws + (int64_t)(slot0 + s2) * (26 * 256) + tid;
                const float ms = oth[(48 + qb / 4) * 128][qb % 4], ls = oth[(50 + qb / 4) * 128][qb % 4];
                if (ls == 0.f) continue;   // an empty slice (or a dead query column) carries no reference point
                const float m_new = fmaxf(mq, ms);
                const float a = __builtin_amdgcn_exp2f((mq - m_new) * SCALE_LOG2E), c = __builtin_amdgcn_exp2f((ms - m_new) * SCALE_LOG2E);
                mq = m_new;
                l = l * a + ls * c;
#pragma unroll
                for (int db = 0; db < 8; ++db) ob[db] = ob[db] * a + oth[(qb * 8 + db) * 128] * c;
            }
        } else {
            static_for<0, 8>([&](auto DBc) { ob[decltype(DBc)::value] = acc_read4<(qb * 8 + decltype(DBc)::value) * 4>(); });
        }
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        const int qrow = row0 + qb * 16 + li;
        if (qrow < p.Nq) {
            const int64_t ooff = b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + lg * 4;
            uint16_t *op = p.o + ooff;
            const uint16_t *oin = INPLACE ? p.o_in + ooff : nullptr;
            if (INPLACE && ntiles == 0) {
                if (oin != op) {
#pragma unroll
                    for (int db = 0; db < 8; ++db) *(u32x2 *)(op + db * 16) = *(const u32x2 *)(oin + db * 16);
                }
            } else {
#pragma unroll
                for (int db = 0; db < 8; ++db) {
                    float x0 = ob[db][0] * inv, x1 = ob[db][1] * inv, x2 = ob[db][2] * inv, x3 = ob[db][3] * inv;
                    u32x2 out;
                    if constexpr (INPLACE) {
                        const u32x2 old = *(const u32x2 *)(oin + db * 16);
                        const float a0 = round_bf16(x0 * p.o_scale), a1 = round_bf16(x1 * p.o_scale);
                        const float a2 = round_bf16(x2 * p.o_scale), a3 = round_bf16(x3 * p.o_scale);
                        out[0] = pack_bf16x2(__uint_as_float(old[0] << 16) + a0, __uint_as_float(old[0] & 0xffff0000u) + a1);
                        out[1] = pack_bf16x2(__uint_as_float(old[1] << 16) + a2, __uint_as_float(old[1] & 0xffff0000u) + a3);
                    } else {
                        out[0] = pack_bf16x2(x0, x1);
                        out[1] = pack_bf16x2(x2, x3);
                    }
                    *(u32x2 *)(op + db * 16) = out;
                }
                if constexpr (WRITE_L) {
                    if (lg == 0) p.l_out[(int64_t)bh * p.Nq + qrow] = 1.0f / (__builtin_amdgcn_exp2f(mq * SCALE_LOG2E) * l);
                }
            }
        }
    });
}

// Work plan for ragged key counts.  HunyuanVideo's text / tail query groups keep ALL 119k keys (13x a normal group); a
// head-parallel rank launches only 3 heads (1 863 items on 512 slots), so one such item -- 6 ms on one workgroup -- would
// be the whole launch.  One 1024-thread workgroup (a few microseconds) builds the plan on the device (the host never
// reads the counts):
//   * L = sum(counts) / slots is a slot's share at perfect balance; items above 1.5 T, T = max(L/4, 4096 keys), are cut
//     into ceil(count / T) slices (<= 64) over their key tiles -- each slice is a workgroup, partial (o, m, l) go through
//     scratch and the last arriver merges (same hand-off as the dense key-split tail);
//   * sliced items go FIRST (longest-first), everything else keeps the natural (head, group) order, which keeps one
//     head's K/V hot in L2 / Infinity Cache;
//   * T doubles until the slices fit `max_slices` (the scratch the host reserved); unused plan entries are -1.
__global__ __launch_bounds__(1024) void attn_plan_kernel(const int32_t *counts, int32_t *plan, int n, int Nk, int slots,
                                                        int max_slices, int cap) {
    __shared__ unsigned long long total;
    __shared__ int n_slices, wave_tot[16], base_s, T_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    auto cnt = [&](int i) {
        const int c = counts[i];
        return c < 0 ? 0 : (c < Nk ? c : Nk);
    };
    if (tid == 0) total = 0, n_slices = 0, base_s = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (int i = tid; i < n; i += 1024) mine += (unsigned)cnt(i);
    atomicAdd(&total, mine);
    __syncthreads();
    long long T = (long long)(total / (unsigned long long)slots) / 4;
    T = T < 4096 ? 4096 : T;
    T = (T + KVT - 1) / KVT * KVT;
    auto slices_of = [&](int c, long long t) {
        if (2 * (long long)c <= 3 * t) return 1;
        const long long k = (c + t - 1) / t;
        return (int)(k > 64 ? 64 : k);
    };
    for (int round = 0; round < 24; ++round) {  // T doubles until the sliced items fit the reserved scratch
        int need = 0;
        for (int i = tid; i < n; i += 1024) {
            const int k = slices_of(cnt(i), T);
            need += k > 1 ? k : 0;
        }
        if (tid == 0) T_s = 0;
        __syncthreads();
        atomicAdd(&T_s, need);
        __syncthreads();
        const int tot = T_s;
        __syncthreads();
        if (tot <= max_slices) break;
        T *= 2;
    }
    for (int i = tid; i < n; i += 1024) {
        const int k = slices_of(cnt(i), T);
        if (k > 1) {
            const int base = atomicAdd(&n_slices, k);
            for (int s2 = 0; s2 < k; ++s2) plan[2 * (base + s2)] = i, plan[2 * (base + s2) + 1] = s2 | (k << 16);
        }
    }
    __syncthreads();
    const int nh = n_slices;
    for (int i0 = 0; i0 < n; i0 += 1024) {  // ordered compaction of the unsliced items
        const int i = i0 + tid;
        const bool light = i < n && slices_of(cnt(i), T) == 1;
        const unsigned long long bal = __ballot(light);
        if (lane == 0) wave_tot[w] = __popcll(bal);
        __syncthreads();
        int off = base_s, tot = 0;
        for (int j = 0; j < 16; ++j) {
            off += j < w ? wave_tot[j] : 0;
            tot += wave_tot[j];
        }
        if (light) {
            const int e = nh + off + __popcll(bal & ((1ull << lane) - 1ull));
            plan[2 * e] = i, plan[2 * e + 1] = 1 << 16;
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    for (int e = nh + base_s + tid; e < cap; e += 1024) plan[2 * e] = -1, plan[2 * e + 1] = 1 << 16;
}

template <bool GATHER, bool INPLACE, bool WRITE_L, bool CSONLY = false>
int launch_attn(const AttnParams &p, hipStream_t stream) {
    auto kern = attn_kernel<GATHER, INPLACE, WRITE_L, CSONLY>;
    const int LDS = chipmunk_get_option("attn_pp") == 1 ? 96 * 1024 : ATTN_LDS_BYTES - (CSONLY ? NSTV * TILE_BYTES : 0);
    static int attr_set = 0;
    if (attr_set != LDS) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = LDS;
    }
    const int64_t nblocks = (int64_t)p.B * p.H * p.G;
    if (nblocks == 0) return CHIPMUNK_OK;
    CM_CHECK((int64_t)p.Nk * p.ks[2] * 2 < (1ll << 32) && (int64_t)p.Nk * p.vs[2] * 2 < (1ll << 32),
             "attention: one head's K or V spans more than 4 GiB (32-bit DMA offsets)");
    AttnParams pp = p;
    pp.probe = chipmunk_get_option("attn_variant");
    // experiment knob: attn_pp = 1 requests enough LDS that only ONE workgroup fits a CU (one wave per SIMD)
    const int wg_per_cu = chipmunk_get_option("attn_pp") == 1 ? 1 : 2;
    // block -> XCD mapping: 1 = every XCD walks its own contiguous (head, group) range, 0 = all XCDs sweep one head
    // together; option value 2 = chunks for the gathered launches only
    const int xo = chipmunk_get_option("attn_xcd_chunks");
    pp.xcd_chunks = xo == 2 ? (GATHER ? 1 : 0) : xo;
    // scratch layout: [tickets: TICKET_BYTES, always left at zero][work order | split partials]
    constexpr size_t TICKET_BYTES = 64 << 10;
    int64_t grid = nblocks;
    if (GATHER && !CSONLY && (nblocks >= 2048 || p.Nk >= 32768) && !chipmunk_get_option("attn_no_order")) {
        const int slots = wg_per_cu * device_cu_count();
        const int max_slices = 3 * 2 * device_cu_count();         // 1536 x 104 KB = 160 MB of partials at most
        const int64_t cap = nblocks + max_slices;                 // plan entries == workgroups launched
        const size_t plan_bytes = (size_t)cap * 2 * sizeof(int32_t);
        const size_t ws_off = (TICKET_BYTES + plan_bytes + 255) & ~(size_t)255;
        unsigned char *sc = (unsigned char *)chipmunk_scratch(stream, ws_off + (size_t)max_slices * 26 * 256 * sizeof(f32x4));
        if (sc && (size_t)max_slices * sizeof(int32_t) <= TICKET_BYTES) {
            int32_t *plan = (int32_t *)(sc + TICKET_BYTES);
            hipLaunchKernelGGL(attn_plan_kernel, dim3(1), dim3(1024), 0, stream, p.counts, plan, (int)nblocks, p.Nk, slots,
                               max_slices, (int)cap);
            pp.plan = plan;
            pp.tickets = (int32_t *)sc;
            pp.ws = (float *)(sc + ws_off);
            pp.xcd_chunks = 0;
            grid = cap;
        }
    }
    // Key-split tail.  Workgroups are dispatched in block order as the 2-per-CU slots free up; with near-equal items
    // the last (nblocks mod slots) of them run alone for a full item time (FLUX: 24 heads x 23 groups = 552 items on
    // 512 slots -> half of the launch is a 8 %-full second round).  Those items (or all of them when the whole grid
    // is under half the machine) are split over their key tiles into up to 8 workgroups each.
    // Dense launches only by default: a gathered FLUX item is ~45 us of which ~11 us are fixed costs every slice pays
    // again, and the split measured 94 -> 105 us there (option attn_split_gather forces it, for the tests).
    if (!CSONLY && !pp.plan && !pp.xcd_chunks && !chipmunk_get_option("attn_no_split") &&
        (!GATHER || chipmunk_get_option("attn_split_gather"))) {
        const int64_t slots = wg_per_cu * (int64_t)device_cu_count();
        const int64_t rem = nblocks <= slots / 2 ? nblocks : nblocks % slots;
        if (rem > 0 && rem * 2 <= slots && rem * sizeof(int32_t) <= TICKET_BYTES) {
            int f = (int)(slots / rem);
            f = f > 8 ? 8 : f;
            const size_t ws_bytes = (size_t)rem * f * 26 * 256 * sizeof(f32x4);
            unsigned char *sc = f > 1 ? (unsigned char *)chipmunk_scratch(stream, TICKET_BYTES + ws_bytes) : nullptr;
            if (sc) {
                pp.tickets = (int32_t *)sc;
                pp.ws = (float *)(sc + TICKET_BYTES);
                pp.nsplit = f;
                pp.split_full = (int)(nblocks - rem);
                grid = (nblocks - rem) + rem * f;
            }
        }
    }
    if constexpr (!CSONLY) {
        if (chipmunk_get_option("attn_w96") == 1) {
            auto k96 = attn_w96_kernel<GATHER, INPLACE, WRITE_L>;
            static bool w96_attr_set = false;
            if (!w96_attr_set) {
                (void)hipFuncSetAttribute((const void *)k96, hipFuncAttributeMaxDynamicSharedMemorySize, W96_LDS_BYTES);
                w96_attr_set = true;
            }
            hipLaunchKernelGGL(k96, dim3((unsigned)grid), dim3(128), W96_LDS_BYTES, stream, pp);
            CM_LAUNCH_CHECK();
            return CHIPMUNK_OK;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, pp);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

int check_common(const void *q, const void *k, const void *v, const void *o, int B, int H, int Nq, int Nk) {
    CM_CHECK(q && k && v && o, "attention: null tensor pointer");
    CM_CHECK(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: B,H,Nq,Nk must be positive (got %d,%d,%d,%d)", B, H, Nq, Nk);
    CM_CHECK((int64_t)B * H * ((Nq + QG - 1) / QG) < (1ll << 31), "attention: too many query groups");
    CM_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0,
             "attention: q/k/v must be 16-byte aligned and o 8-byte aligned");
    return CHIPMUNK_OK;
}
int check_strides(const int64_t *s, const char *name) {
    CM_CHECK(s != nullptr, "attention: %s strides missing", name);
    CM_CHECK(s[2] >= HD && s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0,
             "attention: %s strides must be multiples of 8 elements with a contiguous head dim of 128", name);
    return CHIPMUNK_OK;
}
void contiguous_strides(int64_t *s, int H, int N) {
    s[0] = (int64_t)H * N * HD;
    s[1] = (int64_t)N * HD;
    s[2] = HD;
}

}  // namespace

extern "C" int chipmunk_csp_attn(const void *q, const void *k, const void *v, void *o, const int64_t q_strides[3],
                                 const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
                                 const int32_t *indices, const int32_t *counts, int B, int H, int Nq, int Nk,
                                 int idx_stride, int o_scale, void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");  // csp_attn.cu:327
    CM_CHECK(indices && counts, "csp_attn: indices / counts missing");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    if (int e = check_strides(o_strides, "o")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i], p.os[i] = o_strides[i];
    p.indices = indices, p.counts = counts;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG, p.idx_stride = idx_stride;
    p.o_scale = (float)o_scale;
    p.o_in = p.o;
    return launch_attn<true, true, false>(p, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_attn_out(const void *q, const void *k, const void *v, const void *o_in, void *o_out,
                                     const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                                     const int64_t o_strides[3], const int32_t *indices, const int32_t *counts, int B,
                                     int H, int Nq, int Nk, int idx_stride, int o_scale, void *stream) {
    if (int e = check_common(q, k, v, o_out, B, H, Nq, Nk)) return e;
    CM_CHECK(o_in != nullptr && ((uintptr_t)o_in & 7) == 0, "csp_attn_out: o_in missing or not 8-byte aligned");
    CM_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");
    CM_CHECK(indices && counts, "csp_attn_out: indices / counts missing");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    if (int e = check_strides(o_strides, "o")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o_out;
    p.o_in = (const uint16_t *)o_in;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i], p.os[i] = o_strides[i];
    p.indices = indices, p.counts = counts;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG, p.idx_stride = idx_stride;
    p.o_scale = (float)o_scale;
    return launch_attn<true, true, false>(p, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_128_attn(const void *q, const void *k, const void *v, void *o, const int32_t *indices,
                                     const int32_t *counts, int B, int H, int Nq, int Nk, int idx_stride,
                                     void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(indices && counts, "csp_128_attn: indices / counts missing");
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    contiguous_strides(p.qs, H, Nq), contiguous_strides(p.os, H, Nq);
    contiguous_strides(p.ks, H, Nk), contiguous_strides(p.vs, H, Nk);
    p.indices = indices, p.counts = counts;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG, p.idx_stride = idx_stride;
    p.o_scale = 1.f;
    return launch_attn<true, false, false>(p, (hipStream_t)stream);
}

extern "C" int chipmunk_dense_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                   const int64_t k_strides[3], const int64_t v_strides[3], void *o, float *l, int B,
                                   int H, int Nq, int Nk, void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(l != nullptr, "dense_attn: l output missing");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i];
    contiguous_strides(p.os, H, Nq);
    p.l_out = l;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG;
    p.o_scale = 1.f;
    return launch_attn<false, false, true>(p, (hipStream_t)stream);
}

extern "C" int chipmunk_dense_colsum_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                          const int64_t k_strides[3], const int64_t v_strides[3], const float *pin,
                                          void *o, void *cs, float *l, int B, int H, int Nq, int Nk, int cs_stride,
                                          void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(l && cs && pin, "dense_colsum_attn: p / cs / l missing");
    CM_CHECK(cs_stride >= Nk, "dense_colsum_attn: cs row stride %d < Nk %d", cs_stride, Nk);
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i];
    contiguous_strides(p.os, H, Nq);
    p.l_out = l, p.p_in = pin, p.cs = (uint16_t *)cs, p.cs_stride = cs_stride;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG;
    p.o_scale = 1.f;
    // Two passes: (1) dense attention, (2) a K-only pass that recomputes the scores and reduces the column sums; they
    // share nothing but their inputs.  (A single fused pass was built first: it sat on the 256-VGPR cliff and measured
    // 885 vs 480 us at FLUX size, 36.9 vs 23.5 ms for two HunyuanVideo heads; removed.)
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_attn<false, false, true>(p, st);
    return rc != CHIPMUNK_OK ? rc : launch_attn<false, false, false, true>(p, st);
}
