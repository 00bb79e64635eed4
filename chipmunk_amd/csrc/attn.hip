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
#include "attn_params.h"
#include <type_traits>

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

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

#ifdef ATTN_PROF
// Cycle anatomy of the main loop (tools/attn_prof.py builds a separate library with -DATTN_PROF): per wave of
// workgroup 0, s_memtime at the segment boundaries of every tile, summed per segment.  Not part of the product build.
__device__ unsigned long long g_attn_prof[8 * 8];
#define PROF_DECL unsigned long long pt_, pacc_[7] = {0, 0, 0, 0, 0, 0, 0}; const bool prof_on_ = blockIdx.x == (gridDim.x / 2)
// absolute marks of one workgroup's life (entry, prologue issued, loop entered, loop left, epilogue done): slots 32 + wave * 8 + i
#define PROF_ABS(i) do { if (blockIdx.x == (gridDim.x / 2) && (threadIdx.x & 63) == 0) g_attn_prof[32 + (threadIdx.x >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
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
#define PROF_ABS(i)
#endif

#ifdef ATTN_TIMELINE
// Life marks of EVERY workgroup of a gathered launch on the constant 100 MHz clock (s_memrealtime: comparable across CUs and XCDs):
// tools/attn_timeline.py builds a separate library with -DATTN_TIMELINE.  Slot 0 entry, 1 share located (BAL), then per segment s
// (2 + 4 s ...): count / Q in registers, loop entered, loop left, segment done; 15 = exit.  Not part of the product build.
__device__ unsigned long long g_attn_tl[4096 * 16];
#define TL_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && (i) < 16) g_attn_tl[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int chipmunk_attn_timeline_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_tl), sizeof(g_attn_tl)) == hipSuccess ? 0 : 2;
}
extern "C" int chipmunk_attn_timeline_clear() {
    static unsigned long long z[4096 * 16];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_tl), z, sizeof(z)) == hipSuccess ? 0 : 2;
}
#else
#define TL_MARK(i)
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
//
// BAL = the work-balanced launch of short gathered launches (FLUX: 552 near-equal items on 512 resident slots used to be two
// rounds, the second 8 % full).  The launch's key tiles form one line, item after item; `gridDim.x` persistent workgroups take equal
// shares of it, cut at tile boundaries, so that an item is shared by at most two workgroups when no item is longer than a share.
// A workgroup walks its share from the right: first the HEAD of the item that straddles its right border -- the unnormalised
// (O, m, l) state after those tiles is PUBLISHED, nothing else --, then its whole items, last the TAIL of the item that straddles
// its left border, which does not start from zero but CONTINUES from the state its left neighbour published: online softmax is a
// fold over the key tiles, so the continuation is the uncut item's own arithmetic and the normal epilogue follows; there is no
// merge pass and no partial on the launch's critical path.  The neighbour's head is the first thing it does and the tail the last
// thing this workgroup does, so the state has been published about (share - item) tiles before it is asked for.
// Logical workgroup ids come from one atomic counter (HIP promises no dispatch order): a consumer only ever waits for a workgroup
// that drew its id earlier, i.e. one that is running or done.  The state travels as write-through (sc1) 16-byte stores -> vmcnt(0)
// -> barrier -> relaxed agent flag, and sc1 loads after one relaxed poll (MI355X_MICROARCH.md, hand-off price list).
// (ticket region of the library scratch, 16 384 ints: [0, 1536) the plan / split tickets, [4096, 8192) these, [8192, ..) knorm_max's floats)
constexpr int BAL_CTR = 4096;       // p.tickets[BAL_CTR] = id counter, [BAL_CTR + 1] = workgroups done; both left at zero
constexpr int BAL_FLAG = 4104;      // p.tickets[BAL_FLAG + L] = "the state for workgroup L is published" (reset by its consumer)
constexpr int BAL_MIN_TILES = 3;    // no part of a cut item is shorter than this (a part pays the item's fixed costs again)
constexpr int BAL_MAX_ITEMS = 4096;
constexpr int BAL_STATE_F4 = 26 * 256;   // float4 per published state: 24 of O, m, l per thread

//
// MIX = a gathered launch whose last `nblocks mod slots` items are ROW-SPLIT (round 6): item i >= p.split_full runs as three workgroups of
// 64 query rows -- four waves x ONE 16-row query block (QB = 1) instead of three -- so that the poorly filled last round is made of
// workgroups with a third of the softmax / MFMA work per key tile.  Rows are independent: no partial state, no merge, same bits as the
// unsplit item.  Both bodies live in the one kernel (the thirds must start as slots free up, not behind a launch boundary).
template <bool GATHER, bool INPLACE, bool WRITE_L, bool CSONLY = false, bool BAL = false, bool MIX = false>
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
    PROF_ABS(0);

    TL_MARK(0);
    // ---- BAL: draw a logical id, scan the items' tile counts, locate this workgroup's share [item jl tile cl, item jr tile cr)
    int bal_L = 0, bal_jl = 0, bal_cl = 0, bal_jr = 0, bal_cr = 0;
    if constexpr (BAL) {
        int *bs = (int *)cs_acc;   // [0] id, [1..4] jl cl jr cr, [8..11] wave totals
        if (tid == 0) bs[0] = __hip_atomic_fetch_add(p.tickets + BAL_CTR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int n = p.B * p.H * p.G, per = (n + 255) >> 8;
        // an item's length on the line: its key tiles, at least one (an item without keys still has an epilogue to run)
        auto ntp = [&](int j) {
            int c = p.counts[j];
            c = c < 0 ? 0 : (c < p.Nk ? c : p.Nk);
            const int t = (c + KVT - 1) / KVT;
            return t > 0 ? t : 1;
        };
        const int j0 = tid * per, j1 = j0 + per < n ? j0 + per : n;
        int loc = 0;
        for (int j = j0; j < j1; ++j) loc += ntp(j);
        int incl = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            incl += lane >= d ? t : 0;
        }
        if (lane == 63) bs[8 + w] = incl;
        __syncthreads();
        int wbase = 0, W = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = bs[8 + i];
            wbase += i < w ? t : 0;
            W += t;
        }
        bal_L = bs[0];
        const int nwg = (int)gridDim.x;
        const int a[2] = {(int)((int64_t)W * bal_L / nwg), (int)((int64_t)W * (bal_L + 1) / nwg)};
        int pos = wbase + incl - loc;
        for (int j = j0; j < j1; ++j) {
            const int t = ntp(j);
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (pos <= a[e] && a[e] < pos + t) {
                    int item = j, off = a[e] - pos;
                    if (off < BAL_MIN_TILES) off = 0;                      // a cut this close to an item's border moves onto it
                    else if (t - off < BAL_MIN_TILES) item = j + 1, off = 0;
                    bs[1 + 2 * e] = item, bs[2 + 2 * e] = off;
                }
            pos += t;
        }
        if (tid == 0) {
            if (a[0] >= W) bs[1] = n, bs[2] = 0;
            if (a[1] >= W) bs[3] = n, bs[4] = 0;
        }
        __syncthreads();
        bal_jl = __builtin_amdgcn_readfirstlane(bs[1]), bal_cl = __builtin_amdgcn_readfirstlane(bs[2]);
        bal_jr = __builtin_amdgcn_readfirstlane(bs[3]), bal_cr = __builtin_amdgcn_readfirstlane(bs[4]);
        bal_L = __builtin_amdgcn_readfirstlane(bal_L);
        __syncthreads();   // (the scratch words are the split path's ticket word and the column-sum partials elsewhere)
    }
    // segments, right to left: item jr's tiles [0, cr) if cr > 0, the whole items between, item jl from tile cl
    TL_MARK(1);
    int tl_seg = 0;
    int bal_item = BAL ? (bal_cr > 0 ? bal_jr : bal_jr - 1) : 0;
    if (BAL && bal_item < bal_jl) bal_item = -1;   // an empty share
    while (!BAL || bal_item >= 0) {
    // BAL: everything derived from the thread id is recomputed per segment from an opaque copy -- left alone, the loop-invariant
    // code motion keeps the epilogue's and the prologue's lane constants alive through the main loop (123 spilled VGPRs)
    int tid_o = threadIdx.x;
    if constexpr (BAL) asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // one item (or BAL segment) with QB 16-row query blocks per wave: 3 = a whole 192-row group, 1 = a 64-row third of it (MIX)
    auto item_body = [&](auto qbc, const int third) __attribute__((always_inline)) {
    constexpr int QB = decltype(qbc)::value;
    constexpr int QWB = 16 * QB;     // query rows per wave
    static_assert(QB == 3 || (QB == 1 && GATHER && !CSONLY && !BAL && !WRITE_L), "thirds exist for the gathered forms only");
    const bool bal_publish = BAL && bal_item == bal_jr;                    // (only reached with cr > 0)
    const bool bal_consume = BAL && bal_item == bal_jl && bal_cl > 0;

    int wid0 = p.xcd_chunks ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    if constexpr (BAL) wid0 = bal_item;
    // MIX: the whole items (a multiple of the slot count, hence of 8) are walked XCD by XCD -- every XCD's L2 then holds the K / V of three
    // heads instead of passing all 24 through (probe 16 restores block order; -1.5 % isolated, -1.4 % in the bench for the unsplit launch)
    if constexpr (MIX) wid0 = third >= 0 ? p.split_full + ((int)blockIdx.x - p.split_full) / 3
                              : (p.probe & 16) ? (int)blockIdx.x : xcd_remap(blockIdx.x, p.split_full);
    // sp / nsp: this workgroup's slice of the item's key tiles; slot0: scratch slot of the item's slice 0; tix: its ticket
    int sp = 0, nsp = 1, slot0 = 0, tix = 0, wid = wid0;
    if (BAL || MIX) {
    } else if (!CSONLY && p.plan) {
        const u32x2 entry = *(const u32x2 *)(p.plan + 2 * wid0);   // (one round trip, not two)
        wid = (int)entry[0];
        if (wid < 0) return;
        const int meta = (int)entry[1];
        sp = meta & 0xff, nsp = (meta >> 8) & 0xff;
        slot0 = tix = meta >> 16;
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

    // The item's key count is REQUESTED here and first used below the Q loads: count, the first key tiles' indices and the Q rows make
    // one memory round trip together instead of three in a row (tools/attn_prof.py: 7 k + 9 k cycles of a FLUX item's 114 k passed
    // before its first tile).
    int count_v = GATHER ? p.counts[(int64_t)bh * p.G + g] : p.Nk;
    const IndexRow irow = GATHER ? index_row(p, (int64_t)bh * p.G + g) : IndexRow{nullptr, 0};
    const int32_t *idx = irow.ptr;
    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t *vbase = p.v + b * p.vs[0] + h * p.vs[1];
    const int row0 = g * QG + (QB == 3 ? w * QW : third * 64 + w * QWB);
    // wave 0 streams 64 indices starting at tile T into key slot T % KRING (only the first 32 are tile T's)
    auto issue_keys = [&](int T) {
        if constexpr (GATHER) {
            if (w == 0) {
                int pos = T * KVT + lane;
                pos = pos < irow.width ? pos : irow.width - 1;
                __builtin_amdgcn_global_load_lds(GLB_PTR(idx + pos), LDS_PTR(key_ring + (T & (KRING - 1)) * 64), 4, 0, 0);
            }
        }
    };
    // an unsliced item starts at key tile 0 whatever its count: the first key tiles' indices are requested before anything else, so
    // that their round trip runs beside the Q rows' (it used to start after the Q loads had been issued: one more serial round trip)
    const int bal_tb = bal_consume ? bal_cl : 0;   // BAL: the segment's first key tile
    const bool early_keys = GATHER && !CSONLY && sp == 0 && irow.width > 0;
    if (early_keys) {
#pragma unroll
        for (int T = 0; T < NST; ++T) issue_keys(bal_tb + T);
    }
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase), vrsrc = make_rsrc(vbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u, vstride_b = (uint32_t)p.vs[2] * 2u;  // row strides in bytes

    // ---- Q^T fragments (B operand): lane = query column li, k = lg*8..lg*8+7 of each 32-wide d step
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
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

    if constexpr (GATHER) {
        // vmcnt(0), not a counted wait: index DMAs went out behind this register load, and an LDS-DMA may retire before an older
        // register load; the operand pins every use of the count below this point
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(count_v)::"memory");
        count_v = __builtin_amdgcn_readfirstlane(count_v);
    }
    const int count = count_v;
    TL_MARK(2 + 4 * tl_seg);
    // packed positions >= Nk are masked out by the reference (right_fill, csp_128_attn.cu:314)
    const int valid = count < p.Nk ? count : p.Nk;
    const int ntiles = (valid + KVT - 1) / KVT;
    if (!CSONLY && nsp > 1 && !p.plan) {  // every workgroup of the item derives the same effective split: at least 4 key tiles per slice
        const int cap = ntiles / 4 > 1 ? ntiles / 4 : 1;
        nsp = nsp < cap ? nsp : cap;
        if (sp >= nsp) return;
    }
    int tbeg = (int)((int64_t)ntiles * sp / nsp), tend = (int)((int64_t)ntiles * (sp + 1) / nsp);
    if constexpr (BAL) {
        tbeg = bal_tb < ntiles ? bal_tb : ntiles;
        tend = bal_publish && bal_cr < ntiles ? bal_cr : ntiles;
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

    // every wave stages rows (2w+i)*4 + lg, i = 0..1, of the K tile and of the V tile: 4 DMA instructions per wave
    static_assert((NST & (NST - 1)) == 0 && (KRING & (KRING - 1)) == 0, "ring depths addressed by masks");
    // the gather key of row (2w+i)*4 + lg of tile T, clamped into the key range (memory safety for malformed indices)
    auto tile_key = [&](int T, int i) {
        const int r = (w * 2 + i) * 4 + lg;
        const int pos = T * KVT + r;
        int key = 0;
        if (pos < valid) {
            key = GATHER ? key_ring[(T & (KRING - 1)) * 64 + r] : pos;
            key = key < 0 ? 0 : (key >= p.Nk ? p.Nk - 1 : key);
        }
        return key;
    };
    // buffer-form DMA of one K / V row pair per lane group: per-(b, h) SGPR resource + 32-bit lane offset (row * stride +
    // swizzled 16-byte chunk); `vslot` = T mod NSTV, kept by the caller (5 is not a power of two)
    auto issue_rows = [&](int T, int vslot, int i, int key) {
        const int r = (w * 2 + i) * 4 + lg;
        const uint32_t koff = ((uint32_t)key * kstride_b) + ((uint32_t)(li ^ (r & 15)) << 4);
        const uint32_t voff = ((uint32_t)key * vstride_b) + ((uint32_t)(li ^ ((r & 7) << 1)) << 4);
        blds16(krsrc, koff, 0, Kl + (T & (NST - 1)) * TILE_BYTES + (w * 2 + i) * 1024);
        if constexpr (!CSONLY) blds16(vrsrc, voff, 0, Vl + vslot * TILE_BYTES + (w * 2 + i) * 1024);
    };
    auto issue_data = [&](int T) {   // prologue / column-sum pass form: keys read from the LDS key ring on the spot
        const int vslot = (int)((unsigned)T % (unsigned)NSTV);
#pragma unroll
        for (int i = 0; i < 2; ++i) issue_rows(T, vslot, i, tile_key(T, i));
    };


    f32x4 o[QB][8];
    float m[QB], lsum[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY, lsum[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 8; ++db) o[qb][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const __amdgpu_buffer_rsrc_t bal_rsrc = make_rsrc(BAL ? (const void *)p.ws : (const void *)p.q);
    constexpr uint32_t BAL_STATE_BYTES = BAL_STATE_F4 * 16;

    // ---- prologue: keys of tiles 0..NST-1 synchronously, then the data of tiles 0..NST-2, each group followed by one more
    //      key DMA (the main loop reads a tile's keys one iteration before it issues the tile's data: keys run 7 ahead)
    PROF_ABS(1);
    if (tend > tbeg) {
        if (!early_keys) {
#pragma unroll
            for (int T = 0; T < NST; ++T) issue_keys(tbeg + T);
        }
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll
        for (int T = 0; T < (CSONLY ? 2 : NST - 1); ++T) {
            if (tbeg + T < tend) {
                issue_data(tbeg + T);
                issue_keys(tbeg + T + NST);
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
            const unsigned char *Kb = Kl + (t & (NST - 1)) * TILE_BYTES;
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
            // vmcnt: the tiles' DMA; lgkmcnt: this wave's ds_write of its partial sums into cs_acc -- the raw s_barrier does not
            // wait for LDS traffic, and without it wave 0's flush could read a wave's partial of FOUR TILES AGO (same
            // cs_acc slot) for the tile the wave finished last: seen at HunyuanVideo size as ~1 event per 10^6 wave-tiles, 32
            // adjacent sums of an odd tile ~6 % off, never the same place twice (found by the one-pass route's cross-check)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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

    if constexpr (BAL) {
        if (bal_consume) {
            // the left neighbour's state for this item: one lane polls (relaxed, sleeping), then every thread reads its 26 pieces
            // with sc1 loads (the producer stored sc1: no acquire fence needed) and waits them out with vmcnt(0) -- the tile DMAs
            // issued above are waited for with them; register loads under a counted wait must not have DMAs behind them
            if (tid == 0 && !(p.probe & 32)) {
                int32_t *flag = p.tickets + BAL_FLAG + bal_L;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
                __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            }
            __syncthreads();
            const uint32_t sb = (uint32_t)bal_L * BAL_STATE_BYTES + (uint32_t)tid * 16u;
            if (!(p.probe & 16))   // (timing probes: 16 = no state traffic, 32 = no flag wait either; results are then wrong)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int db = 0; db < 8; ++db)
                    o[qb][db] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bal_rsrc, sb + (qb * 8 + db) * 4096, 0, 16));
            const f32x4 ms = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bal_rsrc, sb + 24 * 4096, 0, 16));
            const f32x4 ls = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bal_rsrc, sb + 25 * 4096, 0, 16));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) m[qb] = ms[qb], lsum[qb] = ls[qb];
        }
    }
    // accumulate forms of an unsplit item with work to do: the epilogue's base rows are fetched ahead of the drain tile
    // (BAL: read in the epilogue's store loop -- held in registers from here they spill in this form of the kernel, and a spill waits
    // the loads out on the spot)
    const bool early_base = INPLACE && !BAL && nsp == 1 && tend > tbeg;
    const bool base_before_drain = early_base;
    u32x4 base[INPLACE ? QWB * 256 / 1024 : 1];
    auto fetch_base = [&]() {
#pragma unroll
        for (int i = 0; i < QWB * 256 / 1024; ++i) {
            const int qrow = row0 + i * 4 + (lane >> 4);
            base[i] = (u32x4){0u, 0u, 0u, 0u};
            if (qrow < p.Nq) base[i] = *(const u32x4 *)(p.o_in + b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + (lane & 15) * 8);
        }
    };
    PROF_DECL;
    PROF_ABS(2);
    TL_MARK(3 + 4 * tl_seg);
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
        bf16x8 pq[QB];   // P^T of the tile whose PV is pending
        f32x4 s[QB][2];
        float alpha[QB], nmsc[QB];            // nmsc = -m*c, kept beside m (recomputed only when the reference point moves)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) pq[qb] = (bf16x8){}, alpha[qb] = 1.f, nmsc[qb] = 0.f;
        // keys of the tile whose data goes out NEXT, read from the key ring one iteration ahead (a read at the point of
        // use costs two exposed LDS round trips per tile: ds_read -> lgkmcnt(0) -> address -> DMA, twice)
        int knext[2] = {0, 0};
        int v5 = (int)((unsigned)tbeg % (unsigned)NSTV);   // tbeg mod 5, advanced with t
        if (tend > tbeg) {
            knext[0] = tile_key(tbeg + NST - 1, 0);
            knext[1] = tile_key(tbeg + NST - 1, 1);
        }
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
                const int vs = v5 + NST - 1 >= NSTV ? v5 + NST - 1 - NSTV : v5 + NST - 1;   // (t + 3) mod 5
                issue_rows(t + NST - 1, vs, 0, knext[0]);
                issue_rows(t + NST - 1, vs, 1, knext[1]);
                issue_keys(t + 2 * (NST - 1) + 1);
                knext[0] = tile_key(t + NST, 0);   // landed: issued 3 iterations ago, this iteration's wait covers it
                knext[1] = tile_key(t + NST, 1);
            }
            PROF_MARK(1);
        };
        auto qk_tile = [&](int t) {     // S^T(t) = K(t) . Q^T, dead keys of a ragged last tile masked
            const unsigned char *Kb = Kl + (t & (NST - 1)) * TILE_BYTES;
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
                for (int qb = 0; qb < QB; ++qb)
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
                        for (int qb = 0; qb < QB; ++qb) s[qb][kt][r] = dead ? -INFINITY : s[qb][kt][r];
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
        // the steady loop's forms: all three query blocks in ONE statement each (hipcc pads every asm statement that writes a
        // VGPR with an s_nop and copies operands around single-instruction helpers: 24 statements -> 2 per tile)
        auto tile_max_x3 = [&](float (&mx)[QB]) {
            if constexpr (QB != 3) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) mx[qb] = tile_max(qb);
            } else
            asm volatile(
                "v_max3_f32 %0, %3, %4, %5\n\tv_max3_f32 %1, %11, %12, %13\n\tv_max3_f32 %2, %19, %20, %21\n\t"
                "v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %14, %15\n\tv_max3_f32 %2, %2, %22, %23\n\t"
                "v_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %1, %1, %16, %17\n\tv_max3_f32 %2, %2, %24, %25\n\t"
                "v_max_f32 %0, %0, %10\n\tv_max_f32 %1, %1, %18\n\tv_max_f32 %2, %2, %26"
                : "=&v"(mx[0]), "=&v"(mx[1]), "=&v"(mx[2])
                : "v"(s[0][0][0]), "v"(s[0][0][1]), "v"(s[0][0][2]), "v"(s[0][0][3]), "v"(s[0][1][0]), "v"(s[0][1][1]), "v"(s[0][1][2]), "v"(s[0][1][3]),
                  "v"(s[1][0][0]), "v"(s[1][0][1]), "v"(s[1][0][2]), "v"(s[1][0][3]), "v"(s[1][1][0]), "v"(s[1][1][1]), "v"(s[1][1][2]), "v"(s[1][1][3]),
                  "v"(s[2][0][0]), "v"(s[2][0][1]), "v"(s[2][0][2]), "v"(s[2][0][3]), "v"(s[2][1][0]), "v"(s[2][1][1]), "v"(s[2][1][2]), "v"(s[2][1][3]));
        };
        auto max_rows_x3 = [&](float (&mx)[QB]) {   // max over the four 16-lane rows of the wave, three values at once
            float t0, t1, t2;
            if constexpr (QB != 3) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) mx[qb] = max_rows(mx[qb]);
            } else
            asm volatile(
                "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\ts_nop 1\n\t"
                "v_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5\n\ts_nop 0\n\t"
                "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5\n\t"
                "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\ts_nop 1\n\t"
                "v_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_permlane16_swap_b32 %2, %5\n\ts_nop 0\n\t"
                "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5"
                : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2));
        };
        auto exp_block = [&](int qb, float nm) {    // p = exp2(s*c - m*c), in place
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], SCALE_LOG2E, nm));
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
            const unsigned char *Vb = Vl + (int)((unsigned)tv % (unsigned)NSTV) * TILE_BYTES;
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
                for (int qb = 0; qb < QB; ++qb) o[qb][db] = mfma16(vf, pq[qb], o[qb][db]);
            }
        };
        auto within_lag = [&](const float (&mx)[QB], float lag) {   // per lane: every query block's tile maximum within `lag` of its reference point
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) ok = ok && mx[qb] <= m[qb] + lag;
            return ok;
        };
        if (tend > tbeg) {
            // ---- first tile: the reference point is its own maximum (exact), nothing to rescale
            tile_sync(tbeg);
            qk_tile(tbeg);
            if (BAL && bal_consume) {
                // a continued item: the reference point is the published one, moved (with the rescale) only if this tile outgrows the lag
                float mx[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) mx[qb] = max_rows(tile_max(qb));
                constexpr float LAG_RAW0 = MAX_LAG / SCALE_LOG2E;
                if (!__all(within_lag(mx, LAG_RAW0))) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const float m_new = max2(m[qb], mx[qb]);
                        const float a = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
                        lsum[qb] *= a;
                        m[qb] = m_new;
#pragma unroll
                        for (int db = 0; db < 8; ++db) o[qb][db] *= a;
                    }
                }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (!(BAL && bal_consume)) m[qb] = max_rows(tile_max(qb));
                nmsc[qb] = -m[qb] * SCALE_LOG2E;
                exp_block(qb, nmsc[qb]);
                row_sum(qb);
                pq[qb] = to_bf16(qb);
            }
            PROF_MARK(3);
            for (int t = tbeg + 1; t < tend; ++t) {
                v5 = v5 + 1 == NSTV ? 0 : v5 + 1;   // t mod 5
                tile_sync(t);
                qk_tile(t);
                bool moved = false;   // wave-uniform: the reference point moved in this tile
                // ---- PV of tile t-1 and the softmax of tile t, hand-interleaved: 8 chunks, each = 3 MFMAs (one 16-wide d
                // block) + the V^T fragment of the block two ahead + one slice of the softmax, fenced so that the slices
                // stay in the shadow of MFMAs that do not depend on them (left to itself hipcc runs the whole softmax first
                // and the MFMAs after it; sched_group_barrier requests did not move it).
                {
                    const unsigned char *Vb = Vl + (v5 == 0 ? NSTV - 1 : v5 - 1) * TILE_BYTES;   // (t - 1) mod 5
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
                    float mx[QB];
                    __builtin_amdgcn_sched_barrier(0);
                    // chunks 0, 1: maxima; then the (rare) reference update in its own block; chunks 2..7: exp2, row sums
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        vr[(db + 2) % 3] = load_v(db + 2);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][db] = mfma16(vr[db % 3], pq[qb], o[qb][db]);
                        if (db == 0) tile_max_x3(mx);
                        else max_rows_x3(mx);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;   // the lag in units of the raw scores
                    if (!__all(within_lag(mx, LAG_RAW))) {
                        moved = true;
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            const float m_new = max2(m[qb], mx[qb]);
                            alpha[qb] = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
                            lsum[qb] *= alpha[qb];   // (o is rescaled once the pending PV has been accumulated)
                            m[qb] = m_new;
                            nmsc[qb] = -m_new * SCALE_LOG2E;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int db = 2; db < 8; ++db) {
                        if (db + 2 < 8) vr[(db + 2) % 3] = load_v(db + 2);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][db] = mfma16(vr[db % 3], pq[qb], o[qb][db]);
                        if (db <= 4) {              // exp2 of one query block per chunk
                            const int qb = db - 2;
                            if (qb < QB) {
                                exp_block(qb, nmsc[qb]);
                                pin(s[qb][0]);
                                pin(s[qb][1]);
                            }
                        } else {                    // row sums
                            const int qb = db - 5;
                            if (qb < QB) {
                                row_sum(qb);
                                pin(lsum[qb]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // P^T of tile t as bf16: only once the last MFMA on tile t-1's P has been issued (a second set of P
                    // registers would push the kernel over 256 VGPRs)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) pq[qb] = to_bf16(qb);
                }
                PROF_MARK(3);
                // rescale AFTER the pending PV has been accumulated: O_t = alpha_t (O_{t-1} + P_{t-1} V_{t-1}) + P_t V_t
                if (moved) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
                        for (int db = 0; db < 8; ++db) o[qb][db] *= alpha[qb];
                    }
                }
                PROF_MARK(4);
            }
            // accumulate forms: the base rows in the epilogue's layout (12 x 16 bytes per lane), requested before the drain tile
            if constexpr (INPLACE) {
                if (base_before_drain) fetch_base();
            }
            pv_mfmas(tend - 1);
        }
    }
    PROF_END(w, tend - tbeg);
    PROF_ABS(3);
    TL_MARK(4 + 4 * tl_seg);

    if constexpr (QB == 3) {
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
    }
    bool published = false;
    if constexpr (QB == 3) {
    if (BAL && bal_publish) {
        published = true;
        // ---- the head of a cut item: the state goes to the right neighbour (slot L + 1), write-through; no epilogue
        const uint32_t sb = (uint32_t)(bal_L + 1) * BAL_STATE_BYTES + (uint32_t)tid * 16u;
        if (!(p.probe & 16))
#pragma unroll
        for (int qb = 0; qb < 3; ++qb)
#pragma unroll
            for (int db = 0; db < 8; ++db)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[qb][db]), bal_rsrc, sb + (qb * 8 + db) * 4096, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){m[0], m[1], m[2], 0.f}), bal_rsrc, sb + 24 * 4096, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){lsum[0], lsum[1], lsum[2], 0.f}), bal_rsrc, sb + 25 * 4096, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave's stores have been acknowledged (and everybody is done with the rings and the key ring)
        if (tid == 0 && !(p.probe & 32)) __hip_atomic_store(p.tickets + BAL_FLAG + bal_L + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    if (!published) {
    // ---- epilogue: O = O^T / l.  A lane holds 4 consecutive d (8 bytes) of one query row per (qb, db): stored as they stand that is
    //      24 eight-byte accesses per lane, each instruction touching 16 rows x 32 bytes (and as many loads for the accumulate forms):
    //      the store tail was 15.8 k of a FLUX item's 114 k cycles (tools/attn_prof.py).  The wave's 48 x 128 bf16 results go through
    //      its own 12 KiB of the (now idle) K/V rings instead -- 8-byte writes, chunk index XOR row so that the 16 lanes of a write land on
    //      16 different chunks -- and leave as 12 whole-row 16-byte accesses per lane (4 rows = 1 KiB per instruction); the accumulation
    //      base is fetched in that same layout BEFORE the drain tile when this workgroup is sure to run the epilogue.
    constexpr int EP_I = QWB * 256 / 1024;   // 16-byte pieces per lane: 12 (a third: 4)
    unsigned char *stage = smem + w * (QWB * 256);
    const int er0 = lane >> 4, ech = lane & 15;
    auto ep_row = [&](int i) { return i * 4 + er0; };   // wave-local row of piece i
    const int64_t obase = b * p.os[0] + h * p.os[1];
    __syncthreads();   // every wave is done with the rings
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l = sum_across_rows(lsum[qb]);
        const float inv = (l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f) * (INPLACE ? p.o_scale : 1.f);   // o_scale = +-1: exact
        const int r = qb * 16 + li;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            // bf16(o_scale * result): what the reference stores before its bf16 reduce-add (csp_attn.cu:294-300)
            const u32x2 out = {pack_bf16x2(o[qb][db][0] * inv, o[qb][db][1] * inv), pack_bf16x2(o[qb][db][2] * inv, o[qb][db][3] * inv)};
            *(u32x2 *)(stage + r * 256 + (((db * 2 + (lg >> 1)) ^ (r & 15)) << 4) + (lg & 1) * 8) = out;
        }
        if constexpr (WRITE_L) {
            // l = 1 / (exp2(m*c) * norm) = 1 / sum_j exp(s_ij / sqrt(D))   (dense_attn.cu:225-227)
            const int qrow = row0 + r;
            if (lg == 0 && qrow < p.Nq) p.l_out[(int64_t)bh * p.Nq + qrow] = 1.0f / (__builtin_amdgcn_exp2f(m[qb] * SCALE_LOG2E) * l);
        }
    }
#pragma unroll
    for (int i = 0; i < EP_I; ++i) {
        const int r = ep_row(i), qrow = row0 + r;
        if (qrow >= p.Nq) continue;
        const int64_t off = obase + (int64_t)qrow * p.os[2] + ech * 8;
        u32x4 v = *(const u32x4 *)(stage + r * 256 + ((ech ^ (r & 15)) << 4));
        if constexpr (INPLACE) {
            const u32x4 old = early_base ? base[i] : *(const u32x4 *)(p.o_in + off);
            if (ntiles == 0) {
                v = old;   // nothing to add: in place leaves o as it is, out of place copies the base
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack_bf16x2(__uint_as_float(old[e] << 16) + __uint_as_float(v[e] << 16),
                                       __uint_as_float(old[e] & 0xffff0000u) + __uint_as_float(v[e] & 0xffff0000u));
            }
        }
        *(u32x4 *)(p.o + off) = v;
    }
#ifdef ATTN_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the mark is taken once the stores have left
#endif
    PROF_ABS(4);
    }
    };   // item_body
    if constexpr (MIX) {
        // the last items of the launch as thirds: workgroup split_full + 3 k + j = rows 64 j .. 64 j + 63 of item split_full + k
        const int k3 = (int)blockIdx.x - p.split_full;
        if (k3 >= 0) item_body(std::integral_constant<int, 1>{}, k3 % 3);
        else item_body(std::integral_constant<int, 3>{}, -1);
    } else {
        item_body(std::integral_constant<int, 3>{}, -1);
    }
    TL_MARK(5 + 4 * tl_seg);
    ++tl_seg;
    if constexpr (!BAL) break;
    bal_item = bal_item > bal_jl ? bal_item - 1 : -1;
    }   // segments
    TL_MARK(15);
    if constexpr (BAL) {
        // the last workgroup out leaves the two counters at zero for the next launch
        if (tid == 0) {
            const int done = __hip_atomic_fetch_add(p.tickets + BAL_CTR + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == (int)gridDim.x - 1) {
                __hip_atomic_store(p.tickets + BAL_CTR, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.tickets + BAL_CTR + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
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
//   * T doubles until the slices fit `max_slices` (the scratch the host reserved); unused plan entries are -1;
//   * the tail: with B workgroups of about equal length on `slots` resident slots the last B mod slots of them run alone for a whole
//     item time (HunyuanVideo: 120 of 15 480 = 2.5 % of the launch; Wan2.1: 4 of 2 052 = a fifth of it).  The last R unsliced items
//     are therefore cut into k = slots / R slices each (>= 8 key tiles per slice, <= 64, within the scratch), placed at the END of
//     the plan: the final round is full and 1 / k as long.
// Plan entry: (item | -1, slice | slices << 8 | scratch slot of the item's slice 0 << 16).
__global__ __launch_bounds__(1024) void attn_plan_kernel(const int32_t *counts, int32_t *plan, int n, int Nk, int slots,
                                                        int max_slices, int cap, int tail) {
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
            for (int s2 = 0; s2 < k; ++s2) plan[2 * (base + s2)] = i, plan[2 * (base + s2) + 1] = s2 | (k << 8) | (base << 16);
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
            plan[2 * e] = i, plan[2 * e + 1] = 1 << 8;
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    // ---- tail: the last R unsliced items as k slices each (see above)
    const int n_light = base_s;
    const int R = slots > 0 ? (nh + n_light) % slots : 0;
    int used = nh + n_light;
    if (tail && R > 0 && R <= n_light && 2 * R <= slots && R <= 1024) {
        const int tail0 = nh + n_light - R;
        int item = -1, tiles = 1 << 30;
        if (tid < R) {
            item = plan[2 * (tail0 + tid)];
            tiles = (cnt(item) + KVT - 1) / KVT;
        }
        if (tid == 0) T_s = 1 << 30;
        __syncthreads();
        atomicMin(&T_s, tiles);
        __syncthreads();
        int k = slots / R;
        k = k > 64 ? 64 : k;
        k = k > T_s / 8 ? T_s / 8 : k;
        k = k > (max_slices - nh) / R ? (max_slices - nh) / R : k;
        if (k >= 2) {
            if (tid < R)
                for (int s2 = 0; s2 < k; ++s2)
                    plan[2 * (tail0 + tid * k + s2)] = item, plan[2 * (tail0 + tid * k + s2) + 1] = s2 | (k << 8) | ((nh + tid * k) << 16);
            used = tail0 + R * k;
        }
        __syncthreads();
    }
    for (int e = used + tid; e < cap; e += 1024) plan[2 * e] = -1, plan[2 * e + 1] = 1 << 8;
}

template <bool GATHER, bool INPLACE, bool WRITE_L, bool CSONLY = false>
int launch_attn(const AttnParams &p, hipStream_t stream) {
    auto kern = attn_kernel<GATHER, INPLACE, WRITE_L, CSONLY>;
    const int LDS = chipmunk_get_option("attn_pp") == 1 ? 96 * 1024 : ATTN_LDS_BYTES - (CSONLY ? NSTV * TILE_BYTES : 0);
    static uint64_t lds_set = 0;
    static int lds_bytes = 0;
    if (lds_bytes != LDS) lds_set = 0, lds_bytes = LDS;   // (the attn_pp experiment knob changes the request)
    ensure_dynamic_lds((const void *)kern, LDS, lds_set);
    const int64_t nblocks = (int64_t)p.B * p.H * p.G;
    if (nblocks == 0) return CHIPMUNK_OK;
    // option attn_csp64 = 1 sends gathered launches to the one-wave-per-SIMD kernel (attn64.hip: three compute waves + a
    // loader wave per 192-row group, ONE workgroup per CU).  Off by default: at equal counts it measured +2.7 % (random
    // keys) / +5.4 % (keys shared between groups) over this file's kernel, but on HunyuanVideo's ragged launches (text /
    // tail groups 13x longer than the rest, one workgroup per CU) 15.5 vs 14.3 ms.
    // Gathered launches over long key ranges go to the two-waves-x-96-rows kernel of attn96.hip (same plan, same scratch):
    // HunyuanVideo 24 heads 12.7 -> 11.1 ms, a head-parallel rank's 1 / 2 / 3 heads 0.71 / 1.12 / 1.81 -> 0.64 / 1.00 / 1.39 ms,
    // Wan2.1 (12 heads x 32 760 keys) 1.97 -> 1.67 ms; short items (FLUX: 21 tiles, 78 vs 130 us: its per-item prologue and
    // epilogue are longer) stay here.  Option attn_csp96: 1 = always, 2 = never.
    const int o96 = chipmunk_get_option("attn_csp96");
    const bool fits96 = GATHER && !CSONLY && !WRITE_L && p.Nk < (1 << 24) && p.ks[2] * 2 < (1 << 24) && p.vs[2] * 2 < (1 << 24) &&
                        (p.idx_stride & 3) == 0;   // (its index rows are read 16 bytes at a time)
    const bool want96 = fits96 && (o96 == 1 || (o96 == 0 && nblocks >= 128 && p.Nk >= 16384));
    const int o64 = chipmunk_get_option("attn_csp64");
    const bool want64 = GATHER && !CSONLY && !WRITE_L && o64 == 1 && p.Nk < (1 << 24) && p.ks[2] * 2 < (1 << 24) && p.vs[2] * 2 < (1 << 24);
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
    if (GATHER && !CSONLY && (nblocks >= 2048 || p.Nk >= 32768 || want64 || want96) && !chipmunk_get_option("attn_no_order")) {
        const int slots = (want64 ? 1 : wg_per_cu) * device_cu_count();
        const int max_slices = 3 * 2 * device_cu_count();         // 1536 x 104 KB = 160 MB of partials at most
        const int64_t cap = nblocks + max_slices;                 // plan entries == workgroups launched
        const size_t plan_bytes = (size_t)cap * 2 * sizeof(int32_t);
        const size_t ws_off = (TICKET_BYTES + plan_bytes + 255) & ~(size_t)255;
        unsigned char *sc = (unsigned char *)chipmunk_scratch(stream, ws_off + (size_t)max_slices * 26 * 256 * sizeof(f32x4));
        if (sc && (size_t)max_slices * sizeof(int32_t) <= TICKET_BYTES) {
            int32_t *plan = (int32_t *)(sc + TICKET_BYTES);
            hipLaunchKernelGGL(attn_plan_kernel, dim3(1), dim3(1024), 0, stream, p.counts, plan, (int)nblocks, p.Nk, slots,
                               max_slices, (int)cap, chipmunk_get_option("attn_no_tail") ? 0 : 1);
            pp.plan = plan;
            pp.tickets = (int32_t *)sc;
            pp.ws = (float *)(sc + ws_off);
            pp.xcd_chunks = 0;
            grid = cap;
            if (want96) return chipmunk_csp96_launch(pp, INPLACE ? 1 : 0, (int)grid, stream);
            if (want64) return chipmunk_csp64_launch(pp, INPLACE ? 1 : 0, (int)grid, stream);
        }
    }
    // Work-balanced launch (BAL, see attn_kernel): gathered launches of a few rounds whose last round would be poorly filled.
    // Option attn_balanced: 1 = always (tests), 3 = by shape, 0 / 2 = never.  NOT in the product library (round 6): the row-split tail below
    // beats it wherever the host can tell the two apart (21 and 42 tiles per item; it is 2-3 % ahead at 84), and the host does not know the
    // counts.  tools/probes/mm1_forms/build.sh compiles it in (-DCHIPMUNK_ATTN_PROBES) and its tests run against that library.
#ifdef CHIPMUNK_ATTN_PROBES
    if constexpr (GATHER && !CSONLY && !WRITE_L) {
        const int ob = chipmunk_get_option("attn_balanced");
        const int64_t slots = wg_per_cu * (int64_t)device_cu_count();
        const int64_t rounds = (nblocks + slots - 1) / slots;
        const bool by_shape = nblocks > slots && rounds <= 8 && (rounds * slots - nblocks) * 4 >= slots && nblocks <= BAL_MAX_ITEMS;
        if (!pp.plan && !pp.xcd_chunks && (ob == 1 || (ob == 3 && by_shape)) && slots <= 4000) {
            const int64_t nwg = nblocks < slots ? nblocks : slots;
            unsigned char *sc = (unsigned char *)chipmunk_scratch(stream, TICKET_BYTES + (size_t)(nwg + 1) * BAL_STATE_F4 * sizeof(f32x4));
            if (sc) {
                auto kb = attn_kernel<GATHER, INPLACE, WRITE_L, CSONLY, true>;
                static uint64_t lds_set_b = 0;
                ensure_dynamic_lds((const void *)kb, ATTN_LDS_BYTES, lds_set_b);
                pp.tickets = (int32_t *)sc;
                pp.ws = (float *)(sc + TICKET_BYTES);
                hipLaunchKernelGGL(kb, dim3((unsigned)nwg), dim3(256), ATTN_LDS_BYTES, stream, pp);
                CM_LAUNCH_CHECK();
                return CHIPMUNK_OK;
            }
        }
    }
#endif
    // Row-split tail (MIX, see attn_kernel): the last `nblocks mod slots` items of a gathered launch of at least one full round run as
    // three 64-row workgroups each, when the thirds get a CU each (3 x rem <= slots / 2: a third is cheap alone on its CU, not when paired --
    // 644 items, 132 in the tail: 72.0 vs 67.5 us for the plain-output form with the looser 3 x rem <= slots).  Option attn_row_split:
    // 0 = by shape, 1 = always (the last min(nblocks, slots / 3) items; tests), 2 = never.
    if constexpr (GATHER && !CSONLY && !WRITE_L) {
        const int orow = chipmunk_get_option("attn_row_split");
        const int64_t slots = wg_per_cu * (int64_t)device_cu_count();
        int64_t rem = nblocks > slots ? nblocks % slots : 0;
        if (orow == 1) rem = nblocks < slots / 3 ? nblocks : slots / 3;
        if (!pp.plan && !pp.xcd_chunks && orow != 2 && rem > 0 && (orow == 1 ? rem * 3 <= slots : rem * 6 <= slots)) {
            auto km = attn_kernel<GATHER, INPLACE, WRITE_L, CSONLY, false, true>;
            static uint64_t lds_set_m = 0;
            ensure_dynamic_lds((const void *)km, ATTN_LDS_BYTES, lds_set_m);
            pp.split_full = (int)(nblocks - rem);
            pp.nsplit = 0;
            hipLaunchKernelGGL(km, dim3((unsigned)(nblocks - rem + 3 * rem)), dim3(256), ATTN_LDS_BYTES, stream, pp);
            CM_LAUNCH_CHECK();
            return CHIPMUNK_OK;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, pp);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

// The one-workgroup-per-CU kernels of attn64.hip pay when their workgroups fill the rounds they occupy: at least one round,
// at most a quarter of the last one empty (FLUX dense: 408 workgroups = 80 % of two rounds, 220 vs 268 us; a
// head-parallel rank's single HunyuanVideo head: 466 = 91 %, 5.97 vs 7.33 ms), and at least 1 024 keys to amortise the
// per-workgroup prologue.
bool fills_rounds(int64_t nwg, int Nk) {
    const int64_t cus = device_cu_count(), rounds = (nwg + cus - 1) / cus;
    return nwg >= cus && nwg * 4 >= 3 * rounds * cus && Nk >= 1024;
}
// dense attention (256-row workgroups); option attn_dense64: 0 = by size, 1 = always, 2 = never
bool use_dense64(int B, int H, int Nq, int Nk) {
    const int o = chipmunk_get_option("attn_dense64");
    if (o) return o == 1 && Nk >= 64;
    return fills_rounds((int64_t)B * H * ((Nq + 255) / 256), Nk);
}
// column-sum pass (a workgroup = four 192-row groups of a head); option attn_colsum64 likewise
bool use_colsum64(int B, int H, int Nq, int Nk) {
    const int o = chipmunk_get_option("attn_colsum64");
    if (o) return o == 1 && Nk >= 64;
    return fills_rounds((int64_t)B * H * (((Nq + QG - 1) / QG + 3) / 4), Nk);
}

int check_common(const void *q, const void *k, const void *v, const void *o, int B, int H, int Nq, int Nk) {
    CM_CHECK(q && k && v && o, "attention: null tensor pointer");
    CM_CHECK(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: B,H,Nq,Nk must be positive (got %d,%d,%d,%d)", B, H, Nq, Nk);
    CM_CHECK((int64_t)B * H * ((Nq + QG - 1) / QG) < (1ll << 31), "attention: too many query groups");
    CM_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0,
             "attention: q/k/v/o must be 16-byte aligned (the epilogues store whole rows as 16-byte pieces)");
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
    CM_CHECK(o_in != nullptr && ((uintptr_t)o_in & 15) == 0, "csp_attn_out: o_in missing or not 16-byte aligned");
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

// chipmunk_csp_attn_out over ragged index rows: row (b, h, g) = indices + idx_offsets[(b*H + h)*G + g], as wide as the distance
// to the next offset (B*H*G + 1 offsets, multiples of 4); what chipmunk_compact_indices lays out
extern "C" int chipmunk_csp_attn_out_ragged(const void *q, const void *k, const void *v, const void *o_in, void *o_out,
                                            const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                                            const int64_t o_strides[3], const int32_t *indices, const int64_t *idx_offsets,
                                            const int32_t *counts, int B, int H, int Nq, int Nk, int o_scale, void *stream) {
    if (int e = check_common(q, k, v, o_out, B, H, Nq, Nk)) return e;
    CM_CHECK(o_in != nullptr && ((uintptr_t)o_in & 15) == 0, "csp_attn_out_ragged: o_in missing or not 16-byte aligned");
    CM_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");
    CM_CHECK(indices && counts && idx_offsets, "csp_attn_out_ragged: indices / offsets / counts missing");
    CM_CHECK(((uintptr_t)indices & 15) == 0, "csp_attn_out_ragged: indices must be 16-byte aligned");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    if (int e = check_strides(o_strides, "o")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o_out;
    p.o_in = (const uint16_t *)o_in;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i], p.os[i] = o_strides[i];
    p.indices = indices, p.counts = counts, p.idx_off = idx_offsets;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG, p.idx_stride = 0;
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

// o_strides == nullptr: contiguous [B, H, Nq, 128]; else (batch, head, row) element strides of `o` (rows of 128 contiguous
// elements), e.g. token-major [B, Nq, H, 128] storage = {Nq*H*128, 128, H*128}
static int output_strides(int64_t os[3], const int64_t *o_strides, int H, int Nq) {
    if (!o_strides) {
        contiguous_strides(os, H, Nq);
        return CHIPMUNK_OK;
    }
    for (int i = 0; i < 3; ++i) os[i] = o_strides[i];
    return check_strides(o_strides, "o");
}

extern "C" int chipmunk_dense_attn_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                           const int64_t k_strides[3], const int64_t v_strides[3], void *o,
                                           const int64_t *o_strides, float *l, int B, int H, int Nq, int Nk, void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(l != nullptr, "dense_attn: l output missing");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i];
    if (int e = output_strides(p.os, o_strides, H, Nq)) return e;
    p.l_out = l;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG;
    p.o_scale = 1.f;
    if (use_dense64(B, H, Nq, Nk)) {
        CM_CHECK((int64_t)Nk * p.ks[2] * 2 < (1ll << 32) && (int64_t)Nk * p.vs[2] * 2 < (1ll << 32),
                 "attention: one head's K or V spans more than 4 GiB (32-bit DMA offsets)");
        return chipmunk_dense64_launch(q, k, v, o, l, p.qs, p.ks, p.vs, p.os, B, H, Nq, Nk, (hipStream_t)stream);
    }
    return launch_attn<false, false, true>(p, (hipStream_t)stream);
}

extern "C" int chipmunk_dense_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                   const int64_t k_strides[3], const int64_t v_strides[3], void *o, float *l, int B,
                                   int H, int Nq, int Nk, void *stream) {
    return chipmunk_dense_attn_strided(q, k, v, q_strides, k_strides, v_strides, o, nullptr, l, B, H, Nq, Nk, stream);
}

extern "C" int chipmunk_dense_colsum_attn_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                                  const int64_t k_strides[3], const int64_t v_strides[3], const float *pin,
                                                  void *o, const int64_t *o_strides, void *cs, float *l, int B, int H, int Nq,
                                                  int Nk, int cs_stride, void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(l && cs && pin, "dense_colsum_attn: p / cs / l missing");
    CM_CHECK(cs_stride >= Nk, "dense_colsum_attn: cs row stride %d < Nk %d", cs_stride, Nk);
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i];
    if (int e = output_strides(p.os, o_strides, H, Nq)) return e;
    p.l_out = l, p.p_in = pin, p.cs = (uint16_t *)cs, p.cs_stride = cs_stride;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG;
    p.o_scale = 1.f;
    // Two passes: (1) dense attention, (2) a K-only pass that recomputes the scores and reduces the column sums; they
    // share nothing but their inputs.  (A single fused pass was built first: it sat on the 256-VGPR cliff and measured
    // 885 vs 480 us at FLUX size, 36.9 vs 23.5 ms for two HunyuanVideo heads; removed.)
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (chipmunk_get_option("attn_fused_colsum") == 5) {   // probe: the K-only pass alone (o and l are not written)
        if (use_colsum64(B, H, Nq, Nk)) return chipmunk_colsum64_launch(p, st);
        return launch_attn<false, false, false, true>(p, st);
    }
    if (use_dense64(B, H, Nq, Nk)) {
        CM_CHECK((int64_t)Nk * p.ks[2] * 2 < (1ll << 32) && (int64_t)Nk * p.vs[2] * 2 < (1ll << 32),
                 "attention: one head's K or V spans more than 4 GiB (32-bit DMA offsets)");
        // one pass (attn64.hip MODE 3: the column sums ride the dense kernel's softmax pipeline) when the partial-sum scratch
        // is available; option attn_fused_colsum = 2 keeps the two passes
        if (chipmunk_get_option("attn_fused_colsum") != 2) {
            // all (batch, head) pairs in one launch if the partial sums fit (21 GB at HunyuanVideo size), else in chunks of heads
            // -- halved until the buffer can be had, down to one head (0.9 GB)
            int hc = H;
            uint16_t *part = nullptr;
            const int hc_min = chipmunk_get_option("attn_fused_colsum") == 4 ? 1 : 0;   // 4: one head per launch (test of the chunked form)
            if (hc_min) hc = 1;
            for (; hc >= 1; hc /= 2) {
                part = (uint16_t *)chipmunk_big_scratch(st, chipmunk_colsum_part_bytes(hc == H ? B : 1, hc, Nq, Nk));
                if (part) break;
            }
            if (part && hc == H) return chipmunk_dense64_colsum_launch(p, part, st);
            if (part) {
                for (int b = 0; b < B; ++b)
                    for (int h0 = 0; h0 < H; h0 += hc) {
                        AttnParams c = p;
                        const int hn = H - h0 < hc ? H - h0 : hc;
                        const int64_t bh0 = (int64_t)b * H + h0;
                        c.q += b * p.qs[0] + h0 * p.qs[1], c.k += b * p.ks[0] + h0 * p.ks[1], c.v += b * p.vs[0] + h0 * p.vs[1];
                        c.o += b * p.os[0] + h0 * p.os[1];
                        c.l_out += bh0 * Nq, c.p_in += bh0 * Nq, c.cs += bh0 * p.G * (int64_t)cs_stride;
                        c.B = 1, c.H = hn;
                        if (int e = chipmunk_dense64_colsum_launch(c, part, st)) return e;
                    }
                return CHIPMUNK_OK;
            }
        }
        rc = chipmunk_dense64_launch(q, k, v, o, l, p.qs, p.ks, p.vs, p.os, B, H, Nq, Nk, st);
    } else {
        rc = launch_attn<false, false, true>(p, st);
    }
    if (rc != CHIPMUNK_OK) return rc;
    // second pass: one wave per 192-row group where that fills the CUs (attn64.hip), the general kernel's K-only pass otherwise
    if (use_colsum64(B, H, Nq, Nk)) return chipmunk_colsum64_launch(p, st);
    return launch_attn<false, false, false, true>(p, st);
}

extern "C" int chipmunk_dense_colsum_attn(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                          const int64_t k_strides[3], const int64_t v_strides[3], const float *pin,
                                          void *o, void *cs, float *l, int B, int H, int Nq, int Nk, int cs_stride,
                                          void *stream) {
    return chipmunk_dense_colsum_attn_strided(q, k, v, q_strides, k_strides, v_strides, pin, o, nullptr, cs, l, B, H, Nq, Nk,
                                              cs_stride, stream);
}

// dense attention + the mask-recompute step's key selection in one call: column sums stay in the per-wave partial rows of the
// one-pass kernel (attn64.hip MODE 3) and the top-k mask kernel adds the three rows of a group itself -- the [B,H,G,Nk] `cs`
// tensor of the reference (3.55 GB at HunyuanVideo size, written by dense_colsum_attn.cu:267-277 and re-read by
// modules/attn.py:76-84) is never materialised.  Same bits as chipmunk_dense_colsum_attn followed by chipmunk_topk_mask.
// Returns CHIPMUNK_ERR_UNSUPPORTED (nothing launched) when the launch does not take the one-pass route in one piece; the
// caller then runs the two operators.
extern "C" int chipmunk_dense_colsum_topk_mask_strided(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                               const int64_t k_strides[3], const int64_t v_strides[3], const float *pin, void *o,
                                               const int64_t *o_strides, float *l, int B, int H, int Nq, int Nk, const void *static_mask,
                                               int64_t static_stride, int static_rows, const void *group_flags, void *mask,
                                               int topk, double random_amount, void *stream) {
    if (int e = check_common(q, k, v, o, B, H, Nq, Nk)) return e;
    CM_CHECK(l && pin && mask, "dense_colsum_topk_mask: p / l / mask missing");
    if (int e = check_strides(q_strides, "q")) return e;
    if (int e = check_strides(k_strides, "k")) return e;
    if (int e = check_strides(v_strides, "v")) return e;
    hipStream_t st = (hipStream_t)stream;
    if (!use_dense64(B, H, Nq, Nk) || chipmunk_get_option("attn_fused_colsum") == 2 || (Nk & 3) || Nk > 1024 * 120)
        return CHIPMUNK_ERR_UNSUPPORTED;
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = q_strides[i], p.ks[i] = k_strides[i], p.vs[i] = v_strides[i];
    CM_CHECK((int64_t)Nk * p.ks[2] * 2 < (1ll << 32) && (int64_t)Nk * p.vs[2] * 2 < (1ll << 32),
             "attention: one head's K or V spans more than 4 GiB (32-bit DMA offsets)");
    if (int e = output_strides(p.os, o_strides, H, Nq)) return e;
    p.l_out = l, p.p_in = pin, p.cs = nullptr, p.cs_stride = 0;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + QG - 1) / QG;
    p.o_scale = 1.f;
    uint16_t *part = (uint16_t *)chipmunk_big_scratch(st, chipmunk_colsum_part_bytes(B, H, Nq, Nk));
    if (!part) return CHIPMUNK_ERR_UNSUPPORTED;
    if (int e = chipmunk_dense64_colsum_launch(p, part, st)) return e;
    const int nrb = ((Nq + 255) / 256) * 2;
    return chipmunk_topk_mask_parts(part, chipmunk_colsum_part_stride(Nk), nrb, p.G, Nq, static_mask, static_stride, static_rows, group_flags, mask, B * H * p.G, Nk,
                                    topk, random_amount, st);
}

extern "C" int chipmunk_dense_colsum_topk_mask(const void *q, const void *k, const void *v, const int64_t q_strides[3],
                                               const int64_t k_strides[3], const int64_t v_strides[3], const float *pin, void *o,
                                               float *l, int B, int H, int Nq, int Nk, const void *static_mask,
                                               int64_t static_stride, int static_rows, const void *group_flags, void *mask,
                                               int topk, double random_amount, void *stream) {
    return chipmunk_dense_colsum_topk_mask_strided(q, k, v, q_strides, k_strides, v_strides, pin, o, nullptr, l, B, H, Nq, Nk,
                                                   static_mask, static_stride, static_rows, group_flags, mask, topk, random_amount,
                                                   stream);
}
