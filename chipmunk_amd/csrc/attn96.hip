// Gathered (column-sparse) attention on gfx950 with ONE wave per SIMD and ALL four SIMDs of a CU computing:
// a 192-row query group (the reference's sparsity granule, modules/attn.py:95-96) = one workgroup of TWO waves x 96
// rows, two workgroups per CU.  (attn64.hip's 64-row waves leave a quarter of the CU idle on 192-row groups; the
// general kernel of attn.hip puts two 48-row waves on every SIMD and is bound by the issue slots they share.)
//   * per wave: O^T (96 x 128 f32) in a[0:191], the eight K fragments of the 32-key tile in a[192:223], the Q^T
//     fragments of the third 32-row block in a[224:255], of the first in 32 VGPRs, of the second staged in LDS and
//     streamed through a 4-fragment window (64 VGPRs of Q beside the softmax made hipcc spill into the accumulator
//     file) -- accumulator registers are only ever named from inline asm (same audit rule as attn64.hip);
//   * swapped layout S^T = K.Q^T on v_mfma_f32_32x32x16_bf16: a lane owns a query column, softmax statistics are
//     lane-local, the bf16 P^T feeds O^T += V^T.P^T from registers;
//   * per 32-key tile ONE stream of 48 MFMAs, QK^T and PV alternating; the softmax of a block runs IN PLACE on its score
//     registers as a pipeline while the other blocks' MFMAs issue, the three pipelines staggered by 16 slots; the PV
//     stream handles query blocks 0 and 1 in PAIRS that share one V^T fragment read (see the main loop's comment; the
//     slot tables and lgkmcnt values come from tools/gen_attn96_sched.py, which asserts every ordering constraint);
//   * gather: each wave stages half of every K and V tile by LDS-DMA (4 + 4 pieces); LDS row 4*pc + lg of a tile holds
//     packed position (pc >> 2)*16 + lg*4 + (pc & 3), so a lane group reads 4 consecutive indices per tile and piece i
//     takes register i; index registers of four tiles in a ring; one s_barrier (two waves) + one counted vmcnt per tile;
//   * work items from the device-built plan of attn.hip; key slices merged by the last arriver (same hand-off).
#include "common.h"
#include "attn64_regs.h"
#include "attn_params.h"
#include "attn64_util.h"
#include "attn96_sched.h"

namespace {

constexpr int KT = 32;                    // keys per tile
constexpr int TB = KT * 256;              // 8 KiB per K or V tile
constexpr int NSL = 4;                    // ring slots, K and V each
constexpr int VRING = NSL * TB;
constexpr int QLDS = 2 * NSL * TB;        // byte offset of the Q^T block 1 staging (8 KiB per wave, same swizzled layout as a K tile)
constexpr int LDS_BYTES = QLDS + 2 * 8192;   // 80 KiB: two workgroups per CU = all 160 KiB
constexpr float SCALE_LOG2E = 0.08838834764f * 1.44269504089f;
constexpr float MAX_LAG = 4.0f;
// timing ablations (results wrong; read SQ_WAVE_CYCLES, not the clock): 1 = no softmax pipelines, 2 = no V^T reads,
// 4 = no DMA / vmcnt / barrier, 8 = no K re-reads and no Q window reads, 16 = exp2 -> multiply, 32 = no row sums
#ifndef A96_ABL
#define A96_ABL 0
#endif

// O^T block (qb, db) += V^T fragment . P^T fragment (the drain after the loop; the loop issues its MFMAs in place)
template <int QB, int DB>
__device__ __forceinline__ void mfma_pv(const u32x4 &vf, const u32x4 &pf) {
    constexpr int oa = (QB * 4 + DB) * 16;
    // (the leading s_nop: the compiler may assemble `vf` with v_mov copies right in front of this statement and cannot know that the
    //  string is an MFMA, so it adds no wait states between a VALU write and the matrix core's read of that register)
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(oa), "i"(oa + 15));
}
template <int KS, int SLOT>
__device__ __forceinline__ void lds_k(uint32_t addr) {
    constexpr int ka = 192 + KS * 4;
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(ka), "i"(ka + 3), "i"(SLOT * TB) : "memory");
}

#ifdef ATTN96_PROF
// cycle anatomy (tools/attn96_prof.py builds a separate library with -DATTN96_PROF): s_memtime at the segment boundaries
// of every tile, summed per segment, both waves of one mid-grid workgroup.  Not part of the product build.
__device__ unsigned long long g_a96_prof[2 * 8];
#define P96_DECL unsigned long long pt_ = 0, pacc_[7] = {0, 0, 0, 0, 0, 0, 0}; const bool prof_on_ = blockIdx.x == 700
#define P96_START() do { if (prof_on_) pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define P96_MARK(i) do { if (prof_on_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } } while (0)
#define P96_END(w, n) do { if (prof_on_ && lane == 0) { for (int i_ = 0; i_ < 7; ++i_) g_a96_prof[(w) * 8 + i_] = pacc_[i_]; g_a96_prof[(w) * 8 + 7] = (n); } } while (0)
#else
#define P96_DECL
#define P96_START()
#define P96_MARK(i)
#define P96_END(w, n)
#endif

template <bool INPLACE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void csp96_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5, l15 = lane & 15, lg = lane >> 4;
#ifdef ATTN96_PROF
    const unsigned long long t_entry_ = __builtin_amdgcn_s_memtime();
#endif
    int wid = blockIdx.x;
    const int wid0 = wid;
    wid = p.plan[2 * wid0];
    if (wid < 0) return;
    const int meta = p.plan[2 * wid0 + 1];
    const int sp = meta & 0xff, nsp = (meta >> 8) & 0xff;   // (attn_plan_kernel: slice | slices << 8 | scratch slot of slice 0 << 16)
    const int slot0 = meta >> 16, tix = meta >> 16;
    const int bh = wid / p.G, g = wid - bh * p.G;
    const int b = bh / p.H, h = bh - b * p.H;
    const int row0 = g * 192 + w * 96;
    const int count = p.counts[(int64_t)bh * p.G + g];
    const int valid = count < p.Nk ? count : p.Nk;   // packed positions >= Nk are masked out (csp_128_attn.cu:314)
    const int ntiles_all = (valid + KT - 1) / KT;
    const int tbeg = (int)((int64_t)ntiles_all * sp / nsp), tend = (int)((int64_t)ntiles_all * (sp + 1) / nsp);
    const int ntiles = tend - tbeg;
    const int T4 = (ntiles + 3) & ~3;

    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t *vbase = p.v + b * p.vs[0] + h * p.vs[1];
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase), vrsrc = make_rsrc(vbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u, vstride_b = (uint32_t)p.vs[2] * 2u;
    const IndexRow irow = index_row(p, (int64_t)bh * p.G + g);
    const int32_t *idx = irow.ptr;

    // ---- lane-constant LDS addresses (tile = [32 rows][256 B]; K chunk swizzle c ^ (r & 15), V chunk swizzle c ^ ((r & 3) << 2))
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    uint32_t kad[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kad[ks] = lds0 + l31 * 256 + (((2 * ks + hf) ^ l15) << 4);
    const uint32_t qad = lds0 + QLDS + w * 8192 + l31 * 256;   // + (((2*ks + hf) ^ l15) << 4) = kad[ks] - lds0 - l31*256
    uint32_t vad[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
        vad[db] = lds0 + VRING + (4 * hf + (l15 >> 2)) * 256 + (((db * 4 + 2 * (lg & 1) + ((l15 & 3) >> 1)) ^ ((l15 >> 2) << 2)) << 4) +
                  (l15 & 1) * 8;
    // DMA: wave w stages pieces 4w + i, i = 0..3, of the K tile and of the V tile; LDS row 4*(4w+i) + lg <- packed position
    // w*16 + lg*4 + i of the tile
    uint32_t kswz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kswz[i] = (uint32_t)(l15 ^ (4 * i + lg)) << 4;
    const uint32_t vswz = (uint32_t)(l15 ^ (lg << 2)) << 4;
    // index rows through a buffer resource sized to the row: no 64-bit lane addresses, and positions past the row's end read 0
    // (hardware range check) -- padding tiles are masked, key 0 is as good as any
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)idx, 0, irow.width * 4, 0x00020000);
    const int nk1 = p.Nk - 1;
    auto load_idx = [&](int T) {   // (idx_stride is a multiple of 4 here: launch_attn sends other launches to the general kernel)
        const int base = ((tbeg + T) * KT + w * 16 + lg * 4) * 4;
        u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(irsrc, base, 0, 0);
        return v4;   // raw: clamped at first use, an iteration later (a clamp here would wait for the load and every DMA in flight)
    };
    auto clamp_idx = [&](u32x4 &v4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)   // memory safety for malformed indices: clamp to [0, Nk-1] (one v_med3 each)
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(v4[e]) : "v"(v4[e]), "v"(nk1));
    };
    auto issue_k1 = [&](const u32x4 &keys, int slot, int i) {
        blds16(krsrc, __umul24(keys[i], kstride_b) + kswz[i], 0, smem + slot * TB + (4 * w + i) * 1024);
    };
    auto issue_v1 = [&](const u32x4 &keys, int slot, int i) {
        blds16(vrsrc, __umul24(keys[i], vstride_b) + vswz, 0, smem + VRING + slot * TB + (4 * w + i) * 1024);
    };
    auto issue_k = [&](const u32x4 &keys, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_k1(keys, slot, i);
    };
    auto issue_v = [&](const u32x4 &keys, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_v1(keys, slot, i);
    };

    // ---- prologue.  Order matters for a lone wave (nothing else runs on its SIMD while it waits): the index loads go out FIRST, the
    //      accumulator file is zeroed while they fly, then the gathers of K(0..3) / V(0..1) are issued and only then Q is loaded and
    //      folded -- Q's load and arithmetic overlap the gather's latency instead of preceding it (three serial round trips -> two).
    //      K(0); "iterations" -3..-1 (iteration i issues K(i+4) and V(i+2); V(-1) does not exist: V(0) goes into
    //      its slot so that the first tile's PV -- P = 0 -- multiplies finite numbers)
    u32x4 ir[4];   // gather keys of four tiles (entry = tile mod 4)
    ir[0] = load_idx(0), ir[1] = load_idx(1), ir[2] = load_idx(2), ir[3] = load_idx(3);
    asm volatile(A96_ZERO_O ::: A64_CLOBBER_ALL);   // accumulator file: O^T = 0
    clamp_idx(ir[0]), clamp_idx(ir[1]), clamp_idx(ir[2]), clamp_idx(ir[3]);
    issue_k(ir[0], 0);
    issue_k(ir[1], 1), issue_v(ir[0], 3);
    issue_k(ir[2], 2), issue_v(ir[0], 0);
    issue_k(ir[3], 3), issue_v(ir[1], 1);
    ir[0] = load_idx(4);
    // ---- Q^T fragments (B operand: lane = query l31, d = ks*16 + hf*8 .. +7); the accumulator file was zeroed above
    // block 0 in 32 VGPRs, block 2 in a[224:255], block 1 staged in LDS (its 32 registers do not fit beside the softmax) and
    // streamed through a two-fragment window, one ds_read_b128 per k step and tile
    u32x4 qv[8];
    // The loop WITHOUT a reference point (see the loop selection below) is possible where |s_ij| <= |q_i| max_j |k_j| =: M_i is
    // provably small for every query of the wave: |q_i|^2 of the wave's rows -> the decision (wave-uniform).
    bool nomax = false;
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1];
        float qss[3] = {0.f, 0.f, 0.f};   // this lane's half of |q|^2 per query block
        // The Q^T fragments.  For the loop without a reference point Q is folded with log2(e)/sqrt(D) HERE, once per item
        // (bf16(q c), round to nearest even): the MFMA then delivers the exponent of 2 itself and the loop has no per-score
        // multiply-add (48 of 200 VALU issues per tile).  The rounding of q c moves an exponent by at most |s| c 2^-9, which the
        // bound below keeps at <= 0.11 (measured at C3 size: error / tolerance 0.016 at unit scale, 0.13 with q x 2; unbounded it
        // reached 1.5 at q x 6).  The running-maximum loop keeps the exact q and its multiply-add: the folded fragments are
        // written optimistically, and a wave that fails the bound fetches its rows again (from L2) and stores them as they are.
        auto load_q = [&](auto foldc) {
            constexpr bool FOLD = decltype(foldc)::value != 0;
            static_for<0, 24>([&](auto f) {
                constexpr int F = decltype(f)::value, qb = F >> 3, ks = F & 7;
                const int qrow = row0 + qb * 32 + l31;
                u32x4 val = {0u, 0u, 0u, 0u};
                if (qrow < p.Nq) val = *(const u32x4 *)(qp + (int64_t)qrow * p.qs[2] + ks * 16 + hf * 8);
                if constexpr (FOLD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(val[e] << 16), hi = __uint_as_float(val[e] & 0xffff0000u);
                        qss[qb] = __builtin_fmaf(lo, lo, qss[qb]);
                        qss[qb] = __builtin_fmaf(hi, hi, qss[qb]);
                        val[e] = pack_bf16x2(lo * SCALE_LOG2E, hi * SCALE_LOG2E);
                    }
                }
                if constexpr (qb == 0) qv[ks] = val;
                else if constexpr (qb == 2) acc_write4<224 + ks * 4>(val);
                else *(u32x4 *)(smem + QLDS + w * 8192 + l31 * 256 + (((2 * ks + hf) ^ l15) << 4)) = val;   // (own wave's rows only)
            });
        };
        load_q(ic<1>{});
        if (p.kmax) {
            const float km = p.kmax[bh];
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                float a = qss[qb], c2 = qss[qb];
                lane_swap32(a, c2);
                // M_i in exponent units, with room for the bf16 rounding of the folded Q (|q'| <= |q c| (1 + 2^-8))
                ok = ok && (__builtin_sqrtf(a + c2) * km * SCALE_LOG2E <= 55.0f);
            }
            nomax = __builtin_amdgcn_ballot_w64(!ok) == 0;
        }
        if (!nomax) load_q(ic<0>{});
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, 8>([&](auto i) { lds_k<decltype(i)::value, 0>(kad[decltype(i)::value]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- main loop: ONE stream of 48 MFMAs per 32-key tile, QK^T and PV strictly alternating (slot sigma = 0..47):
    //   even sigma: S(t)[qb] += K(t)[ks] . Q^T[qb][ks], qb = sigma/16, ks = (sigma%16)/2 -- query block by query block, so
    //               block qb's scores are complete at sigma = 16 qb + 14 and its softmax can start while the other blocks'
    //               MFMAs are still running (a dependent chain on one accumulator, but every other MFMA is a PV one);
    //   odd sigma = 2i+1: PV element i, O[qb'] += V^T fragment . P^T, from the slot tables of attn96_sched.h:
    //               * loop without a reference point (Pair): i < 3 the last three elements of block 2 of tile t-2, i = 3..18 query
    //                 blocks 0 and 1 of tile t-1 in PAIRS on one V^T fragment (16 fragment reads per tile instead of 24; P[0] is
    //                 double-buffered by tile parity because its reads now overlap the next tile's packs), i >= 19 block 2 of t-1;
    //               * running-maximum loop (Major): query-block-major, so that every element of a block precedes the slot in
    //                 which that block's reference point may move (the rescale of its 64 accumulator registers).
    // The softmax of a block is a pipeline (maxima, [rare] reference update + rescale, then exp2 / row sum / bf16 pair, one
    // instruction per stage and slot; the running-maximum form also x = s - m); the three blocks' pipelines are staggered by
    // 16 slots, block 1 runs over the tile seam and block 2 entirely in the next tile.  LDS reads -- two 8-byte halves per V^T
    // fragment into a 4-entry window, the Q^T block-1 window (4 k steps deep), K(t+1) fragment ks after block 2's MFMA on k
    // step ks -- are spread one or two per slot over ALL slots (2-3 after every PV MFMA and none after the QK^T ones made the
    // odd slots 6-8 issues long and left the even ones at 3: a slot cannot be shorter than its MFMA).  The generator asserts
    // that every P^T fragment is complete before the first PV MFMA that reads it and written only after the last one that read
    // its predecessor, the same for the windows, and computes the lgkmcnt in front of each consumer (reads return in order).
    f32x16 s[3];
#pragma unroll
    for (int q2 = 0; q2 < 3; ++q2)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[q2][r] = -INFINITY;   // "tile -1": exp2 -> p = 0
#pragma unroll
    for (int r = 0; r < 7; ++r) s[1][r] = 0.f;   // (block 1's first seven elements are past the exp2 stage at the tile seam: p = 0)
    u32x4 pw[3][2] = {};         // P^T fragments (block, key slab); block 0 of EVEN tiles (its reads for tile t-1 overlap its
    u32x4 pwo[2] = {};           //   packs for tile t: double-buffered by tile parity -- block 0 of ODD tiles lives here)
    u32x2 vlo[4] = {}, vhi[4] = {};   // V^T fragment window, two 8-byte halves per entry (zero: the first PV elements multiply P = 0 by it)
    u32x4 q1w[4] = {};           // Q^T block-1 window
    float m[3] = {-INFINITY, -INFINITY, -INFINITY}, nmsc[3] = {0.f, 0.f, 0.f}, mlag[3] = {-INFINITY, -INFINITY, -INFINITY};
    float lacc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    int vl_prev = KT;            // packed positions that exist in the previous tile (its block 2 is masked in this one)

    auto vhalf_read = [](uint32_t addr, auto offc, u32x2 &dst) __attribute__((always_inline)) {
        constexpr int OFF = decltype(offc)::value;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
    };
    auto q1_read = [&](auto ksc, u32x4 &dst) __attribute__((always_inline)) {
        constexpr int KS = decltype(ksc)::value;
        const uint32_t addr = qad + (kad[KS] - lds0 - l31 * 256);
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
    };
    // dead packed positions of a ragged / padding tile -> -inf (wave-uniform, last tiles only).  Element (r, hf) of a block is
    // LDS row (r&3) + 8*(r>>2) + 4*hf; row 4*pc + lg holds position (pc>>2)*16 + lg*4 + (pc&3) = (r>>3)*16 + (r&3)*4 + 2*((r>>2)&1) + hf
    auto mask_block = [&](f32x16 &sq, int vleft) __attribute__((always_inline)) {
        if (__builtin_expect(vleft < KT, 0)) {
            int thr = vleft - hf;   // opaque: the 16 per-register constants are compared as immediates
            asm volatile("" : "+v"(thr));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r >> 3) * 16 + (r & 3) * 4 + 2 * ((r >> 2) & 1) >= thr) sq[r] = -INFINITY;
        }
    };
    // maxima of the 16 scores a lane holds of a block: step 0 = v_max + v_max3 on registers 0..3, steps 1..3 two v_max3 each
    auto max_step = [&](auto stepc, const f32x16 &sq, float &mx) __attribute__((always_inline)) {
        constexpr int K = decltype(stepc)::value, r0 = K * 4;
        if constexpr (K == 0)
            asm volatile("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "=&v"(mx) : "v"(sq[0]), "v"(sq[1]), "v"(sq[2]), "v"(sq[3]));
        else
            asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(mx) : "v"(sq[r0]), "v"(sq[r0 + 1]), "v"(sq[r0 + 2]), "v"(sq[r0 + 3]));
    };
    auto max_halves = [&](float &mx) __attribute__((always_inline)) {   // the other 16 keys of the block live in lane ^ 32
        float t0;
        asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(mx), "=&v"(t0));
    };
    // the (rare) move of block QB's reference point; the pending PV of that block has been issued at least two slots ago
    auto update_block = [&](auto qq, float mx) __attribute__((always_inline)) {
        constexpr int QB = decltype(qq)::value;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > mlag[QB]) != 0, 0)) {
            constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;
            const float m_new = max2(m[QB], mx * SCALE_LOG2E);   // m in exponent units (the merge of key slices compares them), mx raw
            const float alpha = __builtin_amdgcn_exp2f(m[QB] - m_new);
            lacc[QB][0] *= alpha;
            lacc[QB][1] *= alpha;
            m[QB] = m_new;
            nmsc[QB] = -m_new;
            mlag[QB] = m_new * (1.0f / SCALE_LOG2E) + LAG_RAW;   // raw units: compared with the raw block maximum
            float tmp;
            if constexpr (QB == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" A96_SCALE_QB0 : "=&v"(tmp) : "v"(alpha));
            if constexpr (QB == 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" A96_SCALE_QB1 : "=&v"(tmp) : "v"(alpha));
            if constexpr (QB == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" A96_SCALE_QB2 : "=&v"(tmp) : "v"(alpha));
        }
    };
    // window slot W (0..23) of block QB's x / exp2 / row sum / pack pipeline, in place on its score registers
    auto window = [&](auto qq, auto wc, auto nmc, auto parc) __attribute__((always_inline)) {
        constexpr int QB = decltype(qq)::value, W = decltype(wc)::value, PAR = decltype(parc)::value;   // PAR: parity of the block's tile
        constexpr bool NOMAX = decltype(nmc)::value != 0;
        if constexpr (W < 4 && !NOMAX) {   // x = s c - m (running-maximum form only: the other takes exp2 of the score itself)
            static_for<4 * W, 4 * W + 4>([&](auto ee) {
                constexpr int E = decltype(ee)::value;
                float x = __builtin_fmaf(s[QB][E], SCALE_LOG2E, nmsc[QB]);
                pin(x);
                s[QB][E] = x;
            });
        }
        if constexpr (W >= 1 && W <= 16) {
#if A96_ABL & 16
            float e = s[QB][W - 1] * 0.5f;   // timing ablation: a full-rate multiply in place of the transcendental
#else
            float e = __builtin_amdgcn_exp2f(s[QB][W - 1]);
#endif
            pin(e);
            s[QB][W - 1] = e;
        }
        if constexpr (W >= 2 && W <= 17 && !(A96_ABL & 32)) {
            lacc[QB][W & 1] += s[QB][W - 2];
            pin(lacc[QB][W & 1]);
        }
        if constexpr (W >= 6 && W <= 20 && (W & 1) == 0) {   // (even W = even slots: the odd ones carry the LDS reads)
            constexpr int K = (W - 6) >> 1;   // pair (2K, 2K+1): slab K >> 2, dword K & 3
            uint32_t pk = pack_bf16x2(s[QB][2 * K], s[QB][2 * K + 1]);
            pin(pk);
            if constexpr (QB == 0 && PAR == 1) pwo[K >> 2][K & 3] = pk;
            else pw[QB][K >> 2][K & 3] = pk;
        }
    };

    P96_DECL;
    P96_START();
#ifdef ATTN96_PROF
    if (prof_on_) pacc_[2] = pt_ - t_entry_;   // prologue: kernel entry -> first tile
#endif
    auto tile = [&](auto slc, auto nmc, int t) __attribute__((always_inline)) {
        constexpr int SL = decltype(slc)::value;
        constexpr bool NOMAX = decltype(nmc)::value != 0;   // fixed reference point: no maxima, no update / rescale
        constexpr int VSL = (SL + 3) & 3;        // slot of V(t-1)
        constexpr int KNSL = (SL + 1) & 3;       // slot of K(t+1)
        // K(t+1) and V(t-1) were issued three iterations ago; an iteration is 8 pieces + 1 (4) index loads
        if constexpr (!(A96_ABL & (4 | 128))) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // (ablation 128: no wait, 64: no barrier)
        if constexpr (!(A96_ABL & (4 | 64))) __builtin_amdgcn_s_barrier();
        P96_MARK(0);
        if constexpr (!(A96_ABL & 4)) ir[(SL + 1) & 3] = load_idx(t + 5);
        const int vl_cur = t < ntiles ? valid - (tbeg + t) * KT : 0;   // packed positions of this tile that exist
        float mxa, mxb, mxc;   // block maxima: this tile's block 0, this tile's block 1, the previous tile's block 2
        __builtin_amdgcn_sched_barrier(0);
#ifdef A96_FORCE_MAJOR
        using S = a96s::Major;
#else
        using S = std::conditional_t<NOMAX, a96s::Pair, a96s::Major>;   // slot tables (tools/gen_attn96_sched.py)
#endif
        static_for<0, 48>([&](auto sg) {
            constexpr int SG = decltype(sg)::value;
            // LDS reads return in order: a wait states how many of the youngest may still be in flight
#ifdef A96_TAILWAIT0
            if constexpr (SG == 1 || SG == 3 || SG == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else
#endif
#ifdef A96_ALLWAIT0
            if constexpr ((SG & 1) == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else
#endif
            if constexpr (S::WAIT[SG] >= 0 && !(A96_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(S::WAIT[SG]) : "memory");
            if constexpr ((SG & 1) == 0) {   // ---- QK^T
                constexpr int qb = SG / 16, ks = (SG % 16) / 2, ka = 192 + ks * 4, qa = 224 + ks * 4;
                if constexpr (qb == 2) {
                    if constexpr (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s[2]) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s[2]) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
                } else {
                    const u32x4 &qf = qb == 0 ? qv[ks] : q1w[ks & 3];
                    if constexpr (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], %3, 0" : "=v"(s[qb]) : "i"(ka), "i"(ka + 3), "v"(qf));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], %3, %0" : "+v"(s[qb]) : "i"(ka), "i"(ka + 3), "v"(qf));
                }
                if constexpr (SG % 4 == 2 && SG < 32 && !(A96_ABL & 4)) {   // the DMA of K(t+4) -> slot of K(t), V(t+2) -> slot of V(t-2)
                    constexpr int PC = (SG - 2) / 4;
                    if constexpr (PC == 0) clamp_idx(ir[SL]);   // loaded an iteration ago; V(t+4) reuses the clamped keys two iterations on
                    if constexpr (PC < 4) issue_k1(ir[SL], SL, PC);
                    else issue_v1(ir[(SL + 2) & 3], (SL + 2) & 3, PC - 4);
                }
            } else {                         // ---- PV: O^T[qb][db] += V^T fragment (slab, db) . P^T[qb][slab]
                constexpr int I = (SG - 1) / 2, qbp = S::PV_QB[I], f = S::PV_F[I], e = S::PV_U[I], oa = (qbp * 4 + (f & 3)) * 16;
                const u32x4 vf = {vlo[e][0], vlo[e][1], vhi[e][0], vhi[e][1]};
                // block 0's P^T of tile t-1 (elements 0..2 belong to block 2): the buffer of that tile's parity
                const u32x4 &pf = (qbp == 0 && ((SL + 1) & 1) == 1) ? pwo[f >> 2] : pw[qbp][f >> 2];
                asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(oa), "i"(oa + 15));
            }
            // ---- the slot's LDS reads: halves of V^T fragments of V(t-1), the Q^T block-1 window, K(t+1) fragments
            static_for<0, 2>([&](auto jj) {
                constexpr int J = decltype(jj)::value;
                constexpr int kind = J == 0 ? S::RD_KIND0[SG] : S::RD_KIND1[SG], arg = J == 0 ? S::RD_ARG0[SG] : S::RD_ARG1[SG];
                if constexpr ((kind == 1 || kind == 2) && !(A96_ABL & 2)) {
                    constexpr int en = arg >> 3, fr = arg & 7, off = VSL * TB + (fr >> 2) * 4096 + (kind == 2 ? 2048 : 0);
                    vhalf_read(vad[fr & 3], ic<off>{}, kind == 1 ? vlo[en] : vhi[en]);
                }
                if constexpr (kind == 3 && !(A96_ABL & 8)) q1_read(ic<arg>{}, q1w[arg & 3]);
                if constexpr (kind == 4 && !(A96_ABL & 8)) lds_k<arg, KNSL>(kad[arg]);
            });
            // ---- the three softmax pipelines
            if constexpr (!(A96_ABL & 1)) {
            if constexpr (SG == 0) mask_block(s[2], vl_prev);                       // block 2 of tile t-1
            if constexpr (SG >= 1 && SG <= 4 && !NOMAX) max_step(ic<SG - 1>{}, s[2], mxc);
            if constexpr (SG == 5 && !NOMAX) max_halves(mxc);
            if constexpr (SG == 7 && !NOMAX) update_block(ic<2>{}, mxc);
            if constexpr (SG >= 8 && SG <= 31) window(ic<2>{}, ic<SG - 8>{}, nmc, ic<0>{});
            if constexpr (SG <= 15) window(ic<1>{}, ic<SG + 8>{}, nmc, ic<0>{});                  // block 1 of tile t-1, second part
            if constexpr (SG == 16) mask_block(s[0], vl_cur);                       // block 0 of tile t
            if constexpr (SG >= 17 && SG <= 20 && !NOMAX) max_step(ic<SG - 17>{}, s[0], mxa);
            if constexpr (SG == 21 && !NOMAX) max_halves(mxa);
            if constexpr (SG == 23 && !NOMAX) update_block(ic<0>{}, mxa);
            if constexpr (SG >= 24) window(ic<0>{}, ic<SG - 24>{}, nmc, ic<(SL & 1)>{});
            if constexpr (SG == 32) mask_block(s[1], vl_cur);                       // block 1 of tile t
            if constexpr (SG >= 33 && SG <= 36 && !NOMAX) max_step(ic<SG - 33>{}, s[1], mxb);
            if constexpr (SG == 37 && !NOMAX) max_halves(mxb);
            if constexpr (SG == 39 && !NOMAX) update_block(ic<1>{}, mxb);
            if constexpr (SG >= 40) window(ic<1>{}, ic<SG - 40>{}, nmc, ic<0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        vl_prev = vl_cur;
        P96_MARK(1);
    };

    // tiles 0 .. T4-1 (padding tiles are fully masked), one more tile's worth of slots for the pipelines in flight, then the
    // last three PV elements of block 2
    // No reference point at all where that is provably safe (decided in the prologue): |s_ij| <= M_i <= 55 exponent units over ALL
    // keys of the head, a fortiori over the gathered ones, so p = exp2(s) lies in [2^-56, 2^56]: normal numbers in fp32 and in the
    // bf16 P, row sums and O below 2^56 * 2^17 * max|v| -- nothing overflows, nothing is flushed, and floating point is scale
    // invariant (o = O / l).  Then neither the maxima / update / rescale work nor the subtraction exists in the loop.  Per wave;
    // changes nothing but rounding.  (m = 0 for the merge of key slices: every slice of an item takes the same path.)
    if (nomax) {
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) m[qb] = 0.f, nmsc[qb] = 0.f, mlag[qb] = INFINITY;
    }
    if (nomax) {
        for (int tb = 0;; tb += 4) {
            tile(ic<0>{}, ic<1>{}, tb);
            if (tb >= T4) {
                // the last iteration's V^T fragment reads are still in flight and are consumed after the loop: wait HERE, inside the
                // loop's last block.  The compiler resolves the values leaving the two loops with register copies on the exit edge,
                // and a copy of a register whose ds_read has not landed copies the OLD content (seen: 1 launch in ~15 at 24 heads
                // differing in one 32x32 block of one item -- tools/probes/race96.py; found by the determinism probe at 24 heads)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                break;
            }
            tile(ic<1>{}, ic<1>{}, tb + 1);
            tile(ic<2>{}, ic<1>{}, tb + 2);
            tile(ic<3>{}, ic<1>{}, tb + 3);
        }
    } else {
        for (int tb = 0;; tb += 4) {
            tile(ic<0>{}, ic<0>{}, tb);
            if (tb >= T4) {
                // the last iteration's V^T fragment reads are still in flight and are consumed after the loop: wait HERE, inside the
                // loop's last block.  The compiler resolves the values leaving the two loops with register copies on the exit edge,
                // and a copy of a register whose ds_read has not landed copies the OLD content (seen: 1 launch in ~15 at 24 heads
                // differing in one 32x32 block of one item -- tools/probes/race96.py; found by the determinism probe at 24 heads)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                break;
            }
            tile(ic<1>{}, ic<0>{}, tb + 1);
            tile(ic<2>{}, ic<0>{}, tb + 2);
            tile(ic<3>{}, ic<0>{}, tb + 3);
        }
    }
    // (the last three elements of block 2 of the last real tile: fragments f = 5, 6, 7, read into entries 0..2 by the last iteration)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]), "+v"(vlo[2]), "+v"(vhi[2]));
    mfma_pv<2, 1>((u32x4){vlo[0][0], vlo[0][1], vhi[0][0], vhi[0][1]}, pw[2][1]);
    mfma_pv<2, 2>((u32x4){vlo[1][0], vlo[1][1], vhi[1][0], vhi[1][1]}, pw[2][1]);
    mfma_pv<2, 3>((u32x4){vlo[2][0], vlo[2][1], vhi[2][0], vhi[2][1]}, pw[2][1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ATTN96_PROF
    const unsigned long long t_loop_end_ = __builtin_amdgcn_s_memtime();
#endif
    P96_END(w, T4 + 1);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");

    float lq[3];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        float lp = lacc[qb][0] + lacc[qb][1], lo2 = lp;
        lane_swap32(lp, lo2);
        lq[qb] = lp + lo2;
    }
    auto read_o = [&](auto qq, float (&o)[64]) __attribute__((always_inline)) {
        constexpr int QB = decltype(qq)::value;
        static_for<0, 16>([&](auto ii) {
            constexpr int I = decltype(ii)::value;
            const f32x4 o4 = acc_read4<QB * 64 + I * 4>();
            o[I * 4 + 0] = o4[0], o[I * 4 + 1] = o4[1], o[I * 4 + 2] = o4[2], o[I * 4 + 3] = o4[3];
        });
    };
    // the accumulation base of a query block (INPLACE forms): this lane's 16 x 8 bytes of row qb*32 + l31
    auto load_base = [&](int qb, u32x2 (&old)[16]) __attribute__((always_inline)) {
        const int qrow = row0 + qb * 32 + l31;
        const int64_t ooff = b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + 4 * hf;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32x2 z = {0u, 0u};
            old[i] = qrow < p.Nq ? *(const u32x2 *)(p.o_in + ooff + (i >> 2) * 32 + (i & 3) * 8) : z;
        }
    };
    auto store_o = [&](int qb, const float (&o)[64], float l, const u32x2 (&base)[16]) __attribute__((always_inline)) {
        // O = O^T / l; a lane holds, per d block, four groups of 4 consecutive d of query row qb*32 + l31
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        const int qrow = row0 + qb * 32 + l31;
        if (qrow >= p.Nq) return;
        const int64_t ooff = b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + 4 * hf;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int db = i >> 2, r4 = i & 3;
            const float *o4 = o + db * 16 + r4 * 4;
            const float x0 = o4[0] * inv, x1 = o4[1] * inv, x2 = o4[2] * inv, x3 = o4[3] * inv;
            u32x2 out;
            if constexpr (INPLACE) {
                // bf16 store of o_scale*result, then bf16 reduce-add into the base (csp_attn.cu:294-300)
                const u32x2 old = base[i];
                const float a0 = round_bf16(x0 * p.o_scale), a1 = round_bf16(x1 * p.o_scale);
                const float a2 = round_bf16(x2 * p.o_scale), a3 = round_bf16(x3 * p.o_scale);
                out[0] = pack_bf16x2(__uint_as_float(old[0] << 16) + a0, __uint_as_float(old[0] & 0xffff0000u) + a1);
                out[1] = pack_bf16x2(__uint_as_float(old[1] << 16) + a2, __uint_as_float(old[1] & 0xffff0000u) + a3);
            } else {
                out[0] = pack_bf16x2(x0, x1);
                out[1] = pack_bf16x2(x2, x3);
            }
            *(u32x2 *)(p.o + ooff + db * 32 + r4 * 8) = out;
        }
    };

    if (nsp == 1) {
        // The wave's 96 x 128 bf16 results leave through its own 24 KiB of the (idle) K/V rings: a lane holds 16 x 8 bytes of ONE row per
        // query block -- stored as they stand that is 48 eight-byte stores per lane (and as many base loads for the accumulate forms), 32
        // rows x 16 bytes per instruction.  Through LDS (8-byte writes, chunk index XOR row: the 16 lanes of a write group hit 16
        // different chunks) they become 24 whole-row 16-byte accesses per lane, 4 rows = 1 KiB per instruction; the accumulation base is
        // requested in that layout before the first accumulator is read (one round trip, under the LDS pass).
        constexpr int EP_I = 96 * 256 / 1024;   // 24
        const int er = lane >> 4, ech = lane & 15;
        const int64_t obase = b * p.os[0] + h * p.os[1];
        u32x4 base[EP_I] = {};
        if constexpr (INPLACE) {
#pragma unroll
            for (int j = 0; j < EP_I; ++j) {
                const int qrow = row0 + j * 4 + er;
                if (qrow < p.Nq) base[j] = *(const u32x4 *)(p.o_in + obase + (int64_t)qrow * p.os[2] + ech * 8);
            }
        }
        __syncthreads();   // both waves are done with the rings
        unsigned char *stage = smem + w * (96 * 256);
        static_for<0, 3>([&](auto qq) {
            constexpr int QB = decltype(qq)::value;
            float o[64];
            read_o(qq, o);
            const float l = lq[QB];
            const float inv = (l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f) * (INPLACE ? p.o_scale : 1.f);   // o_scale = +-1: exact
            const int r = QB * 32 + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // bf16(o_scale * result): what the reference stores before its bf16 reduce-add (csp_attn.cu:294-300)
                const u32x2 out = {pack_bf16x2(o[i * 4 + 0] * inv, o[i * 4 + 1] * inv), pack_bf16x2(o[i * 4 + 2] * inv, o[i * 4 + 3] * inv)};
                *(u32x2 *)(stage + r * 256 + ((i ^ (r & 15)) << 4) + hf * 8) = out;   // chunk i = d 8i .. 8i+7: this lane's half hf
            }
        });
#pragma unroll
        for (int j = 0; j < EP_I; ++j) {
            const int r = j * 4 + er, qrow = row0 + r;
            if (qrow >= p.Nq) continue;
            u32x4 v = *(const u32x4 *)(stage + r * 256 + ((ech ^ (r & 15)) << 4));
            if constexpr (INPLACE) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack_bf16x2(__uint_as_float(base[j][e] << 16) + __uint_as_float(v[e] << 16),
                                       __uint_as_float(base[j][e] & 0xffff0000u) + __uint_as_float(v[e] & 0xffff0000u));
            }
            *(u32x4 *)(p.o + obase + (int64_t)qrow * p.os[2] + ech * 8) = v;
        }
#ifdef ATTN96_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (prof_on_ && lane == 0) g_a96_prof[w * 8 + 3] = __builtin_amdgcn_s_memtime() - t_loop_end_;   // epilogue incl. the stores landing
#endif
        return;
    }
    // ---- key-sliced item: publish this slice's (o, m, l) -- [198 values][128 threads], coalesced -- take a ticket; the last
    //      arriver folds ALL slices in slice order and alone runs the epilogue (hand-off as in attn.hip: plain stores ->
    //      barrier -> one agent-scope release -> drained -> relaxed ticket; ticket -> one acquire -> barrier -> plain loads)
    constexpr int SLICE_FLOATS = 26 * 256 * 4;   // the scratch launch_attn reserves per slice (>= 198 * 128)
    int *ticket_s = (int *)smem;
    float *mine = p.ws + (int64_t)(slot0 + sp) * SLICE_FLOATS + tid;
    static_for<0, 3>([&](auto qq) {
        constexpr int QB = decltype(qq)::value;
        float o[64];
        read_o(qq, o);
#pragma unroll
        for (int i = 0; i < 64; ++i) mine[(QB * 64 + i) * 128] = o[i];
        mine[(192 + QB) * 128] = m[QB];
        mine[(195 + QB) * 128] = lq[QB];
    });
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
#pragma unroll 1
    for (int qb = 0; qb < 3; ++qb) {
        float o[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) o[i] = 0.f;
        float mm = -INFINITY, ll = 0.f;
        for (int s2 = 0; s2 < nsp; ++s2) {
            const float *oth = p.ws + (int64_t)(slot0 + s2) * SLICE_FLOATS + tid;
            const float ms = oth[(192 + qb) * 128], ls = oth[(195 + qb) * 128];
            const float m_new = fmaxf(mm, ms);
            if (m_new == -INFINITY) continue;  // nothing so far and an empty slice
            const float a = __builtin_amdgcn_exp2f(mm - m_new);   // (m in exponent units)
            const float c = __builtin_amdgcn_exp2f(ms - m_new);
            mm = m_new;
            ll = ll * a + ls * c;
#pragma unroll
            for (int i = 0; i < 64; ++i) o[i] = o[i] * a + oth[(qb * 64 + i) * 128] * c;
        }
        u32x2 base[16] = {};
        if constexpr (INPLACE) load_base(qb, base);
        store_o(qb, o, ll, base);
    }
}

template <bool INPLACE>
int launch96(const AttnParams &p, int grid, hipStream_t stream) {
    auto kern = csp96_kernel<INPLACE>;
    static uint64_t lds_set = 0;
    ensure_dynamic_lds((const void *)kern, LDS_BYTES, lds_set);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(128), LDS_BYTES, stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

}  // namespace

#ifdef ATTN96_PROF
extern "C" int chipmunk_attn96_prof_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_a96_prof), sizeof(g_a96_prof)) == hipSuccess ? 0 : 2;
}
#endif

// gathered attention over the work plan built by launch_attn (attn.hip); inplace = 1 for the accumulate forms
int chipmunk_csp96_launch(const AttnParams &p, int inplace, int grid, hipStream_t stream) {
    CM_CHECK(p.plan && p.tickets && p.ws, "csp96: work plan missing");
    AttnParams pp = p;
    pp.kmax = chipmunk_knorm_max(p.k, p.ks, p.B, p.H, p.Nk, stream);
    return inplace ? launch96<true>(pp, grid, stream) : launch96<false>(pp, grid, stream);
}
