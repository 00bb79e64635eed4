// Dense attention for long sequences on gfx950: ONE wave per SIMD, 64 query rows per wave, v_mfma_f32_32x32x16_bf16.
//
// The general kernel (attn.hip) runs two 4-wave workgroups per CU with 48 rows per wave on 16x16x32 MFMAs; its loop is
// bound by the issue slots the two waves of a SIMD share.  This kernel is the other point of the design
// space, for the dense launches of HunyuanVideo / Wan size (reference csrc/attn/dense_attn.cu:246-372):
//   * workgroup = 256 query rows = 4 waves x 64 rows, one workgroup per CU, every wave owns the whole 512-entry
//     register file of its SIMD: O^T (64 x 128 f32) in a[0:127], Q^T fragments in a[128:191], the K fragments of the
//     current tile in a[192:255] -- all three only ever touched by inline asm -- and the softmax in the 256 VGPRs;
//   * 64-key tiles; S^T = K.Q^T (swapped operands): a lane owns one query column (lane & 31) and 16 of the 32 keys of a
//     block, so the running maximum / sum are lane-local and the bf16 P^T feeds O^T += V^T.P^T without any cross-lane
//     movement (the k-order permutation is absorbed by the order of the V^T transpose reads);
//   * per tile two phases of 32 MFMAs (32 cycles each: 8 issue slots, of which the fillers may take ~5):
//       A(t): S(t) = K(t).Q^T     beside  exp2 / row sums / bf16 packing of tile t-1
//       B(t): O += V(t-1).P(t-1)  beside  the maxima and s*c - m*c of tile t, the K(t+1) fragment reads, V reads;
//   * K/V tiles by LDS-DMA into 4-slot rings (K four tiles ahead, V two), one s_barrier and one counted vmcnt per tile;
//   * the reference point of the exponentials lags the running maximum by <= 2^4 (as in attn.hip): the rescale of the
//     128 accumulator registers runs a handful of times per item.
// The loop is unrolled over the four ring slots so that every LDS address is base + immediate.
#include "common.h"
#include "attn64_regs.h"
#include "attn_params.h"
#include "attn64_util.h"
#include <type_traits>

namespace {

constexpr int HD = 128;
constexpr int KT = 64;                    // keys per tile
constexpr int WROWS = 64, WGROWS = 256;   // query rows per wave / workgroup
constexpr int TB = KT * HD * 2;           // 16 KiB per K or V tile
constexpr int NSL = 4;                    // ring slots, K and V each
constexpr int VRING = NSL * TB;           // byte offset of the V ring
constexpr int LDS_BYTES = 2 * NSL * TB;   // 128 KiB
constexpr int PSTAGE = 64 * 128;          // MODE 3: a wave's bf16 P tile [64 queries][64 keys] for the column sums on the matrix pipe
constexpr int LDS_BYTES_CSUM = LDS_BYTES + 4 * PSTAGE;   // 160 KiB: all of a CU's LDS
constexpr int CS_EXCH_BYTES = 2 * 4 * 64 * 4;           // MODE 3: [tile parity][wave][64 sums] fp32, the waves' exchange area (see prsrc)
constexpr float SCALE_LOG2E = 0.08838834764f * 1.44269504089f;
constexpr float MAX_LAG = 4.0f;
// timing ablations (tools/attn64_ablate.py builds one library per value; results are wrong, only the clock is read):
// 256 = no vmcnt wait, 128 = no s_barrier, 64 = QK MFMAs write accumulator registers instead of VGPRs, 32 = v_mul instead of v_exp_f32, 1 = no exp2 / row sums / packing, 2 = no maxima / s*c - m*c, 4 = no V^T reads, 8 = no K reads, 16 = no DMA, barrier, vmcnt
#ifndef A64_ABL
#define A64_ABL 0
#endif
#ifndef A64_EB   // exp2 steps of a tile beside its own PV-phase gaps 12..31 (EB) and beside the next tile's QK^T phase (EA); the rest
#define A64_EB 20  // go one per gap into the next PV phase
#endif
#ifndef A64_EA
#define A64_EA 35   // (20 / 35 / 9: with the fixed reference point the first gaps of the PV phase carry no maxima and take nine steps)
#endif
#ifndef A64_CSUM_VALU   // 0 = the unit-weight column sums over the matrix pipe (LDS transpose + 16x16x32 MFMAs against ones) instead of the VALU / DPP
#define A64_CSUM_VALU 1  // reduce-scatter: measured SLOWER (36.3 / 37.3 vs 35.9 ms at 6 heads; see finish_step), kept for tools/attn64_prof.py --mx
#endif
#ifndef A64_VSPREAD   // V^T fragment reads: 0 = one per gap in phase A gaps 0..15, 1 = every other gap 0..30
#define A64_VSPREAD 0
#endif
#ifndef A64_DMA_POS   // where the eight DMA pieces of a tile are issued: 0 = phase A gaps 0..7, 1 = phase B gaps 16..30 (even), 2 = phase A gaps
#define A64_DMA_POS 2  // 16..30 (even; after the V^T reads: 2 718 vs 2 798 cycles per tile for position 0, tools/attn64_cycles.py)
#endif

// the softmax pipeline's step table: elements exponentiated before step k (steps 0..19 = phase B gaps 12..31, 20..51 = the next
// phase A, 52.. = the phase B after it), the step that packs bf16 pair j, and the first read step of key unit u of the column sums
constexpr int a64_ecum(int k) {
    return k <= 0 ? 0 : k <= 20 ? (k * A64_EB) / 20 : k <= 52 ? A64_EB + ((k - 20) * A64_EA) / 32
                                                              : (A64_EB + A64_EA + (k - 52) < 64 ? A64_EB + A64_EA + (k - 52) : 64);
}
constexpr int a64_cstep(int j) {
    int k = 0;
    while (a64_ecum(k) < 2 * j + 2) ++k;
    return k;
}
constexpr int a64_rk(int u) { return u == 0 ? 36 : u == 1 ? 38 : u == 2 ? 68 : 70; }   // reads: phase A / B gaps 16..19
constexpr int a64_mk(int u) { return u == 0 ? 46 : u == 1 ? 48 : u == 2 ? 78 : 80; }   // MFMAs: gaps 26..29, ~450 cycles behind the reads

#ifdef ATTN64_PROF
// cycle anatomy (tools/attn64_prof.py builds a separate library with -DATTN64_PROF): s_memtime at the segment boundaries
// of every tile, summed per segment, for every wave of one mid-grid workgroup.  Not part of the product build.
__device__ unsigned long long g_a64_prof[4 * 8];
#define P64_DECL unsigned long long pt_ = 0, pacc_[7] = {0, 0, 0, 0, 0, 0, 0}; const bool prof_on_ = blockIdx.x == (gridDim.x / 2)
#define P64_START() do { if (prof_on_) pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define P64_MARK(i) do { if (prof_on_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } } while (0)
#define P64_END(w, n) do { if (prof_on_ && lane == 0) { for (int i_ = 0; i_ < 7; ++i_) g_a64_prof[(w) * 8 + i_] = pacc_[i_]; g_a64_prof[(w) * 8 + 7] = (n); } } while (0)
#else
#define P64_DECL
#define P64_START()
#define P64_MARK(i)
#define P64_END(w, n)
#endif

// S^T block (kb, qb) (+)= K fragment (kb, ks) . Q^T fragment (qb, ks); both operands live in the accumulator file
template <int KB, int QB, int KS>
__device__ __forceinline__ void mfma_qk(f32x16 &s) {
    constexpr int ka = 192 + (KB * 8 + KS) * 4, qa = 128 + (QB * 8 + KS) * 4;
    if constexpr (A64_ABL & 64) {   // ablation: the scores accumulate in accumulator registers (wrong, timing only)
        constexpr int oa = (KB * 2 + QB) * 16;
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[%c2:%c3], a[%c4:%c5], a[%c0:%c1]" ::"i"(oa), "i"(oa + 15), "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
        return;
    }
    if constexpr (KS == 0)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
}
// O^T block (qb, db) += V^T fragment . P^T fragment
template <int QB, int DB>
__device__ __forceinline__ void mfma_pv(const u32x4 &vf, const u32x4 &pf) {
    constexpr int oa = (QB * 4 + DB) * 16;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(oa), "i"(oa + 15));
}
// K fragment (kb, ks) of the tile in ring slot SLOT -> a[192 + ...]; completion is counted by the caller (lgkmcnt)
template <int KB, int KS, int SLOT>
__device__ __forceinline__ void lds_k(uint32_t addr) {
    constexpr int ka = 192 + (KB * 8 + KS) * 4;
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(ka), "i"(ka + 3), "i"(SLOT * TB + KB * 8192) : "memory");
}
// ---- cross-lane reduce-scatter steps of the fused column sums (MODE 3).  Two registers a, b hold values of two different
// keys, one per lane (= query); after the step `a` holds, in the lanes whose bit B is 0, key a summed over the lane pair
// (lane, partner) and, in the lanes whose bit B is 1, key b likewise: half the registers per step, no selects.
//   bit 4: v_permlane16_swap (odd rows of a <-> even rows of b), then one add        (inputs written >= 2 instructions ago)
//   bit 3: two bank-masked DPP adds, partner = row_mirror (15 - i: the bits below are flipped too, which the later steps
//          sum over anyway);   bit 2: the same with row_half_mirror (7 - i), banks 0 / 2 keep a, banks 1 / 3 keep b
//   bits 1, 0: inside a quad there is no lane mask: two selects and one DPP add (quad_perm)
// (two pairs per statement: each add sits one instruction behind its swap, which is the wait state the swap's result needs)
__device__ __forceinline__ void cs_pair16x2(float &a, float &b, float &c, float &d) {
    asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void cs_pair8(float &a, const float &b) {
    asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(a) : "v"(b));
}
__device__ __forceinline__ void cs_pair4(float &a, const float &b) {
    asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(a) : "v"(b));
}
template <int CTRL>
__device__ __forceinline__ float cs_pair_quad(float a, float b, bool hi) {
    const float keep = hi ? b : a, give = hi ? a : b;
    return keep + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(give), CTRL, 0xf, 0xf, true));
}
// Largest row norm of K per (batch, head): one 16-lane group per row (a lane squares its 8 elements, DPP row sum), maximum
// over the block, one atomicMax on the float's bits (non-negative floats order like unsigned integers; the buffer is
// zeroed by a memset node before every launch).  730 MB of K at HunyuanVideo size: ~0.2 ms beside a 130 ms launch.
__global__ __launch_bounds__(256) void knorm_max_kernel(const uint16_t *k, const int64_t ks0, const int64_t ks1, const int64_t ks2, int H, int Nk,
                                                        float *out) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const uint16_t *kb = k + b * ks0 + h * ks1;
    const int grp = threadIdx.x >> 4, li = threadIdx.x & 15;
    float best = 0.f;
#pragma unroll 4
    for (int row = blockIdx.x * 16 + grp; row < Nk; row += gridDim.x * 16) {
        const u32x4 v = *(const u32x4 *)(kb + (int64_t)row * ks2 + li * 8);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
            ss = __builtin_fmaf(lo, lo, ss);
            ss = __builtin_fmaf(hi, hi, ss);
        }
        best = fmaxf(best, row16_sum(ss));
    }
    __shared__ float red[256];
    red[threadIdx.x] = best;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((unsigned int *)(out + bh), __float_as_uint(__builtin_sqrtf(red[0])));
}

// MODE 0: dense -- 256-row workgroups, every wave computes 64 rows and stages its quarter of each tile.
// MODE 1 / 2: gathered (csp_128_attn / the accumulate forms csp_attn, csp_attn_out) -- one 192-row query group per
//   workgroup: waves 0..2 compute 64 rows each, wave 3 is the LOADER: it reads the group's index list, forms the per-lane
//   row offsets and issues every LDS-DMA piece of every tile (32 per tile, ~40 cycles of issue each: a quarter of a
//   compute wave's tile time when each wave stages its own share).  Three SIMDs of MFMA work out of four measured 90 % of
//   the four-wave dense kernel's throughput on dense data (the compute waves' tiles get shorter, the chip clocks higher).
//   Work items come from the device-built plan of attn.hip (longest-first order, heavy items sliced over their key tiles
//   and merged by the last arriver).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64_kernel(const AttnParams p) {
    constexpr bool GATHER = MODE == 1 || MODE == 2, INPLACE = MODE == 2;
    constexpr bool CSUM = MODE == 3;   // dense + the column sums of dense_colsum_attn in the same pass (see below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5, l15 = lane & 15, lg = lane >> 4;
    // sp / nsp: this workgroup's slice of the item's key tiles; slot0: scratch slot of the item's slice 0; tix: its ticket
    int wid = blockIdx.x, sp = 0, nsp = 1, slot0 = 0, tix = 0;
    if constexpr (GATHER) {
        const int wid0 = wid;
        wid = p.plan[2 * wid0];
        if (wid < 0) return;
        const int meta = p.plan[2 * wid0 + 1];
        sp = meta & 0xff, nsp = (meta >> 8) & 0xff;
        slot0 = tix = meta >> 16;
    }
    const int bh = wid / p.G, g = wid - bh * p.G;
    const int b = bh / p.H, h = bh - b * p.H;
    const int row0 = g * (GATHER ? 192 : WGROWS) + w * WROWS;
    const int count = GATHER ? p.counts[(int64_t)bh * p.G + g] : p.Nk;
    const int valid = count < p.Nk ? count : p.Nk;   // packed positions >= Nk are masked out (csp_128_attn.cu:314)
    const int ntiles_all = (valid + KT - 1) / KT;
    const int tbeg = (int)((int64_t)ntiles_all * sp / nsp), tend = (int)((int64_t)ntiles_all * (sp + 1) / nsp);
    const int ntiles = tend - tbeg;                  // tiles of this workgroup; local tile t = tile tbeg + t of the item
    const int T4 = (ntiles + 3) & ~3;

    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const uint16_t *vbase = p.v + b * p.vs[0] + h * p.vs[1];
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase), vrsrc = make_rsrc(vbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u, vstride_b = (uint32_t)p.vs[2] * 2u;

    if (GATHER && w == 3) {
        // ---- the loader wave.  LDS row 4*pc + lg of a tile (piece pc, lane group lg) holds packed position lg*16 + pc of the
        // tile: lane group lg reads 16 CONSECUTIVE indices into 16 registers and piece pc takes register pc -- no cross-lane
        // movement (the order of the keys inside a tile is free as long as K, V and the tail mask agree on it).
        // Indices of four tiles live in a register ring (entry = tile mod 4); iteration t loads tile t+5's, issues K(t+4)
        // into the slot K(t) left and V(t+2) into the slot of V(t-2).  vmcnt: an iteration is 4 index loads + 32 pieces (16 +
        // 32 on the unaligned path); at the top of iteration t everything up to iteration t-3 has to have landed -- 72
        // younger operations, more than the counter can express: vmcnt(63), the loosest wait there is, covers it (a single
        // wave cannot keep more than 63 KiB of gathers in flight; key*stride as a 24-bit multiply, checked by the host).
        const IndexRow irow = index_row(p, (int64_t)bh * p.G + g);
        const int32_t *idx = irow.ptr;
        uint32_t kswz[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kswz[i] = (uint32_t)(l15 ^ (4 * i + lg)) << 4;
        const uint32_t vswz = (uint32_t)(l15 ^ (lg << 2)) << 4;
        int ir[4][16];
        // 16 consecutive indices per lane group: four 16-byte loads when the row is 16-byte aligned and the tile lies inside
        // the list (every HunyuanVideo / Wan / FLUX launch), dword loads with clamped positions otherwise
        const bool idx_vec = ((uintptr_t)idx & 15) == 0 && (irow.width & 3) == 0;
        auto load_idx = [&](int T, int (&dst)[16]) {
            const int base = (tbeg + T) * KT + lg * 16;
            if (idx_vec && (tbeg + T) * KT + KT <= irow.width) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const u32x4 v4 = *(const u32x4 *)(idx + base + j4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[j4 * 4 + e] = (int)v4[e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int pos = base + j;
                    pos = pos < irow.width ? pos : irow.width - 1;
                    dst[j] = idx[pos];
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) dst[j] = max(0, min(dst[j], p.Nk - 1));   // memory safety for malformed indices
        };
        auto all_k = [&](const int (&src)[16], int slot) {
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) blds16(krsrc, __umul24((uint32_t)src[pc], kstride_b) + kswz[pc & 3], 0, smem + slot * TB + pc * 1024);
        };
        auto all_v = [&](const int (&src)[16], int slot) {
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) blds16(vrsrc, __umul24((uint32_t)src[pc], vstride_b) + vswz, 0, smem + VRING + slot * TB + pc * 1024);
        };
        load_idx(0, ir[0]), load_idx(1, ir[1]), load_idx(2, ir[2]), load_idx(3, ir[3]);
        all_k(ir[0], 0);
        all_k(ir[1], 1), all_v(ir[0], 3);
        all_k(ir[2], 2), all_v(ir[0], 0);
        all_k(ir[3], 3), all_v(ir[1], 1);
        load_idx(4, ir[0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto loader_tile = [&](auto slc, int t) __attribute__((always_inline)) {
            constexpr int SL = decltype(slc)::value;
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_idx(t + 5, ir[(SL + 1) & 3]);
            all_k(ir[SL], SL);
            all_v(ir[(SL + 2) & 3], (SL + 2) & 3);
        };
        for (int tb = 0;; tb += 4) {
            loader_tile(ic<0>{}, tb);
            if (tb >= T4) break;
            loader_tile(ic<1>{}, tb + 1);
            loader_tile(ic<2>{}, tb + 2);
            loader_tile(ic<3>{}, tb + 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (nsp > 1) {   // the barriers of the slice hand-off below
            __syncthreads();
            __syncthreads();
            if (*(volatile int *)(smem) != nsp - 1) return;
            __syncthreads();
        }
        return;
    }

    // ---- accumulator file: O^T = 0, Q^T fragments (B operand: lane = query l31, d = ks*16 + hf*8 .. +7)
    asm volatile(A64_ZERO_O ::: A64_CLOBBER_ALL);
    float qss[2] = {0.f, 0.f};   // this lane's half of |q|^2 per query block
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1];
        static_for<0, 16>([&](auto f) {
            constexpr int F = decltype(f)::value, qb = F >> 3, ks = F & 7;
            const int qrow = row0 + qb * 32 + l31;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (qrow < p.Nq) val = *(const u32x4 *)(qp + (int64_t)qrow * p.qs[2] + ks * 16 + hf * 8);
            acc_write4<128 + F * 4>(val);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(val[e] << 16), hi = __uint_as_float(val[e] & 0xffff0000u);
                qss[qb] = __builtin_fmaf(lo, lo, qss[qb]);
                qss[qb] = __builtin_fmaf(hi, hi, qss[qb]);
            }
        });
    }

    // ---- lane-constant LDS addresses.  K tile: row-major [64][256 B], 16-byte chunk c of row r stored at c ^ (r & 15);
    //      V tile: chunk c of row r stored at c ^ ((r & 3) << 2).  Both swizzles are applied on the DMA source side.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    uint32_t kad[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kad[ks] = lds0 + l31 * 256 + (((2 * ks + hf) ^ l15) << 4);
    uint32_t vad[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
        vad[db] = lds0 + VRING + (4 * hf + (l15 >> 2)) * 256 + (((db * 4 + 2 * (lg & 1) + ((l15 & 3) >> 1)) ^ ((l15 >> 2) << 2)) << 4) +
                  (l15 & 1) * 8;
    // DMA: wave w stages pieces w*4 + i (4 rows of 256 B each), i = 0..3, of the K tile and of the V tile
    uint32_t kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kofs[i] = (uint32_t)(w * 16 + 4 * i + lg) * kstride_b + ((uint32_t)(l15 ^ (4 * i + lg)) << 4);
    const uint32_t vofs = (uint32_t)(w * 16 + lg) * vstride_b + ((uint32_t)(l15 ^ (lg << 2)) << 4);

    // One DMA piece (4 rows) of tile T.  A ragged last tile and the padding tiles are fetched from the last 64 rows of
    // the key range instead (rows [Nk-64, Nk): in bounds, wave-uniform), which shifts the tile's live keys to the END of
    // its LDS rows; the dead rows in front hold keys of the previous tile and are masked in the scores.
    const int last_base = p.Nk - KT;
    auto tile_base = [&](int T) { return T * KT < last_base ? T * KT : last_base; };
    auto issue_k1 = [&](uint32_t soff, uint32_t ldsw, int slot, int i) {
        blds16(krsrc, kofs[i], soff, smem + ldsw + slot * TB + i * 1024);
    };
    auto issue_v1 = [&](uint32_t soff, uint32_t ldsw, int slot, int i) {
        blds16(vrsrc, vofs, soff + (uint32_t)(4 * i) * vstride_b, smem + ldsw + VRING + slot * TB + i * 1024);
    };
    auto issue_k = [&](int T, int slot) {
        const uint32_t soff = (uint32_t)tile_base(T) * kstride_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_k1(soff, w * 4096, slot, i);
    };
    auto issue_v = [&](int T, int slot) {
        const uint32_t soff = (uint32_t)tile_base(T) * vstride_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_v1(soff, w * 4096, slot, i);
    };

    // ---- prologue: K(0), then the issues of "iterations" -3..-1 (iteration i issues K(i+4) and V(i+2); V(-1) does not
    //      exist: V(0) goes into its slot so that the first tile's PV -- P = 0 -- multiplies finite numbers)
    if constexpr (!GATHER) {
        issue_k(0, 0);
        issue_k(1, 1), issue_v(0, 3);
        issue_k(2, 2), issue_v(0, 0);
        issue_k(3, 3), issue_v(1, 1);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    __syncthreads();
    static_for<0, 16>([&](auto i) {
        constexpr int I = decltype(i)::value;
        lds_k<(I >> 3), (I & 7), 0>(kad[I & 7]);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x16 s[4];        // S^T(t), block kb*2 + qb
    float px[64];       // x = s*c - m*c, then p = exp2(x), of the tile being finished; e = blk*16 + r
    uint32_t pw[2][4][4] = {};   // P^T fragments (qb, key slab u', dword); zero: the first tile's PV multiplies P(-1) = 0
    float m[2] = {-INFINITY, -INFINITY}, nmsc[2] = {0.f, 0.f};
    float mlag[2] = {-INFINITY, -INFINITY};   // m + the lag (in raw score units): the per-tile check is one compare per query block
    float lacc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float alpha[2] = {1.f, 1.f};
    // CSUM: w = exp2(m c) p_i (the summand exp2(s c + log2 p_i) = P_ij w_i), lpq = log2 p_i, the reduction's last nodes,
    // this lane's byte offset inside a tile's 64 partial sums and the wave's row of the partial-sum buffer
    float wq[2] = {0.f, 0.f}, lpq[2] = {-1.0e30f, -1.0e30f}, c4[2] = {0.f, 0.f}, c5 = 0.f;
    uint32_t csoff = 0, ex_w = 0, ex_r = 0;
    int cs_extra = 0;
    float cs_own = 0.f;      // this wave's sums of the previous tile (its partners' are in the exchange area)
    __amdgpu_buffer_rsrc_t prsrc = krsrc;
    // unit-weight loop: the column sums go over the matrix pipe (below).  pst_w / pst_r: this lane's write / read address for
    // key unit 0 inside the wave's P stage, csf: the fragment reads in flight, csacc / csout: one unit's sums / the tile's
    uint32_t pst_w = 0, pst_r = 0, csoff_mx = 0;
    u32x2 csf[8] = {};
    f32x4 csacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float csout = 0.f;
    u32x4 cs_ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if constexpr (CSUM) {
        static_assert(A64_CSUM_VALU, "the matrix-pipe column-sum probe (A64_CSUM_VALU=0) fills all 160 KiB of LDS with its P stage: no room for the exchange area of the in-workgroup combine; build round 5's tree for it");
        pin(cs_ones);
        const uint32_t pst = lds0 + LDS_BYTES + (uint32_t)w * PSTAGE;
        pst_w = pst + l31 * 128 + ((uint32_t)(hf ^ (l31 & 7)) << 4);
        const int prow = lg * 4 + (l15 >> 2);
        pst_r = pst + prow * 128 + ((uint32_t)(((l15 & 3) >> 1) ^ (prow & 7)) << 4) + (l15 & 1) * 8;
        csoff_mx = 2u * (uint32_t)(16 * lg + 8 * ((l15 >> 2) & 1) + 4 * (l15 >> 3) + (l15 & 3));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = row0 + qb * 32 + l31;
            const float pl = qrow < p.Nq ? p.p_in[(int64_t)bh * p.Nq + qrow] : 0.f;
            lpq[qb] = pl > 0.f ? __builtin_amdgcn_logf(pl) : -1.0e30f;   // (rows past Nq and p <= 0 contribute exp2(-huge) = 0)
        }
        // the lane that ends up with key row (kb, u, rr, hf) of the tile: kb = bit 0, u = bit 1, rr = bits 4 3 2 (lsb first)
        const int rr = ((lane >> 4) & 1) | (((lane >> 3) & 1) << 1) | (((lane >> 2) & 1) << 2);
        csoff = 2u * (uint32_t)(32 * (lane & 1) + (rr & 3) + 8 * (2 * ((lane >> 1) & 1) + (rr >> 2)) + 4 * hf);
        // In-workgroup combine (round 6): a 192-row group is three 64-row waves and a workgroup four, so a workgroup's waves belong to
        // TWO groups -- the first nA = 3 - (4g mod 3) waves to one, the rest to the next.  Each wave drops its tile's 64 sums into a 2 KiB
        // exchange area behind the rings (one ds_write_b32 per lane and tile), and one tile later -- behind the tile's barrier -- the first
        // wave of each set adds its partners' (fp32, wave order) and stores ONE bf16 row per set: 2 partial rows per workgroup instead of
        // 4 (5.3 GB instead of 10.6 at HunyuanVideo size, written here and read by the mask kernel), rounded once per set.
        const int nA = 3 - (4 * g) % 3;
        const int lead = w < nA ? 0 : nA, members = w < nA ? nA : 4 - nA;
        cs_extra = w == lead ? members : 0;       // 0: not a set's first wave; else the waves of the set (1..3)
        ex_w = lds0 + LDS_BYTES + (uint32_t)w * 256u + (uint32_t)lane * 4u;
        ex_r = lds0 + LDS_BYTES + (uint32_t)(lead + 1) * 256u + (uint32_t)lane * 4u;
        prsrc = make_rsrc(p.cs_part + ((int64_t)bh * (p.G * 2) + (g * 2 + (w < nA ? 0 : 1))) * p.cs_pstride);
    }
    // "tile -1": the first pass runs steps 20.. of the softmax pipeline on it -- elements 0..19 as if already exponentiated
#pragma unroll
    for (int e = 0; e < 64; ++e) px[e] = e < 20 ? 0.f : -INFINITY;

    // V^T fragment: keys {4hf + 0..3, 8 + 4hf + 0..3} of a 16-key slab, d = db*32 + l31.  Issued through asm: hipcc puts a
    // vmcnt(0) in front of every LDS read it knows about while an LDS-DMA is in flight (it cannot tell the slots apart),
    // which here would wait for the tiles just requested.  The caller waits (lgkmcnt) before the first use.
    auto vfrag_read = [&](auto dbc, auto offc) __attribute__((always_inline)) {
        constexpr int DB = decltype(dbc)::value, OFF = decltype(offc)::value;
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4"
                     : "=&v"(lo), "=&v"(hi) : "v"(vad[DB]), "i"(OFF), "i"(OFF + 2048) : "memory");   // (early clobber: two reads, one address)
        return (u32x4){lo[0], lo[1], hi[0], hi[1]};
    };

    // ---- the softmax of a tile as a pipeline of single-instruction stages over its 64 scores per lane, in SEQUENCE order
    //      i = kb*32 + u*16 + qb*8 + rr  (element r = 8u + rr of block kb*2 + qb): key slab u' = kb*2 + u is complete after
    //      every 16 elements, which is the order the PV MFMAs consume the P^T fragments in.
    //        X_i: px[i] = s*c - m*c          phase B(t), gaps 12..27, four per gap
    //        E_i: px[i] = exp2(px[i])        v_exp_f32 runs at a quarter of the VALU rate: 64 of them are half of a tile's
    //                                        VALU time, so they are spread one per gap over B(t) 12..31 (step k = 0..19),
    //                                        A(t+1) (k = 20..51, five per four gaps) and B(t+1) 0..3 (k = 52..55)
    //        L_i: row sum, one step behind E;  C_j: bf16 pair j = (2j, 2j+1), one step behind E of its second element.
    //      Slab deadlines (first PV MFMA that reads the fragment): slab 0 at B(t+1) gap 0 (packed by k = 17), slab 1 at gap 8
    //      (k = 31), slab 2 at gap 16 (k = 44), slab 3 at gap 24 (k = 57 = gap 5).
    static_assert(64 - A64_EB - A64_EA >= 0 && 64 - A64_EB - A64_EA <= 9, "the last steps must finish before gap 12 of the next PV phase");
    auto ecum = [](int k) constexpr { return a64_ecum(k); };
    auto finish_step = [&](auto kk, auto uwc) __attribute__((always_inline)) {
        constexpr int K = decltype(kk)::value;
        constexpr bool UW = decltype(uwc)::value != 0;   // CSUM with unit weights (reference point = the previous step's normaliser)
        if constexpr (!(A64_ABL & 1)) {
            static_for<ecum(K), ecum(K + 1)>([&](auto ii) {           // E
                constexpr int I = decltype(ii)::value;
                if constexpr (A64_ABL & 32) px[I] = px[I] * 0.5f;   // ablation: the price of v_exp_f32 itself
                else px[I] = __builtin_amdgcn_exp2f(px[I]);
                pin(px[I]);
            });
            static_for<ecum(K - 1), ecum(K)>([&](auto ii) {           // L
                constexpr int I = decltype(ii)::value;
                lacc[(I >> 3) & 1][I & 1] += px[I];
                pin(lacc[(I >> 3) & 1][I & 1]);
            });
            static_for<(ecum(K - 1) >> 1), (ecum(K) >> 1)>([&](auto jj) {   // C
                constexpr int J = decltype(jj)::value, I = 2 * J;
                pw[(I >> 3) & 1][I >> 4][(I & 7) >> 1] = pack_bf16x2(px[I], px[I + 1]);
                pin(pw[(I >> 3) & 1][I >> 4][(I & 7) >> 1]);
            });
            if constexpr (CSUM && UW && !A64_CSUM_VALU) {
                // Column sums on the matrix pipe (round 4).  With unit weights the summand IS the bf16 P^T the PV MFMAs consume, but
                // its fragments hold the queries across lanes and the keys in registers -- the contraction index of every MFMA
                // layout -- so the sum over queries took 102 VALU / DPP operations per tile (~400 of a tile's ~3 100 cycles; the
                // loop is issue-bound).  Instead: every finished fragment pw[qb][u] (query l31 + 32 qb, the 8 keys of 16-key unit u
                // this lane half owns) goes into the wave's [64 queries][128 B] stage as one ds_write_b128 (16-byte chunk
                // (u, hf) stored at chunk ^ (row & 7): conflict-free for the 8-lane groups of the store and the 32-lane groups of
                // the reads); per unit four ds_read_b64_tr_b16 bring back B fragments (n = key column l15, k = 8 queries) and two
                // v_mfma_f32_16x16x32_bf16 against an all-ones A sum the 64 queries: every row of D is the unit's 16 column sums,
                // lane group lg keeps unit lg.  8 stores + 16 reads + 8 MFMAs (128 cycles of the pipe) + 3 selects per tile.
                // The stage is wave-private: LDS executes a wave's operations in order, no barrier.  Schedule (steps K of the
                // tile's pipeline; A gaps 16.. and B gaps 16.. carry no other LDS reads, so the lgkmcnt(0) in front of a unit's
                // MFMAs, four gaps behind its reads, finds them landed):
                //   store of pw[qb][u]: the step after its last pair is packed;   units 0, 1: reads in phase A gaps 16..19, MFMAs in
                //   gaps 26..29;   units 2, 3: the same gaps of phase B;   selects three steps behind;   bf16 store in phase B gap 31.
                // MEASURED (round 4, tools/attn64_prof.py --colsum [--mx]): slower than the reduce-scatter it replaces -- 3 840 vs 3 791
                // cycles per tile in the marked build, 36.3-37.3 vs 35.9 ms per 6-head launch.  The loop has no idle resource to move
                // the work to: a ds_write_b128 costs ~33 cycles of the wave's issue time here (phase B gaps 0..11: +66 cycles for two),
                // the transpose reads ~20 each, and the 16-cycle MFMAs land on a pipe that is 82 % busy in these phases.  Not the default.
                static_assert(a64_cstep(8 * 1 + 7) + 1 < a64_rk(0) && a64_cstep(8 * 3 + 7) + 1 < a64_rk(2), "a unit is read after its second fragment is stored");
                static_for<0, 4>([&](auto uu) {
                    constexpr int U = decltype(uu)::value, S = (U & 1) * 4;
                    if constexpr (K == a64_mk(U) || K == a64_mk(U) + 1) {
                        constexpr int H2 = K - a64_mk(U);
                        if constexpr (H2 == 0 && (U & 1) == 0)
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(csf[0]), "+v"(csf[1]), "+v"(csf[2]), "+v"(csf[3]), "+v"(csf[4]), "+v"(csf[5]), "+v"(csf[6]), "+v"(csf[7]) :: "memory");
                        const u32x4 fr = {csf[S + 2 * H2][0], csf[S + 2 * H2][1], csf[S + 2 * H2 + 1][0], csf[S + 2 * H2 + 1][1]};
                        // (asm: the compiler would put the builtin's result into the accumulator file, which here is O^T's; the
                        // result is read two gaps later, the chained one a gap later: no software wait states needed)
                        if constexpr (H2 == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(csacc[U & 1]) : "v"(cs_ones), "v"(fr));
                        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(csacc[U & 1]) : "v"(cs_ones), "v"(fr));
                    }
                    if constexpr (K == a64_mk(U) + 3) {
                        if constexpr (U == 0) csout = csacc[0][0];
                        else csout = lg == U ? csacc[U & 1][0] : csout;
                        pin(csout);
                    }
                    if constexpr (K == a64_rk(U) || K == a64_rk(U) + 1) {
                        constexpr int H2 = K - a64_rk(U);
                        uint32_t ad = pst_r;
                        if constexpr (U != 0) ad ^= (uint32_t)(U << 5);
                        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4"
                                     : "=&v"(csf[S + 2 * H2]), "=&v"(csf[S + 2 * H2 + 1]) : "v"(ad), "i"(H2 * 4096), "i"(H2 * 4096 + 2048) : "memory");
                    }
                });
                static_for<0, 8>([&](auto ii) {
                    constexpr int U = decltype(ii)::value >> 1, QW = decltype(ii)::value & 1;
                    if constexpr (K == a64_cstep(8 * U + 4 * QW + 3) + 1) {
                        uint32_t ad = pst_w;
                        if constexpr (U != 0) ad ^= (uint32_t)(U << 5);
                        const u32x4 pf = {pw[QW][U][0], pw[QW][U][1], pw[QW][U][2], pw[QW][U][3]};
                        asm volatile("ds_write_b128 %0, %1 offset:%c2" ::"v"(ad), "v"(pf), "i"(QW * 4096) : "memory");
                    }
                });
            } else if constexpr (CSUM) {
                // column sums, in place in px once E / L / C are done with an element (three steps behind E): the weighted sum
                // over the two query blocks lands in the qb = 1 element of each key, then the reduce-scatter over the 32 lanes
                // of a half (a step per level), the four slab nodes (px[8], px[24], px[40], px[56]) -> c4 -> c5
                static_for<ecum(K - 3), ecum(K - 2)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr (UW) {
                        // (v_pk_add_f32 for the two keys of a pair halves this level's operations but returned wrong LOW halves in some
                        // tile slots -- compiler-emitted and asm alike, with plain adds in the same place the sums are right -- and the
                        // build with it was 3.6 % slower: round 4, not pursued)
                        if constexpr (((I >> 3) & 1) == 1) { px[I] += px[I - 8]; pin(px[I]); }
                    } else {
                        if constexpr (((I >> 3) & 1) == 0) px[I] *= wq[0];
                        else px[I] = __builtin_fmaf(px[I], wq[1], px[I - 8]);
                        pin(px[I]);
                    }
                });
                static_for<ecum(K - 4), ecum(K - 3)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr ((I & 11) == 11) cs_pair16x2(px[I - 3], px[I - 2], px[I - 1], px[I]);
                });
                static_for<ecum(K - 5), ecum(K - 4)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr ((I & 11) == 11) cs_pair8(px[I - 3], px[I - 1]);
                });
                static_for<ecum(K - 6), ecum(K - 5)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr ((I & 15) == 15) cs_pair4(px[I - 7], px[I - 3]);
                });
                static_for<ecum(K - 7), ecum(K - 6)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr ((I & 31) == 31) { c4[I >> 5] = cs_pair_quad<0x4E>(px[I - 23], px[I - 7], (lane & 2) != 0); pin(c4[I >> 5]); }
                });
                static_for<ecum(K - 8), ecum(K - 7)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    if constexpr (I == 63) { c5 = cs_pair_quad<0xB1>(c4[0], c4[1], (lane & 1) != 0); pin(c5); }
                });
            }
        }
    };

    P64_DECL;
    P64_START();
    // one tile: ring slot SL = t mod 4 (static), K(t) already in a[192:255]
    auto tile = [&](auto slc, auto nmc, int t) __attribute__((always_inline)) {
        constexpr int SL = decltype(slc)::value;
        constexpr bool NOMAX = decltype(nmc)::value != 0;   // fixed reference point (see below): no maxima, no update, no rescale
        constexpr auto uwc = ic<(CSUM && decltype(nmc)::value == 2) ? 1 : 0>{};
        constexpr int VSL = (SL + 3) & 3;        // slot of V(t-1)
        constexpr int KNSL = (SL + 1) & 3;       // slot of K(t+1)
        // K(t+1) and V(t-1) were issued three iterations ago: the issues of the last two iterations may stay in flight
        uint32_t ldsw = w * 4096, ksoff = 0, vsoff = 0;
        if constexpr (!(A64_ABL & 16)) {
            if constexpr (!(A64_ABL & 256) && !GATHER) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if constexpr (!(A64_ABL & 128)) __builtin_amdgcn_s_barrier();
            // (opaque copies: the 32 destination addresses of the four unrolled tiles are sums the loop recomputes with one
            // SALU op each instead of values the compiler hoists and then spills)
            if constexpr (!GATHER) {
                asm volatile("" : "+s"(ldsw));
                ksoff = (uint32_t)tile_base(t + 4) * kstride_b;
                vsoff = (uint32_t)tile_base(t + 2) * vstride_b;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        P64_MARK(0);

        u32x4 vf[16];
        // ================= phase A: S(t) = K(t).Q^T  ||  tile t-1: steps 20..51 of its softmax pipeline =================
        static_for<0, 32>([&](auto gg) {
            constexpr int G = decltype(gg)::value;
            constexpr int ks = G >> 2, kb = (G >> 1) & 1, qb = G & 1;
            mfma_qk<kb, qb, ks>(s[kb * 2 + qb]);
            if constexpr (!(A64_ABL & 16) && !GATHER && ((A64_DMA_POS == 0 && G < 8) || (A64_DMA_POS == 2 && G >= 16 && (G & 1) == 0))) {
                // the DMA of K(t+4) -> slot of K(t), V(t+2) -> slot of V(t-2): one piece per gap
                constexpr int PC = A64_DMA_POS == 0 ? G : (G - 16) >> 1;
                if constexpr (PC < 4) issue_k1(ksoff, ldsw, SL, PC);
                else issue_v1(vsoff, ldsw, (SL + 2) & 3, PC - 4);
            }
            finish_step(ic<20 + G>{}, uwc);
            if constexpr (A64_VSPREAD ? (G & 1) == 0 : G < 16) {   // all sixteen V^T fragments of tile t-1: the LDS pipe carries V in phase A and K
                constexpr int f = A64_VSPREAD ? G >> 1 : G;        // in phase B (each wave reads both tiles whole: 2 x 64 KiB per CU and tile = 1024 LDS cycles)
                if constexpr (A64_ABL & 4) vf[f] = (u32x4){(uint32_t)t, 1u, 2u, 3u};
                else vf[f] = vfrag_read(ic<(f & 3)>{}, ic<VSL * TB + (f >> 2) * 4096>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        P64_MARK(1);
        // the sixteen fragments have landed (the statement names them: nothing that reads one is scheduled above it)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]), "+v"(vf[7]), "+v"(vf[8]),
                       "+v"(vf[9]), "+v"(vf[10]), "+v"(vf[11]), "+v"(vf[12]), "+v"(vf[13]), "+v"(vf[14]), "+v"(vf[15]));
        __builtin_amdgcn_sched_barrier(0);
        P64_MARK(2);

        // ======= phase B: O += V(t-1).P(t-1)  ||  tile t-1: steps 52..57; tile t: maxima, X, steps 0..19; K(t+1) -> a[192:255] ====
        float mx[2];
        bool moved = false;
        // dense: the first `dead` LDS rows of a ragged / padding tile are not this tile's keys (see tile_base; 0 for a full
        // tile).  gathered: LDS row 4*pc + lg holds packed position lg*16 + pc; positions >= vleft do not exist.
        const int dead = GATHER ? 0 : (t * KT - tile_base(t) < KT ? t * KT - tile_base(t) : KT);
        const int vleft = !GATHER ? KT : (t < ntiles ? valid - (tbeg + t) * KT : 0);
        auto phase_b_gap = [&](auto gg) __attribute__((always_inline)) {
            constexpr int G = decltype(gg)::value;
            constexpr int up = G >> 3, db = (G >> 1) & 3, qb = G & 1;
            const u32x4 pf = {pw[qb][up][0], pw[qb][up][1], pw[qb][up][2], pw[qb][up][3]};
            mfma_pv<qb, db>(vf[G >> 1], pf);
            constexpr bool UWT = decltype(uwc)::value != 0 && !A64_CSUM_VALU;
            if constexpr (G < (CSUM ? (UWT ? 32 : 17) : 12)) finish_step(ic<52 + G>{}, uwc);
            if constexpr (CSUM && G == (UWT ? 31 : 17)) {
                // tile t-1's 64 column sums over this wave's 64 queries go to the exchange area (slot parity of the tile); the set's
                // first wave adds tile t-2's -- its own from last tile, its partners' from the exchange area, complete since this tile's
                // barrier -- and stores them: one dword per lane into the set's partial row (a ragged last tile stores only its own keys,
                // a padding tile nothing).  LDS through asm: a read the compiler knows about waits for every DMA in flight.
                const uint32_t par2 = (uint32_t)(t & 1) * 1024u;            // parity of tile t-2
                const float mine = UWT ? csout : c5;
                if (!cs_extra) {
                    asm volatile("ds_write_b32 %0, %1" ::"v"(ex_w + (par2 ^ 1024u)), "v"(mine) : "memory");
                } else {
                    // (reads, write and wait on ONE path: no edge leaves a read in flight -- tools/audit_async_lds.py checks that)
                    float x1 = 0.f, x2 = 0.f;
                    if (cs_extra >= 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x1) : "v"(ex_r + par2) : "memory");
                    if (cs_extra >= 3) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(x2) : "v"(ex_r + par2) : "memory");
                    asm volatile("ds_write_b32 %0, %1" ::"v"(ex_w + (par2 ^ 1024u)), "v"(mine) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(x1), "+v"(x2)::"memory");   // (in order: everything but the write has returned)
                    const int tb2 = tile_base(t - 2), dd = (t - 2) * KT - tb2;
                    if (t > 1 && dd < KT) {
                        const uint32_t off = UWT ? csoff_mx : csoff;
                        if (dd <= 0 || (int)(off >> 1) >= dd) {
                            const uint16_t val = (uint16_t)(pack_bf16x2((cs_own + x1) + x2, 0.f) & 0xffffu);
                            if (p.probe & 16) __builtin_amdgcn_raw_buffer_store_b16(val, prsrc, off, (uint32_t)tb2 * 2u, 2);
                            else __builtin_amdgcn_raw_buffer_store_b16(val, prsrc, off, (uint32_t)tb2 * 2u, 0);
                        }
                    }
                }
                cs_own = mine;
            }
            if constexpr (!(A64_ABL & 16) && !GATHER && A64_DMA_POS == 1 && G >= 16 && (G & 1) == 0) {
                constexpr int PC = (G - 16) >> 1;
                if constexpr (PC < 4) issue_k1(ksoff, ldsw, SL, PC);
                else issue_v1(vsoff, ldsw, (SL + 2) & 3, PC - 4);
            }
            if constexpr (G < 16 && !(A64_ABL & 8)) {    // K(t+1) fragments, one per gap in the first half (nothing in this phase waits
                constexpr int J = G;                      // on LDS; the last one is 16 gaps old at the end-of-phase wait)
                lds_k<(J >> 3), (J & 7), KNSL>(kad[J & 7]);
            }
            if constexpr (G == 1 && !(A64_ABL & 2)) {   // ragged / padding tile: its first `dead` LDS rows are not this tile's keys
                if (!GATHER && __builtin_expect(dead > 0, 0)) {              // (wave-uniform, last tiles only)
                    int thr = dead - 4 * hf;            // opaque: the 32 per-register row numbers are compared as immediates
                    asm volatile("" : "+v"(thr));
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((blk >> 1) * 32 + (r & 3) + 8 * (r >> 2) < thr) s[blk][r] = -INFINITY;
                }
                if (GATHER && __builtin_expect(vleft < KT, 0)) {             // element (blk, r, hf) = LDS row (blk>>1)*32 + (r&3) + 8*(r>>2) + 4*hf
                    int thr = vleft - hf;               // = position (r&3)*16 + (blk>>1)*8 + 2*(r>>2) + hf
                    asm volatile("" : "+v"(thr));
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r & 3) * 16 + (blk >> 1) * 8 + 2 * (r >> 2) >= thr) s[blk][r] = -INFINITY;
                }
            }
            if constexpr (G >= 2 && G <= 9 && !(A64_ABL & 2) && !NOMAX) {   // maxima of the 32 scores a lane holds per query block: 16 x v_max3 each, two per
                constexpr int j = G - 2;        // block per gap, the four of a gap in one statement (one boundary pad, not four)
                constexpr int kb2 = j >> 2, r0 = (j & 3) * 4;
                if constexpr (j == 0)
                    asm volatile("v_max_f32 %0, %2, %3\n\tv_max_f32 %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
                                 : "=&v"(mx[0]), "=&v"(mx[1])
                                 : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]));
                else
                asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
                             : "+v"(mx[0]), "+v"(mx[1])
                             : "v"(s[kb2 * 2][r0]), "v"(s[kb2 * 2][r0 + 1]), "v"(s[kb2 * 2][r0 + 2]), "v"(s[kb2 * 2][r0 + 3]),
                               "v"(s[kb2 * 2 + 1][r0]), "v"(s[kb2 * 2 + 1][r0 + 1]), "v"(s[kb2 * 2 + 1][r0 + 2]), "v"(s[kb2 * 2 + 1][r0 + 3]));
            }
            if constexpr (G == 10 && !(A64_ABL & 2) && !NOMAX) {  // the other half of the keys lives in lane ^ 32
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    float a = mx[q2], c = mx[q2];
                    lane_swap32(a, c);
                    mx[q2] = max2(a, c);
                    pin(mx[q2]);
                }
            }
            if constexpr (G >= 12 && G < 28 && !(A64_ABL & 2)) {   // X: four elements per gap, sequence order
                static_for<(G - 12) * 4, (G - 11) * 4>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    constexpr int blk = (I >> 5) * 2 + ((I >> 3) & 1), r = ((I >> 4) & 1) * 8 + (I & 7);
                    px[I] = __builtin_fmaf(s[blk][r], SCALE_LOG2E, nmsc[(I >> 3) & 1]);
                    pin(px[I]);
                });
            }
            if constexpr (G >= 12) finish_step(ic<G - 12>{}, uwc);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 12>(phase_b_gap);
        P64_MARK(3);
        if constexpr (!(A64_ABL & 2) && !NOMAX) {   // the (rare) move of the reference point
            constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;
            if (__builtin_amdgcn_ballot_w64(mx[0] > mlag[0]) | __builtin_amdgcn_ballot_w64(mx[1] > mlag[1])) {
                moved = true;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const float m_new = max2(m[q2], mx[q2]);
                    alpha[q2] = __builtin_amdgcn_exp2f((m[q2] - m_new) * SCALE_LOG2E);
                    lacc[q2][0] *= alpha[q2];
                    lacc[q2][1] *= alpha[q2];
                    m[q2] = m_new;
                    nmsc[q2] = -m_new * SCALE_LOG2E;
                    mlag[q2] = m_new + LAG_RAW;
                    if constexpr (CSUM) wq[q2] = fminf(__builtin_amdgcn_exp2f(m_new * SCALE_LOG2E + lpq[q2]), 1.0e37f);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        P64_MARK(4);
        static_for<12, 32>(phase_b_gap);
        P64_MARK(5);
        if (__builtin_expect(moved, 0)) {   // (cold: ~18 KiB of register-by-register code leaves the loop's I-cache footprint) O_t = alpha (O_{t-1} + P_{t-1} V_{t-1}): after the pending PV, before the next one
            float tmp;
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" A64_SCALE_QB0 : "=&v"(tmp) : "v"(alpha[0]));
            asm volatile(A64_SCALE_QB1 : "=&v"(tmp) : "v"(alpha[1]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K(t+1) fragments have landed
        __builtin_amdgcn_sched_barrier(0);
        P64_MARK(6);
    };

    // tiles 0 .. T4-1 (multiples of four: the padding tiles are fully masked), then one more pass whose phase A finishes
    // tile T4-1 and whose phase B accumulates it (its own -- masked -- tile is never finished)
    // Fixed reference point.  |s_ij| <= |q_i| * max_j |k_j| =: M_i (Cauchy-Schwarz; kmax from knorm_max_kernel).  With the
    // exponentials taken against M_i from the first tile on, p = exp2((s - M) c) lies in [2^(-2 M c), 1]: if 2 M c <= 64 for
    // every query of the wave that is inside the normal range of fp32 and bf16, the softmax is the same up to rounding
    // (floating point is scale invariant; o = acc / l and l_out = 1 / (2^(M c) l) do not care which reference was used),
    // and the 32 v_max3, the lane-half exchange, the check and the rescale path leave the loop (46 of ~350 issues per
    // tile).  Unit-variance q, k at HunyuanVideo size: 2 M c ~ 42.  Waves that cannot prove the bound (large-norm inputs)
    // run the running-maximum loop; the choice is per wave and changes nothing but rounding.
    bool nomax = false;
    if constexpr (!GATHER) {
        if (p.kmax) {
            const float km = p.kmax[bh];
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float a = qss[qb], c2 = qss[qb];
                lane_swap32(a, c2);
                const float mq = __builtin_sqrtf(a + c2) * km;   // M_i in raw score units
                ok = ok && (2.0f * mq * SCALE_LOG2E <= 64.0f);
                qss[qb] = mq;
            }
            nomax = __builtin_amdgcn_ballot_w64(!ok) == 0;
            if (nomax) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    m[qb] = qss[qb], nmsc[qb] = -qss[qb] * SCALE_LOG2E, mlag[qb] = INFINITY;
                    if constexpr (CSUM) wq[qb] = fminf(__builtin_amdgcn_exp2f(qss[qb] * SCALE_LOG2E + lpq[qb]), 1.0e37f);
                }
            }
        }
    }
    // CSUM: with the reference point of query i at -log2 p_i (the previous step's normaliser) the summand of the column sums
    // IS the P of the softmax pipeline (w = 1: the weighting costs one add per key instead of a multiply and a
    // multiply-add).  Taken when every query of the wave has a usable p and |s c| + |log2 p| stays far inside the exponent
    // range; p.probe & 4 keeps the weighted form (A/B, tests).
    bool unitw = false;
    if constexpr (CSUM) {
        if (nomax && !(p.probe & 4)) {
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) ok = ok && lpq[qb] > -1.0e29f && qss[qb] * SCALE_LOG2E + __builtin_fabsf(lpq[qb]) <= 80.0f;
            unitw = __builtin_amdgcn_ballot_w64(!ok) == 0;
            if (unitw) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) m[qb] = -lpq[qb] / SCALE_LOG2E, nmsc[qb] = lpq[qb];
            }
        }
    }
    if (CSUM && unitw) {
        if constexpr (CSUM) {
            for (int tb = 0;; tb += 4) {
                tile(ic<0>{}, ic<2>{}, tb);
                if (tb >= T4) break;
                tile(ic<1>{}, ic<2>{}, tb + 1);
                tile(ic<2>{}, ic<2>{}, tb + 2);
                tile(ic<3>{}, ic<2>{}, tb + 3);
            }
        }
    } else if (nomax) {
        for (int tb = 0;; tb += 4) {
            tile(ic<0>{}, ic<1>{}, tb);
            if (tb >= T4) break;
            tile(ic<1>{}, ic<1>{}, tb + 1);
            tile(ic<2>{}, ic<1>{}, tb + 2);
            tile(ic<3>{}, ic<1>{}, tb + 3);
        }
    } else {
        for (int tb = 0;; tb += 4) {
            tile(ic<0>{}, ic<0>{}, tb);
            if (tb >= T4) break;
            tile(ic<1>{}, ic<0>{}, tb + 1);
            tile(ic<2>{}, ic<0>{}, tb + 2);
            tile(ic<3>{}, ic<0>{}, tb + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P64_END(w, T4 + 1);
    if constexpr (CSUM) {
        // the last pass left tile T4-1's sums in cs_own / the exchange area: combine and store them (a padding tile: nothing)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (cs_extra) {
            float x1 = 0.f, x2 = 0.f;
            const uint32_t par = (uint32_t)((T4 - 1) & 1) * 1024u;
            if (cs_extra >= 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x1) : "v"(ex_r + par) : "memory");
            if (cs_extra >= 3) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(x2) : "v"(ex_r + par) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x1), "+v"(x2)::"memory");
            const int tb1 = tile_base(T4 - 1), dd = (T4 - 1) * KT - tb1;
            const uint32_t off = (unitw && !A64_CSUM_VALU) ? csoff_mx : csoff;
            if (dd < KT && (dd <= 0 || (int)(off >> 1) >= dd))
                __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(pack_bf16x2((cs_own + x1) + x2, 0.f) & 0xffffu), prsrc, off, (uint32_t)tb1 * 2u, 0);
        }
    }

    // ---- the accumulators leave the accumulator file: o[(qb*4 + db)*16 + r] = O^T element r of block (qb, db); the row
    //      sums become whole (both lane halves hold the sum over all keys of the query)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float o[128];
    static_for<0, 32>([&](auto ii) {
        constexpr int I = decltype(ii)::value;
        const f32x4 o4 = acc_read4<I * 4>();
        o[I * 4 + 0] = o4[0], o[I * 4 + 1] = o4[1], o[I * 4 + 2] = o4[2], o[I * 4 + 3] = o4[3];
    });
    float lq[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float lp = lacc[qb][0] + lacc[qb][1], lo2 = lp;
        lane_swap32(lp, lo2);
        lq[qb] = lp + lo2;
    }

    if constexpr (GATHER) {
        if (nsp > 1) {
            // ---- key-sliced item: publish this slice's (o, m, l) -- [132 values][192 compute threads], coalesced -- take a
            //      ticket; the last arriver folds ALL slices in slice order (the lane layout is the same in every slice:
            //      the merge is the online-softmax rescale element by element) and alone runs the epilogue.  Same hand-off as
            //      attn.hip: plain stores -> barrier -> one agent-scope release -> drained -> relaxed ticket; last arriver:
            //      ticket -> one agent-scope acquire -> barrier -> plain loads.  The loader wave only takes the barriers.
            constexpr int SLICE_FLOATS = 26 * 256 * 4;   // the scratch launch_attn reserves per slice
            int *ticket_s = (int *)smem;
            float *mine = p.ws + (int64_t)(slot0 + sp) * SLICE_FLOATS + tid;
#pragma unroll
            for (int i = 0; i < 128; ++i) mine[i * 192] = o[i];
            mine[128 * 192] = m[0], mine[129 * 192] = m[1], mine[130 * 192] = lq[0], mine[131 * 192] = lq[1];
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
#pragma unroll
            for (int i = 0; i < 128; ++i) o[i] = 0.f;
            m[0] = m[1] = -INFINITY, lq[0] = lq[1] = 0.f;
            for (int s2 = 0; s2 < nsp; ++s2) {
                const float *oth = p.ws + (int64_t)(slot0 + s2) * SLICE_FLOATS + tid;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const float ms = oth[(128 + qb) * 192], ls = oth[(130 + qb) * 192];
                    const float m_new = fmaxf(m[qb], ms);
                    if (m_new == -INFINITY) continue;  // nothing so far and an empty slice
                    const float a = __builtin_amdgcn_exp2f((m[qb] - m_new) * SCALE_LOG2E);
                    const float c = __builtin_amdgcn_exp2f((ms - m_new) * SCALE_LOG2E);
                    m[qb] = m_new;
                    lq[qb] = lq[qb] * a + ls * c;
#pragma unroll
                    for (int i = 0; i < 64; ++i) o[qb * 64 + i] = o[qb * 64 + i] * a + oth[(qb * 64 + i) * 192] * c;
                }
            }
        }
    }

    // ---- epilogue: O = O^T / l; a lane holds, per (qb, db), four groups of 4 consecutive d of query row qb*32 + l31
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = lq[qb];
        const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
        const int qrow = row0 + qb * 32 + l31;
        if (qrow >= p.Nq) continue;
        const int64_t ooff = b * p.os[0] + h * p.os[1] + (int64_t)qrow * p.os[2] + 4 * hf;
        uint16_t *op = p.o + ooff;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int db = i >> 2, r4 = i & 3;
            const float *o4 = o + (qb * 4 + db) * 16 + r4 * 4;
            float x0 = o4[0] * inv, x1 = o4[1] * inv, x2 = o4[2] * inv, x3 = o4[3] * inv;
            u32x2 out;
            if constexpr (INPLACE) {
                // bf16 store of o_scale*result, then bf16 reduce-add into the base (csp_attn.cu:294-300)
                const u32x2 old = *(const u32x2 *)(p.o_in + ooff + db * 32 + r4 * 8);
                const float a0 = round_bf16(x0 * p.o_scale), a1 = round_bf16(x1 * p.o_scale);
                const float a2 = round_bf16(x2 * p.o_scale), a3 = round_bf16(x3 * p.o_scale);
                out[0] = pack_bf16x2(__uint_as_float(old[0] << 16) + a0, __uint_as_float(old[0] & 0xffff0000u) + a1);
                out[1] = pack_bf16x2(__uint_as_float(old[1] << 16) + a2, __uint_as_float(old[1] & 0xffff0000u) + a3);
            } else {
                out[0] = pack_bf16x2(x0, x1);
                out[1] = pack_bf16x2(x2, x3);
            }
            *(u32x2 *)(op + db * 32 + r4 * 8) = out;
        }
        if (!GATHER && p.l_out && hf == 0)
            p.l_out[(int64_t)bh * p.Nq + qrow] = 1.0f / (__builtin_amdgcn_exp2f(m[qb] * SCALE_LOG2E) * l);
    }
}

// ---- second pass of dense_colsum_attn for long launches: column sums of the softmax over each 192-row query group
// (reference csrc/attn/dense_colsum_attn.cu:230-277; the general kernel's CSONLY pass, attn.hip, is the small-size form).
// cs[bh][g][key] = sum over the group's rows q of exp2(s_qk*c + log2 p_q), p = the previous step's 1 / row sum.
// One WAVE per group (a workgroup = four consecutive groups of one head sharing the K tiles): S = Q.K^T with Q as the A
// operand, so a lane owns one key column (lane & 31) and the sum over the 192 rows is in-lane adds plus one lane-half
// swap per tile -- no cross-wave reduction, nothing order-dependent.  All of Q (192 x 128 bf16, pre-scaled by
// log2e/sqrt(D) like the general kernel's pass) sits in a[0:191], the K fragments of the tile in a[192:255]; the row
// offsets log2 p_q are the C operand of each block's first MFMA (96 VGPRs), so a score costs one v_exp_f32 and one
// v_add_f32.  Per 64-key tile six passes (32 query rows each) of 16 MFMAs; the exp2 / adds of a pass run in the gaps of
// the next one (three score buffers), the K fragments of the next tile are read during the last pass as each
// fragment's final MFMA has been issued.  A ragged last tile is fetched from the last 64 rows of the key range and its
// sums are simply stored again (same values).
template <int QB, int KB, int KS>
__device__ __forceinline__ void mfma_cs(f32x16 &d, const f32x16 &c0) {
    constexpr int qa = (QB * 8 + KS) * 4, ka = 192 + (KB * 8 + KS) * 4;
    if constexpr (KS == 0)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], a[%c4:%c5], %1" : "=&v"(d) : "v"(c0), "i"(qa), "i"(qa + 3), "i"(ka), "i"(ka + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(d) : "i"(qa), "i"(qa + 3), "i"(ka), "i"(ka + 3));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void colsum64_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5, l15 = lane & 15, lg = lane >> 4;
    const int G4 = (p.G + 3) / 4;
    const int wid = blockIdx.x;
    // (readfirstlane: the quotient comes out of the VALU; left in a VGPR, everything derived from it -- the K buffer resource
    // above all -- is "uniform but in vector registers", and every LDS-DMA then becomes a readfirstlane waterfall loop that
    // narrows and restores EXEC in the middle of the MFMA stream)
    const int bh = __builtin_amdgcn_readfirstlane(wid / G4), g = (wid - bh * G4) * 4 + w;   // this wave's 192-row group (>= p.G: nothing to store)
    const int b = __builtin_amdgcn_readfirstlane(bh / p.H), h = bh - b * p.H;
    const int row0 = g * 192;
    const int ntiles = (p.Nk + KT - 1) / KT;
    const int T4 = (ntiles + 3) & ~3;
    const uint16_t *kbase = p.k + b * p.ks[0] + h * p.ks[1];
    const __amdgpu_buffer_rsrc_t krsrc = make_rsrc(kbase);
    const uint32_t kstride_b = (uint32_t)p.ks[2] * 2u;

    // ---- Q fragments (A operand: lane = row l31 of the 32-row block, d = ks*16 + hf*8 .. +7), pre-scaled
    asm volatile("" ::: A64_CLOBBER_ALL);
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1];
        static_for<0, 48>([&](auto f) {
            constexpr int F = decltype(f)::value, qb = F >> 3, ks = F & 7;
            const int qrow = row0 + qb * 32 + l31;
            bf16x8 val = {};
            if (g < p.G && qrow < p.Nq) val = *(const bf16x8 *)(qp + (int64_t)qrow * p.qs[2] + ks * 16 + hf * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = (__bf16)((float)val[e] * SCALE_LOG2E);
            acc_write4<F * 4>(__builtin_bit_cast(u32x4, val));
        });
    }
    // ---- row offsets: element r of a block is row (r&3) + 8*(r>>2) + 4*hf of the 32-row block
    f32x16 offs[6];
#pragma unroll
    for (int qb = 0; qb < 6; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = row0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const float pl = (g < p.G && qr < p.Nq) ? p.p_in[(int64_t)bh * p.Nq + qr] : 0.f;
            offs[qb][r] = pl > 0.f ? __builtin_amdgcn_logf(pl) : -1.0e30f;   // (exp2(-huge) = 0; finite keeps 0 * -inf away)
        }

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    uint32_t kad[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kad[ks] = lds0 + l31 * 256 + (((2 * ks + hf) ^ l15) << 4);
    uint32_t kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kofs[i] = (uint32_t)(w * 16 + 4 * i + lg) * kstride_b + ((uint32_t)(l15 ^ (4 * i + lg)) << 4);
    const int last_base = p.Nk - KT;
    auto tile_base = [&](int T) { return T * KT < last_base ? T * KT : last_base; };
    auto issue_k1 = [&](uint32_t soff, uint32_t ldsw, int slot, int i) { blds16(krsrc, kofs[i], soff, smem + ldsw + slot * TB + i * 1024); };
    auto issue_k = [&](int T, int slot) {
        const uint32_t soff = (uint32_t)tile_base(T) * kstride_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_k1(soff, w * 4096, slot, i);
    };
    issue_k(0, 0), issue_k(1, 1), issue_k(2, 2), issue_k(3, 3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();
    static_for<0, 16>([&](auto i) {
        constexpr int I = decltype(i)::value;
        lds_k<(I >> 3), (I & 7), 0>(kad[I & 7]);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x16 s[3][2];         // scores of three passes in flight, block kb
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[i][kb][r] = -1.0e30f;   // "passes -1, -2": contribute exp2(-huge) = 0
    float cacc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [tile parity][kb]

    // one gap's share of the exp2 / adds: exp2 of elements [E0, E0 + 2) of the 32 a lane holds of one pass (buffer BUF; element
    // E = register E & 15 of block E >> 4), and the adds of the PREVIOUS gap's two exponentials (the add would otherwise wait
    // out the transcendental's latency).  pe / pe_tp / pe_kb: the exponentials in flight and where they go.
    float pe[2] = {0.f, 0.f};
    auto drain2 = [&](auto bufc, auto tpc, auto e0c, auto ptpc, auto pe0c) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, TP = decltype(tpc)::value, E0 = decltype(e0c)::value;
        constexpr int PTP = decltype(ptpc)::value, PE0 = decltype(pe0c)::value;
        (void)TP;
        const float n0 = __builtin_amdgcn_exp2f(s[BUF][E0 >> 4][E0 & 15]);
        const float n1 = __builtin_amdgcn_exp2f(s[BUF][(E0 + 1) >> 4][(E0 + 1) & 15]);
        cacc[PTP][PE0 >> 4] += pe[0];
        cacc[PTP][(PE0 + 1) >> 4] += pe[1];
        pe[0] = n0, pe[1] = n1;
        pin(pe[0]), pin(pe[1]), pin(cacc[PTP][PE0 >> 4]);
    };

    auto tile = [&](auto slc, int t) __attribute__((always_inline)) {
        constexpr int SL = decltype(slc)::value, TP = SL & 1, KNSL = (SL + 1) & 3;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        uint32_t ldsw = w * 4096;
        asm volatile("" : "+s"(ldsw));
        const uint32_t ksoff = (uint32_t)tile_base(t + 4) * kstride_b;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 6>([&](auto qq) {
            constexpr int QB = decltype(qq)::value;
            constexpr int PP = SL * 6 + QB;          // pass counter modulo 24 (4 tiles x 6): buffer = PP % 3
            constexpr int BUF = PP % 3, B1 = (PP + 2) % 3, B2 = (PP + 1) % 3;   // this pass, the previous one, the one before
            static_for<0, 16>([&](auto gg) {
                constexpr int G = decltype(gg)::value, ks = G >> 1, kb = G & 1;
                mfma_cs<QB, kb, ks>(s[BUF][kb], offs[QB]);
                if constexpr (QB == 0 && G >= 8 && G < 12) issue_k1(ksoff, ldsw, SL, G - 8);   // K(t+4) -> the slot K(t) left
                if constexpr (QB == 5 && kb == 1) {     // both fragments of this k step have been issued for the last time
                    lds_k<0, ks, KNSL>(kad[ks]);
                    lds_k<1, ks, KNSL>(kad[ks]);
                }
                // drain: gaps 2..15 take elements 0..27 of the previous pass, gaps 0..1 elements 28..31 of the one before
                // (the elements of the previous gap: gap 2's predecessor is gap 1 = elements 30, 31 of the pass before the previous
                // one; gap 0's predecessor is gap 15 of the previous pass = elements 26, 27 of the pass before it)
                constexpr int TP1 = QB == 0 ? TP ^ 1 : TP, TP2 = QB <= 1 ? TP ^ 1 : TP;   // tile parity of pass P-1 / P-2
                if constexpr (G > 2) drain2(ic<B1>{}, ic<TP1>{}, ic<(G - 2) * 2>{}, ic<TP1>{}, ic<(G - 3) * 2>{});
                else if constexpr (G == 2) drain2(ic<B1>{}, ic<TP1>{}, ic<0>{}, ic<TP2>{}, ic<30>{});
                else if constexpr (G == 1) drain2(ic<B2>{}, ic<TP2>{}, ic<30>{}, ic<TP2>{}, ic<28>{});
                else drain2(ic<B2>{}, ic<TP2>{}, ic<28>{}, ic<TP2>{}, ic<26>{});
                if constexpr (QB == 1 && G == 3) {
                    // the previous tile's sums are complete: lane halves folded with one swap (lanes 0-31: block 0, 32-63: block 1)
                    float a = cacc[TP ^ 1][0], c = cacc[TP ^ 1][1];
                    lane_swap32(a, c);
                    const float tot = a + c;
                    cacc[TP ^ 1][0] = 0.f, cacc[TP ^ 1][1] = 0.f;
                    if (g < p.G && t > 0)
                    {
                        // (branch-free rounding: a divergent NaN path would toggle EXEC here as well)
                        const uint32_t u = __float_as_uint(tot);
                        const uint32_t rne = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16, qn = (u >> 16) | 0x0040u;
                        p.cs[((int64_t)bh * p.G + g) * p.cs_stride + tile_base(t - 1) + lane] = (uint16_t)((u & 0x7fffffffu) > 0x7f800000u ? qn : rne);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K(t+1) fragments have landed
        __builtin_amdgcn_sched_barrier(0);
    };
    // tiles 0 .. T4-1 (padding tiles recompute the last one and store the same sums again), then two more passes' worth:
    // one extra tile whose gaps drain and store tile T4-1
    for (int tb = 0;; tb += 4) {
        tile(ic<0>{}, tb);
        if (tb >= T4) break;
        tile(ic<1>{}, tb + 1);
        tile(ic<2>{}, tb + 2);
        tile(ic<3>{}, tb + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
int launch64(const AttnParams &p, int64_t grid, hipStream_t stream) {
    auto kern = attn64_kernel<MODE>;
    static uint64_t lds_set = 0;
    constexpr int LDS = (MODE == 3 && !A64_CSUM_VALU) ? LDS_BYTES_CSUM + CS_EXCH_BYTES : MODE == 3 ? LDS_BYTES + CS_EXCH_BYTES : LDS_BYTES;
    ensure_dynamic_lds((const void *)kern, LDS, lds_set);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

}  // namespace

#ifdef ATTN64_PROF
extern "C" int chipmunk_attn64_prof_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_a64_prof), sizeof(g_a64_prof)) == hipSuccess ? 0 : 2;
}
#endif

// The largest K row norm per (batch, head) for the fixed reference point of attn64 / attn96 (nullptr: not available, the
// kernels then keep a running maximum).  Lives in the second half of the library scratch's 64 KiB ticket region (the tickets
// of attn.hip's hand-offs use the first few KiB); option attn_nomax = 2 switches it off (A/B, tests).
const float *chipmunk_knorm_max(const uint16_t *k, const int64_t ks[3], int B, int H, int Nk, hipStream_t stream) {
    if (chipmunk_get_option("attn_nomax") == 2 || (size_t)B * H * sizeof(float) > (32 << 10)) return nullptr;
    unsigned char *sc = (unsigned char *)chipmunk_scratch(stream, 64 << 10);
    if (!sc) return nullptr;
    float *km = (float *)(sc + (32 << 10));
    if (hipMemsetAsync(km, 0, (size_t)B * H * sizeof(float), stream) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(knorm_max_kernel, dim3(Nk >= 4096 ? 64 : 8, B * H), dim3(256), 0, stream, k, ks[0], ks[1], ks[2], H, Nk, km);
    return km;
}

// dense attention through the one-wave-per-SIMD kernel (strides in elements, [batch, head, row])
int chipmunk_dense64_launch(const void *q, const void *k, const void *v, void *o, float *l, const int64_t qs[3],
                            const int64_t ks[3], const int64_t vs[3], const int64_t os[3], int B, int H, int Nq, int Nk,
                            hipStream_t stream) {
    AttnParams p = {};
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v, p.o = (uint16_t *)o;
    for (int i = 0; i < 3; ++i) p.qs[i] = qs[i], p.ks[i] = ks[i], p.vs[i] = vs[i], p.os[i] = os[i];
    p.l_out = l;
    p.B = B, p.H = H, p.Nq = Nq, p.Nk = Nk, p.G = (Nq + WGROWS - 1) / WGROWS;
    p.o_scale = 1.f;
    p.kmax = chipmunk_knorm_max(p.k, ks, B, H, Nk, stream);
    return launch64<0>(p, (int64_t)B * H * p.G, stream);
}

// ---- dense_colsum_attn in one pass (MODE 3).  cs[bh][g][j] = sum over the group's rows i of exp2(s_ij c + log2 p_i) =
// sum_i P_ij w_i with the P the softmax pipeline has in registers anyway and w_i = exp2(m_i c) p_i (m = the reference
// point).  S^T puts the queries on the lanes, so the sum over a wave's 64 queries is a 5-level reduce-scatter over the 32
// lanes of a half (cs_pair*: 64 + 63 VALU operations per 64-key tile beside ~2 500 cycles of MFMA) instead of the second
// pass's 96 MFMAs and 192 exponentials per 192 rows; the waves of a workgroup that share a 192-row group add their 64 sums per tile
// through LDS and store one bf16 row per set (2 per workgroup; fp32 rows per wave were 21 GB of traffic each way at HunyuanVideo size),
// and cs_combine_kernel -- or the top-k mask kernel itself, when the caller only wants the mask -- adds the one or two rows of a group
// in fp32 (fixed order: the result does not depend on scheduling).
namespace {
__global__ __launch_bounds__(256) void cs_combine_kernel(const uint16_t *part, int pstride, uint16_t *cs, int NRB, int G, int Nq, int Nk, int cs_stride) {
    const int g = blockIdx.y, bh = blockIdx.z;
    int r0, r1;
    const int nrows = colsum_part_rows(g, NRB / 2, r0, r1);
    const uint16_t *src = part + ((int64_t)bh * NRB + r0) * pstride;
    const int64_t row1 = (int64_t)(r1 - r0) * pstride;     // (only read when nrows == 2)
    uint16_t *dst = cs + ((int64_t)bh * G + g) * cs_stride;
    if (((Nk | cs_stride) & 3) == 0) {
        const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (j >= Nk) return;
        float acc[4];
        {
            const u32x2 x = *(const u32x2 *)(src + j);
            acc[0] = __uint_as_float(x[0] << 16), acc[1] = __uint_as_float(x[0] & 0xffff0000u);
            acc[2] = __uint_as_float(x[1] << 16), acc[3] = __uint_as_float(x[1] & 0xffff0000u);
        }
        if (nrows == 2) {
            const u32x2 x = *(const u32x2 *)(src + row1 + j);
            acc[0] += __uint_as_float(x[0] << 16), acc[1] += __uint_as_float(x[0] & 0xffff0000u);
            acc[2] += __uint_as_float(x[1] << 16), acc[3] += __uint_as_float(x[1] & 0xffff0000u);
        }
        *(u32x2 *)(dst + j) = (u32x2){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
    } else {
        for (int e = 0; e < 4; ++e) {
            const int j = (blockIdx.x * 256 + threadIdx.x) * 4 + e;
            if (j >= Nk) return;
            float acc = bf16_bits_to_f32(src[j]);
            if (nrows == 2) acc += bf16_bits_to_f32(src[row1 + j]);
            dst[j] = f32_to_bf16_bits(acc);
        }
    }
}
}  // namespace

// column-sum pass of dense_colsum_attn, one wave per 192-row group (p.p_in, p.cs, p.cs_stride, p.G = groups of 192)
int chipmunk_colsum64_launch(const AttnParams &p, hipStream_t stream) {
    static uint64_t lds_set = 0;
    ensure_dynamic_lds((const void *)colsum64_kernel, NSL * TB, lds_set);
    const int64_t grid = (int64_t)p.B * p.H * ((p.G + 3) / 4);
    hipLaunchKernelGGL(colsum64_kernel, dim3((unsigned)grid), dim3(256), NSL * TB, stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

int chipmunk_colsum_part_stride(int Nk) { return (chipmunk_get_option("attn_cs_probe") & 1) ? Nk : (Nk + 63) & ~63; }

size_t chipmunk_colsum_part_bytes(int B, int H, int Nq, int Nk) {
    return (size_t)B * H * (((Nq + WGROWS - 1) / WGROWS) * 2) * (size_t)chipmunk_colsum_part_stride(Nk) * sizeof(uint16_t);
}

int chipmunk_dense64_colsum_launch(const AttnParams &p0, uint16_t *part, hipStream_t stream) {
    AttnParams p = p0;
    const int G192 = p.G;
    p.G = (p.Nq + WGROWS - 1) / WGROWS;
    p.cs_part = part, p.cs_pstride = chipmunk_colsum_part_stride(p.Nk);
    if (chipmunk_get_option("attn_cs_probe") & 2) p.probe |= 16;   // non-temporal partial-row stores
    if (chipmunk_get_option("attn_fused_colsum") == 3) p.probe |= 4;   // weighted column sums even where unit weights would do
    p.kmax = chipmunk_knorm_max(p.k, p.ks, p.B, p.H, p.Nk, stream);
    if (int rc = launch64<3>(p, (int64_t)p.B * p.H * p.G, stream)) return rc;
    if (!p.cs) return CHIPMUNK_OK;   // the caller reads the partial rows itself (chipmunk_topk_mask_parts)
    hipLaunchKernelGGL(cs_combine_kernel, dim3((unsigned)((p.Nk + 1023) / 1024), (unsigned)G192, (unsigned)(p.B * p.H)), dim3(256), 0, stream,
                       part, p.cs_pstride, p.cs, p.G * 2, G192, p.Nq, p.Nk, p.cs_stride);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

// gathered attention over the work plan built by launch_attn (attn.hip)
int chipmunk_csp64_launch(const AttnParams &p, int inplace, int grid, hipStream_t stream) {
    CM_CHECK(p.plan && p.tickets && p.ws, "csp64: work plan missing");
    return inplace ? launch64<2>(p, grid, stream) : launch64<1>(p, grid, stream);
}
