// Gathered (column-sparse) attention on gfx950 with ONE wave per SIMD and ALL four SIMDs of a CU computing:
// a 192-row query group (the reference's sparsity granule, modules/attn.py:95-96) = one workgroup of TWO waves x 96
// rows, two workgroups per CU.  (attn64.hip's 64-row waves leave a quarter of the CU idle on 192-row groups; the
// general kernel of attn.hip puts two 48-row waves on every SIMD and is bound by the issue slots they share.)
//   * per wave: O^T (96 x 128 f32) in a[0:191], the eight K fragments of the 32-key tile in a[192:223], the Q^T
//     fragments of the third 32-row block in a[224:255] (the other two blocks' in 64 VGPRs) -- accumulator registers
//     are only ever named from inline asm (same audit rule as attn64.hip);
//   * swapped layout S^T = K.Q^T on v_mfma_f32_32x32x16_bf16: a lane owns a query column, softmax statistics are
//     lane-local, the bf16 P^T feeds O^T += V^T.P^T from registers;
//   * per 32-key tile two phases of 24 MFMAs: A(t) = S(t) = K(t).Q^T beside the V^T(t-1) fragment reads and the DMA
//     issue; B(t) = O += V(t-1).P(t-1) beside the whole softmax of tile t, done IN PLACE on the score registers (a
//     second score buffer does not fit beside 64 VGPRs of Q), and the K(t+1) fragment reads.  Phase B is VALU-bound
//     (~200 VALU issues for 24 MFMAs), phase A runs at the MFMA rate;
//   * gather: each wave stages half of every K and V tile by LDS-DMA (4 + 4 pieces); LDS row 4*pc + lg of a tile holds
//     packed position (pc >> 2)*16 + lg*4 + (pc & 3), so a lane group reads 4 consecutive indices per tile and piece i
//     takes register i; index registers of four tiles in a ring; one s_barrier (two waves) + one counted vmcnt per tile;
//   * work items from the device-built plan of attn.hip; key slices merged by the last arriver (same hand-off).
#include "common.h"
#include "attn64_regs.h"
#include "attn_params.h"
#include "attn64_util.h"

namespace {

constexpr int KT = 32;                    // keys per tile
constexpr int TB = KT * 256;              // 8 KiB per K or V tile
constexpr int NSL = 4;                    // ring slots, K and V each
constexpr int VRING = NSL * TB;
constexpr int QLDS = 2 * NSL * TB;        // byte offset of the Q^T block 1 staging (8 KiB per wave, same swizzled layout as a K tile)
constexpr int LDS_BYTES = QLDS + 2 * 8192;   // 80 KiB: two workgroups per CU = all 160 KiB
constexpr float SCALE_LOG2E = 0.08838834764f * 1.44269504089f;
constexpr float MAX_LAG = 4.0f;

// S^T block qb (+)= K fragment ks . Q^T fragment (qb, ks); Q^T of blocks 0 and 1 in VGPRs, of block 2 in a[224:255]
template <int QB, int KS>
__device__ __forceinline__ void mfma_qk(f32x16 &s, const u32x4 &qv) {
    constexpr int ka = 192 + KS * 4, qa = 224 + KS * 4;
    if constexpr (QB < 2) {
        if constexpr (KS == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], %3, 0" : "=v"(s) : "i"(ka), "i"(ka + 3), "v"(qv));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], %3, %0" : "+v"(s) : "i"(ka), "i"(ka + 3), "v"(qv));
    } else {
        if constexpr (KS == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(ka), "i"(ka + 3), "i"(qa), "i"(qa + 3));
    }
}
template <int QB, int DB>
__device__ __forceinline__ void mfma_pv(const u32x4 &vf, const u32x4 &pf) {
    constexpr int oa = (QB * 4 + DB) * 16;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "i"(oa), "i"(oa + 15));
}
template <int KS, int SLOT>
__device__ __forceinline__ void lds_k(uint32_t addr) {
    constexpr int ka = 192 + KS * 4;
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "i"(ka), "i"(ka + 3), "i"(SLOT * TB) : "memory");
}

template <bool INPLACE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void csp96_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5, l15 = lane & 15, lg = lane >> 4;
    int wid = blockIdx.x;
    const int wid0 = wid;
    wid = p.plan[2 * wid0];
    if (wid < 0) return;
    const int meta = p.plan[2 * wid0 + 1];
    const int sp = meta & 0xffff, nsp = meta >> 16;
    const int slot0 = wid0 - sp, tix = wid0 - sp;
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
    const int32_t *idx = p.indices + ((int64_t)bh * p.G + g) * p.idx_stride;

    // ---- accumulator file: O^T = 0; Q^T fragments (B operand: lane = query l31, d = ks*16 + hf*8 .. +7)
    asm volatile(A96_ZERO_O ::: A64_CLOBBER_ALL);
    // block 0 in 32 VGPRs, block 2 in a[224:255], block 1 staged in LDS (its 32 registers do not fit beside the softmax) and
    // streamed through a two-fragment window, one ds_read_b128 per k step and tile
    u32x4 qv[8];
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1];
        static_for<0, 24>([&](auto f) {
            constexpr int F = decltype(f)::value, qb = F >> 3, ks = F & 7;
            const int qrow = row0 + qb * 32 + l31;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (qrow < p.Nq) val = *(const u32x4 *)(qp + (int64_t)qrow * p.qs[2] + ks * 16 + hf * 8);
            if constexpr (qb == 0) qv[ks] = val;
            else if constexpr (qb == 2) acc_write4<224 + ks * 4>(val);
            else *(u32x4 *)(smem + QLDS + w * 8192 + l31 * 256 + (((2 * ks + hf) ^ l15) << 4)) = val;   // (own wave's rows only)
        });
    }

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
    const bool idx_vec = ((uintptr_t)idx & 15) == 0 && (p.idx_stride & 3) == 0;
    auto load_idx = [&](int T) {
        const int base = (tbeg + T) * KT + w * 16 + lg * 4;
        u32x4 v4;
        if (idx_vec && (tbeg + T) * KT + KT <= p.idx_stride) {
            v4 = *(const u32x4 *)(idx + base);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int pos = base + e;
                pos = pos < p.idx_stride ? pos : p.idx_stride - 1;
                v4[e] = (uint32_t)idx[pos];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = (uint32_t)max(0, min((int)v4[e], p.Nk - 1));   // memory safety for malformed indices
        return v4;
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

    // ---- prologue: K(0); "iterations" -3..-1 (iteration i issues K(i+4) and V(i+2); V(-1) does not exist: V(0) goes into
    //      its slot so that the first tile's PV -- P = 0 -- multiplies finite numbers)
    u32x4 ir[4];   // gather keys of four tiles (entry = tile mod 4)
    ir[0] = load_idx(0), ir[1] = load_idx(1), ir[2] = load_idx(2), ir[3] = load_idx(3);
    issue_k(ir[0], 0);
    issue_k(ir[1], 1), issue_v(ir[0], 3);
    issue_k(ir[2], 2), issue_v(ir[0], 0);
    issue_k(ir[3], 3), issue_v(ir[1], 1);
    ir[0] = load_idx(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, 8>([&](auto i) { lds_k<decltype(i)::value, 0>(kad[decltype(i)::value]); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x16 s[3];                 // S^T(t) of the three query blocks, then x and p IN PLACE
    u32x4 pw0[3] = {};           // P^T fragments of key slab 0 (rewritten once the pending PV has read them)
    u32x4 pw1[3] = {};           // ... of key slab 1 (rewritten after the last MFMA of the phase: a second buffer does not fit)
    float m[3] = {-INFINITY, -INFINITY, -INFINITY}, nmsc[3] = {0.f, 0.f, 0.f}, mlag[3] = {-INFINITY, -INFINITY, -INFINITY};
    float lacc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float alpha[3] = {1.f, 1.f, 1.f};

    auto vfrag_read = [&](auto dbc, auto offc) __attribute__((always_inline)) {
        constexpr int DB = decltype(dbc)::value, OFF = decltype(offc)::value;
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4"
                     : "=v"(lo), "=v"(hi) : "v"(vad[DB]), "i"(OFF), "i"(OFF + 2048) : "memory");
        return (u32x4){lo[0], lo[1], hi[0], hi[1]};
    };
    // softmax element i of a lane's 48 scores per tile, in key-slab order: slab = i / 24, block qb = (i % 24) / 8,
    // register r = slab*8 + i % 8
    auto ecum = [](int n) constexpr { return n <= 0 ? 0 : n >= 13 ? 48 : (n * 48) / 13; };

    u32x4 q1[2];   // Q^T block 1, fragment window (k step parity); fragments 0 and 1 of a tile are read at the end of the tile before
    auto q1_read = [&](auto ksc) __attribute__((always_inline)) {
        constexpr int KS = decltype(ksc)::value;
        const uint32_t addr = qad + (kad[KS] - lds0 - l31 * 256);
        u32x4 fr;
        asm volatile("ds_read_b128 %0, %1" : "=v"(fr) : "v"(addr) : "memory");
        q1[KS & 1] = fr;
    };
    q1_read(ic<0>{});
    q1_read(ic<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q1[0]), "+v"(q1[1]));

    auto tile = [&](auto slc, int t) __attribute__((always_inline)) {
        constexpr int SL = decltype(slc)::value, PAR = SL & 1;
        constexpr int VSL = (SL + 3) & 3;        // slot of V(t-1)
        constexpr int KNSL = (SL + 1) & 3;       // slot of K(t+1)
        // K(t+1) and V(t-1) were issued three iterations ago; an iteration is 8 pieces + 1 (4) index loads
        if (idx_vec) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ir[(SL + 1) & 3] = load_idx(t + 5);
        __builtin_amdgcn_sched_barrier(0);

        u32x4 vf[8];
        // ================= phase A: S(t) = K(t).Q^T  ||  V^T(t-1) -> registers, DMA of K(t+4) / V(t+2) =================
        static_for<0, 24>([&](auto gg) {
            constexpr int G = decltype(gg)::value, ks = G / 3, qb = G % 3;
            if constexpr (qb == 1) {
                // the fragment of this k step has landed: LDS operations return in order, the count is the number of younger
                // ones at this point (window reads, V^T fragment reads of gaps 0..3), enumerated by hand
                // (fragments 0 and 1 were read -- and waited for -- at the end of the previous tile)
                constexpr int YOUNGER = ks == 2 ? 7 : ks == 7 ? 0 : 1;
                if constexpr (ks >= 2) asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(q1[ks & 1]) : "i"(YOUNGER));
            }
            mfma_qk<qb, ks>(s[qb], qb == 0 ? qv[ks] : q1[ks & 1]);
            if constexpr (qb == 1 && ks + 2 < 8) q1_read(ic<ks + 2>{});
            if constexpr (G < 4) vf[G] = vfrag_read(ic<G>{}, ic<VSL * TB>{});   // slab 0; slab 1 follows in phase B, register by register
            if constexpr (G >= 8 && (G & 1) == 0) {
                constexpr int PC = (G - 8) >> 1;
                if constexpr (PC < 4) issue_k1(ir[SL], SL, PC);
                else issue_v1(ir[(SL + 2) & 3], (SL + 2) & 3, PC - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
        __builtin_amdgcn_sched_barrier(0);

        // ================= phase B: O += V(t-1).P(t-1)  ||  the softmax of tile t, in place; K(t+1) -> a[192:223] ========
        float mx[3];
        bool moved = false;
        const int vleft = t < ntiles ? valid - (tbeg + t) * KT : 0;   // packed positions of this tile that exist
        auto gap = [&](auto gg) __attribute__((always_inline)) {
            constexpr int G = decltype(gg)::value, up = G / 12, db = (G % 12) / 3, qb = G % 3;
            const u32x4 &pf = up == 0 ? pw0[qb] : pw1[qb];
            // the V^T fragments of slab 1 are read as the slab-0 fragment of the same d block has been used for the last time
            // (nine gaps before their own first use); two waits cover them
            if constexpr (G == 12) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(vf[4]), "+v"(vf[5]), "+v"(vf[6]));   // (the fourth's two reads, one gap old, may stay in flight)
            if constexpr (G == 21) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[7]));
            mfma_pv<qb, db>(vf[up * 4 + db], pf);
            if constexpr (G < 8) lds_k<G, KNSL>(kad[G]);
            if constexpr (G < 12 && G % 3 == 2) vf[4 + G / 3] = vfrag_read(ic<G / 3>{}, ic<VSL * TB + 4096>{});
            if constexpr (G == 15) q1_read(ic<0>{});   // the next tile's first two Q^T block-1 fragments (same data every tile)
            if constexpr (G == 16) q1_read(ic<1>{});
            if constexpr (G == 1) {
                if (vleft < KT) {
                    // element (qb, r, hf) is LDS row (r&3) + 8*(r>>2) + 4*hf of the tile; row 4*pc + lg holds packed position
                    // (pc>>2)*16 + lg*4 + (pc&3) = (r>>3)*16 + (r&3)*4 + 2*((r>>2)&1) + hf; positions >= vleft do not exist
                    int thr = vleft - hf;   // opaque: the 16 per-register constants are compared as immediates
                    asm volatile("" : "+v"(thr));
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r >> 3) * 16 + (r & 3) * 4 + 2 * ((r >> 2) & 1) >= thr) s[q2][r] = -INFINITY;
                }
            }
            if constexpr (G >= 2 && G <= 5) {   // maxima of the 16 scores a lane holds per block: v_max + 7 x v_max3, two ops per block per gap
                constexpr int j = G - 2, r0 = j * 4;
                if constexpr (j == 0)
                    asm volatile("v_max_f32 %0, %3, %4\n\tv_max_f32 %1, %7, %8\n\tv_max_f32 %2, %11, %12\n\t"
                                 "v_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %9, %10\n\tv_max3_f32 %2, %2, %13, %14"
                                 : "=&v"(mx[0]), "=&v"(mx[1]), "=&v"(mx[2])
                                 : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]),
                                   "v"(s[2][0]), "v"(s[2][1]), "v"(s[2][2]), "v"(s[2][3]));
                else
                    asm volatile("v_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %1, %1, %7, %8\n\tv_max3_f32 %2, %2, %11, %12\n\t"
                                 "v_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %9, %10\n\tv_max3_f32 %2, %2, %13, %14"
                                 : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2])
                                 : "v"(s[0][r0]), "v"(s[0][r0 + 1]), "v"(s[0][r0 + 2]), "v"(s[0][r0 + 3]), "v"(s[1][r0]), "v"(s[1][r0 + 1]),
                                   "v"(s[1][r0 + 2]), "v"(s[1][r0 + 3]), "v"(s[2][r0]), "v"(s[2][r0 + 1]), "v"(s[2][r0 + 2]), "v"(s[2][r0 + 3]));
            }
            if constexpr (G == 6) {   // the other half of the keys lives in lane ^ 32
                float t0, t1, t2;
                asm volatile("v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\ts_nop 1\n\t"
                             "v_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5\n\ts_nop 0\n\t"
                             "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5"
                             : "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2));
            }
            if constexpr (G >= 8 && G < 20) {   // X: x = s*c - m*c in place, four elements per gap
                static_for<(G - 8) * 4, (G - 7) * 4>([&](auto ii) {
                    constexpr int I = decltype(ii)::value, q2 = (I % 24) / 8, r = (I / 24) * 8 + I % 8;
                    float x = __builtin_fmaf(s[q2][r], SCALE_LOG2E, nmsc[q2]);
                    pin(x);   // (the scalar, not the tuple: pinning the 16-register tuples costs ~40 VGPRs of liveness)
                    s[q2][r] = x;
                });
            }
            if constexpr (G >= 9 && G < 22) {   // E: p = exp2(x) in place
                static_for<ecum(G - 9), ecum(G - 8)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value, q2 = (I % 24) / 8, r = (I / 24) * 8 + I % 8;
                    float e = __builtin_amdgcn_exp2f(s[q2][r]);
                    pin(e);
                    s[q2][r] = e;
                });
            }
            if constexpr (G >= 10 && G < 23) {  // L: row sums, one gap behind the exponentials
                static_for<ecum(G - 10), ecum(G - 9)>([&](auto ii) {
                    constexpr int I = decltype(ii)::value, q2 = (I % 24) / 8, r = (I / 24) * 8 + I % 8;
                    lacc[q2][I & 1] += s[q2][r];
                    pin(lacc[q2][I & 1]);
                });
            }
            if constexpr (G >= 17 && G < 21) {  // C, slab 0: three bf16 pairs per gap (the pending PV has left slab 0 behind at gap 12)
                static_for<(G - 17) * 3, (G - 16) * 3>([&](auto jj) {
                    constexpr int J = decltype(jj)::value, q2 = J / 4, d = J % 4;
                    uint32_t pk = pack_bf16x2(s[q2][2 * d], s[q2][2 * d + 1]);
                    pin(pk);
                    pw0[q2][d] = pk;
                });
            }

            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 8>(gap);
        {   // the (rare) move of the reference point
            constexpr float LAG_RAW = MAX_LAG / SCALE_LOG2E;
            if (__builtin_amdgcn_ballot_w64(mx[0] > mlag[0]) | __builtin_amdgcn_ballot_w64(mx[1] > mlag[1]) |
                __builtin_amdgcn_ballot_w64(mx[2] > mlag[2])) {
                moved = true;
#pragma unroll
                for (int q2 = 0; q2 < 3; ++q2) {
                    const float m_new = max2(m[q2], mx[q2]);
                    alpha[q2] = __builtin_amdgcn_exp2f((m[q2] - m_new) * SCALE_LOG2E);
                    lacc[q2][0] *= alpha[q2];
                    lacc[q2][1] *= alpha[q2];
                    m[q2] = m_new;
                    nmsc[q2] = -m_new * SCALE_LOG2E;
                    mlag[q2] = m_new + LAG_RAW;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<8, 24>(gap);
        static_for<0, 12>([&](auto jj) {   // C, slab 1: the pending PV has issued its last MFMA on the old fragments
            constexpr int J = decltype(jj)::value, q2 = J / 4, d = J % 4;
            pw1[q2][d] = pack_bf16x2(s[q2][8 + 2 * d], s[q2][8 + 2 * d + 1]);
        });
        pin(lacc[0][0]), pin(lacc[0][1]), pin(lacc[1][0]), pin(lacc[1][1]), pin(lacc[2][0]), pin(lacc[2][1]);
        if (moved) {   // O_t = alpha (O_{t-1} + P_{t-1} V_{t-1}): after the pending PV, before the next one
            float tmp;
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" A96_SCALE_QB0 : "=&v"(tmp) : "v"(alpha[0]));
            asm volatile(A96_SCALE_QB1 : "=&v"(tmp) : "v"(alpha[1]));
            asm volatile(A96_SCALE_QB2 : "=&v"(tmp) : "v"(alpha[2]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q1[0]), "+v"(q1[1]));   // the window's first two and the K(t+1) fragments have landed
        __builtin_amdgcn_sched_barrier(0);
    };

    // tiles 0 .. T4-1 (padding tiles are fully masked), then one more pass whose phase B accumulates tile T4-1
    for (int tb = 0;; tb += 4) {
        tile(ic<0>{}, tb);
        if (tb >= T4) break;
        tile(ic<1>{}, tb + 1);
        tile(ic<2>{}, tb + 2);
        tile(ic<3>{}, tb + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    auto store_o = [&](int qb, const float (&o)[64], float l) __attribute__((always_inline)) {
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
                const u32x2 old = *(const u32x2 *)(p.o_in + ooff + db * 32 + r4 * 8);
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
        static_for<0, 3>([&](auto qq) {
            float o[64];
            read_o(qq, o);
            store_o(decltype(qq)::value, o, lq[decltype(qq)::value]);
        });
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
            const float a = __builtin_amdgcn_exp2f((mm - m_new) * SCALE_LOG2E);
            const float c = __builtin_amdgcn_exp2f((ms - m_new) * SCALE_LOG2E);
            mm = m_new;
            ll = ll * a + ls * c;
#pragma unroll
            for (int i = 0; i < 64; ++i) o[i] = o[i] * a + oth[(qb * 64 + i) * 128] * c;
        }
        store_o(qb, o, ll);
    }
}

template <bool INPLACE>
int launch96(const AttnParams &p, int grid, hipStream_t stream) {
    auto kern = csp96_kernel<INPLACE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(128), LDS_BYTES, stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

}  // namespace

// gathered attention over the work plan built by launch_attn (attn.hip); inplace = 1 for the accumulate forms
int chipmunk_csp96_launch(const AttnParams &p, int inplace, int grid, hipStream_t stream) {
    CM_CHECK(p.plan && p.tickets && p.ws, "csp96: work plan missing");
    return inplace ? launch96<true>(p, grid, stream) : launch96<false>(p, grid, stream);
}
