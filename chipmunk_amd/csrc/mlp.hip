// Column-sparse ("delta") MLP GEMMs for gfx950.
// Replaces reference csrc/mlp/csp_mlp_mm1.cu, csrc/mlp/csp_mlp_mm2_and_scatter_add.cu (+ the Triton GEMM
// src/chipmunk/triton/csp_mlp_mm2.py) and csrc/indexed_io/scatter_add.cu.
//
// Both GEMMs: workgroup tile = 128 rows (one sparsity group, reference bm = 128) x BN packed columns, K step BK, 4 or 8
// waves, v_mfma_f32_32x32x16_bf16 accumulators, NST-deep LDS ring filled by buffer-form LDS-DMA
// (buffer_load_dwordx4 ... lds) with counted vmcnt across a raw s_barrier; shapes are template parameters, the shipped
// ones are GEMM1 <128 cols, K step 64, 2 stages, 2 workgroups/CU, 4 waves of 64x64> and GEMM2 <256, 32, 3, 2, 8 waves
// of 64x64>.  The gather is the per-lane source offset of the DMA: fc1 rows (mm1) are 2*K contiguous bytes, fc2^T rows
// (mm2) are 2*N2 contiguous bytes, so every gathered piece is a full 128-byte line.  XOR swizzles are applied on the
// source chunk index so the lane-linear LDS image is conflict-free for ds_read_b128 (k-contiguous operands) and
// ds_read_b64_tr_b16 (fc2^T, which is n-contiguous in memory and must be fed k-contiguous to the MFMA).  Epilogues go
// through the freed ring so that global memory only sees 16-byte accesses over whole row segments.
#include "common.h"
#include "attn64_util.h"
#include <type_traits>

namespace {

constexpr int BM = 128;  // rows per workgroup = one sparsity group (reference bm = 128)

#ifdef MLP_PROF
// Cycle anatomy of the GEMM k loops (tools/mlp_prof.py builds a separate library with -DMLP_PROF): s_memtime at the segment
// boundaries of every k step, per wave of one mid-grid workgroup.  Not part of the product build.
__device__ unsigned long long g_mlp_prof[16 * 8];
#define MPROF_DECL unsigned long long pt_ = 0, pacc_[6] = {0, 0, 0, 0, 0, 0}; const bool prof_on_ = blockIdx.x == 161
#define MPROF_START() do { if (prof_on_) pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define MPROF_MARK(i) do { if (prof_on_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; } } while (0)
#define MPROF_ABS(i) do { if (blockIdx.x == 161 && (threadIdx.x & 63) == 0) g_mlp_prof[96 + (threadIdx.x >> 6) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MPROF_END(w, n) do { if (prof_on_ && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 6; ++i_) g_mlp_prof[(w) * 8 + i_] = pacc_[i_]; g_mlp_prof[(w) * 8 + 7] = (n); } } while (0)
extern "C" int chipmunk_mlp_prof_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_prof), sizeof(g_mlp_prof)) == hipSuccess ? 0 : 2;
}
#else
#define MPROF_DECL
#define MPROF_START()
#define MPROF_MARK(i)
#define MPROF_END(w, n)
#define MPROF_ABS(i)
#endif


__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// tanh-GeLU (reference csrc/common/elementwise/gelu.cuh:26-30): x*0.5*(1+tanh(u)) == x*(1 - 1/(1+exp(2u))), u = 0.79788456*(x + 0.044715 x^3);
// evaluated as x - x / (1 + exp2(x * (GA + GB x^2))).  The 2-wide form is the same operation sequence on v_pk_*_f32 (the epilogue
// is VALU-bound: tools/mlp_prof.py, 18 k of a tile's 103 k cycles before this form), element-for-element the same bits.
constexpr float GELU_A = 0.7978845608028654f * 2.0f * 1.44269504089f, GELU_B = GELU_A * 0.044715f;
__device__ __forceinline__ uint32_t pack_bf16x2_v(f32x2 v) { return pack_bf16x2(v[0], v[1]); }   // v_cvt_pk_bf16_f32, RNE
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t u) { return (f32x2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ float gelu_tanh(float x) {
    const float t = __builtin_fmaf(x * x, GELU_B, GELU_A);
    const float e = __builtin_amdgcn_exp2f(x * t);
    return __builtin_fmaf(-x, __builtin_amdgcn_rcpf(e + 1.0f), x);
}
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    const f32x2 t = __builtin_elementwise_fma(x * x, (f32x2){GELU_B, GELU_B}, (f32x2){GELU_A, GELU_A});
    const f32x2 u = x * t;
    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return __builtin_elementwise_fma(-x, r, x);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate out of range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Geometry of a k-contiguous operand tile [ROWS][BK] bf16 staged by LDS-DMA:
//   row stride BK*2 bytes, CPR = BK/8 16-byte chunks per row, one DMA instruction (64 lanes x 16 B) covers RPI rows;
//   chunk c of row r is stored at chunk c ^ swz(r), swz(r) = (r / RPB) & (CPR-1), RPB = rows per 256-byte bank row:
//   conflict-free for the ds_read_b128 lane groups of a 32-row MFMA operand fragment.
template <int BK>
struct KTile {
    static constexpr int ROWB = BK * 2, CPR = BK / 8, RPI = 1024 / ROWB, RPB = 256 / ROWB;
    __device__ static __forceinline__ int swz(int row) { return (row / RPB) & (CPR - 1); }
    // source element offset inside a row for the lane's stored chunk
    __device__ static __forceinline__ int src_chunk_elems(int row, int lane) { return ((lane % CPR) ^ swz(row)) << 3; }
    __device__ static __forceinline__ int lane_row(int inst, int lane) { return inst * RPI + lane / CPR; }
    __device__ static __forceinline__ bf16x8 frag(const unsigned char *tile, int row, int kk, int lane) {
        const int c = kk * 2 + (lane >> 5);
        return *(const bf16x8 *)(tile + row * ROWB + ((c ^ swz(row)) << 4));
    }
};


// Live-tile map shared by both GEMMs.  The host launches G x NTmax workgroups without knowing the per-group counts
// (they live in HBM); every workgroup derives the same compact order from them:
//   * only column tiles below max_g counts[g] are live (a dead workgroup exits in a few hundred cycles);
//   * the live tiles are split into 8 contiguous chunks, one per XCD (block b runs on XCD b % 8);
//   * inside a chunk consecutive tiles walk "NR column tiles x all groups": with ascending index lists the tiles of
//     DIFFERENT groups over the same column-tile position gather overlapping weight rows, so the ~64 workgroups an XCD
//     runs at once re-use each other's rows (and each group's activation tile) out of that XCD's 4 MiB L2 instead of
//     re-fetching them through the fabric.  Measured on FLUX shapes: fabric fetch per launch 860 MB -> see DESIGN.md.
//   * tail split (NSUB > 1): workgroups are dispatched in block order as slots free up, so with T equal tiles on S
//     resident slots the last T mod S tiles run alone for a whole tile time (FLUX: 34 groups x 32 column tiles = 1088
//     tiles on 512 slots -> the third "round" is 12 % full and costs a third of the launch).  When an XCD's leftover
//     is small, each leftover tile is handed out as NSUB quarter-size sub-tiles (64 x 64 outputs, full K, deeper
//     ring) to NSUB workgroups at the END of that XCD's block range.  Every output element is still produced by one
//     workgroup with the same K order, so results are bit-identical to the unsplit schedule.
struct TileMap {
    int g, nt, sub;  // sub < 0: whole tile
    bool live;
};
// The part of the map that is the same for every tile of a workgroup: live column tiles, this XCD's share and where its tail begins.
struct TilePlan {
    int G, NR, NTl, mine, full, base, nsub;
    // workgroup slots of this XCD that carry work: whole tiles, then the tail's sub-tiles
    __device__ __forceinline__ int slots() const { return full + (mine - full) * nsub; }
};
template <int BN>
__device__ __forceinline__ TilePlan plan_tiles(const int32_t *counts, int G, int NTmax, int NR, int slots_per_xcd = 0, int nsub = 1) {
    int cmax = 0;  // wave-parallel max (a scalar loop over G costs ~50 ns per group, per workgroup)
    for (int g = threadIdx.x & 63; g < G; g += 64) cmax = max(cmax, counts[g]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    TilePlan pl;
    pl.G = G, pl.NR = NR, pl.nsub = nsub;
    pl.NTl = min((cmax + BN - 1) / BN, NTmax);
    const int total = pl.NTl * G;
    const int xcd = blockIdx.x & 7;
    const int q = total >> 3, r = total & 7;
    pl.mine = q + (xcd < r ? 1 : 0);  // tiles of this XCD
    pl.full = pl.mine;
    if (nsub > 1 && slots_per_xcd > 0) {
        const int rem = pl.mine % slots_per_xcd;
        if (pl.mine > slots_per_xcd && rem > 0 && rem * nsub <= slots_per_xcd) pl.full = pl.mine - rem;
    }
    pl.base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return pl;
}
__device__ __forceinline__ TileMap tile_at(const TilePlan &pl, int slot) {
    TileMap m;
    m.sub = -1;
    if (slot >= pl.full) {
        const int k = slot - pl.full;
        m.sub = k % pl.nsub;
        slot = pl.full + k / pl.nsub;
    }
    m.live = slot < pl.mine;
    const int t = pl.base + slot;
    const int per = pl.G * pl.NR;
    const int nb = t / per, rem = t - nb * per;
    const int nr = min(pl.NR, pl.NTl - nb * pl.NR);
    m.g = nr > 0 ? rem / nr : 0;
    m.nt = nb * pl.NR + (nr > 0 ? rem - m.g * nr : 0);
    return m;
}

// ------------------------------------------------------------------------------------------------ mm1
struct Mm1Params {
    const uint16_t *a, *b, *bias;
    uint16_t *cache;
    uint16_t *c;
    const int32_t *indices, *counts;
    int M, K, F, NT, NR, probe, slots_per_xcd;
    int update_cache;  // 1: also apply the scatter-add of this tile's deltas to the cache block it already holds in LDS;
                       // 2 (fp8): store the new activation into the cache like the reference's Triton kernel
    const float *scale_a, *scale_b;  // fp8 only: reciprocal quantisation scales (one float each)
};

// One TM x TN output tile (TM rows of group g starting at m_off, packed columns n0 .. n0+TN-1): 4 waves as 2 x 2, each a
// (TM/2) x (TN/2) accumulator of 32x32x16 MFMA tiles; NST-deep LDS ring of [A tile | B tile] stages.
// FP8: operands are OCP e4m3 bytes (BASELINE config C5; reference src/chipmunk/triton/csp_mlp_mm1.py:37-164); BK then
// still counts 2-byte units, i.e. a k step is BK*2 = 128 bytes of every row either way.
// NW = 8 (round 4, option mm1_variant = 10; NOT the default): the same 64 x 64 accumulator per wave, 2 x 4 waves over a 128 x 256 tile -- two
// waves per SIMD out of ONE workgroup per CU instead of two 128 x 128 workgroups: 48 KiB instead of 64 KiB through the L2 -> LDS path per
// 128 x 256 outputs and k step, three stages in 144 KiB.  MEASURED SLOWER: 145-149 vs 117-125 us (FLUX bf16), 231-234 vs 206-222 us
// (Wan fp8), tools/mlp_prof.py: 2 000 ticks per k step against 1 620 for the PAIR of 128 x 128 workgroups, and an epilogue of 12.8 k
// ticks that nothing overlaps (two independent workgroups run one's epilogue under the other's k loop; here all eight waves sit in it).
// Neither the second wave of a SIMD issuing its DMA pieces after its MFMAs (2 200 per step) nor the pieces spread between the MFMA
// groups (no change; the 4-wave form 125 -> 131 us bf16, 219 -> 212 us fp8, inside box noise) helps: the loop waits on the
// landing of the gathered rows, not on their bytes or their issue.
template <int TM, int TN, int BK, int NST, bool FP8 = false, int NW = 4>
__device__ __forceinline__ void mm1_tile(const Mm1Params &p, unsigned char *smem, int g, int m_off, int n0, int cnt) {
    using KT = KTile<BK>;
    constexpr uint32_t ESZ = FP8 ? 1u : 2u;  // operand element size in bytes
    constexpr int A_TILE = TM * BK * 2, B_TILE = TN * BK * 2, STAGE = A_TILE + B_TILE;
    constexpr int A_INST = A_TILE / (1024 * NW), B_INST = B_TILE / (1024 * NW);  // DMA instructions per wave per tile
    constexpr int WNC = NW / 2;                                     // waves across the tile's columns (2 down its rows)
    constexpr int MT = TM / 64, NT4 = TN / (32 * WNC);              // 32-wide m / n tiles per wave
    static_assert(A_INST >= 1 && B_INST >= 1, "tile too small for one DMA instruction per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WNC, wn = w % WNC;
    const int32_t *idxg = p.indices + (int64_t)g * p.F;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a), rb = make_rsrc(p.b);
    uint32_t aoff[A_INST], boff[B_INST];  // byte offsets
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = KT::lane_row(w * A_INST + i, lane);
        aoff[i] = (uint32_t)(g * BM + m_off + row) * p.K * ESZ + KT::src_chunk_elems(row, lane) * 2u;
    }
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        const int row = KT::lane_row(w * B_INST + i, lane);
        const int j = n0 + row;
        const int key = idxg[j < cnt ? j : n0];  // rows past the count re-read a live row and are never stored
        boff[i] = (uint32_t)key * p.K * ESZ + KT::src_chunk_elems(row, lane) * 2u;
    }
    auto issue = [&](int kb, int buf) {
        unsigned char *st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_INST; ++i) blds16(ra, aoff[i], kb * BK * 2, st + (w * A_INST + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_INST; ++i) blds16(rb, boff[i], kb * BK * 2, st + A_TILE + (w * B_INST + i) * 1024);
    };
    // Epilogue operands staged through LDS when the tile's slice of the activation cache fits one ring stage: the
    // TN x TM block cache[idx[n0..], g*BM+m_off ..] (TN rows of TM*2 contiguous bytes) is fetched by LDS-DMA during
    // the LAST k step into the stage no tile needs any more, and the bf16 results leave through the other stage as
    // 16-byte row-major stores.  (Direct form: 8-byte gathered loads and 2-byte stores, 64 of each per lane --
    // measured 34 us of a 148 us launch.)
    // FLAT (tiles whose outputs exceed one stage): the ring starts at the stage that makes the LAST k step compute out of stage 0, the
    // cache block lands behind it in [STAGE, STAGE + TM*TN*2) and the outputs leave through stage 0 plus the bytes behind the cache block.
    constexpr int EPI = TM * TN * 2;
    constexpr bool FLAT = NW == 8 && EPI > STAGE && EPI <= (NST - 1) * STAGE && 2 * EPI <= NST * STAGE && STAGE % (TN * 2) == 0;
    constexpr bool STAGED = EPI <= STAGE || FLAT;
    static_assert(!FP8 || STAGED, "the fp8 form is only built for tile shapes with the staged epilogue");
    constexpr int LPR = TM * 2 / 16;                                  // 16-byte chunks per cache row
    constexpr int C_INST = STAGED ? EPI / (1024 * NW) : 1;            // DMA instructions per wave
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.cache);
    uint32_t coff[C_INST];
    if constexpr (STAGED) {
#pragma unroll
        for (int i = 0; i < C_INST; ++i) {
            const int jj = (w * C_INST + i) * (64 / LPR) + lane / LPR;  // tile-local packed column
            const int j = n0 + jj;
            const int col = idxg[j < cnt ? j : n0];
            coff[i] = ((uint32_t)col * p.M + g * BM + m_off + (((lane % LPR) ^ (jj & (LPR - 1))) << 3)) * 2u;
        }
    }
    auto issue_cache = [&](int buf) {
        if constexpr (STAGED) {
#pragma unroll
            for (int i = 0; i < C_INST; ++i) blds16(rc, coff[i], 0, smem + (FLAT ? STAGE : buf * STAGE) + (w * C_INST + i) * 1024);
        }
    };

    // bf16: the bias seeds the fp32 accumulators, as in the reference (csp_mlp_mm1.cu:347-350) -- its two dependent loads (index,
    // then bias) fly during the prologue instead of in front of the epilogue.  fp8 scales the sum first, so it starts from zero.
    constexpr bool SEED_BIAS = !FP8;
    float bias_v[NT4], sa = 1.f, sb = 1.f;   // loaded here for both forms: in front of the epilogue the round trips would be exposed
    if constexpr (FP8) sa = p.scale_a[0], sb = p.scale_b[0];
    f32x16 acc[MT][NT4];
#pragma unroll
    for (int n4 = 0; n4 < NT4; ++n4) {
        const int j = n0 + wn * (TN / WNC) + n4 * 32 + (lane & 31);
        bias_v[n4] = bf16_bits_to_f32(p.bias[idxg[j < cnt ? j : n0]]);
        const float seed = SEED_BIAS ? bias_v[n4] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][n4][r] = seed;
    }

    const int nkb = (int)((uint32_t)p.K * ESZ / (BK * 2));
    int buf = FLAT ? (NST - (nkb - 1) % NST) % NST : 0;   // FLAT: k step nkb-1 computes out of stage 0
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nkb) issue(s, (buf + s) % NST);
    int nbuf = (buf + NST - 1) % NST;
    MPROF_DECL;
    MPROF_ABS(1);
    MPROF_START();
    for (int kb = 0; kb < nkb; ++kb) {
        // tile kb must have landed; the NST-2 younger tiles may stay in flight across the barrier
        if (kb + NST - 1 <= nkb) wait_vmcnt<(NST - 2) * (A_INST + B_INST)>();
        else wait_vmcnt<0>();
        MPROF_MARK(0);
        __builtin_amdgcn_s_barrier();
        MPROF_MARK(1);
        if (kb + NST - 1 < nkb && p.probe != 1) issue(kb + NST - 1, nbuf);
        if (kb == nkb - 1) issue_cache(nbuf);  // the stage tile kb-1 occupied is free for good
        MPROF_MARK(2);
        if (p.probe == 2) {
            buf = buf + 1 == NST ? 0 : buf + 1;
            nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
            continue;
        }
        const unsigned char *At = smem + buf * STAGE;
        const unsigned char *Bt = At + A_TILE;
        // operand fragments are double-buffered in registers: the ds_reads of k-slice kk+1 are in flight while the
        // MFMAs of slice kk issue (left to itself hipcc emits read -> lgkmcnt(0) -> 4 MFMA -> read ...)
        // one MFMA k slice: bf16 = 16 elements (v_mfma_f32_32x32x16_bf16: 32 bytes of a row, 16 per lane half); fp8 = 64 elements
        // (v_mfma_f32_32x32x64_f8f6f4, the K = 64 form that runs e4m3 at twice the bf16 rate -- the K = 16 fp8 MFMA runs at the
        // bf16 rate: 64 bytes of a row, 32 per lane half = two 16-byte chunks, each swizzled on its own).  A and B fetch the same
        // chunk positions of their rows, so the pairing of k elements inside the instruction is the same on both sides whatever
        // order the hardware walks them in.
        typedef __attribute__((ext_vector_type(8))) int i32x8;
        constexpr int KK = FP8 ? BK * 2 / 64 : BK / 16;
        using Frag = typename std::conditional<FP8, i32x8, bf16x8>::type;
        Frag af[2][MT], bfr[2][NT4];
        auto frag = [&](const unsigned char *tile, int row, int kk) -> Frag {
            if constexpr (FP8) {
                const int c0 = kk * 4 + (lane >> 5) * 2;
                const u32x4 lo = *(const u32x4 *)(tile + row * (BK * 2) + ((c0 ^ KT::swz(row)) << 4));
                const u32x4 hi = *(const u32x4 *)(tile + row * (BK * 2) + (((c0 + 1) ^ KT::swz(row)) << 4));
                return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            } else {
                return KT::frag(tile, row, kk, lane);
            }
        };
        auto load_frags = [&](int kk, int set) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[set][mt] = frag(At, wm * (TM / 2) + mt * 32 + (lane & 31), kk);
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) bfr[set][n4] = frag(Bt, wn * (TN / WNC) + n4 * 32 + (lane & 31), kk);
        };
        load_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + 1 < KK) load_frags(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the next slice's reads ahead of this slice's MFMAs
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n4 = 0; n4 < NT4; ++n4) {
                    if constexpr (FP8)   // formats 0 / 0 = e4m3 x e4m3; literal zero scale operands select the unscaled encoding
                        acc[mt][n4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[kk & 1][mt], bfr[kk & 1][n4], acc[mt][n4], 0, 0, 0, 0, 0, 0);
                    else
                        acc[mt][n4] = mfma32(af[kk & 1][mt], bfr[kk & 1][n4], acc[mt][n4]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        MPROF_MARK(3);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    MPROF_MARK(4);

    if (p.probe == 4) {  // timing probe: no epilogue
        float t = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[mt][n4][r];
        if (t == 123.456f) p.c[0] = 1;
        return;
    }
    // ---- epilogue: lane owns packed column j = lane&31 of each 32x32 tile and rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    //      C[m,j] = bf16(gelu(acc + bias[idx]) - cache[idx, m])     (csp_mlp_mm1.cu:354-390)
    if constexpr (STAGED) {
        // after the loop: buf = stage after the last computed one = the cache stage (NST = 2) or a free one
        const int last = buf == 0 ? NST - 1 : buf - 1;                  // stage of tile nkb-1
        const int cst = last + NST - 1 >= NST ? last - 1 : last + NST - 1;  // nbuf at kb = nkb-1
        unsigned char *Ct = smem + (FLAT ? 1 : cst) * STAGE, *Ot = smem + (FLAT ? 0 : last) * STAGE;
        // output row r of the stage image (FLAT: the rows that do not fit stage 0 continue behind the cache block)
        auto ot_row = [&](int r) { return Ot + r * (TN * 2) + ((FLAT && r >= STAGE / (TN * 2)) ? EPI : 0); };
        constexpr int LPO = TN * 2 / 16;  // 16-byte chunks per output row
        wait_vmcnt<0>();
        __syncthreads();  // cache block landed; every wave is done reading the last tile
        // The output stage is plain row-major: a ds_write_b16 puts 32 consecutive columns of one row (64 contiguous bytes) per half
        // wave, and the 16-byte read-back below walks whole rows, so neither side needs a swizzle (the cache block does: its 32 lanes
        // of a read hit 32 rows of the [column][m] image at one m).  Two values per instruction wherever the ISA has a packed form.
        // The arithmetic is specialised on update_cache (three straight-line copies: with a branch per element group hipcc serialises the groups,
        // one exposed LDS round trip each) and reads the cache values of a whole 32-column tile ahead of that tile's arithmetic.
        auto arith = [&](auto upd_) {
            constexpr int UPD = decltype(upd_)::value;
            u32x2 cv[2][MT][4];
            auto cptr = [&](int n4, int mt, int q4) {
                const int jl = wn * (TN / WNC) + n4 * 32 + (lane & 31), ml = wm * (TM / 2) + mt * 32 + q4 * 8 + (lane >> 5) * 4;
                return Ct + jl * (TM * 2) + (((ml >> 3) ^ (jl & (LPR - 1))) << 4) + (ml & 7) * 2;
            };
            auto load = [&](int n4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) cv[n4 & 1][mt][q4] = *(const u32x2 *)cptr(n4, mt, q4);
            };
            load(0);
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) {
                if (n4 + 1 < NT4) load(n4 + 1);
                const int jl = wn * (TN / WNC) + n4 * 32 + (lane & 31);
                const float bia = bias_v[n4];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int ml = wm * (TM / 2) + mt * 32 + q4 * 8 + (lane >> 5) * 4;
                        const u32x2 c = cv[n4 & 1][mt][q4];
                        const f32x2 c01 = unpack_bf16x2(c[0]), c23 = unpack_bf16x2(c[1]);
                        f32x2 a01 = {acc[mt][n4][q4 * 4 + 0], acc[mt][n4][q4 * 4 + 1]}, a23 = {acc[mt][n4][q4 * 4 + 2], acc[mt][n4][q4 * 4 + 3]};
                        uint32_t d01, d23, n01 = 0, n23 = 0;   // packed deltas as stored (bf16 pairs); the cache block's new values
                        if constexpr (FP8) {
                            // (acc * scale_a) * scale_b + bias in the reference's order -> gelu -> bf16, then a bf16 subtract (csp_mlp_mm1.py:121-133)
                            const f32x2 sav = {sa, sa}, sbv = {sb, sb}, bv = {bia, bia};
                            const uint32_t t01 = pack_bf16x2_v(gelu_tanh2((a01 * sav) * sbv + bv));
                            const uint32_t t23 = pack_bf16x2_v(gelu_tanh2((a23 * sav) * sbv + bv));
                            d01 = pack_bf16x2_v(unpack_bf16x2(t01) - c01), d23 = pack_bf16x2_v(unpack_bf16x2(t23) - c23);
                            if constexpr (UPD == 2) n01 = t01, n23 = t23;   // 2: cache = new activation, what the reference's Triton kernel does (csp_mlp_mm1.py:140)
                        } else {
                            // the bias is already in the sum (SEED_BIAS)
                            d01 = pack_bf16x2_v(gelu_tanh2(a01) - c01), d23 = pack_bf16x2_v(gelu_tanh2(a23) - c23);
                        }
                        uint16_t *op = (uint16_t *)(ot_row(ml) + jl * 2);   // (rows ml .. ml+3: on one side of the split, a multiple of 4)
                        op[0] = (uint16_t)d01, op[TN] = (uint16_t)(d01 >> 16), op[2 * TN] = (uint16_t)d23, op[3 * TN] = (uint16_t)(d23 >> 16);
                        if constexpr (UPD != 0) {
                            // 1: cache += delta in bf16, exactly what csp_scatter_add does (scatter_add.cu:43-98)
                            if constexpr (!(FP8 && UPD == 2))
                                n01 = pack_bf16x2_v(c01 + unpack_bf16x2(d01)), n23 = pack_bf16x2_v(c23 + unpack_bf16x2(d23));
                            *(u32x2 *)cptr(n4, mt, q4) = (u32x2){n01, n23};
                        }
                    }
                }
            }
        };
        if (p.update_cache == 0) arith(ic<0>{});
        else if (p.update_cache == 1) arith(ic<1>{});
        else arith(ic<2>{});
        __syncthreads();
        constexpr int O_INST = EPI / (1024 * NW);  // 1 KiB row-major pieces per wave
#pragma unroll
        for (int i = 0; i < O_INST; ++i) {
            const int r = (w * O_INST + i) * (64 / LPO) + lane / LPO, ch = lane % LPO;
            const u32x4 v = *(const u32x4 *)(ot_row(r) + (ch << 4));
            const int j = n0 + ch * 8;
            uint16_t *cp = p.c + (int64_t)(g * BM + m_off + r) * p.F + j;
            if (j + 8 <= cnt && (p.F & 7) == 0) {
                *(u32x4 *)cp = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (j + e < cnt) cp[e] = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
            }
        }
        if (p.update_cache) {  // the updated cache block goes back the way it came: 1 KiB pieces, TM*2-byte row segments
#pragma unroll
            for (int i = 0; i < C_INST; ++i) {
                const int jj = (w * C_INST + i) * (64 / LPR) + lane / LPR;
                if (n0 + jj < cnt)
                    *(u32x4 *)((unsigned char *)p.cache + coff[i]) = *(const u32x4 *)(Ct + (w * C_INST + i) * 1024 + lane * 16);
            }
        }
    } else {
#pragma unroll
        for (int n4 = 0; n4 < NT4; ++n4) {
            const int j = n0 + wn * (TN / WNC) + n4 * 32 + (lane & 31);
            const bool live = j < cnt;
            const int col = live ? idxg[j] : 0;
            const uint16_t *crow = p.cache + (int64_t)col * p.M;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int m = g * BM + m_off + wm * (TM / 2) + mt * 32 + q4 * 8 + (lane >> 5) * 4;
                    const u32x2 cv = *(const u32x2 *)(crow + m);
                    const float c0 = __uint_as_float(cv[0] << 16), c1 = __uint_as_float(cv[0] & 0xffff0000u);
                    const float c2 = __uint_as_float(cv[1] << 16), c3 = __uint_as_float(cv[1] & 0xffff0000u);
                    const float x0 = gelu_tanh(acc[mt][n4][q4 * 4 + 0]) - c0;   // (bias already in the sum)
                    const float x1 = gelu_tanh(acc[mt][n4][q4 * 4 + 1]) - c1;
                    const float x2 = gelu_tanh(acc[mt][n4][q4 * 4 + 2]) - c2;
                    const float x3 = gelu_tanh(acc[mt][n4][q4 * 4 + 3]) - c3;
                    if (live) {
                        uint16_t *cp = p.c + (int64_t)m * p.F + j;
                        cp[0] = f32_to_bf16_bits(x0);
                        cp[(int64_t)p.F] = f32_to_bf16_bits(x1);
                        cp[2 * (int64_t)p.F] = f32_to_bf16_bits(x2);
                        cp[3 * (int64_t)p.F] = f32_to_bf16_bits(x3);
                    }
                }
            }
        }
    }
#ifdef MLP_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MPROF_ABS(2);
    MPROF_MARK(5);   // epilogue (mark 4 = loop exit edge)
    MPROF_END(w, nkb);
}

template <int BN, int BK, int NST, int WPS, bool FP8 = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, WPS) void mm1_kernel(const Mm1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MPROF_ABS(0);
    constexpr int NSUB = NW == 4 ? 2 * (BN / 64) : 1;  // 64 x 64 sub-tiles per tile (the 8-wave form has no tail split)
    // PERSISTENT workgroups: the grid is the resident slots (WPS per CU); a workgroup walks its XCD's tile list with the stride of the
    // XCD's slots -- the same tile-to-slot order a one-tile-per-workgroup grid is dispatched in, without the relaunch between tiles
    // (kernel arguments, the live-tile map, wave start-up: 5.4 k of a 41 k-cycle tile at the Wan2.1 fp8 shape, tools/mlp_prof.py) and
    // with the previous tile's stores draining under the next tile's index loads.  (Requesting the NEXT tile's gather indices during
    // the current tile -- one exposed round trip less -- measured no gain at the fp8 shape, 209 vs 209-224 us, for 40 more registers.)
    const TilePlan pl = plan_tiles<BN>(p.counts, p.M / BM, p.NT, p.NR, p.slots_per_xcd, NSUB);
    const int nslots = pl.slots(), stride = (int)(gridDim.x >> 3);
    for (int slot = blockIdx.x >> 3; slot < nslots; slot += stride) {
        const TileMap tm = tile_at(pl, slot);
        if (!tm.live) continue;
        const int g = tm.g;
        const int cnt = p.counts[g];
        if (tm.sub < 0) {
            const int n0 = tm.nt * BN;
            if (n0 >= cnt) continue;  // tiles past counts[g] are skipped (csp_mlp_mm1.cu:233-243)
            mm1_tile<BM, BN, BK, NST, FP8, NW>(p, smem, g, 0, n0, cnt);
        } else if constexpr (NW == 4) {
            constexpr int SUB_NST = (NST * (BM + BN)) / 128;  // same LDS bytes, stages of 64 + 64 rows
            const int n0 = tm.nt * BN + (tm.sub >> 1) * 64;
            if (n0 >= cnt || p.probe == 3) continue;  // probe 3: time the launch without its tail
            mm1_tile<64, 64, BK, (SUB_NST > 4 ? 4 : SUB_NST), FP8>(p, smem, g, (tm.sub & 1) * 64, n0, cnt);
        }
        __syncthreads();   // the next tile's DMA lands where this tile's epilogue was reading
    }
}

// The tile shapes and producer / consumer forms measured against the shipped kernels (docs/EXPERIMENTS_r04.md / _r05.md) are
// not part of the product library: tools/probes/mm1_forms/build.sh compiles this file with -DCHIPMUNK_MM1_PROBES into
// tools/bin/forms/libchipmunk_hip.so, where options mm1_variant / mm2_variant select them (same parity tests, tests/test_gpu_mlp_forms.py).
#ifdef CHIPMUNK_MM1_PROBES
#include "mlp_pc.h"
#include "mlp_pp.h"
#endif

// ------------------------------------------------------------------------------------------------ mm2
struct Mm2Params {
    const uint16_t *a, *b;  // a = packed [M,F], b = fc2^T [F,N2]
    uint16_t *c;            // [M,N2], accumulated in place
    const int32_t *indices, *counts;
    int M, F, N2, NT, NR, probe;
    int cus_per_xcd;   // > 0: length-aware placement of a one-round launch (see mm2_kernel), 0: dispatch order = tile order
};

template <int BN, int BK, int NST, int WPS, int NW = 4>
__global__ __launch_bounds__(NW == 8 ? 512 : 256, WPS) void mm2_kernel(const Mm2Params p) {
    static_assert(NW == 4 || NW == 8, "4 waves (2 x 2) or 8 waves (2 x 4)");
    constexpr int WNG = NW / 2;                 // waves along n; wave tile = 64 rows x BN/WNG columns
    using KT = KTile<BK>;
    constexpr int A_TILE = BM * BK * 2, B_TILE = BK * BN * 2, STAGE = A_TILE + B_TILE;
    constexpr int A_INST = A_TILE / (1024 * NW), B_INST = B_TILE / (1024 * NW);
    constexpr int NT4 = BN / WNG / 32;
    static_assert(A_INST >= 1 && B_INST >= 1 && NT4 >= 1, "tile too small for this many waves");
    constexpr int BROWB = BN * 2;            // bytes per gathered fc2^T row slice
    constexpr int BCPR = BN / 8;             // 16-byte chunks per row
    constexpr int BRPI = 1024 / BROWB;       // rows per DMA instruction (2 for BN=256, 4 for BN=128)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WNG, wn = w % WNG;

    // all column tiles of fc2^T are live (N2 is dense); the map only reorders them for L2 reuse (see map_tile)
    const int G = p.M / BM;
    const int xcdq = (G * p.NT) >> 3, xcdr = (G * p.NT) & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= xcdq + (xcd < xcdr ? 1 : 0)) return;
    const int NR = p.NR;
    const int mine = xcdq + (xcd < xcdr ? 1 : 0);
    const int tbase = xcd < xcdr ? xcd * (xcdq + 1) : xcdr * (xcdq + 1) + (xcd - xcdr) * xcdq;
    auto group_of = [&](int tt) {
        const int nb_ = tt / (G * NR), rem_ = tt - nb_ * (G * NR);
        return rem_ / min(NR, p.NT - nb_ * NR);
    };
    int t = tbase + slot;
    // Length-aware placement of a ONE-ROUND launch (round 6).  A tile's k loop is as long as its group's kept count, and a launch of
    // `mine` tiles per XCD on 2 x cus_per_xcd slots lasts as long as its slowest tile.  Workgroups reach the CUs in dispatch order
    // (tools/probes/dispatch_census.hip: local index l lands on CU l mod cus_per_xcd, so l and l + cus share a CU and the indices
    // [mine - cus, cus) have a CU to themselves -- observed, a speed heuristic only), and a workgroup alone on its CU runs ~25 % faster.
    // So this XCD's tiles are ranked by length (longest first, ties by tile order) and handed out by position: the lone positions
    // take the longest tiles, pair (l, l + cus) takes the next longest together with the shortest.  Same tiles per XCD (same L2
    // working set), same arithmetic per tile: only who computes what moves.
    const int cap = p.cus_per_xcd;
    if (cap > 0 && mine > cap && mine <= 64 && mine <= 2 * cap) {
        const int nl = 2 * cap - mine;                      // lone positions: [mine - cap, cap)
        const int want = slot < mine - cap ? nl + slot      // first member of a pair: the longest of what the lone ones left
                         : slot < cap      ? slot - (mine - cap)   // alone on its CU: the longest of all
                                           : mine - 1 - (slot - cap);   // second member (partner of slot - cap): the shortest
        const int len = lane < mine ? p.counts[group_of(tbase + lane)] : -1;
        int rank = 0;
        for (int j = 0; j < mine; ++j) {
            const int lj = __shfl(len, j);
            rank += (lj > len || (lj == len && j < lane)) ? 1 : 0;
        }
        const unsigned long long hit = __ballot(lane < mine && rank == want);
        t = tbase + (int)__builtin_ctzll(hit);
    }
    const int nb = t / (G * NR), rem = t - nb * (G * NR);
    const int nr = min(NR, p.NT - nb * NR);
    const int g = rem / nr, nt = nb * NR + rem - g * nr;
    const int cnt = p.counts[g];
    const int n0 = nt * BN;
    const int ncols = min(BN, p.N2 - n0);  // N2 is a multiple of 8 (checked on the host)
    const int32_t *idxg = p.indices + (int64_t)g * p.F;
    const int nkb = (cnt + BK - 1) / BK;
    if (nkb == 0) return;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a), rb = make_rsrc(p.b);
    uint32_t aoff[A_INST];  // byte offsets
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = KT::lane_row(w * A_INST + i, lane);
        aoff[i] = ((uint32_t)(g * BM + row) * p.F + KT::src_chunk_elems(row, lane)) * 2u;
    }
    // The gather keys of a tile are wave-uniform per DMA row, so they are fetched with SCALAR loads (lgkmcnt): the
    // vector-memory counter then only counts LDS-DMA and the counted vmcnt pipeline below stays intact.  The keys of
    // the tile issued NEXT iteration are fetched one iteration ahead so their latency hides behind the MFMAs.
    constexpr int NKEY = B_INST * BRPI;          // keys per wave per tile (8): consecutive entries of the index list
    static_assert(NKEY % 4 == 0, "a wave's keys are fetched as whole s_load_dwordx4/x8");
    int keys[NKEY];
    auto load_keys = [&](int kb) {
        // counts are multiples of 8 and so is the block start: the block is live or dead as a whole, which lets the
        // compiler fetch it with ONE wide scalar load instead of NKEY single ones
        const int base = kb * BK + w * NKEY;
        // through the CONSTANT address space: the compiler then emits s_load (lgkmcnt).  As a plain global pointer it loaded the keys with
        // a vector global_load -- on the vmcnt counter, so the use of a key one k step later drained every LDS-DMA in flight
        // (s_waitcnt vmcnt(0) in the middle of the ring: the prefetch depth of the ring was never there)
        const __attribute__((address_space(4))) int32_t *kp =
            (const __attribute__((address_space(4))) int32_t *)(idxg + __builtin_amdgcn_readfirstlane(base < cnt ? base : 0));
#pragma unroll
        for (int j = 0; j < NKEY; ++j) keys[j] = kp[j];
    };
    const int rsel = lane / BCPR;                  // which of the instruction's BRPI rows this lane stages
    auto issue = [&](int kb, int buf) {
        unsigned char *st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
            if (!(p.probe & 32)) blds16(ra, aoff[i], kb * BK * 2, st + (w * A_INST + i) * 1024);   // (timing probe 32: no A pieces)
#pragma unroll
        for (int i = 0; i < B_INST; ++i) {
            if (p.probe & 64) continue;                                                        // (timing probe 64: no B pieces)
            int key = keys[i * BRPI];
#pragma unroll
            for (int rr = 1; rr < BRPI; ++rr) key = rsel == rr ? keys[i * BRPI + rr] : key;
            const int r = (w * B_INST + i) * BRPI + rsel;
            int chunk = (lane % BCPR) ^ ((r & 3) << 2);
            chunk = chunk * 8 < ncols ? chunk : 0;  // partial last column tile: stay inside the row
            blds16(rb, ((uint32_t)key * p.N2 + n0 + chunk * 8) * 2u, 0, st + A_TILE + (w * B_INST + i) * 1024);
        }
    };

    f32x16 acc[NT4][2];  // [n tile][m tile]: MFMA rows = n (fc2^T via transpose reads), MFMA cols = m
#pragma unroll
    for (int n4 = 0; n4 < NT4; ++n4)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n4][mt][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nkb) {
            load_keys(s);
            issue(s, s);
        }
    if (NST - 1 < nkb) load_keys(NST - 1);
    const int li = lane & 15, grp = lane >> 4;
    int buf = 0, nbuf = NST - 1;
    MPROF_DECL;
    MPROF_START();
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + NST - 1 <= nkb) wait_vmcnt<(NST - 2) * (A_INST + B_INST)>();
        else wait_vmcnt<0>();
        MPROF_MARK(0);
        __builtin_amdgcn_s_barrier();
        MPROF_MARK(1);
        // An 8-wave workgroup puts waves w and w + 4 on the same SIMD, locked to each other by the barrier: both would issue their DMA
        // pieces first and want the matrix pipe afterwards.  The second wave of each SIMD issues its pieces AFTER its MFMAs instead, so
        // the pair is in complementary phases (same-box A/B, FLUX shape: 154.8 -> 148.7 us; p.probe & 8 switches it off).
        const bool late = NW == 8 && !(p.probe & 8) && w >= NW / 2;
        if (!late && kb + NST - 1 < nkb) {
            issue(kb + NST - 1, nbuf);
            if (kb + NST < nkb) load_keys(kb + NST);  // (moving these behind the MFMAs measured 15 % slower)
        }
        MPROF_MARK(2);
        // Operand fragments by inline-asm LDS reads with hand-counted lgkmcnt (EVERY LDS read of this loop has to stay in asm: one
        // compiler-visible LDS access added here later brings the drain back and is not counted by the waits below -- tests/test_kernel_audit.py
        // checks the loop's ISA for it): hipcc puts s_waitcnt vmcnt(0) in front of every LDS read
        // it knows about while an LDS-DMA is in flight (it cannot tell the ring slots apart), which drained the two stages the ring is
        // meant to keep flying.  Reads return in order; a scalar key load in flight only makes a counted wait conservative.
        constexpr int KK = BK / 16;
        constexpr int RD = 2 + 2 * NT4;   // LDS reads per k slice: two A fragments, two halves per B fragment
        const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + buf * STAGE;
        u32x4 pf[KK][2];
        u32x2 wlo[KK][NT4], whi[KK][NT4];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = wm * 64 + mt * 32 + (lane & 31);
                const uint32_t ad = stage_lds + row * KT::ROWB + (((kk * 2 + (lane >> 5)) ^ KT::swz(row)) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=&v"(pf[kk][mt]) : "v"(ad) : "memory");
            }
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) {
                // lane group grp: n half = grp&1, k half = grp>>1; lane li addresses block row li>>2, cols (li&3)*4
                const int row = kk * 16 + (grp >> 1) * 8 + (li >> 2);
                const int chunk = (wn * (BN / WNG / 8) + n4 * 4 + (grp & 1) * 2 + ((li & 3) >> 1)) ^ ((row & 3) << 2);
                const uint32_t ad = stage_lds + A_TILE + row * BROWB + chunk * 16 + (li & 1) * 8;
                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%c3"
                             : "=&v"(wlo[kk][n4]), "=&v"(whi[kk][n4]) : "v"(ad), "i"(4 * BROWB) : "memory");   // (early clobber: a result
                // register shared with the address would be overwritten by the first read's return if the second one queues behind it)
            }
        }
        if (kb == nkb - 1) {   // packed columns past the count hold garbage: only the last k step can see them (counts are multiples of 8)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bool kdead = kb * BK + kk * 16 + (lane >> 5) * 8 >= cnt;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    asm volatile("" : "+v"(pf[kk][mt]));
                    if (kdead) pf[kk][mt] = (u32x4){0u, 0u, 0u, 0u};
                }
            }
        }
        static_for<0, KK>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            // slice kk has landed once at most the reads of the younger slices are outstanding; the operands pin the MFMAs below the wait
            constexpr int YOUNGER = RD * (KK - 1 - kk);   // (the counter has 4 bits: a smaller number only waits for more)
            asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(YOUNGER > 15 ? 15 : YOUNGER) : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(pf[kk][i]));
#pragma unroll
            for (int i = 0; i < NT4; ++i) asm volatile("" : "+v"(wlo[kk][i]), "+v"(whi[kk][i]));
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) {
                const bf16x8 wfr = __builtin_bit_cast(bf16x8, (u32x4){wlo[kk][n4][0], wlo[kk][n4][1], whi[kk][n4][0], whi[kk][n4][1]});
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[n4][mt] = mfma32(wfr, __builtin_bit_cast(bf16x8, pf[kk][mt]), acc[n4][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (late && kb + NST - 1 < nkb) {
            issue(kb + NST - 1, nbuf);
            if (kb + NST < nkb) load_keys(kb + NST);
        }
        MPROF_MARK(3);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    MPROF_END(w, nkb);

    if (p.probe == 4) {  // timing probe: no epilogue
        float t = 0.f;
#pragma unroll
        for (int n4 = 0; n4 < NT4; ++n4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[n4][mt][r];
        if (t == 123.456f) p.c[0] = 1;
        return;
    }
    // ---- epilogue: lane owns row m = lane&31 of each tile and 4 consecutive n per accumulator quad
    //      C = bf16(acc) + C in bf16  (triton/csp_mlp_mm2.py:100-101)
    constexpr bool STAGED = BM * BN * 2 <= NST * STAGE;
    if constexpr (STAGED) {
        // bf16(acc) goes through the (now free) ring as a row-major [128][BN] tile, 16-byte chunk c of row m stored at
        // chunk c ^ (m & 31) (the 32 lanes of a store hit 32 rows at one column offset); the read-modify-write of C
        // then runs as 16-byte accesses over whole BN*2-byte row segments instead of 8-byte accesses 32 rows apart.
        constexpr int CPRO = BN / 8;  // 16-byte chunks per output row
        __syncthreads();              // every wave is done reading the last operand stage
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int ml = wm * 64 + mt * 32 + (lane & 31);
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int nl = wn * (BN / WNG) + n4 * 32 + q4 * 8 + (lane >> 5) * 4;
                    u32x2 v;
                    v[0] = pack_bf16x2(acc[n4][mt][q4 * 4 + 0], acc[n4][mt][q4 * 4 + 1]);
                    v[1] = pack_bf16x2(acc[n4][mt][q4 * 4 + 2], acc[n4][mt][q4 * 4 + 3]);
                    *(u32x2 *)(smem + ml * (BN * 2) + ((((nl >> 3) ^ (ml & (CPRO - 1) & 31)) << 4) | ((nl & 4) << 1))) = v;
                }
            }
        }
        __syncthreads();
        constexpr int ITEMS = BM * CPRO / (NW * 64);  // 16-byte pieces per thread
        // all of the thread's C pieces are requested before the first is used: written as load -> add -> store per piece, every load
        // sits behind the previous piece's store (same array, may alias) and hipcc drains both with vmcnt(0) -- eight serial HBM round
        // trips at the end of every tile, and the tiles of a launch all end together
        u32x4 olds[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * (NW * 64) + tid;
            const int r = item / CPRO, ch = item % CPRO;
            const int n = n0 + ch * 8;
            olds[it] = n < p.N2 ? *(const u32x4 *)(p.c + (int64_t)(g * BM + r) * p.N2 + n) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * (NW * 64) + tid;
            const int r = item / CPRO, ch = item % CPRO;
            const int n = n0 + ch * 8;
            if (n >= p.N2) continue;  // N2 is a multiple of 8: a chunk is live or dead as a whole
            const u32x4 a = *(const u32x4 *)(smem + r * (BN * 2) + ((ch ^ (r & (CPRO - 1) & 31)) << 4));
            uint16_t *cp = p.c + (int64_t)(g * BM + r) * p.N2 + n;
            const u32x4 old = olds[it];
            u32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[e] = pack_bf16x2(__uint_as_float(a[e] << 16) + __uint_as_float(old[e] << 16),
                                     __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(old[e] & 0xffff0000u));
            *(u32x4 *)cp = out;
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = g * BM + wm * 64 + mt * 32 + (lane & 31);
            uint16_t *crow = p.c + (int64_t)m * p.N2;
#pragma unroll
            for (int n4 = 0; n4 < NT4; ++n4) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n0 + wn * (BN / WNG) + n4 * 32 + q4 * 8 + (lane >> 5) * 4;
                    if (n >= p.N2) continue;
                    const u32x2 old = *(const u32x2 *)(crow + n);
                    const float a0 = round_bf16(acc[n4][mt][q4 * 4 + 0]), a1 = round_bf16(acc[n4][mt][q4 * 4 + 1]);
                    const float a2 = round_bf16(acc[n4][mt][q4 * 4 + 2]), a3 = round_bf16(acc[n4][mt][q4 * 4 + 3]);
                    u32x2 out;
                    out[0] = pack_bf16x2(a0 + __uint_as_float(old[0] << 16), a1 + __uint_as_float(old[0] & 0xffff0000u));
                    out[1] = pack_bf16x2(a2 + __uint_as_float(old[1] << 16), a3 + __uint_as_float(old[1] & 0xffff0000u));
                    *(u32x2 *)(crow + n) = out;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ scatter-add
// unpacked[idx[g,c], g*128 + r] += packed[g*128 + r, c]   (scatter_add.cu:43-98).  One workgroup = (group, 64 packed
// columns): the 128x64 packed tile is transposed through LDS so both the read (128 B per row) and the read-modify-write
// of the column-major cache (256 B per column) are full-line accesses.  Every (group, column) pair is owned by exactly
// one workgroup, so no atomics are needed (the reference needs TMA reduce-add only because of its thread mapping).
constexpr int SC_COLS = 64;
constexpr int SC_LD = 136;  // padded row length of the transposed tile (bf16 elements)

__global__ __launch_bounds__(256) void scatter_add_kernel(const uint16_t *packed, uint16_t *unpacked,
                                                          const int32_t *indices, const int32_t *counts, int M, int F) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[SC_COLS * SC_LD];
    const int g = blockIdx.y, c0 = blockIdx.x * SC_COLS;
    const int cnt = counts[g];
    if (c0 >= cnt) return;
    const int tid = threadIdx.x;
    // load: 128 rows x 8 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = i * 256 + tid;
        const int r = item >> 3, ch = item & 7;
        const u32x4 v = *(const u32x4 *)(packed + (int64_t)(g * 128 + r) * F + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[(ch * 8 + 2 * e) * SC_LD + r] = (uint16_t)(v[e] & 0xffffu);
            tile[(ch * 8 + 2 * e + 1) * SC_LD + r] = (uint16_t)(v[e] >> 16);
        }
    }
    __syncthreads();
    // accumulate: 64 columns x 16 chunks of 8 rows (the four cache pieces of a thread are requested before the first is used: a load behind the
    // previous piece's store would wait for both)
    u32x4 olds[4];
    const uint16_t *src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = i * 256 + tid;
        const int c = item >> 4, ch = item & 15;
        src[i] = nullptr;
        olds[i] = (u32x4){0u, 0u, 0u, 0u};
        if (c0 + c >= cnt) continue;
        const int col = indices[(int64_t)g * F + c0 + c];
        src[i] = unpacked + (int64_t)col * M + g * 128 + ch * 8;
        olds[i] = *(const u32x4 *)src[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = i * 256 + tid;
        const int c = item >> 4, ch = item & 15;
        if (src[i] == nullptr) continue;
        const u32x4 old = olds[i];
        const u32x4 add = *(const u32x4 *)(tile + c * SC_LD + ch * 8);
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[e] = pack_bf16x2(__uint_as_float(old[e] << 16) + __uint_as_float(add[e] << 16),
                                 __uint_as_float(old[e] & 0xffff0000u) + __uint_as_float(add[e] & 0xffff0000u));
        *(u32x4 *)const_cast<uint16_t *>(src[i]) = out;
    }
}

int check_mlp_common(int M, int F, const int32_t *indices, const int32_t *counts) {
    CM_CHECK(indices && counts, "mlp: indices / counts missing");
    CM_CHECK(M > 0 && M % BM == 0, "mlp: M must be a positive multiple of 128 (got %d)", M);
    CM_CHECK(F > 0 && F % 64 == 0, "mlp: F must be a positive multiple of 64 (got %d)", F);
    return CHIPMUNK_OK;
}

int launch_scatter_add(const void *packed, void *unpacked, const int32_t *indices, const int32_t *counts, int M, int F,
                       hipStream_t s) {
    hipLaunchKernelGGL(scatter_add_kernel, dim3(F / SC_COLS, M / BM), dim3(256), 0, s, (const uint16_t *)packed,
                       (uint16_t *)unpacked, indices, counts, M, F);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

template <int BN, int BK, int NST, int WPS, int NW = 4>
int launch_mm2_variant(const Mm2Params &p0, hipStream_t s) {
    constexpr int LDS = NST * (BM * BK * 2 + BK * BN * 2);
    auto kern = mm2_kernel<BN, BK, NST, WPS, NW>;
    static uint64_t lds_set = 0;
    ensure_dynamic_lds((const void *)kern, LDS, lds_set);
    Mm2Params p = p0;
    p.NT = (p.N2 + BN - 1) / BN;
    p.NR = chipmunk_get_option("mm2_nr") > 0 ? chipmunk_get_option("mm2_nr") : 4;
    if (p.NR > p.NT) p.NR = p.NT;
    // length-aware placement: two-workgroups-per-CU forms only (what the census measured); option mm2_order = 1 keeps tile order
    p.cus_per_xcd = (WPS == 2 && !chipmunk_get_option("mm2_order")) ? device_cu_count() / 8 : 0;
    hipLaunchKernelGGL(kern, dim3((((p.M / BM) * p.NT + 7) / 8) * 8), dim3(NW * 64), LDS, s, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

int launch_mm2(const void *a, const void *b, void *c, const int32_t *indices, const int32_t *counts, int M, int F, int N2,
               hipStream_t s) {
    CM_CHECK(N2 > 0 && N2 % 8 == 0, "mm2: N2 must be a positive multiple of 8 (got %d)", N2);
    CM_CHECK((int64_t)M * F < (1ll << 31) && (int64_t)F * N2 < (1ll << 31), "mm2: M*F or F*N2 too large for 32-bit offsets");
    Mm2Params p = {(const uint16_t *)a, (const uint16_t *)b, (uint16_t *)c, indices, counts, M, F, N2, 0, 0, chipmunk_get_option("mm1_probe")};
#ifdef CHIPMUNK_MM1_PROBES
    switch (chipmunk_get_option("mm2_variant")) {
        case 1: return launch_mm2_variant<256, 64, 2, 1>(p, s);
        case 2: return launch_mm2_variant<128, 64, 2, 2>(p, s);
        case 3: return launch_mm2_variant<128, 64, 3, 1>(p, s);
        case 5: return launch_mm2_variant<256, 64, 3, 1>(p, s);
        case 6: return launch_mm2_variant<128, 32, 4, 2>(p, s);
        case 7: return launch_mm2_variant<128, 32, 3, 3>(p, s);
        case 8: return launch_mm2_variant<256, 32, 3, 3>(p, s);
        case 9: return launch_mm2_variant<256, 64, 2, 2>(p, s);
        case 10: return launch_mm2_variant<256, 64, 3, 1, 8>(p, s);
        case 11: return launch_mm2_variant<256, 64, 2, 1, 8>(p, s);
        case 12: return launch_mm2_variant<256, 32, 3, 2, 8>(p, s);
        case 13: return launch_mm2_variant<256, 32, 4, 1, 8>(p, s);
        case 14: return launch_mm2_variant<256, 32, 3, 2>(p, s);  // the 4-wave form of the default
        case 15: return launch_mm2_variant<512, 32, 4, 2, 8>(p, s);  // 128 x 512 tiles, one workgroup per CU, wave tile 64 x 128
        case 16: return launch_mm2_variant<512, 32, 3, 2, 8>(p, s);
        default: break;
    }
#endif
    // 8 waves (2 x 4, 64 x 64 per wave, 4 waves per SIMD) x 2 workgroups per CU: equal to the 4-wave form in
    // isolation, 2 % faster inside the bench loop (A/B on one box: 170.5 -> 167.3 us, twice)
    return launch_mm2_variant<256, 32, 3, 2, 8>(p, s);
}

template <int BN, int BK, int NST, int WPS, bool FP8 = false, int NW = 4>
int launch_mm1_variant(const Mm1Params &p0, hipStream_t s, bool *cache_updated = nullptr) {
    constexpr int STAGE = BM * BK * 2 + BN * BK * 2, LDS = NST * STAGE, EPI = BM * BN * 2;
    // mm1_tile's staged epilogue (the one that can scatter): one stage each for cache block and outputs, or the FLAT layout
    constexpr bool STAGED = EPI <= STAGE || (NW == 8 && EPI <= (NST - 1) * STAGE && 2 * EPI <= NST * STAGE && STAGE % (BN * 2) == 0);
    auto kern = mm1_kernel<BN, BK, NST, WPS, FP8, NW>;
    static uint64_t lds_set = 0;
    ensure_dynamic_lds((const void *)kern, LDS, lds_set);
    Mm1Params p = p0;
    if (!STAGED) p.update_cache = 0;
    if (cache_updated) *cache_updated = p.update_cache != 0;
    p.NT = (p.F + BN - 1) / BN;
    p.NR = chipmunk_get_option("mm1_nr") > 0 ? chipmunk_get_option("mm1_nr") : 4;
    if (p.NR > p.NT) p.NR = p.NT;
    // tail split: WPS workgroups per CU are resident; an XCD's leftover tiles are handed out as sub-tiles at the end of its list.  The
    // persistent grid and the split both count on WPS workgroups REALLY fitting a CU: variants whose LDS does not allow it
    // (<128,64,2,3> / <128,64,2,4>: 3-4 x 64 KiB) get the count their LDS allows -- the output is right either way, the tail balance is not
    constexpr int WPS_FIT = (160 * 1024 / LDS) < WPS ? (160 * 1024 / LDS) : WPS;
    static_assert(WPS_FIT >= 1, "a variant's ring must fit the 160 KiB of a CU");
    const int resident_per_xcd = WPS_FIT * device_cu_count() / 8;
    p.slots_per_xcd = (chipmunk_get_option("mm1_no_split") || NW != 4) ? 0 : resident_per_xcd;
    // persistent grid: the resident slots, or fewer when the launch has fewer tiles than slots (every tile gets its own workgroup)
    const int tiles_per_xcd = ((p.M / BM) * p.NT + 7) / 8;
    const int per_xcd = tiles_per_xcd < resident_per_xcd ? tiles_per_xcd : resident_per_xcd;
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(NW * 64), LDS, s, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

}  // namespace

namespace {
int mm1_entry(const void *a, const void *b, void *c, const void *bias, void *pa_cache, const int32_t *indices,
              const int32_t *counts, int M, int K, int F, hipStream_t stream, int update_cache, bool *cache_updated) {
    CM_CHECK(a && b && c && bias && pa_cache, "csp_mlp_mm1: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    CM_CHECK(K > 0 && K % 64 == 0, "csp_mlp_mm1: K must be a positive multiple of 64 (got %d)", K);
    CM_CHECK((int64_t)F * K < (1ll << 31) && (int64_t)M * K < (1ll << 31) && (int64_t)F * M < (1ll << 31),
             "csp_mlp_mm1: operand too large for 32-bit offsets");
    Mm1Params p = {(const uint16_t *)a, (const uint16_t *)b, (const uint16_t *)bias, (uint16_t *)pa_cache,
                   (uint16_t *)c, indices, counts, M, K, F, 0, 0, chipmunk_get_option("mm1_probe"), 0, update_cache, nullptr, nullptr};
#ifdef CHIPMUNK_MM1_PROBES
    switch (chipmunk_get_option("mm1_variant")) {
        case 1: return launch_mm1_variant<256, 64, 2, 1>(p, stream, cache_updated);
        case 3: return launch_mm1_variant<128, 64, 3, 1>(p, stream, cache_updated);
        case 4: return launch_mm1_variant<256, 32, 3, 2>(p, stream, cache_updated);
        case 5: return launch_mm1_variant<256, 64, 3, 1>(p, stream, cache_updated);
        case 6: return launch_mm1_variant<128, 32, 4, 2>(p, stream, cache_updated);
        case 7: return launch_mm1_variant<128, 32, 3, 3>(p, stream, cache_updated);
        case 8: return launch_mm1_variant<128, 64, 2, 3>(p, stream, cache_updated);
        case 9: return launch_mm1_variant<128, 64, 2, 4>(p, stream, cache_updated);
        case 10: return launch_mm1_variant<256, 64, 3, 1, false, 8>(p, stream, cache_updated);
        case 20:   // producer / consumer form (mlp_pc.h); needs two k steps
            if (K >= 128) {
                if (cache_updated) *cache_updated = update_cache != 0;
                return launch_mm1pc<false>(p, stream);
            }
            [[fallthrough]];
        case 21:   // producer / consumer form with the DMA stream running across tile boundaries (mlp_pp.h); needs six k steps
            if (K >= 384 && chipmunk_get_option("mm1_variant") == 21) {
                if (cache_updated) *cache_updated = update_cache != 0;
                return launch_mm1pp<false>(p, stream);
            }
            [[fallthrough]];
        default: break;
    }
#endif
    return launch_mm1_variant<128, 64, 2, 2>(p, stream, cache_updated);  // measured best (profiles/r01_*); the library's one GEMM1 form
}
}  // namespace

extern "C" int chipmunk_csp_mlp_mm1(const void *a, const void *b, void *c, const void *bias, const void *pa_cache,
                                    const int32_t *indices, const int32_t *counts, int M, int K, int F, void *stream) {
    return mm1_entry(a, b, c, bias, const_cast<void *>(pa_cache), indices, counts, M, K, F, (hipStream_t)stream, 0, nullptr);
}

extern "C" int chipmunk_csp_mlp_mm1_scatter(const void *a, const void *b, void *c, const void *bias, void *pa_cache,
                                            const int32_t *indices, const int32_t *counts, int M, int K, int F,
                                            void *stream) {
    bool done = false;
    if (int e = mm1_entry(a, b, c, bias, pa_cache, indices, counts, M, K, F, (hipStream_t)stream, 1, &done)) return e;
    // tile shapes whose epilogue does not hold the cache block in LDS: the separate scatter-add kernel
    return done ? CHIPMUNK_OK : launch_scatter_add(c, pa_cache, indices, counts, M, F, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_scatter_add(const void *packed, void *unpacked_colmajor, const int32_t *indices,
                                        const int32_t *counts, int M, int F, int num_sms, void *stream) {
    (void)num_sms;
    CM_CHECK(packed && unpacked_colmajor, "csp_scatter_add: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    return launch_scatter_add(packed, unpacked_colmajor, indices, counts, M, F, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_mlp_mm2(const void *mma_a, const void *mma_b, void *mma_c, const int32_t *indices,
                                    const int32_t *counts, int M, int F, int N2, void *stream) {
    CM_CHECK(mma_a && mma_b && mma_c, "csp_mlp_mm2: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    return launch_mm2(mma_a, mma_b, mma_c, indices, counts, M, F, N2, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_mlp_mm2_and_scatter_add(const void *packed, void *unpacked_colmajor,
                                                    const int32_t *indices, const int32_t *counts, const void *mma_a,
                                                    const void *mma_b, void *mma_c, int M, int F, int N2,
                                                    int num_sms_scatter_add, void *stream) {
    (void)num_sms_scatter_add;
    CM_CHECK(packed && unpacked_colmajor && mma_a && mma_b && mma_c, "csp_mlp_mm2_and_scatter_add: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    if (int e = launch_scatter_add(packed, unpacked_colmajor, indices, counts, M, F, (hipStream_t)stream)) return e;
    return launch_mm2(mma_a, mma_b, mma_c, indices, counts, M, F, N2, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_mlp_mm1_fp8(const void *a, const void *b, void *c, const void *bias, void *pa_cache,
                                        const int32_t *indices, const int32_t *counts, const float *scale_a,
                                        const float *scale_b, int M, int K, int F, int update_cache, void *stream) {
    CM_CHECK(a && b && c && bias && pa_cache && scale_a && scale_b, "csp_mlp_mm1_fp8: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    CM_CHECK(K > 0 && K % 128 == 0, "csp_mlp_mm1_fp8: K must be a positive multiple of 128 (got %d)", K);
    CM_CHECK(update_cache >= 0 && update_cache <= 2, "csp_mlp_mm1_fp8: update_cache must be 0, 1 or 2");
    CM_CHECK((int64_t)F * K < (1ll << 31) && (int64_t)M * K < (1ll << 31) && (int64_t)F * M < (1ll << 31),
             "csp_mlp_mm1_fp8: operand too large for 32-bit offsets");
    // same tile machinery as the bf16 kernel (buffer-form DMA, tail split, staged epilogue); a k step is 128 fp8 values
    Mm1Params p = {(const uint16_t *)a, (const uint16_t *)b, (const uint16_t *)bias, (uint16_t *)pa_cache, (uint16_t *)c,
                   indices, counts, M, K, F, 0, 0, chipmunk_get_option("mm1_probe"), 0, update_cache == 1 ? 2 : update_cache == 2 ? 1 : 0, scale_a, scale_b};
#ifdef CHIPMUNK_MM1_PROBES
    if (chipmunk_get_option("mm1_variant") == 10) return launch_mm1_variant<256, 64, 3, 1, true, 8>(p, (hipStream_t)stream);
    if (chipmunk_get_option("mm1_variant") == 20 && K >= 256) return launch_mm1pc<true>(p, (hipStream_t)stream);
    if (chipmunk_get_option("mm1_variant") == 21 && K >= 768) return launch_mm1pp<true>(p, (hipStream_t)stream);
#endif
    return launch_mm1_variant<128, 64, 2, 2, true>(p, (hipStream_t)stream);
}
