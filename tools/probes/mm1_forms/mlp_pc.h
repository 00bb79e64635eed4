// GEMM1 in producer / consumer form (included by mlp.hip inside its anonymous namespace; replaces reference csrc/mlp/csp_mlp_mm1.cu:207-390
// and src/chipmunk/triton/csp_mlp_mm1.py:37-164 exactly as mm1_kernel does -- same operands, same epilogue arithmetic).
//
// Why a second form.  A 128-row sparsity group fixes the reuse of a gathered weight row at 128 outputs, so the bytes that have to come
// through the CU's vector-memory path per MFMA are 32 * (1 + 128 / BN) B/clk/CU at the matrix peak: 64 at BN = 128 (the path moves 43-45
// measured, tools/probes/fill_rate.hip), 48 at BN = 256.  A 128 x 256 tile with a ring deep enough to cover the landing of gathered rows
// (two 48 KiB stages in flight + one being read = 144 KiB) leaves room for ONE workgroup per CU -- and eight waves that all issue their DMA
// pieces, then all want the matrix pipe, overlap nothing (mm1_variant 10: 2 000 cycles per k step for 1 024 of MFMA).  Here the roles are
// split: waves 0-3 (one per SIMD) only read fragments and issue MFMAs (wave tile 64 x 128: 6 ds_read_b128 per 8 MFMAs instead of 4 per 4),
// waves 4-7 (their SIMD partners) only issue the LDS-DMA pieces and wait for them, so a DMA issue stall never sits in front of an MFMA.
// One s_barrier per k step hands a landed stage to the consumers and a drained one back to the producers; the consumers arrive at it BEFORE
// issuing the last k slice's MFMAs, so those cover the latency of the next stage's first fragment reads.
//
// Ring: stage s of a tile lives in slot (first + s) % 3 with `first` chosen so that the LAST k step computes out of slot 2: the two slots that
// are free by then take the tile's 64 KiB cache block ([0, 64 Ki), pieces issued by the producers during the last two k steps), the outputs
// leave through [64 Ki, 128 Ki).
constexpr int PC_TM = 128, PC_TN = 256, PC_NST = 3;
constexpr int PC_A = PC_TM * 128, PC_B = PC_TN * 128, PC_STAGE = PC_A + PC_B, PC_LDS = PC_NST * PC_STAGE;
constexpr int PC_EPI = PC_TM * PC_TN * 2;        // bytes of the cache block, and of the output stage
constexpr int PC_PA = PC_A / 4096, PC_PB = PC_B / 4096, PC_PC = PC_EPI / 4096;   // DMA pieces per producer wave: 4 + 8 per k step, 16 cache
static_assert(PC_PA + PC_PB == 12 && PC_PC == 16 && PC_EPI == 65536 && 2 * PC_EPI <= PC_LDS, "piece counts the vmcnt waits below assume");

#ifdef MLP_PROF
// timeline of one mid-grid workgroup (tools/mlp_prof.py --pc): absolute s_memtime stamps of wave 0 (consumer) and wave 4 (producer), tiles 0 and 1
__device__ unsigned long long g_pc_prof[2 * 2 * 16];
#define PCPROF(tile, role, i) do { if (blockIdx.x == 16 && (threadIdx.x & 63) == 0 && (w & 3) == 0 && (tile) < 2) g_pc_prof[(tile) * 32 + (role) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int chipmunk_pc_prof_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_prof), sizeof(g_pc_prof)) == hipSuccess ? 0 : 2;
}
#else
#define PCPROF(tile, role, i)
#endif

struct PcTileWalk {   // the persistent tile walk both roles run in step (same plan, same skips, hence the same barrier count)
    TilePlan pl;
    int stride, first, nkb;
};

// cache-block piece i (of 16) of producer pw: the first 12 of every producer fill [0, 48 Ki) (slot 0, free during the last two k steps), the
// last 4 fill [48 Ki, 64 Ki) (the head of slot 1, free during the last one)
__device__ __forceinline__ int pc_cid(int pw, int i) { return i < 12 ? pw * 12 + i : 48 + pw * 4 + (i - 12); }

// the outputs of a tile leave as 64 row-major 1 KiB pieces (two 512-byte rows each), 8 per wave
__device__ __forceinline__ void pc_store_outputs(const Mm1Params &p, const u32x4 (&ov)[8], int w, int lane, int g, int n0, int cnt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (w * 8 + i) * 2 + (lane >> 5), ch = lane & 31;
        const int j = n0 + ch * 8;
        uint16_t *cp = p.c + (int64_t)(g * BM + r) * p.F + j;
        if (j + 8 <= cnt && (p.F & 7) == 0) {
            *(u32x4 *)cp = ov[i];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (j + e < cnt) cp[e] = (uint16_t)(ov[i][e >> 1] >> ((e & 1) * 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------- producer: the DMA stream
template <bool FP8>
__device__ __forceinline__ void pc_producer(const Mm1Params &p, unsigned char *smem, const PcTileWalk &tw, int w, int lane) {
    using KT = KTile<64>;
    constexpr uint32_t ESZ = FP8 ? 1u : 2u;
    const int pw = w & 3, nkb = tw.nkb, first = tw.first;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a), rb = make_rsrc(p.b), rc = make_rsrc(p.cache);
    unsigned char *Ct = smem, *Ot = smem + PC_EPI;
    int tno = -1;
    for (int slot = blockIdx.x >> 3; slot < tw.pl.mine; slot += tw.stride) {
        const TileMap tm = tile_at(tw.pl, slot);
        const int g = tm.g, n0 = tm.nt * PC_TN;
        const int cnt = p.counts[g];
        if (n0 >= cnt) continue;   // (uniform over the workgroup: csp_mlp_mm1.cu:233-243)
        ++tno;
        PCPROF(tno, 1, 0);
        const int32_t *idxg = p.indices + (int64_t)g * p.F;
        uint32_t aoff[PC_PA], boff[PC_PB], coff[PC_PC];   // per-lane byte offsets of this wave's pieces
#pragma unroll
        for (int i = 0; i < PC_PA; ++i) {
            const int row = KT::lane_row(pw * PC_PA + i, lane);
            aoff[i] = (uint32_t)(g * BM + row) * p.K * ESZ + KT::src_chunk_elems(row, lane) * 2u;
        }
#pragma unroll
        for (int i = 0; i < PC_PB; ++i) {
            const int row = KT::lane_row(pw * PC_PB + i, lane);
            const int j = n0 + row;
            boff[i] = (uint32_t)idxg[j < cnt ? j : n0] * p.K * ESZ + KT::src_chunk_elems(row, lane) * 2u;   // rows past the count re-read a live row
        }
#pragma unroll
        for (int i = 0; i < PC_PC; ++i) {
            const int jj = pc_cid(pw, i) * 4 + (lane >> 4);   // 4 cache rows (256 B each) per 1 KiB piece
            const int j = n0 + jj;
            coff[i] = ((uint32_t)idxg[j < cnt ? j : n0] * p.M + g * BM + (((lane & 15) ^ (jj & 15)) << 3)) * 2u;
        }
        auto issue = [&](int kb) {
            unsigned char *st = smem + ((first + kb) % PC_NST) * PC_STAGE;
#pragma unroll
            for (int i = 0; i < PC_PA; ++i) blds16(ra, (p.probe & 16) ? 0u : aoff[i], (p.probe & 16) ? 0 : kb * 128, st + (pw * PC_PA + i) * 1024);   // (probe 16: one hot line)
#pragma unroll
            for (int i = 0; i < PC_PB; ++i) blds16(rb, (p.probe & 32) ? 0u : boff[i], (p.probe & 32) ? 0 : kb * 128, st + PC_A + (pw * PC_PB + i) * 1024);
        };
        const bool dma = !(p.probe & 1);        // (timing probe 1: no operand pieces)
        PCPROF(tno, 1, 1);
        if (dma) issue(0);
        if (dma) issue(1);
        PCPROF(tno, 1, 2);
        for (int kb = 0; kb < nkb; ++kb) {
            wait_vmcnt<PC_PA + PC_PB>();        // stage kb has landed; the one issued after it (12 pieces, always) may fly on
            __builtin_amdgcn_s_barrier();       // B(kb): consumers may read stage kb; they are done with stage kb-1
            if (kb + 2 < nkb) {
                if (dma) issue(kb + 2);
            } else if (kb + 2 == nkb) {         // slot 0 is free for good: the first 48 KiB of the cache block (12 pieces)
#pragma unroll
                for (int i = 0; i < 12; ++i) blds16(rc, coff[i], 0, smem + pc_cid(pw, i) * 1024);
            } else {                            // slot 1 too: the last 16 KiB
#pragma unroll
                for (int i = 12; i < PC_PC; ++i) blds16(rc, coff[i], 0, smem + pc_cid(pw, i) * 1024);
            }
        }
        PCPROF(tno, 1, 3);
        wait_vmcnt<0>();
        PCPROF(tno, 1, 4);
        __builtin_amdgcn_s_barrier();           // E0: the cache block has landed, slot 2 is drained
        PCPROF(tno, 1, 5);
        __syncthreads();                        // E1: the consumers' deltas and the updated cache block are in LDS
        PCPROF(tno, 1, 6);
        u32x4 ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = *(const u32x4 *)(Ot + (w * 8 + i) * 1024 + lane * 16);
        u32x4 cvv[PC_PC];
        if (p.update_cache) {
#pragma unroll
            for (int i = 0; i < PC_PC; ++i) cvv[i] = *(const u32x4 *)(Ct + pc_cid(pw, i) * 1024 + lane * 16);
        }
        __syncthreads();                        // E2: LDS is free for the next tile's ring; the stores drain under its prologue
        PCPROF(tno, 1, 7);
        if (!(p.probe & 8)) pc_store_outputs(p, ov, w, lane, g, n0, cnt);   // (timing probe 8: no output stores)
        if (p.update_cache) {                   // the updated cache block goes back the way it came: whole 256-byte row segments
#pragma unroll
            for (int i = 0; i < PC_PC; ++i) {
                const int jj = pc_cid(pw, i) * 4 + (lane >> 4);
                if (n0 + jj < cnt) *(u32x4 *)((unsigned char *)p.cache + coff[i]) = cvv[i];
            }
        }
        PCPROF(tno, 1, 8);
    }
}

// ---------------------------------------------------------------------------------------------- consumer: fragments, MFMAs, epilogue
template <bool FP8>
__device__ __forceinline__ void pc_consumer(const Mm1Params &p, unsigned char *smem, const PcTileWalk &tw, int w, int lane) {
    const int l31 = lane & 31, wm = w >> 1, wn = w & 1, nkb = tw.nkb, first = tw.first;   // 2 x 2 waves of 64 x 128
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned char *Ct = smem, *Ot = smem + PC_EPI;
    // fragment addresses: row l31 of a 32-row block, chunk (slice chunk) ^ swz(row); swz depends on l31 only (blocks start at multiples of 32)
    const uint32_t sw = (uint32_t)(l31 >> 1) & 7u;
    const uint32_t abase = lds0 + (uint32_t)l31 * 128u + (uint32_t)wm * 8192u;
    const uint32_t bbase = lds0 + (uint32_t)l31 * 128u + (uint32_t)(PC_A + wn * 16384);
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    constexpr int KK = FP8 ? 2 : 4;                 // MFMA k slices per k step: 32 B (bf16 x 16) or 64 B (fp8 x 64) of every row
    constexpr int RPF = FP8 ? 2 : 1;                // 16-byte reads per fragment
    constexpr int RD = 6 * RPF;                     // LDS reads per slice
    static_assert(RD <= 15, "the counted lgkmcnt wait has four bits");
    float sa = 1.f, sb = 1.f;
    if constexpr (FP8) sa = p.scale_a[0], sb = p.scale_b[0];
    int tno = -1;
    for (int slot = blockIdx.x >> 3; slot < tw.pl.mine; slot += tw.stride) {
        const TileMap tm = tile_at(tw.pl, slot);
        const int g = tm.g, n0 = tm.nt * PC_TN;
        const int cnt = p.counts[g];
        if (n0 >= cnt) continue;
        ++tno;
        PCPROF(tno, 0, 0);
        const int32_t *idxg = p.indices + (int64_t)g * p.F;
        float bias_v[4];
        f32x16 acc[2][4];
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
            const int j = n0 + wn * 128 + n4 * 32 + l31;
            bias_v[n4] = bf16_bits_to_f32(p.bias[idxg[j < cnt ? j : n0]]);
            const float seed = FP8 ? 0.f : bias_v[n4];   // bf16: the bias seeds the sums (csp_mlp_mm1.cu:347-350); fp8 scales the sum first
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][n4][r] = seed;
        }
        u32x4 fa[2][2][RPF], fb[2][4][RPF];
        // one 16-byte fragment read: piece r of the 6 * RPF a slice needs (A fragments first)
        auto read1 = [&](int kb, int kk, int set, int r) {
            const int q = r / 6, f = r % 6;
            const uint32_t st = (uint32_t)((first + kb) % PC_NST) * PC_STAGE;
            // chunk of this lane's half: bf16 slice kk = chunks 2kk, 2kk+1; fp8 slice kk = chunks 4kk .. 4kk+3, two per half
            const uint32_t c = FP8 ? (uint32_t)(kk * 4 + (lane >> 5) * 2 + q) : (uint32_t)(kk * 2 + (lane >> 5));
            const uint32_t co = st + ((c ^ sw) << 4);
            if (f < 2) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(fa[set][f][q]) : "v"(abase + co), "i"(f * 4096) : "memory");
            else asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(fb[set][f - 2][q]) : "v"(bbase + co), "i"((f - 2) * 4096) : "memory");
        };
        auto mfma1 = [&](int set, int i) {
            const int n4 = i >> 1, mt = i & 1;
            if constexpr (FP8) {
                const u32x4 a0 = fa[set][mt][0], a1 = fa[set][mt][1], b0 = fb[set][n4][0], b1 = fb[set][n4][1];
                const i32x8 av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                const i32x8 bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
                acc[mt][n4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[mt][n4], 0, 0, 0, 0, 0, 0);
            } else {
                acc[mt][n4] = mfma32(__builtin_bit_cast(bf16x8, fa[set][mt][0]), __builtin_bit_cast(bf16x8, fb[set][n4][0]), acc[mt][n4]);
            }
        };
        // One k slice: the 8 MFMAs on fragment set `set`, with the reads of the NEXT slice (if any) dropped into the gaps behind the first MFMAs,
        // RPG per gap -- issued in front of the cluster they cost the wave ~100 cycles per slice in which the matrix pipe runs dry.
        auto slice = [&](int set, bool has_next, int nkb_, int nkk_) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < RPF; ++q) asm volatile("" : "+v"(fa[set][mt][q]));
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
                for (int q = 0; q < RPF; ++q) asm volatile("" : "+v"(fb[set][n4][q]));
            constexpr int RPG = 2 * RPF;
            static_for<0, 8>([&](auto ic_) {
                constexpr int i = decltype(ic_)::value;
                mfma1(set, i);
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) {
#pragma unroll
                    for (int r = i * RPG; r < (i + 1) * RPG && r < RD; ++r) read1(nkb_, nkk_, set ^ 1, r);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        PCPROF(tno, 0, 1);
        __builtin_amdgcn_s_barrier();           // B(0)
        PCPROF(tno, 0, 2);
#pragma unroll
        for (int r = 0; r < RD; ++r) read1(0, 0, 0, r);
        for (int kb = 0; kb < nkb; ++kb) {
            if (p.probe & 2) {                  // (timing probe 2: no fragment reads, no MFMAs)
                __builtin_amdgcn_s_barrier();
                continue;
            }
            static_for<0, KK>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int set = kk & 1;     // KK is even: the sets alternate seamlessly across k steps
                // this slice's fragments have landed (nothing younger is outstanding); after the last slice's reads every read of stage kb has
                // returned: hand the slot back (B(kb+1), or E0 after the last k step) BEFORE that slice's MFMAs, which then cover the latency of
                // the next stage's first fragment reads
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (kk + 1 < KK) {
                    slice(set, true, kb, kk + 1);
                } else {
                    __builtin_amdgcn_s_barrier();
                    slice(set, kb + 1 < nkb, kb + 1, 0);
                }
            });
        }
        PCPROF(tno, 0, 3);
        // ---- epilogue (E0 passed above): lane owns packed column jl = lane&31 of each 32 x 32 tile and rows (r&3) + 8*(r>>2) + 4*(lane>>5):
        //      C[m, j] = bf16(gelu(acc + bias[idx]) - cache[idx, m])     (csp_mlp_mm1.cu:354-390; fp8: csp_mlp_mm1.py:121-133)
        if (p.probe & 4) {                      // (timing probe 4: no epilogue arithmetic)
            float t = 0.f;
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[mt][n4][r];
            if (t == 123.456f) Ot[lane] = 1;
        } else
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
            const int jl = wn * 128 + n4 * 32 + l31;
            const float bia = bias_v[n4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int ml = wm * 64 + mt * 32 + q4 * 8 + (lane >> 5) * 4;
                    unsigned char *cp = Ct + jl * 256 + (((ml >> 3) ^ (jl & 15)) << 4) + (ml & 7) * 2;
                    const u32x2 cv = *(const u32x2 *)cp;
                    const f32x2 c01 = unpack_bf16x2(cv[0]), c23 = unpack_bf16x2(cv[1]);
                    f32x2 a01 = {acc[mt][n4][q4 * 4 + 0], acc[mt][n4][q4 * 4 + 1]}, a23 = {acc[mt][n4][q4 * 4 + 2], acc[mt][n4][q4 * 4 + 3]};
                    uint32_t d01, d23, n01 = 0, n23 = 0;
                    if constexpr (FP8) {
                        // (acc * scale_a) * scale_b + bias in the reference's order -> gelu -> bf16, then a bf16 subtract
                        const f32x2 sav = {sa, sa}, sbv = {sb, sb}, bv = {bia, bia};
                        const uint32_t t01 = pack_bf16x2_v(gelu_tanh2((a01 * sav) * sbv + bv));
                        const uint32_t t23 = pack_bf16x2_v(gelu_tanh2((a23 * sav) * sbv + bv));
                        d01 = pack_bf16x2_v(unpack_bf16x2(t01) - c01), d23 = pack_bf16x2_v(unpack_bf16x2(t23) - c23);
                        if (p.update_cache == 2) n01 = t01, n23 = t23;
                    } else {
                        d01 = pack_bf16x2_v(gelu_tanh2(a01) - c01), d23 = pack_bf16x2_v(gelu_tanh2(a23) - c23);
                    }
                    uint16_t *op = (uint16_t *)(Ot + ml * (PC_TN * 2) + jl * 2);
                    op[0] = (uint16_t)d01, op[PC_TN] = (uint16_t)(d01 >> 16), op[2 * PC_TN] = (uint16_t)d23, op[3 * PC_TN] = (uint16_t)(d23 >> 16);
                    if (p.update_cache) {
                        if (!(FP8 && p.update_cache == 2))   // cache += delta in bf16, what csp_scatter_add does (scatter_add.cu:43-98)
                            n01 = pack_bf16x2_v(c01 + unpack_bf16x2(d01)), n23 = pack_bf16x2_v(c23 + unpack_bf16x2(d23));
                        *(u32x2 *)cp = (u32x2){n01, n23};
                    }
                }
            }
        }
        PCPROF(tno, 0, 4);
        __syncthreads();                        // E1
        PCPROF(tno, 0, 5);
        u32x4 ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = *(const u32x4 *)(Ot + (w * 8 + i) * 1024 + lane * 16);
        __syncthreads();                        // E2
        PCPROF(tno, 0, 6);
        if (!(p.probe & 8)) pc_store_outputs(p, ov, w, lane, g, n0, cnt);
        PCPROF(tno, 0, 7);
    }
}

template <bool FP8>
__global__ __launch_bounds__(512, 1) void mm1pc_kernel(const Mm1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PcTileWalk tw;
    tw.pl = plan_tiles<PC_TN>(p.counts, p.M / BM, p.NT, p.NR, 0, 1);
    tw.stride = (int)(gridDim.x >> 3);
    tw.nkb = (int)((uint32_t)p.K * (FP8 ? 1u : 2u) / 128u);
    tw.first = (2 + 3 * PC_NST - (tw.nkb - 1) % PC_NST) % PC_NST;   // slot of k step 0; k step nkb-1 then computes out of slot 2
#ifdef MLP_PROF
    if (blockIdx.x == 16 && threadIdx.x == 0) g_pc_prof[15] = __builtin_amdgcn_s_memtime();
#endif
    // waves w and w + 4 share a SIMD: one consumer and one producer on each
    if (w >= 4) pc_producer<FP8>(p, smem, tw, w, lane);
    else pc_consumer<FP8>(p, smem, tw, w, lane);
}

template <bool FP8>
int launch_mm1pc(const Mm1Params &p0, hipStream_t s) {
    // the precondition lives with the kernel, not with the dispatch switch: k steps of 128 bytes of every row
    CM_CHECK((int)((uint32_t)p0.K * (FP8 ? 1 : 2) / 128) >= 2, "launch_mm1pc: needs at least two k steps (the cache block's pieces ride the last two)");
    auto kern = mm1pc_kernel<FP8>;
    static uint64_t lds_set = 0;
    ensure_dynamic_lds((const void *)kern, PC_LDS, lds_set);
    Mm1Params p = p0;
    p.NT = (p.F + PC_TN - 1) / PC_TN;
    p.NR = chipmunk_get_option("mm1_nr") > 0 ? chipmunk_get_option("mm1_nr") : 2;
    if (p.NR > p.NT) p.NR = p.NT;
    const int per_xcd_max = device_cu_count() / 8;          // one workgroup per CU
    const int tiles_per_xcd = ((p.M / BM) * p.NT + 7) / 8;
    const int per_xcd = tiles_per_xcd < per_xcd_max ? tiles_per_xcd : per_xcd_max;
    hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(512), PC_LDS, s, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}
