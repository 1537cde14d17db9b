// Fused multi-head self-attention for gfx950 (head dim 64, no mask beyond the token count), flash-style.
//
// Replaces the reference's  softmax(q k^T / sqrt(d)) v  (v2_depthanything/components/transformer_block.py:160-166,
// the SDPA call at :164). Inputs come from the QKV GEMM epilogue (gemm.hip, E_QKV):
//   Q  [B,H,npad,64]  bf16, already multiplied by 1/sqrt(64)
//   K  [B,H,npad,64]  bf16
//   Vt [B,H,64,npadv] bf16 (V transposed, token-contiguous; pad columns are zero)
// Output: token-major [B*npad, F] bf16 (column h*64 + d), i.e. directly the A operand of the proj GEMM.
//
// Work split: one workgroup = 128 query rows of one (batch, head); 4 waves x 32 query rows.
// Per 64-key tile (K tile 8 KiB and Vt tile 8 KiB, LDS-DMA'd into a 2-deep ring, XOR-swizzled like the GEMM):
//   S^T = K Q^T   with 32x32x16 bf16 MFMA  (A = K rows from LDS, B = Q^T held in registers)
//         -> each lane owns ONE query (lane&31) and 16 of the 32 keys of a block: row max/sum are
//            in-register reductions plus a single exchange with lane^32 (wavefront softmax).
//   O^T += Vt P^T with the SAME lane->query mapping, so the online-softmax rescale of O is lane-local.
//         The MFMA contraction index is a free permutation, so P^T fragments are used exactly as the
//         S^T accumulators come out (keys {0-3, 8-11} + 4*(lane>>5) per 16-key block) and the Vt fragment
//         is gathered to match with two 8-byte LDS reads - no cross-lane shuffle of P at all.
// x3 mode: hi/lo bf16 planes for Q, K, V and P (3 MFMAs per product) -> fp32-class accuracy.
//
// Template HD = head dim: 64 (DINOv2 / BEiT) or 32 (SwinV2 windows). With HD = 32 a 128-byte LDS row of the K tile holds a
// PAIR of keys, everything else (DMA chunking, XOR swizzle, fragment reads) is the same code.
// Template MODE: 0 plain, 1 = BEiT relative-position bias (LUT gather), 2 = SwinV2 window attention: the "batch" is
// (image, window), scores get the continuous-position bias LUT plus the 0 / -100 shifted-window mask
// (reference v31_swinv2/components/windowed_attention.py:100-123, :394-439) and output rows are scattered back through the
// window -> image token map (window reverse + un-roll, :171-260) so the proj GEMM sees tokens in image order.

#include "mdpt_kernels.h"
#include "mdpt_prof.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

namespace {

constexpr float kLog2e = 1.4426950408889634f;
// Deferred running maximum of the online softmax: the reference point m of a query row only moves when a tile's maximum exceeds it by
// more than kDeferMax (score units; e^5.5 = 2^7.9), so after the first tile or two NO lane of a wave changes m any more and the wave
// skips the 64-wide rescale of O^T for the rest of the key loop (before: the exact running maximum of at least one of a wave's 64
// queries moves in almost every tile, so the skip never fired). p = exp(s - m) <= e^5.5 is harmless in fp32 / bf16 and the final
// O / l is the same number mathematically; a lane's decision depends on its own query only (batch-invariant bits).
constexpr float kDeferMax = 5.5f;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ opx8 cat44(opx4 a, opx4 b) {
    opx8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = a[e]; r[e + 4] = b[e]; }
    return r;
}

// QB = query blocks of 32 per wave: 2 in bf16 mode (64 queries per wave, 256 per workgroup: every K / Vt fragment read
// from LDS and every LDS-DMA'd tile feeds twice the MFMAs - the kernel is vector-memory/LDS bound otherwise), 1 in x3
// mode (register budget: hi+lo planes of Q, K, V and P).
// SPLIT (latency form for small launches, e.g. batch 1): the four waves of a workgroup share ONE block of 32*QB queries and each
// takes every fourth key tile (private K/V ring per wave, no barrier in the loop); the partial (m, l, O) states are merged through
// LDS at the end. A launch that cannot fill the GPU anyway finishes in a quarter of the key-loop time.
// RUN4 (MODE 2, window width % 4 == 0): four consecutive keys of a lane's accumulator quad sit in one row of the window, so their LUT
// indices tq - tk, tq - tk - 1, ... are consecutive: the table is staged REVERSED and one index + two ds_read2_b32 fetch the four
// biases (before: four index subtractions and four single gathers) - the window attention spent more issue slots on the bias
// gather than on the softmax.
template <bool X3, int QB, int MODE, int HD, bool SPLIT = false, bool RUN4 = false>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnParams p) {
    static_assert(!RUN4 || MODE == 2, "the 4-key bias runs exist for the window attention");
    constexpr bool BIAS = MODE != 0;
    // SwinV2 window attention (round 6): the kernel's VALU work per score was bias add + running max + fma + exp + pack against 16 MFMAs per 4096
    // scores, and the SQ counters read its SIMDs ~85 % busy issuing it (profiles/r06_sq_counters_swinl.md). Three of those go away:
    //   LOG2 : the scores arrive in log2 units - the packed logit scale carries log2(e) (mdpt_finalize), the LDS image of the bias table is
    //          multiplied by it - so exp(s - m) is v_exp_f32(s - m) without the multiply;
    //   CINIT: the bias is the C operand of the first MFMA of every score block (read from LDS straight into the accumulators), not an add behind it;
    //   FIXREF (bf16 operands): cosine attention bounds every score of head h by M_h = logit_scale_h (1 + 2^-6) + 16 (|cos| <= 1 up to operand
    //          rounding, bias = 16 sigmoid(.) < 16: relative_positional_encoder.py:60-93, windowed_attention.py:100-119), and a query's own key scores
    //          at least logit_scale_h (1 - 2^-7). With the constant reference M_h folded into the table, p = exp(s - M_h) lies in [e^-17.6, 1.2] for a
    //          row's maximum - harmless in fp32 / bf16, whose exponent ranges agree - and the running maximum, its cross-lane swap, the exp of the
    //          rescale factor and the O^T rescale are gone. O / l is the same number mathematically; bits depend on the query alone.
    //          The fp16 operand build keeps the running maximum: an fp16 P of 2^-25 would be subnormal.
    // (CINIT also serves BEiT's table bias, MODE 1: the gather lands in the accumulators, 64 adds per lane and tile fewer.)
    constexpr bool LOG2 = MODE == 2, CINIT = BIAS, FIXREF = MODE == 2 && !MDPT_OP_IS_F16;
    constexpr float kExpScale = LOG2 ? 1.0f : kLog2e;  // exp(x) = v_exp_f32(x * kExpScale)
    constexpr int NPL = X3 ? 2 : 1;           // planes per operand
    constexpr int TILE = 64 * HD * 2;         // one [64 keys][HD] (or [HD][64 keys]) bf16 tile
    constexpr int STAGE = 2 * NPL * TILE;     // K planes then Vt planes
    constexpr int KS = HD / 16, DB = HD / 32, CH = HD / 32;  // MFMA k-steps of S, 32-row blocks of O^T, 1 KiB DMA chunks per wave
    constexpr int QPW = 32 * QB, QPB = SPLIT ? QPW : 4 * QPW;
    constexpr int RINGS = SPLIT ? 4 : 1;  // private ring per wave in the split form
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // 1-D grid, XCD-aware: the dispatcher sends block id to XCD id%8, and all q-tiles of one (batch, head) re-read the
    // same K/V (332 KB at N=1297) -> give every (batch, head) to ONE XCD so its K/V stay in that XCD's 4 MB L2
    // (measured before: 1.49 GB fetched per launch vs 0.26 GB algorithmic).
    const int nq = (p.npad + QPB - 1) / QPB, nbh = p.B * p.heads;
    int qt, bhi;
    if ((nbh & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        // A last q-tile with a single active wave (N = 1297: 17 queries) holds a workgroup slot for a full pass over K/V while three
        // of its waves idle. Dispatch those tiles LAST, together: their lone waves then run without a partner on their SIMDs
        // (the rounds before them are made of full tiles only).
        const bool tail_last = !SPLIT && nq > 1 && p.npad - (nq - 1) * QPB <= QPW && p.tail_last;
        if (tail_last) {
            const int per_xcd = nbh >> 3, full = per_xcd * (nq - 1);
            if (slot < full) {
                bhi = (slot / (nq - 1)) * 8 + xcd;
                qt = slot - (slot / (nq - 1)) * (nq - 1);
            } else {
                bhi = (slot - full) * 8 + xcd;
                qt = nq - 1;
            }
        } else {
            bhi = (slot / nq) * 8 + xcd;
            qt = slot - (slot / nq) * nq;
        }
    } else {
        bhi = blockIdx.x / nq;
        qt = blockIdx.x - bhi * nq;
    }
    const int b = bhi / p.heads, h = bhi - b * p.heads;
    const size_t bh = (size_t)bhi;
    const int win = MODE == 2 ? b % p.win_nw : 0;  // window index inside its image
    const int q0 = SPLIT ? qt * QPB : qt * QPB + wave * QPW;
    const bool active = q0 < p.npad;  // tail waves of the last q-tile only help with DMA and barriers

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q][d = 16*ks + 8*half .. +8]
    opx8 qh[QB][KS], ql[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + qb * 32 + l31;
        const int q_ld = q < p.npad ? q : p.npad - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const size_t o = (bh * p.npad + q_ld) * HD + ks * 16 + half * 8;
            qh[qb][ks] = *(const opx8*)(p.q_hi + o);
            if (X3) ql[qb][ks] = *(const opx8*)(p.q_lo + o);
        }
    }

    // ---- BEiT relative position bias: this head's resized table (ext LUT) and the per-key index terms live in LDS
    //      behind the K/V ring; bias(q,k) = lut[tq[q] - tk[k]] (see beit_relpos_kernel)
    // RUN4, re-strided table (AttnParams::bias_row = R = 2 ww - 1, bias_ww = ww; round 6): the reversed table's index of (query q, key k) is
    // [(wh - 1 - yq) R + (ww - 1 - xq)] + [yk R + xk], a row term plus a column term on both sides. Consecutive query tokens of a window row differ
    // by one table entry, the next window row starts R - ww + 1 entries further: with R = 47 (window 24) the 32 queries of a half wave hit 8 ... 9
    // banks twice in every ds_read2_b32 of the bias gather (844 M bank-conflict cycles per 168 launches on SwinV2-L, profiles/r05_sq_counters_swinl.md).
    // The LDS image uses the row stride S = the smallest S >= R with S = ww (mod 32): token i of the window then sits at C - i (mod 32) - 32 consecutive
    // queries read 32 different banks for each of the four keys of a run. Indices are converted where they are loaded; the global table is unchanged.
    const int brow = (RUN4 && p.bias_row > 0) ? p.bias_row : 0;
    const int bstride = brow ? brow + ((p.bias_ww - brow) & 31) : 0;
    const int belen = brow ? (p.bias_elen / brow) * bstride : p.bias_elen;  // entries of the LDS image
    auto restride = [&](int idx) __attribute__((always_inline)) -> int { return brow ? (idx / brow) * bstride + idx % brow : idx; };
    float* lds_lut = (float*)(smem + RINGS * 2 * STAGE);
    int* lds_tk = (int*)(lds_lut + (BIAS ? belen : 0));
    int* lds_reg = lds_tk + (((p.N + 63) >> 6) << 6);  // MODE 2: shifted-window region id of every key
    int tqv[QB], rqv[QB];
    const bool masked = MODE == 2 && p.region != nullptr;
    if (BIAS) {
        const float* lut = p.bias_lut + (size_t)h * p.bias_elen;
        const float bmul = LOG2 ? kLog2e : 1.0f;
        float badd = 0.0f;
        if constexpr (FIXREF) badd = -(p.swin_ls[h] * 1.015625f + 16.0f * kLog2e);  // -M_h in log2 units (swin_ls already carries log2 e)
        if (brow) {
            for (int i = tid; i < belen; i += 256) {
                const int r = i / bstride, c = i - r * bstride;
                lds_lut[i] = c < brow ? lut[p.bias_elen - 1 - (r * brow + c)] * bmul + badd : 0.0f;
            }
        } else {
            for (int i = tid; i < p.bias_elen; i += 256) lds_lut[i] = lut[RUN4 ? p.bias_elen - 1 - i : i] * bmul + badd;
        }
        const int nk = ((p.N + 63) >> 6) << 6;
        for (int i = tid; i < nk; i += 256) lds_tk[i] = restride(p.tk[i < p.npad ? i : p.npad - 1]) * (RUN4 ? 4 : 1);  // RUN4: byte offsets (one add per table read)
        if (MODE == 2)
            for (int i = tid; i < nk; i += 256) lds_reg[i] = masked ? p.region[(size_t)win * p.region_ld + (i < p.N ? i : p.N - 1)] : 0;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q = q0 + qb * 32 + l31;
            tqv[qb] = p.tq[q < p.npad ? q : p.npad - 1];
            if (RUN4) tqv[qb] = restride(p.bias_elen - 1 - tqv[qb]) * 4;  // byte offset into the reversed table: lut[tq - tk - e] = rev[(elen - 1 - tq) + tk + e]
            rqv[qb] = masked ? p.region[(size_t)win * p.region_ld + (q < p.N ? q : p.N - 1)] : 0;
        }
    }

    // ---- staging: 128-byte LDS rows (K: one key at HD 64, a pair of keys at HD 32; Vt: one d), 1 KiB chunks of 8 rows,
    //      CH per wave and plane; the XOR swizzle is applied on the SOURCE address (LDS-DMA writes lane-linear)
    const int lrow = lane >> 3, slot = lane & 7;
    const int ntiles = (p.N + 63) >> 6;
    char* const ring = smem + (SPLIT ? wave * 2 * STAGE : 0);
    auto issue_tile = [&](int tile, int buf) {
        char* s = ring + buf * STAGE;
        const int kv0 = tile * 64;
#pragma unroll
        for (int i = 0; i < (SPLIT ? 4 * CH : CH); ++i) {
            const int c = SPLIT ? i : wave + 4 * i;                   // 1 KiB chunk = 8 LDS rows
            const int sw_stage = ((c & 1) * 4 + (lrow >> 1)) & 7;     // key(row) = (row >> 1) & 7 of row c*8 + lrow
            const int lslot = slot ^ sw_stage;                        // logical 16-byte slot this lane fetches
            const int koff = lslot * 8;
            int krow = HD == 64 ? kv0 + c * 8 + lrow : kv0 + 2 * (c * 8 + lrow) + (lslot >> 2);
            krow = krow < p.npad ? krow : p.npad - 1;  // rows >= N are masked in the softmax
            const size_t ko = (bh * p.npad + krow) * HD + (HD == 64 ? koff : (lslot & 3) * 8);
            const size_t vo = (bh * HD + c * 8 + lrow) * p.npadv + kv0 + koff;
            glds16(p.k_hi + ko, s + c * 1024);
            if (X3) glds16(p.k_lo + ko, s + TILE + c * 1024);
            glds16(p.vt_hi + vo, s + NPL * TILE + c * 1024);
            if (X3) glds16(p.vt_lo + vo, s + NPL * TILE + TILE + c * 1024);
        }
    };

    const int sw_frag = (l31 >> 1) & 7;
    f32x16 o_acc[QB][DB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = FIXREF ? 0.0f : -1.0e30f;
        l_run[qb] = 0.0f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[qb][db][r] = 0.0f;
    }

    if (SPLIT) {
        if (BIAS) __syncthreads();  // the bias tables were staged by all 256 threads
        if (wave < ntiles) issue_tile(wave, 0);
    } else {
        issue_tile(0, 0);
    }
    for (int t = SPLIT ? wave : 0, it = 0; t < ntiles; t += SPLIT ? 4 : 1, ++it) {
        // LDS-DMA completion is tracked by vmcnt; hipcc does NOT reliably wait for it before the barrier
        // (observed: only lgkmcnt(0) in this loop -> rare stale K/V tiles). Wait explicitly, then publish.
        // Split form: the ring is private to the wave, its own covering vmcnt is all a wave needs to read what it fetched.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!SPLIT) __syncthreads();
        if (SPLIT) {
            if (t + 4 < ntiles) issue_tile(t + 4, (it + 1) & 1);
        } else {
            if (t + 1 < ntiles) issue_tile(t + 1, (t + 1) & 1);
        }
        if (!active) continue;
        const char* sK = ring + ((SPLIT ? it : t) & 1) * STAGE;
        const char* sV = sK + NPL * TILE;

        // ---- S^T[key][query] for 2 blocks of 32 keys x QB blocks of 32 queries
        // The first MFMA of every accumulator takes a constant-zero C operand (an inline constant in the instruction) instead of
        // a zeroed register block: the kernel is VALU-bound and clearing 64 registers per tile costs 32 v_mov_b64
        f32x16 s[QB][2];
        const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (CINIT) {  // the bias (log2 units, minus the fixed reference) is where the accumulators start
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (RUN4) {
                        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                        const int tk0 = lds_tk[t * 64 + blk * 32 + 8 * g + 4 * half];
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            const f32x4u b4 = *(const f32x4u*)((const char*)lds_lut + tqv[qb] + tk0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[qb][blk][4 * g + e] = b4[e];
                        }
                    } else {
                        typedef __attribute__((ext_vector_type(4))) int i32x4c;
                        const i32x4c tk4 = *(const i32x4c*)(lds_tk + t * 64 + blk * 32 + 8 * g + 4 * half);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[qb][blk][4 * g + e] = lds_lut[tqv[qb] - tk4[e]];
                    }
                }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = HD == 64 ? (blk * 32 + l31) * 128 + (((ks * 2 + half) ^ sw_frag) << 4)
                                         : (blk * 16 + (l31 >> 1)) * 128 + ((((l31 & 1) * 4 + ks * 2 + half) ^ ((l31 >> 2) & 7)) << 4);
                const opx8 kh = *(const opx8*)(sK + off);
                opx8 kl;
                if (X3) kl = *(const opx8*)(sK + TILE + off);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (X3) {
                        s[qb][blk] = MDPT_MFMA_32x32x16(kl, qh[qb][ks], (ks == 0 && !CINIT) ? zero16 : s[qb][blk], 0, 0, 0);
                        s[qb][blk] = MDPT_MFMA_32x32x16(kh, ql[qb][ks], s[qb][blk], 0, 0, 0);
                        s[qb][blk] = MDPT_MFMA_32x32x16(kh, qh[qb][ks], s[qb][blk], 0, 0, 0);
                    } else {
                        s[qb][blk] = MDPT_MFMA_32x32x16(kh, qh[qb][ks], (ks == 0 && !CINIT) ? zero16 : s[qb][blk], 0, 0, 0);
                    }
                }
            }
        if (BIAS) {
            typedef __attribute__((ext_vector_type(4))) int i32x4;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (CINIT) {
                        // (already in the accumulators)
                    } else if constexpr (RUN4) {
                        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                        const int tk0 = lds_tk[t * 64 + blk * 32 + 8 * g + 4 * half];
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            const f32x4u b4 = *(const f32x4u*)((const char*)lds_lut + tqv[qb] + tk0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[qb][blk][4 * g + e] += b4[e];
                        }
                    } else {
                        const i32x4 tk4 = *(const i32x4*)(lds_tk + t * 64 + blk * 32 + 8 * g + 4 * half);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[qb][blk][4 * g + e] += lds_lut[tqv[qb] - tk4[e]];
                    }
                    if (MODE == 2 && masked) {
                        const i32x4 rk4 = *(const i32x4*)(lds_reg + t * 64 + blk * 32 + 8 * g + 4 * half);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[qb][blk][4 * g + e] += rqv[qb] != rk4[e] ? -100.0f * (LOG2 ? kLog2e : 1.0f) : 0.0f;
                    }
                }
        }
        // key of s[..][blk][r] = t*64 + blk*32 + (r&3) + 8*(r>>2) + 4*half
        if (t * 64 + 64 > p.N) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        s[qb][blk][r] = key < p.N ? s[qb][blk][r] : -1.0e30f;
                    }
        }
        // ---- online softmax per query block: this lane and lane^32 share the query
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if constexpr (FIXREF) {  // fixed reference point (already subtracted through the bias table): p = 2^s, nothing to track or rescale
                f32x2 psum2 = {0.0f, 0.0f};
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float e0 = __builtin_amdgcn_exp2f(s[qb][blk][r]);
                        const float e1 = __builtin_amdgcn_exp2f(s[qb][blk][r + 1]);
                        s[qb][blk][r] = e0;
                        s[qb][blk][r + 1] = e1;
                        psum2 += f32x2{e0, e1};
                    }
                l_run[qb] += psum2[0] + psum2[1];
                continue;
            }
            float mloc = s[qb][0][0];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[qb][blk][r]);
            {  // lane ^ 32 holds the other half of this query's keys: one VALU swap instead of a trip through the LDS crossbar (-2.3 %)
                const unsigned mu = __builtin_bit_cast(unsigned, mloc);
                auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                const unsigned s0 = sw[0], s1 = sw[1];
                mloc = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
            }
            const float m_new = mloc > m_run[qb] + kDeferMax * (LOG2 ? kLog2e : 1.0f) ? mloc : m_run[qb];
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kExpScale);
            const float mb = m_new * kExpScale;
            m_run[qb] = m_new;
            f32x2 psum2 = {0.0f, 0.0f};
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(LOG2 ? s[qb][blk][r] - mb : s[qb][blk][r] * kLog2e - mb);
                    const float e1 = __builtin_amdgcn_exp2f(LOG2 ? s[qb][blk][r + 1] - mb : s[qb][blk][r + 1] * kLog2e - mb);
                    s[qb][blk][r] = e0;
                    s[qb][blk][r + 1] = e1;
                    psum2 += f32x2{e0, e1};  // one v_pk_add_f32 per pair
                }
            const float psum = psum2[0] + psum2[1];
            l_run[qb] = l_run[qb] * alpha + psum;
            if (__any(alpha != 1.0f)) {  // the reference point moved for some query of this wave (first tiles, outlier keys): rescale O^T
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o_acc[qb][db][r] *= alpha;
            }
        }

        // ---- O^T[d][query] += Vt[d][key] P^T[key][query]; 16-key blocks kb = 2*blk + kb2.
        // The S^T accumulators hold, per lane (query = lane&31, hi = lane>>5), keys {0-3, 8-11} + 4*hi of each 16-key
        // block. Packed to bf16 pairs, ONE v_permlane32_swap per register pair exchanges the halves so that the lane
        // ends up with the standard B fragment (keys 16kb + 8*hi + 0..7): lower lanes keep (0,1),(2,3) and receive
        // (4,5),(6,7) from lane+32; upper lanes receive (8,9),(10,11) and keep (12,13),(14,15). Vt fragments are then
        // plain conflict-free ds_read_b128, shared by all query blocks.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2) {
                opx8 ph[QB], pl[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    unsigned hw[4], lw[4];  // packed opx2: [0],[1] = keys (0,1),(2,3)+4hi ; [2],[3] = keys (8,9),(10,11)+4hi
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const f32x2 pp = {s[qb][blk][kb2 * 8 + 2 * w2], s[qb][blk][kb2 * 8 + 2 * w2 + 1]};
                        const opx2 hh = to_op2_bounded(pp);   // one v_cvt_pk_bf16_f32
                        hw[w2] = __builtin_bit_cast(unsigned, hh);
                        if (X3) {
                            const f32x2 rr = pp - __builtin_convertvector(hh, f32x2);
                            lw[w2] = __builtin_bit_cast(unsigned, to_op2_bounded(rr));
                        }
                    }
                    unsigned pw[4], qw[4];
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        auto r = __builtin_amdgcn_permlane32_swap(hw[w2], hw[w2 + 2], false, false);
                        pw[w2] = r[0];
                        pw[w2 + 2] = r[1];
                        if (X3) {
                            auto rl = __builtin_amdgcn_permlane32_swap(lw[w2], lw[w2 + 2], false, false);
                            qw[w2] = rl[0];
                            qw[w2 + 2] = rl[1];
                        }
                    }
                    ph[qb] = __builtin_bit_cast(opx8, (u32x4){pw[0], pw[1], pw[2], pw[3]});
                    if (X3) pl[qb] = __builtin_bit_cast(opx8, (u32x4){qw[0], qw[1], qw[2], qw[3]});
                }
                const int kb = blk * 2 + kb2;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int off = (db * 32 + l31) * 128 + (((2 * kb + half) ^ sw_frag) << 4);
                    const opx8 vh = *(const opx8*)(sV + off);
                    opx8 vl;
                    if (X3) vl = *(const opx8*)(sV + TILE + off);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        if (X3) {
                            o_acc[qb][db] = MDPT_MFMA_32x32x16(vl, ph[qb], o_acc[qb][db], 0, 0, 0);
                            o_acc[qb][db] = MDPT_MFMA_32x32x16(vh, pl[qb], o_acc[qb][db], 0, 0, 0);
                        }
                        o_acc[qb][db] = MDPT_MFMA_32x32x16(vh, ph[qb], o_acc[qb][db], 0, 0, 0);
                    }
                }
            }
    }

    if (SPLIT) {
        // ---- merge the four partial softmax states: waves 1-3 park (m, l, O) in LDS (the rings are dead), wave 0 folds them in
        constexpr int NREG = QB * (2 + DB * 16);
        __syncthreads();
        float* park = (float*)smem;  // [3][NREG][64]
        if (wave > 0) {
            float* dst = park + (size_t)(wave - 1) * NREG * 64 + lane;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                dst[(qb * (2 + DB * 16) + 0) * 64] = m_run[qb];
                dst[(qb * (2 + DB * 16) + 1) * 64] = l_run[qb];
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(qb * (2 + DB * 16) + 2 + db * 16 + r) * 64] = o_acc[qb][db][r];
            }
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* src = park + (size_t)w * NREG * 64 + lane;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float m_w = src[(qb * (2 + DB * 16) + 0) * 64], l_w = src[(qb * (2 + DB * 16) + 1) * 64];
                const float m_new = fmaxf(m_run[qb], m_w);
                const float a = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * kExpScale), bsc = __builtin_amdgcn_exp2f((m_w - m_new) * kExpScale);
                m_run[qb] = m_new;
                l_run[qb] = l_run[qb] * a + l_w * bsc;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        o_acc[qb][db][r] = o_acc[qb][db][r] * a + src[(qb * (2 + DB * 16) + 2 + db * 16 + r) * 64] * bsc;
            }
        }
    }
    // ---- normalise and store: o_acc[qb][db][r] = O[q][d = 32db + (r&3) + 8(r>>2) + 4half]
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv = 1.0f / l_tot;
        const int q = q0 + qb * 32 + l31;
        if (MODE == 2 ? q < p.N : q < p.npad) {
            size_t orow;
            if (MODE == 2) {  // window reverse + un-roll: row of this window token in its image
                const int img = b / p.win_nw;
                orow = ((size_t)img * p.win_nw * p.N + p.rowmap[(size_t)win * p.N + q]) * (p.out_ld ? p.out_ld : p.F) + h * HD;
            } else {
                orow = ((size_t)b * p.npad + q) * (p.out_ld ? p.out_ld : p.F) + h * HD;
            }
            // 16-byte stores (fewest store instructions for the bytes moved): this lane owns
            // d = 8g + 4*half + 0..3 for g = 0..3; one v_permlane32_swap per packed register pair hands the lower lane the
            // whole d = 8g .. 8g+7 run of the even g and the upper lane that of the odd g
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned hx[2][2], lx[2][2];  // [g parity][packed pair]
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            const f32x2 pp = {o_acc[qb][db][(2 * gp + gi) * 4 + 2 * w2] * inv, o_acc[qb][db][(2 * gp + gi) * 4 + 2 * w2 + 1] * inv};
                            const opx2 hh = to_op2_bounded(pp);
                            hx[gi][w2] = __builtin_bit_cast(unsigned, hh);
                            if (X3) {
                                const f32x2 rr = pp - __builtin_convertvector(hh, f32x2);
                                lx[gi][w2] = __builtin_bit_cast(unsigned, to_op2_bounded(rr));
                            }
                        }
                    unsigned oh[4], ol[4];
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        auto r = __builtin_amdgcn_permlane32_swap(hx[0][w2], hx[1][w2], false, false);
                        oh[w2] = r[0];
                        oh[w2 + 2] = r[1];
                        if (X3) {
                            auto rl = __builtin_amdgcn_permlane32_swap(lx[0][w2], lx[1][w2], false, false);
                            ol[w2] = rl[0];
                            ol[w2 + 2] = rl[1];
                        }
                    }
                    const size_t o = orow + db * 32 + 8 * (2 * gp + half);
                    *(u32x4s*)(p.out_hi + o) = u32x4s{oh[0], oh[1], oh[2], oh[3]};
                    if (X3) *(u32x4s*)(p.out_lo + o) = u32x4s{ol[0], ol[1], ol[2], ol[3]};
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Attention-weight dump (diagnostic, not on the hot path): softmax(q k^T (+ relative-position bias)) of ONE block as an
// explicit [B, H, N, N] fp32 tensor - what the reference exposes through its nn.Softmax module when enable_optimizations is
// False (v2_depthanything/components/transformer_block.py:101,126-131; experiments/attention_visualization.py:325-332).
// One workgroup per (batch*head, query row); operands are the Q (pre-scaled) / K planes the fused kernel consumes. The SwinV2
// window attention has the same hookable nn.Softmax (v31_swinv2/components/windowed_attention.py:60-61,119): HD = 32 form below.
// ------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_weights_kernel(const op_t* __restrict__ q_hi, const op_t* __restrict__ q_lo,
                                                           const op_t* __restrict__ k_hi, const op_t* __restrict__ k_lo,
                                                           const float* __restrict__ bias_lut, int bias_elen, const int* tq, const int* tk,
                                                           const int* __restrict__ region, int region_ld, int win_nw,
                                                           float* __restrict__ out, int heads, int N, int npad) {
    // HD = 32: SwinV2 window attention - the "batch" is (image, window), the operands are the L2-normalised, logit-scaled window
    // tokens of swin_qkv_prep, the bias is the continuous-position-bias LUT and `region` the shifted-window region ids (score -100
    // across regions, v31_swinv2/components/windowed_attention.py:100-119, :394-439)
    extern __shared__ float sc[];  // [N] scores, then 8 floats of reduction scratch
    const int row = blockIdx.x, bh = blockIdx.y, h = bh % heads, tid = threadIdx.x;
    float q[HD];
    const op_t* qp = q_hi + ((size_t)bh * npad + row) * HD;
    const op_t* qlp = q_lo ? q_lo + ((size_t)bh * npad + row) * HD : nullptr;
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = (float)qp[d] + (qlp ? (float)qlp[d] : 0.0f);
    const float* lut = bias_lut ? bias_lut + (size_t)h * bias_elen : nullptr;
    const int tqr = lut ? tq[row] : 0;
    const int* reg = region ? region + (size_t)((bh / heads) % win_nw) * region_ld : nullptr;
    const int rq = reg ? reg[row] : 0;
    float lmax = -3.0e38f;
    for (int key = tid; key < N; key += 256) {
        const op_t* kp = k_hi + ((size_t)bh * npad + key) * HD;
        const op_t* klp = k_lo ? k_lo + ((size_t)bh * npad + key) * HD : nullptr;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[d] * ((float)kp[d] + (klp ? (float)klp[d] : 0.0f));
        if (HD == 32) s *= 0.6931471805599453f;  // the window attention's Q carries log2(e) with its logit scale (attn_kernel LOG2): back to natural units
        if (lut) s += lut[tqr - tk[key]];
        if (reg && reg[key] != rq) s += -100.0f;
        sc[key] = s;
        lmax = fmaxf(lmax, s);
    }
    float* red = sc + N;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.0f;
    for (int key = tid; key < N; key += 256) {
        const float e = __expf(sc[key] - m);
        sc[key] = e;
        lsum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lsum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    float* op = out + ((size_t)bh * N + row) * N;
    for (int key = tid; key < N; key += 256) op[key] = sc[key] * inv;
}

}  // namespace

int MDPT_FN(mdpt_launch_attention)(const AttnParams& p, hipStream_t stream) {
    const int hd = p.head_dim ? p.head_dim : 64;
    if ((hd != 64 && hd != 32) || p.F != p.heads * hd || (p.npadv & 63) || p.npadv < ((p.N + 63) & ~63) || p.npad < p.N)
        return (int)hipErrorInvalidValue;
    const bool swin = p.rowmap != nullptr;
    if (swin && (hd != 32 || !p.bias_lut || !p.swin_ls || p.win_nw <= 0 || p.B % p.win_nw)) return (int)hipErrorInvalidValue;
    if (!swin && hd != 64) return (int)hipErrorInvalidValue;
    // 64 queries per wave (256 per workgroup: K / V fragments read once per 64 queries) when that still gives >= 2 workgroups per CU and,
    // at head dim 64, when the query count fills its last 256-query workgroup reasonably: N = 1297 pads to 1536 (+18 %) and N = 577 to 768
    // (+33 %), where 128-query workgroups (11 resp. 5 per head, 116 VGPRs: four waves per SIMD) measure +0.7 % on the ViT-L step and
    // +1.0 % on BEiT-L; N = 5477 (1036x1036) pads by 3 % and stays wide (+1.4 %), and so does the d = 32 window attention (+4.7 %:
    // profiles/r03_attention_queries_per_wave_ab.txt). Both forms give a query the same bits (per-lane softmax state, same key order).
    const long blocks256 = (long)((p.npad + 255) / 256) * p.heads * p.B;
#ifdef MDPT_DEBUG_SWITCHES  // A/B builds only: a stray environment variable must not change which kernel a deployment runs
    static const int wide_env = getenv("MDPT_ATTN_WIDE") ? atoi(getenv("MDPT_ATTN_WIDE")) : -1;
#else
    constexpr int wide_env = -1;
#endif
    const bool fills = hd == 32 || (long)((p.N + 255) / 256) * 256 * 100 <= (long)p.N * 108;
    const bool wide = !p.x3 && (wide_env >= 0 ? wide_env != 0 : (blocks256 >= 512 && fills));
    const bool bias = p.bias_lut != nullptr;
    const int ntk = ((p.N + 63) / 64) * 64;
    // (window attention with the re-strided table image: rows of stride S >= 2 ww - 1, S = ww mod 32 - see the kernel)
    const bool restr = swin && p.bias_run4 && p.bias_row > 0 && p.bias_ww > 0 && p.bias_elen % p.bias_row == 0;
    const size_t lut_entries = restr ? (size_t)(p.bias_elen / p.bias_row) * (p.bias_row + ((p.bias_ww - p.bias_row) & 31)) : (size_t)p.bias_elen;
    const size_t extra = bias ? lut_entries * 4 + (size_t)ntk * 4 * (swin ? 2 : 1) : 0;
    const size_t ring = (size_t)2 * 2 * (p.x3 ? 2 : 1) * 64 * hd * 2;
    if (ring + extra > 160 * 1024) return (int)hipErrorInvalidValue;  // relative-position table does not fit in LDS
    // latency form (opt-in, mdpt_set_latency_mode: the merge changes the summation order): a launch of at most ~one workgroup per CU
    // even at 32 queries per workgroup (batch 1 of the small models) splits the key loop over the four waves instead (bf16 only: four
    // private rings of the x3 operand planes do not fit)
    const long blocks32 = (long)((p.npad + 31) / 32) * p.heads * p.B;
    if (p.allow_split_kv && !swin && !p.x3 && blocks32 <= 320 && 4 * ring + extra <= 160 * 1024) {
        AttnParams ps = p;
        ps.tail_last = 0;
        const unsigned lds_s = (unsigned)(4 * ring + extra);
        MdptProfScope prof_s(bias ? "attn_kernel<false, 1, 1, 64, split>" : "attn_kernel<false, 1, 0, 64, split>",
                             4.0 * p.B * p.heads * (double)p.N * p.N * hd, stream);
        static bool attr0 = false, attr1 = false;
        if (bias) {
            auto kern = attn_kernel<false, 1, 1, 64, true>;
            if (!attr1) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
                attr1 = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks32), dim3(256), lds_s, stream, ps);
        } else {
            auto kern = attn_kernel<false, 1, 0, 64, true>;
            if (!attr0) {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
                attr0 = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks32), dim3(256), lds_s, stream, ps);
        }
        return (int)hipGetLastError();
    }
    static const char* const kNames[2][2][3] = {
        {{"attn_kernel<false, 1, 0, 64>", "attn_kernel<false, 1, 1, 64>", "attn_kernel<false, 1, 2, 32>"},
         {"attn_kernel<false, 2, 0, 64>", "attn_kernel<false, 2, 1, 64>", "attn_kernel<false, 2, 2, 32>"}},
        {{"attn_kernel<true, 1, 0, 64>", "attn_kernel<true, 1, 1, 64>", "attn_kernel<true, 1, 2, 32>"},
         {"attn_kernel<true, 1, 0, 64>", "attn_kernel<true, 1, 1, 64>", "attn_kernel<true, 1, 2, 32>"}}};
    const int mode = swin ? 2 : (bias ? 1 : 0);
    MdptProfScope prof(kNames[p.x3 ? 1 : 0][wide ? 1 : 0][mode], 4.0 * p.B * p.heads * (double)p.N * p.N * hd, stream);
    const unsigned lds = (unsigned)(ring + extra);
    AttnParams pq = p;
    pq.tail_last = 1;  // -4 % per launch when the kernel runs alone (N = 1297); neutral under the two-stream batch split
    const dim3 grid128(((p.npad + 127) / 128) * p.heads * p.B), grid256((unsigned)blocks256), block(256);
#define ATTN_LAUNCH(X3_, QB_, MODE_, HD_, GRID_) ATTN_LAUNCH_R(X3_, QB_, MODE_, HD_, GRID_, false)
#define ATTN_LAUNCH_R(X3_, QB_, MODE_, HD_, GRID_, RUN4_)                                                               \
    do {                                                                                                                \
        auto kern = attn_kernel<X3_, QB_, MODE_, HD_, false, RUN4_>;                                                    \
        static bool attr_done = false;                                                                                  \
        if (!attr_done) {                                                                                               \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return (int)e;                                                                         \
            attr_done = true;                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, GRID_, block, lds, stream, pq);                                                        \
    } while (0)
    // window attention with 4-key bias runs (AttnParams::bias_run4: the caller guarantees window width and token count are multiples of 4)
    if (mode == 2 && p.bias_run4) {
        if (p.x3) ATTN_LAUNCH_R(true, 1, 2, 32, grid128, true);
        else if (wide) ATTN_LAUNCH_R(false, 2, 2, 32, grid256, true);
        else ATTN_LAUNCH_R(false, 1, 2, 32, grid128, true);
    } else if (p.x3) {
        if (mode == 2) ATTN_LAUNCH(true, 1, 2, 32, grid128);
        else if (mode == 1) ATTN_LAUNCH(true, 1, 1, 64, grid128);
        else ATTN_LAUNCH(true, 1, 0, 64, grid128);
    } else if (wide) {
        if (mode == 2) ATTN_LAUNCH(false, 2, 2, 32, grid256);
        else if (mode == 1) ATTN_LAUNCH(false, 2, 1, 64, grid256);
        else ATTN_LAUNCH(false, 2, 0, 64, grid256);
    } else {
        if (mode == 2) ATTN_LAUNCH(false, 1, 2, 32, grid128);
        else if (mode == 1) ATTN_LAUNCH(false, 1, 1, 64, grid128);
        else ATTN_LAUNCH(false, 1, 0, 64, grid128);
    }
#undef ATTN_LAUNCH
#undef ATTN_LAUNCH_R
    return (int)hipGetLastError();
}

int MDPT_FN(mdpt_launch_attn_weights)(const AttnParams& p, float* out_bhnn, hipStream_t stream) {
    const bool swin = p.rowmap != nullptr;
    const int hd = p.head_dim ? p.head_dim : 64;
    if ((swin && (hd != 32 || p.win_nw <= 0)) || (!swin && hd != 64)) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(p.N + 8) * 4;
    if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
    if (swin)
        hipLaunchKernelGGL(attn_weights_kernel<32>, dim3(p.N, p.B * p.heads), dim3(256), lds, stream, p.q_hi, p.q_lo, p.k_hi, p.k_lo, p.bias_lut,
                           p.bias_elen, p.tq, p.tk, p.region, p.region_ld, p.win_nw, out_bhnn, p.heads, p.N, p.npad);
    else
        hipLaunchKernelGGL(attn_weights_kernel<64>, dim3(p.N, p.B * p.heads), dim3(256), lds, stream, p.q_hi, p.q_lo, p.k_hi, p.k_lo, p.bias_lut,
                           p.bias_elen, p.tq, p.tk, (const int*)nullptr, 0, 1, out_bhnn, p.heads, p.N, p.npad);
    return (int)hipGetLastError();
}
