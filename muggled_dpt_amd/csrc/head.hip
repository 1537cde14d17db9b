// Fused tail of the monocular depth head for gfx950 (MI355X):
//
//     x (P/8) bilinear upsample (align_corners=True)  ->  3x3 conv (C/2 -> 32) + ReLU  ->  1x1 conv (32 -> 1) + ReLU | sigmoid
//
// Reference: MonocularDepthHead.forward, v2_depthanything/head_model.py:74-85 (SpatialUpsampleLayer after the first 3x3 conv, then
// proj_1ch) and :100-106; SpatialUpsampleLayer = F.interpolate(scale_factor, "bilinear", align_corners=True), components/misc_helpers.py:39-42.
//
// Unfused, the upsampled C/2-channel map at full image resolution is the largest tensor of the whole forward pass (ViT-L, batch 32:
// 2.08 GB written by the upsample kernel and read back 9x from L2 by the im2col-free conv: profiles/r01_hbm_traffic.md). Here it only
// ever exists as a 16x16-pixel tile (+1 halo) in LDS:
//
//   persistent workgroup of 8 waves = two groups of 4. The 3x3 conv weights live in REGISTERS for the workgroup's whole life: group g
//   keeps the MFMA fragments of the k-steps ks = g, g+2, ... of all nine taps (9 * CIN/32 fragments = 144 VGPRs at CIN = 128), i.e.
//   the contraction is split in two halves whose partial sums meet in the epilogue. Per output tile:
//     1. source patch: the <= 12x12 low-resolution pixels the tile's 18x18 halo interpolates from go global -> LDS by LDS-DMA, issued
//        one tile AHEAD (right after the previous tile's halo was built) so the transfer hides under the previous MFMA phase;
//     2. halo tile: the 18x18 upsampled pixels the tile's 3x3 taps touch are interpolated out of the LDS patch (four 16-byte LDS reads
//        + 3 lerps per 8 channels in packed fp32 math, offsets and weights from per-tile row / column tables) and written to LDS as
//        bf16; pixels outside the image are the conv's zero padding. padded pixel / row pitches make every MFMA fragment read conflict-free at immediate offsets;
//     3. implicit GEMM straight out of LDS: wave q of a group owns tile rows 4q .. 4q+3 (two blocks of 32 pixels); for each tap the
//        pixel fragment is the halo read shifted by (ky, kx) - no im2col, no re-fetch from L2 - and feeds one 32x32x16 bf16 MFMA per
//        block against the register-resident weight fragment (weights = A operand: the accumulator holds C[n][pixel], a lane owns one
//        pixel and 16 of the 32 conv outputs). One LDS read per MFMA: 128 B/clk at the MFMA peak, half the LDS bandwidth;
//     4. the two groups exchange one block each through LDS (fixed order: group 0's partial + group 1's), then per pixel in registers:
//        + bias, ReLU, dot with the 1x1 conv weights (16 per lane + one exchange with lane^32), + bias, ReLU | sigmoid, store in the
//        caller's dtype.
// LDS at CIN = 128: halo 90 KiB + patch 36 KiB + exchange 32 KiB = 158.6 KiB (one workgroup per CU).

#include "mdpt_kernels.h"
#include "mdpt_prof.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int TS = 16, HS = TS + 2;  // output tile side, halo side
constexpr int PS = 12;               // max side of the source patch (launcher checks 17 * scale + 2 <= PS)

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    // 64 lanes x 16 B -> lds_wave_base + lane*16 (LDS-DMA: the destination is wave-uniform base + lane*16, no register round trip)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct TileGeom {
    int b, oy0, ox0, py0, px0, ph, pw;
};

template <int CIN>
__global__ __launch_bounds__(512, 1) void head_tail_kernel(const HeadTailParams p) {
    constexpr int NCH = CIN / 8;               // 16-byte channel chunks per pixel
    constexpr int PIXB = CIN * 2;              // bytes per pixel (bf16)
    constexpr int KSTEPS = CIN / 16;           // MFMA k-steps per tap
    constexpr int KG = KSTEPS / 2;             // ... of which each wave group takes every second one
    constexpr int PPI = 64 / NCH;              // pixels per 1 KiB DMA instruction
    // halo image: pixel pitch PIXB + 16 (an odd number of 16-byte slots: the 16 pixels of a tile row land on 16 different slots of the
    // 256-byte bank row for every channel chunk) and a row pitch that is a multiple of 256 bytes (tile rows r and r+1, which share the
    // 16-lane service groups of a ds_read_b128, see the same slot pattern): conflict-free fragment reads at plain immediate offsets
    constexpr int PIXP = PIXB + 16, ROWB = ((HS * PIXP + 255) / 256) * 256;
    static_assert((PIXP / 16) % 2 == 1 && ROWB % 256 == 0, "halo pitches");
    constexpr int HALO_BYTES = HS * ROWB, PATCH_BYTES = PS * PS * PIXB, XCH_BYTES = 8 * 16 * 64 * 4;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;
    char* const sP = smem + HALO_BYTES;
    float* const sX = (float*)(smem + HALO_BYTES + PATCH_BYTES);  // [8 waves][16 regs][64 lanes] fp32: block exchange between the groups
    // interpolation tables of the tile, y rows then x columns: byte offsets of the two source rows / columns (-1: outside the image), weight
    int* const sT0 = (int*)(smem + HALO_BYTES + PATCH_BYTES + XCH_BYTES);
    int* const sT1 = sT0 + 2 * HS;
    float* const sTL = (float*)(sT1 + 2 * HS);
    float* const sC = (float*)(smem + HALO_BYTES + PATCH_BYTES + XCH_BYTES + 2 * HS * 16);  // [32] conv bias, [32] 1x1 conv weights

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, q = wave & 3;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- this wave's weight fragments -> registers, once: W[n = l31][k = tap*CIN + (2*kg + grp)*16 + half*8 .. +8]
    //      (w_kc image: chunk (k/8), row n, 8 elements: 16 consecutive bytes per lane, 512 per fragment and half)
    opx8 wreg[9][KG];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int chunk = tap * NCH + (2 * kg + grp) * 2 + half;
            wreg[tap][kg] = *(const opx8*)((const char*)p.w_kc + ((size_t)chunk * 32 + l31) * 16);
        }

    const float sy = p.Ho > 1 ? (float)(p.Hi - 1) / (float)(p.Ho - 1) : 0.0f;
    const float sx = p.Wo > 1 ? (float)(p.Wi - 1) / (float)(p.Wo - 1) : 0.0f;
    const int tiles_x = (p.Wo + TS - 1) / TS, tiles_y = (p.Ho + TS - 1) / TS;
    const int ntiles = p.B * tiles_y * tiles_x;
    const float head_b = p.head_b[0];
    if (tid < 64) sC[tid] = tid < 32 ? p.bias[tid] : p.head_w[tid - 32];  // published by the first tile's first barrier

    // tile -> image, origin and the source patch its halo interpolates from (halo rows / columns outside the image are zero padding)
    auto geom = [&](int tile) {
        TileGeom g;
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        g.b = tile / (tiles_x * tiles_y);
        g.oy0 = ty * TS; g.ox0 = tx * TS;
        const int oy_first = g.oy0 == 0 ? 0 : g.oy0 - 1, ox_first = g.ox0 == 0 ? 0 : g.ox0 - 1;
        const int oy_last = min(g.oy0 + TS, p.Ho - 1), ox_last = min(g.ox0 + TS, p.Wo - 1);
        g.py0 = (int)(sy * (float)oy_first); g.px0 = (int)(sx * (float)ox_first);
        g.ph = min((int)(sy * (float)oy_last) + 1, p.Hi - 1) - g.py0 + 1;
        g.pw = min((int)(sx * (float)ox_last) + 1, p.Wi - 1) - g.px0 + 1;  // <= PS (launcher's scale check)
        return g;
    };
    // source patch global -> LDS by LDS-DMA, dense [ph*pw][CIN] image (pixel pitch pw): 1 KiB = PPI pixels per wave instruction
    auto issue_patch = [&](const TileGeom& g) {
        const op_t* src = p.src + (size_t)g.b * p.Hi * p.Wi * CIN;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // opaque: keeps hipcc from hoisting the lane's address part to kernel entry (it gets spilled there)
        const int npix = g.ph * g.pw, c = lane_o % NCH;
        for (int j = wave; j * PPI < npix; j += 8) {
            int pp = j * PPI + lane_o / NCH;
            pp = pp < npix ? pp : npix - 1;  // tail lanes re-fetch the last pixel into the slack behind the image
            const int yy = pp / g.pw, xx = pp - yy * g.pw;
            glds16(src + ((size_t)(g.py0 + yy) * p.Wi + (g.px0 + xx)) * CIN + c * 8, sP + (size_t)j * 1024);
        }
    };

    int tile = blockIdx.x;
    TileGeom g = geom(tile < ntiles ? tile : 0);
    if (tile < ntiles) issue_patch(g);

    int tile_no = 0;
    auto stamp = [&](int slot) {
        if (p.dbg_times && tile_no == 1 && tid == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            p.dbg_times[(size_t)blockIdx.x * 8 + slot] = t;
        }
    };
    for (; tile < ntiles; tile += gridDim.x, ++tile_no) {
        stamp(0);
        // ---- interpolation tables of this tile (align_corners=True): per halo row {byte offset of source row y0, of y1, ly, inside};
        //      per halo column the same in x. Written before, read after the barrier that also publishes the patch.
        if (tid < 2 * HS) {
            const bool isx = tid >= HS;
            const int hidx = isx ? tid - HS : tid;
            const int o = (isx ? g.ox0 : g.oy0) + hidx - 1, lim_o = isx ? p.Wo : p.Ho, lim_i = isx ? p.Wi : p.Hi;
            const float f = (isx ? sx : sy) * (float)o;
            const int i0 = (int)f, i1 = i0 + (i0 < lim_i - 1);
            const int pitch = isx ? PIXB : g.pw * PIXB, org = isx ? g.px0 : g.py0;
            const bool ok = (unsigned)o < (unsigned)lim_o;
            sT0[tid] = ok ? (i0 - org) * pitch : -1;
            sT1[tid] = ok ? (i1 - org) * pitch : -1;
            sTL[tid] = f - (float)i0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the patch has landed
        __syncthreads();
        stamp(1);

        // ---- halo tile out of the patch. A thread owns a column strip: halo column hx, channel chunk c, nine consecutive halo rows. It
        //      keeps the two source rows y0, y1 of the current halo row ALREADY interpolated along x (t = v0 + lx (v1 - v0), fp32) in
        //      registers: going down the strip the source row advances every ~1.75 halo rows, so most steps only do the vertical
        //      lerp (the branches depend on the row only: wave-uniform). bf16 -> fp32 is a shift for the even element of a packed pair;
        //      the odd one is used IN PLACE (its low 16 bits are the neighbour's bits: a relative perturbation below 2^-16, two
        //      hundred times finer than the bf16 rounding of the result)
#if MDPT_OP_IS_F16
        auto to_f2 = [](unsigned d) { return op2_to_f32(d); };  // fp16 operands: two v_cvt_f32_f16
#else
        auto to_f2 = [](unsigned d) { return f32x2{__builtin_bit_cast(float, d << 16), __builtin_bit_cast(float, d)}; };
#endif
        if (tid < 2 * TS * NCH) {
            // strips: halo columns 0..15, chunk c, rows 9 part .. 9 part + 8
            const int part = tid / (TS * NCH), rem = tid - part * (TS * NCH);
            const int hx = rem / NCH, c = rem - hx * NCH;
            const int ox_0 = sT0[HS + hx], ox_1 = sT1[HS + hx];
            const float lx = sTL[HS + hx];
            const char* const pbase = sP + c * 16;
            char* const hcol = sH + hx * PIXP + c * 16;
            auto hlerp = [&](int row_off, f32x2 (&t)[4]) {
                const u32x4 v0 = *(const u32x4*)(pbase + row_off + ox_0);
                const u32x4 v1 = *(const u32x4*)(pbase + row_off + ox_1);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned d0 = v0[w], d1 = v1[w];  // (scalar copies: __builtin_bit_cast on a vector ELEMENT expression reads element 0)
                    const f32x2 a0 = to_f2(d0), a1 = to_f2(d1);
                    t[w] = a0 + lx * (a1 - a0);
                }
            };
            f32x2 t0[4], t1[4];
            int c0 = -2, c1 = -2;  // patch row offsets currently held in t0 / t1
            int oy_0 = sT0[9 * part], oy_1 = sT1[9 * part];
            for (int hy = 9 * part; hy < 9 * part + 9; ++hy) {
                // next row's table entries are fetched before this row's work (LDS latency under the arithmetic)
                const int hn = hy + 1 < HS ? hy + 1 : hy;
                const int ny_0 = sT0[hn], ny_1 = sT1[hn];
                const float ly = sTL[hy];
                u32x4 outw = {0u, 0u, 0u, 0u};
                if ((oy_0 | ox_0) >= 0) {
                    if (oy_0 != c0) {
                        if (oy_0 == c1) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) t0[w] = t1[w];
                        } else {
                            hlerp(oy_0, t0);
                        }
                        c0 = oy_0;
                    }
                    if (oy_1 != c1) {
                        if (oy_1 == c0) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) t1[w] = t0[w];
                        } else {
                            hlerp(oy_1, t1);
                        }
                        c1 = oy_1;
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f32x2 o2 = t0[w] + ly * (t1[w] - t0[w]);
                        outw[w] = __builtin_bit_cast(unsigned, to_op2(o2));
                    }
                }
                *(u32x4*)(hcol + hy * ROWB) = outw;
                oy_0 = ny_0; oy_1 = ny_1;
            }
        }
        // the two right-most halo columns (16, 17), one (row, chunk) item at a time over all threads: full bilinear from four reads
        for (int item = tid; item < 2 * HS * NCH; item += 512) {
            const int c = item % NCH, rest = item / NCH;
            const int hy = rest % HS, hx = TS + rest / HS;
            const int oy_0 = sT0[hy], oy_1 = sT1[hy], ox_0 = sT0[HS + hx], ox_1 = sT1[HS + hx];
            u32x4 outw = {0u, 0u, 0u, 0u};
            if ((oy_0 | ox_0) >= 0) {
                const float ly = sTL[hy], lx = sTL[HS + hx];
                const char* base = sP + c * 16;
                const u32x4 v00 = *(const u32x4*)(base + oy_0 + ox_0);
                const u32x4 v01 = *(const u32x4*)(base + oy_0 + ox_1);
                const u32x4 v10 = *(const u32x4*)(base + oy_1 + ox_0);
                const u32x4 v11 = *(const u32x4*)(base + oy_1 + ox_1);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned d00 = v00[w], d01 = v01[w], d10 = v10[w], d11 = v11[w];
                    const f32x2 a00 = to_f2(d00), a01 = to_f2(d01), a10 = to_f2(d10), a11 = to_f2(d11);
                    const f32x2 u0 = a00 + lx * (a01 - a00);
                    const f32x2 u1 = a10 + lx * (a11 - a10);
                    const f32x2 o2 = u0 + ly * (u1 - u0);
                    outw[w] = __builtin_bit_cast(unsigned, to_op2(o2));
                }
            }
            *(u32x4*)(sH + hy * ROWB + hx * PIXP + c * 16) = outw;
        }
        __syncthreads();
        stamp(2);

        // ---- the next tile's patch streams in under this tile's MFMA phase (the patch region is free from here on)
        const int next = tile + gridDim.x;
        const TileGeom gn = geom(next < ntiles ? next : tile);
        if (next < ntiles) issue_patch(gn);

        // ---- implicit GEMM out of LDS: C[n][pixel] += W[n][tap, ci] * halo[pixel + tap][ci] over this group's k-steps
        //      block blk of this wave = tile rows 4q + 2 blk, 4q + 2 blk + 1; lane pixel (row + (l31 >> 4), l31 & 15)
        f32x16 acc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.0f;
        const int px = l31 & 15, prow = 4 * q + (l31 >> 4);
        // fragment reads run a PAIR of k-steps ahead of the MFMAs (register double buffer, issue order pinned: hipcc's own waits are
        // always lgkmcnt(0), so each one is placed four MFMAs behind the newest reads)
        const char* const hbase = sH + prow * ROWB + px * PIXP + ((2 * grp + half) << 4);
        auto frag = [&](int step, int blk) -> opx8 {
            const int tap = step / KG, kg = step - tap * KG;
            const int ky = tap / 3, kx = tap - 3 * ky;
            return *(const opx8*)(hbase + (ky + 2 * blk) * ROWB + kx * PIXP + kg * 64);
        };
        constexpr int NSTEP = 9 * KG;
        static_assert(NSTEP % 2 == 0, "k-steps are consumed in pairs");
        opx8 xf[2][2][2];  // [buffer][step of the pair][block]
#pragma unroll
        for (int u = 0; u < 2; ++u) { xf[0][u][0] = frag(u, 0); xf[0][u][1] = frag(u, 1); }
#pragma unroll
        for (int pair = 0; pair < NSTEP / 2; ++pair) {
            const int cur = pair & 1;
            // first MFMA of the pair (its lgkmcnt(0) retires the reads issued during the previous pair), THEN the next pair's reads
            acc[0] = MDPT_MFMA_32x32x16(wreg[(2 * pair) / KG][(2 * pair) % KG], xf[cur][0][0], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (pair + 1 < NSTEP / 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { xf[cur ^ 1][u][0] = frag(2 * pair + 2 + u, 0); xf[cur ^ 1][u][1] = frag(2 * pair + 2 + u, 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = MDPT_MFMA_32x32x16(wreg[(2 * pair) / KG][(2 * pair) % KG], xf[cur][0][1], acc[1], 0, 0, 0);
            acc[0] = MDPT_MFMA_32x32x16(wreg[(2 * pair + 1) / KG][(2 * pair + 1) % KG], xf[cur][1][0], acc[0], 0, 0, 0);
            acc[1] = MDPT_MFMA_32x32x16(wreg[(2 * pair + 1) / KG][(2 * pair + 1) % KG], xf[cur][1][1], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        stamp(3);
        // ---- the groups swap one block each (group 0 finishes block 0, group 1 block 1), summing group 0's partial + group 1's.
        //      Exchange image [wave][r / 4][lane][4 floats]: 16-byte accesses at a 16-byte lane stride (conflict-free)
        {
            f32x4* mine = (f32x4*)sX + (size_t)wave * 4 * 64 + lane;
            const f32x16& give = acc[grp ^ 1];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) mine[r4 * 64] = f32x4{give[4 * r4], give[4 * r4 + 1], give[4 * r4 + 2], give[4 * r4 + 3]};
        }
        __syncthreads();
        stamp(4);
        //      relu(conv + bias) . w + b -> relu | sigmoid   (head_model.py:80-85); conv outputs of register group r4: n = 8 r4 + 4 half + 0..3
        float s = 0.0f;
        {
            const f32x4* theirs = (const f32x4*)sX + (size_t)(wave ^ 4) * 4 * 64 + lane;
            const f32x4* cb = (const f32x4*)sC;  // [8] bias, [8] 1x1 conv weights, as groups of 4 consecutive outputs
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 th = theirs[r4 * 64], b4 = cb[2 * r4 + half], w4 = cb[8 + 2 * r4 + half];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mine_v = grp == 0 ? acc[0][4 * r4 + e] : acc[1][4 * r4 + e];
                    const float tot = grp == 0 ? mine_v + th[e] : th[e] + mine_v;
                    s += fmaxf(tot + b4[e], 0.0f) * w4[e];
                }
            }
        }
        s += __shfl_xor(s, 32);
        s += head_b;
        const float dv = p.sigmoid ? 1.0f / (1.0f + __expf(-s)) : fmaxf(s, 0.0f);
        const int oy = g.oy0 + prow + 2 * grp, ox = g.ox0 + px;
        if (half == 0 && oy < p.Ho && ox < p.Wo) {
            const size_t o = ((size_t)g.b * p.Ho + oy) * p.Wo + ox;
            if (p.out_dtype == MDPT_DT_BF16) ((__bf16*)p.out)[o] = (__bf16)dv;
            else if (p.out_dtype == MDPT_DT_F16) ((_Float16*)p.out)[o] = (_Float16)dv;
            else ((float*)p.out)[o] = dv;
        }
        stamp(5);
        g = gn;
        // barriers: the table / patch of the next tile are written after this tile's staging ended (barrier 2) and published by its
        // barrier 1, which also orders this tile's exchange reads before the next exchange writes
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Two-plane form (MDPT_CLASS_HEAD_TAIL at TWO passes, the activation-split form of mdpt_set_class_passes): the head's first conv hands over
// hi + lo 16-bit planes of its output, the upsampled map is interpolated in fp32 from hi + lo and split again into hi / lo HALO planes,
// and every conv weight fragment (one rounded plane, still register-resident) meets both: C += W * halo_lo, C += W * halo_hi. What this
// buys over the single-plane kernel is the rounding of the conv's ACTIVATIONS - the part of this layer's operand rounding that reaches the
// depth map (profiles/r05_precision_budget.md: weights rounded once cost 3e-5 rms, activations rounded once 5.9e-5).
// Two halo planes of all CIN channels do not fit the LDS, so a tile is processed in STAGES of 64 channels: patch (hi + lo) -> halo (hi + lo)
// -> MFMAs of the stage's k-steps, accumulators carried across the stages; the block exchange between the wave groups reuses the halo space.
//   LDS: halo 2 x 49.5 KiB + patch 2 x 18 KiB + tables = 135.8 KiB.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(512, 1) void head_tail2_kernel(const HeadTailParams p) {
    constexpr int SC = 64, NST = CIN / SC;     // channels per stage, stages per tile
    constexpr int NCH = SC / 8;                // 16-byte channel chunks per pixel of a stage
    constexpr int PIXB = SC * 2;               // bytes per pixel of a stage
    constexpr int KSTEPS = CIN / 16, KG = KSTEPS / 2;  // weight fragments per tap: all k-steps / this wave group's
    constexpr int KGS = SC / 32;               // k-steps per tap, wave group and stage
    constexpr int PIXP = PIXB + 16, ROWB = ((HS * PIXP + 255) / 256) * 256;  // (same pitch rules as the single-plane kernel)
    static_assert((PIXP / 16) % 2 == 1 && ROWB % 256 == 0, "halo pitches");
    constexpr int HALO_BYTES = HS * ROWB, PATCH_PLANE = PS * PS * PIXB;
    static_assert(8 * 16 * 64 * 4 <= 2 * HALO_BYTES, "the block exchange lives in the halo space");
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;                                  // [plane: hi, lo][HS rows][ROWB]
    char* const sP = smem + 2 * HALO_BYTES;                 // [plane: hi, lo][patch pixel][PIXB]
    float* const sX = (float*)smem;                         // exchange image (after the tile's last MFMA): [8 waves][16 regs][64 lanes] fp32
    int* const sT0 = (int*)(smem + 2 * HALO_BYTES + 2 * PATCH_PLANE);
    int* const sT1 = sT0 + 2 * HS;
    float* const sTL = (float*)(sT1 + 2 * HS);
    float* const sC = (float*)(smem + 2 * HALO_BYTES + 2 * PATCH_PLANE + 2 * HS * 16);  // [32] conv bias, [32] 1x1 conv weights

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, q = wave & 3;
    const int l31 = lane & 31, half = lane >> 5;

    // this wave's weight fragments -> registers, once (the w_kc image and the k-step split over the wave groups of the single-plane kernel)
    opx8 wreg[9][KG];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int chunk = tap * (CIN / 8) + (2 * kg + grp) * 2 + half;
            wreg[tap][kg] = *(const opx8*)((const char*)p.w_kc + ((size_t)chunk * 32 + l31) * 16);
        }

    const float sy = p.Ho > 1 ? (float)(p.Hi - 1) / (float)(p.Ho - 1) : 0.0f;
    const float sx = p.Wo > 1 ? (float)(p.Wi - 1) / (float)(p.Wo - 1) : 0.0f;
    const int tiles_x = (p.Wo + TS - 1) / TS, tiles_y = (p.Ho + TS - 1) / TS;
    const int ntiles = p.B * tiles_y * tiles_x;
    const float head_b = p.head_b[0];
    if (tid < 64) sC[tid] = tid < 32 ? p.bias[tid] : p.head_w[tid - 32];  // published by the first tile's first barrier

    auto geom = [&](int tile) {
        TileGeom g;
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        g.b = tile / (tiles_x * tiles_y);
        g.oy0 = ty * TS; g.ox0 = tx * TS;
        const int oy_first = g.oy0 == 0 ? 0 : g.oy0 - 1, ox_first = g.ox0 == 0 ? 0 : g.ox0 - 1;
        const int oy_last = min(g.oy0 + TS, p.Ho - 1), ox_last = min(g.ox0 + TS, p.Wo - 1);
        g.py0 = (int)(sy * (float)oy_first); g.px0 = (int)(sx * (float)ox_first);
        g.ph = min((int)(sy * (float)oy_last) + 1, p.Hi - 1) - g.py0 + 1;
        g.pw = min((int)(sx * (float)ox_last) + 1, p.Wi - 1) - g.px0 + 1;  // <= PS (launcher's scale check)
        return g;
    };
    // source patch of stage st (64 channels, hi and lo plane) global -> LDS by LDS-DMA: 1 KiB = 8 pixels per wave instruction
    auto issue_patch = [&](const TileGeom& g, int st) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // (keeps the lane's address part next to its use, see the single-plane kernel)
        const int npix = g.ph * g.pw, c = lane_o % NCH, ninstr = (npix + 7) >> 3;
        const size_t img = (size_t)g.b * p.Hi * p.Wi * CIN + (size_t)st * SC + c * 8;
        for (int j = wave; j < 2 * ninstr; j += 8) {
            const int pl = j >= ninstr ? 1 : 0, jj = j - pl * ninstr;  // wave-uniform
            int pp = jj * 8 + lane_o / NCH;
            pp = pp < npix ? pp : npix - 1;
            const int yy = pp / g.pw, xx = pp - yy * g.pw;
            glds16((pl ? p.src_lo : p.src) + img + ((size_t)(g.py0 + yy) * p.Wi + (g.px0 + xx)) * CIN, sP + pl * PATCH_PLANE + (size_t)jj * 1024);
        }
    };

    int tile = blockIdx.x;
    TileGeom g = geom(tile < ntiles ? tile : 0);
    if (tile < ntiles) issue_patch(g, 0);

    int tile_no = 0;
    auto stamp = [&](int slot) {  // test hook (MDPT_HEAD_DBG builds): phase stamps of every workgroup's second tile
        if (p.dbg_times && tile_no == 1 && tid == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            p.dbg_times[(size_t)blockIdx.x * 8 + slot] = t;
        }
    };
    for (; tile < ntiles; tile += gridDim.x, ++tile_no) {
        stamp(0);
        // ---- interpolation tables of this tile (written before, read after the first stage's first barrier)
        //      (tid_o: an opaque copy of the thread index - per-thread constants of the table / halo phases hoisted out of the tile loop
        //      would live in VGPRs next to the 144 weight registers and the accumulators for the whole kernel, and spill)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        if (tid_o < 2 * HS) {
            const bool isx = tid_o >= HS;
            const int hidx = isx ? tid_o - HS : tid_o;
            const int o = (isx ? g.ox0 : g.oy0) + hidx - 1, lim_o = isx ? p.Wo : p.Ho, lim_i = isx ? p.Wi : p.Hi;
            const float f = (isx ? sx : sy) * (float)o;
            const int i0 = (int)f, i1 = i0 + (i0 < lim_i - 1);
            const int pitch = isx ? PIXB : g.pw * PIXB, org = isx ? g.px0 : g.py0;
            const bool ok = (unsigned)o < (unsigned)lim_o;
            sT0[tid_o] = ok ? (i0 - org) * pitch : -1;
            sT1[tid_o] = ok ? (i1 - org) * pitch : -1;
            sTL[tid_o] = f - (float)i0;
        }
        const int next = tile + gridDim.x;
        const TileGeom gn = geom(next < ntiles ? next : tile);

        f32x16 acc[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.0f;
        const int px = l31 & 15, prow = 4 * q + (l31 >> 4);

        auto stage = [&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the stage's patch has landed
            __syncthreads();                                  // ... everybody's; and every wave is done with the previous stage's halo
            stamp(1 + 3 * st);
            // ---- halo planes of the stage out of the patch planes: value = bilinear(hi + lo) in fp32, stored as hi = op(v), lo = op(v - hi).
            //      A thread owns halo column hx, chunk c and a run of 5 / 5 / 4 / 4 halo rows, the two source rows of the current halo row
            //      held horizontally interpolated in registers (see the single-plane kernel)
            int tid_h = tid;
            asm volatile("" : "+v"(tid_h));  // (see tid_o above)
            {
                const int part = tid_h >> 7, rem = tid_h & 127;
                const int hx = rem >> 3, c = rem & 7;
                const int rb = part < 2 ? 5 * part : 10 + 4 * (part - 2), re = rb + (part < 2 ? 5 : 4);
                const int ox_0 = sT0[HS + hx], ox_1 = sT1[HS + hx];
                const float lx = sTL[HS + hx];
                const char* const pbase = sP + c * 16;
                char* const hcol = sH + hx * PIXP + c * 16;
                auto hlerp = [&](int row_off, f32x2 (&t)[4]) {
                    const u32x4 h0 = *(const u32x4*)(pbase + row_off + ox_0), h1 = *(const u32x4*)(pbase + row_off + ox_1);
                    const u32x4 l0 = *(const u32x4*)(pbase + PATCH_PLANE + row_off + ox_0), l1 = *(const u32x4*)(pbase + PATCH_PLANE + row_off + ox_1);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned dh0 = h0[w], dh1 = h1[w], dl0 = l0[w], dl1 = l1[w];  // (scalar copies: see bug 4 of DESIGN.md section 7)
                        const f32x2 a0 = op2_to_f32(dh0) + op2_to_f32(dl0), a1 = op2_to_f32(dh1) + op2_to_f32(dl1);
                        t[w] = a0 + lx * (a1 - a0);
                    }
                };
                f32x2 t0[4], t1[4];
                int c0 = -2, c1 = -2;
                int oy_0 = sT0[rb], oy_1 = sT1[rb];
                for (int hy = rb; hy < re; ++hy) {
                    const int hn = hy + 1 < HS ? hy + 1 : hy;
                    const int ny_0 = sT0[hn], ny_1 = sT1[hn];
                    const float ly = sTL[hy];
                    u32x4 outh = {0u, 0u, 0u, 0u}, outl = {0u, 0u, 0u, 0u};
                    if ((oy_0 | ox_0) >= 0) {
                        if (oy_0 != c0) {
                            if (oy_0 == c1) {
#pragma unroll
                                for (int w = 0; w < 4; ++w) t0[w] = t1[w];
                            } else {
                                hlerp(oy_0, t0);
                            }
                            c0 = oy_0;
                        }
                        if (oy_1 != c1) {
                            if (oy_1 == c0) {
#pragma unroll
                                for (int w = 0; w < 4; ++w) t1[w] = t0[w];
                            } else {
                                hlerp(oy_1, t1);
                            }
                            c1 = oy_1;
                        }
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const f32x2 o2 = t0[w] + ly * (t1[w] - t0[w]);
                            const opx2 hh = to_op2(o2);
                            outh[w] = __builtin_bit_cast(unsigned, hh);
                            outl[w] = __builtin_bit_cast(unsigned, to_op2_bounded(o2 - __builtin_convertvector(hh, f32x2)))  /* |residue| <= ulp(hi) / 2: no saturation */;
                        }
                    }
                    *(u32x4*)(hcol + hy * ROWB) = outh;
                    *(u32x4*)(hcol + HALO_BYTES + hy * ROWB) = outl;
                    oy_0 = ny_0; oy_1 = ny_1;
                }
            }
            // the two right-most halo columns (16, 17): one (row, chunk) item per thread, full bilinear from four pixel reads per plane
            if (tid_h < 2 * HS * NCH) {
                const int c = tid_h % NCH, rest = tid_h / NCH;
                const int hy = rest % HS, hx = TS + rest / HS;
                const int oy_0 = sT0[hy], oy_1 = sT1[hy], ox_0 = sT0[HS + hx], ox_1 = sT1[HS + hx];
                u32x4 outh = {0u, 0u, 0u, 0u}, outl = {0u, 0u, 0u, 0u};
                if ((oy_0 | ox_0) >= 0) {
                    const float ly = sTL[hy], lx = sTL[HS + hx];
                    const char* base = sP + c * 16;
                    // (in two 8-byte halves: the 144 weight registers + 32 accumulators leave ~70 VGPRs for this phase)
#pragma unroll
                    for (int hw = 0; hw < 2; ++hw) {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        u32x2 v[2][4];
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            v[pl][0] = *(const u32x2*)(base + pl * PATCH_PLANE + oy_0 + ox_0 + hw * 8);
                            v[pl][1] = *(const u32x2*)(base + pl * PATCH_PLANE + oy_0 + ox_1 + hw * 8);
                            v[pl][2] = *(const u32x2*)(base + pl * PATCH_PLANE + oy_1 + ox_0 + hw * 8);
                            v[pl][3] = *(const u32x2*)(base + pl * PATCH_PLANE + oy_1 + ox_1 + hw * 8);
                        }
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            f32x2 a[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned dh = v[0][k][w], dl = v[1][k][w];
                                a[k] = op2_to_f32(dh) + op2_to_f32(dl);
                            }
                            const f32x2 u0 = a[0] + lx * (a[1] - a[0]);
                            const f32x2 u1 = a[2] + lx * (a[3] - a[2]);
                            const f32x2 o2 = u0 + ly * (u1 - u0);
                            const opx2 hh = to_op2(o2);
                            outh[2 * hw + w] = __builtin_bit_cast(unsigned, hh);
                            outl[2 * hw + w] = __builtin_bit_cast(unsigned, to_op2_bounded(o2 - __builtin_convertvector(hh, f32x2)))  /* |residue| <= ulp(hi) / 2: no saturation */;
                        }
                    }
                }
                *(u32x4*)(sH + hy * ROWB + hx * PIXP + c * 16) = outh;
                *(u32x4*)(sH + HALO_BYTES + hy * ROWB + hx * PIXP + c * 16) = outl;
            }
            __syncthreads();
            stamp(2 + 3 * st);
            // ---- the next stage's (or the next tile's first) patch streams in under this stage's MFMA phase
            if (st + 1 < NST) issue_patch(g, st + 1);
            else if (next < ntiles) issue_patch(gn, 0);
            // ---- implicit GEMM out of LDS over the stage's k-steps: plane lo first, then hi (the pass order of the two-pass GEMM form);
            //      step = (plane, tap, k-step of the stage); fragment reads run a pair of steps ahead of the MFMAs
            const char* const hbase = sH + prow * ROWB + px * PIXP + ((2 * grp + half) << 4);
            auto frag = [&](int step, int blk) -> opx8 {
                const int pl = step / (9 * KGS), r = step - pl * (9 * KGS);
                const int tap = r / KGS, kgl = r - tap * KGS;
                const int ky = tap / 3, kx = tap - 3 * ky;
                return *(const opx8*)(hbase + (pl ? 0 : HALO_BYTES) + (ky + 2 * blk) * ROWB + kx * PIXP + kgl * 64);
            };
            constexpr int NSTEP = 2 * 9 * KGS;
            opx8 xf[2][2][2];  // [buffer][step of the pair][block]
#pragma unroll
            for (int u = 0; u < 2; ++u) { xf[0][u][0] = frag(u, 0); xf[0][u][1] = frag(u, 1); }
#pragma unroll
            for (int pair = 0; pair < NSTEP / 2; ++pair) {
                const int cur = pair & 1;
                constexpr int dummy = 0; (void)dummy;
                const int s0 = 2 * pair, s1 = 2 * pair + 1;
                const int r0 = s0 % (9 * KGS), r1 = s1 % (9 * KGS);
                acc[0] = MDPT_MFMA_32x32x16(wreg[r0 / KGS][st * KGS + r0 % KGS], xf[cur][0][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pair + 1 < NSTEP / 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) { xf[cur ^ 1][u][0] = frag(2 * pair + 2 + u, 0); xf[cur ^ 1][u][1] = frag(2 * pair + 2 + u, 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = MDPT_MFMA_32x32x16(wreg[r0 / KGS][st * KGS + r0 % KGS], xf[cur][0][1], acc[1], 0, 0, 0);
                acc[0] = MDPT_MFMA_32x32x16(wreg[r1 / KGS][st * KGS + r1 % KGS], xf[cur][1][0], acc[0], 0, 0, 0);
                acc[1] = MDPT_MFMA_32x32x16(wreg[r1 / KGS][st * KGS + r1 % KGS], xf[cur][1][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(3 + 3 * st);
        };
        stage(std::integral_constant<int, 0>{});
        if constexpr (NST > 1) stage(std::integral_constant<int, 1>{});

        // ---- the groups swap one block each through the (now dead) halo space, summing group 0's partial + group 1's
        __syncthreads();  // every wave is done reading the halo
        {
            f32x4* mine = (f32x4*)sX + (size_t)wave * 4 * 64 + lane;
            const f32x16& give = acc[grp ^ 1];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) mine[r4 * 64] = f32x4{give[4 * r4], give[4 * r4 + 1], give[4 * r4 + 2], give[4 * r4 + 3]};
        }
        __syncthreads();
        float sacc = 0.0f;
        {
            const f32x4* theirs = (const f32x4*)sX + (size_t)(wave ^ 4) * 4 * 64 + lane;
            const f32x4* cb = (const f32x4*)sC;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 th = theirs[r4 * 64], b4 = cb[2 * r4 + half], w4 = cb[8 + 2 * r4 + half];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mine_v = grp == 0 ? acc[0][4 * r4 + e] : acc[1][4 * r4 + e];
                    const float tot = grp == 0 ? mine_v + th[e] : th[e] + mine_v;
                    sacc += fmaxf(tot + b4[e], 0.0f) * w4[e];
                }
            }
        }
        sacc += __shfl_xor(sacc, 32);
        sacc += head_b;
        const float dv = p.sigmoid ? 1.0f / (1.0f + __expf(-sacc)) : fmaxf(sacc, 0.0f);
        const int oy = g.oy0 + prow + 2 * grp, ox = g.ox0 + px;
        if (half == 0 && oy < p.Ho && ox < p.Wo) {
            const size_t o = ((size_t)g.b * p.Ho + oy) * p.Wo + ox;
            if (p.out_dtype == MDPT_DT_BF16) ((__bf16*)p.out)[o] = (__bf16)dv;
            else if (p.out_dtype == MDPT_DT_F16) ((_Float16*)p.out)[o] = (_Float16)dv;
            else ((float*)p.out)[o] = dv;
        }
        stamp(7);
        g = gn;
        // (the next tile's tables are written by threads that have passed this tile's last barrier; its first stage's first barrier orders
        //  this tile's exchange reads before the next halo writes)
    }
}

template <int CIN>
int launch_cin2(const HeadTailParams& p, hipStream_t stream) {
    constexpr unsigned ROWB2 = ((HS * (64 * 2 + 16) + 255) / 256) * 256;
    constexpr unsigned LDS = 2 * HS * ROWB2 + 2 * PS * PS * 128 + 2 * HS * 16 + 256;  // halo planes, patch planes, tables, epilogue constants
    auto kern = head_tail2_kernel<CIN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int ntiles = p.B * ((p.Ho + TS - 1) / TS) * ((p.Wo + TS - 1) / TS);
    const int grid = ntiles < 256 ? ntiles : 256;
    static char prof_name[48] = "";
    if (!prof_name[0]) snprintf(prof_name, sizeof(prof_name), "head_tail2_kernel<%d>", CIN);
    MdptProfScope prof(prof_name, 2.0 * p.B * p.Ho * p.Wo * 32.0 * 9.0 * CIN, stream);  // algorithmic flops (one pass)
#ifdef MDPT_DEBUG_SWITCHES  // A/B builds only
    static const bool dbg_on = getenv("MDPT_HEAD_DBG") != nullptr;
    if (dbg_on) {  // debug hook only (allocates and synchronises): phase stamps of every workgroup's second tile
        static unsigned long long* dbuf = nullptr;
        if (!dbuf && hipMalloc((void**)&dbuf, 256 * 8 * sizeof(unsigned long long)) != hipSuccess) return (int)hipErrorOutOfMemory;
        (void)hipMemsetAsync(dbuf, 0, 256 * 8 * sizeof(unsigned long long), stream);
        HeadTailParams q = p;
        q.dbg_times = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, q);
        (void)hipStreamSynchronize(stream);
        static unsigned long long host[256 * 8];
        (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
        double d[7] = {0, 0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int w = 0; w < grid; ++w)
            if (host[w * 8 + 7] > host[w * 8]) { for (int k = 0; k < 7; ++k) d[k] += (double)(host[w * 8 + k + 1] - host[w * 8 + k]); ++n; }
        if (n) fprintf(stderr, "head_tail2<%d> phases (cycles, mean of %d workgroups' 2nd tile): tables+patch wait %.0f | halo 0 %.0f | mfma 0 %.0f | patch wait 1 %.0f | halo 1 %.0f | mfma 1 %.0f | exchange + epilogue %.0f\n",
                       CIN, n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    return (int)hipGetLastError();
}

template <int CIN>
int launch_cin(const HeadTailParams& p, hipStream_t stream) {
    constexpr unsigned LDS = HS * (((HS * (CIN * 2 + 16) + 255) / 256) * 256) + PS * PS * CIN * 2 + 8 * 16 * 64 * 4 + 2 * HS * 16 + 256;  // halo, patch, exchange, tables, epilogue constants
    auto kern = head_tail_kernel<CIN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int ntiles = p.B * ((p.Ho + TS - 1) / TS) * ((p.Wo + TS - 1) / TS);
    const int grid = ntiles < 256 ? ntiles : 256;  // persistent: the weights are loaded into registers once per workgroup
    static char prof_name[48] = "";
    if (!prof_name[0]) snprintf(prof_name, sizeof(prof_name), "head_tail_kernel<%d>", CIN);
    MdptProfScope prof(prof_name, 2.0 * p.B * p.Ho * p.Wo * 32.0 * 9.0 * CIN, stream);
#ifdef MDPT_DEBUG_SWITCHES  // A/B builds only
    static const bool dbg_on = getenv("MDPT_HEAD_DBG") != nullptr;
#else
    constexpr bool dbg_on = false;
#endif
    if (dbg_on) {  // debug hook only (allocates and synchronises): phase stamps of every workgroup's second tile
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) hipMalloc((void**)&dbuf, 256 * 8 * sizeof(unsigned long long));
        hipMemsetAsync(dbuf, 0, 256 * 8 * sizeof(unsigned long long), stream);
        HeadTailParams q = p;
        q.dbg_times = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, q);
        hipStreamSynchronize(stream);
        static unsigned long long host[256 * 8];
        hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
        double d[5] = {0, 0, 0, 0, 0};
        int n = 0;
        for (int w = 0; w < grid; ++w)
            if (host[w * 8 + 5] > host[w * 8]) { for (int k = 0; k < 5; ++k) d[k] += (double)(host[w * 8 + k + 1] - host[w * 8 + k]); ++n; }
        if (n) fprintf(stderr, "head_tail<%d> phases (cycles, mean of %d workgroups' 2nd tile): tables+patch wait %.0f | halo %.0f | mfma %.0f | exchange wait %.0f | epilogue %.0f\n",
                       CIN, n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

bool MDPT_FN(mdpt_head_tail_supported)(int cin) { return cin == 64 || cin == 128; }

// the 18 halo pixels of a tile side must interpolate from at most PS source pixels: floor(17 * scale) + 2 <= PS with
// scale = (in - 1) / (out - 1) (x1.75 for patch 14: 11, x2 for patch 16: 10)
bool MDPT_FN(mdpt_head_tail_scale_ok)(int Hi, int Wi, int Ho, int Wo) {
    return Ho > 1 && Wo > 1 && (long)17 * (Hi - 1) < (long)(PS - 1) * (Ho - 1) && (long)17 * (Wi - 1) < (long)(PS - 1) * (Wo - 1);
}

int MDPT_FN(mdpt_launch_head_tail)(const HeadTailParams& p, int cin, hipStream_t stream) {
    if (p.B <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ho <= 0 || p.Wo <= 0 || !MDPT_FN(mdpt_head_tail_scale_ok)(p.Hi, p.Wi, p.Ho, p.Wo)) return (int)hipErrorInvalidValue;
    if (p.src_lo) {  // hi + lo planes of the first conv's output: the two-pass (activation-split) form
        if (cin == 128) return launch_cin2<128>(p, stream);
        if (cin == 64) return launch_cin2<64>(p, stream);
        return (int)hipErrorInvalidValue;
    }
    if (cin == 128) return launch_cin<128>(p, stream);
    if (cin == 64) return launch_cin<64>(p, stream);
    return (int)hipErrorInvalidValue;
}
