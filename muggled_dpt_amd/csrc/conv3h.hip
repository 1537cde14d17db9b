// Halo-staged 3x3 convolution (stride 1, pad 1, Cout = 256) for gfx950 (MI355X): the 256-channel residual-conv-unit and projection convs
// of the DPT decoder. Reference call sites: FusionBlock / ResidualConv2D  v2_depthanything/fusion_model.py:148-154,178-182,210-220
// (relu -> conv3x3 -> relu -> conv3x3 + skip, and `x = RCU(reassembled) + upsampled previous level`), the reassembly model's final 3x3
// projection  reassembly_model.py:135,302-309.
//
// The implicit-GEMM form in gemm.hip (A-row generator MDPT_A_CONV3) re-fetches every input pixel from L2 once per tap: nine 32 KB A tiles
// per 64-channel block and output tile (measured 1.25 GB fetched per launch against ~0.2 GB of unique operands). Here an output tile is a
// 16x16 pixel SQUARE of one image (M = 256) times all 256 output channels, and the 18x18-pixel input patch the nine taps of a 64-channel
// block touch is staged in LDS ONCE (41 KB, LDS-DMA, one channel block ahead, spread over the K tiles of the current block). The nine
// taps are nine K tiles whose A fragments are shifted ds_read_b128 of that halo patch:
//
//     K order: k = (cb * 9 + ky * 3 + kx) * 64 + c   (cb = 64-channel block; weights are packed in this order, MDPT_PACK_CONV3)
//     halo image of block cb: pixel (y', x') of the 18x18 patch at  y' * 2304 + x' * 128 + ((chunk ^ (x' & 7)) << 4)
//         - 2304 = 9 * 256: a row shift (ky) does not move the bank pattern; the x-dependent XOR makes the 16 pixels x 16 bytes of every
//           ds_read_b128 service group land on 16 different 16-byte slots of the 256-byte bank row for kx = 0, 1, 2 (checked exhaustively)
//         - the lane part of the address depends on kx only: three lane constants, (cb, ky) enter through one scalar add
//     pixels outside the image read the zero page (= the conv's zero padding); partial tiles compute and drop the rows outside
//
// Main loop = the 8-phase schedule of gemm8_kernel (two wave groups one barrier apart, 16 MFMA 16x16x32 per phase, B half-tiles re-staged
// one / two phases after their last read, counted vmcnt): the B (weight) side is identical; the A side has no per-K-tile DMA at all - one
// halo DMA instruction per wave and K tile (dummy 16-byte-per-lane reads of the zero page into a scratch KiB once the 41 real ones are
// issued, so that the vmcnt arithmetic is the same in every K tile).
// Epilogues are the direct register -> global forms (swapped MFMA operands: a lane owns 4 consecutive channels of one pixel):
//     out = ((conv + bias) [+ x2 bilinear of the coarser level]) [+ skip]  ->  fp32 map and / or bf16 map (optionally ReLU'd)
// `skip` (fp32) is added in the epilogue (its tile is pulled towards the CU by one prefetch DMA per K tile of the main loop); the coarser
// level's 10x10 source patch of the bilinear add is staged in the (then free) LDS once per tile.
// Summation order per output element = K order above, fp32 accumulation in the MFMA pipe, then ((sum + bias) + up) + skip: the generic
// kernels of gemm.hip walk K in the same order and apply the same epilogue expressions in the same order, so a tile-rule change between
// batch sizes does not change a bit (tests/test_gpu_conv3h.py compares the two paths bit for bit).

#include "mdpt_kernels.h"
#include "mdpt_prof.h"
#include "up_bf16.h"
#include "f8_cross.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#pragma clang fp contract(off)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int HROW = 18 * 128;                 // bytes per halo row: 18 pixels x 64 channels bf16
constexpr int HALO_INSTR = 41;                 // ceil(18 * 18 * 128 / 1024) LDS-DMA instructions per halo patch
constexpr int HALO_BYTES = HALO_INSTR * 1024;
constexpr int BTILE = 256 * 128;               // one K tile of weights: 256 rows x 64 k, bf16
constexpr int OFF_H = 2 * BTILE, OFF_SCR = OFF_H + 2 * HALO_BYTES;
constexpr int LDS_BYTES = OFF_SCR + 1024;
constexpr int UPW = 10;                        // side of the coarse patch of the bilinear add (16 fine pixels span <= 9.x coarse ones)
static_assert(UPW * UPW * 1024 <= LDS_BYTES, "coarse patch must fit the ring");

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned long long memtime_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)(bytes < 0xFFFFFFF0ull ? bytes : 0xFFFFFFF0ull), 0x00020000);
}

// NP = 3: split precision (x = hi + lo planes for the input and the weights, hi / lo output planes): the K loop runs three times,
// A_lo * W_hi, A_hi * W_lo, A_hi * W_hi, into the same fp32 accumulators - the pass order of the implicit-GEMM kernels. A pass is just
// ncb more channel blocks whose halo patches / weight tiles come from the other operand planes ("virtual" channel block vcb = pass * ncb + cb).
// NP = 2: the activation-split form, A_lo * W_hi then A_hi * W_hi (input hi + lo planes, ONE weight plane; GemmParams::npass == 2).
// BFOUT: bf16 output planes (false: fp32 map only - the bf16x3 head keeps the fp32 map for its bilinear upsample).
// UPIN (128-channel bf16 form only): the conv's input is the x2 bilinear upsample (align_corners=True) of a bf16 map `up_in` that never
// exists in memory - the 18x18 halo patch of a channel block is INTERPOLATED in LDS from the <= 11x11 source pixels it touches (arithmetic
// of up_bf16.h, rounded to bf16 like the stand-alone upsample kernel): SpatialUpsampleLayer + head conv 1, fusion_model.py:182 ->
// head_model.py:74-76 fused.
// F8 (fp16 build, NP = 2 or 3, Cin % 256 == 0): the cross-term passes run on fp8 planes (f8_cross.h, Conv3hParams::f8): in_lo is the e5m2 residue
// plane, the e5m2 plane of the values sits a8_off bytes behind it, the weights come as e4m3 planes in 128-channel-block K order with one E8M0
// scale per output row. A "virtual channel block" of a cross-term pass is 128 channels - the same 41 KB halo patch and the same 32 KB weight
// K tiles as a 64-channel fp16 block, half as many of them per pass, 8 block-scaled 16x16x128 MFMAs per phase instead of 16.
template <int NQN, int NP, bool SKIP, bool F32OUT, bool RELU, bool UP, bool BFOUT, bool UPIN = false, bool F8 = false>
__global__ __launch_bounds__(512, 1) void conv3h_kernel(const Conv3hParams p) {
    static_assert(NP >= 1 && NP <= 3, "passes");
    static_assert(!F8 || (MDPT_HAVE_F8 && NP >= 2 && !UPIN), "fp8 cross terms: fp16 build, two or three terms");
    constexpr bool X3 = NP >= 2;   // the input has a lo plane (pass 0 reads it) and the output planes are split
    constexpr bool W3 = NP == 3;   // the weights have a lo plane (pass 1 of 3)
    static_assert(!UPIN || (NQN == 1 && !X3 && BFOUT && !F32OUT), "the upsampled-input form exists for the bf16 head conv");
    constexpr int COUT = 128 * NQN;  // output channels: 256 (two 128-column halves per wave) or 128
    static_assert(NQN == 2 || (!SKIP && !UP && !RELU), "the 128-channel form has the bias-only epilogues");
    static_assert(BFOUT || F32OUT, "no output");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned long long t_start = 0, t_first = 0, t_loop = 0;
    if (p.dbg_times) t_start = memtime_now();

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lh = lane >> 4;

    // ---- tile -> (image, tile row, tile column); the dispatcher places block b on XCD b % 8: every XCD gets a contiguous run of tiles
    //      (neighbouring tiles share halo rows / columns in that XCD's L2)
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
        tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    }
    const int img = tile / (tiles_x * tiles_y), trem = tile - img * (tiles_x * tiles_y);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int Y0 = ty * 16, X0 = tx * 16;
    const op_t* const in_img = UPIN ? p.up_in + (size_t)img * p.Hs * p.Ws * p.Cin : p.in + (size_t)img * p.H * p.W * p.Cin;
    const int ncb = p.Cin >> 6;

    // ---- operand staging: LDS-DMA through BUFFER descriptors (buffer_load_dwordx4 ... offen lds). The per-lane byte offsets are
    //      constants of the tile, the position along K is a scalar offset, and a lane whose offset fails the descriptor's bounds check
    //      gets ZEROS written to LDS (tools/probes/buffer_lds_oob.hip): conv zero padding, the slack behind halo pixel 323 and dummy
    //      instructions need no zero page, no 64-bit pointer arithmetic and no selects inside the loop.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int Kw = p.Cin * 9;
    const __amdgpu_buffer_rsrc_t rs_w_hi = plane_rsrc(p.w, (size_t)COUT * Kw * 2);
    const __amdgpu_buffer_rsrc_t rs_w_lo = plane_rsrc(W3 ? p.w_lo : p.w, (size_t)COUT * Kw * 2);
    const __amdgpu_buffer_rsrc_t rs_in_hi = plane_rsrc(in_img, UPIN ? (size_t)p.Hs * p.Ws * p.Cin * 2 : (size_t)p.H * p.W * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rs_in_lo = F8 ? plane_rsrc((const unsigned char*)p.in_lo + (size_t)img * p.H * p.W * p.Cin, (size_t)p.H * p.W * p.Cin)
                                               : plane_rsrc(X3 ? p.in_lo + (size_t)img * p.H * p.W * p.Cin : in_img, (size_t)p.H * p.W * p.Cin * 2);
    // F8: e5m2 plane of the values (three terms), e4m3 weight planes
    const __amdgpu_buffer_rsrc_t rs_in_a8 = (F8 && NP == 3) ? plane_rsrc((const unsigned char*)p.in_lo + p.a8_off + (size_t)img * p.H * p.W * p.Cin, (size_t)p.H * p.W * p.Cin) : rs_in_hi;
    const __amdgpu_buffer_rsrc_t rs_w8 = F8 ? plane_rsrc(p.w8, (size_t)COUT * Kw) : rs_w_hi;
    const __amdgpu_buffer_rsrc_t rs_w8_lo = (F8 && NP == 3) ? plane_rsrc(p.w8_lo, (size_t)COUT * Kw) : rs_w_hi;
    const int ncb8 = F8 ? p.Cin >> 7 : 0;                 // 128-channel blocks of a cross-term pass (even: launcher)
    const int nvcb8 = (NP - 1) * ncb8;                    // ... of all cross-term passes
    const int nvcb = F8 ? nvcb8 + ncb : NP * ncb;         // channel blocks of all passes
    const int T = F8 ? 9 * ncb8 : 9 * ncb;                // K tiles per (cross-term) pass; ncb is even (the launcher checks Cin % 128 == 0)
    // K tile kt of the whole loop -> (operand plane of the weights, byte offset of its K tile inside the plane)
    auto w_pass = [&](int kt) __attribute__((always_inline)) -> int { return NP == 3 ? (kt >= 2 * T ? 2 : (kt >= T ? 1 : 0)) : (NP == 2 ? (kt >= T ? 1 : 0) : 0); };
    // B (weights): DMA instruction i of a wave stages rows r = 8 * (wave + 8 i) + lane / 8 = r0 + 64 i; 16-byte slot lane % 8 holds k-chunk
    // slot ^ ((r >> 1) & 7) (the key is the same for all four i). One lane constant; (K tile kt, i) enter through the scalar offset.
    const unsigned b_voff = (unsigned)((wave * 8 + (lane >> 3)) * Kw + (((lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7)) << 3)) * 2u;
    // F8: byte planes - the row part of a lane offset (a multiple of 256 bytes in the fp16 plane: Kw and Cin are multiples of 128) halves, the
    // 16-byte slot inside the 128-byte K tile stays: off8 = (off16 & ~255) / 2 | (off16 & 255); an out-of-range marker stays out of range
    // (computed where it is used: `salt` is 0, derived from the loop counter so that the conversion is not loop-invariant - hoisted out of the
    // loops the seven converted offsets would be seven more registers for the whole kernel, which the 256-channel forms do not have: 249 of 256 are taken)
    auto to8 = [](unsigned off16, unsigned salt) __attribute__((always_inline)) -> unsigned { return ((off16 & ~255u) >> (1u + salt)) | (off16 & 255u); };
    // F8: the pass a loop belongs to is a COMPILE-TIME constant of the call site (one loop per pass), and what it stages - K tile kt + 2, channel
    // block cb + 1 - lies in that pass or at the start of the next one: every descriptor choice is a two-way scalar select (a three-way
    // choice among descriptors by computed conditions becomes a table in private memory and a waterfall loop around every DMA instruction).
    auto rs_w_of = [&](auto pc) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        constexpr int P = decltype(pc)::value;
        if constexpr (!F8) return rs_w_hi;
        else if constexpr (P >= NP - 1) return rs_w_hi;
        else if constexpr (P == 1) return rs_w8_lo;
        else return rs_w8;
    };
    auto rs_a_of = [&](auto pc) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        constexpr int P = decltype(pc)::value;
        if constexpr (!F8) return rs_in_hi;
        else if constexpr (P >= NP - 1) return rs_in_hi;
        else if constexpr (P == 1) return rs_in_a8;
        else return rs_in_lo;
    };
    auto issue_b = [&](auto pc, int half, int buf, int kt) __attribute__((always_inline)) {  // half-tile `half` (128 weight rows) of K tile kt -> B buffer `buf`
        if constexpr (F8) {
            constexpr int P = decltype(pc)::value, PN = P + 1 < NP ? P + 1 : P;
            const bool nxt = P + 1 < NP && kt >= (P + 1) * T;              // (wave-uniform) the tile opens the next pass
            const __amdgpu_buffer_rsrc_t rs = nxt ? rs_w_of(std::integral_constant<int, PN>{}) : rs_w_of(pc);
            const bool f8t = nxt ? PN < NP - 1 : P < NP - 1;
            const int soff = (kt - (nxt ? PN : P) * T) * 128;
            const unsigned vo = f8t ? to8(b_voff, (unsigned)kt >> 30) : b_voff;
            const int rstep = f8t ? 64 * Kw : 128 * Kw;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((i >> 1) == half)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BTILE + (wave + 8 * i) * 1024), 16, vo, soff + i * rstep, 0, 0);
            return;
        }
        const int ps = w_pass(kt);
        const __amdgpu_buffer_rsrc_t rs = (W3 && ps == 1) ? rs_w_lo : rs_w_hi;
        const int soff = (kt - ps * T) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((i >> 1) == half)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BTILE + (wave + 8 * i) * 1024), 16, b_voff, soff + i * 128 * Kw, 0, 0);
    };
    auto issue_b128 = [&](auto pc, int buf4, int kt) __attribute__((always_inline)) {  // COUT = 128: the whole 16 KB K tile kt -> slot buf4 of a FOUR-deep ring (two instructions per wave)
        if constexpr (F8) {
            constexpr int P = decltype(pc)::value, PN = P + 1 < NP ? P + 1 : P;
            const bool nxt = P + 1 < NP && kt >= (P + 1) * T;
            const __amdgpu_buffer_rsrc_t rs = nxt ? rs_w_of(std::integral_constant<int, PN>{}) : rs_w_of(pc);
            const bool f8t = nxt ? PN < NP - 1 : P < NP - 1;
            const int soff = (kt - (nxt ? PN : P) * T) * 128;
            const unsigned vo = f8t ? to8(b_voff, (unsigned)kt >> 30) : b_voff;
            const int rstep = f8t ? 64 * Kw : 128 * Kw;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf4 * (BTILE / 2) + (wave + 8 * i) * 1024), 16, vo, soff + i * rstep, 0, 0);
            return;
        }
        const int ps = w_pass(kt);
        const __amdgpu_buffer_rsrc_t rs = (W3 && ps == 1) ? rs_w_lo : rs_w_hi;
        const int soff = (kt - ps * T) * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf4 * (BTILE / 2) + (wave + 8 * i) * 1024), 16, b_voff, soff + i * 128 * Kw, 0, 0);
    };
    // A (halo): DMA instruction c = wave + 8 j of a patch covers halo bytes [1024 c, 1024 c + 1024) = 8 pixels x 8 slots; channel block
    // cb at soffset 128 cb. Instructions 41..47 (j = 5 of waves 1-7) do not exist: all lanes out of range, destination = the scratch KiB.
    unsigned h_voff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = wave + 8 * j;
        const int q = c * 8 + (lane >> 3), slot = lane & 7;
        const int yy = (q * 3641) >> 16, xx = q - yy * 18;  // q / 18, q % 18 for q < 400
        const int Y = Y0 + yy - 1, X = X0 + xx - 1;
        const bool ok = c < HALO_INSTR && q < 324 && (unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W;
        h_voff[j] = ok ? ((unsigned)(Y * p.W + X) * (unsigned)p.Cin + (unsigned)((slot ^ (xx & 7)) << 3)) * 2u : OOB;
    }
    auto issue_halo = [&](auto pc, int j, int vcbn, int hbn, bool dummy) __attribute__((always_inline)) {  // dummy (wave-uniform): no next channel block - zeros into the scratch KiB
        const int c = wave + 8 * j;
        const bool real = c < HALO_INSTR && !dummy;
        if constexpr (F8) {
            // virtual blocks [0, ncb8): residue plane, [ncb8, nvcb8) (three terms): the values' e5m2 plane, then the fp16 hi plane's 64-channel blocks
            constexpr int P = decltype(pc)::value, PN = P + 1 < NP ? P + 1 : P;
            const int endp = P < NP - 1 ? (P + 1) * ncb8 : nvcb;
            const bool nxt = P + 1 < NP && vcbn >= endp;               // (wave-uniform) the block opens the next pass
            const __amdgpu_buffer_rsrc_t rs = nxt ? rs_a_of(std::integral_constant<int, PN>{}) : rs_a_of(pc);
            const bool f8b = nxt ? PN < NP - 1 : P < NP - 1;
            const int cbn = vcbn - (nxt ? PN : P) * ncb8;                // (the fp16 pass starts at virtual block (NP - 1) ncb8)
            const unsigned voff = dummy ? OOB : (f8b ? to8(h_voff[j], (unsigned)vcbn >> 30) : h_voff[j]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (real ? OFF_H + hbn * HALO_BYTES + c * 1024 : OFF_SCR)), 16, voff, cbn * 128, 0, 0);
            return;
        }
        const unsigned voff = dummy ? OOB : h_voff[j];
        const bool lo_plane = X3 && vcbn < ncb;  // pass 0 reads the lo plane of the input
        const int cbn = X3 ? (vcbn >= 2 * ncb ? vcbn - 2 * ncb : (vcbn >= ncb ? vcbn - ncb : vcbn)) : vcbn;  // (NP == 2: vcbn < 2 ncb)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lo_plane ? rs_in_lo : rs_in_hi, (lds_ptr_t)(smem + (real ? OFF_H + hbn * HALO_BYTES + c * 1024 : OFF_SCR)), 16, voff,
                                                 cbn * 128, 0, 0);
    };
    // SKIP: the fp32 skip tile (256 KB, read once, no reuse) is added in the EPILOGUE (((conv + bias) + up) + skip, the order of the generic
    // epilogue); one more DMA instruction per wave and K tile pulls pixel `8 slot + wave` of it towards the CU (its 1 KiB lands in the
    // scratch KiB and is ignored: an L2 / MALL prefetch), so the epilogue's loads do not start from HBM with nothing to hide behind.
    const __amdgpu_buffer_rsrc_t rs_skip = plane_rsrc(SKIP ? p.skip + (size_t)img * p.H * p.W * 256 : nullptr, SKIP ? (size_t)p.H * p.W * 1024 : 0);
    // (not in the fp8 forms: two main loops at the 256-register limit - with the prefetch's operands live the fp16 loop reloaded a dozen spilled
    //  registers per pair of channel blocks, a vmcnt(0) each)
    constexpr bool PFSKIP = SKIP && !F8;
    const unsigned pf_voff = (unsigned)lane << 4;
    auto issue_prefetch = [&](int slot) __attribute__((always_inline)) {
        const int pp = slot * 8 + wave;  // tile pixel (row pp >> 4, column pp & 15); slot outside [0, 32): nothing to fetch
        const int Y = Y0 + (pp >> 4), X = X0 + (pp & 15);
        const bool ok = (unsigned)slot < 32u && Y < p.H && X < p.W;  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_skip, (lds_ptr_t)(smem + OFF_SCR), 16, ok ? pf_voff : OOB, ok ? (Y * p.W + X) * 1024 : 0, 0, 0);
    };

    // fragment read offsets. B as in gemm8 (row l15 of a 16-row block, k-chunk lh / 4 + lh, key (row >> 1) & 7);
    // A: pixel x' = l15 + kx of halo row (4 grp + ky + 8 qm + i), k-chunk lh (kk = 0) / 4 + lh (kk = 1: address ^ 64), key x' & 7
    const int bkey = (l15 >> 1) & 7;
    const int b_off0 = (wc * 32 + l15) * 128 + ((lh ^ bkey) << 4), b_off1 = (wc * 32 + l15) * 128 + (((4 + lh) ^ bkey) << 4);
    int a_lane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a_lane[kx] = ((l15 + kx) << 7) + (((lh ^ (l15 + kx)) & 7) << 4) + grp * 4 * HROW + OFF_H;

    f32x4 acc[2][NQN][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NQN; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    opx8 fa[4][2], fb0[2][2], fb1[2][2];
#define PC0 (std::integral_constant<int, 0>{})
#define PC1 (std::integral_constant<int, 1>{})
#define PCL (std::integral_constant<int, NP - 1>{}) /* the fp16 pass (the only pass of the single-pass forms) */
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { PIN(); __builtin_amdgcn_s_barrier(); PIN(); } while (0)
#define LOAD_A(QM_, VA_, ROW0_)                                                                                       \
    do {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
            fa[i][0] = *(const opx8*)(smem + (VA_) + ((ROW0_) + (QM_) * 8 + i) * HROW);                              \
            fa[i][1] = *(const opx8*)(smem + ((VA_) ^ 64) + ((ROW0_) + (QM_) * 8 + i) * HROW);                       \
        }                                                                                                             \
    } while (0)
#define LOAD_B(DST_, QN_, BUF_)                                                                                       \
    do {                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
            DST_[j][0] = *(const opx8*)(smem + (BUF_) * BTILE + (QN_) * 16384 + j * 2048 + b_off0);                  \
            DST_[j][1] = *(const opx8*)(smem + (BUF_) * BTILE + (QN_) * 16384 + j * 2048 + b_off1);                  \
        }                                                                                                             \
    } while (0)
#define LOAD_B_AT(DST_, BYTES_) /* COUT = 128: ring slot at a run-time byte offset */                                 \
    do {                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
            DST_[j][0] = *(const opx8*)(smem + (BYTES_) + j * 2048 + b_off0);                                        \
            DST_[j][1] = *(const opx8*)(smem + (BYTES_) + j * 2048 + b_off1);                                        \
        }                                                                                                             \
    } while (0)
#define MFMA_Q(QM_, QN_, FB_)                                                                                         \
    do {                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                    acc[QM_][QN_][i][j] = MDPT_MFMA_16x16x32(FB_[j][kk], fa[i][kk], acc[QM_][QN_][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
#if MDPT_HAVE_F8
    // E8M0 scales of the weight rows this lane's 16-channel blocks read: byte 2 qn + j = output channel 128 qn + 32 wc + 16 j + l15
    int wsc = 0, wsc_lo = 0;
    const int asc = F8_A_SCALE;  // the activations' constant E8M0 scale (a VGPR operand of every scaled MFMA)
    if constexpr (F8) {
#pragma unroll
        for (int b4 = 0; b4 < 2 * NQN; ++b4) {
            const int n = (b4 >> 1) * 128 + wc * 32 + (b4 & 1) * 16 + l15;
            wsc |= (int)p.s8[n] << (8 * b4);
            if (NP == 3) wsc_lo |= (int)p.s8_lo[n] << (8 * b4);
        }
    }
#define MFMA_Q8_ONE(QM_, QN_, FB_, I_, J_, WS_)                                                                       \
    f8_mfma16_w_first<2 * (QN_) + (J_)>(acc[QM_][QN_][I_][J_], f8_cat(FB_[J_][0], FB_[J_][1]), f8_cat(fa[I_][0], fa[I_][1]), WS_, asc)
#define MFMA_Q8(QM_, QN_, FB_, WS_)                                                                                   \
    do {                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        MFMA_Q8_ONE(QM_, QN_, FB_, 0, 0, WS_); MFMA_Q8_ONE(QM_, QN_, FB_, 0, 1, WS_);                                 \
        MFMA_Q8_ONE(QM_, QN_, FB_, 1, 0, WS_); MFMA_Q8_ONE(QM_, QN_, FB_, 1, 1, WS_);                                 \
        MFMA_Q8_ONE(QM_, QN_, FB_, 2, 0, WS_); MFMA_Q8_ONE(QM_, QN_, FB_, 2, 1, WS_);                                 \
        MFMA_Q8_ONE(QM_, QN_, FB_, 3, 0, WS_); MFMA_Q8_ONE(QM_, QN_, FB_, 3, 1, WS_);                                 \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
#define MQ8A(QM_, QN_, FB_) MFMA_Q8(QM_, QN_, FB_, wsc)
#define MQ8B(QM_, QN_, FB_) MFMA_Q8(QM_, QN_, FB_, wsc_lo)
#endif
#define WAIT_LGKM(N_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" ::: "memory")
#define WAIT_VM_IMM(N_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory")

    if constexpr (UPIN) {
        // ---- 128 output channels, input = x2 upsample of p.up_in, interpolated per channel block:
        //   * source patch of block cb+1 (<= 11x11 pixels x 64 channels bf16, dense [ph * pw][128 B]): two DMA instructions per wave at tap 0 of
        //     block cb into the 16 KB behind a THREE-deep weight ring (K tile kt in slot kt % 3, staged two K tiles ahead, in P1);
        //   * halo patch of block cb+1: 2592 items of 8 channels (pixel, 16-byte chunk), item tid + 512 r is built at tap 3 + r of block cb:
        //     its four corner reads are issued in P1 behind the fragment reads, the arithmetic and the 16-byte LDS store run in P2;
        //     per-item constants (corner offsets, weights, destination) are tile constants held in registers;
        //   * everything else as in the plain 128-channel form. vmcnt at P2: the weights of tile kt+2 and the patch instructions issued in
        //     this and the previous K tile may stay in flight.
        constexpr int PATCH_OFF = 3 * (BTILE / 2);  // the fourth 16 KB of the weight region
        const float sy = (float)(p.Hs - 1) / (float)(p.H - 1), sx = (float)(p.Ws - 1) / (float)(p.W - 1);
        const int oy_first = Y0 == 0 ? 0 : Y0 - 1, ox_first = X0 == 0 ? 0 : X0 - 1;
        const int oy_last = min(Y0 + 16, p.H - 1), ox_last = min(X0 + 16, p.W - 1);
        const int py0 = (int)(sy * (float)oy_first), px0 = (int)(sx * (float)ox_first);
        const int ph = min((int)(sy * (float)oy_last) + 1, p.Hs - 1) - py0 + 1;
        const int pw = min((int)(sx * (float)ox_last) + 1, p.Ws - 1) - px0 + 1;  // <= 11 (launcher's scale check)
        unsigned pt_voff[2];  // patch DMA instruction c = wave + 8 j: patch pixels 8 c .. 8 c + 7, lane / 8 = pixel, lane % 8 = chunk
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = (wave + 8 * j) * 8 + (lane >> 3);
            const int yy = pp / pw, xx = pp - yy * pw;
            pt_voff[j] = pp < ph * pw ? ((unsigned)((py0 + yy) * p.Ws + px0 + xx) * (unsigned)p.Cin + (unsigned)((lane & 7) << 3)) * 2u : OOB;
        }
        auto issue_patch = [&](int cbn) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in_hi, (lds_ptr_t)(smem + PATCH_OFF + (wave + 8 * j) * 1024), 16, pt_voff[j], cbn * 128, 0, 0);
        };
        // item r of this thread: halo pixel (hy, hx), chunk; corner offsets inside the patch, weights, destination in the halo image
        int it_src[6], it_dst[6];
        float it_lx[6], it_ly[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int item = tid + 512 * r, pix = item >> 3, chunk = item & 7;
            const int hy = (pix * 3641) >> 16, hx = pix - hy * 18;
            const int oy = Y0 + hy - 1, ox = X0 + hx - 1;
            const bool inside = item < 2592 && (unsigned)oy < (unsigned)p.H && (unsigned)ox < (unsigned)p.W;
            const int oyc = min(max(oy, 0), p.H - 1), oxc = min(max(ox, 0), p.W - 1);
            const float fy = sy * (float)oyc, fx = sx * (float)oxc;
            const int y0 = (int)fy, x0 = (int)fx;
            const int dy = y0 < p.Hs - 1 ? 1 : 0, dx = x0 < p.Ws - 1 ? 1 : 0;
            it_ly[r] = fy - (float)y0; it_lx[r] = fx - (float)x0;
            // corner (0,0) byte offset | dx << 28 | dy << 29 | outside-the-image (zero padding) << 30 | no such item << 31
            it_src[r] = (((y0 - py0) * pw + (x0 - px0)) * 128 + chunk * 16) | (dx << 28) | (dy << 29) | (inside ? 0 : 1 << 30) | (item < 2592 ? 0 : 1 << 31);
            it_dst[r] = hy * HROW + hx * 128 + ((chunk ^ (hx & 7)) << 4);
        }
        mdpt_u32x4 cq[4];  // the four corners of the item in flight
        auto item_reads = [&](int r) {
            const int o = it_src[r] & 0xFFFFFF, ddx = ((it_src[r] >> 28) & 1) * 128, ddy = ((it_src[r] >> 29) & 1) * pw * 128;
            cq[0] = *(const mdpt_u32x4*)(smem + PATCH_OFF + o);
            cq[1] = *(const mdpt_u32x4*)(smem + PATCH_OFF + o + ddx);
            cq[2] = *(const mdpt_u32x4*)(smem + PATCH_OFF + o + ddy);
            cq[3] = *(const mdpt_u32x4*)(smem + PATCH_OFF + o + ddx + ddy);
        };
        auto item_write = [&](int r, int hbn) {
            mdpt_u32x4 v = mdpt_up_bf16x8(cq[0], cq[1], cq[2], cq[3], it_lx[r], it_ly[r]);
            if (it_src[r] & (1 << 30)) v = mdpt_u32x4{0u, 0u, 0u, 0u};
            if (it_src[r] >= 0) *(mdpt_u32x4*)(smem + OFF_H + hbn * HALO_BYTES + it_dst[r]) = v;
        };

        // prologue: patch 0 and K tiles 0, 1; then every thread builds its items of halo patch 0
        issue_patch(0);
        issue_b128(PC0, 0, 0);
        issue_b128(PC0, 1, 1);
        WAIT_VM_IMM(0);
        BAR();
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            item_reads(r);
            item_write(r, 0);
        }
        WAIT_LGKM(0);
        BAR();
        if (p.dbg_times) t_first = memtime_now();
        if (grp == 1) BAR();
#define CONV_KT128U(U_)                                                                                               \
        do {                                                                                                          \
            constexpr int cbl = (U_) / 9, t9 = (U_) % 9, ky = t9 / 3, kx = t9 % 3;                                     \
            constexpr int p_now = t9 == 0 ? 2 : 0, p_prev = t9 == 1 ? 2 : 0;  /* patch DMA instructions of this / the previous K tile */ \
            constexpr int slot = (U_) % 3, slot2 = ((U_) + 2) % 3, ritem = t9 - 3;                                     \
            const int kt = cbp * 9 + (U_);                                                                            \
            const bool more = (U_) < 16 || !last;            /* K tile kt + 2 exists */                               \
            const bool next_cb = cbl == 0 || !last;          /* channel block cb + 1 exists */                        \
            const int va = a_lane[kx] + cbl * HALO_BYTES;                                                             \
            LOAD_B_AT(fb0, slot * (BTILE / 2)); PIN(); LOAD_A(0, va, ky); PIN();                                      \
            if (more) issue_b128(PC0, slot2, kt + 2);                                                                 \
            if constexpr (p_now != 0) { if (next_cb) issue_patch(cbp + cbl + 1); }                                    \
            if constexpr (ritem >= 0) { if (next_cb) item_reads(ritem); }                                             \
            BAR(); WAIT_LGKM(0); PIN();                                                                               \
            MFMA_Q(0, 0, fb0); BAR();                                                                                 \
            LOAD_A(1, va, ky); PIN();                                                                                 \
            if constexpr (ritem >= 0) { if (next_cb) item_write(ritem, cbl ^ 1); }                                    \
            if (more) { if (next_cb) WAIT_VM_IMM(2 + p_now + p_prev); else WAIT_VM_IMM(2); } else { WAIT_VM_IMM(0); }  \
            if constexpr (t9 == 8) WAIT_LGKM(0);  /* the last halo stores are complete before the barrier both groups pass */ \
            BAR(); WAIT_LGKM(0); PIN();                                                                               \
            MFMA_Q(1, 0, fb0); BAR();                                                                                 \
        } while (0)
        for (int cbp = 0; cbp < nvcb; cbp += 2) {
            const bool last = cbp + 2 >= nvcb;
            CONV_KT128U(0); CONV_KT128U(1); CONV_KT128U(2); CONV_KT128U(3); CONV_KT128U(4); CONV_KT128U(5); CONV_KT128U(6); CONV_KT128U(7); CONV_KT128U(8);
            CONV_KT128U(9); CONV_KT128U(10); CONV_KT128U(11); CONV_KT128U(12); CONV_KT128U(13); CONV_KT128U(14); CONV_KT128U(15); CONV_KT128U(16); CONV_KT128U(17);
        }
#undef CONV_KT128U
    } else
    if constexpr (NQN == 1) {
        // ---- COUT = 128 (head conv 1, head_model.py:74-76): a K tile of weights is 16 KB, a wave owns two 64 x 32 quadrants (qm = 0, 1), so a
        //      K tile has TWO phases of 16 MFMAs and the weights live in a four-deep ring (K tile kt in slot kt & 3, staged three K tiles ahead):
        //   P1: read B, A rows 0-7 | halo DMA (t9 < 6) -> MFMA(0)      P2: read A rows 8-15 | B of tile kt+3, counted vmcnt -> MFMA(1)
        // vmcnt at P2: what this K tile and the previous one issued may stay in flight (2 weight + 0/1 halo instructions each); everything
        // older - tile kt+1's weights above all - has landed one phase and >= one workgroup barrier before P1 of tile kt+1 reads it.
#pragma unroll
        for (int j = 0; j < 6; ++j) issue_halo(PC0, j, 0, 0, false);
        issue_b128(PC0, 0, 0);
        issue_b128(PC0, 1, 1);
        issue_b128(PC0, 2, 2);
        WAIT_VM_IMM(4);
        BAR();
        if (p.dbg_times) t_first = memtime_now();
        if (grp == 1) BAR();
#define CONV_KT128(U_, MQ, PC_)                                                                                          \
        do {                                                                                                          \
            constexpr int cbl = (U_) / 9, t9 = (U_) % 9, ky = t9 / 3, kx = t9 % 3;                                     \
            constexpr int h_now = t9 < 6 ? 1 : 0, h_prev = ((U_) % 9 == 0) ? 0 : (((U_) - 1) % 9 < 6 ? 1 : 0);         \
            const int kt = cbp * 9 + (U_);                                                                            \
            const bool more = (U_) < 15 || !last;      /* K tile kt + 3 exists */                                     \
            const int va = a_lane[kx] + cbl * HALO_BYTES;                                                             \
            const int bslot = (kt & 3) * (BTILE / 2);                                                                 \
            LOAD_B_AT(fb0, bslot); PIN(); LOAD_A(0, va, ky); PIN();                                                   \
            if constexpr (h_now) { if (more) issue_halo(PC_, t9, cbp + cbl + 1, cbl ^ 1, cbl == 1 && last); }         \
            BAR(); WAIT_LGKM(0); PIN();                                                                               \
            MQ(0, 0, fb0); BAR();                                                                                 \
            LOAD_A(1, va, ky); PIN();                                                                                 \
            if (more) { issue_b128(PC_, (kt + 3) & 3, kt + 3); PIN(); WAIT_VM_IMM(4 + h_now + h_prev); } else { WAIT_VM_IMM(0); } \
            BAR(); WAIT_LGKM(0); PIN();                                                                               \
            MQ(1, 0, fb0); BAR();                                                                                 \
        } while (0)
#define CONV_PAIR128(MQ, PC_) CONV_KT128(0, MQ, PC_); CONV_KT128(1, MQ, PC_); CONV_KT128(2, MQ, PC_); CONV_KT128(3, MQ, PC_); CONV_KT128(4, MQ, PC_); CONV_KT128(5, MQ, PC_); CONV_KT128(6, MQ, PC_); CONV_KT128(7, MQ, PC_); CONV_KT128(8, MQ, PC_); \
            CONV_KT128(9, MQ, PC_); CONV_KT128(10, MQ, PC_); CONV_KT128(11, MQ, PC_); CONV_KT128(12, MQ, PC_); CONV_KT128(13, MQ, PC_); CONV_KT128(14, MQ, PC_); CONV_KT128(15, MQ, PC_); CONV_KT128(16, MQ, PC_); CONV_KT128(17, MQ, PC_)
        int cbp = 0;
#if MDPT_HAVE_F8
        if constexpr (F8) {  // the cross-term passes (one loop per pass: the fp16 pass always follows, nothing drains here)
            constexpr bool last = false;
            for (; cbp < ncb8; cbp += 2) { CONV_PAIR128(MQ8A, PC0); }
            if constexpr (NP == 3) { for (; cbp < nvcb8; cbp += 2) { CONV_PAIR128(MQ8B, PC1); } }
        }
#endif
        for (; cbp < nvcb; cbp += 2) {
            const bool last = cbp + 2 >= nvcb;
            CONV_PAIR128(MFMA_Q, PCL);
        }
#undef CONV_PAIR128
#undef CONV_KT128
    } else {
    // ---- prologue: the whole halo patch of channel block 0 (6 instructions per wave), K tile 0 -> even B buffer, K tile 1 -> odd B buffer
#pragma unroll
    for (int j = 0; j < 6; ++j) issue_halo(PC0, j, 0, 0, false);
    issue_b(PC0, 0, 0, 0);
    issue_b(PC0, 1, 0, 0);
    issue_b(PC0, 0, 1, 1);
    issue_b(PC0, 1, 1, 1);
    WAIT_VM_IMM(4);
    BAR();
    if (p.dbg_times) t_first = memtime_now();
    if (grp == 1) BAR();  // stagger: group 1 runs one barrier behind group 0

    // Main loop: one iteration = a PAIR of channel blocks = 18 K tiles, fully unrolled - tap (ky, kx), halo buffer, B buffer, which halo
    // instruction to issue and how many operations may stay in flight are all compile-time constants of the K tile's position U in the pair:
    //     cb = cbp + U / 9 (its patch is in halo buffer U / 9), tap t9 = U % 9 = 3 ky + kx, B buffer U & 1.
    // Four phases per K tile (the 8-phase schedule of gemm8_kernel):
    //   P1: read B0, A rows 0-7                        -> MFMA(0,0)      P2: read B1 | B0 of tile kt+2 (| skip prefetch) -> MFMA(0,1)
    //   P3: read A rows 8-15 | halo DMA (t9 < 6)        -> MFMA(1,1)      P4: B1 of tile kt+2, counted vmcnt                -> MFMA(1,0)
    // vmcnt at P4: the operations issued during this K tile may stay in flight (B0 + B1 of tile kt+2 = 4, + 1 halo instruction, + 1 skip
    // prefetch); everything older (tile kt+1's weights, the halo instructions of earlier K tiles) has landed - one phase and >= one
    // workgroup barrier before its first read. The halo patch of block cb+1 (t9 = 0..5 of block cb) has three more K tiles to land.
#define CONV_KT(U_, MQ, PC_)                                                                                             \
    do {                                                                                                              \
        constexpr int cbl = (U_) / 9, t9 = (U_) % 9, ky = t9 / 3, kx = t9 % 3, bi = (U_) & 1;                          \
        constexpr bool halo_slot = t9 < 6;                                                                            \
        constexpr int inflight = 4 + (halo_slot ? 1 : 0) + (PFSKIP ? 1 : 0);                                          \
        const int kt = cbp * 9 + (U_);                                                                                \
        const bool more = (U_) < 16 || !last;      /* K tile kt + 2 exists */                                         \
        const int va = a_lane[kx] + cbl * HALO_BYTES;                                                                 \
        LOAD_B(fb0, 0, bi); PIN(); LOAD_A(0, va, ky); PIN();                                                          \
        WAIT_LGKM(8); BAR(); WAIT_LGKM(0); PIN();                                                                     \
        MQ(0, 0, fb0); BAR();                                                                                     \
        LOAD_B(fb1, 1, bi); PIN();                                                                                    \
        if (more) { issue_b(PC_, 0, bi, kt + 2); if constexpr (PFSKIP) issue_prefetch(kt - (9 * nvcb - 34)); }                      \
        BAR(); WAIT_LGKM(0); PIN();                                                                                   \
        MQ(0, 1, fb1); BAR();                                                                                     \
        LOAD_A(1, va, ky); PIN();                                                                                     \
        if constexpr (halo_slot) { if (more) issue_halo(PC_, t9, cbp + cbl + 1, cbl ^ 1, cbl == 1 && last); }         \
        BAR(); WAIT_LGKM(0); PIN();                                                                                   \
        MQ(1, 1, fb1); BAR();                                                                                     \
        if (more) { issue_b(PC_, 1, bi, kt + 2); PIN(); WAIT_VM_IMM(inflight); } else { WAIT_VM_IMM(0); }             \
        BAR();                                                                                                        \
        MQ(1, 0, fb0); BAR();                                                                                     \
    } while (0)
#define CONV_PAIR(MQ, PC_) CONV_KT(0, MQ, PC_); CONV_KT(1, MQ, PC_); CONV_KT(2, MQ, PC_); CONV_KT(3, MQ, PC_); CONV_KT(4, MQ, PC_); CONV_KT(5, MQ, PC_); CONV_KT(6, MQ, PC_); CONV_KT(7, MQ, PC_); CONV_KT(8, MQ, PC_); \
        CONV_KT(9, MQ, PC_); CONV_KT(10, MQ, PC_); CONV_KT(11, MQ, PC_); CONV_KT(12, MQ, PC_); CONV_KT(13, MQ, PC_); CONV_KT(14, MQ, PC_); CONV_KT(15, MQ, PC_); CONV_KT(16, MQ, PC_); CONV_KT(17, MQ, PC_)
    int cbp = 0;
#if MDPT_HAVE_F8
    if constexpr (F8) {  // the cross-term passes (one loop per pass: the fp16 pass always follows, nothing drains here)
        constexpr bool last = false;
        for (; cbp < ncb8; cbp += 2) { CONV_PAIR(MQ8A, PC0); }
        if constexpr (NP == 3) { for (; cbp < nvcb8; cbp += 2) { CONV_PAIR(MQ8B, PC1); } }
    }
#endif
    for (; cbp < nvcb; cbp += 2) {
        const bool last = cbp + 2 >= nvcb;  // wave-uniform: the last two K tiles issue nothing and drain
        CONV_PAIR(MFMA_Q, PCL);
    }
#undef CONV_PAIR
#undef CONV_KT
    }
    if (grp == 0) BAR();  // re-join the two groups
#if MDPT_HAVE_F8
    if constexpr (F8) f8_mfma_settle();  // (the compiler does not see the asm MFMAs: f8_cross.h)
#endif
#undef LOAD_A
#undef LOAD_B
#undef LOAD_B_AT
#undef MFMA_Q
#if MDPT_HAVE_F8
#undef MQ8A
#undef MQ8B
#undef MFMA_Q8
#undef MFMA_Q8_ONE
#endif
    if (p.dbg_times) t_loop = memtime_now();

    // ---- epilogue. Lane (l15, lh) of wave (grp, wc) holds, for tile row y = 8 qm + 4 grp + i, pixel x = l15, the four consecutive channels
    //      n = 128 qn + 32 wc + 16 j + 4 lh .. +3 in acc[qm][qn][i][j].
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const size_t plane_px = (size_t)p.H * p.W;
    const __amdgpu_buffer_rsrc_t rs_f = plane_rsrc(F32OUT ? p.out_f32 + (size_t)img * plane_px * COUT : nullptr, F32OUT ? plane_px * COUT * 4 : 0);
    const __amdgpu_buffer_rsrc_t rs_b = plane_rsrc(BFOUT ? p.out_bf + (size_t)img * plane_px * COUT : nullptr, BFOUT ? plane_px * COUT * 2 : 0);
    // lo output plane: the 3-pass form writes one unless the consumer runs a single pass (null pointer: zero-size descriptor, the stores are
    // dropped); the single-pass form of the fp16 build writes one when the consumer runs three passes (mixed-pass mode, wave-uniform test)
    constexpr bool LO_DYN = MDPT_OP_IS_F16 && !X3 && BFOUT && !UPIN;
    const bool want_lo = BFOUT && (X3 || LO_DYN) && p.out_bf_lo != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bl = plane_rsrc(want_lo ? p.out_bf_lo + (size_t)img * plane_px * COUT : nullptr, want_lo ? plane_px * COUT * 2 : 0);
    // ... or, for an F8 consumer (Conv3hParams::out_f8), byte planes: the e5m2 residue plane and (out_a8) the e5m2 plane of the values out_f8 bytes behind it
    const bool lo8 = MDPT_HAVE_F8 && want_lo && p.out_f8 != 0;
    const __amdgpu_buffer_rsrc_t rs_l8 = plane_rsrc(lo8 ? (const unsigned char*)p.out_bf_lo + (size_t)img * plane_px * COUT : nullptr, lo8 ? plane_px * COUT : 0);
    const __amdgpu_buffer_rsrc_t rs_a8 = plane_rsrc(lo8 && p.out_a8 ? (const unsigned char*)p.out_bf_lo + p.out_f8 + (size_t)img * plane_px * COUT : nullptr, lo8 && p.out_a8 ? plane_px * COUT : 0);
    const bool xok = X0 + l15 < p.W;

    // bilinear x2 (align_corners=True) add of the coarser fusion level (fusion_model.py:151,178): stage the <= 10x10 coarse pixels this
    // tile touches (fp32, 1 KiB per pixel, 16-byte chunk c of coarse column cx at slot c ^ (cx & 15)) in the ring, then four reads per value
    int up_x0 = 0, up_x1 = 0, up_py0 = 0, up_pw = 0;
    float up_lx = 0.0f, up_sy = 0.0f;
    if constexpr (UP) {
        up_sy = (float)(p.Hu - 1) / (float)(p.H - 1);
        const float sxs = (float)(p.Wu - 1) / (float)(p.W - 1);
        const int ylast = min(Y0 + 15, p.H - 1), xlast = min(X0 + 15, p.W - 1);
        up_py0 = (int)(up_sy * (float)Y0);
        const int px0 = (int)(sxs * (float)X0);
        const int ph = min((int)(up_sy * (float)ylast) + 1, p.Hu - 1) - up_py0 + 1;
        up_pw = min((int)(sxs * (float)xlast) + 1, p.Wu - 1) - px0 + 1;  // <= UPW (launcher checks the scale)
        const int xq = min(X0 + l15, p.W - 1);
        const float sx = sxs * (float)xq;
        const int x0 = (int)sx, x1 = x0 + (x0 < p.Wu - 1);
        up_lx = sx - (float)x0;
        up_x0 = x0 - px0; up_x1 = x1 - px0;
        __syncthreads();  // every wave is done reading the ring
        const float* src = p.up_src + (size_t)img * p.Hu * p.Wu * 256;
        const int npix = ph * up_pw;
        for (int pp = wave; pp < npix; pp += 8) {
            const int cy = pp / up_pw, cx = pp - cy * up_pw;
            glds16(src + ((size_t)(up_py0 + cy) * p.Wu + (px0 + cx)) * 256 + ((lane ^ (cx & 15)) << 2), smem + pp * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // per-channel constants of both column halves are loaded before the first store (a wait placed after a store would also wait for that
    // store's acknowledgement)
    f32x4 bias_q[NQN][2];
#pragma unroll
    for (int qn = 0; qn < NQN; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bias_q[qn][j] = p.bias ? *(const f32x4*)(p.bias + qn * 128 + wc * 32 + j * 16 + 4 * lh) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const __amdgpu_buffer_rsrc_t rs_s = plane_rsrc(SKIP ? p.skip + (size_t)img * plane_px * 256 : nullptr, SKIP ? plane_px * 1024 : 0);
    // groups g = 2 qm + qn of 8 accumulators (4 tile rows x 2 column blocks); byte offset of (group, i, j) in an fp32 plane
    auto f32_off = [&](int g, int i, int j) -> unsigned {
        const int Y = Y0 + (g >> 1) * 8 + grp * 4 + i;
        return (xok && Y < p.H) ? (unsigned)(Y * p.W + X0 + l15) * (unsigned)(COUT * 4) + (unsigned)((g & 1) * 128 + wc * 32 + j * 16 + 4 * lh) * 4u : OOB;
    };
    auto load_skip = [&](int g, u32x4 (&dst)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_s, f32_off(g, i, j), 0, 0);
    };
    // result of group g, in place, in two steps: base(g) = (conv + bias) [+ x2 bilinear of the coarser level] (LDS only - run for all four
    // groups before the first skip load is issued, so the skip values never compete with the interpolation's registers), then add_skip(g)
    auto base = [&](int g) {
        const int qm = g >> 1, qn = g & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[qm][qn][i][j] += bias_q[qn][j];
        if constexpr (UP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yq = min(Y0 + qm * 8 + grp * 4 + i, p.H - 1);
                const float sy = up_sy * (float)yq;
                const int y0 = (int)sy, y1 = y0 + (y0 < p.Hu - 1);
                const float ly = sy - (float)y0, lx = up_lx;
                const int r0 = (y0 - up_py0) * up_pw, r1 = (y1 - up_py0) * up_pw;
                const int o00 = (r0 + up_x0) * 1024, o01 = (r0 + up_x1) * 1024, o10 = (r1 + up_x0) * 1024, o11 = (r1 + up_x1) * 1024;
                const int k0 = up_x0 & 15, k1 = up_x1 & 15;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ch = qn * 32 + wc * 8 + j * 4 + lh;  // 16-byte chunk of this lane's four channels
                    const f32x4 v00 = *(const f32x4*)(smem + o00 + ((ch ^ k0) << 4)), v01 = *(const f32x4*)(smem + o01 + ((ch ^ k1) << 4));
                    const f32x4 v10 = *(const f32x4*)(smem + o10 + ((ch ^ k0) << 4)), v11 = *(const f32x4*)(smem + o11 + ((ch ^ k1) << 4));
                    acc[qm][qn][i][j] += (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
                }
                // one row's eight LDS reads at a time: the empty asm pins the two results HERE (without it the arithmetic of all 128 reads
                // sinks below the last read and every value read is spilled)
                asm volatile("" : "+v"(acc[qm][qn][i][0]), "+v"(acc[qm][qn][i][1]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto add_skip = [&](int g, const u32x4 (&sk)[4][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[g >> 1][g & 1][i][j] += __builtin_bit_cast(f32x4, sk[i][j]);
    };
    auto store = [&](int g) {
        const int qm = g >> 1, qn = g & 1;
        const unsigned colb = (unsigned)(qn * 128 + wc * 32 + (lh & 1) * 16 + (lh >> 1) * 8) * 2u;  // first of the 8 channels this lane stores as bf16
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + qm * 8 + grp * 4 + i;
            const bool ok = xok && Y < p.H;
            const unsigned pix = (unsigned)(Y * p.W + X0 + l15);
            f32x4 v[2] = {acc[qm][qn][i][0], acc[qm][qn][i][1]};
            if constexpr (F32OUT) {
#pragma unroll
                for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs_f, f32_off(g, i, j), 0, 0);
            }
            if constexpr (RELU) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = fmaxf(v[j][e], 0.0f);
            }
            if constexpr (BFOUT) {
                unsigned hw_[2][2], lw_[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const f32x2 pp = {v[j][2 * w2], v[j][2 * w2 + 1]};
                        const opx2 hh = to_op2(pp);
                        hw_[j][w2] = __builtin_bit_cast(unsigned, hh);
                        if (X3 || (LO_DYN && want_lo)) {
                            const f32x2 rr = pp - __builtin_convertvector(hh, f32x2);
                            lw_[j][w2] = __builtin_bit_cast(unsigned, to_op2(rr));
                        }
                    }
                unsigned ph[4], pl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    auto r = __builtin_amdgcn_permlane16_swap(hw_[0][w2], hw_[1][w2], false, false);
                    ph[w2] = r[0];
                    ph[w2 + 2] = r[1];
                    if (X3 || (LO_DYN && want_lo)) {
                        auto rl = __builtin_amdgcn_permlane16_swap(lw_[0][w2], lw_[1][w2], false, false);
                        pl[w2] = rl[0];
                        pl[w2 + 2] = rl[1];
                    }
                }
                // (no branch around the stores: hipcc would wait for every store's acknowledgement)
                const unsigned boff = ok ? pix * (unsigned)(COUT * 2) + colb : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ph[0], ph[1], ph[2], ph[3]}, rs_b, boff, 0, 0);
#if MDPT_HAVE_F8
                if ((X3 || LO_DYN) && lo8) {  // (wave-uniform) fp8 form: 8 e5m2 bytes of the residue (and of the values) per lane, from the same fp32 values / fp16 roundings
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2v;
                    unsigned l8[2], a8[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x2 h0 = __builtin_convertvector(__builtin_bit_cast(opx2, hw_[j][0]), f32x2), h1 = __builtin_convertvector(__builtin_bit_cast(opx2, hw_[j][1]), f32x2);
                        l8[j] = f8_lo8x4(v[j][0], v[j][1], v[j][2], v[j][3], h0[0], h0[1], h1[0], h1[1]);
                        a8[j] = f8_a8x4(h0[0], h0[1], h1[0], h1[1]);
                    }
                    auto rl8 = __builtin_amdgcn_permlane16_swap(l8[0], l8[1], false, false);
                    const unsigned boff8 = ok ? pix * (unsigned)COUT + (colb >> 1) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2v{rl8[0], rl8[1]}, rs_l8, boff8, 0, 0);
                    auto ra8 = __builtin_amdgcn_permlane16_swap(a8[0], a8[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2v{ra8[0], ra8[1]}, rs_a8, boff8, 0, 0);  // (zero records without an a8 plane: dropped)
                } else
#endif
                // (unconditional where the form can have a lo plane at all: without one the descriptor has zero records and the store is dropped)
                if constexpr (X3 || LO_DYN) __builtin_amdgcn_raw_buffer_store_b128(u32x4{pl[0], pl[1], pl[2], pl[3]}, rs_bl, boff, 0, 0);
            }
        }
    };
    if constexpr (SKIP) {
        // vmcnt retires in order: a wait for loads issued after a store also waits for that store's acknowledgement. So group g+1's loads
        // are issued BEFORE group g's stores (one group = 8 x 16 bytes per lane in flight; the results are kept in the accumulators)
        u32x4 sk[4][2];
        load_skip(0, sk);
#pragma unroll
        for (int g = 0; g < 4; ++g) base(g);
        __builtin_amdgcn_sched_barrier(0);
        add_skip(0, sk);
        load_skip(1, sk);
        store(0);
        add_skip(1, sk);
        load_skip(2, sk);
        store(1);
        add_skip(2, sk);
        load_skip(3, sk);
        store(2);
        add_skip(3, sk);
        store(3);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (NQN == 2 || (g & 1) == 0) { base(g); store(g); }
    }
    if (p.dbg_times && tid == 0) {
        const unsigned long long t_issued = memtime_now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = p.dbg_times + (size_t)blockIdx.x * 6;
        d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = memtime_now();
        d[4] = t_issued;
        d[5] = __builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 3 << 11);  // HW_REG_XCC_ID
    }
}
#undef PIN
#undef BAR
#undef WAIT_LGKM
#undef WAIT_VM_IMM
#undef PC0
#undef PC1
#undef PCL

template <int NQN, int NP, bool SKIP, bool F32OUT, bool RELU, bool UP, bool BFOUT = true, bool UPIN = false, bool F8 = false>
int launch_variant(const Conv3hParams& p, hipStream_t stream) {
    auto kern = conv3h_kernel<NQN, NP, SKIP, F32OUT, RELU, UP, BFOUT, UPIN, F8>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = p.B * ((p.H + 15) / 16) * ((p.W + 15) / 16);
    static char prof_name[80] = "";
    if (!prof_name[0])
        snprintf(prof_name, sizeof(prof_name), "conv3h_kernel<%d, %s, %d, %d, %d, %d>", 128 * NQN, UPIN ? "bf16 up2-in" : (F8 ? (NP == 3 ? "x3f8" : "x2f8") : (NP == 3 ? "x3" : (NP == 2 ? "x2a" : "bf16"))), (int)SKIP, (int)F32OUT, (int)RELU,
                 (int)UP);
    MdptProfScope prof(prof_name, 2.0 * p.B * p.H * p.W * (128.0 * NQN) * 9.0 * p.Cin, stream);  // algorithmic flops (one pass, whatever the mode)
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), LDS_BYTES, stream, p);
    return (int)hipGetLastError();
}

// the epilogue combinations the decoder uses
template <int NP, bool F8 = false>
int launch_mode(const Conv3hParams& p, hipStream_t stream) {
    constexpr int X3 = NP;  // (pass count of every variant below)
    const bool skip = p.skip != nullptr, f32 = p.out_f32 != nullptr, up = p.up_src != nullptr;
    if (p.Cout == 128) {
        if constexpr (NP == 1) {
            if (p.up_in) return launch_variant<1, 1, false, false, false, false, true, true>(p, stream);
        }
        if (p.out_bf) return launch_variant<1, X3, false, false, false, false, true, false, F8>(p, stream);
        return launch_variant<1, X3, false, true, false, false, false, false, F8>(p, stream);
    }
    if (up) return launch_variant<2, X3, true, true, true, true, true, false, F8>(p, stream);
    if (!skip && !f32) return launch_variant<2, X3, false, false, true, false, true, false, F8>(p, stream);
    if (skip && !f32) return launch_variant<2, X3, true, false, false, false, true, false, F8>(p, stream);
    return launch_variant<2, X3, false, true, true, false, true, false, F8>(p, stream);
}

}  // namespace

// the combinations the decoder uses (anything else runs the implicit-GEMM path of gemm.hip)
bool MDPT_FN(mdpt_conv3h_supported)(const Conv3hParams& p) {
    if (p.B <= 0 || p.H < 2 || p.W < 2 || p.Cin <= 0 || (p.Cin & 127) || !p.w) return false;
    if ((p.in != nullptr) == (p.up_in != nullptr)) return false;  // exactly one input form
    if (p.up_in) {
        // upsampled input: 128 output channels, bf16 planes out, bias only; 18 halo pixels must interpolate from <= 11 source pixels
        if (p.Cout != 128 || p.in_lo || !p.out_bf || p.out_f32 || p.skip || p.up_src || p.relu_bf || p.Hs < 2 || p.Ws < 2) return false;
        if ((long)17 * (p.Hs - 1) >= (long)9 * (p.H - 1) || (long)17 * (p.Ws - 1) >= (long)9 * (p.W - 1)) return false;
        if ((size_t)p.Hs * p.Ws * p.Cin * 2 >= 0xFFFFFFF0ull) return false;
    }
    if (p.Cout != 256 && p.Cout != 128) return false;
    if ((size_t)p.H * p.W * p.Cout * 4 >= 0xFFFFFFF0ull || (size_t)p.H * p.W * p.Cin * 2 >= 0xFFFFFFF0ull) return false;  // 32-bit byte offsets inside one image plane
    if (p.f8) {  // fp8 cross terms: fp16 build, pairs of 128-channel blocks, byte planes within 32-bit offsets (implied by the fp16 plane's check below)
        if (!MDPT_OP_IS_F16 || !p.in_lo || p.up_in || (p.Cin & 255) || !p.w8 || !p.s8 || p.w_lo || ((p.w8_lo != nullptr) != (p.s8_lo != nullptr)) || (p.w8_lo && !p.a8_off)) return false;
    }
    if (p.out_f8 && (!MDPT_OP_IS_F16 || !p.out_bf_lo)) return false;
    const bool x3 = p.in_lo != nullptr;  // multi-pass: three passes with a lo plane of the weights, two (activation-split) without
    // (a multi-pass conv whose consumer runs one pass writes no lo plane: out_bf_lo may be null)
    if (!x3 && (p.w_lo || (p.out_bf_lo && (!MDPT_OP_IS_F16 || p.up_in || !p.out_bf)))) return false;  // single pass + lo output: fp16 build only
    const bool skip = p.skip != nullptr, f32 = p.out_f32 != nullptr, relu = p.relu_bf != 0, up = p.up_src != nullptr;
    if (p.Cout == 128) return !skip && !relu && !up && ((p.out_bf != nullptr) != f32);  // bias -> bf16 planes, or bias -> fp32 map
    if (!p.out_bf) return false;
    if (up) {
        // the 16 fine rows / columns of a tile must interpolate from <= UPW coarse ones: floor(15 * scale) + 3 <= UPW
        if (p.Hu < 1 || p.Wu < 1 || (long)15 * (p.Hu - 1) >= (long)(UPW - 2) * (p.H - 1) || (long)15 * (p.Wu - 1) >= (long)(UPW - 2) * (p.W - 1)) return false;
        return skip && f32 && relu;
    }
    if (!skip && !f32 && relu) return true;
    if (skip && !f32 && !relu) return true;
    if (!skip && f32 && relu) return true;
    return false;
}

int MDPT_FN(mdpt_launch_conv3h)(const Conv3hParams& p, hipStream_t stream) {
    if (!MDPT_FN(mdpt_conv3h_supported)(p)) return (int)hipErrorInvalidValue;
#if MDPT_OP_IS_F16
    if (p.f8) return p.w8_lo ? launch_mode<3, true>(p, stream) : launch_mode<2, true>(p, stream);
#endif
    return p.in_lo ? (p.w_lo ? launch_mode<3>(p, stream) : launch_mode<2>(p, stream)) : launch_mode<1>(p, stream);
}
