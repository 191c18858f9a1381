// fp32 implicit-GEMM convolution on the CDNA4 matrix cores, two arithmetic routes in one kernel template:
//
//   SPLIT = false  v_mfma_f32_32x32x2_f32: exact f32 fmaf-chain numerics at 64 FLOP/clk/SIMD (157 TFLOP/s chip peak).
//   SPLIT = true   (round 3, the networks' default) every fp32 operand as the exact sum of three bf16 terms, six of the nine cross
//                  products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32 accuracy (the dropped terms are <= 2^-23 of a
//                  product) at 2.67x the fp32 pipe's rate per clock.  See the comment at the split main loop, DESIGN.md 3.1b.
//
// Mapping (one 256-thread workgroup = 4 waves, one per SIMD, 2x2 over the block tile):
//   GEMM M = B*Ho*Wo output pixels (A rows, gathered NHWC pixels: 32 channels = one 128-B line)
//   GEMM N = Cout                  (B rows, packed weights [Cout][KH*KW*Cin])
//   GEMM K = KH*KW*Cin walked in 32-float chunks; Cin % 32 == 0 so a chunk never straddles a tap.
// LDS image: [row][32 floats], the 16-B chunk index XOR-swizzled with (row>>1)&7 so that the
// ds_read_b128 fragment reads (16-lane groups, 64-bank rows) are conflict-free.
// MFMA operand trick: lane (i = l&31, h = l>>5) reads 4 consecutive k (one ds_read_b128) and
// feeds element e to the e-th MFMA; A and B use the same k for the same (h, e), and the MFMA
// sums over k, so any k permutation is legal.
// D layout (32x32): col(n) = lane&31, row(m) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
//   -> each store instruction writes two 128-B channel runs: coalesced NHWC epilogue.
// History (measured on MI355X, profiles/r01_conv_*.txt): the bring-up kernel staged tiles through VGPRs with two barriers
// per chunk (120 TFLOP/s on the decoder conv); double-buffered LDS + register-pinned fragment prefetch alone: null;
// global_load_lds with a zero page: 128; buffer_load...lds with range-check padding: 138.  Tried and measured null,
// removed: start-time stagger of co-resident blocks; s_setprio around the MFMAs; an LDS-transposed epilogue with 16-B
// stores; persistent tiles with the next tile's first DMA issued before the epilogue; a 256x128 eight-wave tile (kept as
// tile 5, not selected); raster panels sized to the XCD's co-resident tiles; skipping the chunks of taps that are outside
// the image for a whole tile (ASPP dilation 24 / 36 on a 90x90 map: up to 27 % of the chunks) -- every tile is resident at
// once, so the launch lasts as long as its full-price centre tiles; with scattered m-tiles and 64x64 tiles to mix cheap and
// expensive ones per CU: +6 % on the dilation-24 conv, 0 on DeepLabv3 end to end.
#include "kernels.h"
#include "igemm_epilogue.h"

#include <algorithm>
#include <cstdlib>

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------
// conv_igemm_dma_f32: the tiles go global -> LDS DIRECTLY
// (buffer_load_dwordx4 ... offen lds: no VGPR staging, no ds_write pass, no vmcnt ladder), two LDS stages, one
// barrier per 32-deep chunk, fragment reads double-buffered in registers.  One wave-instruction of the DMA writes
// a lane-linear 1 KiB (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE offset.
// Padding is done by the buffer descriptor's RANGE CHECK: an out-of-range lane makes the DMA write zeros into LDS
// (probed on MI355X, tools/probe_buffer_lds.hip), so a padding tap / a row beyond M / a channel beyond Cout is just
// the sentinel voffset 0x80000000.  Per chunk the issue is 8 DMAs + ~12 VALU: the per-lane offsets are loop
// invariants, the chunk position is the scalar soffset, tap validity is one bit of a per-row mask.
// Elimination runs on the decoder conv (profiles/): VGPR-staged loads cost ~9 %, the ds_write pass ~5 %.
// Workgroup timelines (tools/probe_conv_trace.hip, profiles/r01_conv_wg_timeline.txt): with two workgroups per CU the
// main loop keeps the MFMA pipe busy 93 % of its cycles and only 1.6 % of them are spent in the per-chunk wait+barrier;
// a lone workgroup reaches 83 %.  What short-K layers lose is outside the loop, because the co-resident workgroup is
// in the same phase (both were dispatched together): prologue (index math + first DMA) and epilogue.  The first
// epilogue (per-element guards, 64-bit addresses, run-time activation switch: ~85 instructions per element) took
// 11-17 k cycles per 128x128 tile and the prologue 6 k -- 27 % of a K=256 tile; with the buffer-descriptor epilogue
// below (5 instructions per element) and the division-free 1x1 prologue they are 3-8 k and 4 k (+6 % end to end).
// De-phasing the two workgroups of a CU (delaying bid+256 by 4-40 k cycles) was measured before and after that change:
// null to negative every time (the late workgroup's prologue/epilogue slow down under the neighbour's DMA + MFMA stream).
// ---------------------------------------------------------------------------------------------------------
#ifdef FS_TRACE
// tools/probe_conv_trace.hip only (never in libfloodseg.so): per-workgroup timeline, 8 x u64 per workgroup:
// [0] start, [1] first stage landed, [2] main loop done, [3] epilogue done (shader clock, s_memtime),
// [4] cycles spent in the per-chunk wait+barrier, [5] HW_ID | XCC_ID << 32, [6] start (100 MHz wall clock), [7] end (wall)
__device__ unsigned long long fs_trace_buf[8 * 65536];
// round 5: end-of-chunk stamps of the first 8 K chunks of the split main loop (scalar registers; written out by lane 0 at the end)
__device__ unsigned long long fs_trace_chunks[8 * 65536];
#define FS_TRACE_DECL unsigned long long tr_start = __builtin_readcyclecounter(), tr_wall = wall_clock64(), tr_ready = 0, tr_loop = 0, tr_wait = 0; \
    unsigned long long tr_c0 = 0, tr_c1 = 0, tr_c2 = 0, tr_c3 = 0, tr_c4 = 0, tr_c5 = 0, tr_c6 = 0, tr_c7 = 0;
#define FS_TRACE_CHUNK(KC_) { const unsigned long long tc_ = __builtin_readcyclecounter(); \
    if ((KC_) == 0) tr_c0 = tc_; else if ((KC_) == 1) tr_c1 = tc_; else if ((KC_) == 2) tr_c2 = tc_; else if ((KC_) == 3) tr_c3 = tc_; \
    else if ((KC_) == 4) tr_c4 = tc_; else if ((KC_) == 5) tr_c5 = tc_; else if ((KC_) == 6) tr_c6 = tc_; else if ((KC_) == 7) tr_c7 = tc_; }
#define FS_TRACE_SYNC() { const unsigned long long tw = __builtin_readcyclecounter(); FS_DMA_PUBLISH() tr_wait += __builtin_readcyclecounter() - tw; }
#else
#define FS_TRACE_DECL
#define FS_TRACE_CHUNK(KC_)
#define FS_TRACE_SYNC() FS_DMA_PUBLISH()
#endif
// A wave's buffer_load...lds writes are complete when ITS vmcnt reaches 0; the other waves may read them only after that.
// __syncthreads() does not imply it: the compiler tracks LDS-DMA only against the SAME wave's later ds_reads and is free
// to wait for vmcnt after the barrier (it did, in the 128x64 instantiation, once two unrelated global loads were added
// ahead of the loop -- a race between a wave's DMA and its neighbours' fragment reads).  So the wait is explicit:
// s_waitcnt vmcnt(0) lgkmcnt(0) (this wave's fragment reads of the stage that is about to be overwritten are done as
// well; expcnt left at "no wait"), then the barrier.
#define FS_DMA_PUBLISH() { __builtin_amdgcn_s_waitcnt(0x0070); __syncthreads(); }
// WGM x WGN waves per workgroup (2x2 = 256 threads, two workgroups per CU; 4x2 = 512 threads, one per CU).
// DUAL = true: concatenated-K GEMM of two 1x1 convs (ConvParams::in2): the K chunks beyond the first conv's come from a second
// map with its own pixel stride / conv stride.  A separate instantiation (no residual input: the shortcut IS the second
// operand), so the plain kernel's register budget -- 252 of the 256 VGPRs that let two workgroups share a CU -- is untouched.
// LDS floats of one workgroup: two stages of (pixel tile + filter tile)
// (split filters are staged in whole 16-row groups per wave: a 96-column tile stages 128 rows, the last 32 of them zeros)
constexpr int conv_split_rows(int BN, int waves) { return (BN / 16 + waves - 1) / waves * waves * 16; }
template <int BM, int BN, bool SPLIT, int WAVES = 4>
constexpr int conv_tile_lds_floats() { return 2 * (BM * 32 + (SPLIT ? 3 * conv_split_rows(BN, WAVES) * 16 : BN * 32)); }

// One BM x BN output tile at (m0, n0): prologue, main loop, epilogue.  `lds` = conv_tile_lds_floats() floats, 1 KiB aligned; every
// wave of the workgroup calls it with the same arguments.
template <int BM, int BN, int WGM, int WGN, bool DUAL, bool SPLIT>
__device__ __forceinline__ void conv_tile(ConvParams p, const int m0, const int n0, float* __restrict__ lds, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins do not exist in the host pass, which only needs the launch stub
    constexpr int BK = 32;
    constexpr int NT = 64 * WGM * WGN;   // threads
    constexpr int RSTEP = NT / 8;        // tile rows staged per DMA pass (8 lanes per 128-B row)
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    // SPLIT: a filter row of a chunk is 3 planes x 32 bf16 = 3 x 64 B; one DMA wave-instruction covers 16 rows of one plane
    constexpr int BNL = SPLIT ? conv_split_rows(BN, NT / 64) : BN;  // filter rows staged (>= BN: every wave stages whole 16-row groups)
    constexpr int BPL = BNL * 16;  // floats of LDS per filter plane and stage
    constexpr int RA = BM / RSTEP, RB = SPLIT ? 3 * (BNL / 16) / (NT / 64) : BN / RSTEP;
    constexpr int RB1 = SPLIT ? (BNL / 16) / (NT / 64) : RB;  // of them per plane
    static_assert(BN % 32 == 0 && WN % 32 == 0 && (SPLIT || BN % RSTEP == 0), "tile columns: whole 32-column MFMA blocks per wave");
    constexpr int STAGE = BM * BK + (SPLIT ? 3 * BPL : BN * BK);
    FS_TRACE_DECL

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hh = lane >> 5;
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    const int M = p.B * p.Ho * p.Wo;
    const int K = p.KH * p.KW * p.Cin + (DUAL ? p.Cin2 : 0);
    const int cpt = p.Cin >> 5;
    const int nchunks1 = p.KH * p.KW * cpt;
    const int nchunks = nchunks1 + (DUAL ? p.Cin2 >> 5 : 0);

    const int sc = t & 7, r0 = t >> 3;
    const int swz = sc ^ ((r0 >> 1) & 7);  // logical chunk this lane fetches into LDS slot sc (same key for rows r0+32j)
    constexpr unsigned SENT = 0x80000000u;  // >= num_records of both descriptors (tensors < 2 GiB, checked by the launcher)
    // A descriptor starts `pad` rows and `pad` pixels BEFORE the tensor so that every tap offset (soffset) is >= 0;
    // lanes that are valid by the tap mask always land inside the tensor.
    const long long pad_off = ((long long)p.pad * p.W + p.pad) * p.ld_in;
    const unsigned a_bytes = (unsigned)(((long long)p.B * p.H * p.W * p.ld_in + pad_off) * 4);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in - pad_off), 0, a_bytes, 0x00020000);
    const int ldw = p.ld_wgt ? p.ld_wgt : K;  // filter row stride (> K for a K-slice of wider rows)
    const __amdgpu_buffer_rsrc_t b_rsrc =
        SPLIT ? __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt3, 0, 2u * p.plane_bytes + (unsigned)((((long long)p.Cout - 1) * ldw + K) * 2), 0x00020000)
              : __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (unsigned)((((long long)p.Cout - 1) * ldw + K) * 4), 0x00020000);
    unsigned a_voff[RA], a_mask[RA];
    if (p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0) {
        // 1x1 stride-1 conv / plain GEMM (most launches): output pixel m IS input pixel m, no coordinate split needed
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = m0 + r0 + RSTEP * j;
            a_voff[j] = (unsigned)((m * p.ld_in + swz * 4) * 4);  // rows >= M: the tap mask keeps them out of range
            a_mask[j] = m < M ? 1u : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = m0 + r0 + RSTEP * j;
            const bool v = m < M;
            const int mm = v ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_voff[j] = (unsigned)(((b * p.H * p.W + oy * p.stride * p.W + ox * p.stride) * p.ld_in + swz * 4) * 4);
            // tap (r, q) is inside the image iff its row is and its column is: KH + KW tests instead of KH * KW
            unsigned colmask = 0, mask = 0;
            for (int q2 = 0; q2 < p.KW; ++q2)
                if ((unsigned)(ox * p.stride - p.pad + q2 * p.dil) < (unsigned)p.W) colmask |= 1u << q2;
            for (int r = 0; r < p.KH; ++r)
                if ((unsigned)(oy * p.stride - p.pad + r * p.dil) < (unsigned)p.H) mask |= colmask << (r * p.KW);
            a_mask[j] = v ? mask : 0u;
        }
    }
    // DUAL: the second operand's descriptor and per-row offsets (pixel (b, oy*stride2, ox*stride2) of the H2 x W2 map)
    const __amdgpu_buffer_rsrc_t a2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.in2 : p.in), 0, DUAL ? (unsigned)((long long)p.B * p.H2 * p.W2 * p.ld_in2 * 4) : 0u, 0x00020000);
    unsigned a2_voff[DUAL ? RA : 1];
    if (DUAL) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = m0 + r0 + RSTEP * j;
            const int mm = m < M ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a2_voff[j] = (unsigned)((((b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.ld_in2 + swz * 4) * 4);
        }
    }
    unsigned b_voff[RB1];
    if (SPLIT) {
        // lane -> (row, 16-B slot) of a 16-row x 64-B group; the slot holds piece slot ^ ((row >> 2) & 3) of the row's 32 bf16, which
        // makes the fragment reads (16 consecutive rows, one piece) hit 16 different 16-B columns of the 256-B bank row
#pragma unroll
        for (int j = 0; j < RB1; ++j) {
            const int row = 16 * (wv + (NT / 64) * j) + (lane >> 2);
            const int n = n0 + row;
            b_voff[j] = (n < p.Cout && row < BN) ? (unsigned)(n * ldw * 2 + (((lane & 3) ^ ((row >> 2) & 3)) * 16)) : SENT;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RB1; ++j) {
            const int n = n0 + r0 + RSTEP * j;
            b_voff[j] = n < p.Cout ? (unsigned)((n * ldw + swz * 4) * 4) : SENT;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int sw = (l31 >> 1) & 7;

    int tap_r = 0, tap_s = 0, cc = 0;  // chunk being fetched
    int fetch = 0;                     // its index along K (DUAL: chunks >= nchunks1 belong to the second operand)
    unsigned b_soff = 0;               // its byte offset along k in the packed filters

    // One DMA row (A rows first, then B rows): ROW_ in [0, RA+RB).
#ifdef FS_TRACE  // elimination experiments (results invalid, timing only; profiles/r05_experiments.txt section 17): dbg & 64 no pixel-row DMAs,
                 // & 128 (below) no pixel-fragment LDS reads, & 256 no filter-row DMAs
#define FS_DMA_SKIP(ROW_) (((ROW_) < RA && (p.dbg & 64)) || ((ROW_) >= RA && (p.dbg & 256)))
#else
#define FS_DMA_SKIP(ROW_) false
#endif
#define FS_DMA_ROW(STG, ROW_)                                                                                     \
    if (FS_DMA_SKIP(ROW_)) {                                                                                      \
    } else if ((ROW_) < RA) {                                                                                     \
        const int j = (ROW_) < RA ? (ROW_) : 0;                                                                   \
        if (DUAL && second) {                                                                                     \
            const unsigned vo = (a_mask[j] & 1u) ? a2_voff[DUAL ? j : 0] : SENT;                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a2_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + (8 * wv_u + RSTEP * j) * BK), \
                                                     16, vo, a_soff, 0, 0);                                       \
        } else {                                                                                                  \
            const unsigned vo = ((a_mask[j] >> tap_bit) & 1u) ? a_voff[j] : SENT;                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + (8 * wv_u + RSTEP * j) * BK), \
                                                     16, vo, a_soff, 0, 0);                                       \
        }                                                                                                         \
    } else if (SPLIT) {                                                                                           \
        const int jj = (ROW_) >= RA ? (ROW_) - RA : 0;                                                            \
        const int pl = jj / RB1, j = jj % RB1;                                                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + BM * BK + pl * BPL + (wv_u + (NT / 64) * j) * 256), \
                                                 16, b_voff[j], b_soff + (unsigned)pl * p.plane_bytes, 0, 0);     \
    } else {                                                                                                      \
        const int j = (ROW_) >= RA ? (ROW_) - RA : 0;                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + BM * BK + (8 * wv_u + RSTEP * j) * BK), \
                                                 16, b_voff[j], b_soff, 0, 0);                                    \
    }
#define FS_DMA_ADVANCE()                                                                                          \
    {                                                                                                             \
        ++fetch;                                                                                                  \
        if (p.korder == 0) {                                                                                      \
            if (++cc == cpt) { cc = 0; if (++tap_s == p.KW) { tap_s = 0; ++tap_r; } }                             \
        } else {                                                                                                  \
            if (++tap_s == p.KW) { tap_s = 0; if (++tap_r == p.KH) { tap_r = 0; ++cc; } }                         \
        }                                                                                                         \
    }
#define FS_DMA_ALL(STG)                                                                                           \
    {                                                                                                             \
        const bool second = DUAL && fetch >= nchunks1;                                                            \
        const int tap_bit = tap_r * p.KW + tap_s;                                                                 \
        const unsigned a_soff = second ? (unsigned)((fetch - nchunks1) * 128)                                     \
                                       : (unsigned)((((tap_r * p.W + tap_s) * p.dil) * p.ld_in + cc * 32) * 4);  \
        _Pragma("unroll") for (int rw = 0; rw < RA + RB; ++rw) { FS_DMA_ROW(STG, rw) }                            \
        b_soff += SPLIT ? 64 : 128;                                                                               \
    }

#define FS_FRAGS(STG, S_, A_, B_)                                                                                 \
    {                                                                                                             \
        const float* a_src = lds + (STG) * STAGE;                                                                 \
        const float* b_src = a_src + BM * BK;                                                                     \
        const int cidx = (2 * (S_) + hh) ^ sw;                                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                            \
            A_[i] = *reinterpret_cast<const f32x4*>(&a_src[(wm * WM + i * 32 + l31) * BK + 4 * cidx]);            \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
            B_[j] = *reinterpret_cast<const f32x4*>(&b_src[(wn * WN + j * 32 + l31) * BK + 4 * cidx]);            \
    }

#define FS_MMA(A_, B_)                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[i][e], B_[j][e], acc[i][j], 0, 0, 0);

    // (Round 5 measured the alternative -- both stages requested before the first wait, a counted vmcnt in between: the first stage
    //  lands 700 cycles later, the first chunk ends 800 cycles earlier, nothing end to end: profiles/r05_experiments.txt section 12.)
    FS_DMA_ALL(0)
    FS_DMA_ADVANCE()
    FS_DMA_PUBLISH()  // stage 0 has landed for every wave
#ifdef FS_TRACE
    tr_ready = __builtin_readcyclecounter();
#endif
    // Software pipeline across the chunk boundary: the barrier that hands stage cur^1 over (and frees stage cur) sits
    // between sub-steps 2 and 3; the DMA of chunk kc+2 and the first fragment reads of chunk kc+1 are issued right
    // behind it and hide under the 16 MFMAs of sub-step 3, so the MFMA pipe does not drain at a chunk boundary.
    int cur = 0;
    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
    float rv[TM][TN][16];  // residual tile, requested where the last DMA would have been (see below)
    const int em_base = m0 + wm * WM + 4 * hh, en_base = n0 + wn * WN + l31;
    float sc_n[TN], sh_n[TN];  // per-channel scale / shift of this lane's columns: fetched now, needed after the loop
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = en_base + j * 32;
        sc_n[j] = (n < p.Cout && p.scale) ? p.scale[n] : 1.f;
        sh_n[j] = (n < p.Cout && p.shift) ? p.shift[n] : 0.f;
    }
    if (nchunks > 1) {
        FS_DMA_ALL(1)
        FS_DMA_ADVANCE()
    }
    if constexpr (SPLIT) {
        // Split operands: every fp32 value x is the exact sum h + m + l of three bf16 terms (h = bf16(x), m = bf16(x - h),
        // l = bf16(x - h - m), round to nearest even; 3 x 8 significant bits with signed residues cover the 24 of fp32).  The filters
        // arrive split (three planes); the pixels are split here, in registers, right after the fragment read.  Of the nine cross
        // products the six of order <= 2^-16 go to the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: exact products, fp32 accumulate):
        //   x * w = hh + (hm + mh) + (mm + hl + lh) + [ml + lm + ll <= 2^-23 |x w|, dropped: below the rounding of one fp32 add]
        // A chunk of 32 k = two MFMA steps of 16 k; lane (i, hh) owns k = 16 s + 8 hh .. + 7 of row i in both operands.
        bf16x8 A3[TM][3], B3[TN][3], A3n[TM][3], B3n[TN][3];
        f32x4 araw[TM][2];
#ifdef FS_TRACE
        const bool FS_NO_A_READ = (p.dbg & 128) != 0;  // elimination experiment: the pixel fragments are not read from LDS
        f32x4 fake_a = {1.f + lane, 2.f, 3.f, 4.f};
        asm volatile("" : "+v"(fake_a));
#define FS_FAKE_A_OPAQUE() asm volatile("" : "+v"(fake_a));
#else
        constexpr bool FS_NO_A_READ = false;
        const f32x4 fake_a = {0.f, 0.f, 0.f, 0.f};
#define FS_FAKE_A_OPAQUE()
#endif
        u32x4 ah[TM][3];
#define FS_READ3(STG, S_, RAW_, B_)                                                                               \
    {                                                                                                             \
        const float* a_src = lds + (STG) * STAGE;                                                                 \
        const float* b_src = a_src + BM * BK;                                                                     \
        const int c0 = (4 * (S_) + 2 * hh) ^ sw;                                                                  \
        if (FS_NO_A_READ) {                                                                                       \
            FS_FAKE_A_OPAQUE()  /* per use: the split of the fake fragments is not to be hoisted out of the loop */ \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) { RAW_[i][0] = fake_a; RAW_[i][1] = fake_a; }          \
        } else {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                          \
            RAW_[i][0] = *reinterpret_cast<const f32x4*>(&a_src[(wm * WM + i * 32 + l31) * BK + 4 * c0]);         \
            RAW_[i][1] = *reinterpret_cast<const f32x4*>(&a_src[(wm * WM + i * 32 + l31) * BK + 4 * (c0 ^ 1)]);   \
        }                                                                                                         \
        }                                                                                                         \
        const int bslot = (2 * (S_) + hh) ^ ((l31 >> 2) & 3);                                                     \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                      \
                B_[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&b_src[pl * BPL + (wn * WN + j * 32 + l31) * 16 + 4 * bslot])); \
    }
        // One MFMA step: 6 TM TN MFMAs (32 cycles each on the SIMD's matrix pipe, 8 of them holding the issue port) with the split
        // of the NEXT step's pixels threaded through them in program order -- every slot is fenced (sched_barrier), because the
        // compiler otherwise issues the MFMAs back to back and the ~36 TM VALU instructions of the split after them.  Units of the
        // split: level 0 of every pair (cvt_pk, two unpacks, subtract), then level 1 of every pair, one unit per slot from slot S0 on
        // (the fragment reads land under the first slots); a pair's final cvt_pk rides in the slot after its level 1.
        constexpr int NS = 6 * TM * TN, NP = 4 * TM, NU = 3 * NP, S0 = NS >= 24 ? 4 : NS / 6;
        float rq[NP][2];
#define FS_STEP3(A_, B_, RAW_, AN_)                                                                               \
    _Pragma("unroll") for (int k = 0; k < NS; ++k) {                                                              \
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                     \
        const int term = k / (TM * TN), i_ = (k % (TM * TN)) / TN, j_ = k % TN;                                   \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i_][PA[term]], B_[j_][PB[term]], acc[i_][j_], 0, 0, 0); \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                          \
            const int lvl = u / NP, pr = u % NP, pi = pr / 4, pe = pr % 4;                                        \
            const int v_ = lvl < 2 ? u : NP + pr;  /* level unit (of the pair's level 1 for a final) */            \
            const int sl = S0 + (2 * NP <= NS - S0 ? v_ : v_ * (NS - S0) / (2 * NP));                             \
            if ((lvl < 2 ? sl : (sl + 1 < NS ? sl + 1 : NS - 1)) != k) continue;                                  \
            const float x0 = lvl == 0 ? RAW_[pi][pe >> 1][2 * (pe & 1)] : rq[pr][0];                              \
            const float x1 = lvl == 0 ? RAW_[pi][pe >> 1][2 * (pe & 1) + 1] : rq[pr][1];                          \
            unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));         \
            asm volatile("" : "+v"(pk));  /* anchors the unit in THIS slot: the optimiser would sink it to its use */ \
            ah[pi][lvl][pe] = pk;                                                                                 \
            if (lvl < 2) {                                                                                        \
                rq[pr][0] = x0 - __builtin_bit_cast(float, pk << 16);                                             \
                rq[pr][1] = x1 - __builtin_bit_cast(float, pk & 0xffff0000u);                                     \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) AN_[i][pl] = __builtin_bit_cast(bf16x8, ah[i][pl]);
        FS_READ3(0, 0, araw, B3)
        {   // prologue: split step 0's pixels (no MFMA to hide under yet)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int lvl = u / NP, pr = u % NP, pi = pr / 4, pe = pr % 4;
                const float x0 = lvl == 0 ? araw[pi][pe >> 1][2 * (pe & 1)] : rq[pr][0];
                const float x1 = lvl == 0 ? araw[pi][pe >> 1][2 * (pe & 1) + 1] : rq[pr][1];
                const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
                ah[pi][lvl][pe] = pk;
                if (lvl < 2) {
                    rq[pr][0] = x0 - __builtin_bit_cast(float, pk << 16);
                    rq[pr][1] = x1 - __builtin_bit_cast(float, pk & 0xffff0000u);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) A3[i][pl] = __builtin_bit_cast(bf16x8, ah[i][pl]);
        }
        unsigned touch0 = 0, touch1 = 0;
        int touch_kc = (!DUAL && p.res && p.res_touch) ? max(nchunks - 2, 0) : -1;  // the chunk after whose barrier the residual lines are touched
        asm volatile("" : "+s"(touch_kc));  // opaque: the loop is not to be versioned on it
        for (int kc = 0; kc < nchunks; ++kc) {
            FS_READ3(cur, 1, araw, B3n)
            __builtin_amdgcn_sched_barrier(0);
            FS_STEP3(A3, B3, araw, A3n)
            __builtin_amdgcn_sched_barrier(0);
            FS_TRACE_SYNC()  // every wave's reads of stage `cur` are done, chunk kc + 1 has landed
            if (kc + 2 < nchunks) {
                FS_DMA_ALL(cur)
                FS_DMA_ADVANCE()
            }
            if (kc == touch_kc) {
                // the residual tile is BM rows x LPR = BN * 4 / 128 lines; thread t touches lines t and t + NT of the tile's BM * LPR (by
                // line, not by row pair: on the 128 x 96 tile a row has three lines -- ADVICE r5)
                const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (unsigned)((long long)M * p.ld_res * 4), 0x00020000);
                constexpr int LPR = BN / 32;  // lines per row
                const int l0 = t, l1 = t + NT;
                const int r0t = l0 / LPR, c0t = l0 % LPR, r1t = l1 / LPR, c1t = l1 % LPR;
                const bool ok0 = l0 < BM * LPR && n0 + c0t * 32 < p.Cout, ok1 = l1 < BM * LPR && n0 + c1t * 32 < p.Cout;
                touch0 = __builtin_amdgcn_raw_buffer_load_b32(t_rsrc, ok0 ? (unsigned)((m0 + r0t) * p.ld_res + n0 + c0t * 32) * 4u : 0x80000000u, 0, 0);
                touch1 = __builtin_amdgcn_raw_buffer_load_b32(t_rsrc, ok1 ? (unsigned)((m0 + r1t) * p.ld_res + n0 + c1t * 32) * 4u : 0x80000000u, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            FS_READ3(cur ^ 1, 0, araw, B3)  // (after the last chunk: a stale stage, read and never used)
            __builtin_amdgcn_sched_barrier(0);
            FS_STEP3(A3n, B3n, araw, A3)
            __builtin_amdgcn_sched_barrier(0);
            FS_TRACE_CHUNK(kc)
            cur ^= 1;
        }
        asm volatile("" ::"v"(touch0), "v"(touch1));  // keeps the two dead loads (and their registers) alive to here
        if (!DUAL && p.res) igemm_load_residual(rv, p, M, em_base, en_base);
#undef FS_READ3
#undef FS_STEP3
#undef FS_FAKE_A_OPAQUE
    } else {
    if (!DUAL && p.res && nchunks <= 2) igemm_load_residual(rv, p, M, em_base, en_base);
    FS_FRAGS(0, 0, a0, b0)
    for (int kc = 0; kc < nchunks; ++kc) {
        // sub-step 0's fragments were requested under the previous chunk's last MFMAs: start multiplying at once and
        // slip the reads of sub-step 1 in behind the first MFMA (a wait placed before them would cover the new reads too)
        FS_FRAGS(cur, 1, a1, b1)
        FS_MMA(a0, b0)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        FS_FRAGS(cur, 2, a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        FS_MMA(a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        FS_FRAGS(cur, 3, a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        FS_MMA(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        FS_TRACE_SYNC()  // __syncthreads: all fragment reads of stage `cur` are done (lgkmcnt(0)) and chunk kc+1 has landed (vmcnt(0))
        if (kc + 2 < nchunks) {
            FS_DMA_ALL(cur)
            FS_DMA_ADVANCE()
        }
        if (!DUAL && p.res && kc + 3 == nchunks) igemm_load_residual(rv, p, M, em_base, en_base);  // one and a quarter chunks to land
        if (kc + 1 < nchunks) FS_FRAGS(cur ^ 1, 0, a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        FS_MMA(a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    }
#ifdef FS_TRACE
    tr_loop = __builtin_readcyclecounter();
#endif
#undef FS_DMA_ROW
#undef FS_DMA_ADVANCE
#undef FS_DMA_ALL
#undef FS_DMA_SKIP
#undef FS_FRAGS
#undef FS_MMA

    // ---- epilogue
#ifdef FS_TRACE
    if ((p.dbg & 16) && p.ld_out >= 0) return;  // timing experiment: skip the epilogue (the test keeps the main loop alive)
#endif
    if constexpr (SPLIT && !DUAL && BN == 96 && TM == 1) {
        // the Segmenter's qkv Linear: K / V column tiles leave as the attention's operand planes (ConvParams::kv_k, igemm_epilogue.h)
        const int Dm = p.kv_heads * 64;
        if (p.kv_k && n0 >= Dm) {
            const int kind = n0 >= 2 * Dm ? 2 : 1;
            igemm_epilogue_kv<TN>(acc, sc_n, sh_n, p, kind, em_base, en_base - kind * Dm, en_base < p.Cout, lane);
            return;
        }
    }
    if (!DUAL && p.res) {
        if (p.relu == 1) igemm_epilogue<1, true>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
        else if (p.relu == 2) igemm_epilogue<2, true>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
        else igemm_epilogue<0, true>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
    } else {
        if (p.relu == 1) igemm_epilogue<1, false>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
        else if (p.relu == 2) igemm_epilogue<2, false>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
        else igemm_epilogue<0, false>(acc, rv, sc_n, sh_n, p, M, em_base, en_base);
    }
#ifdef FS_TRACE
    if (t == 0) {
        unsigned long long* o = fs_trace_buf + 8 * (size_t)(bid & 65535);
        o[0] = tr_start; o[1] = tr_ready; o[2] = tr_loop; o[3] = __builtin_readcyclecounter(); o[4] = tr_wait;
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o[6] = tr_wall; o[7] = wall_clock64();
        unsigned long long* c = fs_trace_chunks + 8 * (size_t)(bid & 65535);
        c[0] = tr_c0; c[1] = tr_c1; c[2] = tr_c2; c[3] = tr_c3; c[4] = tr_c4; c[5] = tr_c5; c[6] = tr_c6; c[7] = tr_c7;
    }
#endif
#endif
}

template <int BM, int BN, int WGM = 2, int WGN = 2, bool DUAL = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_dma_f32(ConvParams p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PM = 8;  // m-tiles per raster panel (panels sized to the ~64 tiles co-resident on an XCD: same time, +3 % L2 misses)
    __shared__ __attribute__((aligned(1024))) float lds[conv_tile_lds_floats<BM, BN, SPLIT, WGM * WGN>()];
    const int nblk = gridDim.x, bid = blockIdx.x;
#ifdef FS_TRACE
    // experiment: de-phase the workgroups that share a CU (dispatch order puts bid and bid + 256 on the same CU)
    if ((p.dbg & 32) && ((bid >> 8) & 1))
        for (int i = 0; i < (p.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(4);  // 256 cycles each
#endif
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int per_group = tiles_m * tiles_n;
    const int grp = lid / per_group;           // grouped GEMM: consecutive logical ids walk one group's tiles
    const int lig = lid - grp * per_group;
    const int panel = lig / (PM * tiles_n);
    const int within = lig - panel * (PM * tiles_n);
    const int prow = min(PM, tiles_m - panel * PM);
    const int m0 = (panel * PM + within % prow) * BM, n0 = (within / prow) * BN;
    p.in += (long long)grp * p.g_in;
    p.wgt += (long long)grp * p.g_wgt;
    if (SPLIT) p.wgt3 = (const char*)p.wgt3 + (long long)grp * p.g_wgt * 2;  // the group's rows inside every plane
    p.out += (long long)grp * p.g_out;
    p.kv_b = grp;
    conv_tile<BM, BN, WGM, WGN, DUAL, SPLIT>(p, m0, n0, lds, bid);
#endif
}

namespace {
struct TileCfg { int bm, bn; const char* name; };
const TileCfg kTiles[7] = {{0, 0, "auto"}, {128, 128, "igemm128x128"}, {128, 64, "igemm128x64"},
                           {64, 64, "igemm64x64"}, {64, 128, "igemm64x128"}, {0, 0, "(retired)"}, {128, 96, "split128x96"}};

int pick_tile(const ConvParams& p) {
    // Cost model fitted to the MI355X tile sweeps (profiles/r01_conv_tile_sweep*.txt): per-tile MFMA efficiency by tile
    // shape; a CU that hosts a single workgroup runs at ~0.8 of the rate it reaches with two or more co-resident ones;
    // the launch takes as long as its most loaded CU (ceil(tiles / 256) workgroups).
    const int M = p.B * p.Ho * p.Wo;
    // Tile 6 (128 x 96, split route, round 5) is a candidate only where 96 divides the columns: the Segmenter's Linears (d_model 384 / 768
    // and their multiples) -- 4052 token rows x 1536 columns are exactly 512 such tiles, two per CU, where 128 x 64 leaves 768 (1.5 per
    // slot).  No PSPNet / DeepLab layer qualifies (their channel counts are powers of two), so their choices are untouched.
    // (Rounds 4-5, profiles/r04_experiments.txt section 9, r05_experiments.txt section 18: 256 x 128 tiles -- four waves of 64 x 128, or eight of
    //  32 x 128 -- won 3-12 % on isolated single-round launches and nothing in a window; removed in round 6, the model stays as fitted.)
    const double eff[7] = {0, 1.00, 0.92, 0.80, 0.92, 0, 0.97};
    int best = 1;
    double best_t = 1e300;
    const int last = p.in2 ? 2 : p.wgt3 ? 3 : 4;
    for (int c = 1; c <= 6; ++c) {
        if (c > last && !(c == 6 && p.wgt3 && !p.in2 && p.Cout % 96 == 0)) continue;
        const int bm = kTiles[c].bm, bn = kTiles[c].bn;
        if (p.Cout < bn && bn > 64) continue;
        const long tiles = (long)cdiv(M, bm) * cdiv(p.Cout, bn) * (p.groups > 1 ? p.groups : 1);
        const long per_cu = (tiles + 255) / 256;
        const double t = (double)per_cu * bm * bn / (eff[c] * (per_cu >= 2 ? 1.0 : 0.8));
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}
}  // namespace

const char* conv_igemm_tile_name(const ConvParams& p, int tile) {
    tile &= 0xff;
    if (tile <= 0 || tile > 6 || tile == 5) tile = pick_tile(p);
    if (p.in2 && p.wgt3) return tile == 2 ? "split128x64cat" : "split128x128cat";
    if (p.in2) return tile == 2 ? "igemm128x64cat" : "igemm128x128cat";  // the concatenated-K instantiations are kernels of their own
    if (p.wgt3) return tile == 1 ? "split128x128" : tile == 2 ? "split128x64" : tile == 4 ? "split64x128" : tile == 6 ? "split128x96" : "split64x64";
    return kTiles[tile].name;
}

namespace {
// filter rows of the split bank are bf16: 16-B DMA pieces need a row stride that is a multiple of 8 elements
bool ldw_ok(const ConvParams& p) { return (p.ld_wgt ? p.ld_wgt : p.KH * p.KW * p.Cin + (p.in2 ? p.Cin2 : 0)) % 8 == 0; }

__global__ void split_bf16x3_kernel(const float* __restrict__ w, long long n, __bf16* __restrict__ planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    planes[i] = h;
    planes[n + i] = m;
    planes[2 * n + i] = (__bf16)(r - (float)m);
}
}  // namespace

int launch_split_bf16x3(const float* w, long long n, void* planes, hipStream_t s) {
    FS_REQUIRE(w && planes && n > 0 && n % 8 == 0, "split_bf16x3: bad arguments (n=%lld)", n);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, (__bf16*)planes);
    FS_HIP(hipGetLastError());
    return 0;
}

namespace {
// argument checks of launch_conv_igemm
int check_conv_params(const ConvParams& p) {
    FS_REQUIRE(p.Cin % 32 == 0, "conv_igemm: Cin=%d must be a multiple of 32", p.Cin);
    if (p.in2) {
        FS_REQUIRE(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.res == nullptr && p.groups <= 1,
                   "conv_igemm: a concatenated-K launch takes two 1x1 convs, the first with stride 1, and no residual");
        FS_REQUIRE(p.Cin2 >= 32 && p.Cin2 % 32 == 0 && p.ld_in2 % 4 == 0 && p.ld_in2 >= p.Cin2 && p.stride2 >= 1 && ((uintptr_t)p.in2 & 15) == 0,
                   "conv_igemm: bad second operand (Cin2=%d ld=%d stride=%d)", p.Cin2, p.ld_in2, p.stride2);
        FS_REQUIRE((p.H2 - 1) / p.stride2 + 1 == p.Ho && (p.W2 - 1) / p.stride2 + 1 == p.Wo, "conv_igemm: second operand %dx%d / stride %d does not give %dx%d",
                   p.H2, p.W2, p.stride2, p.Ho, p.Wo);
        FS_REQUIRE((int64_t)p.B * p.H2 * p.W2 * p.ld_in2 * 4 < (int64_t)1 << 31, "conv_igemm: second operand must be smaller than 2 GiB");
    }
    FS_REQUIRE(p.ld_in % 4 == 0 && p.ld_in >= p.Cin, "conv_igemm: bad ld_in=%d (Cin=%d)", p.ld_in, p.Cin);
    FS_REQUIRE(p.ld_out >= p.Cout, "conv_igemm: bad ld_out=%d (Cout=%d)", p.ld_out, p.Cout);
    FS_REQUIRE(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0, "conv_igemm: unaligned operand");
    FS_REQUIRE(p.Ho == (p.H + 2 * p.pad - p.dil * (p.KH - 1) - 1) / p.stride + 1 &&
               p.Wo == (p.W + 2 * p.pad - p.dil * (p.KW - 1) - 1) / p.stride + 1,
               "conv_igemm: output geometry %dx%d inconsistent with input %dx%d", p.Ho, p.Wo, p.H, p.W);
    // the DMA kernel addresses both operands with 32-bit byte offsets and uses 0x80000000 as the out-of-range sentinel
    FS_REQUIRE(((int64_t)p.B * p.H * p.W * p.ld_in + ((int64_t)p.pad * p.W + p.pad) * p.ld_in) * 4 < (int64_t)1 << 31,
               "conv_igemm: input tensor must be smaller than 2 GiB");
    FS_REQUIRE((int64_t)p.Cout * std::max(p.ld_wgt, p.KH * p.KW * p.Cin + (p.in2 ? p.Cin2 : 0)) * 4 < (int64_t)1 << 31, "conv_igemm: filter bank must be smaller than 2 GiB");
    FS_REQUIRE(p.ld_wgt == 0 || (p.ld_wgt >= p.KH * p.KW * p.Cin + (p.in2 ? p.Cin2 : 0) && p.ld_wgt % 4 == 0), "conv_igemm: bad ld_wgt=%d", p.ld_wgt);
    FS_REQUIRE(p.KH * p.KW <= 32, "conv_igemm: at most 32 filter taps");
    // the epilogue stores (and reads the residual) through 32-bit-offset buffer descriptors as well
    FS_REQUIRE((int64_t)p.B * p.Ho * p.Wo * p.ld_out * 4 < (int64_t)1 << 31 && ((uintptr_t)p.out & 3) == 0,
               "conv_igemm: output tensor must be smaller than 2 GiB");
    FS_REQUIRE(p.res == nullptr || (int64_t)p.B * p.Ho * p.Wo * p.ld_res * 4 < (int64_t)1 << 31, "conv_igemm: residual tensor must be smaller than 2 GiB");
    if (!(p.res == nullptr || p.ld_res >= p.Cout)) return fail("conv_igemm: bad ld_res");
    return 0;
}
}  // namespace

int launch_conv_igemm(const ConvParams& p, hipStream_t s, int tile) {
    FS_TRY(check_conv_params(p));
    tile &= 0xff;
    if (tile <= 0 || tile > 6 || tile == 5) tile = pick_tile(p);
    FS_REQUIRE(tile < 6 || (p.wgt3 && !p.in2), "conv_igemm: tile 6 (128 x 96) exists on the split-operand route only");
    const int M = p.B * p.Ho * p.Wo;
    const int bm = kTiles[tile].bm, bn = kTiles[tile].bn;
    const int tm = cdiv(M, bm), tn = cdiv(p.Cout, bn);
    const int groups = p.groups > 1 ? p.groups : 1;
    FS_REQUIRE(groups == 1 || p.res == nullptr, "conv_igemm: grouped GEMM has no residual input");
    const dim3 grid(tm * tn * groups), block(256);
    if (p.kv_k) {
        const int Dm = p.kv_heads * 64;
        FS_REQUIRE(tile == 6 && p.wgt3 && !p.in2 && !p.res && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.relu == 0,
                   "conv_igemm: the K / V plane epilogue belongs to a plain Linear on the 128 x 96 split tile");
        FS_REQUIRE(p.kv_vt && p.kv_heads >= 1 && Dm % 96 == 0 && p.Cout == 3 * Dm && p.kv_N == M && p.kv_Npad % 32 == 0 && p.kv_Npad >= M &&
                       p.kv_Npad <= tm * bm && (int64_t)3 * p.kv_plane_bytes < (int64_t)1 << 31 &&
                       (int64_t)p.kv_plane_bytes == (int64_t)groups * p.kv_heads * p.kv_Npad * 128 && (((uintptr_t)p.kv_k | (uintptr_t)p.kv_vt) & 15) == 0,
                   "conv_igemm: bad K / V plane geometry (heads %d, tokens %d, padded %d)", p.kv_heads, p.kv_N, p.kv_Npad);
    }
    if (p.wgt3) {
        FS_REQUIRE(((uintptr_t)p.wgt3 & 15) == 0 && p.plane_bytes % 16 == 0 && (int64_t)3 * p.plane_bytes < (int64_t)1 << 31 && ldw_ok(p),
                   "conv_igemm: bad split filter bank (plane_bytes=%u)", p.plane_bytes);
        FS_REQUIRE((int64_t)groups * p.g_wgt * 2 <= (int64_t)p.plane_bytes, "conv_igemm: the groups' filters (%d x %lld) exceed a plane of the split bank", groups, p.g_wgt);
    }
    if (p.in2 && p.wgt3) {
        if (tile == 2) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 64, 4, 1, true, true>), grid, block, 0, s, p, tm, tn);
        else if (tile == 1) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 128, 4, 1, true, true>), grid, block, 0, s, p, tm, tn);
        else return fail("conv_igemm: concatenated-K launches use tile 1 or 2");
        FS_HIP(hipGetLastError());
        return 0;
    }
    if (p.in2) {  // concatenated-K instantiations exist for the two 128-row tiles
        if (tile == 2) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 64, 2, 2, true>), grid, block, 0, s, p, tm, tn);
        else if (tile == 1) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 128, 2, 2, true>), grid, block, 0, s, p, tm, tn);
        else return fail("conv_igemm: concatenated-K launches use tile 1 or 2");
        FS_HIP(hipGetLastError());
        return 0;
    }
    if (p.wgt3) {  // split-operand route: same tiles, bf16 matrix pipe
        switch (tile) {
            // 4 x 1 waves (32 pixel rows x the whole tile width each): every pixel of the tile is split ONCE per workgroup (2 x 2 waves
            // split it twice).  Under this kernel the chip is power-limited (1.25-1.8 GHz, box dependent), so VALU work saved comes
            // back as clock: K = 2048 layers +6 %, the others +0..2 % (profiles/r03_split_operands.txt)
            case 1: hipLaunchKernelGGL((conv_igemm_dma_f32<128, 128, 4, 1, false, true>), grid, block, 0, s, p, tm, tn); break;
            case 2: hipLaunchKernelGGL((conv_igemm_dma_f32<128, 64, 4, 1, false, true>), grid, block, 0, s, p, tm, tn); break;
            case 3: hipLaunchKernelGGL((conv_igemm_dma_f32<64, 64, 2, 2, false, true>), grid, block, 0, s, p, tm, tn); break;
            // round 5: 64 rows x 128 columns, 2 x 2 waves of 32 x 64 (3 split instructions per MFMA against 6 in the 64 x 64 tile): the
            // shape for GEMMs with few rows and many columns (the Segmenter's Linears: 4052 token rows)
            case 4: hipLaunchKernelGGL((conv_igemm_dma_f32<64, 128, 2, 2, false, true>), grid, block, 0, s, p, tm, tn); break;
            // round 5: 128 rows x 96 columns (4 x 1 waves of 32 x 96), for column counts that are multiples of 96 -- see pick_tile
            case 6: hipLaunchKernelGGL((conv_igemm_dma_f32<128, 96, 4, 1, false, true>), grid, block, 0, s, p, tm, tn); break;
            default: return fail("conv_igemm: the split-operand route has tiles 1, 2, 3, 4 and 6");
        }
        FS_HIP(hipGetLastError());
        return 0;
    }
#ifdef FS_TRACE
    const size_t dyn = (p.dbg & 2) ? 56 * 1024 : 0;  // timing experiment: push occupancy to one block per CU
#else
    const size_t dyn = 0;
#endif
#define FS_CONV_LAUNCH(BM_, BN_) hipLaunchKernelGGL((conv_igemm_dma_f32<BM_, BN_>), grid, block, dyn, s, p, tm, tn);
    switch (tile) {
        case 1: FS_CONV_LAUNCH(128, 128) break;
        case 2: FS_CONV_LAUNCH(128, 64) break;
        case 3: FS_CONV_LAUNCH(64, 64) break;
        default: FS_CONV_LAUNCH(64, 128) break;
    }
#undef FS_CONV_LAUNCH
    FS_HIP(hipGetLastError());
    return 0;
}


}  // namespace fs
