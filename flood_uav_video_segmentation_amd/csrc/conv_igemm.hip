// fp32 implicit-GEMM convolution on the CDNA4 matrix cores.
//
//   v_mfma_f32_32x32x2_f32: exact f32 fmaf-chain numerics at 64 FLOP/clk/SIMD (157 TFLOP/s chip peak).
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

#include <algorithm>
#include <cstdlib>

namespace fs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
#if defined(__HIP_DEVICE_COMPILE__)
// Epilogue of the DMA kernel: y = act(acc * scale + shift (+ residual)) stored through a buffer descriptor whose range
// check IS the row / channel guard (num_records = M * ld * 4: rows >= M fall outside, lanes with n >= Cout carry the
// sentinel offset), so there is no per-element branch, no 64-bit address arithmetic and the activation is resolved at
// compile time: 5 instructions per element instead of ~85 (the first version spent ~11 k cycles per 128x128 tile here,
// as much as two K chunks).  The row offset is added on the VALU: the scalar offset of a buffer op is not range-checked.
// Each dword store instruction covers two full 128-B lines.  (Measured and rejected: transposed accumulators -- filter
// fragment as the MFMA's A operand -- so that a lane owns 4 consecutive channels of one pixel and stores 16 B: 4x fewer
// store instructions but 32 separate 32-B segments each; epilogue 8 k -> 12-15 k cycles, 818 -> 790 FPS.  And an
// epilogue interleaved with the next tile's main loop was priced with fake stores: it slows the loop by what it saves.)
// Residual tile (the bottleneck shortcut): TM*TN*16 dword loads per lane, issued by the main loop one chunk before its
// end so that they land under the last MFMAs instead of at the head of the epilogue.
template <int TM, int TN>
__device__ __forceinline__ void igemm_load_residual(float (&rv)[TM][TN][16], const ConvParams& p, int M, int m_base, int n_base) {
    constexpr unsigned SENT = 0x80000000u;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (unsigned)((long long)M * p.ld_res * 4), 0x00020000);
    const unsigned row_r = (unsigned)p.ld_res * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned vr = n < p.Cout ? (unsigned)(m_base + i * 32) * row_r + (unsigned)n * 4u : SENT;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                rv[i][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vr + (unsigned)((e & 3) + 8 * (e >> 2)) * row_r, 0, 0));
        }
    }
}

template <int ACT, bool RES, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const f32x16 (&acc)[TM][TN], const float (&rv)[TM][TN][16], const float (&sc)[TN],
                                               const float (&sh)[TN], const ConvParams& p, int M, int m_base, int n_base) {
    constexpr unsigned SENT = 0x80000000u;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (unsigned)((long long)M * p.ld_out * 4), 0x00020000);
    const unsigned row_o = (unsigned)p.ld_out * 4u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + j * 32;
        const bool nok = n < p.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m_base + i * 32;
            const unsigned vo = nok ? (unsigned)mb * row_o + (unsigned)n * 4u : SENT;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] * sc[j] + sh[j];
                if (RES) v += rv[i][j][e];
                if (ACT == 1) v = fmaxf(v, 0.f);
                else if (ACT == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // nn.GELU (erf form)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo + (unsigned)((e & 3) + 8 * (e >> 2)) * row_o, 0, 0);
            }
        }
    }
}
#endif

#ifdef FS_TRACE
// tools/probe_conv_trace.hip only (never in libfloodseg.so): per-workgroup timeline, 8 x u64 per workgroup:
// [0] start, [1] first stage landed, [2] main loop done, [3] epilogue done (shader clock, s_memtime),
// [4] cycles spent in the per-chunk wait+barrier, [5] HW_ID | XCC_ID << 32, [6] start (100 MHz wall clock), [7] end (wall)
__device__ unsigned long long fs_trace_buf[8 * 65536];
#define FS_TRACE_DECL unsigned long long tr_start = __builtin_readcyclecounter(), tr_wall = wall_clock64(), tr_ready = 0, tr_loop = 0, tr_wait = 0;
#define FS_TRACE_SYNC() { const unsigned long long tw = __builtin_readcyclecounter(); FS_DMA_PUBLISH() tr_wait += __builtin_readcyclecounter() - tw; }
#else
#define FS_TRACE_DECL
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
template <int BM, int BN, int WGM = 2, int WGN = 2, bool DUAL = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_dma_f32(ConvParams p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins do not exist in the host pass, which only needs the launch stub
    constexpr int BK = 32;
    constexpr int NT = 64 * WGM * WGN;   // threads
    constexpr int RSTEP = NT / 8;        // tile rows staged per DMA pass (8 lanes per 128-B row)
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RA = BM / RSTEP, RB = BN / RSTEP;
    constexpr int PM = 8;  // m-tiles per raster panel (panels sized to the ~64 tiles co-resident on an XCD: same time, +3 % L2 misses)
    constexpr int STAGE = (BM + BN) * BK;
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    FS_TRACE_DECL

    const int nblk = gridDim.x, bid = blockIdx.x;
#ifdef FS_TRACE
    // experiment: de-phase the workgroups that share a CU (dispatch order puts bid and bid + 256 on the same CU)
    if ((p.dbg & 32) && ((bid >> 8) & 1))
        for (int i = 0; i < (p.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(16);  // 1024 cycles each
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
    p.out += (long long)grp * p.g_out;

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
        __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, (unsigned)((((long long)p.Cout - 1) * ldw + K) * 4), 0x00020000);
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
    unsigned b_voff[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = n0 + r0 + RSTEP * j;
        b_voff[j] = n < p.Cout ? (unsigned)((n * ldw + swz * 4) * 4) : SENT;
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
#define FS_DMA_ROW(STG, ROW_)                                                                                     \
    if ((ROW_) < RA) {                                                                                            \
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
        b_soff += 128;                                                                                            \
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
#ifdef FS_TRACE
    tr_loop = __builtin_readcyclecounter();
#endif
#undef FS_DMA_ROW
#undef FS_DMA_ADVANCE
#undef FS_DMA_ALL
#undef FS_FRAGS
#undef FS_MMA

    // ---- epilogue
#ifdef FS_TRACE
    if ((p.dbg & 16) && p.ld_out >= 0) return;  // timing experiment: skip the epilogue (the test keeps the main loop alive)
#endif
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
    }
#endif
#endif
}

namespace {
struct TileCfg { int bm, bn; const char* name; };
const TileCfg kTiles[6] = {{0, 0, "auto"}, {128, 128, "igemm128x128"}, {128, 64, "igemm128x64"},
                           {64, 64, "igemm64x64"}, {64, 128, "igemm64x128"}, {256, 128, "igemm256x128"}};

int pick_tile(const ConvParams& p) {
    // Cost model fitted to the MI355X tile sweeps (profiles/r01_conv_tile_sweep*.txt): per-tile MFMA efficiency by tile
    // shape; a CU that hosts a single workgroup runs at ~0.8 of the rate it reaches with two or more co-resident ones;
    // the launch takes as long as its most loaded CU (ceil(tiles / 256) workgroups).
    const int M = p.B * p.Ho * p.Wo;
    const double eff[5] = {0, 1.00, 0.92, 0.80, 0.92};
    int best = 1;
    double best_t = 1e300;
    for (int c = 1; c <= (p.in2 ? 2 : 4); ++c) {
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
    if (tile <= 0 || tile > 5) tile = pick_tile(p);
    if (p.in2) return tile == 2 ? "igemm128x64cat" : "igemm128x128cat";  // the concatenated-K instantiations are kernels of their own
    return kTiles[tile].name;
}

int launch_conv_igemm(const ConvParams& p, hipStream_t s, int tile) {
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
    tile &= 0xff;
    if (tile <= 0 || tile > 5) tile = pick_tile(p);
    const int M = p.B * p.Ho * p.Wo;
    const int bm = kTiles[tile].bm, bn = kTiles[tile].bn;
    const int tm = cdiv(M, bm), tn = cdiv(p.Cout, bn);
    const int groups = p.groups > 1 ? p.groups : 1;
    FS_REQUIRE(groups == 1 || p.res == nullptr, "conv_igemm: grouped GEMM has no residual input");
    const dim3 grid(tm * tn * groups), block(256);
    if (p.in2) {  // concatenated-K instantiations exist for the two 128-row tiles
        if (tile == 2) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 64, 2, 2, true>), grid, block, 0, s, p, tm, tn);
        else if (tile == 1) hipLaunchKernelGGL((conv_igemm_dma_f32<128, 128, 2, 2, true>), grid, block, 0, s, p, tm, tn);
        else return fail("conv_igemm: concatenated-K launches use tile 1 or 2");
        FS_HIP(hipGetLastError());
        return 0;
    }
    if (tile == 5) {  // 8-wave workgroup, one per CU
        hipLaunchKernelGGL((conv_igemm_dma_f32<256, 128, 4, 2>), grid, dim3(512), 0, s, p, tm, tn);
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
