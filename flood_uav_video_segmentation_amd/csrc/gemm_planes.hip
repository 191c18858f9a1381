// Split-operand GEMM whose ROW operand arrives pre-split (round 4): out[m][n] = act(scale[n] * sum_k A[m][k] * W[n][k] + shift[n]).
//
// conv_igemm_dma_f32<..., SPLIT = true> stages fp32 pixels and splits every fragment into its three bf16 terms in registers, once
// per workgroup and 16-deep step: 36 VALU instructions per 24 MFMAs and lane.  Here the PRODUCER of the row operand writes the three
// bf16 planes (the Winograd input transform, winograd.hip) and this kernel's main loop is LDS-DMA, fragment reads and
// v_mfma_f32_32x32x16_bf16 -- nothing else.
//
//   x = h + m + l exactly (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)); x * w = hh + (hm + mh) + (mm + hl + lh) + [dropped:
//   ml + lm + ll <= 2^-23 |x w|] -- the same six products in the same order as the in-register kernel, so the two kernels give
//   bit-identical sums for the same operands (tests/test_gpu_ops.py).
//
// MEASURED (MI355X, profiles/r04_experiments.txt, r04_plane_gemm_bench.txt): taking the split out of the main loop buys nothing.
// Sustained, interleaved runs on random operands: layer4.conv1 (16200 x 512 x 2048) 150.0 us here vs 152.1 in registers; layer4.conv3
// 163.9 vs 156.0; layer3.conv3 (K = 256) 47.1 vs 44.9.  Both kernels sit at the clock the chip holds under dense bf16 MFMA load
// (223-227 fp32-equivalent TFLOP/s = 1.35 PFLOP/s executed on the K = 2048 shape: the guide's bare-loop figure): the MFMAs themselves
// are the power, not the 36 VALU instructions beside them.  Inside the network the route LOSES: V as planes is 1.5x the bytes for the
// transform to write and the GEMM to read (head: 352 vs 324 us, transforms +0.05 ms per window).  So the route is opt-in
// (FS_OPT_PLANE_OPERANDS) and parity-tested, not the default.  The DEV build's v_mfma_f32_16x16x32_bf16 variant of this kernel
// (below) is what DID move: +8-10 % on the K = 2048 x N = 512 shape (138 us), +-2 % on the others.
//
// Workgroup = 8 waves (512 threads, ONE per CU: both operands as planes are 72 KB per 32-deep stage, two stages fill the CU's LDS)
// over a 256 x BN tile, BN = 128 or 64; wave w owns rows 32 w .. 32 w + 31 and all BN columns, so its row fragments are its own
// DMA (no other wave reads them) and the filter planes are shared by all eight waves -- half the filter traffic per CU of two
// 128-row workgroups.  LDS image of both operands: [plane][row][32 bf16] (64-B rows), the 16-B piece index XOR-swizzled with
// (row >> 2) & 3: a fragment read (one piece of 16 rows) covers all 64 banks once.  One DMA wave-instruction = 16 rows of one plane.
// Padding rows (m >= M, n >= N) are the buffer descriptor's range check (sentinel offset -> zeros in LDS), as in conv_igemm.hip.
#include "igemm_epilogue.h"
#include "kernels.h"

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FS_PUBLISH() { __builtin_amdgcn_s_waitcnt(0x0070); __syncthreads(); }  // vmcnt(0) lgkmcnt(0), then the barrier (conv_igemm.hip)

template <int BN>
__global__ __launch_bounds__(512) void gemm_planes_bf16x3(PlaneGemmParams p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, TN = BN / 32;
    constexpr int A_PL = BM * 16, B_PL = BN * 16;       // floats of LDS per plane and stage (64-B rows)
    constexpr int STAGE = 3 * (A_PL + B_PL);
    constexpr int NB = 3 * (BN / 16);                   // filter DMA pieces per stage, dealt round-robin over the 8 waves
    constexpr int RB = (NB + 7) / 8;
    constexpr int PM = 4;                               // m-tiles per raster panel
    constexpr unsigned SENT = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;  // blocks b, b + 8 share an XCD: give each XCD a contiguous run of tiles
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int per_group = tiles_m * tiles_n;
    const int grp = lid / per_group;
    const int lig = lid - grp * per_group;
    const int panel = lig / (PM * tiles_n);
    const int within = lig - panel * (PM * tiles_n);
    const int prow = min(PM, tiles_m - panel * PM);
    const int m0 = (panel * PM + within % prow) * BM, n0 = (within / prow) * BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int nchunks = p.K >> 5;

    const char* a_base = (const char*)p.a3 + (long long)grp * p.g_a * 2;
    const char* b_base = (const char*)p.b3 + (long long)grp * p.g_b * 2;
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 2u * p.a_plane_bytes + (unsigned)((((long long)p.M - 1) * p.ld_a + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 2u * p.b_plane_bytes + (unsigned)((((long long)p.N - 1) * p.ld_b + p.K) * 2), 0x00020000);
    // lane -> (row, 16-B slot) of a 16-row x 64-B piece; the slot holds piece slot ^ ((row >> 2) & 3) of the row's 32 bf16
    const unsigned piece = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    unsigned a_voff[2], b_voff[RB];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 32 * wv + 16 * j + (lane >> 2);
        a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.ld_a * 2) + piece : SENT;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int idx = wv + 8 * j;  // piece idx = plane * (BN / 16) + row group
        const int n = n0 + 16 * (idx % (BN / 16)) + (lane >> 2);
        b_voff[j] = (idx < NB && n < p.N) ? (unsigned)n * (unsigned)(p.ld_b * 2) + piece : SENT;
    }

    f32x16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    unsigned k_off = 0;  // byte offset along k of the chunk being fetched
#define FS_DMA(STG)                                                                                                \
    {                                                                                                              \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + pl * A_PL + (32 * wv + 16 * j) * 16), \
                                                         16, a_voff[j], k_off + (unsigned)pl * p.a_plane_bytes, 0, 0);  \
        _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                                           \
            const int idx = wv + 8 * j;                                                                            \
            if (idx < NB)                                                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + 3 * A_PL + (idx / (BN / 16)) * B_PL + (idx % (BN / 16)) * 256), \
                                                         16, b_voff[j], k_off + (unsigned)(idx / (BN / 16)) * p.b_plane_bytes, 0, 0); \
        }                                                                                                          \
        k_off += 64;                                                                                               \
    }
#define FS_READ(STG, S_, A_, B_)                                                                                   \
    {                                                                                                              \
        const float* a_src = lds + (STG) * STAGE + (32 * wv + l31) * 16;                                           \
        const float* b_src = lds + (STG) * STAGE + 3 * A_PL + l31 * 16;                                            \
        const int slot = 4 * ((2 * (S_) + hh) ^ ((l31 >> 2) & 3));                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
            A_[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&a_src[pl * A_PL + slot]));        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                       \
                B_[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&b_src[pl * B_PL + j * 512 + slot])); \
    }
    // the six cross products, smallest first: (l h) (h l) (m m) (m h) (h m) (h h) -- the order of conv_igemm_dma_f32's split loop
#define FS_STEP(A_, B_)                                                                                            \
    _Pragma("unroll") for (int term = 0; term < 6; ++term) {                                                       \
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[PA[term]], B_[j][PB[term]], acc[0][j], 0, 0, 0); \
    }

    FS_DMA(0)
    FS_PUBLISH()
    if (nchunks > 1) FS_DMA(1)
    const int em_base = m0 + 32 * wv + 4 * hh, en_base = n0 + l31;
    float sc_n[TN], sh_n[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = en_base + j * 32;
        sc_n[j] = (n < p.N && p.scale) ? p.scale[n] : 1.f;
        sh_n[j] = (n < p.N && p.shift) ? p.shift[n] : 0.f;
    }
    bf16x8 A3[3], B3[TN][3], A3n[3], B3n[TN][3];
    int cur = 0;
    FS_READ(0, 0, A3, B3)
    for (int kc = 0; kc < nchunks; ++kc) {
        FS_READ(cur, 1, A3n, B3n)
        __builtin_amdgcn_sched_barrier(0);
        FS_STEP(A3, B3)
        __builtin_amdgcn_sched_barrier(0);
        FS_PUBLISH()  // every wave's reads of stage `cur` are done, chunk kc + 1 has landed
        if (kc + 2 < nchunks) FS_DMA(cur)
        __builtin_amdgcn_sched_barrier(0);
        FS_READ(cur ^ 1, 0, A3, B3)  // (after the last chunk: a stale stage, read and never used)
        __builtin_amdgcn_sched_barrier(0);
        FS_STEP(A3n, B3n)
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
#undef FS_DMA
#undef FS_READ
#undef FS_STEP

    ConvParams e{};  // the shared epilogue's view of the output
    e.out = p.out + (long long)grp * p.g_out;
    e.ld_out = p.ld_out;
    e.Cout = p.N;
    float rv[1][TN][16];
    if (p.relu == 1) igemm_epilogue<1, false>(acc, rv, sc_n, sh_n, e, p.M, em_base, en_base);
    else if (p.relu == 2) igemm_epilogue<2, false>(acc, rv, sc_n, sh_n, e, p.M, em_base, en_base);
    else igemm_epilogue<0, false>(acc, rv, sc_n, sh_n, e, p.M, em_base, en_base);
#endif
}


#ifdef FS_DEV
// Development variant (make DEV=1 only): the same GEMM on v_mfma_f32_16x16x32_bf16.  Under dense bf16 MFMA load the chip holds a
// higher clock with this shape (MI355X guide, DVFS item 7: 1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP).
// A 32-deep chunk is ONE MFMA step; the wave tile 32 x 128 = 2 x 8 blocks of 16 x 16; filter fragments are streamed in four
// groups of two column blocks (6 ds_read_b128 per 24 MFMAs), double-buffered in registers.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void gemm_planes_bf16x3_s16(PlaneGemmParams p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BN = 128;
    constexpr int A_PL = BM * 16, B_PL = BN * 16;
    constexpr int STAGE = 3 * (A_PL + B_PL);
    constexpr int NB = 3 * (BN / 16);
    constexpr int RB = NB / 8;
    constexpr int PM = 4;
    constexpr unsigned SENT = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int per_group = tiles_m * tiles_n;
    const int grp = lid / per_group;
    const int lig = lid - grp * per_group;
    const int panel = lig / (PM * tiles_n);
    const int within = lig - panel * (PM * tiles_n);
    const int prow = min(PM, tiles_m - panel * PM);
    const int m0 = (panel * PM + within % prow) * BM, n0 = (within / prow) * BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int nchunks = p.K >> 5;
    const char* a_base = (const char*)p.a3 + (long long)grp * p.g_a * 2;
    const char* b_base = (const char*)p.b3 + (long long)grp * p.g_b * 2;
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 2u * p.a_plane_bytes + (unsigned)((((long long)p.M - 1) * p.ld_a + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 2u * p.b_plane_bytes + (unsigned)((((long long)p.N - 1) * p.ld_b + p.K) * 2), 0x00020000);
    // swizzle key {0, 2, 3, 1}[(row >> 2) & 3]: conflict-free for the 16x16x32 fragment pattern (lane = row & 15, piece = lane >> 4)
    const int keyd = (0x78 >> (2 * ((lane >> 4) & 3))) & 3;  // DMA lane: row = lane >> 2 -> (row >> 2) & 3 = (lane >> 4) & 3
    const unsigned piece = (unsigned)(((lane & 3) ^ keyd) * 16);
    unsigned a_voff[2], b_voff[RB];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 32 * wv + 16 * j + (lane >> 2);
        a_voff[j] = m < p.M ? (unsigned)m * (unsigned)(p.ld_a * 2) + piece : SENT;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int idx = wv + 8 * j;
        const int n = n0 + 16 * (idx % (BN / 16)) + (lane >> 2);
        b_voff[j] = n < p.N ? (unsigned)n * (unsigned)(p.ld_b * 2) + piece : SENT;
    }
    f32x4v acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v(0.f);
    unsigned k_off = 0;
#define FS_DMA(STG)                                                                                                \
    {                                                                                                              \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + pl * A_PL + (32 * wv + 16 * j) * 16), \
                                                         16, a_voff[j], k_off + (unsigned)pl * p.a_plane_bytes, 0, 0);  \
        _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                                           \
            const int idx = wv + 8 * j;                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(lds + (STG) * STAGE + 3 * A_PL + (idx / (BN / 16)) * B_PL + (idx % (BN / 16)) * 256), \
                                                     16, b_voff[j], k_off + (unsigned)(idx / (BN / 16)) * p.b_plane_bytes, 0, 0); \
        }                                                                                                          \
        k_off += 64;                                                                                               \
    }
    const int keyr = (0x78 >> (2 * ((l15 >> 2) & 3))) & 3;
    const int frag = l15 * 16 + 4 * (kq ^ keyr);  // this lane's 16-B piece inside a 16-row block of 64-B rows
#define FS_READ_A(STG, A_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
            A_[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&lds[(STG) * STAGE + pl * A_PL + (32 * wv + 16 * i) * 16 + frag]));
#define FS_READ_B(STG, G_, B_)                                                                                     \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                               \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                           \
            B_[jj][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&lds[(STG) * STAGE + 3 * A_PL + pl * B_PL + (2 * (G_) + jj) * 256 + frag]));
#define FS_GROUP(G_, A_, B_)                                                                                       \
    _Pragma("unroll") for (int term = 0; term < 6; ++term) {                                                       \
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                       \
                acc[i][2 * (G_) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[i][PA[term]], B_[jj][PB[term]], acc[i][2 * (G_) + jj], 0, 0, 0); \
    }
    FS_DMA(0)
    FS_PUBLISH()
    if (nchunks > 1) FS_DMA(1)
    bf16x8 Aa[2][3], Ab[2][3], B0[2][3], B1[2][3];
    FS_READ_A(0, Aa)
    FS_READ_B(0, 0, B0)
    // two chunks per trip so that the fragment double buffers alternate without copies
#define FS_CHUNK(A_, AN_)                                                                                          \
    {                                                                                                              \
        FS_READ_B(cur, 1, B1)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_GROUP(0, A_, B0)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_READ_B(cur, 2, B0)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_GROUP(1, A_, B1)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_READ_B(cur, 3, B1)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_GROUP(2, A_, B0)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_PUBLISH()                                                                                               \
        if (kc + 2 < nchunks) FS_DMA(cur)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_READ_A(cur ^ 1, AN_)                                                                                    \
        FS_READ_B(cur ^ 1, 0, B0)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        FS_GROUP(3, A_, B1)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        cur ^= 1;                                                                                                  \
    }
    int cur = 0;
    for (int kc = 0; kc < nchunks; ++kc) {
        FS_CHUNK(Aa, Ab)
        _Pragma("unroll") for (int i = 0; i < 2; ++i)
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) Aa[i][pl] = Ab[i][pl];
    }
#undef FS_CHUNK
#undef FS_DMA
#undef FS_READ_A
#undef FS_READ_B
#undef FS_GROUP
    // epilogue: D block 16 x 16: column n = lane & 15, rows 4 * (lane >> 4) + r
    float* out = p.out + (long long)grp * p.g_out;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (unsigned)((long long)p.M * p.ld_out * 4), 0x00020000);
    const unsigned row_o = (unsigned)p.ld_out * 4u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + 16 * j + l15;
        const float sc = (n < p.N && p.scale) ? p.scale[n] : 1.f, sh = (n < p.N && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + 32 * wv + 16 * i + 4 * kq;
            const unsigned vo = n < p.N ? (unsigned)mb * row_o + (unsigned)n * 4u : SENT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] * sc + sh;
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo + (unsigned)r * row_o, 0, 0);
            }
        }
    }
#endif
}
#endif  // FS_DEV

int launch_gemm_planes(const PlaneGemmParams& p, hipStream_t s, int bn) {
    FS_REQUIRE(p.a3 && p.b3 && p.out && p.M >= 1 && p.N >= 1 && p.K >= 32 && p.K % 32 == 0, "gemm_planes: bad arguments (M=%d N=%d K=%d)", p.M, p.N, p.K);
    FS_REQUIRE(p.ld_a >= p.K && p.ld_a % 8 == 0 && p.ld_b >= p.K && p.ld_b % 8 == 0 && p.ld_out >= p.N, "gemm_planes: bad strides (ld_a=%d ld_b=%d ld_out=%d)",
               p.ld_a, p.ld_b, p.ld_out);
    FS_REQUIRE(((uintptr_t)p.a3 & 15) == 0 && ((uintptr_t)p.b3 & 15) == 0 && ((uintptr_t)p.out & 3) == 0 && p.a_plane_bytes % 16 == 0 && p.b_plane_bytes % 16 == 0,
               "gemm_planes: unaligned operand");
    const int groups = p.groups > 1 ? p.groups : 1;
    FS_REQUIRE((p.g_a * 2) % 16 == 0 && (p.g_b * 2) % 16 == 0, "gemm_planes: group strides must keep 16-B alignment");
    // 32-bit byte offsets inside one group's view of every plane set, 0x80000000 as the out-of-range sentinel
    FS_REQUIRE(2ll * p.a_plane_bytes + ((long long)(p.M - 1) * p.ld_a + p.K) * 2 < (1ll << 31), "gemm_planes: row operand must be smaller than 2 GiB");
    FS_REQUIRE(2ll * p.b_plane_bytes + ((long long)(p.N - 1) * p.ld_b + p.K) * 2 < (1ll << 31), "gemm_planes: filter planes must be smaller than 2 GiB");
    FS_REQUIRE((long long)p.M * p.ld_out * 4 < (1ll << 31), "gemm_planes: output must be smaller than 2 GiB");
    // the last group's rows must still lie inside a plane
    FS_REQUIRE((long long)(groups - 1) * p.g_a * 2 + ((long long)(p.M - 1) * p.ld_a + p.K) * 2 <= (long long)p.a_plane_bytes, "gemm_planes: the groups' rows exceed a plane of the row operand");
    FS_REQUIRE((long long)(groups - 1) * p.g_b * 2 + ((long long)(p.N - 1) * p.ld_b + p.K) * 2 <= (long long)p.b_plane_bytes, "gemm_planes: the groups' filters exceed a plane of the filter bank");
    const int var = bn >> 8;  // development builds (make DEV=1) only: experiment variants of the kernel
    bn &= 0xff;
#ifndef FS_DEV
    FS_REQUIRE(var == 0, "gemm_planes: bn must be 0, 64 or 128");
#endif
    if (bn != 64 && bn != 128) bn = p.N <= 64 ? 64 : 128;
    const int tm = cdiv(p.M, 256), tn = cdiv(p.N, bn);
    const dim3 grid(tm * tn * groups), block(512);
#ifdef FS_DEV
    if (var == 2 && bn == 128) { hipLaunchKernelGGL(gemm_planes_bf16x3_s16, grid, block, 0, s, p, tm, tn); FS_HIP(hipGetLastError()); return 0; }
#endif
    if (bn == 128) hipLaunchKernelGGL((gemm_planes_bf16x3<128>), grid, block, 0, s, p, tm, tn);
    else hipLaunchKernelGGL((gemm_planes_bf16x3<64>), grid, block, 0, s, p, tm, tn);
    FS_HIP(hipGetLastError());
    return 0;
}

// x -> its three bf16 terms, elementwise: planes[t][i], t = 0, 1, 2 (the row-operand twin of launch_split_bf16x3; producers that
// can afford it write the planes themselves)
}  // namespace fs
