// Epilogue of the matrix-core GEMM kernel (conv_igemm.hip): y = act(acc * scale + shift (+ residual)) for the 32x32
// accumulator blocks of v_mfma_f32_32x32x*: col(n) = lane & 31, row(m) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include "kernels.h"

namespace fs {
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// erf for the GELU epilogue (nn.GELU, erf form: segm/model/blocks.py:16-28).  Branch-free: erf(|x|) = 1 - 2^(-|x| P(|x|)), P = the degree-10
// fit of -log2(erfc(x)) / x on [0, 4] (beyond 4, erfc < 2^-25: the result is 1 in fp32), 11 FMAs + one v_exp_f32 against the ~35
// instructions of the library erff, whose two argument ranges a wave executes one after the other.  Absolute error <= 1.0e-7 (the library's:
// 6e-8); in 0.5 v (1 + erf(v / sqrt 2)) that is below the rounding of the sum: measured against float64 the GELU's error is the same
// (tests/test_gpu_ops.py::test_gelu_epilogue_against_float64).  Round 5: the Segmenter's fc1 launches 41.6 -> 3x us, profiles/r05_experiments.txt 15.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x), 4.f);  // (a NaN becomes 4 here; the caller's product with v keeps it a NaN)
    float q = -1.434946029e-07f;
    q = fmaf(q, a, 3.633772167e-06f);
    q = fmaf(q, a, -4.095856275e-05f);
    q = fmaf(q, a, 2.688578097e-04f);
    q = fmaf(q, a, -1.106124604e-03f);
    q = fmaf(q, a, 2.616208512e-03f);
    q = fmaf(q, a, -3.566103114e-04f);
    q = fmaf(q, a, -2.759680524e-02f);
    q = fmaf(q, a, 1.482741833e-01f);
    q = fmaf(q, a, 9.184474349e-01f);
    q = fmaf(q, a, 1.627907038e+00f);
    return copysignf(1.f - __builtin_amdgcn_exp2f(-(q * a)), x);
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
                else if (ACT == 2) v = 0.5f * v * (1.f + gelu_erf(v * 0.70710678118654752f));  // nn.GELU (erf form)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo + (unsigned)((e & 3) + 8 * (e >> 2)) * row_o, 0, 0);
            }
        }
    }
}

// ---- the Segmenter's qkv Linear: K / V column tiles written as the attention's operand planes (ConvParams::kv_k)
typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ep_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ep_f32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> the three bf16 terms of each (x = h + m + l exactly, round-to-nearest residues), packed {x0 term, x1 term}
__device__ __forceinline__ void ep_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((ep_f32x2){x0, x1}, ep_bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((ep_f32x2){r0, r1}, ep_bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((ep_f32x2){q0, q1}, ep_bf16x2));
}

// kind 1: K tile, kind 2: V tile.  m_base = first row of this lane's rows (block base + 4 * hh), n_rel = this lane's first column relative to
// the start of K (resp. V) inside the qkv row.  Row r of the 32-row block is key kb + r of image p.kv_b; a lane holds rows
// (e & 3) + 8 * (e >> 2) + 4 * hh of one column (channel d of one head) per 32-column block.
//   K  [bh][key][d]:   rows e, e + 1 are consecutive keys; lanes d, d + 1 exchange one of them (DPP), so every lane stores dwords (two
//                      channels of one key) -- a store instruction covers four 64-B runs.
//   V^T [bh][d][pos]:  inside 16 keys the position is (k & 3) + 4 (k >> 3) + 8 ((k >> 2) & 1): a lane's registers 8 g .. 8 g + 7 ARE
//                      positions 8 hh .. 8 hh + 7 of key group 2 (kb / 32) + g in that order -- one 16-B store per plane and group.
template <int TN>
__device__ __forceinline__ void igemm_epilogue_kv(const f32x16 (&acc)[1][TN], const float (&sc)[TN], const float (&sh)[TN], const ConvParams& p, int kind,
                                                  int m_base, int n_rel, bool col_ok, int lane) {
    constexpr unsigned SENT = 0x80000000u;
    const int hh = lane >> 5, odd = lane & 1;
    const int N = p.kv_N, Npad = p.kv_Npad;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(kind == 1 ? p.kv_k : p.kv_vt), 0, 3u * p.kv_plane_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_rel + j * 32;
        const int bh = p.kv_b * p.kv_heads + (n >> 6), d = n & 63;
        if (kind == 1) {
#pragma unroll
            for (int ep = 0; ep < 8; ++ep) {
                const int e0 = 2 * ep, key0 = m_base + (e0 & 3) + 8 * (e0 >> 2);
                const float v0 = key0 < N ? acc[0][j][e0] * sc[j] + sh[j] : 0.f;
                const float v1 = key0 + 1 < N ? acc[0][j][e0 + 1] * sc[j] + sh[j] : 0.f;
                unsigned H, Mm, L;
                ep_split_pair(v0, v1, H, Mm, L);  // low halves: key0, high halves: key0 + 1, channel d
                // even lanes keep key0 and receive the neighbour's (channel d + 1); odd lanes keep key0 + 1 and receive channel d - 1's
                const unsigned send_hm = odd ? ((H & 0xffffu) | (Mm << 16)) : ((H >> 16) | (Mm & 0xffff0000u));
                const unsigned send_l = odd ? (L & 0xffffu) : (L >> 16);
                const unsigned recv_hm = (unsigned)__builtin_amdgcn_mov_dpp((int)send_hm, 0xB1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]
                const unsigned recv_l = (unsigned)__builtin_amdgcn_mov_dpp((int)send_l, 0xB1, 0xf, 0xf, true);
                const unsigned oH = odd ? ((recv_hm & 0xffffu) | (H & 0xffff0000u)) : ((H & 0xffffu) | (recv_hm << 16));
                const unsigned oM = odd ? ((recv_hm >> 16) | (Mm & 0xffff0000u)) : ((Mm & 0xffffu) | (recv_hm & 0xffff0000u));
                const unsigned oL = odd ? ((recv_l & 0xffffu) | (L & 0xffff0000u)) : ((L & 0xffffu) | (recv_l << 16));
                const int key = key0 + odd;
                const unsigned vo = (col_ok && key < Npad) ? (unsigned)(((bh * Npad + key) * 64 + (d & ~1)) * 2) : SENT;
                __builtin_amdgcn_raw_buffer_store_b32(oH, rsrc, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(oM, rsrc, vo, (int)p.kv_plane_bytes, 0);
                __builtin_amdgcn_raw_buffer_store_b32(oL, rsrc, vo, (int)(2u * p.kv_plane_bytes), 0);
            }
        } else {
            const int kb = m_base - 4 * hh;  // first key of the 32-row block (a multiple of 32)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                ep_u32x4 H, Mm, L;
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const int e0 = 8 * g + 2 * t2, key0 = m_base + (e0 & 3) + 8 * (e0 >> 2);
                    const float v0 = key0 < N ? acc[0][j][e0] * sc[j] + sh[j] : 0.f;
                    const float v1 = key0 + 1 < N ? acc[0][j][e0 + 1] * sc[j] + sh[j] : 0.f;
                    unsigned h, m, l;
                    ep_split_pair(v0, v1, h, m, l);
                    H[t2] = h; Mm[t2] = m; L[t2] = l;
                }
                const unsigned vo = (col_ok && kb < Npad) ? (unsigned)(((bh * 64 + d) * Npad + kb + 16 * g + 8 * hh) * 2) : SENT;
                __builtin_amdgcn_raw_buffer_store_b128(H, rsrc, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(Mm, rsrc, vo, (int)p.kv_plane_bytes, 0);
                __builtin_amdgcn_raw_buffer_store_b128(L, rsrc, vo, (int)(2u * p.kv_plane_bytes), 0);
            }
        }
    }
}
#endif


}  // namespace fs
