// Epilogue of the matrix-core GEMM kernels (conv_igemm.hip, gemm_planes.hip): y = act(acc * scale + shift (+ residual)) for the 32x32
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
#endif


}  // namespace fs
