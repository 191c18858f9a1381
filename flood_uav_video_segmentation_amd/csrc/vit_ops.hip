// Segmenter (ViT encoder + mask-transformer decoder) kernels that are not plain nn.Linear:
// patchify, token assembly, LayerNorm, fp32-MFMA attention, mask head.
// Reference arithmetic: segm/model/vit.py:17-137, blocks.py:39-95, decoder.py:80-102, utils.py:65-76.
#include "kernels.h"

#include <type_traits>

namespace fs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ patchify (zero padded right/bottom)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ in, const float* __restrict__ in2, int B1,
                                                       float* __restrict__ out, int B, int H, int W, int P, int gh, int gw) {
    const int Kc = 3 * P * P;
    const int64_t total = (int64_t)B * gh * gw * Kc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % Kc);
        const int64_t row = i / Kc;
        const int px = col % P, py = (col / P) % P, c = col / (P * P);
        const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((int64_t)gw * gh));
        const int y = gy * P + py, x = gx * P + px;
        const float* src = b < B1 ? in + (size_t)b * 3 * H * W : in2 + (size_t)(b - B1) * 3 * H * W;
        out[i] = (y < H && x < W) ? src[((size_t)c * H + y) * W + x] : 0.f;
    }
}

// P % 4 == 0: a thread moves four consecutive pixels of a patch row (four scalar loads -- frame rows are not 16-B aligned -- one 16-B store)
// with 32-bit index arithmetic; the element-wise kernel above spent its time on 64-bit divisions (20 us for 2 x 12 MB).
__global__ __launch_bounds__(256) void patchify4_kernel(const float* __restrict__ in, const float* __restrict__ in2, int B1,
                                                        float* __restrict__ out, unsigned total4, int H, int W, int P, int gh, int gw) {
    const unsigned P4 = (unsigned)P / 4, Kc4 = 3u * P * P4;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const unsigned col4 = i % Kc4, row = i / Kc4;
        const unsigned px4 = col4 % P4, t = col4 / P4, py = t % (unsigned)P, c = t / (unsigned)P;
        const unsigned gx = row % (unsigned)gw, r2 = row / (unsigned)gw, gy = r2 % (unsigned)gh, b = r2 / (unsigned)gh;
        const int y = (int)(gy * P + py), x = (int)(gx * P + 4 * px4);
        const float* src = ((int)b < B1 ? in + (size_t)b * 3 * H * W : in2 + (size_t)(b - B1) * 3 * H * W) + (size_t)c * H * W;
        const int yc = min(y, H - 1);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float q = src[(size_t)yc * W + min(x + e, W - 1)];  // clamped: always a valid address; zero padding by the select
            v[e] = (y < H && x + e < W) ? q : 0.f;
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

int launch_patchify(const float* in, const float* in2, int B1, float* out, int B, int H, int W, int P, int gh, int gw, hipStream_t s) {
    FS_REQUIRE(B1 >= 0 && B1 <= B && (B1 == B || in2) && (B1 == 0 || in), "patchify: bad frame split B1=%d of B=%d", B1, B);
    const int64_t total = (int64_t)B * gh * gw * 3 * P * P;
    if (P % 4 == 0 && total / 4 < ((int64_t)1 << 31) && ((uintptr_t)out & 15) == 0) {
        hipLaunchKernelGGL(patchify4_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total / 4, 256), 16384)), dim3(256), 0, s, in, in2, B1, out,
                           (unsigned)(total / 4), H, W, P, gh, gw);
        FS_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, in, in2, B1,
                       out, B, H, W, P, gh, gw);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ token assembly
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ X, int B, int N, int D4) {
    const int64_t total = (int64_t)B * (N + 1) * D4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d4 = (int)(i % D4);
        const int t = (int)((i / D4) % (N + 1));
        const int b = (int)(i / ((int64_t)D4 * (N + 1)));
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos)[(size_t)t * D4 + d4];
        const f32x4 v = t == 0 ? reinterpret_cast<const f32x4*>(cls)[d4]
                               : reinterpret_cast<const f32x4*>(emb)[((size_t)b * N + t - 1) * D4 + d4];
        reinterpret_cast<f32x4*>(X)[i] = v + pe;
    }
}

int launch_vit_assemble(const float* emb, const float* cls, const float* pos, float* X, int B, int N, int D, hipStream_t s) {
    FS_REQUIRE(D % 4 == 0, "vit_assemble: D must be a multiple of 4");
    const int64_t total = (int64_t)B * (N + 1) * (D / 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, emb, cls, pos,
                       X, B, N, D / 4);
    FS_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void dec_assemble_kernel(const float* __restrict__ Y, const float* __restrict__ cls_emb,
                                                           float* __restrict__ Z, int B, int N, int K, int D4) {
    const int64_t total = (int64_t)B * (N + K) * D4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d4 = (int)(i % D4);
        const int t = (int)((i / D4) % (N + K));
        const int b = (int)(i / ((int64_t)D4 * (N + K)));
        reinterpret_cast<f32x4*>(Z)[i] = t < N ? reinterpret_cast<const f32x4*>(Y)[((size_t)b * N + t) * D4 + d4]
                                               : reinterpret_cast<const f32x4*>(cls_emb)[(size_t)(t - N) * D4 + d4];
    }
}

int launch_dec_assemble(const float* Y, const float* cls_emb, float* Z, int B, int N, int K, int D, hipStream_t s) {
    FS_REQUIRE(D % 4 == 0, "dec_assemble: D must be a multiple of 4");
    const int64_t total = (int64_t)B * (N + K) * (D / 4);
    hipLaunchKernelGGL(dec_assemble_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, Y, cls_emb, Z,
                       B, N, K, D / 4);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ LayerNorm: one wave per row, D <= 1024
// NI = chunks of 64 float4 a row has.  Everything a row needs -- its values, gamma, beta -- is requested before anything is used
// (round 6: the gamma / beta loads sat behind the two reductions, a second memory round trip in a kernel that is one round trip
// long); lanes past the row's end load from clamped addresses and contribute exact zeros, as before.
template <int NI>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out, int rows, int D4,
                                                        int rows_per_batch, int drop_first) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int orow = row;
    if (drop_first) {
        const int b = row / rows_per_batch, t = row - b * rows_per_batch;
        if (t == 0) return;  // wave-uniform
        orow = b * (rows_per_batch - 1) + t - 1;
    }
    const f32x4* x = reinterpret_cast<const f32x4*>(in) + (size_t)row * D4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[NI], g[NI], bt[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int cc = min(lane + 64 * i, D4 - 1);
        v[i] = x[cc];
        g[i] = reinterpret_cast<const f32x4*>(gamma)[cc];
        bt[i] = reinterpret_cast<const f32x4*>(beta)[cc];
    }
    // an empty asm that "reads" gamma and beta: the compiler otherwise sinks their loads into the conditional store at the end; the
    // scheduling barrier keeps all nine loads in front of the first use of any of them
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(g[i]), "v"(bt[i]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        v[i] = lane + 64 * i < D4 ? v[i] : z;
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
#pragma unroll
    for (int i = NI; i < 4; ++i) sum += 0.f;  // four terms whatever NI is (x + 0.0 is not x for x = -0.0): the same float in every instantiation
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)(D4 * 4);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {  // element by element into ONE running sum, valid chunks only (as the round-1 kernel did)
        const bool ok = lane + 64 * i < D4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq = ok ? sq + d * d : sq;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = 1.f / sqrtf(sq / (float)(D4 * 4) + 1e-5f);
    f32x4* y = reinterpret_cast<f32x4*>(out) + (size_t)orow * D4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (v[i][e] - mean) * rstd * g[i][e] + bt[i][e];
        if (c < D4) y[c] = r;
    }
}

int launch_layernorm(const float* in, const float* gamma, const float* beta, float* out, int rows, int D, int rows_per_batch,
                     int drop_first, hipStream_t s) {
    FS_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm: D=%d must be a multiple of 4 and <= 1024", D);
    const dim3 grid(cdiv(rows, 4)), block(256);
    switch (cdiv(D / 4, 64)) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, in, gamma, beta, out, rows, D / 4, rows_per_batch, drop_first); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, in, gamma, beta, out, rows, D / 4, rows_per_batch, drop_first); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, in, gamma, beta, out, rows, D / 4, rows_per_batch, drop_first); break;
        default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, in, gamma, beta, out, rows, D / 4, rows_per_batch, drop_first); break;
    }
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ attention, fp32 MFMA, head_dim 64
// Workgroup = 128 queries of one (batch, head); wave w owns queries 32w..32w+31, one per lane (lane&31).
// Per 32-key block:   S^T = K * Q^T          A = K rows (lane i = key),  B = Q^T (lane j = query)   32 MFMAs
//                     online softmax          every lane owns ONE query: max/sum are in-register + 1 shuffle
//                     O^T += V^T * P^T        B operand = the S^T accumulator registers as they stand   32 MFMAs
// (the 32x32 accumulator has its column on the lane and rows (r&3)+8(r>>2)+4h in registers: fed back as
//  the B operand, step r consumes key (r&3)+8(r>>2) from lane-half 0 and that key + 4 from lane-half 1,
//  and the A operand reads V at exactly those two keys.)
// K tile rows are padded to 68 floats: ds_read_b128 of a 16-lane group then hits 16 distinct 4-bank slots.
#ifndef FS_ATT_EXP
#define FS_ATT_EXP 0  // tools/probe_attention.hip only: elimination experiments (1 no exp, 2 no PV MFMAs, 3 no K/V re-staging, 4 no S MFMAs)
#endif
constexpr int ATT_DH = 64;
constexpr int ATT_KT = 64;   // keys per LDS tile
constexpr int ATT_LDK = 68;
#ifndef ATT_USE_DMA
#define ATT_USE_DMA 1  // 0: the register-staged kernel (A/B builds)
#endif
constexpr int ATT_NW = 4;    // waves per workgroup = 128 queries per staged K/V tile.  Measured on ViT-S/16 (N = 1937): 2 waves
                             // (64 queries, 4 workgroups/CU) 36 TFLOP/s -- the K/V staging per query doubles; 4 waves 52+

// Key split (flash-decoding style): ViT-S/16 at 713x713 has 16 query tiles x 6 heads x 2 frames = 192 workgroups for 256
// CUs, one wave per SIMD, so the softmax VALU work of a wave is never hidden behind another wave's MFMAs.  With
// nsplit > 1 workgroup (query tile, split) walks only its share of the key tiles and stores the UNNORMALISED O^T plus
// (running max, running sum) per query; attention_combine_kernel merges the splits.  nsplit == 1 writes `out` directly.
// __launch_bounds__(.., 3): three waves per SIMD (the compiler fits 154 VGPRs instead of 210, no spills), so that the three
// workgroups the key split aims at per CU are really co-resident: 96.8 -> 101.9 TFLOP/s on ViT-S/16 (profiles/r02_experiments.txt).
template <bool SPLIT>
__global__ __launch_bounds__(64 * ATT_NW, 3) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                     float* __restrict__ part_o, float* __restrict__ part_ml, int N,
                                                                     int heads, float scale, int nsplit) {
    __shared__ __attribute__((aligned(16))) float Ks[ATT_KT * ATT_LDK];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_KT * ATT_DH];
    constexpr int NT = 64 * ATT_NW;
    constexpr int LPT = ATT_KT * 16 / NT;  // float4 of K (and of V) staged per thread
    const int D = heads * ATT_DH, ld = 3 * D;
    const int b = blockIdx.z, head = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int qtiles = SPLIT ? gridDim.x / nsplit : gridDim.x;
    const int qt = SPLIT ? blockIdx.x % qtiles : blockIdx.x, split = SPLIT ? blockIdx.x / qtiles : 0;
    const int q = qt * (32 * ATT_NW) + wv * 32 + l31;
    const int qc = min(q, N - 1);
    const float* base = qkv + (size_t)b * N * ld + head * ATT_DH;

    // this lane's query row, dims 32*hh .. 32*hh+31, pre-scaled by scale * log2(e): the softmax then runs on v_exp_f32
    // (2^x, 1 ulp) directly -- exp(s - m) == 2^(s' - m') with s' = s * log2(e)
    const float scale2 = scale * 1.44269504088896340736f;
    float qreg[32];
    {
        const f32x4* qp = reinterpret_cast<const f32x4*>(base + (size_t)qc * ld + 32 * hh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 v = qp[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) qreg[4 * u + e] = v[e] * scale2;
        }
    }
    f32x16 acc_o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles_all = (N + ATT_KT - 1) / ATT_KT;
    const int kt0 = SPLIT ? (ntiles_all * split) / nsplit : 0;             // launcher: nsplit <= ntiles_all, so never empty
    const int ntiles = SPLIT ? (ntiles_all * (split + 1)) / nsplit : ntiles_all;
    f32x4 kreg[LPT], vreg[LPT];  // next K/V tile, in flight while the current one is multiplied
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int idx = t + NT * j;
            const int row = idx >> 4, c4 = idx & 15;
            const int key = min(kt * ATT_KT + row, N - 1);  // clamped; rows beyond N are zeroed when stored
            const float* rp = base + (size_t)key * ld + c4 * 4;
            kreg[j] = *reinterpret_cast<const f32x4*>(rp + D);
            vreg[j] = *reinterpret_cast<const f32x4*>(rp + 2 * D);
        }
    };
    fetch(kt0);
    for (int kt = kt0; kt < ntiles; ++kt) {
        __syncthreads();  // previous tile fully consumed
        if (FS_ATT_EXP != 3 || kt == kt0)
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int idx = t + NT * j;
            const int row = idx >> 4, c4 = idx & 15;
            const bool ok = kt * ATT_KT + row < N;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&Ks[row * ATT_LDK + c4 * 4]) = ok ? kreg[j] : z;
            *reinterpret_cast<f32x4*>(&Vs[row * ATT_DH + c4 * 4]) = ok ? vreg[j] : z;
        }
        __syncthreads();
        if (kt + 1 < ntiles && FS_ATT_EXP != 3) fetch(kt + 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key0 = kt * ATT_KT + kb * 32;
            if (key0 >= N) break;  // block-uniform
            f32x16 sT;
#pragma unroll
            for (int e = 0; e < 16; ++e) sT[e] = 0.f;
            const float* krow = &Ks[(kb * 32 + l31) * ATT_LDK + 32 * hh];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 4 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FS_ATT_EXP != 4 || (u == 0 && e == 0)) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[4 * u + e], sT, 0, 0, 0);
            }
            // mask keys beyond N, running max over this lane's 16 keys and the other half's 16
            if (key0 + 32 > N) {  // block-uniform: only the last block has keys to mask
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sT[r] = -INFINITY;
            }
            float mloc = sT[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sT[r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);                        // finite: key0 < N guarantees one valid key
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 2^(-inf) = 0 on the first block
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sT[r] = FS_ATT_EXP == 1 ? sT[r] - m_new : __builtin_amdgcn_exp2f(sT[r] - m_new);
                lsum += sT[r];
            }
            l_run = l_run * alpha + lsum;
            m_run = m_new;
            if (__any(alpha != 1.f)) {  // the running max settles after a few blocks: skip the 32 rescaling multiplies then
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc_o[i][e] *= alpha;
            }
            // O^T += V^T P^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow_r = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v0 = Vs[krow_r * ATT_DH + l31];
                const float v1 = Vs[krow_r * ATT_DH + 32 + l31];
                if (FS_ATT_EXP == 2 && r > 0) continue;
                acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sT[r], acc_o[0], 0, 0, 0);
                acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sT[r], acc_o[1], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        if (q < N) {
            const size_t row = ((size_t)(b * heads + head) * nsplit + split) * N + q;
            float* op = part_o + row * ATT_DH;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e];
                    *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
                }
            if (hh == 0) {
                part_ml[2 * row] = m_run;
                part_ml[2 * row + 1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.f / l_tot;
    if (q < N) {
        float* op = out + ((size_t)b * N + q) * D + head * ATT_DH;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e] * inv;
                *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
            }
    }
}

// ------------------------------------------------------------------ the same kernel with K / V staged global -> LDS DIRECTLY
// (round 3).  The register-staged form above spends, per 64 keys and thread, 8 global loads into 32 VGPRs, 8 ds_write_b128 and
// two barriers on moving K and V (measured: ~7 % of the launch, tools/probe_attention.hip).  Here a stage is 32 keys, K and V
// of the NEXT stage are requested with buffer_load_dwordx4 ... lds (four 1-KiB pieces per wave, no VGPRs) right behind the barrier
// that publishes the current one, and land under its 64 MFMAs; two LDS stage buffers, one barrier per 32 keys.  A piece is
// lane-linear in LDS (4 rows x 256 B), so rows cannot be padded: the K image is XOR-swizzled instead -- 16-B chunk j of key row r
// lives at chunk j ^ (r & 15), applied to the per-lane SOURCE offset -- which keeps the fragment reads (ds_read_b128 of one
// column chunk across 16 different rows) conflict-free; V is read along rows (32 consecutive floats) and stays linear.  Keys
// beyond N are zero-filled by the descriptor's range check (sentinel offset) and masked to -inf as before.  Same arithmetic,
// same order: bit-identical to the register-staged kernel.  Measured (ViT-S/16, B = 2, 14 launches): 1.7225 ms against 1.7297 ms --
// no gain: with three workgroups per CU the staging was already hidden behind the neighbours' MFMAs; what it frees is 32 VGPRs
// (122 instead of 154).  A fourth workgroup per CU with a 5-way key split (960 workgroups) was measured too: 1.85 ms, worse (more
// partials to merge, same pipe).  Kept as the shipped form (ATT_USE_DMA 0 builds the register-staged one for A/B).
template <bool SPLIT>
__global__ __launch_bounds__(64 * ATT_NW, 3) void attention_dma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                     float* __restrict__ part_o, float* __restrict__ part_ml, int N,
                                                                     int heads, float scale, int nsplit) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ST = 32;  // keys per stage
    __shared__ __attribute__((aligned(1024))) float Ks[2][ST * ATT_DH];
    __shared__ __attribute__((aligned(1024))) float Vs[2][ST * ATT_DH];
    const int D = heads * ATT_DH, ld = 3 * D;
    const int b = blockIdx.z, head = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int qtiles = SPLIT ? gridDim.x / nsplit : gridDim.x;
    const int qt = SPLIT ? blockIdx.x % qtiles : blockIdx.x, split = SPLIT ? blockIdx.x / qtiles : 0;
    const int q = qt * (32 * ATT_NW) + wv * 32 + l31;
    const int qc = min(q, N - 1);
    const float* base = qkv + (size_t)b * N * ld + head * ATT_DH;

    const float scale2 = scale * 1.44269504088896340736f;
    float qreg[32];
    {
        const f32x4* qp = reinterpret_cast<const f32x4*>(base + (size_t)qc * ld + 32 * hh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 v = qp[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) qreg[4 * u + e] = v[e] * scale2;
        }
    }
    f32x16 acc_o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // this workgroup's keys: the same 64-key tile ranges as the register-staged kernel, walked 32 keys at a time
    const int ntiles_all = (N + ATT_KT - 1) / ATT_KT;
    const int kt0 = SPLIT ? (ntiles_all * split) / nsplit : 0;
    const int kt1 = SPLIT ? (ntiles_all * (split + 1)) / nsplit : ntiles_all;
    const int s0 = 2 * kt0, s1 = min(2 * kt1, (N + ST - 1) / ST);

    // DMA pieces of this wave: rows 8 wv + 4 j + (lane >> 4), j = 0, 1; physical chunk lane & 15
    constexpr unsigned SENT = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(((long long)(N - 1) * ld + 3 * D - head * ATT_DH) * 4), 0x00020000);
    const int prow = lane >> 4, pch = lane & 15;
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * wv + 4 * j + prow;
            const int key = stage * ST + row;
            const unsigned ro = key < N ? (unsigned)(key * ld) * 4u : SENT;
            const unsigned kvo = ro == SENT ? SENT : ro + (unsigned)((D + 4 * (pch ^ (row & 15))) * 4);
            const unsigned vvo = ro == SENT ? SENT : ro + (unsigned)((2 * D + 4 * pch) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(&Ks[buf][(8 * wv + 4 * j) * ATT_DH]), 16, kvo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(&Vs[buf][(8 * wv + 4 * j) * ATT_DH]), 16, vvo, 0, 0, 0);
        }
    };
    const int xk = (l31 & 15) ^ (8 * hh);  // physical chunk of logical chunk 8 hh + u in row l31 = xk ^ u

    if (s0 < s1) issue(s0, 0);
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        // this wave's pieces of stage s have landed (vmcnt), everybody's have (barrier), and everybody is done reading the other buffer
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        if (s + 1 < s1) issue(s + 1, buf ^ 1);
        const int key0 = s * ST;
        f32x16 sT;
#pragma unroll
        for (int e = 0; e < 16; ++e) sT[e] = 0.f;
        const float* krow = &Ks[buf][l31 * ATT_DH];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + 4 * (xk ^ u));
#pragma unroll
            for (int e = 0; e < 4; ++e) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[4 * u + e], sT, 0, 0, 0);
        }
        if (key0 + 32 > N) {  // block-uniform: only the last stage has keys to mask
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sT[r] = -INFINITY;
        }
        float mloc = sT[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sT[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sT[r] = __builtin_amdgcn_exp2f(sT[r] - m_new);
            lsum += sT[r];
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[i][e] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow_r = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float v0 = Vs[buf][krow_r * ATT_DH + l31];
            const float v1 = Vs[buf][krow_r * ATT_DH + 32 + l31];
            acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sT[r], acc_o[0], 0, 0, 0);
            acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sT[r], acc_o[1], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        if (q < N) {
            const size_t row = ((size_t)(b * heads + head) * nsplit + split) * N + q;
            float* op = part_o + row * ATT_DH;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e];
                    *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
                }
            if (hh == 0) {
                part_ml[2 * row] = m_run;
                part_ml[2 * row + 1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.f / l_tot;
    if (q < N) {
        float* op = out + ((size_t)b * N + q) * D + head * ATT_DH;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e] * inv;
                *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
            }
    }
#endif
}

// ------------------------------------------------------------------ attention on the bf16 matrix cores with SPLIT operands
// (round 3, the default route; same idea as the split-operand conv kernel, conv_igemm.hip / DESIGN.md 3.1b): every fp32 value of
// Q, K, V and of the probabilities P is the exact sum of three bf16 terms; the six cross products of order <= 2^-16 run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the three dropped ones are <= 2^-23 of a product.  Softmax stays in fp32.
//   * a pre-pass (attention_split_kv_kernel) writes K as three planes [bh][key][64] and V TRANSPOSED as three planes
//     [bh][64][key] (bf16; keys padded with zeros to a multiple of 32; inside every 16 keys the order is 0-3, 8-11, 4-7, 12-15,
//     which is the order in which a lane of the S^T accumulator holds its eight keys of a 16-key step -- so P needs no shuffle);
//   * a stage is 32 keys: 12 KB of K planes + 12 KB of V planes, DMA'd straight into LDS (two stage buffers, one barrier per
//     stage, 3 workgroups per CU as before); K rows are 128 B (chunk ^ (row >> 1) & 7), V^T rows 64 B (piece ^ (row >> 2) & 3):
//     both fragment reads are one conflict-free ds_read_b128 per plane;
//   * S^T = K Q^T: 4 steps of 16 d x 6 terms; O^T += V^T P^T: 2 d-blocks x 2 steps of 16 keys x 6 terms: 48 MFMAs of 32 cycles per
//     32 keys instead of 64 of 64 cycles; Q is split once per workgroup, P once per stage (72 VALU instructions).
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
typedef float af32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void att_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){x0, x1}, abf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){r0, r1}, abf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((af32x2){q0, q1}, abf16x2));
}

// thread = (bh, key, 8 consecutive d); Kp / Vtp: [3][BH][Npad][64] and [3][BH][64][Npad] bf16
__global__ __launch_bounds__(256) void attention_split_kv_kernel(const float* __restrict__ qkv, unsigned short* __restrict__ Kp,
                                                                 unsigned short* __restrict__ Vtp, int B, int N, int Npad, int heads) {
    const int64_t total = (int64_t)B * heads * Npad * 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int dg = (int)(i & 7);
    const int key = (int)((i >> 3) % Npad);
    const int bh = (int)((i >> 3) / Npad);
    const int b = bh / heads, head = bh - b * heads;
    const int D = heads * ATT_DH;
    const size_t plane = (size_t)B * heads * Npad * ATT_DH;
    float k[8], v[8];
    if (key < N) {
        const float* src = qkv + ((size_t)b * N + key) * 3 * D + head * ATT_DH + 8 * dg;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(src + D), k1 = *reinterpret_cast<const f32x4*>(src + D + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + 2 * D), v1 = *reinterpret_cast<const f32x4*>(src + 2 * D + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] = k0[e]; k[4 + e] = k1[e]; v[e] = v0[e]; v[4 + e] = v1[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = v[e] = 0.f;
    }
    au32x4 kh, km, kl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned h, m, l;
        att_split_pair(k[2 * e], k[2 * e + 1], h, m, l);
        kh[e] = h; km[e] = m; kl[e] = l;
    }
    const size_t ko = ((size_t)bh * Npad + key) * ATT_DH + 8 * dg;
    *reinterpret_cast<au32x4*>(Kp + ko) = kh;
    *reinterpret_cast<au32x4*>(Kp + plane + ko) = km;
    *reinterpret_cast<au32x4*>(Kp + 2 * plane + ko) = kl;
    // V transposed; position of key k inside its group of 16: (k & 3) + 4 (k >> 3) + 8 ((k >> 2) & 1)
    const int kk = key & 15, pos = (key & ~15) + (kk & 3) + 4 * (kk >> 3) + 8 * ((kk >> 2) & 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e];
        const __bf16 hb = (__bf16)x;
        const float r = x - (float)hb;
        const __bf16 mb = (__bf16)r;
        const __bf16 lb = (__bf16)(r - (float)mb);
        const size_t vo = ((size_t)bh * ATT_DH + 8 * dg + e) * Npad + pos;
        Vtp[vo] = __builtin_bit_cast(unsigned short, hb);
        Vtp[plane + vo] = __builtin_bit_cast(unsigned short, mb);
        Vtp[2 * plane + vo] = __builtin_bit_cast(unsigned short, lb);
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(64 * ATT_NW, 3) void attention_bf16x3_kernel(const float* __restrict__ qkv, const unsigned short* __restrict__ Kp,
                                                                        const unsigned short* __restrict__ Vtp, float* __restrict__ out,
                                                                        float* __restrict__ part_o, float* __restrict__ part_ml, int N, int Npad,
                                                                        int heads, float scale, int nsplit, unsigned plane_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ST = 32;     // keys per stage
    constexpr int PL = 1024;   // floats of LDS per plane and stage (4 KB: 32 keys x 128 B, or 64 d-rows x 64 B)
    __shared__ __attribute__((aligned(1024))) float Ks[2][3 * PL];
    __shared__ __attribute__((aligned(1024))) float Vs[2][3 * PL];
    const int D = heads * ATT_DH, ld = 3 * D;
    const int b = blockIdx.z, head = blockIdx.y, bh = b * heads + head;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int qtiles = SPLIT ? gridDim.x / nsplit : gridDim.x;
    const int qt = SPLIT ? blockIdx.x % qtiles : blockIdx.x, split = SPLIT ? blockIdx.x / qtiles : 0;
    const int q = qt * (32 * ATT_NW) + wv * 32 + l31;
    const int qc = min(q, N - 1);
    const float scale2 = scale * 1.44269504088896340736f;
    f32x16 acc_o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles_all = (N + ATT_KT - 1) / ATT_KT;  // the same key ranges per split as the fp32 kernels
    const int kt0 = SPLIT ? (ntiles_all * split) / nsplit : 0;
    const int kt1 = SPLIT ? (ntiles_all * (split + 1)) / nsplit : ntiles_all;
    const int s0 = 2 * kt0, s1 = min(2 * kt1, (N + ST - 1) / ST);

    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, 3u * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vtp, 0, 3u * plane_bytes, 0x00020000);
    // DMA pieces of this wave, per plane: K rows 8 wv + (lane >> 3), 16-B slot lane & 7; V^T rows 16 wv + (lane >> 2), slot lane & 3
    const int krow = 8 * wv + (lane >> 3), vrow = 16 * wv + (lane >> 2);
    const unsigned k_voff = (unsigned)((((size_t)bh * Npad + krow) * ATT_DH + 8 * ((lane & 7) ^ ((krow >> 1) & 7))) * 2);
    const unsigned v_voff = (unsigned)((((size_t)bh * ATT_DH + vrow) * Npad + 8 * ((lane & 3) ^ ((vrow >> 2) & 3))) * 2);
    auto issue = [&](int stage, int buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(&Ks[buf][pl * PL + wv * 256]), 16, k_voff,
                                                     (unsigned)pl * plane_bytes + (unsigned)(stage * ST * ATT_DH * 2), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(&Vs[buf][pl * PL + wv * 256]), 16, v_voff,
                                                     (unsigned)pl * plane_bytes + (unsigned)(stage * ST * 2), 0, 0);
        }
    };
    const int kx = (l31 >> 1) & 7, vx = (l31 >> 2) & 3;
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};  // (row-operand term, column-operand term): h l, l h, m m, h m, m h, h h

    if (s0 < s1) issue(s0, 0);  // requested BEFORE this lane's query is fetched and split (round 5): the two latencies overlap
    // Q of this lane's query: d = 16 ks + 8 hh .. + 7 for the four steps, scaled, split once
    abf16x8 Qp[4][3];
    {
        const float* qp = qkv + ((size_t)b * N + qc) * ld + head * ATT_DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * ks), c = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
            au32x4 h, m, l;
            const float x[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hu, mu, lu;
                att_split_pair(x[2 * e] * scale2, x[2 * e + 1] * scale2, hu, mu, lu);
                h[e] = hu; m[e] = mu; l[e] = lu;
            }
            Qp[ks][0] = __builtin_bit_cast(abf16x8, h);
            Qp[ks][1] = __builtin_bit_cast(abf16x8, m);
            Qp[ks][2] = __builtin_bit_cast(abf16x8, l);
        }
    }
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        __builtin_amdgcn_s_waitcnt(0x0070);  // this wave's pieces of stage s have landed; the barrier: everybody's have, and nobody reads the other buffer any more
        __syncthreads();
        if (s + 1 < s1) issue(s + 1, buf ^ 1);
        const int key0 = s * ST;
        f32x16 sT;
#pragma unroll
        for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            abf16x8 kf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                kf[pl] = __builtin_bit_cast(abf16x8, *reinterpret_cast<const au32x4*>(&Ks[buf][pl * PL + l31 * 32 + 4 * ((2 * ks + hh) ^ kx)]));
#pragma unroll
            for (int tm = 0; tm < 6; ++tm) sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[TA[tm]], Qp[ks][TB[tm]], sT, 0, 0, 0);
        }
        if (key0 + 32 > N) {  // block-uniform: only the last stage has keys to mask
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sT[r] = -INFINITY;
        }
        float mloc = sT[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sT[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sT[r] = __builtin_amdgcn_exp2f(sT[r] - m_new);
            lsum += sT[r];
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[i][e] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // this lane's eight keys of the step are accumulator registers 8 ks .. 8 ks + 7 (keys 16 ks + 4 hh + 0-3 and + 8-11)
            au32x4 ph, pm, pl3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hu, mu, lu;
                att_split_pair(sT[8 * ks + 2 * e], sT[8 * ks + 2 * e + 1], hu, mu, lu);
                ph[e] = hu; pm[e] = mu; pl3[e] = lu;
            }
            const abf16x8 pp[3] = {__builtin_bit_cast(abf16x8, ph), __builtin_bit_cast(abf16x8, pm), __builtin_bit_cast(abf16x8, pl3)};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                abf16x8 vf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    vf[pl] = __builtin_bit_cast(abf16x8, *reinterpret_cast<const au32x4*>(&Vs[buf][pl * PL + (i * 32 + l31) * 16 + 4 * ((2 * ks + hh) ^ vx)]));
#pragma unroll
                for (int tm = 0; tm < 6; ++tm) acc_o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[TA[tm]], pp[TB[tm]], acc_o[i], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        if (q < N) {
            const size_t row = ((size_t)(b * heads + head) * nsplit + split) * N + q;
            float* op = part_o + row * ATT_DH;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e];
                    *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
                }
            if (hh == 0) {
                part_ml[2 * row] = m_run;
                part_ml[2 * row + 1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.f / l_tot;
    if (q < N) {
        float* op = out + ((size_t)b * N + q) * D + head * ATT_DH;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc_o[i][4 * g + e] * inv;
                *reinterpret_cast<f32x4*>(op + i * 32 + 8 * g + 4 * hh) = v;
            }
    }
#endif
}

// merge the key splits of one query: O = sum_s e^(m_s - M) O_s / sum_s e^(m_s - M) l_s ; thread = (query row, float4 of dh)
// NS > 0: the split count at compile time -- every (max, sum) pair and partial row is requested before the first is used (round 6: the maxima
// were one pass of loads, the weighted sum a second one with each split's loads behind the previous split's arithmetic).  Same operations in
// the same order as the generic form (NS == 0).
typedef float f32x2c __attribute__((ext_vector_type(2)));
template <int NS>
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                float* __restrict__ out, int B, int N, int heads, int nsplit) {
    const int64_t total = (int64_t)B * heads * N * 16;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i & 15);
    const int64_t r = i >> 4;
    const int q = (int)(r % N);
    const int bh = (int)(r / N);
    const int b = bh / heads, head = bh - b * heads;
    float L = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (NS > 0) {
        f32x2c ml[NS > 0 ? NS : 1];
        f32x4 v[NS > 0 ? NS : 1];
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            const size_t row = ((size_t)bh * NS + sp) * N + q;
            ml[sp] = *reinterpret_cast<const f32x2c*>(part_ml + 2 * row);
            v[sp] = *reinterpret_cast<const f32x4*>(part_o + row * ATT_DH + c4 * 4);
        }
        float M = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) M = fmaxf(M, ml[sp][0]);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            const float w = __builtin_amdgcn_exp2f(ml[sp][0] - M);   // the partial maxima are in log2 units (see the kernel)
            L += w * ml[sp][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += w * v[sp][e];
        }
    } else {
        float M = -INFINITY;
        for (int sp = 0; sp < nsplit; ++sp) M = fmaxf(M, part_ml[2 * (((size_t)bh * nsplit + sp) * N + q)]);
        for (int sp = 0; sp < nsplit; ++sp) {
            const size_t row = ((size_t)bh * nsplit + sp) * N + q;
            const float w = __builtin_amdgcn_exp2f(part_ml[2 * row] - M);
            L += w * part_ml[2 * row + 1];
            const f32x4 v = *reinterpret_cast<const f32x4*>(part_o + row * ATT_DH + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += w * v[e];
        }
    }
    const float inv = 1.f / L;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= inv;
    *reinterpret_cast<f32x4*>(out + ((size_t)b * N + q) * (heads * ATT_DH) + head * ATT_DH + c4 * 4) = o;
}

static void launch_attention_combine(const float* part_o, const float* part_ml, float* out, int B, int N, int heads, int ns, hipStream_t s) {
    const int64_t tot = (int64_t)B * heads * N * 16;
    const dim3 grid((unsigned)cdiv64(tot, 256)), block(256);
    switch (ns) {
        case 2: hipLaunchKernelGGL(attention_combine_kernel<2>, grid, block, 0, s, part_o, part_ml, out, B, N, heads, ns); break;
        case 3: hipLaunchKernelGGL(attention_combine_kernel<3>, grid, block, 0, s, part_o, part_ml, out, B, N, heads, ns); break;
        case 4: hipLaunchKernelGGL(attention_combine_kernel<4>, grid, block, 0, s, part_o, part_ml, out, B, N, heads, ns); break;
        default: hipLaunchKernelGGL(attention_combine_kernel<0>, grid, block, 0, s, part_o, part_ml, out, B, N, heads, ns); break;
    }
}

int attention_splits(int /*B*/, int N, int heads) {
    // aim at ~3 workgroups per CU (256 CUs) for the usual batch of two key frames: their waves share a SIMD, so one's softmax
    // overlaps another's MFMAs.  Decided on ONE image's work, never on the batch: the merge order of the key splits is part of
    // the result, and a frame's output must not depend on the batch it is computed in (the key-frame cache relies on it).
    const int units = cdiv(N, 32 * ATT_NW) * heads;
    const int ntiles = cdiv(N, ATT_KT);
    int ns = (384 + units / 2) / units;
    ns = std::min(ns, std::min(4, ntiles / 2));
    return std::max(ns, 1);
}

size_t attention_scratch_floats(int B, int N, int heads) {
    const int ns = attention_splits(B, N, heads);
    return ns > 1 ? (size_t)B * heads * ns * N * (ATT_DH + 2) : 0;
}

// floats of workspace the split-operand route needs for the K / V^T planes (two tensors x three bf16 planes, keys padded to 32)
size_t attention_split_floats(int B, int N, int heads) { return (size_t)2 * 3 * B * heads * ((N + 31) / 32 * 32) * (ATT_DH / 2); }

// softmax(Q K^T * scale) V with split operands (see attention_bf16x3_kernel).  planes: attention_split_floats(B, N, heads) floats;
// scratch: as launch_attention_f32 (same key splits, same merge).  planes_ready: the K / V^T planes were already written by the
// producer of qkv (the qkv Linear's epilogue, conv_igemm.hip) -- the pre-pass is skipped.
int launch_attention_split(const float* qkv, float* out, int B, int N, int heads, float scale, float* scratch, float* planes, hipStream_t s, bool planes_ready) {
    FS_REQUIRE(B >= 1 && N >= 1 && heads >= 1 && planes, "attention: bad shape");
    const int Npad = (N + 31) / 32 * 32;
    const size_t plane_elems = (size_t)B * heads * Npad * ATT_DH;
    FS_REQUIRE(plane_elems * 2 * 3 < ((size_t)1 << 31) && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)planes & 15) == 0, "attention: K / V planes too large or unaligned");
    unsigned short* Kp = reinterpret_cast<unsigned short*>(planes);
    unsigned short* Vtp = Kp + 3 * plane_elems;
    if (!planes_ready) {
        const int64_t total = (int64_t)B * heads * Npad * 8;
        hipLaunchKernelGGL(attention_split_kv_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, qkv, Kp, Vtp, B, N, Npad, heads);
        FS_HIP(hipGetLastError());
    }
    const int qtiles = cdiv(N, 32 * ATT_NW);
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const int ns = scratch ? attention_splits(B, N, heads) : 1;
    if (ns == 1) {
        hipLaunchKernelGGL(attention_bf16x3_kernel<false>, dim3(qtiles, heads, B), dim3(64 * ATT_NW), 0, s, qkv, Kp, Vtp, out, nullptr, nullptr, N, Npad,
                           heads, scale, 1, plane_bytes);
        FS_HIP(hipGetLastError());
        return 0;
    }
    float* part_o = scratch;
    float* part_ml = scratch + (size_t)B * heads * ns * N * ATT_DH;
    hipLaunchKernelGGL(attention_bf16x3_kernel<true>, dim3(qtiles * ns, heads, B), dim3(64 * ATT_NW), 0, s, qkv, Kp, Vtp, out, part_o, part_ml, N, Npad,
                       heads, scale, ns, plane_bytes);
    FS_HIP(hipGetLastError());
    launch_attention_combine(part_o, part_ml, out, B, N, heads, ns, s);
    FS_HIP(hipGetLastError());
    return 0;
}

int launch_attention_f32(const float* qkv, float* out, int B, int N, int heads, float scale, float* scratch, hipStream_t s) {
    FS_REQUIRE(B >= 1 && N >= 1 && heads >= 1, "attention: bad shape");
    const int qtiles = cdiv(N, 32 * ATT_NW);
    const int ns = scratch ? attention_splits(B, N, heads) : 1;
    // the 32-bit byte offsets of the direct-to-LDS form cover one image's [N][3D] rows; beyond 2 GiB (never in practice) the
    // register-staged kernel takes over
    const bool dma = ATT_USE_DMA && (int64_t)N * 3 * heads * ATT_DH * 4 < ((int64_t)1 << 31) && ((uintptr_t)qkv & 15) == 0;
    if (ns == 1) {
        if (dma) hipLaunchKernelGGL(attention_dma_kernel<false>, dim3(qtiles, heads, B), dim3(64 * ATT_NW), 0, s, qkv, out, nullptr, nullptr, N, heads, scale, 1);
        else hipLaunchKernelGGL(attention_f32_kernel<false>, dim3(qtiles, heads, B), dim3(64 * ATT_NW), 0, s, qkv, out, nullptr, nullptr, N, heads,
                           scale, 1);
        FS_HIP(hipGetLastError());
        return 0;
    }
    float* part_o = scratch;
    float* part_ml = scratch + (size_t)B * heads * ns * N * ATT_DH;
    if (dma) hipLaunchKernelGGL(attention_dma_kernel<true>, dim3(qtiles * ns, heads, B), dim3(64 * ATT_NW), 0, s, qkv, out, part_o, part_ml, N, heads, scale, ns);
    else hipLaunchKernelGGL(attention_f32_kernel<true>, dim3(qtiles * ns, heads, B), dim3(64 * ATT_NW), 0, s, qkv, out, part_o, part_ml, N, heads,
                       scale, ns);
    FS_HIP(hipGetLastError());
    launch_attention_combine(part_o, part_ml, out, B, N, heads, ns, s);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ split-K merge of a Linear: out = sum_s part[s] + bias (+ res)
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ bias,
                                                             const float* __restrict__ res, float* __restrict__ out, int64_t total4, int N4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
        for (int sidx = 1; sidx < nsplit; ++sidx) v += reinterpret_cast<const f32x4*>(part)[(int64_t)sidx * total4 + i];
        if (bias) v += reinterpret_cast<const f32x4*>(bias)[i % N4];
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

int launch_splitk_combine(const float* part, int nsplit, const float* bias, const float* res, float* out, int rows, int N, hipStream_t s) {
    FS_REQUIRE(part && out && nsplit >= 2 && rows >= 1 && N % 4 == 0, "splitk_combine: bad arguments");
    const int64_t total4 = (int64_t)rows * N / 4;
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total4, 256), 4096)), dim3(256), 0, s, part, nsplit, bias, res,
                       out, total4, N / 4);
    FS_HIP(hipGetLastError());
    return 0;
}

// The same merge followed by the next block's LayerNorm in one pass (blocks.py:89-95: x = x + mlp(norm2(x)) of block i is the input
// of norm1 of block i + 1): one wave per row, D <= 1024; the merged row goes to `out`, its LayerNorm to `ln_out`.  The row never
// returns from HBM between the two, and a launch disappears per block.  Same operations in the same order as splitk_combine_kernel
// followed by layernorm_kernel, hence bit-identical to the two launches.
// NS > 0: the number of partials at compile time (2, 3, 4: what linear_splits returns) -- the loads of all of them, of the bias and of the
// shortcut are then issued together; with a run-time count every load waited for its predecessor (round 6).  NS = 0: run-time count.
template <int NS, int NI>
__global__ __launch_bounds__(256) void splitk_combine_ln_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ bias,
                                                                const float* __restrict__ res, float* __restrict__ out, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ ln_out, int rows, int D4) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t total4 = (int64_t)rows * D4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[NI], g[NI], bt[NI];
    // NS > 0: the partial sums, bias, residual, gamma and beta of the whole row are requested before anything is used (round 6: they came
    // chunk by chunk, each chunk's six loads behind the previous chunk's store, and gamma / beta behind both reductions -- four to five
    // memory round trips in a 10-us kernel).  Lanes past the row's end read clamped addresses and contribute exact zeros.  Same sums in
    // the same order as the generic form (NS == 0: any number of partials, loaded as they are added).
    if (NS > 0) {
        f32x4 pv[NI][NS > 0 ? NS : 1], bv[NI], rv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int cc = min(lane + 64 * i, D4 - 1);
            const int64_t idx = (int64_t)row * D4 + cc;
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) pv[i][sidx] = reinterpret_cast<const f32x4*>(part)[(int64_t)sidx * total4 + idx];
            bv[i] = bias ? reinterpret_cast<const f32x4*>(bias)[cc] : z;
            rv[i] = res ? reinterpret_cast<const f32x4*>(res)[idx] : z;
            g[i] = reinterpret_cast<const f32x4*>(gamma)[cc];
            bt[i] = reinterpret_cast<const f32x4*>(beta)[cc];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 a = pv[i][0];
#pragma unroll
            for (int sidx = 1; sidx < NS; ++sidx) a += pv[i][sidx];
            if (bias) a += bv[i];
            if (res) a += rv[i];
            const int c = lane + 64 * i;
            if (c < D4) reinterpret_cast<f32x4*>(out)[(int64_t)row * D4 + c] = a;
            v[i] = c < D4 ? a : z;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane + 64 * i;
            v[i] = z;
            g[i] = bt[i] = z;
            if (c < D4) {
                const int64_t idx = (int64_t)row * D4 + c;
                f32x4 a = reinterpret_cast<const f32x4*>(part)[idx];
                for (int sidx = 1; sidx < nsplit; ++sidx) a += reinterpret_cast<const f32x4*>(part)[(int64_t)sidx * total4 + idx];
                if (bias) a += reinterpret_cast<const f32x4*>(bias)[c];
                if (res) a += reinterpret_cast<const f32x4*>(res)[idx];
                reinterpret_cast<f32x4*>(out)[idx] = a;
                v[i] = a;
                g[i] = reinterpret_cast<const f32x4*>(gamma)[c];
                bt[i] = reinterpret_cast<const f32x4*>(beta)[c];
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
#pragma unroll
    for (int i = NI; i < 4; ++i) sum += 0.f;  // four terms whatever NI is (x + 0.0 is not x for x = -0.0): the same float in every instantiation
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)(D4 * 4);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const bool ok = lane + 64 * i < D4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq = ok ? sq + d * d : sq;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = 1.f / sqrtf(sq / (float)(D4 * 4) + 1e-5f);
    f32x4* y = reinterpret_cast<f32x4*>(ln_out) + (size_t)row * D4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (v[i][e] - mean) * rstd * g[i][e] + bt[i][e];
            y[c] = r;
        }
    }
}

int launch_splitk_combine_ln(const float* part, int nsplit, const float* bias, const float* res, float* out, const float* gamma, const float* beta,
                             float* ln_out, int rows, int N, hipStream_t s) {
    FS_REQUIRE(part && out && ln_out && gamma && beta && nsplit >= 2 && rows >= 1 && N % 4 == 0 && N >= 4 && N <= 1024, "splitk_combine_ln: bad arguments");
    const dim3 grid(cdiv(rows, 4)), block(256);
    const bool narrow = N <= 512;  // two chunks of 64 float4 per row (ViT-S 384, ViT-B 768 takes four)
#define FS_CLN(NS_)                                                                                                                              \
    {                                                                                                                                            \
        if (narrow) hipLaunchKernelGGL((splitk_combine_ln_kernel<NS_, 2>), grid, block, 0, s, part, nsplit, bias, res, out, gamma, beta, ln_out, rows, N / 4); \
        else hipLaunchKernelGGL((splitk_combine_ln_kernel<NS_, 4>), grid, block, 0, s, part, nsplit, bias, res, out, gamma, beta, ln_out, rows, N / 4);        \
    }
    switch (nsplit) {
        case 2: FS_CLN(2) break;
        case 3: FS_CLN(3) break;
        case 4: FS_CLN(4) break;
        default: FS_CLN(0) break;
    }
#undef FS_CLN
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ mask head: one wave per patch token
__global__ __launch_bounds__(256) void mask_head_kernel(const float* __restrict__ pp, const float* __restrict__ cc,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out, int B, int N, int K, int D4) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // b*N + i
    if (row >= B * N) return;
    const int b = row / N, i = row - b * N;
    const int T = N + K;
    const f32x4* p = reinterpret_cast<const f32x4*>(pp) + ((size_t)b * T + i) * D4;
    float pn = 0.f;
    f32x4 pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        pv[u] = c < D4 ? p[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        pn += pv[u][0] * pv[u][0] + pv[u][1] * pv[u][1] + pv[u][2] * pv[u][2] + pv[u][3] * pv[u][3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pn += __shfl_xor(pn, off, 64);
    const float pinv = 1.f / sqrtf(pn);
    float mval = 0.f;   // lane k keeps the mask value of class k (K <= 64)
    float msum = 0.f;
    for (int k = 0; k < K; ++k) {
        const f32x4* c = reinterpret_cast<const f32x4*>(cc) + ((size_t)b * T + N + k) * D4;
        float dot = 0.f, cn = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ci = lane + 64 * u;
            if (ci < D4) {
                const f32x4 cv = c[ci];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot += pv[u][e] * cv[e];
                    cn += cv[e] * cv[e];
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dot += __shfl_xor(dot, off, 64);
            cn += __shfl_xor(cn, off, 64);
        }
        const float mk = dot * pinv / sqrtf(cn);
        if (lane == k) mval = mk;
        msum += mk;
    }
    // LayerNorm over the K class scores (decoder.mask_norm)
    const float mean = msum / (float)K;
    float d = lane < K ? mval - mean : 0.f;
    float var = d * d;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
    const float rstd = 1.f / sqrtf(var / (float)K + 1e-5f);
    if (lane < K) out[((size_t)b * K + lane) * N + i] = d * rstd * gamma[lane] + beta[lane];
}

int launch_mask_head(const float* pp, const float* cc, const float* gamma, const float* beta, float* out, int B, int N, int K,
                     int D, hipStream_t s) {
    FS_REQUIRE(D % 4 == 0 && D <= 1024 && K >= 1 && K <= 64, "mask_head: unsupported D=%d / K=%d", D, K);
    hipLaunchKernelGGL(mask_head_kernel, dim3(cdiv(B * N, 4)), dim3(256), 0, s, pp, cc, gamma, beta, out, B, N, K, D / 4);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
