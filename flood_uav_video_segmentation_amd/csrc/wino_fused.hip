// Fused Winograd F(4x4,3x3) convolution for the 3x3 stride-1 pad-1 convs with FEW input channels (Cin = 64 / 128: the deep
// stem's layer0.3 / layer0.6 and conv2 of layer1 / layer2, model/resnet.py:110-116, 67-69) + eval BatchNorm + ReLU.
//
// The two-pass Winograd of winograd.hip (transform -> grouped GEMM -> transform) does not pay at these sizes: V and M are
// 2.25x the map and would cross HBM twice (layer0.3 at 713^2: 590 MB for a 19-GFLOP conv), and a K = 64 GEMM runs the
// implicit-GEMM kernel at 57 TFLOP/s.  Here nothing but the input map and the output map touches HBM:
//
//   workgroup = NT = 16*WM tiles (4x4 outputs each) x NC = 16*WN output channels, all 36 Winograd positions at once;
//   wave (wm, wn) owns 16 tiles x 16 channels: 36 accumulators of v_mfma_f32_16x16x4_f32 = 144 registers per lane;
//   K loop in stages of 16 input channels:
//     * every thread loads the 6x6 patch of ONE (tile, channel) straight from the NHWC map (16 lanes = 16 consecutive
//       channels = one 64-B segment; padding = the buffer descriptor's range check), transforms it in registers (B^T d B)
//       and writes its 36 values to LDS: V[stage & 1][xi][tile][16 ch] -- a wave's 64 lanes write 256 contiguous bytes;
//     * per position xi and sub-chunk: A = one ds_read_b128 (lane (m, q): V[xi][tile m][4q..4q+3], 1 KiB contiguous per
//       wave, conflict-free), B = one 16-B global load of the packed filter bank U[xi][Cin/16][Cout][16] (lane (n, q):
//       U[..][cout n][4q..4q+3], 1 KiB contiguous per wave, L2-resident: 0.6-2.4 MB per conv), 4 MFMAs (element e of both
//       = k 4q+e: the MFMA sums over k, any k permutation is legal as long as A and B agree);
//     * LDS is double-buffered by stage (2 x 72 KiB at 32 tiles): one barrier per 16 channels.  The waves of the second tile block (wm = 1) run
//       "MFMA then transform", those of the first "transform then MFMA", so that the two waves sharing a SIMD keep the
//       matrix pipe and the VALU busy at the same time instead of in lock-step (MI355X_MICROARCH.md, two waves per SIMD, 9);
//   epilogue: a lane holds all 36 positions of 4 (tile, channel) pairs: A^T m A in registers, BatchNorm + ReLU, stores
//   (16 lanes = 16 consecutive channels of one pixel).
// FLOPs executed: 36/16 = 2.25 multiplies per output instead of 9.  fp32 error ~1e-5 relative (as F(4,3) elsewhere).
#include "kernels.h"
#include "winograd.h"

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- packed filter bank: U[xi][I/16][O][16] = (G g G^T)[xi] of filter (o, c), in double, rounded once.  Thread per (o, c).
__global__ __launch_bounds__(256) void wino4_filter_packed_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int chunk_major) {
    const int64_t total = (int64_t)O * I;
    const int ts = chunk_major ? 32 : 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int o = (int)(i / I), c = (int)(i - (int64_t)o * I);
        const float* g = chunk_major ? w + (i >> 5) * 288 + (i & 31) : w + i * 9;
        double t[6][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double col[6];
            Wino<4>::g(g[(0 * 3 + s) * ts], g[(1 * 3 + s) * ts], g[(2 * 3 + s) * ts], col);
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r][s] = col[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double u[6];
            Wino<4>::g(t[r][0], t[r][1], t[r][2], u);
#pragma unroll
            for (int q = 0; q < 6; ++q) U[(((size_t)(r * 6 + q) * (I >> 4) + (c >> 4)) * O + o) * 16 + (c & 15)] = (float)u[q];
        }
    }
}

int launch_wino4_filter_packed(const float* w, float* U, int O, int I, hipStream_t s, int chunk_major) {
    FS_REQUIRE(I % 32 == 0 && O % 16 == 0, "wino_fused: Cin %% 32 == 0 and Cout %% 16 == 0 required (Cin=%d Cout=%d)", I, O);
    const int64_t total = (int64_t)O * I;
    hipLaunchKernelGGL(wino4_filter_packed_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w, U, O, I, chunk_major);
    FS_HIP(hipGetLastError());
    return 0;
}

struct WinoFusedParams {
    const float* in;
    int ld_in;
    const float* U;      // [36][Cin/16][Cout][16]
    const float* scale;  // [Cout] or nullptr
    const float* shift;
    float* out;
    int ld_out;
    int B, H, W, Cin, Cout, relu;
    int th, tw, T;       // tiles per column / row of one image, total tiles
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void wino4_fused_kernel(WinoFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 16 * WM, NC = 16 * WN, NTHR = 64 * WM * WN;
    constexpr int ITEMS = NT * 16 / NTHR;  // (tile, channel) patches a thread transforms per sub-chunk (1 when WN == 4)
    static_assert(NT * 16 % NTHR == 0, "threads must divide the patches of a sub-chunk");
    constexpr int SUB = 36 * NT * 16;      // floats of one stage image V[xi][tile][16]: 72 KiB at 32 tiles
    __shared__ __attribute__((aligned(1024))) float lds[2 * SUB];
    constexpr unsigned BAD = 0x40000000u;  // row / column outside the image: pushes the byte offset beyond num_records (< 1 GiB)

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wv % WN, wm = wv / WN;
    const int ncb = p.Cout / NC;
    const int cb = blockIdx.x % ncb, tb = blockIdx.x / ncb;  // the workgroups of one tile block (same input) are neighbours
    const int n0 = cb * NC;
    const int nstages = p.Cin >> 4;

    // ---- transform role: ITEMS x (tile, channel) per sub-chunk; the tile is fixed for the whole kernel
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (unsigned)((long long)p.B * p.H * p.W * p.ld_in * 4), 0x00020000);
    unsigned rowoff[ITEMS][6], coloff[ITEMS][6];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = t + it * NTHR;
        const int tile = tb * NT + (item >> 4);
        const bool tv = tile < p.T;
        const int tt = tv ? tile : 0;
        const int tx = tt % p.tw, ty = (tt / p.tw) % p.th, b = tt / (p.tw * p.th);
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            rowoff[it][k] = (tv && (unsigned)(y0 + k) < (unsigned)p.H) ? (unsigned)(((b * p.H + y0 + k) * p.W) * p.ld_in * 4) : BAD;
            coloff[it][k] = ((unsigned)(x0 + k) < (unsigned)p.W) ? (unsigned)(((x0 + k) * p.ld_in + (item & 15)) * 4) : BAD;
        }
    }

    // ---- MFMA role
    const int m16 = lane & 15, q4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, (unsigned)((long long)36 * p.Cin * p.Cout * 4), 0x00020000);
    const unsigned b_voff = (unsigned)(((n0 + wn * 16 + m16) * 16 + 4 * q4) * 4);
    const unsigned u_chunk = (unsigned)p.Cout * 64u;              // bytes of one 16-channel slab [Cout][16]
    const unsigned u_pos = (unsigned)(p.Cin >> 4) * u_chunk;      // bytes of one Winograd position
    const int a_off = (wm * 16 + m16) * 16 + 4 * q4;              // floats, inside one V[xi] plane

    f32x4 acc[36];
#pragma unroll
    for (int g = 0; g < 36; ++g) acc[g] = f32x4(0.f);

    auto transform = [&](int stage, int buf) {
        {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const unsigned soff = (unsigned)(stage * 16 * 4);
                float d[6][6];
#pragma unroll
                for (int y = 0; y < 6; ++y)
#pragma unroll
                    for (int x = 0; x < 6; ++x)
                        d[y][x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, rowoff[it][y] + coloff[it][x], soff, 0));
                float r[6][6];
#pragma unroll
                for (int x = 0; x < 6; ++x) {  // B^T along y
                    float col[6], tc[6];
#pragma unroll
                    for (int y = 0; y < 6; ++y) col[y] = d[y][x];
                    Wino<4>::bt(col, tc);
#pragma unroll
                    for (int y = 0; y < 6; ++y) r[y][x] = tc[y];
                }
                float* dst = lds + buf * SUB + t + it * NTHR;
#pragma unroll
                for (int y = 0; y < 6; ++y) {  // ... then along x
                    float o[6];
                    Wino<4>::bt(r[y], o);
#pragma unroll
                    for (int x = 0; x < 6; ++x) dst[(y * 6 + x) * (NT * 16)] = o[x];
                }
            }
        }
    };

    auto multiply = [&](int stage, int buf) {
        {
            const float* vsrc = lds + buf * SUB + a_off;
            const unsigned soff0 = (unsigned)stage * u_chunk;
            // two positions in flight: the 4 MFMAs of one position depend on each other (40-cycle latency on a 32-cycle issue)
#pragma unroll
            for (int g = 0; g < 36; g += 2) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(vsrc + g * (NT * 16));
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(vsrc + (g + 1) * (NT * 16));
                const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, b_voff, soff0 + (unsigned)g * u_pos, 0));
                const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, b_voff, soff0 + (unsigned)(g + 1) * u_pos, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0[e], acc[g], 0, 0, 0);
                    acc[g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b1[e], acc[g + 1], 0, 0, 0);
                }
            }
        }
    };

    transform(0, 0);
    for (int s = 0; s < nstages; ++s) {
        __syncthreads();  // stage s is complete in buffer s & 1; everyone is done reading buffer (s + 1) & 1
        const bool more = s + 1 < nstages;
        if (more) transform(s + 1, (s + 1) & 1);
        multiply(s, s & 1);
    }

    // ---- epilogue: lane (n = lane & 15, q = lane >> 4) holds, for r = 0..3, all 36 positions of (tile wm*16 + 4q + r, channel n)
    const int n = n0 + wn * 16 + m16;
    const float sc = p.scale ? p.scale[n] : 1.f, sh = p.shift ? p.shift[n] : 0.f;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (unsigned)((long long)p.B * p.H * p.W * p.ld_out * 4), 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = tb * NT + wm * 16 + 4 * q4 + r;
        const bool tv = tile < p.T;
        const int tt = tv ? tile : 0;
        const int tx = tt % p.tw, ty = (tt / p.tw) % p.th, b = tt / (p.tw * p.th);
        float half[4][6];  // A^T m: rows
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float col[6], y4[4];
#pragma unroll
            for (int y = 0; y < 6; ++y) col[y] = acc[y * 6 + x][r];
            Wino<4>::at(col, y4);
#pragma unroll
            for (int y = 0; y < 4; ++y) half[y][x] = y4[y];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            float o4[4];
            Wino<4>::at(half[y], o4);
            const int oy = 4 * ty + y;
            const bool rowok = tv && oy < p.H;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int ox = 4 * tx + x;
                float v = o4[x] * sc + sh;
                if (p.relu) v = fmaxf(v, 0.f);
                const unsigned vo = (rowok && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * p.ld_out + n) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo, 0, 0);
            }
        }
    }
#endif
}

// Winograd-domain filter bank size (floats) of the fused kernel
size_t wino_fused_bank_floats(int Cin, int Cout) { return (size_t)36 * Cin * Cout; }

bool wino_fused_supported(int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
    return KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && Cin % 32 == 0 && Cin >= 32 && Cin <= 256 && Cout % 64 == 0;
}

// variant: 0 = by tile count, 1 = 32 tiles x 64 channels (8 waves), 2 = 16 tiles x 64 channels (4 waves, two workgroups per CU)
int launch_wino4_fused(const float* in, int ld_in, const float* U, const float* scale, const float* shift, float* out, int ld_out, int B, int H,
                       int W, int Cin, int Cout, int relu, hipStream_t s, int variant) {
    FS_REQUIRE(wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1), "wino_fused: unsupported shape (Cin=%d Cout=%d)", Cin, Cout);
    FS_REQUIRE(ld_in >= Cin && ld_out >= Cout && ((uintptr_t)U & 15) == 0 && ((uintptr_t)in & 3) == 0, "wino_fused: bad strides / alignment");
    FS_REQUIRE((int64_t)B * H * W * ld_in * 4 < (int64_t)1 << 30 && (int64_t)B * H * W * ld_out * 4 < (int64_t)1 << 31,
               "wino_fused: input map must be smaller than 1 GiB, output smaller than 2 GiB");
    WinoFusedParams p{};
    p.in = in; p.ld_in = ld_in; p.U = U; p.scale = scale; p.shift = shift; p.out = out; p.ld_out = ld_out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.th = cdiv(H, 4); p.tw = cdiv(W, 4);
    p.T = B * p.th * p.tw;
    const int ncb = Cout / 64;
    if (variant == 0) variant = (int64_t)cdiv(p.T, 32) * ncb >= 384 ? 1 : 2;  // enough 32-tile blocks to fill the chip 1.5x over?
    if (variant == 1) hipLaunchKernelGGL((wino4_fused_kernel<2, 4>), dim3(cdiv(p.T, 32) * ncb), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((wino4_fused_kernel<1, 4>), dim3(cdiv(p.T, 16) * ncb), dim3(256), 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
