// Winograd F(m x m, 3x3) transforms, m = 4 (Lavin & Gray matrices) or m = 6 (points 0, +-1, +-2, +-1/2, inf, the NNPACK
// set), around the grouped fp32-MFMA GEMM.  Replaces the direct 3x3 stride-1 pad == dil convolution + eval BatchNorm +
// ReLU of the PSPNet classifier head (reference model/pspnet.py:70-73: Conv2d(4096, 512, 3, padding=1, bias=False),
// BatchNorm2d, ReLU) and of the dilated bottleneck conv2 layers (model/resnet.py:67-69 after model/pspnet.py:55-64).
// F(6,3) needs 64 products per 36 outputs (1.78 per output) against 36 per 16 (2.25) for F(4,3), and a 90x90 map is
// exactly 15 x 15 tiles of 6x6; its fp32 error is ~1.5x that of F(4,3) (measured 1.5e-5 vs 1.0e-5 relative on a
// 512-channel conv; direct: 3e-7).  All kernels are HBM-bound elementwise-style passes over channel groups (NHWC).
#include "kernels.h"
#include "winograd.h"

namespace fs {

// ---- filter transform U = G g G^T, once at load, in double (rounded once to fp32): thread per (o, c)
template <int MT>
__global__ __launch_bounds__(256) void winograd_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I, int chunk_major) {
    constexpr int A = Wino<MT>::A;
    const int64_t total = (int64_t)O * I;
    const int ts = chunk_major ? 32 : 1;  // distance between two taps of one (o, c)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // OIHW: [o][c][3][3];  chunk-major: [o][c/32][3][3][c%32] -- (o*I + c)/32 = o*(I/32) + c/32 since I % 32 == 0
        const float* g = chunk_major ? w + (i >> 5) * 288 + (i & 31) : w + i * 9;
        double t[A][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double col[A];
            Wino<MT>::g(g[(0 * 3 + s) * ts], g[(1 * 3 + s) * ts], g[(2 * 3 + s) * ts], col);
#pragma unroll
            for (int r = 0; r < A; ++r) t[r][s] = col[r];
        }
#pragma unroll
        for (int r = 0; r < A; ++r) {
            double u[A];
            Wino<MT>::g(t[r][0], t[r][1], t[r][2], u);
#pragma unroll
            for (int q = 0; q < A; ++q) U[(size_t)(r * A + q) * total + i] = (float)u[q];
        }
    }
}

int launch_winograd_filter(const float* w_oihw, float* U, int O, int I, int mt, hipStream_t s, int chunk_major) {
    FS_REQUIRE(mt == 3 || mt == 4 || mt == 6, "winograd: tile size must be 3, 4 or 6");
    FS_REQUIRE(!chunk_major || I % 32 == 0, "winograd_filter: a chunk-major bank needs Cin %% 32 == 0");
    const int64_t total = (int64_t)O * I;
    const dim3 grid((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535));
    if (mt == 3) hipLaunchKernelGGL(winograd_filter_kernel<3>, grid, dim3(256), 0, s, w_oihw, U, O, I, chunk_major);
    else if (mt == 4) hipLaunchKernelGGL(winograd_filter_kernel<4>, grid, dim3(256), 0, s, w_oihw, U, O, I, chunk_major);
    else hipLaunchKernelGGL(winograd_filter_kernel<6>, grid, dim3(256), 0, s, w_oihw, U, O, I, chunk_major);
    FS_HIP(hipGetLastError());
    return 0;
}

// ---- input / output transforms.  Both are separable (rows, then columns); a workgroup takes ONE tile x 128 channels and splits
// the two 1-D passes over A x 32 threads through LDS: thread (x, cq) loads column x of the patch for channel quad cq (A
// float4 loads, 512 contiguous bytes per 32 lanes), transforms it and parks the A results in LDS; after the barrier thread
// (r, cq) transforms row r and stores A float4 results.  (Round 1 gave a thread a whole (m+2)^2 patch: 64 loads, 175-214
// VGPRs, and a 90x90x256 map is only 900 waves -- under one per SIMD, every load latency exposed: 15 us for 46 MB.  Split
// this way the same map is 3600 waves of 8 loads each.)  The arithmetic (B^T d B, A^T m A; which products are formed and in
// which order) is unchanged.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wo_u32x4 __attribute__((ext_vector_type(4)));

template <int MT>
__global__ __launch_bounds__((MT + 2) * 32) void winograd_input_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ V, int B, int H,
                                                                       int W, int C4, int th, int tw, int dil, long long s_pos, long long s_tile) {
    constexpr int A = Wino<MT>::A;
    __shared__ f32x4 tmp[A][A][32];  // [r][x][channel quad]
    // dilation d: the conv splits into d*d independent undilated convs on the pixel lattices (py + d*i, px + d*j);
    // tile t = (b, py, px, ty, tx) covers lattice rows m*ty-1 .. m*ty+m of phase (py, px)
    const unsigned T = (unsigned)(B * dil * dil * th * tw);  // < 2^31 (launcher)
    const int cq = threadIdx.x & 31, lane_x = threadIdx.x >> 5;  // lane_x: column in pass 1, row in pass 2
    // grid = (channel blocks, tiles): the tile index is decomposed with 32-bit divisions (a flat 64-bit block index cost six 64-bit
    // divisions per tile in every thread -- several times the transform's own arithmetic)
    const int c4 = blockIdx.x * 32 + cq;
    const bool cok = c4 < C4;
    for (unsigned t = blockIdx.y; t < T; t += gridDim.y) {
        const unsigned tw_u = (unsigned)tw, th_u = (unsigned)th, dd = (unsigned)(dil * dil);
        const unsigned q1 = t / tw_u, q2 = q1 / th_u, q3 = q2 / dd;
        const int tx = (int)(t - q1 * tw_u), ty = (int)(q1 - q2 * th_u), ph = (int)(q2 - q3 * dd), b = (int)q3;
        const int py = ph / dil, px = ph - py * dil;
        const int y0 = py + dil * (ty * MT - 1), x0 = px + dil * (tx * MT - 1);  // pad = dil <=> lattice pad 1
        {
            const int ix = x0 + dil * lane_x;
            const float* base = in + (size_t)b * H * W * ld_in + (size_t)c4 * 4;
            f32x4 col[A];
#pragma unroll
            for (int y = 0; y < A; ++y) {
                const int iy = y0 + dil * y;
                const bool ok = cok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                col[y] = ok ? *reinterpret_cast<const f32x4*>(base + ((size_t)iy * W + ix) * ld_in) : f32x4(0.f);
            }
            f32x4 tc[A];
            Wino<MT>::bt(col, tc);
#pragma unroll
            for (int r = 0; r < A; ++r) tmp[r][lane_x][cq] = tc[r];
        }
        __syncthreads();
        {
            f32x4 row[A], o[A];
#pragma unroll
            for (int x = 0; x < A; ++x) row[x] = tmp[lane_x][x][cq];
            Wino<MT>::bt(row, o);
            if (cok) {
#pragma unroll
                for (int q = 0; q < A; ++q) *reinterpret_cast<f32x4*>(V + (size_t)(lane_x * A + q) * s_pos + (size_t)t * s_tile + (size_t)c4 * 4) = o[q];
            }
        }
        __syncthreads();  // tmp is reused by the next tile of the grid-stride loop
    }
}

// Layout of the Winograd-domain tensors V [positions x tiles x C] and M [positions x tiles x N].  Tile-major ([tile][position][c]:
// one tile's 36 / 64 positions are one contiguous record, so a transform workgroup writes / reads ONE 16-64 KB region instead of
// 64 chunks of 512 B scattered 0.5 MB apart -- the grouped GEMM addresses group g as rows of pixel stride G*C starting at g*C)
// whenever the strided view stays below the 2 GiB a buffer descriptor can address; position-major otherwise.
WinoLayout winograd_layout(int mt, long long T, int C) {
    const long long G = (long long)(mt + 2) * (mt + 2);
    WinoLayout l;
    l.tile_major = T * G * C * 4 < ((long long)1 << 31);
    l.s_pos = l.tile_major ? C : T * C;
    l.s_tile = l.tile_major ? G * C : C;
    return l;
}

int launch_winograd_input(const float* in, int ld_in, float* V, int B, int H, int W, int C, int dil, int mt, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0 && dil >= 1, "winograd_input: C must be a multiple of 4");
    FS_REQUIRE(mt == 3 || mt == 4 || mt == 6, "winograd: tile size must be 3, 4 or 6");
    FS_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)V & 15) == 0, "winograd_input: unaligned operand");
    const int th = (cdiv(H, dil) + mt - 1) / mt, tw = (cdiv(W, dil) + mt - 1) / mt;
    const long long T = (long long)B * dil * dil * th * tw;
    FS_REQUIRE(T < (1ll << 31), "winograd_input: too many tiles");
    const dim3 grid((unsigned)cdiv(C / 4, 32), (unsigned)std::min<long long>(T, 65535));
    const WinoLayout lay = winograd_layout(mt, T, C);
    if (mt == 3) hipLaunchKernelGGL((winograd_input_kernel<3>), grid, dim3(5 * 32), 0, s, in, ld_in, V, B, H, W, C / 4, th, tw, dil, lay.s_pos, lay.s_tile);
    else if (mt == 4) hipLaunchKernelGGL((winograd_input_kernel<4>), grid, dim3(6 * 32), 0, s, in, ld_in, V, B, H, W, C / 4, th, tw, dil, lay.s_pos, lay.s_tile);
    else hipLaunchKernelGGL((winograd_input_kernel<6>), grid, dim3(8 * 32), 0, s, in, ld_in, V, B, H, W, C / 4, th, tw, dil, lay.s_pos, lay.s_tile);
    FS_HIP(hipGetLastError());
    return 0;
}

// ---- output transform + scale/shift + activation: pass 1 thread (q, oq) transforms column q of M, pass 2 thread (a, oq), a < m,
// transforms row a, applies BatchNorm + ReLU and stores its m pixels.
template <int MT>
__global__ __launch_bounds__((MT + 2) * 32) void winograd_output_kernel(const float* __restrict__ M, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift, float* __restrict__ out, int ld_out, int B,
                                                                        int H, int W, int N4, int th, int tw, int relu, int dil, long long s_pos,
                                                                        long long s_tile, unsigned out_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in the host pass, which only needs the launch stub)
    constexpr int A = Wino<MT>::A;
    __shared__ f32x4 tmp[MT][A][32];  // [a][q][channel quad]
    const unsigned T = (unsigned)(B * dil * dil * th * tw);
    const int oq = threadIdx.x & 31, lane_q = threadIdx.x >> 5;
    const int n4 = blockIdx.x * 32 + oq;  // grid = (channel blocks, tiles), as in the input transform
    const bool nok = n4 < N4;
    for (unsigned t = blockIdx.y; t < T; t += gridDim.y) {
        const unsigned tw_u = (unsigned)tw, th_u = (unsigned)th, dd = (unsigned)(dil * dil);
        const unsigned q1 = t / tw_u, q2 = q1 / th_u, q3 = q2 / dd;
        const int tx = (int)(t - q1 * tw_u), ty = (int)(q1 - q2 * th_u), ph = (int)(q2 - q3 * dd), b = (int)q3;
        const int py = ph / dil, px = ph - py * dil;
        {
            f32x4 col[A], y[MT];
#pragma unroll
            for (int r = 0; r < A; ++r)
                col[r] = nok ? *reinterpret_cast<const f32x4*>(M + (size_t)(r * A + lane_q) * s_pos + (size_t)t * s_tile + (size_t)n4 * 4) : f32x4(0.f);
            Wino<MT>::at(col, y);
#pragma unroll
            for (int a = 0; a < MT; ++a) tmp[a][lane_q][oq] = y[a];
        }
        __syncthreads();
        if (lane_q < MT && nok) {
            const int a = lane_q;
            f32x4 row[A], y[MT];
#pragma unroll
            for (int q = 0; q < A; ++q) row[q] = tmp[a][q][oq];
            Wino<MT>::at(row, y);
            const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + (size_t)n4 * 4) : f32x4(1.f);
            const f32x4 sh = shift ? *reinterpret_cast<const f32x4*>(shift + (size_t)n4 * 4) : f32x4(0.f);
            const int oy = py + dil * (ty * MT + a);
            // Straight-line stores through a buffer descriptor whose range check drops the pixels beyond the map's edge (sentinel offset):
            // written as `if (ox >= W) continue; store` every store sat in a basic block of its own behind an s_waitcnt vmcnt(0) -- which on
            // gfx9 also waits for the PREVIOUS store's acknowledgement: six serial round trips per thread (round 6).
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
            const unsigned rowoff = (unsigned)(((b * H + oy) * W) * ld_out + n4 * 4) * 4u;
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                const int ox = px + dil * (tx * MT + c);
                f32x4 v = y[c] * sc + sh;
                if (relu) v = __builtin_elementwise_max(v, f32x4(0.f));
                const unsigned vo = (oy < H && ox < W) ? rowoff + (unsigned)(ox * ld_out) * 4u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wo_u32x4, v), o_rsrc, vo, 0, 0);
            }
        }
        __syncthreads();
    }
#endif
}

int launch_winograd_output(const float* M, const float* scale, const float* shift, float* out, int ld_out, int B, int H, int W, int N,
                           int relu, int dil, int mt, hipStream_t s) {
    FS_REQUIRE(N % 4 == 0 && ld_out % 4 == 0 && dil >= 1, "winograd_output: N must be a multiple of 4");
    FS_REQUIRE(mt == 3 || mt == 4 || mt == 6, "winograd: tile size must be 3, 4 or 6");
    FS_REQUIRE(((uintptr_t)M & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "winograd_output: unaligned operand");
    const int th = (cdiv(H, dil) + mt - 1) / mt, tw = (cdiv(W, dil) + mt - 1) / mt;
    const long long T = (long long)B * dil * dil * th * tw;
    FS_REQUIRE(T < (1ll << 31), "winograd_output: too many tiles");
    // the stores go through a buffer descriptor with 32-bit byte offsets (as the implicit-GEMM epilogue's do)
    const long long out_bytes = ((long long)B * H * W - 1) * ld_out * 4 + (long long)N * 4;
    FS_REQUIRE(out_bytes < (1ll << 31), "winograd_output: output tensor must be smaller than 2 GiB");
    const dim3 grid((unsigned)cdiv(N / 4, 32), (unsigned)std::min<long long>(T, 65535));
    const WinoLayout lay = winograd_layout(mt, T, N);
    if (mt == 3)
        hipLaunchKernelGGL(winograd_output_kernel<3>, grid, dim3(5 * 32), 0, s, M, scale, shift, out, ld_out, B, H, W, N / 4, th, tw, relu, dil,
                           lay.s_pos, lay.s_tile, (unsigned)out_bytes);
    else if (mt == 4)
        hipLaunchKernelGGL(winograd_output_kernel<4>, grid, dim3(6 * 32), 0, s, M, scale, shift, out, ld_out, B, H, W, N / 4, th, tw, relu, dil,
                           lay.s_pos, lay.s_tile, (unsigned)out_bytes);
    else
        hipLaunchKernelGGL(winograd_output_kernel<6>, grid, dim3(8 * 32), 0, s, M, scale, shift, out, ld_out, B, H, W, N / 4, th, tw, relu, dil,
                           lay.s_pos, lay.s_tile, (unsigned)out_bytes);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
