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

// F(4,3) B^T d and A^T m on scalars, every multiply-add an explicit fma: all kernel forms then execute the same operations in the
// same order whatever the optimiser would have contracted -- their results are bit-identical (tests/test_gpu_ops.py), which
// lets the launcher pick the form by batch size without a frame's result depending on its batch.
// LDS image of one stage: V[position][tile][16 channels], 64 B per (position, tile) row.  The MFMA A fragment of lane (m, q) is the
// 16-B chunk q of tile row m; ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, ...) and rows 4 apart share
// their banks (4 x 64 B = the 64-bank width), so with a linear image every group had 2-4 lanes on the same banks (rocprofv3:
// SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles of the first version).  Chunk c of tile row m is therefore stored at chunk
// c ^ g(m >> 2), g = {0, 2, 3, 1}: within every lane group the four rows of one residue m % 4 then land on four different chunks.
// Returns the float offset of channel k (0..15) of tile row m inside a position's plane.
__device__ __forceinline__ int wf_voff(int m, int k) {
    const int g = (0x78 >> ((m >> 1) & 6)) & 3;  // {0, 2, 3, 1}[(m >> 2) & 3], two bits each, packed in 0b01'11'10'00
    return m * 16 + 4 * ((k >> 2) ^ g) + (k & 3);
}

__device__ __forceinline__ void wf_bt(const float d[6], float t[6]) {
    const float s12 = d[1] + d[2], d12 = d[1] - d[2], s34 = d[3] + d[4], d43 = d[4] - d[3], d42 = d[4] - d[2], d31 = d[3] - d[1];
    t[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
    t[1] = __builtin_fmaf(-4.f, s12, s34);
    t[2] = __builtin_fmaf(4.f, d12, d43);
    t[3] = __builtin_fmaf(2.f, d31, d42);
    t[4] = __builtin_fmaf(-2.f, d31, d42);
    t[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
}
__device__ __forceinline__ void wf_at(const float m[6], float y[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = (m[0] + s12) + s34;
    y[1] = __builtin_fmaf(2.f, d34, d12);
    y[2] = __builtin_fmaf(4.f, s34, s12);
    y[3] = __builtin_fmaf(8.f, d34, d12) + m[5];
}

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
    // POOL form (round 5): MaxPool2d(3, stride 2, padding 1) of the ReLU output fused into the epilogue (model/resnet.py:116-117):
    // `out` is not written; pool[b][pr][pc][c] (zero-initialised by the launcher) receives the maxima.  Workgroups own 4 x 4 BLOCKS of tiles.
    float* pool;
    int ld_pool, Hp, Wp, nby, nbx;
};

#if defined(__HIP_DEVICE_COMPILE__)
// Epilogue of both kernel forms: lane (n = lane & 15, q = lane >> 4) holds, for r = 0..3, all 36 positions of (tile tile0 + r,
// channel n).  A^T m A in registers, BatchNorm + ReLU, 16 stores per tile.  Pixel (y, x) of a tile is at byte offset
// base + (y * W + x) * ld * 4, the second term uniform (the store's scalar offset); the four tiles of a lane are consecutive, so
// one division pair finds the first and the others follow by carry.  16 lanes = 16 consecutive channels of one pixel per store.
__device__ __forceinline__ void wino4_epilogue(const f32x4 (&acc)[36], const WinoFusedParams& p, int tile, int n) {
    const float sc = p.scale ? p.scale[n] : 1.f, sh = p.shift ? p.shift[n] : 0.f;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (unsigned)((long long)p.B * p.H * p.W * p.ld_out * 4), 0x00020000);
    int tx = tile % p.tw, ty = (tile / p.tw) % p.th, b = tile / (p.tw * p.th);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool tv = tile < p.T;
        const unsigned base = (unsigned)((((b * p.H + 4 * ty) * p.W + 4 * tx) * p.ld_out + n) * 4);
        float half[4][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float col[6], y4[4];
#pragma unroll
            for (int y = 0; y < 6; ++y) col[y] = acc[y * 6 + x][r];
            wf_at(col, y4);
#pragma unroll
            for (int y = 0; y < 4; ++y) half[y][x] = y4[y];
        }
        const int ny = tv ? p.H - 4 * ty : 0, nx = p.W - 4 * tx;  // valid rows / columns of this tile (>= 4 inside the image)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            float o4[4];
            wf_at(half[y], o4);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float v = __builtin_fmaf(o4[x], sc, sh);
                if (p.relu) v = fmaxf(v, 0.f);
                const unsigned vo = (y < ny && x < nx) ? base : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, vo, (unsigned)((y * p.W + x) * p.ld_out * 4), 0);
            }
        }
        ++tile;
        if (++tx == p.tw) {
            tx = 0;
            if (++ty == p.th) { ty = 0; ++b; }
        }
    }
}

// Epilogue of the POOL form.  A workgroup owns a 4 x 4 block of tiles = 16 x 16 conv outputs; lane (n, q) holds the four tiles of
// tile row q (ix = r = 0..3, left to right), channel n.  conv -> BN -> ReLU exactly as above, then MaxPool2d(3, 2, 1): pooled row
// i covers conv rows 2i-1 .. 2i+1, so a tile row (4 conv rows, first one even) feeds pooled rows 2q (rows 0, 1), 2q + 1 (rows
// 1, 2, 3: complete) and 2q + 2 (row 3), block-local; columns alike with a carry from tile to tile.  Rows 2q + 2 are completed with
// the neighbouring lane group's first slot (one cross-lane read); what is left incomplete is the block's rim -- local pooled rows
// and columns 0 and 8 -- which neighbouring workgroups also contribute to: those 32 of 81 cells go out as integer atomic max
// (every value is >= +0 after the ReLU and the map starts at +0, so signed-integer order IS float order, and a maximum does not
// depend on the order of its operands: the result is bit-identical to conv -> store -> maxpool, run to run), the 49 inner ones as
// plain stores.  Conv outputs outside the image contribute 0 = the identity here (MaxPool2d pads with -inf; the values are >= 0).
__device__ __forceinline__ void wino4_epilogue_pool(const f32x4 (&acc)[36], const WinoFusedParams& p, int tb, int q, int n, int lane) {
    const float sc = p.scale ? p.scale[n] : 1.f, sh = p.shift ? p.shift[n] : 0.f;
    const int bx = tb % p.nbx, by = (tb / p.nbx) % p.nby, b = tb / (p.nbx * p.nby);
    const int R0 = 16 * by + 4 * q;      // first conv row of this lane's tile row
    float h[3][9];                       // [row slot][block-local pooled column]
    float carry[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float half[4][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            float col[6], y4[4];
#pragma unroll
            for (int y = 0; y < 6; ++y) col[y] = acc[y * 6 + x][r];
            wf_at(col, y4);
#pragma unroll
            for (int y = 0; y < 4; ++y) half[y][x] = y4[y];
        }
        const int C0 = 16 * bx + 4 * r;  // first conv column of this tile
        float o[4][4];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            float o4[4];
            wf_at(half[y], o4);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float v = __builtin_fmaf(o4[x], sc, sh);
                v = fmaxf(v, 0.f);
                o[y][x] = (R0 + y < p.H && C0 + x < p.W) ? v : 0.f;
            }
        }
        float rp[3][4];  // vertical part: rows {0,1}, {1,2,3}, {3}
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            rp[0][x] = fmaxf(o[0][x], o[1][x]);
            rp[1][x] = fmaxf(fmaxf(o[1][x], o[2][x]), o[3][x]);
            rp[2][x] = o[3][x];
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {  // horizontal part: local pooled column 2r = {carry, 0, 1}, 2r + 1 = {1, 2, 3}, carry = 3
            h[sl][2 * r] = fmaxf(fmaxf(carry[sl], rp[sl][0]), rp[sl][1]);
            h[sl][2 * r + 1] = fmaxf(fmaxf(rp[sl][1], rp[sl][2]), rp[sl][3]);
            carry[sl] = rp[sl][3];
        }
    }
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) h[sl][8] = carry[sl];
    // local pooled row 2q + 2 = this tile row's slot 2 and the next tile row's slot 0 (lane + 16); the last tile row keeps its own
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float below = __shfl_down(h[0][j], 16, 64);
        if (q < 3) h[2][j] = fmaxf(h[2][j], below);
    }
    int* pool = reinterpret_cast<int*>(p.pool);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        if (sl == 0 && q != 0) continue;  // row 2q is the previous tile row's combined slot 2
        const int li = 2 * q + sl;        // block-local pooled row 0..8
        const int pr = 8 * by + li;
        const bool row_rim = li == 0 || li == 8;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int pc = 8 * bx + j;
            if (pr >= p.Hp || pc >= p.Wp) continue;
            const int off = ((b * p.Hp + pr) * p.Wp + pc) * p.ld_pool + n;
            const int bits = __builtin_bit_cast(int, h[sl][j]);
            if (row_rim || j == 0 || j == 8) atomicMax(pool + off, bits);
            else pool[off] = bits;
        }
    }
}
#endif

template <int WM, int WN, bool POOL = false>
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
        int tx, ty, b;
        bool tv;
        if (POOL) {  // tile (item >> 4) = (iy, ix) of the 4 x 4 block tb = (b, by, bx)
            const int li = item >> 4;
            tx = 4 * (tb % p.nbx) + (li & 3);
            ty = 4 * ((tb / p.nbx) % p.nby) + (li >> 2);
            b = tb / (p.nbx * p.nby);
            tv = tx < p.tw && ty < p.th;
            if (!tv) tx = ty = 0;
        } else {
            const int tile = tb * NT + (item >> 4);
            tv = tile < p.T;
            const int tt = tv ? tile : 0;
            tx = tt % p.tw, ty = (tt / p.tw) % p.th, b = tt / (p.tw * p.th);
        }
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
    const int a_off = wf_voff(wm * 16 + m16, 4 * q4);              // floats, inside one V[xi] plane

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
                    wf_bt(col, tc);
#pragma unroll
                    for (int y = 0; y < 6; ++y) r[y][x] = tc[y];
                }
                float* dst = lds + buf * SUB + wf_voff((t + it * NTHR) >> 4, t & 15);
#pragma unroll
                for (int y = 0; y < 6; ++y) {  // ... then along x
                    float o[6];
                    wf_bt(r[y], o);
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

    if constexpr (POOL) wino4_epilogue_pool(acc, p, tb, q4, n0 + wn * 16 + m16, lane);
    else wino4_epilogue(acc, p, tb * NT + wm * 16 + 4 * q4, n0 + wn * 16 + m16);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Warp-specialised form (variant 3): the same arithmetic, but the two jobs have their own waves.  A workgroup = 16 tiles x 64
// output channels = 8 waves: waves 0-3 ("M") hold the 36 accumulators of 16 channels each and do nothing but MFMAs, their V
// fragment reads and their filter loads, PD position pairs ahead; waves 4-7 ("T") load and transform the patches (one (tile,
// channel) per thread and stage) TWO stages ahead into a ring of three LDS stage images.  Why: a wave's vector-memory
// operations complete in order (vmcnt), so in a wave that does both jobs every wait for a filter fragment also waits for the
// patch loads issued before it -- the patch latency lands in the MFMA stream (measured: 44 of 123 us).  With the roles in
// different waves the M waves never wait for anything but their own filter stream, the T waves have a whole stage of slack
// per transform, and the SIMD issues T's VALU work in the shadow of M's MFMAs (one M wave and one T wave per SIMD).
// One barrier per stage (16 input channels).
#ifdef FS_TRACE
// tools/probe_wino_trace.hip only: per-workgroup stamps (shader clock).  [0..15]: M wave 0 -- start, first barrier passed, end of
// stage 0..7 (as many as there are), [14] epilogue done; [16..31]: T wave 4 -- start, after transform 0, 1, then after each loop transform
__device__ unsigned long long fs_wino_trace[32 * 16384];
#define FS_WT(slot) { if (lane == 0 && blockIdx.x < 16384) fs_wino_trace[32 * blockIdx.x + (slot)] = __builtin_readcyclecounter(); }
#else
#define FS_WT(slot) {}
#endif

// The kernel is PERSISTENT: one workgroup per CU walks the (tile block, channel block) pairs blk, blk + gridDim.x, ... (the grid is
// a multiple of the channel blocks, so a workgroup keeps its channel block and its filter stream), and the T waves' run-ahead
// crosses the block boundary: while the M waves transform and store the outputs of block i (no MFMA work: ~6 k cycles), the T
// waves already load and transform the first two stages of block i + 1, whose first-touch latency (~7 k cycles) was the other
// exposed piece of a one-block workgroup.  Stages are counted globally (S = block * stages + stage) for the LDS ring and the
// barrier schedule; both roles execute exactly `total` barriers.  Measured: layer0.3 91 -> 86 us, layer0.6 164 -> 150 us (the
// two-workgroup form above still wins there, 80 / 140 us); in the steady state an M stage takes 7-10 k cycles with every CU streaming
// -- the T waves' transform (7-9 k cycles per stage; s_setprio on them changes nothing) and the M waves' MFMAs do not overlap.
// (Built and measured before settling on four M waves x 16 channels: two waves sharing each filter stream through the L1 -- half
// the L2 traffic -- make the T waves transform two patches per thread and become the critical path, 100 us against 83 on
// layer0.3; with the loads pipelined it spills.  Workgroup timelines, tools/probe_wino_trace.hip: an M stage takes 5.1 k cycles
// for 4.6 k of MFMA issue when the T waves are idle and 6.3-7.4 k while they transform -- the fp32 MFMA runs at the vector rate
// and the T waves' VALU work does not hide under it.)
// PD: filter prefetch distance in position pairs; must divide 18 (a pair's slot is pair % PD in every stage).
template <int PD>
__global__ __launch_bounds__(512, 2) void wino4_ws_kernel(WinoFusedParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(18 % PD == 0, "PD must divide 18");
    constexpr int NT = 16, NC = 64, NB = 3;  // LDS stage images in the ring (36 KiB each): the T waves run NB - 1 stages ahead
    constexpr int SUB = 36 * NT * 16;        // floats of one stage image V[xi][tile][16 ch]
    __shared__ __attribute__((aligned(1024))) float lds[NB * SUB];
    constexpr unsigned BAD = 0x40000000u;

    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ncb = p.Cout / NC;
    const int nblk = ((p.T + NT - 1) / NT) * ncb, G = gridDim.x;  // G % ncb == 0 (launcher)
    const int n0 = ((int)blockIdx.x % ncb) * NC;
    const int nstages = p.Cin >> 4;
    const int total = ((nblk - (int)blockIdx.x + G - 1) / G) * nstages;  // global stages of this workgroup

    if (wv >= 4) {
        // ================================================================ T waves: patches -> V, one (tile, channel) per thread and stage
        const int tt = t - 256;
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (unsigned)((long long)p.B * p.H * p.W * p.ld_in * 4), 0x00020000);
        // byte offsets of the 36 patch elements of this thread's (tile, channel) in the current block (BAD added for rows / columns
        // outside the image): computed once per block -- the T waves have registers to spare, and every VALU instruction they issue
        // is one the M wave of the same SIMD cannot overlap (the fp32 MFMA and the VALU share the vector ALUs)
        unsigned voff[6][6];
        auto setup_block = [&](int blk) {
            const int tile = (blk / ncb) * NT + (tt >> 4);
            const bool tv = tile < p.T;
            const int tq = tv ? tile : 0;
            const int tx = tq % p.tw, ty = (tq / p.tw) % p.th, b = tq / (p.tw * p.th);
            const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
            for (int y = 0; y < 6; ++y) {
                const unsigned ro = (tv && (unsigned)(y0 + y) < (unsigned)p.H) ? (unsigned)(((b * p.H + y0 + y) * p.W) * p.ld_in * 4) : BAD;
#pragma unroll
                for (int x = 0; x < 6; ++x)
                    voff[y][x] = ro + (((unsigned)(x0 + x) < (unsigned)p.W) ? (unsigned)(((x0 + x) * p.ld_in + (tt & 15)) * 4) : BAD);
            }
        };
        // The patch of the NEXT stage is requested before the current one is transformed (two register sets): a transform never
        // waits for its own loads, only for ones issued a whole transform earlier.  Loads are issued in global stage order, so the
        // (block, stage) of the next load is a pair of counters.
        float d[2][6][6];
        int ld_stage = 0, ld_blk = blockIdx.x;
        auto patch_load = [&](int set) {
            if (ld_stage == 0) setup_block(ld_blk);
            const unsigned soff = (unsigned)(ld_stage * 64);
#pragma unroll
            for (int y = 0; y < 6; ++y)
#pragma unroll
                for (int x = 0; x < 6; ++x) d[set][y][x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, voff[y][x], soff, 0));
            if (++ld_stage == nstages) {
                ld_stage = 0;
                ld_blk += G;
            }
        };
        const int v_off = wf_voff(tt >> 4, tt & 15);
        auto transform = [&](int ring, int set) {
#pragma unroll
            for (int x = 0; x < 6; ++x) {  // B^T along y
                float col[6], tc[6];
#pragma unroll
                for (int y = 0; y < 6; ++y) col[y] = d[set][y][x];
                wf_bt(col, tc);
#pragma unroll
                for (int y = 0; y < 6; ++y) d[set][y][x] = tc[y];
            }
            float* dst = lds + ring * SUB + v_off;
#pragma unroll
            for (int y = 0; y < 6; ++y) {  // ... then along x
                float o[6];
                wf_bt(d[set][y], o);
#pragma unroll
                for (int x = 0; x < 6; ++x) dst[(y * 6 + x) * (NT * 16)] = o[x];
            }
        };
        // global stage j is transformed from register set j & 1 into ring image j % NB; both parities written out (static register indices)
        int done = 0, ring = 0;
        auto produce = [&]() {
            if (done & 1) {
                if (done + 1 < total) patch_load(0);
                transform(ring, 1);
            } else {
                if (done + 1 < total) patch_load(1);
                transform(ring, 0);
            }
            ++done;
            if (++ring == NB) ring = 0;
        };
        if (wv == 4) FS_WT(16)
        patch_load(0);
        produce();
        if (wv == 4) FS_WT(17)
        for (int S = 0; S < total; ++S) {
            __syncthreads();  // barrier S: global stage S is complete, the M waves are done with stage S - 1 -> its image may be overwritten
            while (done < total && done < S + NB) produce();
            if (wv == 4 && S < 12) FS_WT(18 + S)
        }
        return;
    }

    // ================================================================ M waves: 36 positions x 16 tiles x 16 channels each
    const int wn = wv;
    const int m16 = lane & 15, q4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, (unsigned)((long long)36 * p.Cin * p.Cout * 4), 0x00020000);
    const unsigned b_voff = (unsigned)(((n0 + wn * 16 + m16) * 16 + 4 * q4) * 4);
    const unsigned u_chunk = (unsigned)p.Cout * 64u;          // bytes of one 16-channel slab [Cout][16]
    const unsigned u_pos = (unsigned)(p.Cin >> 4) * u_chunk;  // bytes of one Winograd position
    const int a_off = wf_voff(m16, 4 * q4);

    f32x4 acc[36];
#pragma unroll
    for (int g = 0; g < 36; ++g) acc[g] = f32x4(0.f);
    f32x4 bq[PD][2];
    auto load_b = [&](int stage, int pair, int slot) {
        const unsigned so = (unsigned)stage * u_chunk + (unsigned)(2 * pair) * u_pos;
        bq[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, b_voff, so, 0));
        bq[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, b_voff, so + u_pos, 0));
    };
#pragma unroll
    for (int k = 0; k < PD; ++k) load_b(0, k, k);
    if (wv == 0) FS_WT(0)

    int blk = blockIdx.x, ms = 0, ring = 0;  // block being multiplied, stage inside it, ring image of the global stage
    for (int S = 0; S < total; ++S) {
        __syncthreads();
        if (wv == 0 && S == 0) FS_WT(1)
        const float* vsrc = lds + ring * SUB + a_off;
        const bool more = S + 1 < total;
        const int ns = ms + 1 == nstages ? 0 : ms + 1;  // the stage after this one: the next block starts at its stage 0 (same filters)
        f32x4 aq[2][2];
        aq[0][0] = *reinterpret_cast<const f32x4*>(vsrc);
        aq[0][1] = *reinterpret_cast<const f32x4*>(vsrc + NT * 16);
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            if (k + 1 < 18) {
                aq[(k + 1) & 1][0] = *reinterpret_cast<const f32x4*>(vsrc + (2 * k + 2) * (NT * 16));
                aq[(k + 1) & 1][1] = *reinterpret_cast<const f32x4*>(vsrc + (2 * k + 3) * (NT * 16));
            }
            // two accumulator chains alternate: the next MFMA of a chain may issue 40 cycles after the previous one, the pipe takes one per 32
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * k] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[k & 1][0][e], bq[k % PD][0][e], acc[2 * k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[2 * k + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[k & 1][1][e], bq[k % PD][1][e], acc[2 * k + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k + PD < 18) load_b(ms, k + PD, k % PD);
            else if (more) load_b(ns, k + PD - 18, k % PD);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wv == 0 && S < 12) FS_WT(2 + S)
        if (++ring == NB) ring = 0;
        if (++ms == nstages) {  // block complete: outputs, then the next block's accumulators start from zero
            wino4_epilogue(acc, p, (blk / ncb) * NT + 4 * q4, n0 + wn * 16 + m16);
            if (wv == 0 && blk == (int)blockIdx.x) FS_WT(14)
#pragma unroll
            for (int g = 0; g < 36; ++g) acc[g] = f32x4(0.f);
            ms = 0;
            blk += G;
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
    // Measured on MI355X (profiles/r03_wino_fused.txt): the 16-tile form with two workgroups per CU (2) wins when the launch is
    // several rounds of workgroups deep (stem convs: 81 / 141 us against 91 / 164), the warp-specialised form (3) when a CU gets
    // one workgroup or none (layer1 / layer2 conv2: 27 / 39 us against 32 / 50).  All forms give bit-identical results (same
    // products, same order), so the choice may depend on the batch.
    if (variant == 0) variant = (int64_t)cdiv(p.T, 16) * ncb > 512 ? 2 : 3;
    if (variant == 3) {  // persistent: one workgroup per CU, a multiple of the channel blocks
        const int nblk = cdiv(p.T, 16) * ncb;
        hipLaunchKernelGGL((wino4_ws_kernel<6>), dim3(std::min(nblk, 256 - 256 % ncb)), dim3(512), 0, s, p);
    }
    else if (variant == 1) hipLaunchKernelGGL((wino4_fused_kernel<2, 4>), dim3(cdiv(p.T, 32) * ncb), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((wino4_fused_kernel<1, 4>), dim3(cdiv(p.T, 16) * ncb), dim3(256), 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

// The same convolution + BatchNorm + ReLU followed by MaxPool2d(3, stride 2, padding 1) in ONE kernel (the deep stem's layer0.6 and
// the max-pool behind it, model/resnet.py:114-117): the full-resolution map is never written.  pool: [B][Hp][Wp][ld_pool] with
// Hp = (H - 1) / 2 + 1; it is zeroed here (the rim cells of every 16 x 16 block are integer atomic maxima against +0).
int launch_wino4_fused_pool(const float* in, int ld_in, const float* U, const float* scale, const float* shift, float* pool, int ld_pool, int B,
                            int H, int W, int Cin, int Cout, hipStream_t s) {
    FS_REQUIRE(wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1), "wino_fused_pool: unsupported shape (Cin=%d Cout=%d)", Cin, Cout);
    FS_REQUIRE(ld_in >= Cin && ld_pool >= Cout && ((uintptr_t)U & 15) == 0 && ((uintptr_t)in & 3) == 0 && ((uintptr_t)pool & 3) == 0, "wino_fused_pool: bad strides / alignment");
    const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    FS_REQUIRE((int64_t)B * H * W * ld_in * 4 < (int64_t)1 << 30 && (int64_t)B * Hp * Wp * ld_pool * 4 < (int64_t)1 << 31,
               "wino_fused_pool: input map must be smaller than 1 GiB, pooled map smaller than 2 GiB");
    WinoFusedParams p{};
    p.in = in; p.ld_in = ld_in; p.U = U; p.scale = scale; p.shift = shift; p.out = nullptr; p.ld_out = 0;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = 1;
    p.th = cdiv(H, 4); p.tw = cdiv(W, 4);
    p.T = B * p.th * p.tw;
    p.pool = pool; p.ld_pool = ld_pool; p.Hp = Hp; p.Wp = Wp; p.nby = cdiv(p.th, 4); p.nbx = cdiv(p.tw, 4);
    FS_HIP(hipMemsetAsync(pool, 0, (size_t)B * Hp * Wp * ld_pool * sizeof(float), s));
    hipLaunchKernelGGL((wino4_fused_kernel<1, 4, true>), dim3((unsigned)(B * p.nby * p.nbx * (Cout / 64))), dim3(256), 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
