/*
 * floodseg_test.h -- op-level hooks of libfloodseg.so: the building blocks behind the networks (implicit-GEMM conv, Winograd forms,
 * stem, pooling, attention, layout copies), for the parity tests (tests/test_gpu_ops.py) and the measurement tools (tools/).
 *
 * They are NOT part of the product's symbol surface (include/floodseg.h): the library exports ONE extra symbol, fs_test_hooks(), that
 * returns a table of function pointers.  The table is append-only; `size` is sizeof(fs_test_api) of the library that was built, so a
 * caller built against a longer table can tell which members exist.  Conventions (device pointers, fs_stream, return codes,
 * fs_last_error) are those of floodseg.h.  No reference call site binds to anything in this file.
 */
#ifndef FLOODSEG_TEST_H_
#define FLOODSEG_TEST_H_

#include "floodseg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* conv2d `tile` argument: | FS_CONV_CHUNK_MAJOR = the filters are packed chunk-major ([O][I/32][KH][KW][32]) */
#define FS_CONV_CHUNK_MAJOR 0x400

typedef struct fs_test_api {
    size_t size;

    /* OIHW -> OHWI filter repack (what fs_finalize does once per conv) */
    int (*pack_conv_weight)(const float* oihw, float* ohwi, int O, int I, int KH, int KW, fs_stream stream);

    /* Conv2d (+ per-channel scale/shift, + residual, + ReLU (relu = 1) / GELU (2)) on the fp32 matrix cores; Cin % 32 == 0.
     * in/out/res are NHWC with pixel strides ld_*; wgt_ohwi from pack_conv_weight.  tile: 0 = cost-model choice, 1..4 force the
     * workgroup tile 128x128, 128x64, 64x64, 64x128 (tests / sweeps), optionally | FS_CONV_CHUNK_MAJOR.  Anything else is refused. */
    int (*conv2d_nhwc)(const float* in, int ld_in, const float* wgt_ohwi, const float* scale, const float* shift, const float* res,
                       int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int dil, int relu, int tile, fs_stream stream);

    /* The same convolution with SPLIT operands: every fp32 filter value and every fp32 pixel is the exact sum of three bf16 terms
     * (round-to-nearest residues) and the six cross products of order <= 2^-16 run on the bf16 matrix cores with fp32 accumulation (the
     * three dropped ones are <= 2^-23 of the product: below the rounding of one fp32 add).  split_bf16x3 writes the three planes
     * (3 * n bf16, n % 8 == 0) of a packed filter bank; conv2d_nhwc_split takes them in place of wgt_ohwi (tiles 0..4 and 6 = 128x96,
     * which the cost model considers where 96 divides Cout: the Segmenter's Linears).
     * Non-finite and out-of-range operands (tests/test_gpu_ops.py::test_conv_non_finite_operands): the split is exact for every finite
     * fp32 value up to the largest bf16, |x| <= 3.3895e38 (and flushes nothing above 2^-110).  An operand that is +-inf, NaN, or finite
     * with 3.3895e38 < |x| <= FLT_MAX makes EVERY output it contributes to NaN on this route; the fp32-MFMA route (conv2d_nhwc,
     * FS_OPT_NO_SPLIT_BF16) follows IEEE like the reference's convolution.  On BOTH routes the fused ReLU epilogue is max(v, 0) and maps
     * a NaN to 0 (torch's F.relu keeps it): a caller that must detect corrupt frames checks its inputs. */
    int (*split_bf16x3)(const float* w, int64_t n, void* planes, fs_stream stream);
    int (*conv2d_nhwc_split)(const float* in, int ld_in, const void* wgt_planes, const float* scale, const float* shift, const float* res,
                             int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                             int pad, int dil, int relu, int tile, fs_stream stream);

    /* Multi-head attention of the Segmenter (segm/model/blocks.py:39-66): out[b][n][h*64 + d] = softmax_keys(q k^T * scale) v for
     * qkv = [B][N][3 * heads * 64] (q | k | v, head-major inside each third), head_dim 64.  split_operands = 0: fp32 matrix cores;
     * 1: the split-operand route (three bf16 terms per fp32 value of q, k, v and of the probabilities, bf16 matrix cores, fp32
     * accumulation and softmax).  workspace: attention_workspace_floats(B, N, heads, split_operands) floats. */
    size_t (*attention_workspace_floats)(int B, int N, int heads, int split_operands);
    int (*attention)(const float* qkv, float* out, int B, int N, int heads, float scale, int split_operands, float* workspace,
                     fs_stream stream);

    /* 3x3 stride-1 conv with padding == dilation as Winograd F(m x m,3x3) (transforms + (m+2)^2 grouped MFMA GEMMs); the networks use
     * it for every such conv with Cin >= 256.  tile_m: 3, 4, 6, or 0 = whichever needs the fewest GEMM rows for this map (a 90x90 map is
     * exactly 15x15 tiles of 6x6).  workspace: winograd_workspace_floats(..., same tile_m) floats of device memory. */
    size_t (*winograd_workspace_floats)(int B, int H, int W, int Cin, int Cout, int dil, int tile_m);
    int (*conv3x3_winograd_nhwc)(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* out,
                                 int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m, float* workspace,
                                 fs_stream stream);

    /* 3x3 stride-1 pad-1 conv with FEW input channels (32 <= Cin <= 256, Cin % 32 == 0, Cout % 64 == 0) as ONE fused Winograd
     * F(4x4,3x3) kernel: input transform, the 36 position GEMMs on the fp32 matrix cores and the output transform (+ scale/shift,
     * ReLU) without the Winograd-domain tensors ever reaching HBM (the deep stem's 64-channel convs, conv2 of layer1:
     * model/resnet.py:110-116, 67-69).  workspace: winograd_fused_workspace_floats(Cin, Cout) floats (the packed filter bank, rebuilt by
     * every call of this test entry; the network builds it once at fs_finalize).  variant: 0 = by workgroup count, 1 = 32 tiles x 64
     * channels per workgroup, 2 = 16 x 64 (two workgroups per CU), 3 = 16 x 64 warp-specialised; all give bit-identical results. */
    size_t (*winograd_fused_workspace_floats)(int Cin, int Cout);
    int (*conv3x3_winograd_fused_nhwc)(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* out,
                                       int ld_out, int B, int H, int W, int Cin, int Cout, int relu, int variant, float* workspace,
                                       fs_stream stream);
    /* the same conv (+ BatchNorm + ReLU) followed by MaxPool2d(3, stride 2, padding 1) as ONE launch (the deep stem's layer0.6 +
     * max-pool): pool = [B][(H-1)/2+1][(W-1)/2+1][Cout]; bit-identical to the two calls it replaces. */
    int (*conv3x3_winograd_fused_pool_nhwc)(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                            float* pool, int B, int H, int W, int Cin, int Cout, float* workspace, fs_stream stream);

    /* stem convolution read from NCHW frames; wgt_hwio: [KH][KW][3][Cout] (weight.permute(2,3,1,0)).  _split: three bf16 terms per fp32
     * value on the bf16 matrix cores (what the network handles run unless FS_OPT_NO_SPLIT_BF16); Cout % 32 == 0, <= 128. */
    int (*stem_conv_nchw)(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc, int B,
                          int H, int W, int Cout, int KH, int KW, int stride, int pad, fs_stream stream);
    int (*stem_conv_nchw_split)(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc,
                                int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, fs_stream stream);

    int (*maxpool3x3s2_nhwc)(const float* in, float* out, int B, int H, int W, int C, fs_stream stream);
    int (*adaptive_avgpool_nhwc)(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin, fs_stream stream);
    int (*nchw_to_nhwc)(const float* in, float* out, int B, int C, int HW, fs_stream stream);
    int (*nhwc_to_nchw)(const float* in, float* out, int B, int C, int HW, fs_stream stream);
} fs_test_api;

const fs_test_api* fs_test_hooks(void);

#ifdef __cplusplus
}
#endif
#endif /* FLOODSEG_TEST_H_ */
