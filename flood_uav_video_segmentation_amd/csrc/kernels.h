// Launcher declarations for the hand-written gfx950 kernels.
// All tensors are fp32. Activations inside the library are NHWC ("pixel-major"):
// element (b, y, x, c) lives at ((b*H + y)*W + x) * ld + c, ld >= C (ld > C when the
// tensor is a channel slice of a wider concat buffer).
#pragma once
#include "common.h"

namespace fs {

// ---------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//   out[m][n] = act( scale[n] * sum_k A[m][k] * Wt[n][k] + shift[n] (+ res[m][n]) )
//   m = (b, oy, ox) output pixel, n = output channel, k = (r, s, c) filter tap/channel.
// Restates nn.Conv2d + eval BatchNorm2d + ReLU (+ residual add) of the reference's
// Bottleneck (model/resnet.py:76-96) and PSPNet heads (model/pspnet.py:21-26, 70-76).
// ---------------------------------------------------------------------------------
struct ConvParams {
    const float* in;    int ld_in;   // NHWC input
    const float* wgt;                // [Cout][KH*KW*Cin], k ordered (r, s, c), c fastest
    int ld_wgt;                      // floats between two filter rows; 0 = KH*KW*Cin (+ Cin2).  > K: the launch multiplies a K-slice
                                     // of wider rows (split-K groups of one nn.Linear: group g takes columns g*Cin .. of W[out][in])
    const float* scale;              // [Cout] or nullptr (=1)
    const float* shift;              // [Cout] or nullptr (=0)
    const float* res;   int ld_res;  // residual, NHWC at output resolution, or nullptr
    float* out;         int ld_out;  // NHWC output
    int B, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad, dil;
    // second input of a concatenated-K GEMM (nullptr = none): a 1x1 conv with stride `stride2` over an H2 x W2 x Cin2 map whose
    // output geometry is Ho x Wo; its Cin2 weights follow the first conv's K in every filter row.  The bottleneck block with a
    // projection shortcut runs conv3 and downsample as ONE launch this way (model/resnet.py:86-94); first conv must be 1x1 s1.
    const float* in2;   int ld_in2;
    int Cin2, stride2, H2, W2;
    int relu;  // epilogue activation: 0 none, 1 ReLU, 2 GELU (erf)
    int res_touch;  // split route (round 5): != 0 -> after the barrier of the last-but-one chunk every lane touches two 128-B lines of the
                    // residual tile (dead loads), so that the epilogue's 64 residual loads per lane find them in L2: -3 % on layer3 /
                    // layer4 conv3, -7 % on layer1 / layer2 conv3 (profiles/r05_experiments.txt section 3); same results bit for bit
    int korder;  // weight k order: 0 = (r, s, c) ; 1 = (c/32, r, s, c%32)
    // grouped GEMM (Winograd: one GEMM per transform position): group g uses in + g*g_in, wgt + g*g_wgt, out + g*g_out
    int groups;  // 0 or 1 = plain
    long long g_in, g_wgt, g_out;  // strides in floats
    // Split-operand route (conv_igemm_dma_f32<..., SPLIT = true>): the SAME filters as `wgt`, each fp32 value written as the exact sum
    // of three bf16 terms (round-to-nearest residues), stored as three planes [3][rows][ld_wgt or K] of bf16 (split_bf16x3).
    // nullptr = the fp32-MFMA kernel.  plane_bytes = bytes of one plane.
    const void* wgt3;
    unsigned plane_bytes;
    // Segmenter qkv Linear (round 6, 128 x 96 tile of the split route only): kv_k != nullptr -> the launch is grouped by IMAGE (groups = B,
    // M = the tokens of one image, so a tile's rows are keys of one image and 32-row blocks are aligned with the attention's 16-key groups)
    // and the epilogue of the K and V column tiles writes the attention kernel's operand planes -- K as three bf16 planes [bh][Npad][64],
    // V transposed as three planes [bh][64][Npad] in the key order of the S^T accumulator (vit_ops.hip::attention_split_kv_kernel, whose
    // work this is), keys N .. Npad-1 as zeros -- instead of fp32 rows; the Q columns are stored as usual.  Cout = 3 * kv_heads * 64.
    unsigned short* kv_k;
    unsigned short* kv_vt;
    int kv_N, kv_Npad, kv_heads;
    unsigned kv_plane_bytes;   // bytes of one plane of K (= of V^T): B * heads * Npad * 64 * 2
    int kv_b;                  // image of this workgroup's group (set by the kernel)
#ifdef FS_TRACE  // tools/probe_conv_trace.hip builds only -- the field does not exist in libfloodseg.so
    int dbg;     // timing experiments (results are wrong when != 0): 2 = one block per CU, 16 = skip the epilogue,
                 // 32 | n << 8 = start workgroups bid+256.. n*256 cycles late
#endif
};
// tile: 0 = heuristic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128
int launch_conv_igemm(const ConvParams& p, hipStream_t s, int tile = 0);
const char* conv_igemm_tile_name(const ConvParams& p, int tile = 0);
// planes[t][i] (t = 0, 1, 2; bf16) with w[i] == planes[0][i] + planes[1][i] + planes[2][i] exactly (ConvParams::wgt3)
int launch_split_bf16x3(const float* w, long long n, void* planes, hipStream_t s);


// ---------------------------------------------------------------------------------
// Stem convolution with Cin = 3 read straight from the caller's NCHW frame
// (model/resnet.py:110 3x3 s2 p1; torchvision ResNet 7x7 s2 p3), + BN + ReLU, NHWC out.
// ---------------------------------------------------------------------------------
// Where the B input images of a forward pass live.  Plain mode (ncrops == 0): images 0 .. B1-1 are in[b] and images
// B1 .. B-1 are in2[b - B1], both NCHW [*,3,H,W] (the two key frames of a window are two tensors in the reference's API,
// flow/model.py:189-204: reading both in place replaces a torch.cat).  Crop mode (ncrops > 0, the sliding-crop route of
// flow/base.py:182-209): `in` and `in2` are two FULL frames [1,3,FH,FW]; image b < ncrops is the H x W window of `in` whose
// top-left corner is (cy[b], cx[b]), image ncrops + b the same window of `in2` -- the crops are never copied out.
struct FrameSrc {
    const float* in;
    const float* in2;
    int B1;
    int ncrops, FH, FW;
    short cy[32], cx[32];
};
static inline FrameSrc frames_plain(const float* in, const float* in2, int B1) {
    FrameSrc f{};
    f.in = in;
    f.in2 = in2;
    f.B1 = B1;
    return f;
}

struct StemParams {
    FrameSrc src;
    const float* wgt; // [KH*KW*3][Cout]  (tap-major, Cout fastest)
    const float* scale; const float* shift;
    float* out; int ld_out;  // NHWC [B,Ho,Wo,Cout]
    int B, H, W, Ho, Wo, Cout, KH, KW, stride, pad;
    int split;  // != 0: the split-operand route (three bf16 terms per fp32 value, bf16 matrix cores, fp32 accumulation), round 5
};
int launch_stem_conv(const StemParams& p, hipStream_t s);

// MaxPool2d(kernel 3, stride 2, padding 1) NHWC (model/resnet.py:117).
int launch_maxpool3x3s2(const float* in, int ld_in, float* out, int ld_out, int B, int H, int W, int C,
                        int Ho, int Wo, hipStream_t s);

// AdaptiveAvgPool2d(bin) NHWC -> [B][bin*bin][C] (model/pspnet.py:22; torchvision ASPPPooling).
int launch_adaptive_avgpool(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin,
                            hipStream_t s);

// bins 1, 2, 3 from the 6x6 cell means (all windows equal-sized: H, W multiples of 6)
int launch_ppm_pool_combine(const float* cell_mean, float* out1, float* out2, float* out3, int B, int C, hipStream_t s);

// Small-M 1x1 convolution (+scale/shift+ReLU): out[m][n] = act(scale[n]*dot(in[m], w[n]) + shift[n]).
// Used for the pooled PPM / ASPP-pooling branches (M = B*bin*bin <= ~100).
int launch_rowdot_1x1(const float* in, int ld_in, const float* wgt, const float* scale, const float* shift,
                      float* out, int ld_out, int M, int K, int N, int relu, hipStream_t s);
// up to four such problems of one (K, N, ld_in, ld_out) in ONE launch (the four pyramid levels: M = B, 4B, 9B, 36B)
struct RowdotProblem { const float* in; const float* wgt; const float* scale; const float* shift; float* out; int M; };
struct RowdotBatch { RowdotProblem p[4]; };
int launch_rowdot_1x1_batch(const RowdotBatch& pb, int nprob, int ld_in, int ld_out, int K, int N, int relu, hipStream_t s);

// Bilinear (align_corners as given) upsample of a tiny [B][hi*wi][C] map into a channel
// slice of an NHWC buffer (PPM: model/pspnet.py:33, align_corners=True).
int launch_upsample_into(const float* in, int hi, int wi, float* out, int ld_out, int B, int Ho, int Wo,
                         int C, int align_corners, hipStream_t s);

// Final classifier 1x1 conv with bias, NHWC in -> NCHW out (model/pspnet.py:75).
int launch_classifier_nchw(const float* in, int ld_in, const float* wgt /*[K][C]*/, const float* bias,
                           float* out /*[B,K,H,W]*/, int B, int HW, int C, int K, hipStream_t s);

// ---------------------------------------------------------------------------------
// Layout / weight packing
// ---------------------------------------------------------------------------------
int launch_pack_oihw_to_ohwi(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s);
int launch_pack_oihw_to_hwio(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s);
// [O][I/32][KH][KW][32]: the taps of one 32-channel slab are consecutive along k (ConvParams::korder == 1), so the
// pixel lines a 3x3 conv re-reads for its 9 taps are touched back to back and hit in L2.
int launch_pack_oihw_chunk_major(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s);
// out[(tap*O + o)*nc + c] = w[o][c0 + c][tap]: input-channel slice of an OIHW bank as a [taps*O][nc] 1x1 filter matrix
int launch_pack_slice_tap_major(const float* w, float* out, int O, int I, int c0, int nc, int taps, hipStream_t s);
// PSPNet head finish (net_ops.hip): v = act(scale * (T + sum_b conv-of-upsampled-pyramid term from Z_b) + shift), then the classifier
// 1x1 conv + bias (model/pspnet.py:75) on v while it is in registers: T (the raw backbone conv sums) is read once and never written
// back, logits come out NCHW [B][K][H][W].  scratch: ppm_term_scratch_floats(B, H, C) floats (row-collapsed pyramid term)
size_t ppm_term_scratch_floats(int B, int H, int C);
int launch_ppm_term_classify(const float* T, int ld, const float* const Z[4], const int bins[4], float* scratch, const float* scale,
                             const float* shift, int B, int H, int W, int C, int relu, const float* cls_w /*[K][C]*/, const float* cls_b,
                             float* logits_nchw, int K, hipStream_t s);
// out[o][0..Ka) = sa[o] * wa[o][:], out[o][Ka..Ka+Kb) = sb[o] * wb[o][:]; shift_out[o] = ha[o] + hb[o]: the filter bank and bias of
// BN_a(conv_a(x)) + BN_b(conv_b(y)) written as one GEMM over the concatenated K (both 1x1)
int launch_concat_scaled_filters(const float* wa, const float* sa, const float* ha, int Ka, const float* wb, const float* sb, const float* hb,
                                 int Kb, float* out, float* shift_out, int O, hipStream_t s);
int launch_nchw_to_nhwc(const float* in, float* out, int ld_out, int B, int C, int HW, hipStream_t s);
int launch_nhwc_to_nchw(const float* in, int ld_in, float* out, int B, int C, int HW, hipStream_t s);

// ---------------------------------------------------------------------------------
// Flow / interpolation tail (flow/model.py:184-249, flow/base.py:275-276)
// ---------------------------------------------------------------------------------
// F.grid_sample(mode=bilinear, padding_mode=border), NCHW in/out, N = 1..B.
int launch_grid_sample_nchw(const float* in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg,
                            float* out, int align_corners, hipStream_t s);
// Same on NHWC tensors (feature-based mode, C = 4096).
int launch_grid_sample_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, const float* grid, int Hg,
                            int Wg, float* out, int ld_out, int align_corners, hipStream_t s);
// F.interpolate(mode=bilinear), NCHW.
int launch_resize_bilinear_nchw(const float* in, int BC, int Hi, int Wi, float* out, int Ho, int Wo,
                                int align_corners, hipStream_t s);
// NHWC bilinear resize (feature maps).
int launch_resize_bilinear_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, float* out, int ld_out,
                                int Ho, int Wo, int align_corners, hipStream_t s);
// out = wa*a + wb*b  (b may be nullptr -> out = wa*a)
int launch_blend(const float* a, float wa, const float* b, float wb, float* out, int64_t numel, hipStream_t s);

// Fused predict_segmentation tail.
struct SegTailParams {
    const float* lo_prev;   // [K, h, w]   decoder logits of the previous key frame
    const float* lo_next;   // [K, h, w]   or nullptr (single-frame)
    const float* const* grids_left;   // n-1 device pointers [Hg,Wg,2] (ignored when no_warp)
    const float* const* grids_right;  // n-1 device pointers
    int K, h, w;            // low-res logits geometry
    int Hg, Wg;             // grid geometry
    int H, W;               // output frame size
    int n;                  // frame_delta
    int no_warp;
    float* out_logits;      // [n, K, H, W] or nullptr
    uint8_t* out_mask;      // [n, H, W] argmax or nullptr
    float* scratch;         // >= 2*(n-1)*K*Hg*Wg floats when warping
    // sliding-crop mode (flow/base.py:204-205, 226-234): canvas[n,K,cH,cW] += softmax_K(frame logits) at (y0, x0), count += 1
    double* canvas;         // float64 [n, K, cH, cW] or nullptr
    double* count;          // float64 [cH, cW]
    int cH, cW, y0, x0;
};
int launch_seg_tail(const SegTailParams& p, hipStream_t s);

// Fused predict_feature tail (flow/model.py:131-171): warp chains at grid resolution + every map of the decoder's batch in one launch.
struct FeatTailParams {
    const float* f_prev;    // NHWC [fh][fw][C] encoder features of the previous key frame
    const float* f_next;    // the next key frame's, or nullptr (single frame: stack holds one map)
    const float* const* grids_left;   // n-1 device pointers [Hg,Wg,2] (ignored when no_warp)
    const float* const* grids_right;
    const float* grid0;     // [H0,W0,2] the default (identity) grid the key-frame map is resampled through (warp mode)
    int C, fh, fw, Hg, Wg, H0, W0, n, no_warp;
    float* stack;           // NHWC [n | 1][fh][fw][C]: the decoder's batch
    float* scratch;         // >= 2*(n-1)*Hg*Wg*C floats when warping with two key frames
};
int launch_feat_tail(const FeatTailParams& p, hipStream_t s);

// Sliding crops in one pass (flow/base.py:182-209 after the network): every pixel of the full frame from the crops covering it.
struct CropsFuseParams {
    const float* lo_prev;   // [nc, K, h, w] per-crop decoder logits of the previous key frame
    const float* lo_next;   // the same for the next key frame, or nullptr (single frame)
    const float* scratch;   // warp mode: [nc][2][n-1][K][Hg][Wg] warped maps (filled by the launcher)
    int nc;
    short cy[64], cx[64];   // crop offsets, in the reference's crop order
    int K, h, w, Hg, Wg, ch, cw, n, no_warp;
    double* canvas;         // [n, K, H, W] crop-averaged softmax (already divided by the count) or nullptr
    uint8_t* mask;          // [n, H, W] its argmax or nullptr
    int H, W;
    float sy_lo, sx_lo, sy_g, sx_g;
};
// grids: [nc][2(n-1)][Hg][Wg][2] (fs_crop_grids' output) in warp mode; scratch: nc * 2(n-1) * K * Hg * Wg floats
int launch_crops_fuse(CropsFuseParams p, const float* grids, float* scratch, hipStream_t s);

// argmax over channel dim of NCHW logits -> uint8 (first max wins; flow/base.py:276).
int launch_argmax_u8(const float* in, int B, int K, int64_t HW, uint8_t* out, hipStream_t s);
// argmax of the align_corners=True bilinear upsample of NCHW logits, without materialising it
// (flow/base.py:275-276: F.interpolate(output,(1072,1920)) then max(1)[1]).
int launch_resize_argmax_u8(const float* in, int B, int K, int Hi, int Wi, uint8_t* out, int Ho, int Wo,
                            hipStream_t s);
// Sliding-crop accumulation of compute_output / compute_predict_crop (flow/base.py:182-234): softmax over K of one
// crop's logits added into a float64 canvas + per-pixel crop count; finish = divide by the count (+ argmax).
int launch_softmax_accumulate(const float* logits, int n, int K, int h, int w, double* canvas, double* count, int H, int W, int y0,
                              int x0, hipStream_t s);
int launch_canvas_finish(double* canvas, const double* count, int n, int K, int64_t HW, uint8_t* mask, hipStream_t s);
// argmax over K of the align_corners=True bilinear resize (in float64) of the crop-averaged canvas (flow/base.py:275-276)
int launch_resize_crop(const float* in, int B, int K, int Hi, int Wi, int Hfull, int Wfull, int align_corners, float* logits,
                       uint8_t* mask, int Ho, int Wo, hipStream_t s);
int launch_canvas_resize_argmax(const double* canvas, int n, int K, int Hi, int Wi, uint8_t* mask, int Ho, int Wo, hipStream_t s);
// crop_motion_vector (flow/transform.py:215-261) for every crop x every grid of a window in one launch
struct CropGridParams {
    const float* grids[32];  // ngrids device pointers, each [Hg, Wg, 2] fp32
    int ngrids, Hg, Wg;
    int H, W;                // frame size the grids are normalised to
    int ncrops, fh, fw;      // output grids [ncrops][ng_total][fh][fw][2]; this launch fills grids g0 .. g0+ngrids-1 of its crops
    int ng_total, g0;        // (a window with more than 32 grids or crops is cut into several launches by fs_crop_grids)
    short bho[32], bwo[32], bh[32], bw[32];          // block range per crop (Python-rounded on the host)
    float off_h[32], off_w[32], den_h[32], den_w[32];  // pixel offset of the crop; bh * pixels-per-block, bw * pixels-per-block
    float* out;
};
int launch_crop_grids(const CropGridParams& p, hipStream_t s);
// rgb[i] = palette[mask[i]] (flow/base.py:308-312)
int launch_colorize(const uint8_t* mask, const uint8_t* palette, int K, uint8_t* rgb, int64_t numel, hipStream_t s);
// H.264 block motion vectors -> forward / inverse sampling grids (dataset/flow/extract_motion_vectors.py:21-43), float64
int launch_mv_to_grids(const int* mv, int n, int stride, int hb, int wb, int bs, int H, int W, int* owners, double* grid,
                       double* inv_grid, hipStream_t s);
// intersection / union / target histograms (util/util.py:52-63), int64[3][K] accumulated.
int launch_iou_hist(const uint8_t* pred, const uint8_t* target, int64_t numel, int K, int ignore_index,
                    long long* hist3K, hipStream_t s);


// ---------------------------------------------------------------------------------
// Segmenter / ViT pieces (segm/model/vit.py, blocks.py, decoder.py); token matrices are row-major
// [rows][D] -- i.e. the same "NHWC with ld" layout, so every nn.Linear runs on conv_igemm_f32.
// ---------------------------------------------------------------------------------
// Zero-padded (right/bottom) im2col of non-overlapping PxP patches: NCHW frame -> [B*gh*gw][3*P*P],
// column order (c, py, px) = Conv2d(k=s=P) weight flattening (segm/model/vit.py:28-35, utils.py:65-76).
// in: images 0 .. B1-1, in2: images B1 .. B-1 (nullptr when B1 == B), as in StemParams
int launch_patchify(const float* in, const float* in2, int B1, float* out, int B, int H, int W, int P, int gh, int gw, hipStream_t s);
// X[b][0] = cls + pos[0];  X[b][1+i] = emb[b*N+i] + pos[1+i]   (segm/model/vit.py:112-131)
int launch_vit_assemble(const float* emb, const float* cls, const float* pos, float* X, int B, int N, int D, hipStream_t s);
// Z[b][i<N] = Y[b*N+i];  Z[b][N+k] = cls_emb[k]                (segm/model/decoder.py:84-86)
int launch_dec_assemble(const float* Y, const float* cls_emb, float* Z, int B, int N, int K, int D, hipStream_t s);
// nn.LayerNorm(D) per row; drop_first > 0: input rows are [B][rows_per_batch] and row 0 of every batch
// (the cls token) is skipped in the output (segm/model/segmenter.py:41-42).
int launch_layernorm(const float* in, const float* gamma, const float* beta, float* out, int rows, int D, int rows_per_batch,
                     int drop_first, hipStream_t s);
// softmax(q k^T * scale) v for all heads, fp32 MFMA flash-style; qkv rows are [3][heads][64] (blocks.py:56-77).
// scratch: attention_scratch_floats(B, N, heads) floats for the key-split partials (nullptr = single pass)
size_t attention_scratch_floats(int B, int N, int heads);
int launch_attention_f32(const float* qkv, float* out, int B, int N, int heads, float scale, float* scratch, hipStream_t s);
// the same on the bf16 matrix cores with split operands (three bf16 terms per fp32 value, fp32 accumulate: vit_ops.hip);
// planes: attention_split_floats(B, N, heads) floats of workspace for the K / V^T planes
size_t attention_split_floats(int B, int N, int heads);
int launch_attention_split(const float* qkv, float* out, int B, int N, int heads, float scale, float* scratch, float* planes, hipStream_t s, bool planes_ready = false);
// masks[b][k][i] = LayerNorm_K( <pp[b][i]/|pp|, cc[b][N+k]/|cc|> )  (segm/model/decoder.py:90-100), NCHW out
// out[r][c] = sum_s part[s][r][c] + bias[c] (+ res[r][c]): merges the split-K partial products of a Linear
int launch_splitk_combine(const float* part, int nsplit, const float* bias, const float* res, float* out, int rows, int N, hipStream_t s);
// the same merge + nn.LayerNorm(N) of the merged rows into ln_out, one pass (fc2 of block i feeding norm1 of block i + 1)
int launch_splitk_combine_ln(const float* part, int nsplit, const float* bias, const float* res, float* out, const float* gamma, const float* beta,
                             float* ln_out, int rows, int N, hipStream_t s);
int launch_mask_head(const float* pp, const float* cc, const float* gamma, const float* beta, float* out, int B, int N, int K,
                     int D, hipStream_t s);


// ---------------------------------------------------------------------------------
// Winograd F(m x m, 3x3), m = 4 or 6, for stride-1 3x3 convolutions with pad == dil and many input channels (the PSPNet
// head conv and the dilated bottleneck conv2 layers).  (m+2)^2 GEMMs [tiles x Cin] x [Cin x Cout] on the fp32 MFMA
// kernel replace the direct conv: 2.25 (m = 4) or 1.78 (m = 6) multiplies per output instead of 9; fp32 error ~1e-5
// relative (m = 4) / ~1.5e-5 (m = 6), direct 3e-7.
//   V[xi][t][c]  = (B^T d B)[xi]      input transform,  xi in 0..(m+2)^2-1, t = (b, phase, ty, tx) m x m-output tiles
//   M[xi][t][o]  = sum_c V[xi][t][c] * U[xi][o][c]      (grouped conv_igemm_dma_f32)
//   out          = act(scale * (A^T M A) + shift)       output transform
// ---------------------------------------------------------------------------------
// chunk_major != 0: `w` is the direct kernel's packed bank [O][I/32][3][3][32] (ConvParams::korder == 1) instead of OIHW
int launch_winograd_filter(const float* w, float* U /*[(m+2)^2][O][I]*/, int O, int I, int mt, hipStream_t s, int chunk_major = 0);
// dil > 1 (pad == dil): the conv is d*d independent undilated convs on the pixel lattices (py + d*i, px + d*j);
// tiles are enumerated (b, py, px, ty, tx).
int launch_winograd_input(const float* in, int ld_in, float* V /*[(m+2)^2][T][C]*/, int B, int H, int W, int C, int dil, int mt,
                          hipStream_t s);
int launch_winograd_output(const float* M /*[(m+2)^2][T][N]*/, const float* scale, const float* shift, float* out, int ld_out, int B,
                           int H, int W, int N, int relu, int dil, int mt, hipStream_t s);
// element (position xi, tile t, channel c) of V (C = Cin) / M (C = Cout) lives at xi * s_pos + t * s_tile + c
struct WinoLayout { bool tile_major; long long s_pos, s_tile; };
WinoLayout winograd_layout(int mt, long long T, int C);
// the grouped GEMM between the transforms: group g = position g, rows = tiles (fills the V / M addressing of a ConvParams)
static inline void winograd_gemm_params(ConvParams& p, int mt, int T, int Cin, int Cout) {
    const WinoLayout a = winograd_layout(mt, T, Cin), o = winograd_layout(mt, T, Cout);
    p.ld_in = (int)a.s_tile;
    p.g_in = a.s_pos;
    p.ld_out = (int)o.s_tile;
    p.g_out = o.s_pos;
}
static inline int winograd_tiles(int B, int H, int W, int dil, int mt) {
    return B * dil * dil * ((cdiv(H, dil) + mt - 1) / mt) * ((cdiv(W, dil) + mt - 1) / mt);
}
// GEMM rows the grouped launch really multiplies: (m+2)^2 groups x tiles rounded up to the 64-row tile
static inline double winograd_gemm_rows(int B, int H, int W, int dil, int mt) {
    return (double)(mt + 2) * (mt + 2) * (double)(cdiv(winograd_tiles(B, H, W, dil, mt), 64) * 64);
}
// the cheaper tile size for this map (a 90x90 map is 15x15 tiles of 6x6 exactly, but 23x23 of 4x4).  Decided on the
// geometry of ONE image, so that a frame's result does not depend on the batch it is processed in.
// Round 6: F(3,3) joins where BOTH larger forms are mostly empty tiles (lattices of <= 3 pixels: ASPP's dilation 36 on a 90x90 map).
static inline int winograd_pick_m(int /*B*/, int H, int W, int dil) {
    const int m = winograd_gemm_rows(1, H, W, dil, 6) < winograd_gemm_rows(1, H, W, dil, 4) ? 6 : 4;
    return winograd_gemm_rows(1, H, W, dil, 3) < winograd_gemm_rows(1, H, W, dil, m) ? 3 : m;
}

// ---------------------------------------------------------------------------------
// Fused Winograd F(4x4,3x3) for 3x3 stride-1 pad-1 convs with few input channels (wino_fused.hip): input transform, the 36
// position GEMMs and the output transform (+BN, ReLU) in ONE kernel -- only the input and output maps touch HBM.
// U: packed bank [36][Cin/16][Cout][16] from launch_wino4_filter_packed (w: OIHW, or the chunk-major bank when chunk_major).
// ---------------------------------------------------------------------------------
size_t wino_fused_bank_floats(int Cin, int Cout);
bool wino_fused_supported(int Cin, int Cout, int KH, int KW, int stride, int pad, int dil);
int launch_wino4_filter_packed(const float* w, float* U, int O, int I, hipStream_t s, int chunk_major = 0);
// variant: 0 = by workgroup count, 1 = 32 tiles x 64 channels per workgroup (8 waves), 2 = 16 tiles x 64 channels (4 waves, two
// workgroups per CU), 3 = 16 x 64 warp-specialised (4 MFMA waves + 4 transform waves); bit-identical results
int launch_wino4_fused(const float* in, int ld_in, const float* U, const float* scale, const float* shift, float* out, int ld_out, int B, int H,
                       int W, int Cin, int Cout, int relu, hipStream_t s, int variant = 0);

// Round 5: the same conv + BatchNorm + ReLU with MaxPool2d(3, stride 2, padding 1) fused into the epilogue (model/resnet.py:114-117:
// layer0.6 -> layer0.8 -> maxpool): pool = [B][(H-1)/2+1][(W-1)/2+1][ld_pool], zeroed by the launcher; the full-resolution map is
// never written.  Bit-identical to launch_wino4_fused followed by launch_maxpool3x3s2.
int launch_wino4_fused_pool(const float* in, int ld_in, const float* U, const float* scale, const float* shift, float* pool, int ld_pool, int B,
                            int H, int W, int Cin, int Cout, hipStream_t s);

}  // namespace fs
