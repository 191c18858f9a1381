/*
 * floodseg.h -- C ABI of the MI355X-native key-frame segmentation + flow-interpolation hot path.
 *
 * Drop-in boundary for lenke182/flood-uav-video-segmentation (citations are reference file:line):
 *   - the two callables the reference's FlowModel invokes on its network,
 *       self.model.encoder(x) / self.model.decoder(f)            flow/model.py:39-40,57-58,76-79,120,129,177,189-204
 *     for FlowPSPNet (model/pspnet.py:113-141) and FlowDeepLabv3 (model/deeplabv3.py:47-54);
 *   - the torch ops FlowModel applies around them,
 *       F.grid_sample(bilinear, border)                          flow/model.py:157,248
 *       F.interpolate(bilinear, align_corners=True)              flow/model.py:42,68,86,103,139,150,159,179,193,206,218,228
 *       (n-p)/n * a + p/n * b                                    flow/model.py:104,168,170,234-236
 *   - the post-processing inside the reference's timed region,
 *       F.interpolate(.., (1072,1920)); max(1)[1]; uint8         flow/base.py:275-277
 *   - the metric histogram intersectionAndUnionGPU               util/util.py:52-63
 *
 * Conventions
 *   - every pointer named *_dev / in / out is a DEVICE pointer (tensor.data_ptr()); nothing is freed or
 *     retained by the library except its own packed weights and workspace;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it.  fs_finalize, fs_reserve and fs_profile_dump synchronise the device.  A forward synchronises (and
 *     allocates) only when it has to GROW the library-owned workspace or build a Winograd filter bank for a geometry it has
 *     not seen: call fs_reserve(handle, B, H, W) once for the largest batch / frame size you will use and no later forward
 *     allocates, frees or synchronises (required before HIP-graph capture and before using one handle from several streams);
 *   - all functions return 0 on success; on failure a non-zero code, and fs_last_error() holds the text
 *     (the Python shim raises RuntimeError with it -- the reference itself only raises/asserts);
 *   - NCHW tensors are the reference's layout; "NHWC" tensors are pixel-major with an explicit pixel
 *     stride `ld` (in floats) so that channel slices of wider buffers can be addressed.
 */
#ifndef FLOODSEG_H_
#define FLOODSEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_net* fs_handle;
typedef void* fs_stream;

enum { FS_ARCH_PSPNET = 0, FS_ARCH_DEEPLABV3 = 1, FS_ARCH_SEGMENTER = 2 };

typedef struct fs_config {
    int arch;        /* FS_ARCH_*                               flow/base.py:94-103            */
    int layers;      /* ResNet depth 50 | 101 | 152             model/pspnet.py:45-50          */
    int classes;     /* K                                       dataset/flow/config.yaml:2     */
    /* FS_ARCH_SEGMENTER only (ignored otherwise)               model/vit.py:13-56             */
    int patch;       /* patch size P (32 as shipped, 16 for ViT-S/16)                          */
    int d_model;     /* 768 | 384 ...; head_dim is fixed to 64 => heads = d_model / 64         */
    int n_layers;    /* encoder blocks (12)                                                    */
    int dec_layers;  /* mask-transformer blocks (2)                                            */
    int image_size;  /* construction size: pos_embed holds (image_size/P)^2 + 1 rows           */
    /* Explicit A/B options of the convolutional networks (0 = the shipped default).  They select between arithmetically
     * different but parity-tested evaluation routes; nothing is read from the process environment.                     */
    int flags;          /* FS_OPT_* bits                                                                                */
    int winograd_tile;  /* 0 = per-map choice of F(4x4,3x3) / F(6x6,3x3); 4 or 6 forces one tile size                   */
} fs_config;

enum {
    FS_OPT_NO_WINOGRAD = 1,   /* every 3x3 conv on the direct implicit-GEMM kernel                                      */
    FS_OPT_NO_FUSED_HEAD = 2, /* fs_segment_forward = fs_decoder_forward(fs_encoder_forward(x)) over the 4096-ch concat */
    FS_OPT_NO_FUSED_SHORTCUT = 4, /* projection blocks: downsample and conv3 as two launches instead of one concatenated-K GEMM */
    FS_OPT_NO_FUSED_WINOGRAD = 8, /* the 3x3 convs with Cin <= 128 (deep stem, conv2 of layer1) on the direct kernel instead of the
                                    one-kernel Winograd F(4x4,3x3) (implied by FS_OPT_NO_WINOGRAD)                             */
    FS_OPT_NO_SPLIT_BF16 = 16, /* implicit-GEMM launches (1x1 / 3x3 convs, Winograd GEMMs, nn.Linear) on the fp32 matrix-core kernel
                                 (v_mfma_f32_32x32x2_f32) instead of the split-operand one: each fp32 operand as the exact sum of three
                                 bf16 terms, six cross products on the bf16 matrix cores, fp32 accumulation (fs_conv2d_nhwc_split) */
    FS_OPT_PLANE_OPERANDS = 32, /* round 4, opt-in A/B route: the Winograd input transform writes V as three bf16 planes (each value split
                                 once) and the position GEMMs run on fs_gemm_bf16x3_planes instead of splitting fp32 rows in registers
                                 inside the GEMM.  Measured slower end to end (1.5x the V bytes, profiles/r04_experiments.txt), so it
                                 is not the default; ignored with FS_OPT_NO_SPLIT_BF16 */
    FS_OPT_CHAIN = 64,         /* round 5, opt-in A/B route: conv3 (+ shortcut) of a layer1 / layer2 bottleneck and conv1 of the NEXT block as
                                 ONE chained launch (conv_chain_dma_f32: a workgroup multiplies the pixel rows it has just stored by the
                                 next filters) instead of two.  Bit-identical results either way.  Measured equal in layer1 and slower
                                 in layer2 (profiles/r05_experiments.txt), so it is not the default; needs the split-operand route
                                 (ignored with FS_OPT_NO_SPLIT_BF16) */
    FS_OPT_NO_FUSED_POOL = 256, /* round 5 A/B switch: the deep stem's last conv (layer0.6) and the max-pool behind it (model/resnet.py:114-117) as two
                                 launches instead of one (the one-kernel Winograd with MaxPool2d(3, 2, 1) in its epilogue: the 357 x 357 x 128 map
                                 is never written).  Bit-identical results either way */
    FS_OPT_ATT_PIPELINED = 512, /* round 5, opt-in A/B route (Segmenter): the split-operand attention software-pipelined inside a wave on a balanced
                                 grid (fs_attention mode 2) instead of the stage-serial kernel with equal key splits (mode 1) */
    FS_OPT_NO_RES_TOUCH = 128  /* round 5 A/B switch: without it the split-operand conv kernels touch the lines of a bottleneck's shortcut tile
                                 (dead loads) before the last K chunk of their main loop, so that the epilogue's residual loads hit in L2.
                                 Same results bit for bit either way */
};

int fs_version(void);
const char* fs_last_error(void);

/* ---- network lifecycle ------------------------------------------------------------------------- */
int fs_create(const fs_config* cfg, fs_handle* out);
int fs_destroy(fs_handle h);
/* Copy one state_dict tensor (canonical name) into the library.
 * on_device != 0: `data` is a device pointer. */
int fs_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                   fs_stream stream);
/* Fold eval-mode BatchNorm into per-channel scale/shift, repack filters OIHW -> OHWI. Synchronises. */
int fs_finalize(fs_handle h, fs_stream stream);
/* Feature-map geometry produced by fs_encoder_forward for an H x W frame. */
int fs_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw);
/* Bytes of library-owned activation workspace a forward at this geometry needs (allocated lazily, or by fs_reserve). */
size_t fs_workspace_bytes(fs_handle h, int B, int H, int W);
/* Grow every library-owned block that fs_encoder_forward / fs_decoder_forward / fs_segment_forward(2) / fs_segment_crops over
 * at most B frames (crops) of H x W will touch -- activation buffers, pooled maps, the Winograd V / M workspace, the resized
 * position table of the Segmenter -- to its final size, and build (on `stream`) the Winograd filter banks this geometry
 * selects.  Synchronises the device.  After it, a forward at this geometry performs no hipMalloc / hipFree / device sync;
 * fs_reserved_bytes() is how a caller checks: it does not change across such forwards.  Blocks only ever grow: reserving
 * several geometries in turn leaves the handle ready for all of them. */
int fs_reserve(fs_handle h, int B, int H, int W, fs_stream stream);
/* Bytes of workspace + lazily built filter banks the handle holds right now (packed weights not included). */
size_t fs_reserved_bytes(fs_handle h);

/* model.encoder(x):  in NCHW [B,3,H,W]  ->  out NHWC [B,fh,fw,C] (ld = C; a channels_last torch tensor)
 * FS_ARCH_SEGMENTER: out = the ViT tokens after the final LayerNorm, cls token dropped, [B, gh*gw, D]
 * (segm/model/segmenter.py:37-42); decoder = MaskTransformer -> [B,K,gh,gw] (before the bilinear resize). */
int fs_encoder_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nhwc, fs_stream stream);
/* model.decoder(f):  in NHWC [B,fh,fw,C]  ->  out NCHW [B,K,fh,fw] */
int fs_decoder_forward(fs_handle h, const float* feat_nhwc, int B, int fh, int fw, float* out_nchw, fs_stream stream);
/* model.decoder(model.encoder(x)) in one call -- the composition the segmentation-mode paths evaluate
 * (flow/model.py:39-40 forward_segmentation, :189-191 and :202-204 predict_segmentation; single-frame inference):
 * in NCHW [B,3,H,W] -> out NCHW [B,K,fh,fw] with (fh, fw) from fs_feature_shape.  Same result as the two calls above; the
 * PSPNet head skips the 4096-channel concat: model/pspnet.py:28-34 (PPM upsample + cat) and :70-73 (3x3 conv) are linear,
 * so the pyramid's share of the conv is evaluated on the pooled b x b maps and added before BatchNorm + ReLU. */
int fs_segment_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nchw, fs_stream stream);
/* The same two calls on a batch held in TWO tensors: images 0..Ba-1 = in_a [Ba,3,H,W], images Ba..Ba+Bb-1 = in_b [Bb,3,H,W].
 * FlowModel hands its two key frames over as separate tensors (frame_prev, frame_next: flow/model.py:189-191 and :202-204 run
 * the network once on each); both go through the network as one batch here, read in place -- no concatenation copy.
 * Results are identical to the one-tensor calls on cat(in_a, in_b).  Bb may be 0 (in_b ignored). */
int fs_encoder_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nhwc,
                        fs_stream stream);
int fs_segment_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nchw,
                        fs_stream stream);

/* Sliding-crop route (the reference's default `no_cropping=False`, flow/base.py:182-209): decoder(encoder(.)) of `ncrops`
 * windows [ch x cw] of the full frame frame_a [1,3,FH,FW] -- and, when frame_b != NULL, of the same windows of frame_b -- as ONE
 * batch, read in place from the frames (flow/base.py:199-200 clones every crop; here none is copied).  crop_y / crop_x: host
 * arrays of the windows' top-left corners.  out: NCHW [ncrops * (frame_b ? 2 : 1), K, fh, fw], the crops of frame_a first;
 * image i of the result is bit-identical to fs_segment_forward on the cloned crop.  Convolutional networks only. */
int fs_segment_crops(fs_handle h, const float* frame_a, const float* frame_b, int FH, int FW, int ncrops, const int* crop_y,
                     const int* crop_x, int ch, int cw, float* out_nchw, fs_stream stream);

/* ---- per-op profiling (HIP events on `stream` around every launch of the next forward calls) -- */
int fs_profile_enable(fs_handle h, int on);
/* Synchronises the recorded events; writes one line per op: "name kernel flops bytes ms\n". */
int fs_profile_dump(fs_handle h, char* buf, size_t buflen);

/* ---- flow / interpolation ops -------------------------------------------------------------------- */
int fs_grid_sample_nchw(const float* in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg, float* out,
                        int align_corners, fs_stream stream);
int fs_grid_sample_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg,
                        float* out, int ld_out, int align_corners, fs_stream stream);
int fs_resize_bilinear_nchw(const float* in, int BC, int Hi, int Wi, float* out, int Ho, int Wo, int align_corners,
                            fs_stream stream);
int fs_resize_bilinear_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, float* out, int ld_out, int Ho,
                            int Wo, int align_corners, fs_stream stream);
/* out = wa*a + wb*b (b may be NULL) */
int fs_blend(const float* a, float wa, const float* b, float wb, float* out, int64_t numel, fs_stream stream);

/* Fused tail of FlowModel.predict_segmentation (flow/model.py:184-241) from the two low-resolution
 * decoder outputs: upsample, (optional) warp chains at grid resolution, linear fusion, and optionally
 * the per-frame argmax.  grids_left/right: host arrays of n-1 device pointers [Hg,Wg,2] (ignored when
 * no_warp).  scratch: >= 2*(n-1)*K*Hg*Wg floats (warp mode).  lo_next may be NULL (single frame). */
int fs_seg_tail(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                float* out_logits, uint8_t* out_mask, float* scratch, fs_stream stream);

/* Fused tail of FlowModel.predict_feature between the encoder and its one batched decoder call (flow/model.py:131-171), on the
 * NHWC feature maps fs_encoder_forward returns (C % 4 == 0, one image per key frame):
 *   warp chains at grid resolution (:135-151), stack[0] = up(grid_sample(f_prev, grid0 [H0,W0,2], align_corners=True)) (:154-159),
 *   stack[p] = (n-p)/n * up(fwd[p-1]) + p/n * up(bwd[n-p-1]) (:166-171), up = bilinear align_corners=True to fh x fw, skipped when the
 *   map already has that size (:138,:149,:158);  no_warp: stack[0] = f_prev, stack[p] = (n-p)/n * f_prev + p/n * f_next.
 * stack: NHWC [n][fh][fw][C] ([1][fh][fw][C] when f_next is NULL) = the decoder's batch; the upsampled chain maps and the H0 x W0
 * resample are never materialised.  Bit-identical to fs_grid_sample_nhwc -> fs_resize_bilinear_nhwc -> fs_blend per map.
 * grids_left / grids_right: host arrays of n-1 device pointers [Hg,Wg,2]; scratch: >= 2*(n-1)*Hg*Wg*C floats (warp mode, f_next given). */
int fs_feat_tail(const float* f_prev, const float* f_next, int C, int fh, int fw, const float* const* grids_left,
                 const float* const* grids_right, int Hg, int Wg, const float* grid0, int H0, int W0, int n, int no_warp, float* stack,
                 float* scratch, fs_stream stream);

/* The same tail feeding the sliding-crop canvas instead of returning logits: compute_predict_crop's softmax over K
 * (flow/base.py:226-234) of every output frame is added to canvas[n,K,cH,cW] (float64) at the crop's offset (y0, x0) and
 * count[cH,cW] += 1 over the crop (flow/base.py:204-205) -- fs_seg_tail + fs_softmax_accumulate without the [n,K,H,W] logits
 * in HBM, bit-identical to that pair.  Crops that overlap must be accumulated by successive calls on one stream. */
int fs_seg_tail_accumulate(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                           const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                           double* canvas, double* count, int cH, int cW, int y0, int x0, float* scratch, fs_stream stream);

int fs_argmax_u8(const float* in, int B, int K, int64_t HW, uint8_t* out, fs_stream stream);
int fs_resize_argmax_u8(const float* in, int B, int K, int Hi, int Wi, uint8_t* out, int Ho, int Wo, fs_stream stream);
/* hist3K: int64[3][K] = {intersection, |pred|, |target|}, accumulated (caller zeroes). */
int fs_iou_hist(const uint8_t* pred, const uint8_t* target, int64_t numel, int K, int ignore_index, long long* hist3K,
                fs_stream stream);

/* rgb[numel,3] = palette[K,3][mask[numel]]  (flow/base.py:308-312) */
int fs_colorize(const uint8_t* mask, const uint8_t* palette, int K, uint8_t* rgb, int64_t numel, fs_stream stream);
/* H.264 block motion vectors -> forward / inverse sampling grids (dataset/flow/extract_motion_vectors.py:21-43).
 * mv: int32 [n, stride>=7] rows (source, w, h, src_x, src_y, dst_x, dst_y, ...) on the device; grids: float64 [hb, wb, 2];
 * owners: 2*hb*wb ints of scratch.  A block hit by several vectors takes the LAST one, as the reference's loop does. */
int fs_mv_to_grids(const int* mv, int n, int stride, int hb, int wb, int block, int H, int W, int* owners, double* grid,
                   double* inv_grid, fs_stream stream);

/* Sliding-crop inference (flow/base.py:182-234): canvas[n,K,H,W] (float64, zeroed by the caller) += softmax_K(logits[n,K,h,w])
 * at (y0, x0); count[H,W] += 1 over the crop.  fs_canvas_finish divides by the count (flow/base.py:208) and, if mask != NULL,
 * writes the per-frame argmax. */
int fs_softmax_accumulate(const float* logits, int n, int K, int h, int w, double* canvas, double* count, int H, int W, int y0,
                          int x0, fs_stream stream);
int fs_canvas_finish(double* canvas, const double* count, int n, int K, int64_t HW, uint8_t* mask, fs_stream stream);

/* argmax over K of F.interpolate(canvas, (Ho, Wo), bilinear, align_corners=True) evaluated in float64 (flow/base.py:275-276
 * on compute_output's result), without the [n,K,Ho,Wo] intermediate.  canvas must already be divided by the count. */
int fs_canvas_resize_argmax(const double* canvas, int n, int K, int Hi, int Wi, uint8_t* mask, int Ho, int Wo, fs_stream stream);
/* crop_motion_vector (flow/transform.py:215-261) for every crop x every grid of a window in one launch.  grids: host array of
 * `ngrids` device pointers, each fp32 [Hg, Wg, 2] normalised to the H x W frame (any number: more than 32 grids or crops are cut into several launches); crop c is the ch x cw window at
 * (crop_y[c], crop_x[c]).  out: fp32 [ncrops][ngrids][ch/16][cw/16][2]: block range by Python's round(), coordinates
 * renormalised to the crop, half-pixel bilinear resize (cv2.INTER_LINEAR) to (ch/16) x (cw/16). */
int fs_crop_grids(const float* const* grids, int ngrids, int Hg, int Wg, int H, int W, int ncrops, const int* crop_y,
                  const int* crop_x, int ch, int cw, float* out, fs_stream stream);

/* compute_output after the network (flow/base.py:182-209, 226-234) for ALL crops of a window in one pass: lo_prev / lo_next =
 * per-crop decoder logits [ncrops,K,h,w] (lo_next NULL: single frame), crop_grids = fs_crop_grids' output
 * [ncrops][2(n-1)][Hg][Wg][2] (ignored when no_warp), crop (y, x) offsets in the reference's crop order.  Every pixel of the
 * H x W frame is written ONCE: per covering crop the predict_segmentation tail and the fp32 softmax over K, summed in float64 in
 * crop order, divided by the number of covering crops -> canvas[n,K,H,W] (float64, may be NULL) and / or its argmax mask[n,H,W]
 * (may be NULL).  Bit-identical to fs_seg_tail_accumulate per crop + fs_canvas_finish.  K <= 8, at most 64 crops;
 * scratch: ncrops * 2(n-1) * K * Hg * Wg floats (warp mode). */
int fs_crops_fuse(const float* lo_prev, const float* lo_next, const float* crop_grids, int ncrops, const int* crop_y, const int* crop_x,
                  int K, int h, int w, int Hg, int Wg, int ch, int cw, int n, int no_warp, double* canvas, uint8_t* mask, int H, int W,
                  float* scratch, fs_stream stream);

/* ---- building blocks (exposed for op-level parity tests and for other host code) ---------------- */
int fs_pack_conv_weight(const float* oihw, float* ohwi, int O, int I, int KH, int KW, fs_stream stream);
/* Conv2d (+ per-channel scale/shift, + residual, + ReLU (relu = 1) / GELU (2)) on the fp32 matrix cores; Cin % 32 == 0.
 * in/out/res are NHWC with pixel strides ld_*; wgt_ohwi from fs_pack_conv_weight.  tile: 0 = cost-model choice, 1..5 force the
 * workgroup tile 128x128, 128x64, 64x64, 64x128, 256x128 (tests / sweeps); | FS_CONV_CHUNK_MAJOR = the filters are packed
 * chunk-major ([O][I/32][KH][KW][32]).  Any other bit is refused with a non-zero return. */
#define FS_CONV_CHUNK_MAJOR 0x400
int fs_conv2d_nhwc(const float* in, int ld_in, const float* wgt_ohwi, const float* scale, const float* shift,
                   const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                   int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream);
/* The same convolution with SPLIT operands: every fp32 filter value and every fp32 pixel is written as the exact sum of three bf16
 * terms (round-to-nearest residues) and the six cross products of order <= 2^-16 run on the bf16 matrix cores with fp32
 * accumulation (the three dropped ones are <= 2^-23 of the product: below the rounding of one fp32 add).  fs_split_bf16x3 writes
 * the three planes (3 * n bf16, n % 8 == 0) of a packed filter bank; fs_conv2d_nhwc_split takes them in place of wgt_ohwi
 * (tiles 0..4, 6 and 7; 4 = 64x128 has no fp32-route twin of the same wave layout; 6 = 128x96 and 7 = 256x128 on eight waves exist on
 * this route only: the cost model considers 6 where 96 divides Cout -- the Segmenter's Linears; 7 is a forced tile (sweeps)).
 * Non-finite and out-of-range operands (tests/test_gpu_ops.py::test_conv_non_finite_operands): the split is exact for every
 * finite fp32 value up to the largest bf16, |x| <= 3.3895e38 (and flushes nothing above 2^-110: the low-order term of a smaller
 * value may be a bf16 denormal).  An operand that is +-inf, NaN, or finite with 3.3895e38 < |x| <= FLT_MAX makes EVERY output it
 * contributes to NaN on this route (its leading bf16 term is infinite and the residue inf - inf); the fp32-MFMA route
 * (fs_conv2d_nhwc, FS_OPT_NO_SPLIT_BF16) follows IEEE like the reference's convolution: +-inf where the sum diverges, NaN for
 * NaN operands and inf - inf.  Outputs the operand does not contribute to are unaffected on both routes.  On BOTH routes the
 * fused ReLU epilogue is max(v, 0) and maps a NaN to 0 (torch's F.relu keeps it): a caller that must detect corrupt frames
 * checks its inputs -- an image normalised from 8-bit pixels and finite trained weights never reach any of these cases. */
/* Multi-head attention of the Segmenter (segm/model/blocks.py:39-66): out[b][n][h*64 + d] = softmax_keys(q k^T * scale) v for
 * qkv = [B][N][3 * heads * 64] (q | k | v, head-major inside each third), head_dim 64.  split_operands = 0: fp32 matrix cores;
 * 1: the split-operand route (three bf16 terms per fp32 value of q, k, v and of the probabilities, bf16 matrix cores, fp32
 * accumulation and softmax); 2: the same route software-pipelined inside a wave (round 5 experiment: the matrix cores compute the
 * scores of the next 32 keys while the vector ALU does the softmax of the current ones; bit-identical outputs; its stage loop runs
 * 1.5x faster but it needs 198 registers, i.e. two workgroups per CU instead of three, and loses that on the grid of 768 workgroups:
 * profiles/r05_experiments.txt section 16).  workspace: fs_attention_workspace_floats(B, N, heads, split_operands) floats. */
size_t fs_attention_workspace_floats(int B, int N, int heads, int split_operands);
int fs_attention(const float* qkv, float* out, int B, int N, int heads, float scale, int split_operands, float* workspace, fs_stream stream);
int fs_split_bf16x3(const float* w, int64_t n, void* planes, fs_stream stream);
int fs_conv2d_nhwc_split(const float* in, int ld_in, const void* wgt_planes, const float* scale, const float* shift,
                         const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                         int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream);
/* Round 5: two dependent 1x1 convolutions over the same M pixel rows in ONE launch (conv_chain_dma_f32; the networks use it at the
 * layer1 / layer2 bottleneck boundaries of model/resnet.py:76-96: conv3 + shortcut of block i, then conv1 of block i + 1):
 *   mid[M][C1] = act1(scale1 * (in @ W1a^T + in2 @ W1b^T) + shift1 + res)        in [M][K1]; in2 [M][K1b] optional (then no res);
 *   out[M][C2] = act2(scale2 * (mid @ W2^T) + shift2)                            res [M][C1] optional, may BE mid (in place)
 * All tensors dense (pixel stride = channel count).  w1_planes: fs_split_bf16x3 of the [C1][K1 + K1b] filter rows, w2_planes of
 * [C2][C1]; scale* / shift* may be NULL.  tile: 0 = by M, 1 = 128x128, 2 = 128x64, 3 = 64x64, 6 = 64x128 (rows x columns per
 * workgroup).  Every output is bit-identical to the two fs_conv2d_nhwc_split calls it replaces. */
int fs_conv_chain_nhwc(const float* in, int K1, const float* in2, int K1b, const void* w1_planes, const float* scale1, const float* shift1,
                       const float* res, float* mid, int C1, int relu1, const void* w2_planes, const float* scale2, const float* shift2,
                       float* out, int C2, int relu2, int M, int tile, fs_stream stream);
/* 3x3 stride-1 conv with padding == dilation as Winograd F(m x m,3x3) (transforms + (m+2)^2 grouped MFMA GEMMs); the network
 * uses it for every such conv with Cin >= 256.  tile_m: 4, 6, or 0 = whichever needs fewer GEMM rows for this map (a 90x90
 * map is exactly 15x15 tiles of 6x6).  workspace: fs_winograd_workspace_floats(..., same tile_m) floats of device memory. */
size_t fs_winograd_workspace_floats(int B, int H, int W, int Cin, int Cout, int dil, int tile_m);
int fs_conv3x3_winograd_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* out,
                             int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m, float* workspace,
                             fs_stream stream);
/* Round 4: the split-operand GEMM whose ROW operand is pre-split as well.  out[m][n] = act(scale[n] * sum_k A[m][k] W[n][k] + shift[n])
 * with A [M][K] and W [N][K] each given as three bf16 planes (fs_split_bf16x3 layout: plane t starts t * plane_elems bf16 after the
 * base; element (r, k) at (r * ld + k) inside a plane; K % 32 == 0, ld % 8 == 0); `groups` > 1: group g adds g * g_a / g_b elements
 * inside every plane and g * g_out floats to out (the Winograd position GEMMs).  bn: 128 or 64 output columns per 256-row workgroup
 * tile, 0 = by N.  Same six cross products in the same order as fs_conv2d_nhwc_split: bit-identical sums for the same operands.
 * With FS_OPT_PLANE_OPERANDS the networks use it for the Winograd GEMMs, whose input transform then writes the planes (split once per
 * value instead of once per 128 output channels inside the GEMM). */
int fs_gemm_bf16x3_planes(const void* a_planes, int64_t a_plane_elems, int ld_a, const void* b_planes, int64_t b_plane_elems, int ld_b,
                          const float* scale, const float* shift, float* out, int ld_out, int M, int N, int K, int relu, int groups,
                          int64_t g_a, int64_t g_b, int64_t g_out, int bn, fs_stream stream);
/* fs_conv3x3_winograd_nhwc on that route: filter bank split at call time, input transform writes the planes of V. */
size_t fs_winograd_planes_workspace_floats(int B, int H, int W, int Cin, int Cout, int dil, int tile_m);
int fs_conv3x3_winograd_planes_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* out,
                                    int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m, float* workspace,
                                    fs_stream stream);
/* 3x3 stride-1 pad-1 conv with FEW input channels (32 <= Cin <= 256, Cin % 32 == 0, Cout % 64 == 0) as ONE fused Winograd
 * F(4x4,3x3) kernel: input transform, the 36 position GEMMs on the fp32 matrix cores and the output transform (+ scale/shift,
 * ReLU) without the Winograd-domain tensors ever reaching HBM.  The network uses it for the deep stem's 64-channel convs and
 * conv2 of layer1 / layer2 (model/resnet.py:110-116, 67-69).  workspace: fs_winograd_fused_workspace_floats(Cin, Cout) floats
 * (the packed filter bank, rebuilt by every call of this test entry; the network builds it once at fs_finalize).
 * variant: 0 = by workgroup count, 1 = 32 tiles x 64 channels per workgroup, 2 = 16 tiles x 64 channels (two workgroups per CU),
 * 3 = 16 x 64 warp-specialised (four MFMA waves + four transform waves); all give bit-identical results. */
size_t fs_winograd_fused_workspace_floats(int Cin, int Cout);
int fs_conv3x3_winograd_fused_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* out,
                                   int ld_out, int B, int H, int W, int Cin, int Cout, int relu, int variant, float* workspace,
                                   fs_stream stream);
/* wgt_hwio: [KH][KW][3][Cout] (weight.permute(2,3,1,0)) */
int fs_stem_conv_nchw(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc,
                      int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, fs_stream stream);
/* Round 5: the same stem convolution with split operands (three bf16 terms per fp32 value, six cross products on the bf16 matrix cores,
 * fp32 accumulation: the arithmetic of fs_conv2d_nhwc_split); what the network handles run unless FS_OPT_NO_SPLIT_BF16.  Cout % 32 == 0, <= 128. */
int fs_stem_conv_nchw_split(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc,
                      int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, fs_stream stream);
/* Round 5: fs_conv3x3_winograd_fused_nhwc (+ BatchNorm + ReLU) followed by MaxPool2d(3, stride 2, padding 1) as ONE launch (the
 * deep stem's layer0.6 + max-pool): pool = [B][(H-1)/2+1][(W-1)/2+1][Cout]; bit-identical to the two calls it replaces. */
int fs_conv3x3_winograd_fused_pool_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* pool,
                                        int B, int H, int W, int Cin, int Cout, float* workspace, fs_stream stream);
int fs_maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, fs_stream stream);
int fs_adaptive_avgpool_nhwc(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin, fs_stream stream);
int fs_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, fs_stream stream);
int fs_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, fs_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOODSEG_H_ */
