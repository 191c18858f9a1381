// Segmenter executor: ViT encoder + mask-transformer decoder as a fixed launch sequence.
// Restates model/vit.py:13-56 -> segm/model/segmenter.py:32-48, vit.py:108-137, blocks.py:16-95,
// decoder.py:80-102, utils.py:22-40,65-76.  Every nn.Linear runs on conv_igemm_f32 (1x1 "conv" over
// the token matrix), attention on attention_f32_kernel.
#include "net.h"

namespace fs {
namespace {

int copy_param(fs_net* h, const std::string& name, int64_t expect, float** dst) {
    const RawTensor* t;
    FS_TRY(fetch(h, name, &t));
    FS_REQUIRE(expect < 0 || t->numel() == expect, "'%s' has %lld elements, expected %lld", name.c_str(), (long long)t->numel(),
               (long long)expect);
    FS_TRY(dev_alloc(h, dst, (size_t)t->numel()));
    FS_HIP(hipMemcpy(*dst, t->d, (size_t)t->numel() * sizeof(float), hipMemcpyDeviceToDevice));
    return 0;
}

int make_linear(fs_net* h, Linear& l, const std::string& prefix, int in, int out, bool bias, hipStream_t s) {
    l.name = prefix;
    l.in = in;
    l.out = out;
    FS_TRY(copy_param(h, prefix + ".weight", (int64_t)in * out, &l.w));
    if (bias) FS_TRY(copy_param(h, prefix + ".bias", out, &l.b));
    FS_HIP(hipDeviceSynchronize());  // the copy above ran on the null stream; `s` need not be ordered behind it
    return split_attach(h, l.w, (size_t)in * out, s);
}

int make_norm(fs_net* h, LNorm& n, const std::string& prefix, int D) {
    n.D = D;
    FS_TRY(copy_param(h, prefix + ".weight", D, &n.g));
    FS_TRY(copy_param(h, prefix + ".bias", D, &n.b));
    return 0;
}

int make_block(fs_net* h, VitBlock& b, const std::string& p, int D, hipStream_t s) {
    FS_TRY(make_norm(h, b.n1, p + "norm1", D));
    FS_TRY(make_norm(h, b.n2, p + "norm2", D));
    FS_TRY(make_linear(h, b.qkv, p + "attn.qkv", D, 3 * D, true, s));
    FS_TRY(make_linear(h, b.proj, p + "attn.proj", D, D, true, s));
    FS_TRY(make_linear(h, b.fc1, p + "mlp.fc1", D, 4 * D, true, s));
    FS_TRY(make_linear(h, b.fc2, p + "mlp.fc2", 4 * D, D, true, s));
    return 0;
}

// A Linear whose 64x64 tiles do not even give every CU two workgroups (ViT-S/16 fc2: 61 x 6 = 366 tiles for 256 CUs, 110 of
// them with two and the rest with one) while its K is long: cut K into `split` slices -- grouped launch, group g multiplies
// columns g*K/split .. of the [out][in] weight rows (ConvParams::ld_wgt) into its own partial buffer -- and merge the partials
// with bias + residual in one small pass.  S/16 fc2 59 -> 4x us (profiles/r02_experiments.txt).  0 = no split.
int linear_splits(const Linear& l, int rows_per_image, int act, bool split_route) {
    // decided on ONE image's rows (for the usual batch of two key frames: ~768 workgroups), never on the batch: a frame's
    // result must not depend on the batch it is computed in (the key-frame cache relies on it)
    if (act != 0 || l.in < 768) return 0;
    // the tile the launch will get (conv_igemm.hip::pick_tile): 128 x 96 where 96 divides the columns -- a tile of the split-operand
    // route only (the fp32-MFMA route keeps the 64 x 64 estimate it was fitted with) -- else 64 x 64; aim at two workgroups
    // per CU for a pair of images
    const bool t96 = split_route && l.out % 96 == 0;
    const long tiles = t96 ? (long)cdiv(rows_per_image, 128) * (l.out / 96) : (long)cdiv(rows_per_image, 64) * cdiv(l.out, 64);
    const long want = t96 ? 256 : 384;
    int split = (int)std::min<long>(4, (want + tiles / 2) / std::max<long>(tiles, 1));
    while (split >= 2 && (l.in % (32 * split) != 0 || l.in / split < 384)) --split;
    return split >= 2 ? split : 0;
}

// out[rows][l.out] = act(in[rows][l.in] @ W^T + b (+ res))
// ln / ln_out: when the Linear is split-K, the merge pass also writes LayerNorm(ln)(out) into ln_out (the next block's norm1) and
// *ln_done is set; otherwise the caller runs that LayerNorm itself.
int run_linear(fs_net* h, const Linear& l, const float* in, int rows, float* out, const float* res, int act, hipStream_t s, float* part = nullptr,
               int rows_per_image = 0, const LNorm* ln = nullptr, float* ln_out = nullptr, bool* ln_done = nullptr) {
    const int split = part ? linear_splits(l, rows_per_image ? rows_per_image : rows, act, h->use_split) : 0;
    if (ln_done) *ln_done = false;
    if (split) {
        ConvParams p{};
        p.in = in; p.ld_in = l.in; p.wgt = l.w; p.ld_wgt = l.in; p.out = part; p.ld_out = l.out;
        p.B = 1; p.H = rows; p.W = 1; p.Cin = l.in / split; p.Ho = rows; p.Wo = 1; p.Cout = l.out;
        p.KH = p.KW = 1; p.stride = 1; p.dil = 1;
        p.groups = split;
        p.g_in = l.in / split;
        p.g_wgt = l.in / split;
        p.g_out = (long long)rows * l.out;
        split_use(h, p);
        const double flops = 2.0 * rows * (double)l.in * l.out;
        FS_TRY(prof_begin(h, l.name, conv_igemm_tile_name(p), flops, 4.0 * ((double)rows * (l.in + (split + 1.0) * l.out) + (double)l.in * l.out), s));
        FS_TRY(launch_conv_igemm(p, s));
        if (ln && ln_out && ln->D == l.out) {
            FS_TRY(launch_splitk_combine_ln(part, split, l.b, res, out, ln->g, ln->b, ln_out, rows, l.out, s));
            if (ln_done) *ln_done = true;
        } else {
            FS_TRY(launch_splitk_combine(part, split, l.b, res, out, rows, l.out, s));
        }
        return prof_end(h, s);
    }
    ConvParams p{};
    p.in = in;
    p.ld_in = l.in;
    p.wgt = l.w;
    p.scale = nullptr;
    p.shift = l.b;
    p.res = res;
    p.ld_res = l.out;
    p.out = out;
    p.ld_out = l.out;
    p.B = 1;
    p.H = rows;
    p.W = 1;
    p.Cin = l.in;
    p.Ho = rows;
    p.Wo = 1;
    p.Cout = l.out;
    p.KH = p.KW = 1;
    p.stride = 1;
    p.pad = 0;
    p.dil = 1;
    p.relu = act;
    p.res_touch = res != nullptr && h->res_touch;
    split_use(h, p);
    const double flops = 2.0 * rows * (double)l.in * l.out;
    FS_TRY(prof_begin(h, l.name, conv_igemm_tile_name(p), flops, 4.0 * ((double)rows * (l.in + l.out) + (double)l.in * l.out), s));
    FS_TRY(launch_conv_igemm(p, s));
    return prof_end(h, s);
}

// qkv = in @ W^T + b for B images of `tokens` rows, grouped by image; the Q columns go to out[rows][3D] as fp32, the K and V columns to the
// attention's operand planes (layout of launch_attention_split: K planes, then V^T planes, keys padded to a multiple of 32)
int run_linear_qkv(fs_net* h, const Linear& l, const float* in, int B, int tokens, float* out, float* planes, hipStream_t s) {
    const int D = l.in, heads = D / 64, Npad = (tokens + 31) / 32 * 32;
    const size_t plane_elems = (size_t)B * heads * Npad * 64;
    ConvParams p{};
    p.in = in; p.ld_in = l.in; p.wgt = l.w; p.shift = l.b; p.out = out; p.ld_out = l.out;
    p.B = 1; p.H = tokens; p.W = 1; p.Cin = l.in; p.Ho = tokens; p.Wo = 1; p.Cout = l.out;
    p.KH = p.KW = 1; p.stride = 1; p.dil = 1;
    p.groups = B;
    p.g_in = (long long)tokens * l.in;
    p.g_wgt = 0;
    p.g_out = (long long)tokens * l.out;
    p.kv_k = reinterpret_cast<unsigned short*>(planes);
    p.kv_vt = p.kv_k + 3 * plane_elems;
    p.kv_N = tokens; p.kv_Npad = Npad; p.kv_heads = heads;
    p.kv_plane_bytes = (unsigned)(plane_elems * 2);
    split_use(h, p);
    FS_REQUIRE(p.wgt3, "segmenter: the fused qkv epilogue needs the split filter bank");
    const double rows = (double)B * tokens;
    FS_TRY(prof_begin(h, l.name, conv_igemm_tile_name(p, 6), 2.0 * rows * (double)l.in * l.out, 4.0 * (rows * (l.in + l.out) + (double)l.in * l.out), s));
    FS_TRY(launch_conv_igemm(p, s, 6));
    return prof_end(h, s);
}

int run_norm(fs_net* h, const LNorm& n, const float* in, float* out, int rows, int rows_per_batch, int drop_first, hipStream_t s) {
    FS_TRY(prof_begin(h, "layernorm", "layernorm", 0, 8.0 * rows * n.D, s));
    FS_TRY(launch_layernorm(in, n.g, n.b, out, rows, n.D, rows_per_batch, drop_first, s));
    return prof_end(h, s);
}

struct VitWs {
    float *X, *Xn, *QKV, *A, *Hd, *patches, *emb, *att, *part, *kv;  // kv: K / V^T planes of the split-operand attention
};

int vit_workspace(fs_net* h, int B, int tokens, VitWs* ws) {
    const size_t D = (size_t)h->cfg.d_model, T = (size_t)B * tokens;
    const size_t P2 = (size_t)3 * h->cfg.patch * h->cfg.patch;
    const size_t att = attention_scratch_floats(B, tokens, (int)(D / 64));
    const size_t kv = h->use_split ? attention_split_floats(B, tokens, (int)(D / 64)) : 0;
    const size_t need = T * D * 3 + T * 3 * D + T * 4 * D + T * P2 + T * D + att + 4 * T * D + kv + 64;  // + split-K partials (<= 4 slices)
    FS_TRY(ws_grow(h, &h->vit_ws, &h->vit_ws_elems, need, false));
    float* p = h->vit_ws;
    ws->X = p; p += T * D;
    ws->Xn = p; p += T * D;
    ws->A = p; p += T * D;
    ws->QKV = p; p += T * 3 * D;
    ws->Hd = p; p += T * 4 * D;
    ws->patches = p; p += T * P2;
    ws->emb = p; p += T * D;
    ws->att = att ? p : nullptr;
    p += att;
    ws->part = p;
    p += 4 * T * D;
    ws->kv = kv ? p + ((64 - ((uintptr_t)p / 4) % 64) % 64) : nullptr;  // 256-B aligned inside the block (the + 64 of `need`)
    return 0;
}

// pre-LN transformer block, in place on X (blocks.py:89-95).  n1_done: ws.Xn already holds norm1(X) (written by the previous block's
// fc2 merge pass); next_n1: the following block's norm1, computed by this block's fc2 merge when that Linear is split-K
// (*next_done tells the caller).
int run_block(fs_net* h, const VitBlock& blk, const VitWs& ws, int B, int tokens, hipStream_t s, bool n1_done = false, const LNorm* next_n1 = nullptr,
              bool* next_done = nullptr) {
    const int D = h->cfg.d_model, heads = D / 64, rows = B * tokens;
    if (!n1_done) FS_TRY(run_norm(h, blk.n1, ws.X, ws.Xn, rows, tokens, 0, s));
    // Round 6: the qkv Linear writes the attention's K / V^T operand planes from its epilogue (conv_igemm.hip, ConvParams::kv_k): the
    // pre-pass that read the fp32 rows back is gone (one launch and 31 MB per block); bit-identical planes.  Needs the split route and
    // column tiles of 96 that do not straddle q | k | v (d_model % 96 == 0: 384, 768).
    const bool fused_qkv = h->use_fused_qkv && ws.kv && D % 96 == 0;
    if (fused_qkv) FS_TRY(run_linear_qkv(h, blk.qkv, ws.Xn, B, tokens, ws.QKV, ws.kv, s));
    else FS_TRY(run_linear(h, blk.qkv, ws.Xn, rows, ws.QKV, nullptr, 0, s));
    const double aflops = 4.0 * B * heads * (double)tokens * tokens * 64;
    FS_TRY(prof_begin(h, blk.qkv.name + ".attention", ws.kv ? "attention_split" : "attention_f32", aflops, 4.0 * rows * 4.0 * D, s));
    if (ws.kv) FS_TRY(launch_attention_split(ws.QKV, ws.A, B, tokens, heads, 0.125f, ws.att, ws.kv, s, fused_qkv));
    else FS_TRY(launch_attention_f32(ws.QKV, ws.A, B, tokens, heads, 0.125f, ws.att, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(run_linear(h, blk.proj, ws.A, rows, ws.X, ws.X, 0, s, ws.part, tokens));  // x = x + proj(attn)
    FS_TRY(run_norm(h, blk.n2, ws.X, ws.Xn, rows, tokens, 0, s));
    FS_TRY(run_linear(h, blk.fc1, ws.Xn, rows, ws.Hd, nullptr, 2, s));       // GELU
    FS_TRY(run_linear(h, blk.fc2, ws.Hd, rows, ws.X, ws.X, 0, s, ws.part, tokens, next_n1, ws.Xn, next_done));  // x = x + mlp(x) (+ the next norm1)
    return 0;
}

// a stack of blocks on X: each block's fc2 merge also produces the next block's norm1 where it can
int run_blocks(fs_net* h, const std::vector<VitBlock>& blocks, const VitWs& ws, int B, int tokens, hipStream_t s) {
    bool have_n1 = false;
    for (size_t i = 0; i < blocks.size(); ++i) {
        bool next = false;
        FS_TRY(run_block(h, blocks[i], ws, B, tokens, s, have_n1, i + 1 < blocks.size() ? &blocks[i + 1].n1 : nullptr, &next));
        have_n1 = next;
    }
    return 0;
}

}  // namespace

int vit_finalize(fs_handle h, hipStream_t s) {
    const fs_config& c = h->cfg;
    const int D = c.d_model, P = c.patch, K = c.classes;
    FS_REQUIRE(D % 64 == 0 && D >= 64 && D <= 1024, "segmenter: d_model=%d must be a multiple of 64 (head_dim 64), <= 1024", D);
    FS_REQUIRE(P >= 4 && P % 4 == 0 && (3 * P * P) % 32 == 0, "segmenter: unsupported patch size %d", P);
    FS_REQUIRE(c.image_size % P == 0 && c.image_size >= P, "segmenter: image_size %d not divisible by patch %d", c.image_size, P);
    FS_REQUIRE(c.n_layers >= 1 && c.dec_layers >= 1, "segmenter: layer counts must be positive");
    h->pos_g0 = c.image_size / P;
    FS_TRY(make_linear(h, h->patch_embed, "encoder.patch_embed.proj", 3 * P * P, D, true, s));
    FS_TRY(copy_param(h, "encoder.cls_token", D, &h->cls_token));
    FS_TRY(copy_param(h, "encoder.pos_embed", (int64_t)(1 + h->pos_g0 * h->pos_g0) * D, &h->pos_embed));
    h->enc_blocks.resize((size_t)c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) FS_TRY(make_block(h, h->enc_blocks[(size_t)i], "encoder.blocks." + std::to_string(i) + ".", D, s));
    FS_TRY(make_norm(h, h->enc_norm, "encoder.norm", D));
    h->dec_blocks.resize((size_t)c.dec_layers);
    for (int i = 0; i < c.dec_layers; ++i) FS_TRY(make_block(h, h->dec_blocks[(size_t)i], "decoder.blocks." + std::to_string(i) + ".", D, s));
    FS_TRY(make_linear(h, h->proj_dec, "decoder.proj_dec", D, D, true, s));
    FS_TRY(make_norm(h, h->dec_norm, "decoder.decoder_norm", D));
    FS_TRY(make_norm(h, h->mask_norm, "decoder.mask_norm", K));
    FS_TRY(copy_param(h, "decoder.cls_emb", (int64_t)K * D, &h->cls_emb));
    // `patches @ proj_patch` uses the parameter as [in][out]: store the transpose so it is a Linear weight
    for (int which = 0; which < 2; ++which) {
        Linear& l = which ? h->proj_classes : h->proj_patch;
        const std::string name = which ? "decoder.proj_classes" : "decoder.proj_patch";
        const RawTensor* t;
        FS_TRY(fetch(h, name, &t));
        FS_REQUIRE(t->numel() == (int64_t)D * D, "'%s' must be [%d,%d]", name.c_str(), D, D);
        l.name = name;
        l.in = l.out = D;
        FS_TRY(dev_alloc(h, &l.w, (size_t)D * D));
        FS_TRY(launch_nchw_to_nhwc(t->d, l.w, D, 1, D, D, s));  // [in][out] -> [out][in]
        FS_TRY(split_attach(h, l.w, (size_t)D * D, s));
    }
    FS_HIP(hipStreamSynchronize(s));
    return 0;
}

int vit_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw) {
    const int P = h->cfg.patch;
    if (C) *C = h->cfg.d_model;
    if (fh) *fh = (H + P - 1) / P;
    if (fw) *fw = (W + P - 1) / P;
    return 0;
}

// position embedding for a gh x gw grid (segm/model/vit.py:122-130, utils.py:22-40: bilinear, align_corners=False); the
// resized table is kept until another geometry asks (fs_reserve builds it ahead of the first forward)
static int vit_pos_embed(fs_net* h, int gh, int gw, hipStream_t s, const float** pos) {
    const int D = h->cfg.d_model, N = gh * gw;
    *pos = h->pos_embed;
    if (gh == h->pos_g0 && gw == h->pos_g0) return 0;
    if (gh != h->pos_gh || gw != h->pos_gw) {
        FS_TRY(ws_grow(h, &h->pos_cur, &h->pos_elems, (size_t)(1 + N) * D, false));
        FS_HIP(hipMemcpyAsync(h->pos_cur, h->pos_embed, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, s));
        FS_TRY(launch_resize_bilinear_nhwc(h->pos_embed + D, D, 1, D, h->pos_g0, h->pos_g0, h->pos_cur + D, D, gh, gw, 0, s));
        h->pos_gh = gh;
        h->pos_gw = gw;
    }
    *pos = h->pos_cur;
    return 0;
}

int vit_reserve(fs_handle h, int B, int H, int W, hipStream_t s) {
    const int P = h->cfg.patch;
    const int gh = (H + P - 1) / P, gw = (W + P - 1) / P;
    VitWs ws;
    FS_TRY(vit_workspace(h, B, gh * gw + 1 + h->cfg.classes, &ws));
    const float* pos = nullptr;
    return vit_pos_embed(h, gh, gw, s, &pos);
}

int vit_encoder(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out_tokens, hipStream_t s) {
    FS_REQUIRE(out_tokens && B >= 1 && H >= 1 && W >= 1, "fs_encoder_forward(segmenter): bad arguments");
    FS_REQUIRE(src.ncrops == 0, "fs_segment_crops: the Segmenter has no sliding-crop route (the reference never wires ViT into flow/base.py:182-209)");
    const float* in_nchw = src.in;
    const float* in2 = src.in2;
    const int B1 = src.B1;
    const int P = h->cfg.patch, D = h->cfg.d_model;
    const int gh = (H + P - 1) / P, gw = (W + P - 1) / P, N = gh * gw;
    VitWs ws;
    FS_TRY(vit_workspace(h, B, N + 1 + h->cfg.classes, &ws));
    const float* pos = nullptr;
    FS_TRY(vit_pos_embed(h, gh, gw, s, &pos));
    FS_TRY(prof_begin(h, "patchify", "patchify", 0, 8.0 * B * N * 3.0 * P * P, s));
    FS_TRY(launch_patchify(in_nchw, in2, B1, ws.patches, B, H, W, P, gh, gw, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(run_linear(h, h->patch_embed, ws.patches, B * N, ws.emb, nullptr, 0, s));
    FS_TRY(launch_vit_assemble(ws.emb, h->cls_token, pos, ws.X, B, N, D, s));
    FS_TRY(run_blocks(h, h->enc_blocks, ws, B, N + 1, s));
    return run_norm(h, h->enc_norm, ws.X, out_tokens, B * (N + 1), N + 1, 1, s);  // final norm, cls token dropped
}

int vit_decoder(fs_handle h, const float* tokens, int B, int gh, int gw, float* out_nchw, hipStream_t s) {
    FS_REQUIRE(tokens && out_nchw && B >= 1 && gh >= 1 && gw >= 1, "fs_decoder_forward(segmenter): bad arguments");
    const int D = h->cfg.d_model, K = h->cfg.classes, N = gh * gw, T = N + K;
    VitWs ws;
    FS_TRY(vit_workspace(h, B, N + 1 + K, &ws));
    FS_TRY(run_linear(h, h->proj_dec, tokens, B * N, ws.emb, nullptr, 0, s));
    FS_TRY(launch_dec_assemble(ws.emb, h->cls_emb, ws.X, B, N, K, D, s));
    FS_TRY(run_blocks(h, h->dec_blocks, ws, B, T, s));
    FS_TRY(run_norm(h, h->dec_norm, ws.X, ws.Xn, B * T, T, 0, s));
    FS_TRY(run_linear(h, h->proj_patch, ws.Xn, B * T, ws.A, nullptr, 0, s));     // all rows; only the patch rows are read
    FS_TRY(run_linear(h, h->proj_classes, ws.Xn, B * T, ws.QKV, nullptr, 0, s)); // all rows; only the class rows are read
    FS_TRY(prof_begin(h, "mask_head", "mask_head", 2.0 * B * N * (double)K * D, 4.0 * B * N * (double)D, s));
    FS_TRY(launch_mask_head(ws.A, ws.QKV, h->mask_norm.g, h->mask_norm.b, out_nchw, B, N, K, D, s));
    return prof_end(h, s);
}

}  // namespace fs
