// Network plan + executor: dilated ResNet backbones with PSPNet / DeepLabv3 heads.
// Restates the module structure of model/resnet.py:60-165, model/pspnet.py:16-141 and
// (from the public torchvision definition, parity unpinned) model/deeplabv3.py:11-54.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/floodseg.h"
#include "kernels.h"

namespace fs {

struct RawTensor {
    float* d = nullptr;  // device copy owned by the net until finalize()
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// Winograd filter banks of one conv, built on first use for the tile size the map at hand selects (a handle that only ever
// sees one geometry never pays for the other bank).  Shared by the copies of a ConvBN the executor makes.
struct WinoBank {
    float* U4 = nullptr;  // [36][Cout][Cin]  F(4x4,3x3)
    float* U6 = nullptr;  // [64][Cout][Cin]  F(6x6,3x3)
    float* U3 = nullptr;  // [25][Cout][Cin]  F(3x3,3x3)
    // recorded on the stream the bank was built on, right behind the filter transform; [0] = U4, [1] = U6, [2] = U3.  A forward on any
    // other stream waits for it before its first read (the events belong to the handle: fs_net::bank_events)
    hipEvent_t ready[3] = {nullptr, nullptr, nullptr};
    hipStream_t built_on[3] = {nullptr, nullptr, nullptr};
};

// conv + (eval BatchNorm | bias) + optional ReLU, ready to launch
struct ConvBN {
    std::string name;
    float* w = nullptr;  // OHWI packed (HWIO for the Cin=3 stem)
    float* scale = nullptr;
    float* shift = nullptr;
    int Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0, dil = 1, relu = 0;
    int korder = 0;  // 1: filters packed chunk-major (3x3 convs)
    std::shared_ptr<WinoBank> wino;  // non-null: 3x3 stride-1 conv with pad == dil and Cin >= 256 (Winograd-eligible)
    float* wf = nullptr;  // non-null: packed F(4,3) bank [36][Cin/16][Cout][16] of the one-kernel Winograd (wino_fused.hip): 3x3 s1 p1, Cin <= 128
    int out_size(int in) const { return (in + 2 * pad - dil * (KH - 1) - 1) / stride + 1; }
};

struct Bottleneck {
    ConvBN c1, c2, c3, ds;
    bool has_ds = false;
    // projection block: relu(bn3(conv3(f)) + bn_ds(downsample(x))) (model/resnet.py:86-94) as ONE GEMM over the concatenated K:
    // filters [Cout][Cin3 + Cin_ds] with the two BatchNorm scales folded into their halves, shift = shift3 + shift_ds
    ConvBN c3ds;
    int ds_cin = 0, ds_stride = 1;
};

// nn.Linear / nn.LayerNorm of the Segmenter (weights kept [out][in] exactly as torch stores them)
struct Linear {
    std::string name;
    float* w = nullptr;
    float* b = nullptr;
    int in = 0, out = 0;
};
struct LNorm {
    float* g = nullptr;
    float* b = nullptr;
    int D = 0;
};
struct VitBlock {
    LNorm n1, n2;
    Linear qkv, proj, fc1, fc2;
};

struct ProfRec {
    std::string name, kernel;
    double flops = 0, bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

}  // namespace fs

struct fs_net {
    fs_config cfg{};
    bool finalized = false;
    std::map<std::string, fs::RawTensor> raw;
    std::vector<float*> owned;  // every device allocation made for packed parameters

    // backbone
    bool deep_stem = true;  // semseg ResNet (3x 3x3) vs torchvision (7x7)
    fs::ConvBN stem[3];
    std::vector<fs::Bottleneck> blocks;
    std::vector<int> layer_end;  // index into blocks after each of the 4 stages
    // PSPNet head
    fs::ConvBN ppm[4];
    int bins[4] = {1, 2, 3, 6};
    fs::ConvBN cls_conv;  // 3x3 4096->512
    // fused encoder+decoder route (fs_segment_forward): the head conv over the 2048 backbone channels only, and the
    // [9*512][512] filter matrices that take the pooled pyramid maps straight to their share of the head conv (net_ops.hip)
    fs::ConvBN cls_main;
    fs::ConvBN ppm_z[4];
    float* seg_feat = nullptr;  // internal feature map for fs_segment_forward on the heads without a fused route
    size_t seg_feat_elems = 0;
    float* cls_w = nullptr;  // [K][512]
    float* cls_b = nullptr;
    // DeepLabv3 head
    fs::ConvBN aspp[4];    // 1x1, 3x3 d12, d24, d36
    fs::ConvBN aspp_pool;  // 1x1 on the global pool
    fs::ConvBN project, head_conv;
    int cls_cin = 512;

    // Segmenter (ViT encoder + mask transformer): segm/model/{vit,blocks,decoder}.py
    fs::Linear patch_embed;          // Conv2d(3, D, k=s=P) flattened to [D][3*P*P]
    float* cls_token = nullptr;      // [D]
    float* pos_embed = nullptr;      // [1 + g0*g0][D] as constructed
    int pos_g0 = 0;
    float* pos_cur = nullptr;        // [1 + gh*gw][D] resized for the current geometry
    int pos_gh = 0, pos_gw = 0;
    std::vector<fs::VitBlock> enc_blocks, dec_blocks;
    fs::LNorm enc_norm, dec_norm, mask_norm;
    fs::Linear proj_dec, proj_patch, proj_classes;  // proj_patch/_classes stored transposed ([out][in])
    float* cls_emb = nullptr;        // [K][D]
    float* vit_ws = nullptr;
    size_t vit_ws_elems = 0;
    float* wino_ws = nullptr;  // V [36][T][Cin] followed by M [36][T][Cout]
    size_t wino_ws_elems = 0;
    std::vector<hipEvent_t> bank_events;  // "bank built" events of the Winograd filter banks (WinoBank::ready)
    size_t bank_elems = 0;                // floats held by those banks
    size_t pos_elems = 0;                 // floats of pos_cur
    size_t ws_allocs = 0;                 // workspace / bank allocations made so far (none after fs_reserve, tests/test_gpu_net.py)
    // explicit options of fs_config (include/floodseg.h): nothing is read from the environment
    bool use_winograd = true;    // !(flags & FS_OPT_NO_WINOGRAD)
    int wino_force_m = 0;        // winograd_tile: 3 | 4 | 6 forces F(3,3) / F(4,3) / F(6,3); 0 = the cheapest one for the map at hand
    bool use_fused_head = true;  // !(flags & FS_OPT_NO_FUSED_HEAD)
    bool use_fused_shortcut = true;  // !(flags & FS_OPT_NO_FUSED_SHORTCUT)
    bool use_fused_winograd = true;  // !(flags & FS_OPT_NO_FUSED_WINOGRAD)
    bool use_split = true;           // !(flags & FS_OPT_NO_SPLIT_BF16): the implicit-GEMM launches take the split-operand kernel
    bool use_fused_qkv = true;       // !(flags & FS_OPT_NO_FUSED_QKV): the Segmenter's qkv Linear writes the attention's K / V^T planes in its epilogue
    bool use_fused_pool = true;      // !(flags & FS_OPT_NO_FUSED_POOL): layer0.6 + maxpool as one launch (deep stem, one-kernel Winograd route)
    bool res_touch = true;           // !(flags & FS_OPT_NO_RES_TOUCH): the conv kernels touch a shortcut tile's lines into L2 before their last K chunk
    // fp32 filter bank -> its three bf16 planes (split_bf16x3), keyed by the bank's first float; value = (planes, floats in the bank)
    std::map<const float*, std::pair<void*, size_t>> split_banks;
    int device = 0;              // HIP device the handle's memory lives on (current device at fs_create)

    // workspace
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    float* small = nullptr;  // pooled maps / tiny intermediates
    size_t small_elems = 0;

    // profiling
    bool profiling = false;
    std::vector<fs::ProfRec> prof;

    int feat_channels() const {
        if (cfg.arch == FS_ARCH_SEGMENTER) return cfg.d_model;
        return cfg.arch == FS_ARCH_PSPNET ? 4096 : 2048;
    }
};

namespace fs {
int net_create(const fs_config* cfg, fs_handle* out);
int net_destroy(fs_handle h);
int check_device(fs_net* h, const char* what);
int net_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                    hipStream_t s);
int net_finalize(fs_handle h, hipStream_t s);
int net_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw);
size_t net_workspace_bytes(fs_handle h, int B, int H, int W);
int net_reserve(fs_handle h, int B, int H, int W, hipStream_t s);
size_t net_reserved_bytes(fs_handle h);
// src: where the B frames live (kernels.h FrameSrc: one tensor, two tensors, or crop windows of two full frames)
int net_encoder(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out_nhwc, hipStream_t s);
int net_segment(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out_nchw, hipStream_t s);
int net_decoder(fs_handle h, const float* feat, int B, int fh, int fw, float* out_nchw, hipStream_t s);
int net_profile_dump(fs_handle h, char* buf, size_t n);

// shared with vit_net.hip
int dev_alloc(fs_net* h, float** p, size_t elems);
int ws_grow(fs_net* h, float** p, size_t* have, size_t need, bool zero);
// split-operand route: split_attach builds (on `s`, right behind the kernel that wrote the bank) and registers the three bf16 planes
// of a GEMM filter bank; split_use points a launch at them when its `wgt` lies inside a registered bank (no-op otherwise)
int split_attach(fs_net* h, const float* bank, size_t elems, hipStream_t s);
void split_use(const fs_net* h, ConvParams& p);
int vit_reserve(fs_handle h, int B, int H, int W, hipStream_t s);
int fetch(fs_net* h, const std::string& name, const RawTensor** out);
int prof_begin(fs_net* h, const std::string& name, const char* kernel, double flops, double bytes, hipStream_t s);
int prof_end(fs_net* h, hipStream_t s);
int vit_finalize(fs_handle h, hipStream_t s);
int vit_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw);
int vit_encoder(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out_tokens, hipStream_t s);
int vit_decoder(fs_handle h, const float* tokens, int B, int gh, int gw, float* out_nchw, hipStream_t s);
}  // namespace fs
