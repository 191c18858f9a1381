// Network plan + executor: dilated ResNet backbones with PSPNet / DeepLabv3 heads.
// Restates the module structure of model/resnet.py:60-165, model/pspnet.py:16-141 and
// (from the public torchvision definition, parity unpinned) model/deeplabv3.py:11-54.
#pragma once
#include <map>
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

// conv + (eval BatchNorm | bias) + optional ReLU, ready to launch
struct ConvBN {
    std::string name;
    float* w = nullptr;  // OHWI packed (HWIO for the Cin=3 stem)
    float* scale = nullptr;
    float* shift = nullptr;
    int Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0, dil = 1, relu = 0;
    int out_size(int in) const { return (in + 2 * pad - dil * (KH - 1) - 1) / stride + 1; }
};

struct Bottleneck {
    ConvBN c1, c2, c3, ds;
    bool has_ds = false;
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
    float* cls_w = nullptr;  // [K][512]
    float* cls_b = nullptr;
    // DeepLabv3 head
    fs::ConvBN aspp[4];    // 1x1, 3x3 d12, d24, d36
    fs::ConvBN aspp_pool;  // 1x1 on the global pool
    fs::ConvBN project, head_conv;
    int cls_cin = 512;

    // workspace
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    float* small = nullptr;  // pooled maps / tiny intermediates
    size_t small_elems = 0;

    // profiling
    bool profiling = false;
    std::vector<fs::ProfRec> prof;

    int feat_channels() const { return cfg.arch == FS_ARCH_PSPNET ? 4096 : 2048; }
};

namespace fs {
int net_create(const fs_config* cfg, fs_handle* out);
int net_destroy(fs_handle h);
int net_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                    hipStream_t s);
int net_finalize(fs_handle h, hipStream_t s);
int net_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw);
size_t net_workspace_bytes(fs_handle h, int B, int H, int W);
int net_encoder(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nhwc, hipStream_t s);
int net_decoder(fs_handle h, const float* feat, int B, int fh, int fw, float* out_nchw, hipStream_t s);
int net_profile_dump(fs_handle h, char* buf, size_t n);
}  // namespace fs
