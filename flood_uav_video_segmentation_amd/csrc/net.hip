// Host-side executor: builds the layer plan from the loaded state_dict and runs it with the
// hand-written kernels.  No torch types; device memory is raw hipMalloc owned by the handle.
#include "net.h"

#include <cmath>
#include <cstdarg>
#include <cstring>

namespace fs {

std::string& last_error() {
    static thread_local std::string e;
    return e;
}
int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return 1;
}

int dev_alloc(fs_net* h, float** p, size_t elems) {
    FS_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(elems, 4) * sizeof(float)));
    h->owned.push_back(*p);
    return 0;
}

// The ONE place a forward may allocate: grows a library-owned workspace block (never shrinks).  Growing synchronises the
// device first -- work enqueued earlier may still use the old block -- which is why fs_reserve() exists: it does all the
// growing for a geometry up front, and no later forward at that geometry (or a smaller one) comes through the `need > have` arm.
int ws_grow(fs_net* h, float** p, size_t* have, size_t need, bool zero) {
    if (need <= *have) return 0;
    FS_HIP(hipDeviceSynchronize());
    if (*p) FS_HIP(hipFree(*p));
    *p = nullptr;
    *have = 0;
    FS_HIP(hipMalloc(reinterpret_cast<void**>(p), need * sizeof(float)));
    if (zero) FS_HIP(hipMemset(*p, 0, need * sizeof(float)));
    *have = need;
    ++h->ws_allocs;
    return 0;
}

int split_attach(fs_net* h, const float* bank, size_t elems, hipStream_t s) {
    if (!h->use_split || elems % 8 != 0 || h->split_banks.count(bank)) return 0;
    float* planes = nullptr;
    FS_TRY(dev_alloc(h, &planes, (3 * elems + 1) / 2));  // 3 planes of bf16
    FS_TRY(launch_split_bf16x3(bank, (long long)elems, planes, s));
    h->split_banks[bank] = {planes, elems};
    return 0;
}

void split_use(const fs_net* h, ConvParams& p) {
    p.wgt3 = nullptr;
    if (!h->use_split || p.Cin % 32 != 0 || h->split_banks.empty()) return;
    auto it = h->split_banks.upper_bound(p.wgt);
    if (it == h->split_banks.begin()) return;
    --it;
    const size_t off = (size_t)(p.wgt - it->first), elems = it->second.second;
    const int K = p.KH * p.KW * p.Cin + (p.in2 ? p.Cin2 : 0);
    if (off >= elems || off % 8 != 0 || (p.ld_wgt ? p.ld_wgt : K) % 8 != 0 || elems * 6 >= ((size_t)1 << 31)) return;
    p.wgt3 = (const char*)it->second.first + off * 2;
    p.plane_bytes = (unsigned)(elems * 2);
}

int fetch(fs_net* h, const std::string& name, const RawTensor** out) {
    auto it = h->raw.find(name);
    if (it == h->raw.end()) return fail("missing weight '%s'", name.c_str());
    *out = &it->second;
    return 0;
}

int to_host(const RawTensor& t, std::vector<float>& v) {
    v.resize((size_t)t.numel());
    FS_HIP(hipMemcpy(v.data(), t.d, v.size() * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

namespace {

bool wino_eligible(const ConvBN& c, bool hwio) {
    return !hwio && c.KH == 3 && c.KW == 3 && c.stride == 1 && c.pad == c.dil && c.Cin >= 128 && c.Cin % 32 == 0 && c.Cout % 4 == 0;
}

// The F(mt x mt, 3x3) filter bank of a Winograd-eligible conv, transformed (in double, rounded once) from the direct kernel's
// chunk-major bank the first time this tile size is needed.  Stream-ordered: the transform runs on `s` ahead of its first use.
// A forward on ANOTHER stream than the one the bank was built on waits for the build's event first (fs_reserve builds the
// banks ahead of time; without it the first forward that needs a bank allocates and builds it -- never under graph capture).
int wino_bank(fs_net* h, const ConvBN& c, int mt, hipStream_t s, const float** U) {
    const int k = mt == 6 ? 1 : mt == 3 ? 2 : 0;
    float*& slot = mt == 6 ? c.wino->U6 : mt == 3 ? c.wino->U3 : c.wino->U4;
    if (!slot) {
        FS_REQUIRE(c.korder == 1, "winograd: conv '%s' has no chunk-major filter bank", c.name.c_str());
        float* bank = nullptr;  // published only once the transform has been enqueued: a failed launch must not leave a half-built bank behind
        const size_t elems = (size_t)(mt + 2) * (mt + 2) * c.Cout * c.Cin;
        FS_TRY(dev_alloc(h, &bank, elems));
        FS_TRY(launch_winograd_filter(c.w, bank, c.Cout, c.Cin, mt, s, 1));
        FS_TRY(split_attach(h, bank, elems, s));
        hipEvent_t ev = nullptr;
        FS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        FS_HIP(hipEventRecord(ev, s));
        c.wino->ready[k] = ev;
        c.wino->built_on[k] = s;
        h->bank_events.push_back(ev);
        h->bank_elems += elems + (h->use_split ? (3 * elems + 1) / 2 : 0);
        ++h->ws_allocs;
        slot = bank;
    } else if (c.wino->built_on[k] != s && c.wino->ready[k]) {
        FS_HIP(hipStreamWaitEvent(s, c.wino->ready[k], 0));
    }
    *U = slot;
    return 0;
}

// conv weight + BatchNorm prefix (or bias name, or neither)
int make_conv(fs_net* h, ConvBN& c, const std::string& wname, const std::string& bn, const std::string& bias, int stride,
              int pad, int dil, int relu, bool hwio, hipStream_t s) {
    const RawTensor* w;
    FS_TRY(fetch(h, wname, &w));
    FS_REQUIRE(w->shape.size() == 4, "weight '%s' is not 4-D", wname.c_str());
    c.name = wname;
    c.Cout = (int)w->shape[0];
    c.Cin = (int)w->shape[1];
    c.KH = (int)w->shape[2];
    c.KW = (int)w->shape[3];
    c.stride = stride;
    c.pad = pad;
    c.dil = dil;
    c.relu = relu;
    FS_TRY(dev_alloc(h, &c.w, (size_t)w->numel()));
    if (hwio) {
        FS_TRY(launch_pack_oihw_to_hwio(w->d, c.w, c.Cout, c.Cin, c.KH, c.KW, s));
    } else if (c.KH * c.KW > 1 && c.Cin % 32 == 0) {
        c.korder = 1;
        FS_TRY(launch_pack_oihw_chunk_major(w->d, c.w, c.Cout, c.Cin, c.KH, c.KW, s));
    } else {
        FS_TRY(launch_pack_oihw_to_ohwi(w->d, c.w, c.Cout, c.Cin, c.KH, c.KW, s));
    }
    if (!hwio && c.Cin % 32 == 0) FS_TRY(split_attach(h, c.w, (size_t)w->numel(), s));
    if (wino_eligible(c, hwio)) c.wino = std::make_shared<WinoBank>();  // the banks themselves: wino_bank(), on first use
    if (!hwio && c.korder == 1 && c.Cin <= 128 && wino_fused_supported(c.Cin, c.Cout, c.KH, c.KW, c.stride, c.pad, c.dil)) {
        // few input channels (deep stem, conv2 of layer1 / layer2): the one-kernel Winograd's packed bank, at most 2.4 MB, built now
        FS_TRY(dev_alloc(h, &c.wf, wino_fused_bank_floats(c.Cin, c.Cout)));
        FS_TRY(launch_wino4_filter_packed(c.w, c.wf, c.Cout, c.Cin, s, 1));
    }
    if (!bn.empty()) {
        const RawTensor *g, *b, *m, *v;
        FS_TRY(fetch(h, bn + ".weight", &g));
        FS_TRY(fetch(h, bn + ".bias", &b));
        FS_TRY(fetch(h, bn + ".running_mean", &m));
        FS_TRY(fetch(h, bn + ".running_var", &v));
        FS_REQUIRE(g->numel() == c.Cout && b->numel() == c.Cout && m->numel() == c.Cout && v->numel() == c.Cout,
                   "BatchNorm '%s' does not match %d output channels", bn.c_str(), c.Cout);
        std::vector<float> hg, hb, hm, hv, sc((size_t)c.Cout), sh((size_t)c.Cout);
        FS_TRY(to_host(*g, hg));
        FS_TRY(to_host(*b, hb));
        FS_TRY(to_host(*m, hm));
        FS_TRY(to_host(*v, hv));
        for (int i = 0; i < c.Cout; ++i) {
            // ATen eval batch_norm: alpha = invstd * weight; beta = bias - mean * alpha  (eps = 1e-5)
            const float invstd = 1.0f / std::sqrt(hv[i] + 1e-5f);
            sc[i] = invstd * hg[i];
            sh[i] = hb[i] - hm[i] * sc[i];
        }
        FS_TRY(dev_alloc(h, &c.scale, (size_t)c.Cout));
        FS_TRY(dev_alloc(h, &c.shift, (size_t)c.Cout));
        FS_HIP(hipMemcpy(c.scale, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
        FS_HIP(hipMemcpy(c.shift, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
    } else if (!bias.empty()) {
        const RawTensor* b;
        FS_TRY(fetch(h, bias, &b));
        FS_REQUIRE(b->numel() == c.Cout, "bias '%s' does not match %d output channels", bias.c_str(), c.Cout);
        FS_TRY(dev_alloc(h, &c.shift, (size_t)c.Cout));
        FS_HIP(hipMemcpy(c.shift, b->d, (size_t)c.Cout * sizeof(float), hipMemcpyDeviceToDevice));
    }
    return 0;
}

}  // namespace

int prof_begin(fs_net* h, const std::string& name, const char* kernel, double flops, double bytes, hipStream_t s) {
    if (!h->profiling) return 0;
    ProfRec r;
    r.name = name;
    r.kernel = kernel;
    r.flops = flops;
    r.bytes = bytes;
    FS_HIP(hipEventCreate(&r.e0));
    FS_HIP(hipEventCreate(&r.e1));
    FS_HIP(hipEventRecord(r.e0, s));
    h->prof.push_back(r);
    return 0;
}
int prof_end(fs_net* h, hipStream_t s) {
    if (!h->profiling) return 0;
    FS_HIP(hipEventRecord(h->prof.back().e1, s));
    return 0;
}

namespace {

// Winograd workspace: V followed by M (fp32)
size_t wino_v_floats(size_t G, size_t T, int Cin) { return (G * T * Cin + 7) / 8 * 8; }
size_t wino_ws_floats(size_t G, size_t T, int Cin, int Cout) { return wino_v_floats(G, T, Cin) + G * T * Cout; }

// 3x3 s1 p1 conv as Winograd F(4x4,3x3): input transform -> 36 grouped GEMMs -> output transform (+BN, ReLU)
int run_conv_winograd(fs_net* h, const ConvBN& c, const float* in, int ld_in, int B, int H, int W, float* out, int ld_out,
                      hipStream_t s) {
    const int mt = h->wino_force_m ? h->wino_force_m : winograd_pick_m(B, H, W, c.dil);
    const int G = (mt + 2) * (mt + 2);
    const int T = winograd_tiles(B, H, W, c.dil, mt);
    const float* U = nullptr;
    FS_TRY(wino_bank(h, c, mt, s, &U));
    const size_t v_elems = (size_t)G * T * c.Cin, m_elems = (size_t)G * T * c.Cout;
    FS_TRY(ws_grow(h, &h->wino_ws, &h->wino_ws_elems, wino_ws_floats(G, T, c.Cin, c.Cout), false));
    float* V = h->wino_ws;
    float* Mb = h->wino_ws + wino_v_floats(G, T, c.Cin);
    ConvParams p{};
    p.in = V;
    p.ld_in = c.Cin;
    p.wgt = U;
    p.out = Mb;
    p.ld_out = c.Cout;
    p.B = 1;
    p.H = T;
    p.W = 1;
    p.Cin = c.Cin;
    p.Ho = T;
    p.Wo = 1;
    p.Cout = c.Cout;
    p.KH = p.KW = 1;
    p.stride = 1;
    p.dil = 1;
    p.groups = G;
    p.g_wgt = (long long)c.Cout * c.Cin;
    winograd_gemm_params(p, mt, T, c.Cin, c.Cout);  // V / M are tile-major when that fits a buffer descriptor (winograd.hip)
    split_use(h, p);
    const double flops = 2.0 * G * (double)T * c.Cin * c.Cout;
    FS_TRY(prof_begin(h, c.name + ".wino_in", "winograd_input", 0, 4.0 * ((double)B * H * W * c.Cin + (double)v_elems), s));
    FS_TRY(launch_winograd_input(in, ld_in, V, B, H, W, c.Cin, c.dil, mt, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(prof_begin(h, c.name + ".wino_gemm", conv_igemm_tile_name(p), flops, 4.0 * ((double)v_elems + (double)G * c.Cout * c.Cin + (double)m_elems), s));
    FS_TRY(launch_conv_igemm(p, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(prof_begin(h, c.name + ".wino_out", "winograd_output", 0, 4.0 * ((double)m_elems + (double)B * H * W * c.Cout), s));
    FS_TRY(launch_winograd_output(Mb, c.scale, c.shift, out, ld_out, B, H, W, c.Cout, c.relu, c.dil, mt, s));
    return prof_end(h, s);
}

// Winograd pays when its 36 GEMM rows per 4x4 tile undercut the 9 taps per pixel of the direct conv even after the
// tile-edge / lattice-phase waste (large dilations on a small map leave mostly-empty tiles): 36*T < 0.8 * 9*M
bool takes_winograd(const fs_net* h, const ConvBN& c, int B, int H, int W, bool has_res) {
    if (!(c.wino && h->use_winograd && !has_res)) return false;
    // Round 5: the 128-channel convs (conv2 of layer2) have both Winograd forms.  The one-kernel form (wino_fused.hip) is a serial chain of
    // Cin / 16 stages per workgroup: while one image gives it >= 1500 4x4 tiles it fills the chip several rounds deep and wins; on layer2's
    // 90 x 90 map (529 tiles per image: 134 workgroups for a key-frame pair, 8 stages each) the three-launch form is faster -- 34 vs 38 us
    // at B = 2, 26 vs 37 us at B = 1 (profiles/r05_experiments.txt section 10).  Decided on ONE image's geometry, never on the batch.
    if (c.Cin < 256 && c.wf && h->use_fused_winograd && (long)cdiv(H, 4) * cdiv(W, 4) >= 1500) return false;
    const int mt = h->wino_force_m ? h->wino_force_m : winograd_pick_m(B, H, W, c.dil);
    const double wino_rows = (double)(mt + 2) * (mt + 2) * winograd_tiles(B, H, W, c.dil, mt);
    const double direct_rows = 9.0 * (double)B * c.out_size(H) * c.out_size(W);
    // dilation <= 4 (the dilated ResNet stages, the heads): 0.8; the ASPP dilations leave the lattices of a 90x90 map
    // only 8, 4 and 3 pixels wide and the transforms touch every pixel through 2048 channels, so they must save more
    return wino_rows < (c.dil <= 4 ? 0.8 : 0.7) * direct_rows;
}

// fs_reserve: the Winograd workspace this conv needs at this geometry (0 = direct kernel), its filter bank built on `s`
int reserve_conv(fs_net* h, const ConvBN& c, int B, int H, int W, hipStream_t s, size_t* need) {
    if (!takes_winograd(h, c, B, H, W, false)) return 0;
    const int mt = h->wino_force_m ? h->wino_force_m : winograd_pick_m(B, H, W, c.dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)winograd_tiles(B, H, W, c.dil, mt);
    *need = std::max(*need, wino_ws_floats(G, T, c.Cin, c.Cout));
    const float* U = nullptr;
    return wino_bank(h, c, mt, s, &U);
}

// The one-kernel Winograd (wino_fused.hip) takes a 3x3 s1 p1 conv with Cin <= 128 when ONE image gives it enough 4x4 tiles to
// spread over the chip (decided per image, never on the batch: a frame's result must not depend on the batch it is computed in)
bool takes_fused_winograd(const fs_net* h, const ConvBN& c, int B, int H, int W, int ld_in, int ld_out, bool has_res) {
    // the kernel addresses its input with 30-bit and its output with 31-bit byte offsets (wino_fused.hip): maps beyond that
    // (batches of more than ~30 frames at 713^2) stay on the direct kernel -- a size limit, not a batch-dependent result: both
    // routes are within the parity tolerance of each other, and the key-frame cache never mixes such batches with small ones
    const bool fits = (int64_t)B * H * W * ld_in * 4 < ((int64_t)1 << 30) && (int64_t)B * H * W * ld_out * 4 < ((int64_t)1 << 31);
    return c.wf && h->use_fused_winograd && !has_res && fits && (long)cdiv(H, 4) * cdiv(W, 4) >= 500;
}

// launch parameters of one conv + (BatchNorm | bias) + activation (+ residual) on the implicit-GEMM kernel
ConvParams conv_params(const fs_net* h, const ConvBN& c, const float* in, int ld_in, int B, int H, int W, float* out, int ld_out, const float* res,
                       int ld_res) {
    ConvParams p{};
    p.in = in;
    p.ld_in = ld_in;
    p.wgt = c.w;
    p.scale = c.scale;
    p.shift = c.shift;
    p.res = res;
    p.ld_res = ld_res;
    p.out = out;
    p.ld_out = ld_out;
    p.B = B;
    p.H = H;
    p.W = W;
    p.Cin = c.Cin;
    p.Ho = c.out_size(H);
    p.Wo = c.out_size(W);
    p.Cout = c.Cout;
    p.KH = c.KH;
    p.KW = c.KW;
    p.stride = c.stride;
    p.pad = c.pad;
    p.dil = c.dil;
    p.relu = c.relu;
    p.korder = c.korder;
    p.res_touch = res != nullptr && h->res_touch;
    split_use(h, p);
    return p;
}

int run_conv(fs_net* h, const ConvBN& c, const float* in, int ld_in, int B, int H, int W, float* out, int ld_out,
             const float* res, int ld_res, hipStream_t s) {
    if (takes_winograd(h, c, B, H, W, res != nullptr)) return run_conv_winograd(h, c, in, ld_in, B, H, W, out, ld_out, s);
    if (takes_fused_winograd(h, c, B, H, W, ld_in, ld_out, res != nullptr)) {
        const double tiles = (double)B * cdiv(H, 4) * cdiv(W, 4);
        FS_TRY(prof_begin(h, c.name, "wino_fused", 2.0 * 36.0 * tiles * c.Cin * c.Cout,
                          4.0 * ((double)B * H * W * (c.Cin + c.Cout) + 36.0 * c.Cin * c.Cout), s));
        FS_TRY(launch_wino4_fused(in, ld_in, c.wf, c.scale, c.shift, out, ld_out, B, H, W, c.Cin, c.Cout, c.relu, s));
        return prof_end(h, s);
    }
    ConvParams p = conv_params(h, c, in, ld_in, B, H, W, out, ld_out, res, ld_res);
    const double M = (double)B * p.Ho * p.Wo;
    const double flops = 2.0 * M * c.Cout * c.KH * c.KW * c.Cin;
    const double bytes = 4.0 * ((double)B * H * W * c.Cin + (double)c.Cout * c.KH * c.KW * c.Cin + M * c.Cout * (res ? 2 : 1));
    FS_TRY(prof_begin(h, c.name, conv_igemm_tile_name(p), flops, bytes, s));
    FS_TRY(launch_conv_igemm(p, s));
    return prof_end(h, s);
}

int ensure_workspace(fs_net* h, size_t buf_elems, size_t small_elems) {
    if (buf_elems > h->buf_elems) {
        for (int i = 0; i < 4; ++i) {
            size_t have = h->buf_elems;
            FS_TRY(ws_grow(h, &h->buf[i], &have, buf_elems, false));
        }
        h->buf_elems = buf_elems;
    }
    // zeroed once: the fixed-stride pyramid slots have rows no kernel ever writes (levels with fewer than 36 cells) and
    // the grouped Z GEMM multiplies them like any other row
    return ws_grow(h, &h->small, &h->small_elems, small_elems, true);
}

struct Geometry {
    int H1, W1, H2, W2, H3, W3;  // after stem conv, after maxpool, after layer2 (stride 8 map)
};

Geometry geometry(const fs_net* h, int H, int W) {
    Geometry g;
    g.H1 = h->stem[0].out_size(H);
    g.W1 = h->stem[0].out_size(W);
    g.H2 = (g.H1 + 2 - 3) / 2 + 1;
    g.W2 = (g.W1 + 2 - 3) / 2 + 1;
    g.H3 = (g.H2 + 2 - 3) / 2 + 1;
    g.W3 = (g.W2 + 2 - 3) / 2 + 1;
    return g;
}

size_t encoder_buf_elems(const fs_net* h, int B, int H, int W) {
    const Geometry g = geometry(h, H, W);
    const size_t a = (size_t)B * g.H1 * g.W1 * 128;
    const size_t b = (size_t)B * g.H2 * g.W2 * 256;
    const size_t c = (size_t)B * g.H3 * g.W3 * 2048;
    return std::max(a, std::max(b, c));
}

// pooled maps (1+4+9+36 cells x 2048), then per pyramid level a fixed-stride slot of 36 cells: reduced (x512) and Z (x9*512)
size_t small_elems_for(int B) { return (size_t)B * (50 * 2048 + 4 * 36 * (512 + 9 * 512)) + 1024; }

}  // namespace

// ---------------------------------------------------------------------------------------------
int net_create(const fs_config* cfg, fs_handle* out) {
    FS_REQUIRE(cfg && out, "fs_create: null argument");
    FS_REQUIRE(cfg->arch == FS_ARCH_PSPNET || cfg->arch == FS_ARCH_DEEPLABV3 || cfg->arch == FS_ARCH_SEGMENTER,
               "fs_create: unknown arch %d", cfg->arch);
    FS_REQUIRE(cfg->arch == FS_ARCH_SEGMENTER || cfg->layers == 50 || cfg->layers == 101 || cfg->layers == 152,
               "fs_create: layers must be 50, 101 or 152");
    FS_REQUIRE(cfg->classes >= 1 && cfg->classes <= 255, "fs_create: classes out of range");
    FS_REQUIRE((cfg->flags & ~(FS_OPT_NO_WINOGRAD | FS_OPT_NO_FUSED_HEAD | FS_OPT_NO_FUSED_SHORTCUT | FS_OPT_NO_FUSED_WINOGRAD | FS_OPT_NO_SPLIT_BF16 |
                               FS_OPT_NO_RES_TOUCH | FS_OPT_NO_FUSED_POOL | FS_OPT_NO_FUSED_QKV)) == 0,
               "fs_create: unknown option bits 0x%x", cfg->flags);
    FS_REQUIRE(cfg->winograd_tile == 0 || cfg->winograd_tile == 3 || cfg->winograd_tile == 4 || cfg->winograd_tile == 6,
               "fs_create: winograd_tile must be 0, 3, 4 or 6");
    fs_net* h = new fs_net();
    h->cfg = *cfg;
    h->use_winograd = !(cfg->flags & FS_OPT_NO_WINOGRAD);
    h->wino_force_m = cfg->winograd_tile;
    h->use_fused_head = !(cfg->flags & FS_OPT_NO_FUSED_HEAD);
    h->use_fused_shortcut = !(cfg->flags & FS_OPT_NO_FUSED_SHORTCUT);
    h->use_fused_winograd = !(cfg->flags & (FS_OPT_NO_FUSED_WINOGRAD | FS_OPT_NO_WINOGRAD));
    h->use_split = !(cfg->flags & FS_OPT_NO_SPLIT_BF16);
    h->res_touch = !(cfg->flags & FS_OPT_NO_RES_TOUCH);
    h->use_fused_pool = !(cfg->flags & FS_OPT_NO_FUSED_POOL);
    h->use_fused_qkv = h->use_split && !(cfg->flags & FS_OPT_NO_FUSED_QKV);
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail("fs_create: no HIP device");
    }
    h->deep_stem = cfg->arch == FS_ARCH_PSPNET;
    *out = h;
    return 0;
}

// The handle's weights and workspace live on ONE device and every launch goes to the calling thread's current device:
// a call made while another device is current would run device-A kernels on device-B pointers.  Refuse it.
int check_device(fs_net* h, const char* what) {
    int cur = -1;
    FS_HIP(hipGetDevice(&cur));
    FS_REQUIRE(cur == h->device, "%s: the handle was created on HIP device %d but device %d is current", what, h->device, cur);
    return 0;
}

int net_destroy(fs_handle h) {
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    for (auto& kv : h->raw)
        if (kv.second.d) (void)hipFree(kv.second.d);
    for (float* p : h->owned) (void)hipFree(p);
    for (int i = 0; i < 4; ++i)
        if (h->buf[i]) (void)hipFree(h->buf[i]);
    if (h->small) (void)hipFree(h->small);
    if (h->vit_ws) (void)hipFree(h->vit_ws);
    if (h->seg_feat) (void)hipFree(h->seg_feat);
    if (h->wino_ws) (void)hipFree(h->wino_ws);
    if (h->pos_cur) (void)hipFree(h->pos_cur);
    for (hipEvent_t e : h->bank_events) (void)hipEventDestroy(e);
    for (auto& r : h->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    delete h;
    return 0;
}

int net_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                    hipStream_t s) {
    FS_REQUIRE(h && name && data && (shape || ndim == 0), "fs_load_weight: null argument");
    FS_REQUIRE(!h->finalized, "fs_load_weight: network already finalized");
    FS_TRY(check_device(h, "fs_load_weight"));
    RawTensor t;
    t.shape.assign(shape, shape + ndim);
    const int64_t n = t.numel();
    FS_REQUIRE(n > 0, "fs_load_weight: '%s' is empty", name);
    auto it = h->raw.find(name);
    if (it != h->raw.end()) {
        (void)hipFree(it->second.d);
        h->raw.erase(it);
    }
    FS_HIP(hipMalloc(reinterpret_cast<void**>(&t.d), (size_t)n * sizeof(float)));
    FS_HIP(hipMemcpyAsync(t.d, data, (size_t)n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    h->raw[name] = t;
    return 0;
}

int net_finalize(fs_handle h, hipStream_t s) {
    FS_REQUIRE(h, "fs_finalize: null handle");
    FS_REQUIRE(!h->finalized, "fs_finalize: already finalized");
    FS_TRY(check_device(h, "fs_finalize"));
    FS_HIP(hipStreamSynchronize(s));
    if (h->cfg.arch == FS_ARCH_SEGMENTER) {
        FS_TRY(vit_finalize(h, s));
        FS_HIP(hipDeviceSynchronize());
        for (auto& kv : h->raw) {
            (void)hipFree(kv.second.d);
            kv.second.d = nullptr;
        }
        h->raw.clear();
        h->finalized = true;
        return 0;
    }
    const bool psp = h->cfg.arch == FS_ARCH_PSPNET;
    const std::string bb = psp ? "" : "backbone.";
    // ---- stem
    if (psp) {
        FS_TRY(make_conv(h, h->stem[0], "layer0.0.weight", "layer0.1", "", 2, 1, 1, 1, true, s));
        FS_TRY(make_conv(h, h->stem[1], "layer0.3.weight", "layer0.4", "", 1, 1, 1, 1, false, s));
        FS_TRY(make_conv(h, h->stem[2], "layer0.6.weight", "layer0.7", "", 1, 1, 1, 1, false, s));
    } else {
        FS_TRY(make_conv(h, h->stem[0], "backbone.conv1.weight", "backbone.bn1", "", 2, 3, 1, 1, true, s));
    }
    FS_REQUIRE(h->stem[0].Cin == 3, "stem conv must take 3 input channels");
    // ---- residual stages
    int nblk[4] = {3, 4, 6, 3};
    if (h->cfg.layers == 101) nblk[2] = 23;
    if (h->cfg.layers == 152) { nblk[1] = 8; nblk[2] = 36; }
    h->blocks.clear();
    h->layer_end.clear();
    for (int L = 1; L <= 4; ++L) {
        for (int b = 0; b < nblk[L - 1]; ++b) {
            Bottleneck blk;
            const std::string pre = bb + "layer" + std::to_string(L) + "." + std::to_string(b) + ".";
            int stride = 1, dil = 1;
            if (L == 2 && b == 0) stride = 2;
            if (psp) {
                // model/pspnet.py:55-64: every conv2 of layer3/4 gets dilation 2/4, stride 1
                if (L == 3) dil = 2;
                if (L == 4) dil = 4;
            } else {
                // torchvision replace_stride_with_dilation=[False, True, True]
                if (L == 3) dil = b == 0 ? 1 : 2;
                if (L == 4) dil = b == 0 ? 2 : 4;
            }
            FS_TRY(make_conv(h, blk.c1, pre + "conv1.weight", pre + "bn1", "", 1, 0, 1, 1, false, s));
            FS_TRY(make_conv(h, blk.c2, pre + "conv2.weight", pre + "bn2", "", stride, dil, dil, 1, false, s));
            FS_TRY(make_conv(h, blk.c3, pre + "conv3.weight", pre + "bn3", "", 1, 0, 1, 1, false, s));
            blk.has_ds = b == 0;
            if (blk.has_ds) {
                FS_TRY(make_conv(h, blk.ds, pre + "downsample.0.weight", pre + "downsample.1", "", stride, 0, 1, 0, false, s));
                if (blk.c3.KH == 1 && blk.ds.KH == 1 && blk.c3.Cin % 32 == 0 && blk.ds.Cin % 32 == 0 && blk.c3.scale && blk.ds.scale) {
                    ConvBN& f = blk.c3ds;
                    f.name = pre + "conv3+downsample";
                    f.Cin = blk.c3.Cin;
                    f.Cout = blk.c3.Cout;
                    f.relu = 1;
                    blk.ds_cin = blk.ds.Cin;
                    blk.ds_stride = blk.ds.stride;
                    FS_TRY(dev_alloc(h, &f.w, (size_t)f.Cout * (blk.c3.Cin + blk.ds.Cin)));
                    FS_TRY(dev_alloc(h, &f.shift, (size_t)f.Cout));
                    FS_TRY(launch_concat_scaled_filters(blk.c3.w, blk.c3.scale, blk.c3.shift, blk.c3.Cin, blk.ds.w, blk.ds.scale, blk.ds.shift,
                                                        blk.ds.Cin, f.w, f.shift, f.Cout, s));
                    FS_TRY(split_attach(h, f.w, (size_t)f.Cout * (blk.c3.Cin + blk.ds.Cin), s));
                }
            }
            h->blocks.push_back(blk);
        }
        h->layer_end.push_back((int)h->blocks.size());
    }
    // ---- heads
    const int K = h->cfg.classes;
    if (psp) {
        for (int i = 0; i < 4; ++i) {
            const std::string pre = "ppm.features." + std::to_string(i) + ".";
            FS_TRY(make_conv(h, h->ppm[i], pre + "1.weight", pre + "2", "", 1, 0, 1, 1, false, s));
            FS_REQUIRE(h->ppm[i].Cin == 2048 && h->ppm[i].Cout == 512, "unexpected PPM conv shape");
        }
        FS_TRY(make_conv(h, h->cls_conv, "decoder.0.weight", "decoder.1", "", 1, 1, 1, 1, false, s));
        FS_REQUIRE(h->cls_conv.Cin == 4096, "decoder.0 must take 4096 channels");
        h->cls_cin = h->cls_conv.Cout;
        {   // fused route: backbone half of the head conv (own packed filters, BatchNorm shared) + pyramid filter matrices
            const RawTensor* w;
            FS_TRY(fetch(h, "decoder.0.weight", &w));
            const int O = h->cls_conv.Cout;
            ConvBN& m = h->cls_main;
            m = h->cls_conv;
            m.name = "decoder.0.weight[:, :2048]";
            m.Cin = 2048;
            m.w = nullptr;
            m.wino = std::make_shared<WinoBank>();  // its own banks (Cin = 2048), built on first use like every other conv's
            // In the chunk-major layout [O][I/32][3][3][32] the first 2048 input channels of output channel o are the first
            // 64 chunk blocks of its row: the backbone half of the bank is a strided copy of the full one, no repack.
            FS_TRY(dev_alloc(h, &m.w, (size_t)O * 2048 * 9));
            m.korder = 1;
            FS_REQUIRE(h->cls_conv.korder == 1, "decoder.0: chunk-major bank expected");
            FS_HIP(hipMemcpy2DAsync(m.w, (size_t)2048 * 9 * sizeof(float), h->cls_conv.w, (size_t)4096 * 9 * sizeof(float),
                                    (size_t)2048 * 9 * sizeof(float), (size_t)O, hipMemcpyDeviceToDevice, s));
            FS_TRY(split_attach(h, m.w, (size_t)O * 2048 * 9, s));
            float* zw = nullptr;  // the four [9*O][512] matrices back to back: groups of one grouped GEMM
            FS_TRY(dev_alloc(h, &zw, (size_t)4 * 9 * O * 512));
            for (int i = 0; i < 4; ++i) {
                ConvBN& z = h->ppm_z[i];
                z.name = "decoder.0.weight[:, ppm" + std::to_string(h->bins[i]) + "]";
                z.Cin = 512;
                z.Cout = 9 * O;
                z.w = zw + (size_t)i * 9 * O * 512;
                FS_TRY(launch_pack_slice_tap_major(w->d, z.w, O, 4096, 2048 + 512 * i, 512, 9, s));
            }
            FS_TRY(split_attach(h, zw, (size_t)4 * 9 * O * 512, s));
        }
        const RawTensor *w, *b;
        FS_TRY(fetch(h, "decoder.4.weight", &w));
        FS_TRY(fetch(h, "decoder.4.bias", &b));
        FS_REQUIRE(w->shape.size() == 4 && w->shape[0] == K && w->shape[1] == h->cls_cin, "decoder.4.weight has wrong shape");
        FS_TRY(dev_alloc(h, &h->cls_w, (size_t)w->numel()));
        FS_TRY(dev_alloc(h, &h->cls_b, (size_t)K));
        FS_HIP(hipMemcpy(h->cls_w, w->d, (size_t)w->numel() * sizeof(float), hipMemcpyDeviceToDevice));
        FS_HIP(hipMemcpy(h->cls_b, b->d, (size_t)K * sizeof(float), hipMemcpyDeviceToDevice));
    } else {
        const int rates[4] = {0, 12, 24, 36};
        FS_TRY(make_conv(h, h->aspp[0], "classifier.0.convs.0.0.weight", "classifier.0.convs.0.1", "", 1, 0, 1, 1, false, s));
        for (int i = 1; i < 4; ++i) {
            const std::string pre = "classifier.0.convs." + std::to_string(i) + ".";
            FS_TRY(make_conv(h, h->aspp[i], pre + "0.weight", pre + "1", "", 1, rates[i], rates[i], 1, false, s));
        }
        FS_TRY(make_conv(h, h->aspp_pool, "classifier.0.convs.4.1.weight", "classifier.0.convs.4.2", "", 1, 0, 1, 1, false, s));
        FS_TRY(make_conv(h, h->project, "classifier.0.project.0.weight", "classifier.0.project.1", "", 1, 0, 1, 1, false, s));
        FS_TRY(make_conv(h, h->head_conv, "classifier.1.weight", "classifier.2", "", 1, 1, 1, 1, false, s));
        h->cls_cin = h->head_conv.Cout;
        const RawTensor *w, *b;
        FS_TRY(fetch(h, "classifier.4.weight", &w));
        FS_TRY(fetch(h, "classifier.4.bias", &b));
        FS_REQUIRE(w->shape.size() == 4 && w->shape[0] == K && w->shape[1] == h->cls_cin, "classifier.4.weight has wrong shape");
        FS_TRY(dev_alloc(h, &h->cls_w, (size_t)w->numel()));
        FS_TRY(dev_alloc(h, &h->cls_b, (size_t)K));
        FS_HIP(hipMemcpy(h->cls_w, w->d, (size_t)w->numel() * sizeof(float), hipMemcpyDeviceToDevice));
        FS_HIP(hipMemcpy(h->cls_b, b->d, (size_t)K * sizeof(float), hipMemcpyDeviceToDevice));
    }
    FS_HIP(hipStreamSynchronize(s));
    FS_HIP(hipDeviceSynchronize());
    for (auto& kv : h->raw) {
        (void)hipFree(kv.second.d);
        kv.second.d = nullptr;
    }
    h->raw.clear();
    h->finalized = true;
    return 0;
}

int net_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw) {
    FS_REQUIRE(h && h->finalized, "fs_feature_shape: network not finalized");
    if (h->cfg.arch == FS_ARCH_SEGMENTER) return vit_feature_shape(h, H, W, C, fh, fw);
    const Geometry g = geometry(h, H, W);
    if (C) *C = h->feat_channels();
    if (fh) *fh = g.H3;
    if (fw) *fw = g.W3;
    return 0;
}

size_t net_workspace_bytes(fs_handle h, int B, int H, int W) {
    if (!h || !h->finalized) return 0;
    if (h->cfg.arch == FS_ARCH_SEGMENTER) {
        const int P = h->cfg.patch, D = h->cfg.d_model;
        const size_t T = (size_t)B * (((H + P - 1) / P) * ((W + P - 1) / P) + 1 + h->cfg.classes);
        return (T * D * 15 + T * 3 * P * P + 64) * sizeof(float);
    }
    return (4 * encoder_buf_elems(h, B, H, W) + small_elems_for(B)) * sizeof(float);
}

// fs_reserve: every library-owned block a forward over B frames of H x W touches -- activation buffers, pooled maps, the
// internal feature map of the unfused routes, the Winograd V / M workspace of the largest eligible conv -- is grown to its
// final size now, and the Winograd filter banks this geometry selects are built on `s`.  A later fs_encoder_forward /
// fs_decoder_forward / fs_segment_forward / fs_segment_crops with the same H x W and at most B frames allocates nothing.
int net_reserve(fs_handle h, int B, int H, int W, hipStream_t s) {
    FS_REQUIRE(h && h->finalized, "fs_reserve: network not finalized");
    FS_TRY(check_device(h, "fs_reserve"));
    FS_REQUIRE(B >= 1 && H >= 1 && W >= 1, "fs_reserve: bad arguments (B=%d H=%d W=%d)", B, H, W);
    if (h->cfg.arch == FS_ARCH_SEGMENTER) return vit_reserve(h, B, H, W, s);
    FS_REQUIRE(H >= 33 && W >= 33, "fs_reserve: frames of at least 33 x 33 (got %d x %d)", H, W);
    const Geometry g = geometry(h, H, W);
    const int fh = g.H3, fw = g.W3;
    const size_t px = (size_t)B * fh * fw;
    const bool psp = h->cfg.arch == FS_ARCH_PSPNET;
    size_t buf = encoder_buf_elems(h, B, H, W);
    buf = std::max(buf, psp ? std::max(px * 512, ppm_term_scratch_floats(B, fh, h->cls_main.Cout)) : px * 1280);
    FS_TRY(ensure_workspace(h, buf, small_elems_for(B)));
    if (!psp || !h->use_fused_head) FS_TRY(ws_grow(h, &h->seg_feat, &h->seg_feat_elems, px * (size_t)h->feat_channels(), false));
    size_t wino = 0;
    if (h->deep_stem) {
        FS_TRY(reserve_conv(h, h->stem[1], B, g.H1, g.W1, s, &wino));
        FS_TRY(reserve_conv(h, h->stem[2], B, g.H1, g.W1, s, &wino));
    }
    int curH = g.H2, curW = g.W2;
    for (const Bottleneck& blk : h->blocks) {
        FS_TRY(reserve_conv(h, blk.c2, B, curH, curW, s, &wino));
        curH = blk.c2.out_size(curH);
        curW = blk.c2.out_size(curW);
    }
    if (psp) {
        FS_TRY(reserve_conv(h, h->use_fused_head ? h->cls_main : h->cls_conv, B, fh, fw, s, &wino));
        FS_TRY(reserve_conv(h, h->cls_conv, B, fh, fw, s, &wino));  // fs_decoder_forward on its own (feature mode)
    } else {
        for (int i = 1; i < 4; ++i) FS_TRY(reserve_conv(h, h->aspp[i], B, fh, fw, s, &wino));
        FS_TRY(reserve_conv(h, h->head_conv, B, fh, fw, s, &wino));
    }
    return ws_grow(h, &h->wino_ws, &h->wino_ws_elems, wino, false);
}

size_t net_reserved_bytes(fs_handle h) {
    if (!h) return 0;
    return (4 * h->buf_elems + h->small_elems + h->seg_feat_elems + h->wino_ws_elems + h->vit_ws_elems + h->pos_elems + h->bank_elems) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
namespace {
// Pyramid pooling up to the reduced maps (model/pspnet.py:22-26): adaptive average pools (bins 1, 2, 3, 6) of the 2048
// backbone channels -> 1x1 conv + BN + ReLU.  Layout in h->small: pooled maps back to back (1 + 4 + 9 + 36 cells per
// image x 2048), then per level a fixed-stride slot of B*36 rows x 512 (the fused head's grouped Z GEMM wants one stride).
int pyramid_reduce(fs_net* h, const float* feat, int ld_feat, int B, int H, int W, hipStream_t s) {
    float* pooled = h->small;
    float* reduced = h->small + (size_t)B * 50 * 2048;
    size_t pool_off[4];
    {
        size_t po = 0;
        for (int i = 0; i < 4; ++i) {
            pool_off[i] = po;
            po += (size_t)B * h->bins[i] * h->bins[i] * 2048;
        }
    }
    const bool even = H % 6 == 0 && W % 6 == 0 && h->bins[0] == 1 && h->bins[1] == 2 && h->bins[2] == 3 && h->bins[3] == 6;
    if (even) {  // one pass over the 2048-channel map instead of four
        FS_TRY(prof_begin(h, "ppm.pool6+combine", "adaptive_avgpool", 0, 4.0 * B * H * W * 2048.0, s));
        FS_TRY(launch_adaptive_avgpool(feat, ld_feat, pooled + pool_off[3], B, H, W, 2048, 6, s));
        FS_TRY(launch_ppm_pool_combine(pooled + pool_off[3], pooled + pool_off[0], pooled + pool_off[1], pooled + pool_off[2], B, 2048, s));
        FS_TRY(prof_end(h, s));
    }
    RowdotBatch pb{};
    double flops = 0;
    for (int i = 0; i < 4; ++i) {
        const int bin = h->bins[i];
        const int cells = bin * bin;
        if (!even) {
            FS_TRY(prof_begin(h, "ppm.pool" + std::to_string(bin), "adaptive_avgpool", 0, 4.0 * B * H * W * 2048.0, s));
            FS_TRY(launch_adaptive_avgpool(feat, ld_feat, pooled + pool_off[i], B, H, W, 2048, bin, s));
            FS_TRY(prof_end(h, s));
        }
        const ConvBN& c = h->ppm[i];
        pb.p[i] = RowdotProblem{pooled + pool_off[i], c.w, c.scale, c.shift, reduced + (size_t)i * B * 36 * 512, B * cells};
        flops += 2.0 * B * cells * 2048.0 * 512.0;
    }
    // the four levels' 1x1 conv + BN + ReLU (model/pspnet.py:23-25) as ONE launch: M = B, 4B, 9B, 36B rows of the same (K, N)
    FS_TRY(prof_begin(h, "ppm.features.*.1.weight", "rowdot_1x1", flops, 4.0 * 4 * 2048.0 * 512.0, s));
    FS_TRY(launch_rowdot_1x1_batch(pb, 4, 2048, 512, 2048, 512, 1, s));
    FS_TRY(prof_end(h, s));
    return 0;
}

// ResNet backbone (+ pyramid pooling for PSPNet).  out != nullptr: the reference's encoder output (PSPNet: 4096-channel
// concat with the upsampled pyramid; ld_out = feat_channels()).  out == nullptr (fused PSPNet route): the 2048 backbone
// channels stay in a workspace buffer (*feat2048, ld 2048) and the pyramid stops at the pooled+reduced maps in h->small.
int encoder_core(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out, float** feat2048, hipStream_t s) {
    const bool fused = out == nullptr;
    const Geometry g = geometry(h, H, W);
    FS_TRY(ensure_workspace(h, encoder_buf_elems(h, B, H, W), small_elems_for(B)));
    const bool psp = h->cfg.arch == FS_ARCH_PSPNET;
    float *X = h->buf[0], *F1 = h->buf[1], *F2 = h->buf[2], *F3 = h->buf[3];

    // ---- stem
    {
        const ConvBN& c = h->stem[0];
        StemParams p{};
        p.src = src;
        p.wgt = c.w;
        p.scale = c.scale;
        p.shift = c.shift;
        p.out = X;
        p.ld_out = c.Cout;
        p.B = B;
        p.H = H;
        p.W = W;
        p.Ho = g.H1;
        p.Wo = g.W1;
        p.Cout = c.Cout;
        p.KH = c.KH;
        p.KW = c.KW;
        p.stride = c.stride;
        p.pad = c.pad;
        p.split = h->use_split;
        const double M = (double)B * g.H1 * g.W1;
        FS_TRY(prof_begin(h, c.name, h->use_split ? "stem_conv_split" : "stem_conv", 2.0 * M * c.Cout * c.KH * c.KW * 3, 4.0 * (B * 3.0 * H * W + M * c.Cout), s));
        FS_TRY(launch_stem_conv(p, s));
        FS_TRY(prof_end(h, s));
    }
    int C = h->stem[0].Cout;
    bool pooled = false;
    if (h->deep_stem) {
        FS_TRY(run_conv(h, h->stem[1], X, C, B, g.H1, g.W1, F1, h->stem[1].Cout, nullptr, 0, s));
        const ConvBN& c = h->stem[2];
        // Round 5: layer0.6 + BN + ReLU + the max-pool behind it as ONE launch whenever the conv takes the one-kernel Winograd (decided
        // per image geometry like that route itself): the 357 x 357 x 128 map -- the largest tensor of the network -- is never written
        pooled = h->use_fused_pool && c.relu == 1 && c.scale && takes_fused_winograd(h, c, B, g.H1, g.W1, h->stem[1].Cout, c.Cout, false) &&
                 g.H2 == (g.H1 - 1) / 2 + 1 && g.W2 == (g.W1 - 1) / 2 + 1;
        if (pooled) {
            const double tiles = (double)B * cdiv(g.H1, 4) * cdiv(g.W1, 4);
            FS_TRY(prof_begin(h, c.name + "+maxpool", "wino_fused_pool", 2.0 * 36.0 * tiles * c.Cin * c.Cout,
                              4.0 * ((double)B * g.H1 * g.W1 * c.Cin + 2.0 * (double)B * g.H2 * g.W2 * c.Cout + 36.0 * c.Cin * c.Cout), s));
            FS_TRY(launch_wino4_fused_pool(F1, h->stem[1].Cout, c.wf, c.scale, c.shift, X, c.Cout, B, g.H1, g.W1, c.Cin, c.Cout, s));
            FS_TRY(prof_end(h, s));
        } else {
            FS_TRY(run_conv(h, c, F1, h->stem[1].Cout, B, g.H1, g.W1, X, c.Cout, nullptr, 0, s));
        }
        C = c.Cout;
    }
    if (!pooled) {
        const double M = (double)B * g.H2 * g.W2;
        FS_TRY(prof_begin(h, "maxpool", "maxpool3x3s2", 0, 4.0 * ((double)B * g.H1 * g.W1 * C + M * C), s));
        FS_TRY(launch_maxpool3x3s2(X, C, F1, C, B, g.H1, g.W1, C, g.H2, g.W2, s));
        FS_TRY(prof_end(h, s));
        std::swap(X, F1);
    }
    // ---- residual stages.  X holds the block input; F1..F3 are free.
    int curH = g.H2, curW = g.W2;
    const int nblocks = (int)h->blocks.size();
    for (int bi = 0; bi < nblocks; ++bi) {
        const Bottleneck& blk = h->blocks[bi];
        const bool last = bi == nblocks - 1;
        const int oH = blk.c2.out_size(curH), oW = blk.c2.out_size(curW);
        FS_TRY(run_conv(h, blk.c1, X, C, B, curH, curW, F1, blk.c1.Cout, nullptr, 0, s));
        FS_TRY(run_conv(h, blk.c2, F1, blk.c1.Cout, B, curH, curW, F2, blk.c2.Cout, nullptr, 0, s));
        const int Cn = blk.c3.Cout;
        float* dst = last ? (fused ? F1 : out) : nullptr;  // F1 (conv1's output) is free again once conv2 has run
        const int ld_dst = last && !fused ? h->feat_channels() : Cn;
        ConvBN c3 = blk.c3;
        c3.relu = 1;  // ReLU after the residual add (model/resnet.py:93-94)
        if (blk.has_ds && blk.c3ds.w && h->use_fused_shortcut) {
            // conv3 and the projection shortcut as one launch over the concatenated K: the shortcut map is never written
            if (!dst) dst = F3;
            ConvParams p{};
            p.in = F2; p.ld_in = blk.c2.Cout; p.wgt = blk.c3ds.w; p.shift = blk.c3ds.shift; p.out = dst; p.ld_out = ld_dst;
            p.B = B; p.H = oH; p.W = oW; p.Cin = blk.c3ds.Cin; p.Ho = oH; p.Wo = oW; p.Cout = Cn;
            p.KH = p.KW = 1; p.stride = 1; p.dil = 1; p.relu = 1;
            p.in2 = X; p.ld_in2 = C; p.Cin2 = blk.ds_cin; p.stride2 = blk.ds_stride; p.H2 = curH; p.W2 = curW;
            split_use(h, p);
            const double M = (double)B * oH * oW;
            const double fl = 2.0 * M * Cn * (p.Cin + p.Cin2), by = 4.0 * (M * p.Cin + (double)B * curH * curW * p.Cin2 + (double)Cn * (p.Cin + p.Cin2) + M * Cn);
            FS_TRY(prof_begin(h, blk.c3ds.name, conv_igemm_tile_name(p), fl, by, s));
            FS_TRY(launch_conv_igemm(p, s));
            FS_TRY(prof_end(h, s));
            if (!last) std::swap(X, F3);
        } else if (blk.has_ds) {
            FS_TRY(run_conv(h, blk.ds, X, C, B, curH, curW, F3, Cn, nullptr, 0, s));
            if (!dst) dst = F3;  // in place over the shortcut
            FS_TRY(run_conv(h, c3, F2, blk.c2.Cout, B, oH, oW, dst, ld_dst, F3, Cn, s));
            if (!last) std::swap(X, F3);
        } else {
            if (!dst) dst = X;  // in place over the identity shortcut
            FS_TRY(run_conv(h, c3, F2, blk.c2.Cout, B, oH, oW, dst, ld_dst, X, C, s));
        }
        C = Cn;
        curH = oH;
        curW = oW;
    }
    FS_REQUIRE(curH == g.H3 && curW == g.W3 && C == 2048, "encoder geometry mismatch");
    if (fused) {
        *feat2048 = F1;
        out = F1;
    }
    if (!psp) return 0;
    if (fused) return 0;  // net_segment runs the pyramid itself (on its side stream) and never upsamples it
    // ---- pyramid pooling: pooled -> 1x1 conv + BN + ReLU -> bilinear(ac=True) into channels 2048+512*i
    FS_TRY(pyramid_reduce(h, out, 4096, B, curH, curW, s));
    float* reduced = h->small + (size_t)B * 50 * 2048;
    for (int i = 0; i < 4; ++i) {
        const int bin = h->bins[i];
        FS_TRY(prof_begin(h, "ppm.up" + std::to_string(bin), "upsample_into", 0, 4.0 * B * curH * curW * 512.0, s));
        FS_TRY(launch_upsample_into(reduced + (size_t)i * B * 36 * 512, bin, bin, out + 2048 + 512 * i, 4096, B, curH, curW, 512, 1, s));
        FS_TRY(prof_end(h, s));
    }
    return 0;
}
}  // namespace

namespace {
int check_frames(const FrameSrc& f, int B, const char* what) {
    if (f.ncrops) return 0;  // the stem launcher validates crop windows against the frame
    FS_REQUIRE(B >= 1 && f.B1 >= 0 && f.B1 <= B && (f.B1 == 0 || f.in) && (f.B1 == B || f.in2), "%s: bad frame batch (B1=%d of B=%d)", what, f.B1, B);
    return 0;
}
}  // namespace

int net_encoder(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out, hipStream_t s) {
    FS_REQUIRE(h && h->finalized, "fs_encoder_forward: network not finalized");
    FS_TRY(check_device(h, "fs_encoder_forward"));
    FS_TRY(check_frames(src, B, "fs_encoder_forward"));
    if (h->cfg.arch == FS_ARCH_SEGMENTER) return vit_encoder(h, src, B, H, W, out, s);
    FS_REQUIRE(out && H >= 33 && W >= 33, "fs_encoder_forward: bad arguments (B=%d H=%d W=%d)", B, H, W);
    return encoder_core(h, src, B, H, W, out, nullptr, s);
}

// decoder(encoder(x)) in one call (flow/model.py:39-40, 189-191, 202-204; single-frame inference).  PSPNet takes the
// fused route (no 4096-channel concat, see net_ops.hip); the other heads run encoder + decoder over an internal feature map.
int net_segment(fs_handle h, const FrameSrc& src, int B, int H, int W, float* out_nchw, hipStream_t s) {
    FS_REQUIRE(h && h->finalized, "fs_segment_forward: network not finalized");
    FS_TRY(check_device(h, "fs_segment_forward"));
    FS_REQUIRE(out_nchw && B >= 1 && H >= 1 && W >= 1, "fs_segment_forward: bad arguments (B=%d H=%d W=%d)", B, H, W);
    FS_TRY(check_frames(src, B, "fs_segment_forward"));
    int C = 0, fh = 0, fw = 0;
    if (h->cfg.arch != FS_ARCH_PSPNET || !h->use_fused_head) {
        FS_TRY(net_feature_shape(h, H, W, &C, &fh, &fw));
        const size_t need = h->cfg.arch == FS_ARCH_SEGMENTER ? (size_t)B * (fh * fw + 1) * C : (size_t)B * fh * fw * C;
        FS_TRY(ws_grow(h, &h->seg_feat, &h->seg_feat_elems, need, false));
        FS_TRY(net_encoder(h, src, B, H, W, h->seg_feat, s));
        return net_decoder(h, h->seg_feat, B, fh, fw, out_nchw, s);
    }
    FS_REQUIRE(H >= 33 && W >= 33, "fs_segment_forward: bad arguments (B=%d H=%d W=%d)", B, H, W);
    const Geometry g = geometry(h, H, W);
    fh = g.H3;
    fw = g.W3;
    // four rotating buffers: the backbone's largest map, and (tiny inputs) the row-collapsed pyramid term of the head
    FS_TRY(ensure_workspace(h, std::max(encoder_buf_elems(h, B, H, W), ppm_term_scratch_floats(B, fh, h->cls_main.Cout)), small_elems_for(B)));
    float* feat = nullptr;
    FS_TRY(encoder_core(h, src, B, H, W, nullptr, &feat, s));
    // (Running the pyramid branch -- pool, four tiny 1x1 convs, the Z GEMM, ~0.08 ms -- on a side stream under the head's
    //  Winograd GEMM was measured twice: +0.2 % in round 3, -2 % in round 5 (profiles/r05_experiments.txt section 5): the branch's
    //  kernels are bandwidth- and latency-bound and take memory slots from the GEMM they were meant to hide under.)
    hipStream_t ps = s;
    FS_TRY(pyramid_reduce(h, feat, 2048, B, fh, fw, ps));
    // free workspace buffers now: every h->buf[] except `feat`
    float *T = nullptr, *R = nullptr;  // head conv output; row-collapsed pyramid term
    for (int i = 0; i < 4; ++i)
        if (h->buf[i] != feat) {
            R = T;
            T = h->buf[i];
        }
    FS_REQUIRE(ppm_term_scratch_floats(B, fh, h->cls_main.Cout) <= h->buf_elems, "fs_segment_forward: workspace too small for the pyramid term");
    const int K = h->cfg.classes, O = h->cls_main.Cout;
    const size_t px = (size_t)B * fh * fw;
    // Z_b = reduced_b x W_b: [B*bin*bin] x [9*O]
    float* reduced = h->small + (size_t)B * 50 * 2048;
    float* zbuf = reduced + (size_t)4 * B * 36 * 512;
    const float* Z[4];
    {   // one grouped GEMM, 4 groups x [B*36 rows (the level's B*bin^2 real ones first)] x [9*O] x 512
        const ConvBN& z = h->ppm_z[0];
        const int rows = B * 36;
        ConvParams p{};
        p.in = reduced; p.ld_in = 512; p.wgt = z.w; p.out = zbuf; p.ld_out = z.Cout;
        p.B = 1; p.H = rows; p.W = 1; p.Cin = 512; p.Ho = rows; p.Wo = 1; p.Cout = z.Cout;
        p.KH = p.KW = 1; p.stride = 1; p.dil = 1;
        p.groups = 4;
        p.g_in = (long long)rows * 512;
        p.g_wgt = (long long)z.Cout * 512;
        p.g_out = (long long)rows * z.Cout;
        split_use(h, p);
        FS_TRY(prof_begin(h, "decoder.0.weight[:, ppm]", conv_igemm_tile_name(p), 2.0 * B * 50 * 512.0 * z.Cout,
                          4.0 * (4.0 * z.Cout * 512 + B * 50.0 * (512 + z.Cout)), ps));
        FS_TRY(launch_conv_igemm(p, ps));
        FS_TRY(prof_end(h, ps));
        for (int i = 0; i < 4; ++i) Z[i] = zbuf + (size_t)i * p.g_out;
    }
    ConvBN raw = h->cls_main;  // raw sums: BatchNorm + ReLU are applied after the pyramid term has been added
    raw.scale = raw.shift = nullptr;
    raw.relu = 0;
    FS_TRY(run_conv(h, raw, feat, 2048, B, fh, fw, T, O, nullptr, 0, s));
    // pyramid term + BatchNorm + ReLU + (Dropout2d: identity in eval) + the classifier 1x1 conv in one pass over the raw sums
    FS_TRY(prof_begin(h, "decoder.0.pyramid_term+decoder.4", "ppm_term_classify", 2.0 * px * O * (24.0 + K), 4.0 * px * O, s));
    FS_TRY(launch_ppm_term_classify(T, O, Z, h->bins, R, h->cls_main.scale, h->cls_main.shift, B, fh, fw, O, 1, h->cls_w, h->cls_b, out_nchw, K, s));
    return prof_end(h, s);
}

// ---------------------------------------------------------------------------------------------
int net_decoder(fs_handle h, const float* feat, int B, int fh, int fw, float* out_nchw, hipStream_t s) {
    FS_REQUIRE(h && h->finalized, "fs_decoder_forward: network not finalized");
    FS_TRY(check_device(h, "fs_decoder_forward"));
    if (h->cfg.arch == FS_ARCH_SEGMENTER) return vit_decoder(h, feat, B, fh, fw, out_nchw, s);
    FS_REQUIRE(feat && out_nchw && B >= 1 && fh >= 1 && fw >= 1, "fs_decoder_forward: bad arguments");
    const int K = h->cfg.classes;
    const size_t px = (size_t)B * fh * fw;
    if (h->cfg.arch == FS_ARCH_PSPNET) {
        FS_TRY(ensure_workspace(h, px * 512, small_elems_for(B)));
        float* T = h->buf[0];
        FS_TRY(run_conv(h, h->cls_conv, feat, 4096, B, fh, fw, T, 512, nullptr, 0, s));
        FS_TRY(prof_begin(h, "decoder.4", "classifier_nchw", 2.0 * px * 512.0 * K, 4.0 * px * 512.0, s));
        FS_TRY(launch_classifier_nchw(T, 512, h->cls_w, h->cls_b, out_nchw, B, fh * fw, 512, K, s));
        return prof_end(h, s);
    }
    // DeepLabv3 head (torchvision DeepLabHead): ASPP -> project -> 3x3 -> 1x1
    FS_TRY(ensure_workspace(h, px * 1280, small_elems_for(B)));
    float *cat = h->buf[0], *P = h->buf[1], *Q = h->buf[2];
    for (int i = 0; i < 4; ++i) FS_TRY(run_conv(h, h->aspp[i], feat, 2048, B, fh, fw, cat + 256 * i, 1280, nullptr, 0, s));
    float* pooled = h->small;
    float* reduced = h->small + (size_t)B * 2048;
    FS_TRY(prof_begin(h, "aspp.pool", "adaptive_avgpool", 0, 4.0 * px * 2048.0, s));
    if (fh % 6 == 0 && fw % 6 == 0) {
        // one window per image is only B * 64 workgroups with 254 serial pixels per thread (0.14 ms at 90x90): sum 6x6 cells
        // first (4608 workgroups), then the 36 cell means -- the PSPNet pyramid's two-stage pooling; bins 2 / 3 land in scratch
        float* cells = h->small + (size_t)B * 4096;
        float* spare = cells + (size_t)B * 36 * 2048;
        FS_TRY(launch_adaptive_avgpool(feat, 2048, cells, B, fh, fw, 2048, 6, s));
        FS_TRY(launch_ppm_pool_combine(cells, pooled, spare, spare + (size_t)B * 4 * 2048, B, 2048, s));
    } else {
        FS_TRY(launch_adaptive_avgpool(feat, 2048, pooled, B, fh, fw, 2048, 1, s));
    }
    FS_TRY(prof_end(h, s));
    FS_TRY(prof_begin(h, h->aspp_pool.name, "rowdot_1x1", 2.0 * B * 2048.0 * 256.0, 4.0 * 2048.0 * 256.0, s));
    FS_TRY(launch_rowdot_1x1(pooled, 2048, h->aspp_pool.w, h->aspp_pool.scale, h->aspp_pool.shift, reduced, 256, B, 2048, 256, 1, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(prof_begin(h, "aspp.up", "upsample_into", 0, 4.0 * px * 256.0, s));
    FS_TRY(launch_upsample_into(reduced, 1, 1, cat + 1024, 1280, B, fh, fw, 256, 0, s));
    FS_TRY(prof_end(h, s));
    FS_TRY(run_conv(h, h->project, cat, 1280, B, fh, fw, P, 256, nullptr, 0, s));
    FS_TRY(run_conv(h, h->head_conv, P, 256, B, fh, fw, Q, 256, nullptr, 0, s));
    FS_TRY(prof_begin(h, "classifier.4", "classifier_nchw", 2.0 * px * 256.0 * K, 4.0 * px * 256.0, s));
    FS_TRY(launch_classifier_nchw(Q, 256, h->cls_w, h->cls_b, out_nchw, B, fh * fw, 256, K, s));
    return prof_end(h, s);
}

int net_profile_dump(fs_handle h, char* buf, size_t n) {
    FS_REQUIRE(h && buf && n > 0, "fs_profile_dump: bad arguments");
    FS_HIP(hipDeviceSynchronize());
    size_t off = 0;
    buf[0] = 0;
    for (auto& r : h->prof) {
        float ms = 0.f;
        FS_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        const int w = snprintf(buf + off, n - off, "%s %s %.0f %.0f %.6f\n", r.name.c_str(), r.kernel.c_str(), r.flops, r.bytes, ms);
        if (w < 0 || (size_t)w >= n - off) return fail("fs_profile_dump: buffer too small");
        off += (size_t)w;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    h->prof.clear();
    return 0;
}

}  // namespace fs
