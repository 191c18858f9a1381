// Probe (GPU box only): per-workgroup timeline of conv_igemm_dma_f32 for one conv shape.
// Built with -DFS_TRACE (instrumentation that never ships in libfloodseg.so):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFS_TRACE -I flood_uav_video_segmentation_amd/csrc -I include \
//         tools/probe_conv_trace.hip -o tools/bin/probe_conv_trace
// usage: probe_conv_trace B H W Cin Cout K pad dil tile [dbg [groups [residual [split]]]]  > trace.csv   (analysed by tools/analyze_conv_trace.py)
//        split = 1: the split-operand instantiation (three bf16 planes of the filters, bf16 MFMA)
//        groups > 1: grouped GEMM as the Winograd path launches it (use K = 1, W = 1, H = rows per group)
#include "../flood_uav_video_segmentation_amd/csrc/conv_igemm.hip"

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <vector>

namespace fs {
std::string& last_error() { static std::string e; return e; }
int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    fprintf(stderr, "error: %s\n", buf);
    return 1;
}
}  // namespace fs

int main(int argc, char** argv) {
    if (argc < 10) { fprintf(stderr, "usage: %s B H W Cin Cout K pad dil tile [dbg]\n", argv[0]); return 2; }
    const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]), K = atoi(argv[6]),
              pad = atoi(argv[7]), dil = atoi(argv[8]), tile = atoi(argv[9]);
    const int Ho = H + 2 * pad - dil * (K - 1), Wo = W + 2 * pad - dil * (K - 1);
    const int groups = argc > 11 ? atoi(argv[11]) : 1;
    const size_t n_in = (size_t)groups * B * H * W * Cin, n_w = (size_t)groups * Cout * K * K * Cin, n_out = (size_t)groups * B * Ho * Wo * Cout;
    float *in, *wgt, *out;
    if (hipMalloc(&in, n_in * 4) || hipMalloc(&wgt, n_w * 4) || hipMalloc(&out, n_out * 4)) return 3;
    {   // pseudo-random operands: MFMA power (hence the sustained clock) depends on the data toggling
        std::vector<float> h(std::max(n_in, n_w));
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        if (hipMemcpy(in, h.data(), n_in * 4, hipMemcpyHostToDevice) || hipMemcpy(wgt, h.data(), n_w * 4, hipMemcpyHostToDevice)) return 3;
    }
    fs::ConvParams p{};
    p.in = in; p.ld_in = Cin; p.wgt = wgt; p.out = out; p.ld_out = Cout;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
    p.KH = p.KW = K; p.stride = 1; p.pad = pad; p.dil = dil; p.relu = 1;
    p.dbg = argc > 10 ? atoi(argv[10]) : 0;
    if (argc > 12 && atoi(argv[12])) {  // residual input (a bottleneck's conv3): same shape as the output
        float* res = nullptr;
        if (hipMalloc(&res, n_out * 4) != hipSuccess || hipMemset(res, 0, n_out * 4) != hipSuccess) return 3;
        p.res = res;
        p.ld_res = Cout;
    }
    if (argc > 13 && atoi(argv[13])) {  // split-operand route
        void* planes = nullptr;
        if (hipMalloc(&planes, n_w * 6) != hipSuccess || fs::launch_split_bf16x3(wgt, (long long)n_w, planes, 0)) return 3;
        p.wgt3 = planes;
        p.plane_bytes = (unsigned)(n_w * 2);
    }
    if (groups > 1) {
        p.groups = groups;
        p.g_in = (long long)B * H * W * Cin;
        p.g_wgt = (long long)Cout * K * K * Cin;
        p.g_out = (long long)B * Ho * Wo * Cout;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = getenv("FS_WARM") ? atoi(getenv("FS_WARM")) : 20;  // launches before the traced one (a long run shows the clock the card SUSTAINS)
    for (int i = 0; i < warm; ++i) if (fs::launch_conv_igemm(p, 0, tile)) return 4;
    hipDeviceSynchronize();
    if (getenv("FS_AVG")) {  // average of N back-to-back launches instead of one traced launch (A/B of a dbg setting); no timeline dump
        const int n = atoi(getenv("FS_AVG"));
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) if (fs::launch_conv_igemm(p, 0, tile)) return 4;
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        printf("avg %.2f us over %d launches (dbg=%d)\n", 1e3 * t / n, n, p.dbg);
        return 0;
    }
    hipEventRecord(e0, 0);
    if (fs::launch_conv_igemm(p, 0, tile)) return 4;
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int M = B * Ho * Wo;
    const double gf = 2.0 * groups * M * Cout * K * K * Cin * 1e-9;
    std::vector<unsigned long long> tr(8 * 65536);
    if (hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fs::fs_trace_buf), tr.size() * 8) != hipSuccess) return 5;
    printf("# %s M=%d N=%d K=%d  %.4f ms  %.1f TFLOP/s (single launch incl. launch overhead)\n", fs::conv_igemm_tile_name(p, tile), M, Cout,
           K * K * Cin, ms, gf / ms);
    if (getenv("FS_CHUNKS")) {  // round 5: per-chunk durations of the first 8 chunks (median / p10 / p90 over workgroups), split kernel only
        std::vector<unsigned long long> ch(8 * 65536);
        if (hipMemcpyFromSymbol(ch.data(), HIP_SYMBOL(fs::fs_trace_chunks), ch.size() * 8) != hipSuccess) return 5;
        std::vector<long long> d[9];
        for (int b = 0; b < 65536; ++b) {
            const unsigned long long* o = &tr[8 * (size_t)b];
            const unsigned long long* c = &ch[8 * (size_t)b];
            if (o[3] == 0 || c[0] == 0) continue;
            d[0].push_back((long long)(c[0] - o[1]));
            for (int k = 1; k < 8; ++k) if (c[k]) d[k].push_back((long long)(c[k] - c[k - 1]));
            d[8].push_back((long long)(o[1] - o[0]));
        }
        auto pr = [&](const char* nm, std::vector<long long>& v) {
            if (v.empty()) return;
            std::sort(v.begin(), v.end());
            printf("  %-22s n=%zu  p10 %7lld  p50 %7lld  p90 %7lld\n", nm, v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
        };
        pr("prologue -> stage 0", d[8]);
        for (int k = 0; k < 8; ++k) { char nm[32]; snprintf(nm, sizeof nm, "chunk %d", k); pr(nm, d[k]); }
        return 0;
    }
    printf("bid,start,ready,loop,end,wait,hwid,xcc,wall0,wall1\n");
    for (int b = 0; b < 65536; ++b) {
        const unsigned long long* o = &tr[8 * (size_t)b];
        if (o[3] == 0) continue;
        printf("%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", b, o[0], o[1], o[2], o[3], o[4], o[5] & 0xffffffffull, o[5] >> 32, o[6], o[7]);
    }
    return 0;
}
