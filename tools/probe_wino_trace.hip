// Probe (GPU box only): per-workgroup timeline of the warp-specialised fused Winograd kernel (wino_fused.hip, variant 3).
// Built with -DFS_TRACE (instrumentation that never ships in libfloodseg.so):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFS_TRACE -I flood_uav_video_segmentation_amd/csrc -I include \
//         tools/probe_wino_trace.hip -o tools/bin/probe_wino_trace
// usage: probe_wino_trace B H W Cin Cout [variant]
#include "../flood_uav_video_segmentation_amd/csrc/wino_fused.hip"

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
    if (argc < 6) { fprintf(stderr, "usage: %s B H W Cin Cout [variant]\n", argv[0]); return 2; }
    const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    const int variant = argc > 6 ? atoi(argv[6]) : 3;
    const size_t n_in = (size_t)B * H * W * Cin, n_w = (size_t)Cout * Cin * 9, n_out = (size_t)B * H * W * Cout, n_u = (size_t)36 * Cin * Cout;
    float *in, *wgt, *out, *U;
    if (hipMalloc(&in, n_in * 4) || hipMalloc(&wgt, n_w * 4) || hipMalloc(&out, n_out * 4) || hipMalloc(&U, n_u * 4)) return 3;
    {
        std::vector<float> h(std::max(n_in, n_w));
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        if (hipMemcpy(in, h.data(), n_in * 4, hipMemcpyHostToDevice) || hipMemcpy(wgt, h.data(), n_w * 4, hipMemcpyHostToDevice)) return 3;
    }
    if (fs::launch_wino4_filter_packed(wgt, U, Cout, Cin, 0, 0)) return 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        if (fs::launch_wino4_fused(in, Cin, U, nullptr, nullptr, out, Cout, B, H, W, Cin, Cout, 1, 0, variant)) return 4;
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) fs::launch_wino4_fused(in, Cin, U, nullptr, nullptr, out, Cout, B, H, W, Cin, Cout, 1, 0, variant);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int tiles = B * ((H + 3) / 4) * ((W + 3) / 4), nblk = (tiles + 15) / 16 * (Cout / 64), nwg = variant == 3 ? std::min(nblk, 256) : nblk, nst = Cin / 16;
    printf("shape B=%d %dx%d Cin=%d Cout=%d variant %d: %.1f us per launch, %d workgroups (stamps: the FIRST block of each), %.1f TFLOP/s on the matrix cores\n", B, H, W, Cin, Cout, variant,
           ms / iters * 1e3, nwg, 2.0 * 36 * tiles * Cin * Cout / (ms / iters * 1e-3) / 1e12);
    if (variant < 3) return 0;
    std::vector<unsigned long long> tr((size_t)32 * 16384);
    if (hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fs::fs_wino_trace), tr.size() * 8)) return 5;
    const int n = std::min(nwg, 16384);
    auto med = [&](auto f) { std::vector<double> v; for (int w = 0; w < n; ++w) v.push_back(f(tr.data() + 32 * w)); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("median over %d workgroups, shader cycles:\n", n);
    printf("  M wave: start -> first barrier passed   %8.0f\n", med([](const unsigned long long* t) { return (double)(t[1] - t[0]); }));
    for (int s = 0; s < nst && s < 12; ++s)
        printf("  M wave: stage %d (144 MFMAs = 4608 issue cycles) %8.0f\n", s, med([s](const unsigned long long* t) { return (double)(t[2 + s] - (s ? t[1 + s] : t[1])); }));
    printf("  M wave: epilogue                         %8.0f\n", med([nst](const unsigned long long* t) { return (double)(t[14] - t[1 + nst]); }));
    printf("  M wave: whole workgroup                  %8.0f\n", med([](const unsigned long long* t) { return (double)(t[14] - t[0]); }));
    printf("  T wave: transform 0                      %8.0f\n", med([](const unsigned long long* t) { return (double)(t[17] - t[16]); }));
    for (int s = 0; s < nst && s < 12; ++s)
        printf("  T wave: loop iteration %d (barrier wait + transforms) %8.0f\n", s, med([s](const unsigned long long* t) { return (double)(t[18 + s] - t[17 + s]); }));
    double first = 1e300, last = 0;
    for (int w = 0; w < n; ++w) { first = std::min(first, (double)tr[32 * w]); last = std::max(last, (double)tr[32 * w + 14]); }
    printf("  first workgroup start -> last workgroup end: %.0f cycles\n", last - first);
    return 0;
}
