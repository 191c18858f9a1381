// Probe (GPU box only): where a wave of the split-operand attention kernels spends a stage -- per-wave cycle counters written by the
// FS_ATT_TRACE build of csrc/vit_ops.hip (pipelined kernel: wait + barrier / phase 1 / phase 2).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFS_ATT_TRACE -I flood_uav_video_segmentation_amd/csrc -I include tools/probe_attention_trace.hip -o tools/bin/probe_attention_trace
// usage: probe_attention_trace [B N heads]
#include "../flood_uav_video_segmentation_amd/csrc/vit_ops.hip"

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
    fprintf(stderr, "error: %s\n", buf);
    return 1;
}
}  // namespace fs

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2, N = argc > 2 ? atoi(argv[2]) : 2026, heads = argc > 3 ? atoi(argv[3]) : 6;
    const size_t nq = (size_t)B * N * 3 * heads * 64, no = (size_t)B * N * heads * 64;
    float *qkv, *out, *scr, *planes;
    const size_t ns = fs::attention_scratch_floats(B, N, heads), np = fs::attention_split_floats(B, N, heads);
    if (hipMalloc(&qkv, nq * 4) || hipMalloc(&out, no * 4) || hipMalloc(&scr, (ns + 4) * 4) || hipMalloc(&planes, (np + 64) * 4)) return 3;
    std::vector<float> h(nq);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.5f / (1 << 23)); }
    if (hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice)) return 3;
    for (int i = 0; i < 5; ++i) if (fs::launch_attention_split(qkv, out, B, N, heads, 0.125f, ns ? scr : nullptr, planes, 0, true)) return 4;
    (void)hipDeviceSynchronize();
    const int splits = 4;  // (upper bound of waves: 256 runs per image x 4 waves)
    const size_t waves = std::min<size_t>((size_t)B * 256 * 4, 65536);
    std::vector<unsigned long long> t(8 * waves);
    if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(fs::fs_att_trace), t.size() * 8)) return 5;
    const char* names[7] = {"wait + barrier", "phase 1 (softmax || next S^T)", "phase 2 (P.V)", "whole segment", "", "start -> Q split done", "-> first stages landed"};
    printf("B=%d N=%d heads=%d key splits=%d: %zu waves; cycles PER STAGE and wave (whole kernel: per wave)\n", B, N, heads, splits, waves);
    for (int k = 0; k < 8; ++k) {
        if (k == 4) continue;
        std::vector<double> v;
        for (size_t w = 0; w < waves; ++w) {
            const double st = (double)t[8 * w + 4];
            if (st > 0) v.push_back(k < 3 ? t[8 * w + k] / st : (double)t[8 * w + k]);
        }
        if (k == 7) {
            printf("  %-32s p10 %8.0f  p50 %8.0f  p90 %8.0f\n", "start -> stage loop", v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
            v.clear();
            for (size_t w = 0; w < waves; ++w) if (t[8 * w + 4]) v.push_back((double)t[8 * w + 3] - (double)t[8 * w + 7] - (double)(t[8 * w] + t[8 * w + 1] + t[8 * w + 2]));
            std::sort(v.begin(), v.end());
            printf("  %-32s p10 %8.0f  p50 %8.0f  p90 %8.0f\n", "after the loop (store partials)", v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
            continue;
        }
        std::sort(v.begin(), v.end());
        printf("  %-32s p10 %8.0f  p50 %8.0f  p90 %8.0f\n", names[k], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
    return 0;
}
