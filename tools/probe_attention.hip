// Probe (GPU box only): elimination experiments on attention_f32_kernel -- which part of a key tile costs what.
//   for e in 0 1 2 3 4; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFS_ATT_EXP=$e -I flood_uav_video_segmentation_amd/csrc -I include \
//         tools/probe_attention.hip -o tools/bin/probe_attention_$e; done
// usage: probe_attention_E B N heads     (results are wrong for E != 0: timing only)
#include "../flood_uav_video_segmentation_amd/csrc/vit_ops.hip"

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
    float *qkv, *out, *scr;
    const size_t ns = fs::attention_scratch_floats(B, N, heads);
    if (hipMalloc(&qkv, nq * 4) || hipMalloc(&out, no * 4) || hipMalloc(&scr, (ns + 4) * 4)) return 3;
    std::vector<float> h(nq);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    if (hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice)) return 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) if (fs::launch_attention_f32(qkv, out, B, N, heads, 0.125f, ns ? scr : nullptr, 0)) return 4;
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) if (fs::launch_attention_f32(qkv, out, B, N, heads, 0.125f, ns ? scr : nullptr, 0)) return 4;
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double gf = 4.0 * B * heads * (double)N * N * 64 * 1e-9;
    printf("FS_ATT_EXP=%d B=%d N=%d heads=%d splits=%d  %.4f ms  %.1f TFLOP/s (nominal FLOPs)\n", FS_ATT_EXP, B, N, heads,
           fs::attention_splits(B, N, heads), ms, gf / ms);
    return 0;
}
