// Diagnostic harness for k_scan_q16 (not part of the product): the filter launch on a random bf16 corpus, timed, and --
// in the -DQ16_PROFILE build -- the in-kernel clock (shader cycles per tick of the constant 100 MHz counter around the tile
// loop, MI355X_MICROARCH.md "DVFS give-back" item 6) and where a wave's cycles go (tile statement / filter slow path /
// end-of-tile vmcnt wait and pacing / barrier), medians over the waves, after >= 2 s of back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQ16_PROFILE=1048576 -o tools/bin/q16_prof tools/q16_bench.hip
//   tools/bin/q16_prof <B> [z|r] [tau] [rows]
#ifndef Q16_SRC
#define Q16_SRC "../code-rag_amd/csrc/crag_scan_q16.hip"
#endif
#include Q16_SRC
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char** argv) {
    const long long n = argc > 4 ? atoll(argv[4]) : 6000000; const int D = 768, B = argc > 1 ? atoi(argv[1]) : 1024;
    unsigned short *rows, *q; float *xn, *tau; unsigned long long* cand; uint32_t *cnt, *prog;
    const size_t cnt_words = 1048576 + 1024 * 4 * 16 + 64;
    (void)hipMalloc(&rows, n * D * 2); (void)hipMalloc(&q, 2048 * D * 2); (void)hipMalloc(&xn, (n + 512) * 4); (void)hipMalloc(&tau, 2048 * 4);
    (void)hipMalloc(&cand, 2048ull * 2 * CRAG_CAND_CAP * 8); (void)hipMalloc(&cnt, 4 * cnt_words); (void)hipMalloc(&prog, 4 * CRAG_PROGRESS_WORDS(256));
    std::vector<unsigned short> h(8 << 20);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // ~ +-0.01..0.03, random mantissas
    for (long long off = 0; off < n * D; off += h.size()) (void)hipMemcpy(rows + off, h.data(), std::min<long long>(h.size(), n * D - off) * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(q, h.data(), 2048 * D * 2, hipMemcpyHostToDevice);
    if (argc > 2 && argv[2][0] == 'z') (void)hipMemset(rows, 0, n * D * 2);  // all-zero corpus: least data toggling
    (void)hipMemset(xn, 0, (n + 512) * 4); (void)hipMemset(cnt, 0, 4 * cnt_words);
    std::vector<float> t(2048, argc > 3 ? (float)atof(argv[3]) : -1e30f);
    (void)hipMemcpy(tau, t.data(), 8192, hipMemcpyHostToDevice);
    MfmaArgs a; memset(&a, 0, sizeof(a)); a.rows = rows; a.xnorm = xn; a.n_rows = n; a.row_begin = 0; a.row_end = n; a.queries = q; a.n_queries = B;
    a.dim_pad = D; a.metric = CRAG_L2; a.tau = tau; a.cand = cand; a.cnt = cnt;
    a.segcnt = cnt; a.n_seg = crag_q16_segments(B, 256); a.seg_cap = std::max(32, 2 * CRAG_CAND_CAP / a.n_seg); a.cand_stride = (long long)a.n_seg * a.seg_cap;
    // Q16_AHEAD: tiles ahead | (pace mask << 8) as MfmaArgs::ahead_max; 0 = no pacing (no progress words)
    const int ahead = getenv("Q16_AHEAD") ? atoi(getenv("Q16_AHEAD")) : (CRAG_AHEAD_MAX_DEFAULT | ((CRAG_PACE_EVERY_DEFAULT - 1) << 8));
    a.progress = (B > 256 && (ahead & 0xff)) ? prog : nullptr; a.ahead_max = ahead;
    auto launch = [&]() { if (a.progress) (void)hipMemsetAsync(prog, 0, 4 * CRAG_PROGRESS_WORDS(256), nullptr); return crag_launch_scan_q16(a, 1, 256, nullptr); };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = (int)std::max(20.0, 2.5e3 / (2.0 * B * n * D / 1.3e12));  // >= 2 s back to back
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < reps; ++i) launch(); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("B=%d rows=%lld  %.3f ms  %.1f GB/s  %.1f TFLOP/s = %.3f of 2500  (%d launches, %s)\n", B, n, ms, n * D * 2.0 / ms / 1e6, 2.0 * B * n * D / ms / 1e9,
           2.0 * B * n * D / ms / 1e9 / 2500, reps, hipGetErrorString(hipGetLastError()));
#ifdef Q16_PROFILE
    // one isolated launch for the timeline of a launch (its stamps are the ones read below)
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms1; (void)hipEventElapsedTime(&ms1, e0, e1);
    const int nqg = (B + 255) / 256, grid_waves = 8 * std::max(1, 32 / nqg) * nqg * 4;
    std::vector<unsigned long long> pc((size_t)grid_waves * 8);
    (void)hipMemcpy(pc.data(), cnt + Q16_PROFILE, pc.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, stmt, slow, per_tile, entry, pro, fin;
    for (int w = 0; w < grid_waves; ++w) {
        const unsigned long long* p = &pc[(size_t)w * 8];
        if (!p[7] || !p[6] || !p[1]) continue;
        clk.push_back(100e6 * (double)p[0] / (double)p[1] / 1e9);
        stmt.push_back((double)p[2] / p[0]); slow.push_back((double)p[3] / p[0]);
        entry.push_back((double)p[4]); pro.push_back(((double)p[5] - (double)p[4]) / 100.0); fin.push_back((double)p[5] + (double)p[1]);
        per_tile.push_back((double)p[0] / p[6]);
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    printf("in-kernel clock %.3f GHz (median of %zu waves; min %.3f max %.3f)   cycles per tile %.0f (MFMA issue alone: 3072)\n", med(clk), clk.size(),
           clk.empty() ? 0.0 : clk.front(), clk.empty() ? 0.0 : clk.back(), med(per_tile));
    printf("share of the tile loop: statement %.3f  hit test + slow path %.3f\n", med(stmt), med(slow));
    // timeline of the isolated launch, microseconds from the first wave's entry
    if (!entry.empty()) {
        const double t0 = *std::min_element(entry.begin(), entry.end());
        for (auto& v : entry) v = (v - t0) / 100.0;
        for (auto& v : fin) v = (v - t0) / 100.0;
        const double e_med = med(entry), e_max = entry.back(), p_med = med(pro), p_max = pro.back();
        const double f_med = med(fin), f_min = fin.front(), f_max = fin.back();
        printf("isolated launch %.1f us by events | wave entry: median %.1f max %.1f us | prologue (query load + first tile): median %.1f max %.1f us | "
               "tile loop ends: min %.1f median %.1f max %.1f us\n", ms1 * 1e3, e_med, e_max, p_med, p_max, f_min, f_med, f_max);
    }
#endif
    return 0;
}
