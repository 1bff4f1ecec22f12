// Ablation harness for the query-stationary MFMA scan (not part of the product): times
// k_scan_qreg2<48,1> on random data with tau = -inf (no appends).
#define USE_Q2 1
#include "../code-rag_amd/csrc/crag_scan_qreg2.hip"
#define LAUNCH crag_launch_scan_qreg2
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>
int main(int argc, char** argv) {
    const long long n = argc > 4 ? atoll(argv[4]) : 4000000; const int D = 768, B = argc > 1 ? atoi(argv[1]) : 256;
    unsigned short *rows, *q; float *xn, *tau; unsigned long long* cand; uint32_t* cnt;
    hipMalloc(&rows, n * D * 2); hipMalloc(&q, 1024 * D * 2); hipMalloc(&xn, (n + 512) * 4); hipMalloc(&tau, 1024 * 4);
    hipMalloc(&cand, 1024ull * CRAG_CAND_CAP * 8); hipMalloc(&cnt, 4 * (1024 * 512 + 64));
    std::vector<unsigned short> h(8 << 20);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // ~ +-0.01..0.03
    for (long long off = 0; off < n * D; off += h.size()) hipMemcpy(rows + off, h.data(), std::min<long long>(h.size(), n * D - off) * 2, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), 1024 * D * 2, hipMemcpyHostToDevice);
    if (argc > 2 && argv[2][0] == 'z') hipMemset(rows, 0, n * D * 2);  // all-zero corpus: least data toggling
    hipMemset(xn, 0, (n + 512) * 4); hipMemset(cnt, 0, 4 * (1024 * 512 + 64));
    std::vector<float> t(1024, argc > 3 ? (float)atof(argv[3]) : -1e30f);  // argv[3]: tau (random corpus: score sigma ~ 6.7e-3)
    hipMemcpy(tau, t.data(), 4096, hipMemcpyHostToDevice);
    MfmaArgs a; memset(&a, 0, sizeof(a)); a.rows = rows; a.xnorm = xn; a.n_rows = n; a.row_begin = 0; a.row_end = n; a.queries = q; a.n_queries = B;
    a.dim_pad = D; a.metric = CRAG_L2; a.tau = tau; a.sample = nullptr; a.sample_stride = 0; a.cand = cand; a.cnt = cnt;
#ifdef USE_Q2
    a.segcnt = cnt; a.n_seg = crag_qreg2_segments(B, 256); a.seg_cap = CRAG_CAND_CAP / a.n_seg; a.cand_stride = (long long)a.n_seg * a.seg_cap;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) LAUNCH(a, 1, 256, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) LAUNCH(a, 1, 256, nullptr); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    { std::vector<uint32_t> sc((size_t)B * 512); hipMemcpy(sc.data(), cnt, sc.size() * 4, hipMemcpyDeviceToHost); double tot = 0; for (auto v : sc) tot += v;
      printf("appends per query per launch: %.1f\n", tot / B / 13); }
    printf("B=%d  %.3f ms  %.1f GB/s  %.1f TFLOP/s  (%s)\n", B, ms, n * D * 2.0 / ms / 1e6, 2.0 * B * n * D / ms / 1e9, hipGetErrorString(hipGetLastError()));
#ifdef Q2_PROFILE
    uint32_t pc[9]; hipMemset(cnt, 0, 4 * (1024 * 512 + 64)); LAUNCH(a, 1, 256, nullptr); hipDeviceSynchronize();
    hipMemcpy(pc, cnt + Q2_PROFILE, 36, hipMemcpyDeviceToHost);
    printf("entry spread %u  first entry -> last exit %u (ticks of the same clock)\n", pc[7] - ~pc[6], pc[8] - ~pc[6]);
    printf("per wave (100 MHz ticks): vmcnt wait %.1f  barrier %.1f  loop %.1f  query load %.1f  first tiles %.1f  (%u waves)\n", 1.0 * pc[0] / pc[3], 1.0 * pc[1] / pc[3], 1.0 * pc[2] / pc[3], 1.0 * pc[4] / pc[3], 1.0 * pc[5] / pc[3], pc[3]);
#endif
    return 0;
}
