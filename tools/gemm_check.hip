// development harness (GPU box only): the FP64 matrix-core products of the setup on the shapes the 256^3 Stokes setup
// launches most (profiles/r03_u_setup_prof_256.txt), every kernel variant against the 64 x 64 one (same bits expected).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I hymls_amd/csrc tools/gemm_check.hip -o tools/gemm_check.bin -ldl
#include "../hymls_amd/csrc/device_hip.hip"
#include <random>
#include <cstring>
#include <vector>
using namespace hymls::dev;
struct Shape { int M, N, K, batch; };
template <int VAR, bool BIG>
static void launch(double* C, int64_t ld, int64_t sA, int M, int N, int K, int batch, int off) {
  // the trailing update of a front: C = A22 -= A21 A12, all inside one column-major front of leading dimension ld
  double* A22 = C + off + ld * off; const double* A21 = C + off; const double* A12 = C + ld * off;
  if (BIG) hipLaunchKernelGGL((k_gemm_f64_big<1, 0, 0, VAR>), dim3((M + 127) / 128, (N + 127) / 128, batch), dim3(256), 0, 0, A22, ld, sA, A21, ld, sA, A12, ld, sA, M, N, K);
  else hipLaunchKernelGGL((k_gemm_f64<1, 0, 0, VAR>), dim3((M + 63) / 64, (N + 63) / 64, batch), dim3(256), 0, 0, A22, ld, sA, A21, ld, sA, A12, ld, sA, M, N, K);
}
int main() {
  const Shape shapes[] = {{546, 546, 49, 723}, {339, 339, 35, 723}, {320, 320, 32, 723}, {791, 791, 32, 224}, {2287, 2287, 128, 6},
                          {3793, 3793, 128, 6}, {5965, 5965, 128, 6}, {1200, 1200, 128, 6}, {600, 600, 128, 3},
                          {3793, 3793, 256, 6}, {3793, 3793, 512, 6}, {2287, 2287, 512, 6}, {1200, 1200, 512, 6}, {128, 3793, 384, 6}, {3793, 128, 384, 6}};
  int rc = 0;
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K, batch = sh.batch, ld = std::max(M, N) + K;   // (the front is square: rows and columns 0 .. max(M, N) + K)
    const int64_t sA = (int64_t)ld * ld + 3;
    const size_t tot = (size_t)sA * batch;
    std::vector<double> h(std::min<size_t>(tot, (size_t)1 << 24));
    std::mt19937_64 g(M);
    std::uniform_real_distribution<double> U(-1, 1);
    for (auto& v : h) v = U(g);
    double *d0, *d1, *dref;
    HIP_CHECK(hipMalloc(&d0, tot * 8)); HIP_CHECK(hipMalloc(&d1, tot * 8)); HIP_CHECK(hipMalloc(&dref, tot * 8));
    for (size_t o = 0; o < tot; o += h.size()) HIP_CHECK(hipMemcpy(d0 + o, h.data(), std::min(h.size(), tot - o) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"64 var0", "64 var1", "big var0", "big var1", "big var2"};
    std::printf("M %d N %d K %d batch %d ld %d\n", M, N, K, batch, ld);
    std::vector<double> r0(std::min<size_t>(tot, (size_t)1 << 22)), r1(r0.size());
    for (int v = 0; v < 5; v++) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; rep++) {
        HIP_CHECK(hipMemcpy(d1, d0, tot * 8, hipMemcpyDeviceToDevice));
        HIP_CHECK(hipDeviceSynchronize());
        hipEventRecord(e0, 0);
        switch (v) {
          case 0: launch<0, false>(d1, ld, sA, M, N, K, batch, K); break;
          case 1: launch<1, false>(d1, ld, sA, M, N, K, batch, K); break;
          case 2: launch<0, true>(d1, ld, sA, M, N, K, batch, K); break;
          case 3: launch<1, true>(d1, ld, sA, M, N, K, batch, K); break;
          case 4: launch<2, true>(d1, ld, sA, M, N, K, batch, K); break;
        }
        hipEventRecord(e1, 0);
        HIP_CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      // compare the first and the last member with the 64 x 64 var0 result
      size_t bad = 0;
      if (v == 0) HIP_CHECK(hipMemcpy(dref, d1, tot * 8, hipMemcpyDeviceToDevice));
      else {
        for (size_t o : {(size_t)0, tot - r0.size()}) {
          HIP_CHECK(hipMemcpy(r0.data(), dref + o, r0.size() * 8, hipMemcpyDeviceToHost));
          HIP_CHECK(hipMemcpy(r1.data(), d1 + o, r1.size() * 8, hipMemcpyDeviceToHost));
          bad += std::memcmp(r0.data(), r1.data(), r0.size() * 8) != 0;
        }
      }
      std::printf("  %-9s %8.3f ms  %6.2f TFLOP/s  C traffic %5.2f TB/s %s\n", names[v], best, 2.0 * M * N * K * batch / (best * 1e9),
                  16.0 * M * N * batch / (best * 1e9), bad ? "DIFFERENT BITS" : "");
      rc |= bad != 0;
    }
    hipFree(d0); hipFree(d1); hipFree(dref);
  }
  return rc;
}
