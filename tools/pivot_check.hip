// development harness (GPU box only): the blocked pivot-piece kernel against the scalar one on random pieces, with timings.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I hymls_amd/csrc tools/pivot_check.hip -o /tmp/pivot_check -ldl
#include "../hymls_amd/csrc/device_hip.hip"
#include <random>
#include <cstring>
using namespace hymls::dev;
int main() {
  const int ws[] = {128, 127, 100, 96, 65, 64, 49, 33, 32, 17, 5};
  int rc = 0;
  HIP_CHECK(hipFuncSetAttribute((const void*)k_big_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((PIECE * PIECE + 2 * PIECE) * sizeof(double))));
  HIP_CHECK(hipFuncSetAttribute((const void*)k_big_pivot_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((PIECE * PIECE + PIVB_LT * PB) * sizeof(double))));
  for (int w : ws) {
    const int nslot = 512, ld = w + 5;
    const int64_t sA = (int64_t)ld * w + 7, lds = w + 3, sS = lds * w + 11, sT = 2 * PIECE * PIECE;
    std::vector<double> hA((size_t)sA * nslot);
    std::mt19937_64 g(w);
    std::uniform_real_distribution<double> U(-1, 1);
    for (auto& v : hA) v = U(g);
    for (int s = 0; s < nslot; s++) for (int i = 0; i < w; i++) hA[(size_t)s * sA + i + (size_t)ld * i] += (i % 3 == 0 ? -1 : 1) * 0.3 * w;
    double *dA, *dS[2], *dT[2]; int32_t* dflag;
    HIP_CHECK(hipMalloc(&dA, hA.size() * 8));
    for (int v = 0; v < 2; v++) { HIP_CHECK(hipMalloc(&dS[v], (size_t)sS * nslot * 8)); HIP_CHECK(hipMalloc(&dT[v], (size_t)sT * nslot * 8)); HIP_CHECK(hipMemset(dS[v], 0, (size_t)sS * nslot * 8)); HIP_CHECK(hipMemset(dT[v], 0, (size_t)sT * nslot * 8)); }
    HIP_CHECK(hipMalloc(&dflag, 32)); HIP_CHECK(hipMemset(dflag, 0, 32));
    HIP_CHECK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    const int Wk = (w + PB - 1) / PB * PB;
    for (int v = 0; v < 2; v++) {
      for (int rep = 0; rep < 6; rep++) {
        if (rep == 1) hipEventRecord(e0, 0);
        if (v == 0) hipLaunchKernelGGL(k_big_pivot, dim3(nslot), dim3(PIVOT_T), (size_t)(w * w + 2 * w) * sizeof(double), 0, dA, ld, sA, w, dS[0], lds, sS, dT[0], sT, dflag);
        else hipLaunchKernelGGL(k_big_pivot_blk, dim3(nslot), dim3(PIVB_T), (size_t)(Wk * Wk + PIVB_LT * PB) * sizeof(double), 0, dA, ld, sA, w, dS[1], lds, sS, dT[1], sT, dflag);
      }
      hipEventRecord(e1, 0); HIP_CHECK(hipDeviceSynchronize());
      hipEventElapsedTime(&ms[v], e0, e1);
    }
    std::vector<double> S0((size_t)sS * nslot), S1(S0.size()), T0((size_t)sT * nslot), T1(T0.size());
    HIP_CHECK(hipMemcpy(S0.data(), dS[0], S0.size() * 8, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(S1.data(), dS[1], S1.size() * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(T0.data(), dT[0], T0.size() * 8, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(T1.data(), dT[1], T1.size() * 8, hipMemcpyDeviceToHost));
    int32_t hf[8]; HIP_CHECK(hipMemcpy(hf, dflag, 32, hipMemcpyDeviceToHost));
    double dmaxS = 0, refS = 0, dmaxT = 0, refT = 0;
    for (size_t i = 0; i < S0.size(); i++) { dmaxS = std::max(dmaxS, std::abs(S0[i] - S1[i])); refS = std::max(refS, std::abs(S0[i])); }
    for (size_t i = 0; i < T0.size(); i++) { dmaxT = std::max(dmaxT, std::abs(T0[i] - T1[i])); refT = std::max(refT, std::abs(T0[i])); }
    double gr; std::memcpy(&gr, hf + 2, 8);
    const bool ok = dmaxS <= 1e-11 * refS && dmaxT <= 1e-11 * refT && std::isfinite(dmaxS) && std::isfinite(dmaxT) && refS > 0;
    std::printf("w %3d: slab diff %.2e (max %.2e)  Lf/Uf diff %.2e (max %.2e)  flag %d growth %.3g  scalar %.1f us  blocked %.1f us per launch of %d  %s\n", w, dmaxS, refS, dmaxT,
                refT, hf[0], gr, 1e3 * ms[0] / 5, 1e3 * ms[1] / 5, nslot, ok ? "OK" : "MISMATCH");
    if (!ok) rc = 1;
    hipFree(dA); hipFree(dflag); for (int v = 0; v < 2; v++) { hipFree(dS[v]); hipFree(dT[v]); }
  }
  return rc;
}
