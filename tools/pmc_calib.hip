// pmc_calib.hip -- calibration of the FETCH_SIZE counter on gfx950 for the access patterns of the
// factor-panel kernels.  Every kernel reads a buffer of known size exactly once; the FETCH_SIZE a
// rocprofv3 --pmc pass reports for it, divided by the bytes read, is the correction factor for
// that pattern.  Build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// P1: aligned streaming read, 8 B per lane, grid-stride
__global__ void calib_stream_aligned(const double* __restrict__ a, int64_t n, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
// P2: the same, but the buffer starts 24 bytes into a 128-byte line (every wave load straddles 5 lines)
__global__ void calib_stream_shift24(const double* __restrict__ a, int64_t n, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
// P3: panels: one workgroup walks panels of `rows` x `w` doubles (column-major, contiguous, ld = rows odd),
// lane = row, loop over the columns with 4 loads in flight -- the L21 part of the fused interior solve
__global__ void calib_panels(const double* __restrict__ a, int rows, int w, int64_t npanels, double* out) {
  double s = 0;
  for (int64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const double* P = a + p * (int64_t)rows * w;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
      int k = 0;
      for (; k + 3 < w; k += 4) {
        const double l0 = P[r + (int64_t)rows * k], l1 = P[r + (int64_t)rows * (k + 1)], l2 = P[r + (int64_t)rows * (k + 2)],
                     l3 = P[r + (int64_t)rows * (k + 3)];
        s += (l0 + l1) + (l2 + l3);
      }
      for (; k < w; k++) s += P[r + (int64_t)rows * k];
    }
  }
  if (s == 1.2345e300) out[0] = s;
}
// P4: only the strictly lower triangle + the rows below of each panel (what the forward sweep reads),
// P5: only the upper triangle incl. diagonal of the first w rows (what the backward sweep reads of the same panel)
__global__ void calib_panels_lower(const double* __restrict__ a, int rows, int w, int64_t npanels, double* out) {
  double s = 0;
  for (int64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const double* P = a + p * (int64_t)rows * w;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
      const int kmax = r < w ? r : w;
      for (int k = 0; k < kmax; k++) s += P[r + (int64_t)rows * k];
    }
  }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void calib_panels_upper(const double* __restrict__ a, int rows, int w, int64_t npanels, double* out) {
  double s = 0;
  for (int64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const double* P = a + p * (int64_t)rows * w;
    for (int r = threadIdx.x; r < w; r += blockDim.x)
      for (int k = r; k < w; k++) s += P[r + (int64_t)rows * k];
  }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  const int64_t n = (int64_t)1 << 28;   // 2 GiB of doubles
  double *a, *out;
  CK(hipMalloc(&a, (n + 64) * sizeof(double)));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, (n + 64) * sizeof(double)));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(calib_stream_aligned, dim3(8192), dim3(256), 0, 0, a, n, out);
    hipLaunchKernelGGL(calib_stream_shift24, dim3(8192), dim3(256), 0, 0, a + 3, n, out);
    struct { int rows, w; } shapes[] = {{40, 8}, {150, 20}, {546, 49}};
    for (auto& sh : shapes) {
      const int64_t np = n / ((int64_t)sh.rows * sh.w);
      hipLaunchKernelGGL(calib_panels, dim3(8192), dim3(256), 0, 0, a + 1, sh.rows, sh.w, np, out);
      hipLaunchKernelGGL(calib_panels_lower, dim3(8192), dim3(256), 0, 0, a + 1, sh.rows, sh.w, np, out);
      hipLaunchKernelGGL(calib_panels_upper, dim3(8192), dim3(256), 0, 0, a + 1, sh.rows, sh.w, np, out);
      std::printf("CALIB shape rows %d w %d panels %lld bytes_full %lld bytes_lower %lld bytes_upper %lld\n", sh.rows, sh.w, (long long)np,
                  (long long)(np * sh.rows * sh.w * 8), (long long)(np * ((int64_t)sh.w * (sh.w - 1) / 2 + (int64_t)(sh.rows - sh.w) * sh.w) * 8),
                  (long long)(np * ((int64_t)sh.w * (sh.w + 1) / 2) * 8));
    }
    CK(hipDeviceSynchronize());
  }
  std::printf("CALIB stream bytes %lld\n", (long long)(n * 8));
  return 0;
}
