// stream_widths.hip -- what the memory system delivers for the request mixes of the interior-solve kernels (development aid,
// round 3).  Every kernel reads `bytes` once; the printed rate is bytes / event time.
//   A  grid-stride stream, 8 B per lane            B  grid-stride stream, 16 B per lane (double2)
//   C  one workgroup per 0.57 MB chunk (the factor slab of one subdomain), 69 696 chunks, 8 workgroups per CU; a thread issues
//      U loads of 8 B (rows on lanes, stride = a column) and consumes them, step after step            U = 4, 8
//   D  the same with a workgroup barrier after every S steps (the level barriers of the fused solve)   S = 8
//   E  as C with 16 B per lane (two rows per lane)
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_widths.hip -o tools/stream_widths.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void kA(const double* __restrict__ a, int64_t n, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) out[0] = s;
}
__global__ void kB(const double2* __restrict__ a, int64_t n2, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
// chunk of `len` doubles per workgroup, viewed as columns of 256 rows: thread = row, U columns in flight
template <int U, int S>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) kC(const double* __restrict__ a, int64_t len, double* out) {
  const double* p = a + (int64_t)blockIdx.x * len + threadIdx.x;
  const int ncol = (int)(len / 256);
  double s = 0;
  int step = 0;
  for (int k = 0; k + U <= ncol; k += U) {
    double l[U];
#pragma unroll
    for (int u = 0; u < U; u++) l[u] = p[(int64_t)256 * (k + u)];
#pragma unroll
    for (int u = 0; u < U; u++) s += l[u];
    if (S > 0 && (++step % S) == 0) __syncthreads();
  }
  if (s == 1.2345e300) out[0] = s;
}
template <int U>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) kE(const double2* __restrict__ a, int64_t len2, double* out) {
  const double2* p = a + (int64_t)blockIdx.x * len2 + threadIdx.x;
  const int ncol = (int)(len2 / 256);
  double s = 0;
  for (int k = 0; k + U <= ncol; k += U) {
    double2 l[U];
#pragma unroll
    for (int u = 0; u < U; u++) l[u] = p[(int64_t)256 * (k + u)];
#pragma unroll
    for (int u = 0; u < U; u++) s += l[u].x + l[u].y;
  }
  if (s == 1.2345e300) out[0] = s;
}

int main() {
  const int64_t nchunk = 69696, len = 71680;          // 0.573 MB per chunk = 280 columns of 256 rows; 40.0 GB in all
  const int64_t n = nchunk * len;
  double *a, *out;
  CK(hipMalloc(&a, n * sizeof(double)));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, n * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, float ms) { std::printf("%-58s %8.3f ms  %6.2f TB/s\n", name, ms, n * 8.0 / ms / 1e9); };
  for (int rep = 0; rep < 2; rep++) {
    float ms;
#define TIME(name, launch) CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report(name, ms);
    TIME("A  grid-stride, 8 B per lane", hipLaunchKernelGGL(kA, dim3(16384), dim3(256), 0, 0, a, n, out));
    TIME("B  grid-stride, 16 B per lane", hipLaunchKernelGGL(kB, dim3(16384), dim3(256), 0, 0, (const double2*)a, n / 2, out));
    TIME("C4 chunk per workgroup, 4 x 8 B in flight per thread", hipLaunchKernelGGL((kC<4, 0>), dim3(nchunk), dim3(256), 0, 0, a, len, out));
    TIME("C8 chunk per workgroup, 8 x 8 B in flight per thread", hipLaunchKernelGGL((kC<8, 0>), dim3(nchunk), dim3(256), 0, 0, a, len, out));
    TIME("C2 chunk per workgroup, 2 x 8 B in flight per thread", hipLaunchKernelGGL((kC<2, 0>), dim3(nchunk), dim3(256), 0, 0, a, len, out));
    TIME("D4 as C4 + a barrier every 8 steps", hipLaunchKernelGGL((kC<4, 8>), dim3(nchunk), dim3(256), 0, 0, a, len, out));
    TIME("D4' as C4 + a barrier every 2 steps", hipLaunchKernelGGL((kC<4, 2>), dim3(nchunk), dim3(256), 0, 0, a, len, out));
    TIME("E2 chunk per workgroup, 2 x 16 B in flight per thread", hipLaunchKernelGGL((kE<2>), dim3(nchunk), dim3(256), 0, 0, (const double2*)a, len / 2, out));
    TIME("E4 chunk per workgroup, 4 x 16 B in flight per thread", hipLaunchKernelGGL((kE<4>), dim3(nchunk), dim3(256), 0, 0, (const double2*)a, len / 2, out));
  }
  return 0;
}
