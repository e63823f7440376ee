// probe: does a compact moving write window (few waves, grid-stride) reach the hipMemset rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int T>
__global__ __launch_bounds__(T) void fill_grid(u4 *out, size_t nvec, uint32_t val) {
  u4 v = {val, val, val, val};
  const size_t stride = (size_t)gridDim.x * T;
  for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < nvec; i += stride) out[i] = v;
}
// window variant: the grid sweeps the buffer in windows of `win` vectors; inside a window blocks interleave
template <int T, int U>
__global__ __launch_bounds__(T) void fill_grid_u(u4 *out, size_t nvec, uint32_t val) {
  u4 v = {val, val, val, val};
  const size_t stride = (size_t)gridDim.x * T;
  size_t i = (size_t)blockIdx.x * T + threadIdx.x;
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) out[i + u * stride] = v;
  }
  for (; i < nvec; i += stride) out[i] = v;
}
template <typename F>
static float run(F launch, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[reps / 2];
}
int main() {
  const size_t bytes = (size_t)2982 * 1080 * 1920, nvec = bytes / 16;
  u4 *buf; CK(hipMalloc(&buf, bytes));
  printf("memset %.3f ms\n", run([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }));
  for (int g : {8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384}) {
    float a = run([&] { hipLaunchKernelGGL(fill_grid<256>, dim3(g), dim3(256), 0, 0, buf, nvec, 1u); });
    float b = run([&] { hipLaunchKernelGGL((fill_grid_u<256, 4>), dim3(g), dim3(256), 0, 0, buf, nvec, 1u); });
    float c = run([&] { hipLaunchKernelGGL(fill_grid<1024>, dim3(g), dim3(1024), 0, 0, buf, nvec, 1u); });
    float d = run([&] { hipLaunchKernelGGL(fill_grid<64>, dim3(g), dim3(64), 0, 0, buf, nvec, 1u); });
    printf("grid %5d: T256 %.3f ms (%.0f) | T256 unroll4 %.3f (%.0f) | T1024 %.3f (%.0f) | T64 %.3f (%.0f)\n", g, a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6);
  }
  return 0;
}
