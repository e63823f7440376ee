// probe: cache-policy bits on 16-byte streaming stores (buffer_store_dwordx4 aux: bit0 sc0, bit1 nt, bit4 sc1 on gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int AUX>
__global__ __launch_bounds__(256) void fill_2d(unsigned char *out, int rows, int H, int W, uint32_t val) {
  const int y0 = blockIdx.x * rows, y1 = min(y0 + rows, H);
  unsigned char *o = out + (size_t)blockIdx.y * H * W;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(o, 0, H * W, 0x00020000);
  u4 v = {val, val, val, val};
  const int chunks = W / 16;
  const int n = (y1 - y0) * chunks;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int y = y0 + i / chunks, x = i % chunks;
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (y * W + x * 16), 0, AUX);
  }
}
// a small reader that follows the fill: does the fill leave write-back work behind?
__global__ void reader(const float4 *in, float4 *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
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
  const int slots = 2982, H = 1080, W = 1920;
  const size_t bytes = (size_t)slots * H * W;
  unsigned char *buf; CK(hipMalloc(&buf, bytes));
  const size_t rn = (size_t)400 << 20;  // 400 MB copy as the "next kernel"
  float4 *ra, *rb; CK(hipMalloc(&ra, rn)); CK(hipMalloc(&rb, rn));
  const int strips = (H + 15) / 16;
  auto rd = [&] { hipLaunchKernelGGL(reader, dim3((unsigned)(rn / 16 / 256)), dim3(256), 0, 0, ra, rb, rn / 16); };
  printf("reader alone %.3f ms\n", run(rd));
  printf("memset alone %.3f ms ; memset+reader %.3f ms\n", run([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }), run([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); rd(); }));
#define T(AUX) { float f = run([&] { hipLaunchKernelGGL(fill_2d<AUX>, dim3(strips, slots), dim3(256), 0, 0, buf, 16, H, W, 1u); }); \
                 float g = run([&] { hipLaunchKernelGGL(fill_2d<AUX>, dim3(strips, slots), dim3(256), 0, 0, buf, 16, H, W, 1u); rd(); }); \
                 printf("aux %2d: fill %.3f ms (%.0f GB/s) ; fill+reader %.3f ms\n", AUX, f, bytes / f / 1e6, g); }
  T(0) T(1) T(2) T(3) T(16) T(17) T(18) T(19)
  return 0;
}
