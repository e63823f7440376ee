// standalone write-bandwidth probe: which store pattern reaches the fill ceiling on MI355X?
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/fill_bench scratch/fill_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st(u4 *p, u4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// V1: each block owns blk_vec consecutive 16-byte vectors, threads stride by 256
template <bool NT>
__global__ __launch_bounds__(256) void fill_block(u4 *out, size_t nvec, int blk_vec, uint32_t val) {
  const size_t base = (size_t)blockIdx.x * blk_vec;
  u4 v = {val, val, val, val};
  for (int i = threadIdx.x; i < blk_vec; i += 256) {
    if (base + i < nvec) st<NT>(out + base + i, v);
  }
}

// V2: grid stride
template <bool NT>
__global__ __launch_bounds__(256) void fill_grid(u4 *out, size_t nvec, uint32_t val) {
  u4 v = {val, val, val, val};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) st<NT>(out + i, v);
}

// V4: each wave owns a contiguous run of its block's region
template <bool NT>
__global__ __launch_bounds__(256) void fill_wave(u4 *out, size_t nvec, int blk_vec, uint32_t val) {
  const size_t base = (size_t)blockIdx.x * blk_vec;
  const int per_wave = blk_vec / 4;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  u4 v = {val, val, val, val};
  for (int i = l; i < per_wave; i += 64) {
    const size_t idx = base + (size_t)w * per_wave + i;
    if (idx < nvec) st<NT>(out + idx, v);
  }
}

// V5: 2-D grid like the mask kernel: y = mask slot, x = strip
template <bool NT>
__global__ __launch_bounds__(256) void fill_2d(u4 *out, int strips, int blk_vec, int slot_vec, uint32_t val) {
  const size_t base = (size_t)blockIdx.y * slot_vec + (size_t)blockIdx.x * blk_vec;
  const int n = min(blk_vec, slot_vec - (int)blockIdx.x * blk_vec);
  u4 v = {val, val, val, val};
  for (int i = threadIdx.x; i < n; i += 256) st<NT>(out + base + i, v);
}

// V8: 2-D grid, T threads, division-based indexing like the mask kernel
template <bool NT, int T>
__global__ __launch_bounds__(T) void fill_2d_t(u4 *out, int rows, int H, int chunks, uint32_t val) {
  const int y0 = blockIdx.x * rows, y1 = min(y0 + rows, H);
  u4 *o = out + (size_t)blockIdx.y * H * chunks;
  u4 v = {val, val, val, val};
  const int n = (y1 - y0) * chunks;
  for (int i = threadIdx.x; i < n; i += T) {
    const int y = y0 + i / chunks, x = i % chunks;
    st<NT>(o + (size_t)y * chunks + x, v);
  }
}
// V6: each lane writes 4 consecutive vectors (64 B), block region contiguous
template <bool NT>
__global__ __launch_bounds__(256) void fill_lane4(u4 *out, size_t nvec, int blk_vec, uint32_t val) {
  const size_t base = (size_t)blockIdx.x * blk_vec;
  u4 v = {val, val, val, val};
  for (int i = threadIdx.x * 4; i < blk_vec; i += 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + i + k < nvec) st<NT>(out + base + i + k, v);
  }
}
// V7: block-contig with T threads
template <bool NT, int T>
__global__ __launch_bounds__(T) void fill_block_t(u4 *out, size_t nvec, int blk_vec, uint32_t val) {
  const size_t base = (size_t)blockIdx.x * blk_vec;
  u4 v = {val, val, val, val};
  for (int i = threadIdx.x; i < blk_vec; i += T) {
    if (base + i < nvec) st<NT>(out + base + i, v);
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static void run(const char *name, size_t bytes, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < 10; ++i) {
    CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  printf("%-44s med %.3f ms  min %.3f ms  -> %.0f GB/s\n", name, t[5], t[0], bytes / t[5] / 1e6);
  fflush(stdout);
}

int main() {
  const int slots = 2982, H = 1080, W = 1920;
  const size_t bytes = (size_t)slots * H * W;
  const size_t nvec = bytes / 16;
  u4 *buf;
  CK(hipMalloc(&buf, bytes));
  run("hipMemsetAsync", bytes, [&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); });
  char name[128];
  for (int rows : {64}) {
    const int strips = (H + rows - 1) / rows;
    snprintf(name, sizeof name, "2d rows=%3d T=256 nt=0", rows);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_2d_t<false, 256>), dim3(strips, slots), dim3(256), 0, 0, buf, rows, H, W / 16, 1u); });
    snprintf(name, sizeof name, "2d rows=%3d T=256 nt=1", rows);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_2d_t<true, 256>), dim3(strips, slots), dim3(256), 0, 0, buf, rows, H, W / 16, 1u); });
    snprintf(name, sizeof name, "2d rows=%3d T=512 nt=0", rows);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_2d_t<false, 512>), dim3(strips, slots), dim3(512), 0, 0, buf, rows, H, W / 16, 1u); });
    snprintf(name, sizeof name, "2d rows=%3d T=512 nt=1", rows);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_2d_t<true, 512>), dim3(strips, slots), dim3(512), 0, 0, buf, rows, H, W / 16, 1u); });
    snprintf(name, sizeof name, "2d rows=%3d T=1024 nt=0", rows);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_2d_t<false, 1024>), dim3(strips, slots), dim3(1024), 0, 0, buf, rows, H, W / 16, 1u); });
  }
  for (int blk_bytes : {61440, 65536, 122880, 126976, 131072, 131072+4096, 196608, 245760, 262144}) {
    const int bv = blk_bytes / 16;
    const unsigned g = (unsigned)((nvec + bv - 1) / bv);
    snprintf(name, sizeof name, "block-contig %7d B T=256", blk_bytes);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_block_t<false, 256>), dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
    snprintf(name, sizeof name, "block-contig %7d B T=512", blk_bytes);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_block_t<false, 512>), dim3(g), dim3(512), 0, 0, buf, nvec, bv, 1u); });
    snprintf(name, sizeof name, "block-contig %7d B T=1024", blk_bytes);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_block_t<false, 1024>), dim3(g), dim3(1024), 0, 0, buf, nvec, bv, 1u); });
    snprintf(name, sizeof name, "block-contig %7d B T=64", blk_bytes);
    run(name, bytes, [&] { hipLaunchKernelGGL((fill_block_t<false, 64>), dim3(g), dim3(64), 0, 0, buf, nvec, bv, 1u); });
    snprintf(name, sizeof name, "lane4        %7d B T=256", blk_bytes);
    run(name, bytes, [&] { hipLaunchKernelGGL(fill_lane4<false>, dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
  }
  for (int nt = 0; nt < 0; ++nt) {
    for (int blk_bytes : {30720, 65536, 262144}) {
      const int bv = blk_bytes / 16;
      const unsigned g = (unsigned)((nvec + bv - 1) / bv);
      snprintf(name, sizeof name, "block-contig %6d B nt=%d", blk_bytes, nt);
      if (nt) run(name, bytes, [&] { hipLaunchKernelGGL(fill_block<true>, dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
      else run(name, bytes, [&] { hipLaunchKernelGGL(fill_block<false>, dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
      snprintf(name, sizeof name, "wave-contig  %6d B nt=%d", blk_bytes, nt);
      if (nt) run(name, bytes, [&] { hipLaunchKernelGGL(fill_wave<true>, dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
      else run(name, bytes, [&] { hipLaunchKernelGGL(fill_wave<false>, dim3(g), dim3(256), 0, 0, buf, nvec, bv, 1u); });
    }
    for (int k : {2, 4, 8, 16, 64}) {
      snprintf(name, sizeof name, "grid-stride  %4d blocks/CU nt=%d", k, nt);
      if (nt) run(name, bytes, [&] { hipLaunchKernelGGL(fill_grid<true>, dim3(256 * k), dim3(256), 0, 0, buf, nvec, 1u); });
      else run(name, bytes, [&] { hipLaunchKernelGGL(fill_grid<false>, dim3(256 * k), dim3(256), 0, 0, buf, nvec, 1u); });
    }
    {
      const int slot_vec = H * W / 16, bv = 16 * W / 16, strips = (H + 15) / 16;
      snprintf(name, sizeof name, "2d strips(16 rows) x slots nt=%d", nt);
      if (nt) run(name, bytes, [&] { hipLaunchKernelGGL(fill_2d<true>, dim3(strips, slots), dim3(256), 0, 0, buf, strips, bv, slot_vec, 1u); });
      else run(name, bytes, [&] { hipLaunchKernelGGL(fill_2d<false>, dim3(strips, slots), dim3(256), 0, 0, buf, strips, bv, slot_vec, 1u); });
    }
  }
  CK(hipFree(buf));
  return 0;
}
