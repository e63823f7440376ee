// probe (round 2): which store / load structure lets a letterbox-shaped kernel approach the write ceiling?
// Output (B,3,S,S) fp32 = 315 MB for B=64, S=640; source rows 1080p (5760 B each, one per content output row for
// the exact x3 case) or 4K (11520 B, two per content row).  No resampling arithmetic: only the memory structure.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int S = 640, B = 64;

// A: flat fill in address order, grid-stride
__global__ __launch_bounds__(256) void fill_flat(f4 *out, size_t nvec) {
  f4 v = {1.f, 1.f, 1.f, 1.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, out + i);
}

// B: unit = (image, strip of R rows): 3 planes x R rows x 2560 B, contiguous per plane.  persistent (grid-stride over units)
template <int R>
__global__ __launch_bounds__(256) void fill_strip(float *out, int units_per_img) {
  f4 v = {1.f, 1.f, 1.f, 1.f};
  const int total = B * units_per_img;
  for (int u = blockIdx.x; u < total; u += gridDim.x) {
    const int b = u / units_per_img, s = u - b * units_per_img;
    for (int c = 0; c < 3; ++c) {
      f4 *p = reinterpret_cast<f4 *>(out + ((size_t)(b * 3 + c) * S + (size_t)s * R) * S);
      for (int i = threadIdx.x; i < R * S / 4; i += 256) __builtin_nontemporal_store(v, p + i);
    }
  }
}

// C: the round-1 tile pattern: 128 x 8 tile, 3 planes, one 16-byte store per lane and plane
__global__ __launch_bounds__(256) void fill_tile(float *out) {
  f4 v = {1.f, 1.f, 1.f, 1.f};
  const int b = blockIdx.y, tx = blockIdx.x % 5, ty = blockIdx.x / 5;
  const int g = threadIdx.x & 31, r = threadIdx.x >> 5;
  for (int c = 0; c < 3; ++c)
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + ((size_t)(b * 3 + c) * S + ty * 8 + r) * S + tx * 128 + 4 * g));
}

// D: strip units + source rows streamed through registers (rows_per_out source rows of row_bytes per content row);
// content rows are [140, 500).  The loaded data is folded into the stored value so the loads stay.
template <int R>
__global__ __launch_bounds__(256) void mix_strip(float *out, const u4 *src, int src_rows, int row_vec, int rows_per_out, int step, int units_per_img) {
  const int total = B * units_per_img;
  for (int u = blockIdx.x; u < total; u += gridDim.x) {
    const int b = u / units_per_img, s = u - b * units_per_img;
    uint32_t acc = 0;
    const int y0 = s * R;
    if (y0 + R > 140 && y0 < 500) {
      const int n = R * rows_per_out * row_vec;
      for (int i = threadIdx.x; i < n; i += 256) {
        const int rr = i / row_vec, x = i - rr * row_vec;
        const int oy = y0 + rr / rows_per_out - 140;
        if (oy >= 0 && oy < 360) {
          const int sy = oy * step + 1 + (rr % rows_per_out);
          u4 t = __builtin_nontemporal_load(src + ((size_t)b * src_rows + sy) * row_vec + x);
          acc += t.x ^ t.y ^ t.z ^ t.w;
        }
      }
    }
    f4 v = {(float)acc, 1.f, 1.f, 1.f};
    for (int c = 0; c < 3; ++c) {
      f4 *p = reinterpret_cast<f4 *>(out + ((size_t)(b * 3 + c) * S + (size_t)y0) * S);
      for (int i = threadIdx.x; i < R * S / 4; i += 256) __builtin_nontemporal_store(v, p + i);
    }
  }
}

// E: like D but loads of unit u+1 are issued before the stores of unit u (software pipelining, K loads per lane kept in registers)
template <int R, int K>
__global__ __launch_bounds__(256) void mix_strip_pipe(float *out, const u4 *src, int src_rows, int row_vec, int rows_per_out, int step, int units_per_img) {
  const int total = B * units_per_img;
  u4 hold[K];
  auto issue = [&](int u) {
    const int b = u / units_per_img, s = u - b * units_per_img;
    const int y0 = s * R;
    const int n = R * rows_per_out * row_vec;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = threadIdx.x + 256 * k;
      hold[k] = u4{0, 0, 0, 0};
      if (i < n && y0 + R > 140 && y0 < 500) {
        const int rr = i / row_vec, x = i - rr * row_vec;
        const int oy = y0 + rr / rows_per_out - 140;
        if (oy >= 0 && oy < 360) {
          const int sy = oy * step + 1 + (rr % rows_per_out);
          hold[k] = __builtin_nontemporal_load(src + ((size_t)b * src_rows + sy) * row_vec + x);
        }
      }
    }
  };
  int u = blockIdx.x;
  if (u < total) issue(u);
  for (; u < total; u += gridDim.x) {
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += hold[k].x ^ hold[k].y ^ hold[k].z ^ hold[k].w;
    if (u + gridDim.x < total) issue(u + gridDim.x);
    const int b = u / units_per_img, s = u - b * units_per_img;
    f4 v = {(float)acc, 1.f, 1.f, 1.f};
    for (int c = 0; c < 3; ++c) {
      f4 *p = reinterpret_cast<f4 *>(out + ((size_t)(b * 3 + c) * S + (size_t)s * R) * S);
      for (int i = threadIdx.x; i < R * S / 4; i += 256) __builtin_nontemporal_store(v, p + i);
    }
  }
}

template <typename F>
static float run(F launch, int reps = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[reps / 2] * 1000.f;
}

int main() {
  const size_t out_bytes = (size_t)B * 3 * S * S * 4;
  float *out; CK(hipMalloc(&out, out_bytes));
  const size_t src_bytes = (size_t)B * 2160 * 11520;
  constexpr int NSRC = 3;   // rotated between launches: cold reads (the Infinity Cache holds 256 MB)
  u4 *srcs[NSRC];
  for (int i = 0; i < NSRC; ++i) { CK(hipMalloc(&srcs[i], src_bytes)); CK(hipMemset(srcs[i], 1 + i, src_bytes)); }
  int rot = 0;
  auto next_src = [&]() { rot = (rot + 1) % NSRC; return srcs[rot]; };
  u4 *src = srcs[0];
  const double MB = out_bytes / 1e6;
  printf("output %.1f MB\n", MB);
  printf("memset: %.1f us\n", run([&] { CK(hipMemsetAsync(out, 0, out_bytes, 0)); }));
  for (int g : {128, 256, 512, 1024}) printf("A flat fill grid %4d: %.1f us\n", g, run([&] { hipLaunchKernelGGL(fill_flat, dim3(g), dim3(256), 0, 0, (f4 *)out, out_bytes / 16); }));
  for (int g : {256, 512, 1024, 5120}) {
    printf("B strip R=8  grid %4d: %.1f us\n", g, run([&] { hipLaunchKernelGGL(fill_strip<8>, dim3(g), dim3(256), 0, 0, out, S / 8); }));
    printf("B strip R=4  grid %4d: %.1f us\n", g, run([&] { hipLaunchKernelGGL(fill_strip<4>, dim3(g), dim3(256), 0, 0, out, S / 4); }));
    printf("B strip R=16 grid %4d: %.1f us\n", g, run([&] { hipLaunchKernelGGL(fill_strip<16>, dim3(g), dim3(256), 0, 0, out, S / 16); }));
  }
  printf("C tile 128x8 (25600 wg): %.1f us\n", run([&] { hipLaunchKernelGGL(fill_tile, dim3(5 * 80, B), dim3(256), 0, 0, out); }));
  printf("(sources rotate over %d buffers: cold reads)\n", NSRC);
  struct Cfg { const char *name; int src_rows, row_vec, rows_per_out, step; double rd_mb; } cfgs[] = {
      {"1080p (1 row of 5760 B per content row)", 1080, 360, 1, 3, 64 * 360 * 5760 / 1e6},
      {"4K (2 rows of 11520 B per content row)", 2160, 720, 2, 6, 64 * 360 * 2 * 11520 / 1e6}};
  for (auto &c : cfgs) {
    printf("--- %s: reads %.1f MB + writes %.1f MB\n", c.name, c.rd_mb, MB);
    for (int g : {256, 512, 1024, 2048, 5120}) {
      float t8 = run([&] { src = next_src(); hipLaunchKernelGGL(mix_strip<8>, dim3(g), dim3(256), 0, 0, out, src, c.src_rows, c.row_vec, c.rows_per_out, c.step, S / 8); });
      float t4 = run([&] { src = next_src(); hipLaunchKernelGGL(mix_strip<4>, dim3(std::min(g * 2, 10240)), dim3(256), 0, 0, out, src, c.src_rows, c.row_vec, c.rows_per_out, c.step, S / 4); });
      printf("D mix R=8 grid %4d: %.1f us (%.0f GB/s) | R=4 grid x2: %.1f us (%.0f GB/s)\n", g, t8, (c.rd_mb + MB) / t8 * 1e3, t4, (c.rd_mb + MB) / t4 * 1e3);
    }
    for (int g : {256, 512, 1024}) {
      // K loads of 16 B per lane cover R*rows_per_out*row_vec vectors: 1080p R=8: 2880/256 -> 12; 4K R=4: 5760/256 -> 23
      float t = c.rows_per_out == 1
                    ? run([&] { src = next_src(); hipLaunchKernelGGL((mix_strip_pipe<8, 12>), dim3(g), dim3(256), 0, 0, out, src, c.src_rows, c.row_vec, c.rows_per_out, c.step, S / 8); })
                    : run([&] { src = next_src(); hipLaunchKernelGGL((mix_strip_pipe<4, 23>), dim3(g), dim3(256), 0, 0, out, src, c.src_rows, c.row_vec, c.rows_per_out, c.step, S / 4); });
      printf("E pipelined grid %4d: %.1f us (%.0f GB/s)\n", g, t, (c.rd_mb + MB) / t * 1e3);
    }
  }
  return 0;
}
