// probe (r03): can the zero fill of the masks leave out the rows under each slot's box and still run at the memset
// rate?  The fill walks a table of 4 KiB blocks (start line, lines) that covers everything EXCEPT rows [ya, yb) of
// every slot, one workgroup per CU, in address order; the rows under the boxes are then written by a second kernel
// (zeros + nothing computed here: the store pattern only), 16-row strips, 8 workgroups per CU.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/fill_skip scratch/fill_skip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <cmath>
#include <cstdlib>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_plain(u4 *out, size_t nvec) {
  const u4 z = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) __builtin_nontemporal_store(z, out + i);
}

// table entry: x = first 128-byte line, y = 16-byte chunks (<= 256).  The table is laid out per workgroup: entry
// (w, t) -- block v = t G + w of the virtual sequence, so that at any moment the workgroups write neighbouring blocks --
// sits at tab[w T + t]; every lane fetches one entry of the next 64 iterations while the current 64 are stored, and an
// iteration takes its entry with v_readlane.
// the product's fill with the workgroup's four waves kept in step: one bare s_barrier (no wait for the stores) per trip
template <int EVERY, int T = 256>
__global__ __launch_bounds__(T) void fill_plain_bar(u4 *out, size_t nvec) {
  const u4 z = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * T;
  int k = 0;
  for (size_t base = (size_t)blockIdx.x * T; base < nvec; base += stride) {
    if (base + threadIdx.x < nvec) __builtin_nontemporal_store(z, out + base + threadIdx.x);
    if (++k == EVERY) { __builtin_amdgcn_s_barrier(); k = 0; }
  }
}

// barrier-paced fill, workgroup -> block mapping varied: XCD x (= workgroup id % 8) owns a contiguous eighth of every window
template <int MAP>
__global__ __launch_bounds__(256) void fill_bar_map(u4 *out, size_t nvec) {
  const u4 z = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  const unsigned w = blockIdx.x, G = gridDim.x;
  const unsigned slot = MAP == 1 ? (w % 8) * (G / 8) + w / 8 : MAP == 2 ? (w / 8) + (w % 8) * (G / 8) : w;
  for (size_t base = (size_t)slot * 256; base < nvec; base += stride) {
    if (base + threadIdx.x < nvec) __builtin_nontemporal_store(z, out + base + threadIdx.x);
    __builtin_amdgcn_s_barrier();
  }
}
// each wave writes 2 consecutive KiB per trip (a workgroup 8 KiB): window 2 MiB
__global__ __launch_bounds__(256) void fill_bar_2k(u4 *out, size_t nvec) {
  const u4 z = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 512;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * 512 + wave * 128 + lane; base < nvec; base += stride) {
    __builtin_nontemporal_store(z, out + base);
    if (base + 64 < nvec) __builtin_nontemporal_store(z, out + base + 64);
    __builtin_amdgcn_s_barrier();
  }
}

// barrier-paced fill with at most K + 1 stores of a wave in flight (s_waitcnt vmcnt(K) after every store)
template <int K, bool BAR>
__global__ __launch_bounds__(256) void fill_throttle(u4 *out, size_t nvec) {
  const u4 z = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t base = (size_t)blockIdx.x * 256; base < nvec; base += stride) {
    if (base + threadIdx.x < nvec) __builtin_nontemporal_store(z, out + base + threadIdx.x);
    if (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (K == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (K == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (K == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if (K == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
  }
}

typedef uint32_t u16v __attribute__((ext_vector_type(16)));
// Every entry is a whole 4 KiB block (its first 128-byte line): a run's last block ends at the run's end (it may overlap
// the block before), a run shorter than a block spills into the box rows behind it, which the row pass rewrites.  The
// entries of 16 iterations are one 64-byte scalar load (s_load_dwordx16, counted by lgkmcnt: waiting for it does not
// drain the stores, which a vector load's vmcnt wait would), issued 16 iterations ahead.  T is a multiple of 16.
__global__ __launch_bounds__(256) void fill_table(const uint32_t *__restrict__ tab, uint32_t T, u4 *out) {
  const u4 z = {0u, 0u, 0u, 0u};
  const uint32_t *mytab = tab + (size_t)blockIdx.x * T;
  u4 *o = out + threadIdx.x;
  u16v cur, nxt;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cur) : "s"(mytab));
  for (uint32_t t0 = 0; t0 < T; t0 += 16) {
    const uint32_t *np = mytab + (t0 + 16 < T ? t0 + 16 : t0);
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(nxt) : "s"(np));
#pragma unroll
    for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(z, o + (size_t)cur[i] * 8);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt));
    cur = nxt;
  }
}

// Every workgroup keeps its own residue class of 4 KiB blocks (block k belongs to workgroup k mod G, as in the plain
// fill) and leaves out the blocks whose bit is clear: bits[w TW + t / 32] bit t % 32 = block t G + w is to be written.
template <int MODE>   // 0: the 32 iterations of a bitmap word unrolled; 1: a loop, one store per trip; 2: a loop with a bare s_barrier per trip
__global__ __launch_bounds__(256) void fill_class(const uint32_t *__restrict__ bits, uint32_t T, uint32_t TW, u4 *out) {
  const u4 z = {0u, 0u, 0u, 0u};
  const uint32_t *mybits = bits + (size_t)blockIdx.x * TW;
  u4 *o = out + (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  uint32_t cur, nxt;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cur) : "s"(mybits));
  for (uint32_t t0 = 0; t0 < T; t0 += 32) {
    const uint32_t *np = mybits + (t0 + 32 < T ? t0 / 32 + 1 : t0 / 32);
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(nxt) : "s"(np));
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (cur & (1u << i)) __builtin_nontemporal_store(z, o + (size_t)(t0 + i) * step);
    } else {
      u4 *oo = o + (size_t)t0 * step;
#pragma unroll 1
      for (int i = 0; i < 32; ++i, oo += step) {
        if ((cur >> i) & 1u) __builtin_nontemporal_store(z, oo);
        if (MODE == 2) __builtin_amdgcn_s_barrier();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt));
    cur = nxt;
  }
}

// rows [ya, yb) of every slot, whole rows, zeros: grid (strips of 16 rows, slots)
__global__ __launch_bounds__(256) void fill_rows(const int2 *__restrict__ rows, int H, int W16, u4 *out) {
  const u4 z = {0u, 0u, 0u, 0u};
  const int slot = blockIdx.y;
  const int2 r = rows[slot];
  const int ra = max((int)blockIdx.x * 16, r.x), rb = min((int)blockIdx.x * 16 + 16, r.y);
  if (ra >= rb) return;
  u4 *o = out + (size_t)slot * H * W16 + (size_t)ra * W16;
  const int n = (rb - ra) * W16;
  for (int i = threadIdx.x; i < n; i += 256) __builtin_nontemporal_store(z, o + i);
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
  const int S = 2982, H = 1080, W = 1920, W16 = W / 16;
  const size_t slot_bytes = (size_t)H * W, bytes = (size_t)S * slot_bytes, nvec = bytes / 16;
  u4 *buf; CK(hipMalloc(&buf, bytes));
  std::mt19937 rng(5);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<int2> rows(S);
  size_t skipped = 0;
  for (int s = 0; s < S; ++s) {
    const double h = std::exp(std::log(24.0) + U(rng) * (std::log(960.0) - std::log(24.0)));
    const double cy = 140 * 3 + U(rng) * 360 * 3 - 420;   // centres over the content rows of a letterboxed 1080p frame
    int ya = (int)std::floor(cy + 420 - h / 2), yb = (int)std::ceil(cy + 420 + h / 2);
    ya = std::max(0, std::min(H, ya)); yb = std::max(ya, std::min(H, yb));
    rows[s] = make_int2(ya, yb);
    skipped += (size_t)(yb - ya) * W;
  }
  // the table: runs between the boxes' rows (a run crosses the slot boundary), cut into 4 KiB blocks
  std::vector<uint32_t> tab;
  const bool aligned = getenv("FILL_ALIGNED") != nullptr;
  if (aligned) {   // every 4 KiB-aligned block that is not wholly inside one slot's box rows
    for (size_t p = 0; p < bytes; p += 4096) {
      const size_t s = p / slot_bytes;
      const size_t a = s * slot_bytes + (size_t)rows[s].x * W, b = s * slot_bytes + (size_t)rows[s].y * W;
      if (p >= a && p + 4096 <= b) continue;
      tab.push_back((uint32_t)(std::min(p, bytes - 4096) / 128));
    }
  } else {
  size_t pos = 0;   // bytes
  auto emit = [&](size_t a, size_t b) {
    for (size_t p = a; p < b; p += 4096) {
      size_t q = p + 4096 <= b ? p : (b - a >= 4096 ? b - 4096 : a);   // whole blocks only
      if (q + 4096 > bytes) q = bytes - 4096;
      tab.push_back((uint32_t)(q / 128));
    }
  };
  for (int s = 0; s < S; ++s) {
    const size_t a = (size_t)s * slot_bytes + (size_t)rows[s].x * W, b = (size_t)s * slot_bytes + (size_t)rows[s].y * W;
    emit(pos, a);
    pos = b;
  }
  emit(pos, bytes);
  }
  printf("slots %d, box rows hold %.1f %% of the bytes, table %zu entries (%.1f MB)\n", S, 100.0 * skipped / bytes, tab.size(), tab.size() * 4 / 1e6);
  uint32_t *dtab; CK(hipMalloc(&dtab, (tab.size() + 16 * 4096) * 4));
  auto upload = [&](int G) -> uint32_t {   // per-workgroup layout for a grid of G workgroups
    const uint32_t T = (uint32_t)(((tab.size() + G - 1) / G + 15) / 16 * 16);
    std::vector<uint32_t> tr((size_t)T * G, tab.back());
    for (size_t v = 0; v < tab.size(); ++v) tr[(v % G) * T + v / G] = tab[v];
    CK(hipMemcpy(dtab, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
    return T;
  };
  int2 *drows; CK(hipMalloc(&drows, S * 8)); CK(hipMemcpy(drows, rows.data(), S * 8, hipMemcpyHostToDevice));

  const float tm = run([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); });
  printf("memset           %.3f ms  %.0f GB/s\n", tm, bytes / tm / 1e6);
  for (int g : {128, 192, 256, 320, 512}) {
    const float a = run([&] { hipLaunchKernelGGL(fill_plain, dim3(g), dim3(256), 0, 0, buf, nvec); });
    const uint32_t T = upload(g);
    const float b = run([&] { hipLaunchKernelGGL(fill_table, dim3(g), dim3(256), 0, 0, dtab, T, buf); });
    printf("grid %4d: plain %.3f ms (%.0f GB/s) | table %.3f ms (%.0f GB/s of the %.2f GB it writes)\n", g, a, bytes / a / 1e6, b,
           tab.size() * 4096.0 / b / 1e6, tab.size() * 4096.0 / 1e9);
  }
  for (int g : {128, 256}) {
    const size_t nb = bytes / 4096;   // (the tail below a block is ignored by this probe)
    const uint32_t T = (uint32_t)((nb + g - 1) / g), TW = (T + 31) / 32;
    std::vector<uint32_t> bits((size_t)TW * g, 0u);
    size_t written = 0;
    for (size_t k = 0; k < nb; ++k) {
      const size_t p = k * 4096, sl = p / slot_bytes;
      const size_t a = sl * slot_bytes + (size_t)rows[sl].x * W, b = sl * slot_bytes + (size_t)rows[sl].y * W;
      if (p >= a && p + 4096 <= b) continue;
      const size_t w = k % g, t = k / g;
      bits[w * TW + t / 32] |= 1u << (t % 32);
      ++written;
    }
    uint32_t *dbits; CK(hipMalloc(&dbits, bits.size() * 4)); CK(hipMemcpy(dbits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> all(bits.size(), 0u);
    const uint32_t T2 = (uint32_t)(nb / g);   // control: every block t < T2 of every class, nothing beyond (the loops store wherever a bit is set)
    for (size_t w = 0; w < (size_t)g; ++w)
      for (uint32_t t = 0; t < T2; ++t) all[w * TW + t / 32] |= 1u << (t % 32);
    uint32_t *dall; CK(hipMalloc(&dall, all.size() * 4)); CK(hipMemcpy(dall, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    const float c0 = run([&] { hipLaunchKernelGGL(fill_class<0>, dim3(g), dim3(256), 0, 0, dbits, T, TW, buf); });
    const float c1 = run([&] { hipLaunchKernelGGL(fill_class<1>, dim3(g), dim3(256), 0, 0, dbits, T, TW, buf); });
    const float c2 = run([&] { hipLaunchKernelGGL(fill_class<2>, dim3(g), dim3(256), 0, 0, dbits, T, TW, buf); });
    const float d0 = run([&] { hipLaunchKernelGGL(fill_class<0>, dim3(g), dim3(256), 0, 0, dall, T2, TW, buf); });
    const float d1 = run([&] { hipLaunchKernelGGL(fill_class<1>, dim3(g), dim3(256), 0, 0, dall, T2, TW, buf); });
    const float d2 = run([&] { hipLaunchKernelGGL(fill_class<2>, dim3(g), dim3(256), 0, 0, dall, T2, TW, buf); });
    const double wb = written * 4096.0, fb = (double)T2 * g * 4096.0;
    printf("grid %4d: residue classes + skip bitmap (%.2f GB): unrolled %.3f ms (%.0f GB/s) | loop %.3f (%.0f) | loop + s_barrier %.3f (%.0f)\n", g, wb / 1e9,
           c0, wb / c0 / 1e6, c1, wb / c1 / 1e6, c2, wb / c2 / 1e6);
    printf("grid %4d: the same kernels, nothing skipped (%.2f GB): unrolled %.3f ms (%.0f GB/s) | loop %.3f (%.0f) | loop + s_barrier %.3f (%.0f)\n", g, fb / 1e9,
           d0, fb / d0 / 1e6, d1, fb / d1 / 1e6, d2, fb / d2 / 1e6);
    CK(hipFree(dall));
    CK(hipFree(dbits));
  }
  for (int rep = 0; rep < 4; ++rep) {
    const float a = run([&] { hipLaunchKernelGGL(fill_plain, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b1 = run([&] { hipLaunchKernelGGL(fill_plain_bar<1>, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b2 = run([&] { hipLaunchKernelGGL(fill_plain_bar<2>, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b4 = run([&] { hipLaunchKernelGGL(fill_plain_bar<4>, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float m = run([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }, 20);
    printf("A/B %d: plain %.4f ms (%.0f) | barrier every trip %.4f (%.0f) | every 2nd %.4f (%.0f) | every 4th %.4f (%.0f) | memset %.4f (%.0f)\n", rep, a, bytes / a / 1e6,
           b1, bytes / b1 / 1e6, b2, bytes / b2 / 1e6, b4, bytes / b4 / 1e6, m, bytes / m / 1e6);
  }
  for (int rep = 0; rep < 2; ++rep) {
    const float a = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 512>), dim3(256), dim3(512), 0, 0, buf, nvec); }, 20);
    const float b = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 512>), dim3(128), dim3(512), 0, 0, buf, nvec); }, 20);
    const float c = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 1024>), dim3(256), dim3(1024), 0, 0, buf, nvec); }, 20);
    const float d = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 1024>), dim3(128), dim3(1024), 0, 0, buf, nvec); }, 20);
    const float e = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 128>), dim3(256), dim3(128), 0, 0, buf, nvec); }, 20);
    const float f = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 256>), dim3(512), dim3(256), 0, 0, buf, nvec); }, 20);
    printf("barrier every trip %d: 256 x 512 %.4f ms (%.0f) | 128 x 512 %.4f (%.0f) | 256 x 1024 %.4f (%.0f) | 128 x 1024 %.4f (%.0f) | 256 x 128 %.4f (%.0f) | 512 x 256 %.4f (%.0f)\n", rep,
           a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6, e, bytes / e / 1e6, f, bytes / f / 1e6);
  }
  for (int rep = 0; rep < 3; ++rep) {
    const float a = run([&] { hipLaunchKernelGGL(fill_bar_map<0>, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b = run([&] { hipLaunchKernelGGL(fill_bar_map<1>, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float c = run([&] { hipLaunchKernelGGL(fill_bar_2k, dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float d = run([&] { hipLaunchKernelGGL(fill_bar_2k, dim3(128), dim3(256), 0, 0, buf, nvec); }, 20);
    printf("barrier-paced %d: identity map %.4f ms (%.0f) | XCD-contiguous map %.4f (%.0f) | 2 KiB per wave and trip, grid 256 %.4f (%.0f) | grid 128 %.4f (%.0f)\n", rep,
           a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6);
  }
  for (int rep = 0; rep < 2; ++rep) {
    const float r0 = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 256>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float a = run([&] { hipLaunchKernelGGL((fill_throttle<1, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b = run([&] { hipLaunchKernelGGL((fill_throttle<2, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float c = run([&] { hipLaunchKernelGGL((fill_throttle<4, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float d = run([&] { hipLaunchKernelGGL((fill_throttle<8, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float e = run([&] { hipLaunchKernelGGL((fill_throttle<16, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float f = run([&] { hipLaunchKernelGGL((fill_throttle<2, false>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float g = run([&] { hipLaunchKernelGGL((fill_throttle<4, false>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float h = run([&] { hipLaunchKernelGGL((fill_throttle<8, false>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    printf("stores in flight %d: barrier only %.4f ms (%.0f) | barrier + vmcnt 1/2/4/8/16: %.4f (%.0f) %.4f (%.0f) %.4f (%.0f) %.4f (%.0f) %.4f (%.0f) | vmcnt 2/4/8 alone: %.4f (%.0f) %.4f (%.0f) %.4f (%.0f)\n", rep,
           r0, bytes / r0 / 1e6, a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6, e, bytes / e / 1e6, f, bytes / f / 1e6, g, bytes / g / 1e6, h, bytes / h / 1e6);
  }
  for (int rep = 0; rep < 6; ++rep) {
    const float r0 = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 256>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float a = run([&] { hipLaunchKernelGGL((fill_throttle<6, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float b = run([&] { hipLaunchKernelGGL((fill_throttle<8, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float c = run([&] { hipLaunchKernelGGL((fill_throttle<10, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    const float d = run([&] { hipLaunchKernelGGL((fill_throttle<12, true>), dim3(256), dim3(256), 0, 0, buf, nvec); }, 20);
    printf("A/B throttle %d: barrier only %.4f (%.0f) | + vmcnt 6 %.4f (%.0f) | 8 %.4f (%.0f) | 10 %.4f (%.0f) | 12 %.4f (%.0f)\n", rep, r0, bytes / r0 / 1e6,
           a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6);
  }
  for (int gsz : {224, 240, 248, 256}) {
    const float a = run([&] { hipLaunchKernelGGL((fill_plain_bar<1, 256>), dim3(gsz), dim3(256), 0, 0, buf, nvec); }, 20);
    printf("barrier-paced, grid %d: %.4f ms (%.0f)\n", gsz, a, bytes / a / 1e6);
  }
  const float tr = run([&] { hipLaunchKernelGGL(fill_rows, dim3((H + 15) / 16, S), dim3(256), 0, 0, drows, H, W16, buf); });
  printf("rows under the boxes, zeros only: %.3f ms (%.0f GB/s of %.2f GB)\n", tr, skipped / tr / 1e6, skipped / 1e9);
  // both in a row (the pair that would replace fill + box pass)
  const uint32_t T256 = upload(256);
  const float tb = run([&] {
    hipLaunchKernelGGL(fill_table, dim3(256), dim3(256), 0, 0, dtab, T256, buf);
    hipLaunchKernelGGL(fill_rows, dim3((H + 15) / 16, S), dim3(256), 0, 0, drows, H, W16, buf);
  });
  printf("table fill + rows: %.3f ms (%.0f GB/s of all %.2f GB)\n", tb, bytes / tb / 1e6, bytes / 1e9);
  // check: every byte zero after a 0xff prefill
  CK(hipMemset(buf, 0xff, bytes));
  hipLaunchKernelGGL(fill_table, dim3(256), dim3(256), 0, 0, dtab, T256, buf);
  hipLaunchKernelGGL(fill_rows, dim3((H + 15) / 16, S), dim3(256), 0, 0, drows, H, W16, buf);
  CK(hipDeviceSynchronize());
  std::vector<unsigned char> host(slot_bytes * 4);
  CK(hipMemcpy(host.data(), (unsigned char *)buf + slot_bytes * 100, host.size(), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (unsigned char c : host) bad += c != 0;
  printf("coverage check (4 slots): %zu bytes not written\n", bad);
  return 0;
}
