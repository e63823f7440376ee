// Round 3 probe: where does a hand-written device-to-device copy stop on this MI355X?
// The guide measures 6.29 TB/s for a float4 copy; round 2 quoted 5.2 TB/s from torch's copy_ and from letterbox-shaped
// pattern kernels at 447 MB.  Kernels below move bytes only: 16-byte loads and stores, plain or non-temporal, U loads in
// flight per lane, simple (one element per lane, huge grid) / grid-stride / persistent block-contiguous forms, at
// 0.45 / 0.85 / 4 GB of read + write bytes.  Source / destination pairs rotate over 3 buffers, so nothing is served from
// the 256 MiB Infinity Cache.  Also: asymmetric streams with the letterbox's read : write ratio (1080p 133 : 315 MB,
// 4K 531 : 315 MB).
//   hipcc -O3 --offload-arch=gfx950 scratch/copy_ceiling.hip -o scratch/copy_ceiling && scratch/copy_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u4 ld(const u4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u4 *p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// one 16-byte element per lane, grid = n / 256
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_simple(u4 *dst, const u4 *src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) st<NTS>(dst + i, ld<NTL>(src + i));
}

// grid-stride: every iteration a lane loads U elements `gridDim * 256` apart, then stores them
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_gs(u4 *dst, const u4 *src, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ld<NTL>(src + i + k * stride);
#pragma unroll
    for (int k = 0; k < U; ++k) st<NTS>(dst + i + k * stride, v[k]);
  }
  for (; i < n; i += stride) st<NTS>(dst + i, ld<NTL>(src + i));
}

// persistent, block-contiguous: a workgroup takes chunks of U * 4 KiB in address order (chunk c -> block c % grid), all
// U loads of the chunk in flight, then the stores; the loads of the NEXT chunk are issued before this chunk's stores (PIPE)
template <int U, bool NTL, bool NTS, bool PIPE>
__global__ __launch_bounds__(256) void copy_chunk(u4 *dst, const u4 *src, size_t n) {
  const size_t chunk = (size_t)U * 256;
  const size_t nchunk = n / chunk;   // n is a multiple of chunk (host)
  u4 v[U], w[U];
  size_t c = blockIdx.x;
  if (c >= nchunk) return;
#pragma unroll
  for (int k = 0; k < U; ++k) v[k] = ld<NTL>(src + c * chunk + k * 256 + threadIdx.x);
  for (; c < nchunk; c += gridDim.x) {
    const size_t nx = c + gridDim.x;
    if (PIPE && nx < nchunk) {
#pragma unroll
      for (int k = 0; k < U; ++k) w[k] = ld<NTL>(src + nx * chunk + k * 256 + threadIdx.x);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) st<NTS>(dst + c * chunk + k * 256 + threadIdx.x, v[k]);
    if (PIPE) {
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = w[k];
    } else if (nx < nchunk) {
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = ld<NTL>(src + nx * chunk + k * 256 + threadIdx.x);
    }
  }
}

// asymmetric stream: per chunk of 4 KiB * UW written, UR * 4 KiB are read (the letterbox's ratio); the loaded words are
// folded into the stored value so the loads stay
template <int UR, int UW>
__global__ __launch_bounds__(256) void mix_chunk(u4 *dst, const u4 *src, size_t nchunk) {
  for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    u4 v[UR];
#pragma unroll
    for (int k = 0; k < UR; ++k) v[k] = ld<true>(src + (c * UR + k) * 256 + threadIdx.x);
    u4 a = v[0];
#pragma unroll
    for (int k = 1; k < UR; ++k) a ^= v[k];
#pragma unroll
    for (int k = 0; k < UW; ++k) st<true>(dst + (c * UW + k) * 256 + threadIdx.x, a);
  }
}

template <int U>
__global__ __launch_bounds__(256) void fill_chunk(u4 *dst, size_t nchunk) {
  const u4 a = {1, 2, 3, 4};
  for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x)
#pragma unroll
    for (int k = 0; k < U; ++k) st<true>(dst + (c * U + k) * 256 + threadIdx.x, a);
}

template <int U>
__global__ __launch_bounds__(256) void read_chunk(u4 *dst, const u4 *src, size_t nchunk) {
  u4 a = {0, 0, 0, 0};
  for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ld<true>(src + (c * U + k) * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < U; ++k) a ^= v[k];
  }
  if (a.x == 0x12345678u) dst[threadIdx.x] = a;   // never true for the test pattern: keeps the loads
}

static hipEvent_t e0, e1;
template <typename F>
static double time_us(F &&launch, int reps = 15) {
  std::vector<float> t;
  for (int r = 0; r < reps + 3; ++r) {
    CK(hipEventRecord(e0, 0));
    launch(r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;
}

int main() {
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t kMax = (size_t)2 << 30;   // bytes per buffer at the largest size
  u4 *src[3], *dst[3];
  for (int i = 0; i < 3; ++i) {
    CK(hipMalloc(&src[i], kMax));
    CK(hipMalloc(&dst[i], kMax));
    CK(hipMemset(src[i], 0x5a + i, kMax));
    CK(hipMemset(dst[i], 0, kMax));
  }
  CK(hipDeviceSynchronize());
  const double total_mb[3] = {447.3, 845.4, 4096.0};
  for (int si = 0; si < 3; ++si) {
    // bytes per direction, rounded to a multiple of 1 MiB so that every chunk form divides it
    size_t bytes = (size_t)(total_mb[si] * 1e6 / 2);
    bytes = bytes / (1 << 20) * (1 << 20);
    if (bytes > kMax) bytes = kMax;
    const size_t n = bytes / 16;
    const double gb = 2.0 * bytes / 1e9;
    printf("\n== copy of %.1f MB (read + write %.1f MB), cold (3 rotating buffer pairs) ==\n", bytes / 1e6, 2 * bytes / 1e6);
    auto rep = [&](const char *name, double us) { printf("%-58s %8.1f us  %7.1f GB/s read+write\n", name, us, gb / (us * 1e-6)); fflush(stdout); };
    rep("hipMemcpyDtoDAsync", time_us([&](int r) { CK(hipMemcpyAsync(dst[r % 3], src[r % 3], bytes, hipMemcpyDeviceToDevice, 0)); }));
    rep("simple, plain ld / plain st", time_us([&](int r) { copy_simple<false, false><<<dim3((unsigned)((n + 255) / 256)), 256>>>(dst[r % 3], src[r % 3], n); }));
    rep("simple, nt ld / nt st", time_us([&](int r) { copy_simple<true, true><<<dim3((unsigned)((n + 255) / 256)), 256>>>(dst[r % 3], src[r % 3], n); }));
    rep("simple, plain ld / nt st", time_us([&](int r) { copy_simple<false, true><<<dim3((unsigned)((n + 255) / 256)), 256>>>(dst[r % 3], src[r % 3], n); }));
    for (int grid : {256, 512, 1024, 2048, 4096}) {
      char nm[128];
      snprintf(nm, sizeof nm, "grid-stride U=4 nt/nt, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_gs<4, true, true><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "grid-stride U=8 nt/nt, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_gs<8, true, true><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "grid-stride U=8 plain/plain, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_gs<8, false, false><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "chunk U=8 nt/nt, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_chunk<8, true, true, false><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "chunk U=8 nt/nt pipelined, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_chunk<8, true, true, true><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "chunk U=16 nt/nt pipelined, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_chunk<16, true, true, true><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
      snprintf(nm, sizeof nm, "chunk U=4 nt/nt pipelined, grid %d", grid);
      rep(nm, time_us([&](int r) { copy_chunk<4, true, true, true><<<grid, 256>>>(dst[r % 3], src[r % 3], n); }));
    }
  }
  // pure directions and the letterbox's ratios: 315 MB written = 76800 chunks of 4 KiB
  {
    const size_t wchunks = 76800;   // x 4 KiB = 314.6 MB
    auto rep2 = [&](const char *name, double mb, double us) { printf("%-58s %8.1f us  %7.1f GB/s of %.1f MB\n", name, us, mb * 1e6 / 1e9 / (us * 1e-6), mb); fflush(stdout); };
    printf("\n== pure and asymmetric streams (cold) ==\n");
    for (int grid : {256, 512, 1024, 2048}) {
      char nm[128];
      snprintf(nm, sizeof nm, "fill 314.6 MB, U=4, grid %d", grid);
      rep2(nm, 314.6, time_us([&](int r) { fill_chunk<4><<<grid, 256>>>(dst[r % 3], wchunks / 4); }));
      snprintf(nm, sizeof nm, "read 530.8 MB, U=8, grid %d", grid);
      rep2(nm, 530.8, time_us([&](int r) { read_chunk<8><<<grid, 256>>>(dst[r % 3], src[r % 3], 129600 / 8); }));
      // 1080p: 132.7 MB read : 314.6 MB written ~ 3 : 7 -> UR 3, UW 7 over 10971 chunks (read 134.8, write 314.6)
      snprintf(nm, sizeof nm, "mix 3 read : 7 write (1080p letterbox), grid %d", grid);
      rep2(nm, 449.4, time_us([&](int r) { mix_chunk<3, 7><<<grid, 256>>>(dst[r % 3], src[r % 3], wchunks / 7); }));
      // 4K: 530.8 : 314.6 ~ 27 : 16 -> UR 27 is too many registers; 5 : 3 (524.3 : 314.6)
      snprintf(nm, sizeof nm, "mix 5 read : 3 write (4K letterbox), grid %d", grid);
      rep2(nm, 838.9, time_us([&](int r) { mix_chunk<5, 3><<<grid, 256>>>(dst[r % 3], src[r % 3], wchunks / 3); }));
      snprintf(nm, sizeof nm, "mix 10 read : 6 write (4K letterbox), grid %d", grid);
      rep2(nm, 838.9, time_us([&](int r) { mix_chunk<10, 6><<<grid, 256>>>(dst[r % 3], src[r % 3], wchunks / 6); }));
    }
  }
  // correctness of one copy form (the others share the indexing style)
  {
    const size_t n = ((size_t)256 << 20) / 16;
    CK(hipMemset(dst[0], 0, n * 16));
    copy_chunk<8, true, true, true><<<1024, 256>>>(dst[0], src[1], n);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(1024);
    CK(hipMemcpy(h.data(), reinterpret_cast<char *>(dst[0]) + n * 16 - 4096, 4096, hipMemcpyDeviceToHost));
    bool ok = true;
    for (uint32_t x : h) ok = ok && x == 0x5b5b5b5bu;
    printf("\ncopy check: %s\n", ok ? "ok" : "MISMATCH");
  }
  return 0;
}
