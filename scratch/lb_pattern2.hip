// Round 3 probe: what does the letterbox's ADDRESS PATTERN cost, with no arithmetic and no LDS?  Persistent workgroups
// (256 lanes) walk units of 2 output rows of a (64,3,640,640) fp32 tensor; content rows 140..499 read their source
// rows (1080p: row 3dy+1, 5760 B; 4K: rows 6dy+2 and 6dy+3, 11520 B each) with 16-byte nt loads, fold them and store
// the unit's 3 x 2 x 2560 bytes with 16-byte nt stores.  Variants: planar output as the real tensor (three segments a
// plane apart) or the same bytes written as ONE linear stream; source rows where they lie (strided) or read as ONE
// linear stream of the same length.  Cold sources (3 rotating buffers).
//   hipcc -O3 --offload-arch=gfx950 scratch/lb_pattern2.hip -o scratch/lb_pattern2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int S = 640, B = 64, UPI = 320, TOP = 140, CH = 360;

// RPO: source rows per output row; ROWV: 16-byte chunks per source row; STEP: source rows between output rows
template <int RPO, int ROWV, int STEP, bool PLANAR, bool STRIDED>
__global__ __launch_bounds__(256) void lbpat(u4 *__restrict__ out, const u4 *__restrict__ src, int src_rows) {
  constexpr int NV = 2 * RPO * ROWV;            // chunks a content unit reads
  constexpr int KF = (NV + 255) / 256;
  const int tid = threadIdx.x;
  for (int u = blockIdx.x; u < B * UPI; u += gridDim.x) {
    const int b = u / UPI, strip = u - b * UPI, y0 = 2 * strip;
    u4 acc = {1, 2, 3, 4};
    if (y0 >= TOP && y0 < TOP + CH) {
      u4 v[KF];
#pragma unroll
      for (int j = 0; j < KF; ++j) {
        const int c = tid + 256 * j;
        if (c < NV) {
          const int rr = c / ROWV, x = c - rr * ROWV;          // constant divisor
          size_t a;
          if (STRIDED) {
            const int oy = y0 - TOP + rr / RPO;
            const int sy = oy * STEP + (STEP / 2 - (RPO == 2 ? 1 : 0)) + (rr % RPO);
            a = ((size_t)b * src_rows + sy) * ROWV + x;
          } else {
            a = ((size_t)b * (CH / 2) + (y0 - TOP) / 2) * NV + c;   // the same bytes as one linear stream
          }
          v[j] = __builtin_nontemporal_load(src + a);
        } else v[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < KF; ++j) acc ^= v[j];
    }
    // 3 planes x 2 rows x 160 chunks = 960 chunks
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      if (c < 960) {
        size_t a;
        if (PLANAR) {
          const int p = c / 320, w = c - p * 320;
          a = ((size_t)(b * 3 + p) * S + y0) * (S / 4) + w;
        } else {
          a = (size_t)u * 960 + c;
        }
        __builtin_nontemporal_store(acc, out + a);
      }
    }
  }
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
  const size_t src_bytes = (size_t)B * 2160 * 11520;   // 4K batch
  u4 *src[3], *out;
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&src[i], src_bytes)); CK(hipMemset(src[i], 0x11 * (i + 1), src_bytes)); }
  CK(hipMalloc(&out, (size_t)B * 3 * S * S * 4));
  CK(hipDeviceSynchronize());
  auto rep = [&](const char *name, double mb, double us) { printf("%-64s %8.1f us  %7.1f GB/s of %.1f MB\n", name, us, mb * 1e6 / 1e9 / (us * 1e-6), mb); fflush(stdout); };
  for (int grid : {256, 512, 768, 1024, 2048}) {
    char nm[160];
    snprintf(nm, sizeof nm, "1080p planar out, strided rows, grid %d", grid);
    rep(nm, 447.3, time_us([&](int r) { lbpat<1, 360, 3, true, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p linear out, strided rows, grid %d", grid);
    rep(nm, 447.3, time_us([&](int r) { lbpat<1, 360, 3, false, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p planar out, linear rows,  grid %d", grid);
    rep(nm, 447.3, time_us([&](int r) { lbpat<1, 360, 3, true, false><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p linear out, linear rows,  grid %d", grid);
    rep(nm, 447.3, time_us([&](int r) { lbpat<1, 360, 3, false, false><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "4K    planar out, strided rows, grid %d", grid);
    rep(nm, 845.4, time_us([&](int r) { lbpat<2, 720, 6, true, true><<<grid, 256>>>(out, src[r % 3], 2160); }));
    snprintf(nm, sizeof nm, "4K    linear out, strided rows, grid %d", grid);
    rep(nm, 845.4, time_us([&](int r) { lbpat<2, 720, 6, false, true><<<grid, 256>>>(out, src[r % 3], 2160); }));
    snprintf(nm, sizeof nm, "4K    planar out, linear rows,  grid %d", grid);
    rep(nm, 845.4, time_us([&](int r) { lbpat<2, 720, 6, true, false><<<grid, 256>>>(out, src[r % 3], 2160); }));
    snprintf(nm, sizeof nm, "4K    linear out, linear rows,  grid %d", grid);
    rep(nm, 845.4, time_us([&](int r) { lbpat<2, 720, 6, false, false><<<grid, 256>>>(out, src[r % 3], 2160); }));
  }
  return 0;
}
