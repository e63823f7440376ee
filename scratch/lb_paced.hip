// probe (r03, second session): the letterbox as a PACED sweep.  The zero fill showed what the memory system rewards: one
// workgroup per CU, each trip one 4 KiB block, the blocks of a trip adjacent (a 1 MiB window sweeping the output in
// address order), the four waves kept in step.  Here the same sweep over the (64,3,640,640) fp32 tensor, and every block
// also READS the source rows its output rows need (1080p -> 640x360: source row 3 dy + 1 of each content row, whole
// rows of 5760 bytes, all three planes of an image read the same rows), DEPTH trips ahead, with no arithmetic: the
// I/O skeleton of such a kernel.  Cold sources (3 rotating buffers).
//   hipcc -O3 --offload-arch=gfx950 scratch/lb_paced.hip -o scratch/lb_paced
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int S = 640, B = 64, TOP = 140, CH = 360, BLK_PER_PLANE = S * S / 1024, BLK_PER_IMG = 3 * BLK_PER_PLANE;
constexpr int NBLK = B * BLK_PER_IMG;     // 76,800 blocks of 4 KiB

// the source chunks (16 B) block k needs: the rows of its content output rows.  STEP 3 / ROWV 360 (1080p) or STEP 6 /
// ROWV 720 and two rows per output row (4K).  Returns how many chunks and, for chunk c, its address.
template <int STEP, int ROWV, int RPO>
__device__ __forceinline__ int block_rows(int k, int &b, int &ya, int &yb) {
  b = k / BLK_PER_IMG;
  const int q = (k - b * BLK_PER_IMG) % BLK_PER_PLANE;
  const int px0 = q * 1024;
  ya = max(px0 / S, TOP);
  yb = min((px0 + 1023) / S + 1, TOP + CH);
  return ya < yb ? (yb - ya) * RPO * ROWV : 0;
}

template <int STEP, int ROWV, int RPO, int KF, int DEPTH, bool READ>
__global__ __launch_bounds__(256) void lbpaced(u4 *__restrict__ out, const u4 *__restrict__ src, int src_rows) {
  const int tid = threadIdx.x, G = gridDim.x;
  const int T = (NBLK + G - 1) / G;
  u4 v[DEPTH][KF];
  auto issue = [&](int t, u4 (&dst)[KF]) {
    const int k = t * G + blockIdx.x;
#pragma unroll
    for (int j = 0; j < KF; ++j) dst[j] = u4{0u, 0u, 0u, 0u};
    if (!READ || k >= NBLK) return;
    int b, ya, yb;
    const int n = block_rows<STEP, ROWV, RPO>(k, b, ya, yb);
#pragma unroll
    for (int j = 0; j < KF; ++j) {
      const int c = tid + 256 * j;
      if (c < n) {
        const int rr = c / ROWV, x = c - rr * ROWV;
        const int sy = (ya - TOP + rr / RPO) * STEP + (STEP / 2 - (RPO == 2 ? 1 : 0)) + (rr % RPO);
        dst[j] = __builtin_nontemporal_load(src + ((size_t)b * src_rows + sy) * ROWV + x);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, v[d]);
  for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d, k = t * G + blockIdx.x;
      u4 acc = {1u, 2u, 3u, 4u};
#pragma unroll
      for (int j = 0; j < KF; ++j) acc ^= v[d][j];
      issue(t + DEPTH, v[d]);
      if (t < T && k < NBLK) __builtin_nontemporal_store(acc, out + (size_t)k * 256 + tid);
      __builtin_amdgcn_s_barrier();
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
  auto rep = [&](const char *name, double us) { printf("%-72s %8.1f us\n", name, us); fflush(stdout); };
  rep("output alone, paced sweep (no reads), grid 256", time_us([&](int r) { lbpaced<3, 360, 1, 5, 4, false><<<256, 256>>>(out, src[r % 3], 1080); }));
  for (int grid : {256, 512}) {
    char nm[160];
    snprintf(nm, sizeof nm, "1080p: rows read 2 trips ahead, grid %d", grid);
    rep(nm, time_us([&](int r) { lbpaced<3, 360, 1, 5, 2, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p: rows read 4 trips ahead, grid %d", grid);
    rep(nm, time_us([&](int r) { lbpaced<3, 360, 1, 5, 4, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p: rows read 8 trips ahead, grid %d", grid);
    rep(nm, time_us([&](int r) { lbpaced<3, 360, 1, 5, 8, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
    snprintf(nm, sizeof nm, "1080p: rows read 16 trips ahead, grid %d", grid);
    rep(nm, time_us([&](int r) { lbpaced<3, 360, 1, 5, 16, true><<<grid, 256>>>(out, src[r % 3], 1080); }));
  }
  return 0;
}
