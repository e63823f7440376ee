// Operator-list revert on the device, batched.
//
// Replaces the per-mask / per-segment Python loops of
//   pipeline_utils.revert_to_origin          lmi_utils/gadget_utils/pipeline_utils.py:422-476
//   pipeline_utils.revert_mask(s)_to_origin  lmi_utils/gadget_utils/pipeline_utils.py:374-418
//     -> resize_image :23-68 (F.interpolate bilinear, align_corners=False; integer masks go through fp32 and are
//        truncated on the way back), fit_im_to_size :72-142 (centred crop / zero pad), torch.flip
// as called from Yolov8.predict (object_detectors/yolov8_lmi/model.py:308-318).  Semantics: oracle/geometry.py.
//
// Kernels
//  revert_points_kernel      one lane per (x, y) pair; the operator list (<= 16 steps, by value) applied last to
//                            first in fp32 with IEEE division, then rint (half to even) and clamp at 0.
//  revert_masks_kernel<T>    one stage = [centred crop / zero pad] -> [bilinear resize] -> [flip], fused: the crop
//                            and pad are index arithmetic on the resize taps, the flip mirrors the store address.
//                            16 bytes per lane per store; write-bound when it enlarges (640x360 -> 1920x1080).
//  revert_resize_kernel<T>   the resize stage with the source rows staged in LDS (outputs with dh + dw > 128).
//  revert_resize_lines_kernel  the same for fp32 rows that are not whole 128-byte lines: results through LDS, stores
//                            cut by absolute address.
#include "common.h"
#include <mutex>

namespace {

struct PointOps {
  int n;
  int kind[LMI_MAX_POINT_OPS];
  float x[LMI_MAX_POINT_OPS], y[LMI_MAX_POINT_OPS];
};

__global__ __launch_bounds__(256) void revert_points_kernel(const float *pts, float *out, long long npairs, PointOps P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  float x = pts[2 * i], y = pts[2 * i + 1];
  for (int k = P.n - 1; k >= 0; --k) {  // operations are undone last to first
    const float a = P.x[k], b = P.y[k];
    switch (P.kind[k]) {
      case LMI_POINT_DIV: x = x / a; y = y / b; break;
      case LMI_POINT_SUB: x = x - a; y = y - b; break;
      case LMI_POINT_FLIP_X: x = a - x; break;
      case LMI_POINT_FLIP_Y: y = b - y; break;
      case LMI_POINT_FLIP_XY: x = a - x; y = b - y; break;
      default: break;
    }
  }
  out[2 * i] = fmaxf(rintf(x), 0.0f);
  out[2 * i + 1] = fmaxf(rintf(y), 0.0f);
}

struct MaskStage {
  const void *src;
  void *dst;
  long long m;
  int sh, sw;      // source planes
  int oy, ox;      // virtual image V[y][x] = src[y + oy][x + ox], zero outside the source
  int vh, vw;      // size of V (after the centred crop / pad)
  int dh, dw;      // output planes
  int resize;      // 1: bilinear V -> (dh, dw); 0: dh == vh, dw == vw, copy
  int flip_lr, flip_ud;
};

// aten upsample_bilinear2d, align_corners=False, with the contractions of torch's x86 build (oracle/geometry.py:
// only the fused forms reproduce F.interpolate bit for bit): the source coordinate is ONE fused multiply-add; then
// either aten's separable kernel -- each pair of taps combined as fma(a, wa, b*wb), horizontal first -- or, for
// outputs with dh + dw <= 128, its per-pixel kernel with four weight products (aten_linear_variant in the oracle).
__device__ __forceinline__ void bilinear_src(float scale, int d, int in_n, int &i0, int &i1, float &l0, float &l1) {
  float s = fmaf(scale, (float)d + 0.5f, -0.5f);
  s = fmaxf(s, 0.0f);
  i0 = min((int)s, in_n - 1);
  i1 = i0 + ((i0 < in_n - 1) ? 1 : 0);
  l1 = fminf(fmaxf(s - (float)i0, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}

template <typename T>
struct Px;
template <>
struct Px<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ float load(const float *p) { return *p; }
  static __device__ __forceinline__ void load_group(const float *p, float (&f)[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef f4 f4_anywhere __attribute__((aligned(4)));
    const f4 t = *reinterpret_cast<const f4_anywhere *>(p);
    f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
  }
  static __device__ __forceinline__ float from_interp(float v) { return v; }
  // one 16-byte store at whatever address the row gives (widths that are not whole chunks): the target runs in
  // unaligned-access mode, the instruction is the same global_store_dwordx4
  static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef f4 f4_anywhere __attribute__((aligned(4)));
    f4 o = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<f4_anywhere *>(p));
  }
};
template <>
struct Px<uint8_t> {
  static constexpr int N = 16;
  static __device__ __forceinline__ float load(const uint8_t *p) { return (float)*p; }
  static __device__ __forceinline__ void load_group(const uint8_t *p, float (&f)[16]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef u4 u4_anywhere __attribute__((aligned(1)));
    const u4 t = *reinterpret_cast<const u4_anywhere *>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = (float)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
  }
  // `.to(uint8)` of the interpolated fp32 value: truncation toward zero (values stay inside 0..255)
  static __device__ __forceinline__ uint8_t from_interp(float v) { return (uint8_t)(int)v; }
  static __device__ __forceinline__ void store(uint8_t *p, const uint8_t (&v)[16]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = v[4 * q] | (v[4 * q + 1] << 8) | (v[4 * q + 2] << 16) | ((uint32_t)v[4 * q + 3] << 24);
    typedef u4 u4_anywhere __attribute__((aligned(1)));
    u4 o = {w[0], w[1], w[2], w[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<u4_anywhere *>(p));
  }
};

// grid: x = chunk columns (N pixels each) in groups of 64 lanes, y = output rows in groups of 4, z = mask
template <typename T>
__global__ __launch_bounds__(256) void revert_masks_kernel(MaskStage S) {
  constexpr int N = Px<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = blockIdx.y * 4 + wave;  // output row before the flip
  const int x0 = (blockIdx.x * 64 + lane) * N;
  if (y >= S.dh || x0 >= S.dw) return;
  const long long mi = blockIdx.z;
  const T *src = reinterpret_cast<const T *>(S.src) + (size_t)mi * S.sh * S.sw;
  T *dst = reinterpret_cast<T *>(S.dst) + (size_t)mi * S.dh * S.dw;

  auto tap = [&](int vy, int vx) -> float {  // V[vy][vx] as fp32
    const int sy = vy + S.oy, sx = vx + S.ox;
    if (sy < 0 || sy >= S.sh || sx < 0 || sx >= S.sw) return 0.0f;
    return Px<T>::load(src + (size_t)sy * S.sw + sx);
  };

  T v[N];
  if (S.resize) {
    const float scale_h = (float)S.vh / (float)S.dh, scale_w = (float)S.vw / (float)S.dw;
    const bool weights4 = S.dh + S.dw <= 128;
    int iy0, iy1; float ly0, ly1;
    bilinear_src(scale_h, y, S.vh, iy0, iy1, ly0, ly1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int x = x0 + k;
      int ix0, ix1; float lx0, lx1;
      bilinear_src(scale_w, min(x, S.dw - 1), S.vw, ix0, ix1, lx0, lx1);
      float r;
      if (weights4) {
        const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
        r = fmaf(w11, tap(iy1, ix1), fmaf(w10, tap(iy1, ix0), fmaf(w00, tap(iy0, ix0), w01 * tap(iy0, ix1))));
      } else {
        const float t0 = fmaf(tap(iy0, ix0), lx0, tap(iy0, ix1) * lx1);
        const float t1 = fmaf(tap(iy1, ix0), lx0, tap(iy1, ix1) * lx1);
        r = fmaf(t0, ly0, t1 * ly1);
      }
      v[k] = Px<T>::from_interp(r);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int sy = y + S.oy, sx = x0 + k + S.ox;
      const bool in = sy >= 0 && sy < S.sh && sx >= 0 && sx < S.sw && x0 + k < S.dw;
      v[k] = in ? src[(size_t)sy * S.sw + sx] : (T)0;
    }
  }

  const int yo = S.flip_ud ? S.dh - 1 - y : y;
  T *row = dst + (size_t)yo * S.dw;
  const bool full = x0 + N <= S.dw;
  if (!S.flip_lr) {
    if (full) {
      Px<T>::store(row + x0, v);
    } else {
      for (int k = 0; k < N && x0 + k < S.dw; ++k) row[x0 + k] = v[k];
    }
  } else {
    const int xe = S.dw - x0 - N;  // mirrored chunk starts here
    if (full) {
      T r[N];
#pragma unroll
      for (int k = 0; k < N; ++k) r[k] = v[N - 1 - k];
      Px<T>::store(row + xe, r);
    } else {
      for (int k = 0; k < N && x0 + k < S.dw; ++k) row[S.dw - 1 - (x0 + k)] = v[k];
    }
  }
}

// The virtual source rows the output rows [ra, rb) read, staged in LDS as fp32 (crop offset and zero pad folded in, a
// repeat of the last column so that the right-hand tap is always the next float); returns the first staged row.
// Ends with a workgroup barrier.
template <typename T>
__device__ __forceinline__ int stage_source_rows(const MaskStage &S, const T *src, int ra, int rb, float *L, int ls) {
  constexpr int N = Px<T>::N;
  const float scale_h = (float)S.vh / (float)S.dh;
  int i0, i1, j0, j1; float l0, l1;
  bilinear_src(scale_h, ra, S.vh, i0, i1, l0, l1);
  const int vy_first = i0;
  bilinear_src(scale_h, rb - 1, S.vh, j0, j1, l0, l1);
  const int nrows = j1 - vy_first + 1;   // fits the LDS buffer by the host's choice of rows_per_block
  // groups of N source pixels: one 16-byte load (at whatever address) where the group lies inside the source,
  // pixel by pixel at the pad border and for the repeated last column
  const int groups = (ls + N - 1) / N;
  // four groups per lane and pass, their global loads issued before the first is converted: a strip's staging is a
  // handful of passes, and one HBM round trip per pass was a third of a workgroup's lifetime
  constexpr int U = 4;
  const int total = nrows * groups;
  for (int base = 0; base < total; base += U * (int)blockDim.x) {
    float f[U][N];
    bool vec[U];
    int rr[U], cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * (int)blockDim.x + (int)threadIdx.x;
      const int r = idx / groups, c0 = (idx - r * groups) * N;
      const int sy = vy_first + r + S.oy, sx0 = c0 + S.ox;
      rr[u] = r; cc[u] = c0;
      vec[u] = idx < total && sy >= 0 && sy < S.sh && sx0 >= 0 && sx0 + N <= S.sw && c0 + N <= S.vw;
      if (vec[u]) Px<T>::load_group(src + (size_t)sy * S.sw + sx0, f[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * (int)blockDim.x + (int)threadIdx.x;
      if (idx >= total) continue;
      const int r = rr[u], c0 = cc[u];
      float *d = L + r * ls + c0;
      if (vec[u]) {
#pragma unroll
        for (int k = 0; k < N; k += 4) *reinterpret_cast<float4 *>(d + k) = make_float4(f[u][k], f[u][k + 1], f[u][k + 2], f[u][k + 3]);
      } else {
        const int sy = vy_first + r + S.oy;
        for (int k = 0; k < N && c0 + k < ls; ++k) {
          const int sx = min(c0 + k, S.vw - 1) + S.ox;
          d[k] = (sy >= 0 && sy < S.sh && sx >= 0 && sx < S.sw) ? Px<T>::load(src + (size_t)sy * S.sw + sx) : 0.0f;
        }
      }
    }
  }
  __syncthreads();
  return vy_first;
}

// The resize stage for outputs with dh + dw > 128 (aten's separable kernel), in the structure of mask_box_kernel:
// grid (strips of rows_per_block output rows, masks).  A workgroup stages the virtual source rows its strip reads
// (crop offset and zero pad folded in, converted to fp32, plus a repeat of the last column so that the right-hand tap
// is always the next float) in LDS; every lane owns one 16-byte chunk column, computes the horizontal weights of its
// pixels once and walks down its rows, keeping the horizontally interpolated source-row pair in registers across the
// output rows that share it; each wave sweeps adjacent rows.  Same arithmetic as revert_masks_kernel (which gathers
// its 4 taps per pixel from global memory: 0.5 TB/s where this form reaches several), same stores.
template <typename T>
__global__ __launch_bounds__(256) void revert_resize_kernel(MaskStage S, int rows_per_block) {
  constexpr int N = Px<T>::N;
  extern __shared__ __attribute__((aligned(16))) float L[];
  const long long mi = blockIdx.y;
  const int ra = blockIdx.x * rows_per_block, rb = min(ra + rows_per_block, S.dh);
  if (ra >= rb) return;
  const T *src = reinterpret_cast<const T *>(S.src) + (size_t)mi * S.sh * S.sw;
  T *dst = reinterpret_cast<T *>(S.dst) + (size_t)mi * S.dh * S.dw;
  const float scale_h = (float)S.vh / (float)S.dh, scale_w = (float)S.vw / (float)S.dw;
  const int ls = (S.vw + 1 + 3) & ~3;   // staged row: vw columns + a repeat of the last, padded to whole 16-byte groups
  const int vy_first = stage_source_rows<T>(S, src, ra, rb, L, ls);
  const int chunks = (S.dw + N - 1) / N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = min(chunks, 64);
  const int R = 64 / lpr, rsub = lane / lpr;
  const int lc = lane - rsub * lpr;
  if (rsub >= R) return;
  const int q = (rb - ra + 3) >> 2;
  const int y0w = ra + wave * q;
  const int y_first = y0w + rsub, y_last = min(y0w + q, rb), ystep = R;
  typedef float taps __attribute__((ext_vector_type(2), aligned(4)));
  for (int chunk = lc; chunk < chunks; chunk += lpr) {
    const int x0 = chunk * N;
    int oa[N];
    float wa[N], wb[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      int ia, ib;
      bilinear_src(scale_w, min(x0 + k, S.dw - 1), S.vw, ia, ib, wa[k], wb[k]);
      oa[k] = ia * 4;
    }
    auto hrow = [&](int iy, int k) -> float {
      const taps a = *reinterpret_cast<const taps *>(reinterpret_cast<const char *>(L + (iy - vy_first) * ls) + oa[k]);
      return fmaf(a.x, wa[k], a.y * wb[k]);
    };
    float tu[N], tl[N];
    int cur0 = -1, cur1 = -1;
    for (int y = y_first; y < y_last; y += ystep) {
      int iy0, iy1; float ly0, ly1;
      bilinear_src(scale_h, y, S.vh, iy0, iy1, ly0, ly1);
      if (iy0 != cur0 || iy1 != cur1) {
        if (iy0 == cur1) {
#pragma unroll
          for (int k = 0; k < N; ++k) tu[k] = tl[k];
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k) tu[k] = hrow(iy0, k);
        }
        if (iy1 == iy0) {
#pragma unroll
          for (int k = 0; k < N; ++k) tl[k] = tu[k];
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k) tl[k] = hrow(iy1, k);
        }
        cur0 = iy0; cur1 = iy1;
      }
      T v[N];
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = Px<T>::from_interp(fmaf(tu[k], ly0, tl[k] * ly1));
      const int yo = S.flip_ud ? S.dh - 1 - y : y;
      T *row = dst + (size_t)yo * S.dw;
      const bool full = x0 + N <= S.dw;
      if (!S.flip_lr) {
        if (full) {
          Px<T>::store(row + x0, v);
        } else {
          for (int k = 0; k < N && x0 + k < S.dw; ++k) row[x0 + k] = v[k];
        }
      } else {
        if (full) {
          T r[N];
#pragma unroll
          for (int k = 0; k < N; ++k) r[k] = v[N - 1 - k];
          Px<T>::store(row + (S.dw - x0 - N), r);
        } else {
          for (int k = 0; k < N && x0 + k < S.dw; ++k) row[S.dw - 1 - (x0 + k)] = v[k];
        }
      }
    }
  }
}

// uint8 masks: revert_resize_kernel rebuilt around what the PMC pass and the ISA of that kernel showed.
//  * With 16 adjacent pixels per lane the horizontal taps of neighbouring lanes lie 16 * (vw / dw) floats apart in the
//    staged rows -- 5.3 at 640 -> 1920, 8 at 640 -> 1280: every 12th (8th) lane on the same LDS bank, ten bank-conflict
//    cycles for every LDS issue cycle.  Here a lane owns four runs of 4 adjacent pixels, 4 * lpr pixels apart: for each
//    of them neighbouring lanes read taps 4 * (vw / dw) floats apart (at most two lanes per bank), and a wave's store
//    instruction still covers one contiguous run of the row (4 bytes per lane instead of 16).
//  * The kernel is then bound by its vector ALU (81 % busy), so the row loop is kept to the arithmetic: the vertical
//    source pair and weight of every row come from a table in LDS (once per strip instead of ten instructions per row
//    and lane); the two horizontally interpolated source rows live in two register sets whose roles (upper / lower)
//    alternate as the sweep moves down, instead of being copied from one set to the other; adjacent pixels are paired
//    for the packed multiply-add; the flip is a template parameter; groups that lie wholly inside the row store without
//    per-run range checks.
template <bool FLIP>
__global__ __launch_bounds__(256) void revert_resize_u8_kernel(MaskStage S, int rows_per_block) {
  constexpr int N = 16, SUBS = 4;
  typedef float f2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float L[];
  __shared__ int2 rowtab[64];   // per strip row: {upper | lower << 16 source row, fp32 bits of the lower row's weight}; rows_per_block <= 64
  const long long mi = blockIdx.y;
  const int ra = blockIdx.x * rows_per_block, rb = min(ra + rows_per_block, S.dh);
  if (ra >= rb) return;
  const uint8_t *src = reinterpret_cast<const uint8_t *>(S.src) + (size_t)mi * S.sh * S.sw;
  uint8_t *dst = reinterpret_cast<uint8_t *>(S.dst) + (size_t)mi * S.dh * S.dw;
  const float scale_h = (float)S.vh / (float)S.dh, scale_w = (float)S.vw / (float)S.dw;
  const int ls = (S.vw + 1 + 3) & ~3;
  if ((int)threadIdx.x < rb - ra) {
    int i0, i1; float l0, l1;
    bilinear_src(scale_h, ra + threadIdx.x, S.vh, i0, i1, l0, l1);
    rowtab[threadIdx.x] = make_int2(i0 | (i1 << 16), __float_as_int(l1));
  }
  const int vy_first = stage_source_rows<uint8_t>(S, src, ra, rb, L, ls);   // ends with the barrier that also covers rowtab
  const int quads = (S.dw + 3) / 4;                      // runs of 4 pixels in a row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = (rb - ra + 3) >> 2;                      // every wave sweeps a quarter of the strip's rows, group by group
  const int y0w = ra + wave * q;
  typedef float taps __attribute__((ext_vector_type(2), aligned(4)));
  typedef uint32_t u32_anywhere __attribute__((aligned(1)));
  // A group: up to SUBS * 64 runs of a row; lane lc owns its runs g0 + j * lpr + lc.  A row's last group may be narrow
  // (1080 pixels = 256 + 14 runs): its lanes then share out R adjacent rows per instruction instead of idling, so a
  // group costs what its width is worth.
  for (int g0 = 0; g0 < quads; g0 += SUBS * 64) {
    const int lpr = min((min(quads - g0, SUBS * 64) + SUBS - 1) / SUBS, 64);   // lanes per row in this group
    const int R = 64 / lpr, rsub = lane / lpr;
    const int lc = lane - rsub * lpr;
    if (rsub >= R) continue;
    const int y_first = y0w + rsub, y_last = min(y0w + q, rb), ystep = R;
    const bool interior = (g0 + SUBS * lpr) * 4 <= S.dw;  // every run of every lane lies wholly inside the row
    int oa[N];
    float wa[N], wb[N];
    int xo[SUBS];                                         // where a run is stored, relative to the row
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int x = (g0 + (k >> 2) * lpr + lc) * 4 + (k & 3);
      int ia, ib;
      bilinear_src(scale_w, min(x, S.dw - 1), S.vw, ia, ib, wa[k], wb[k]);
      oa[k] = ia * 4;
    }
#pragma unroll
    for (int j = 0; j < SUBS; ++j) {
      const int x = (g0 + j * lpr + lc) * 4;
      xo[j] = FLIP ? S.dw - x - 4 : x;
    }
    // returns the OR of the 16 values' bits: 0 iff the lane's stretch of this source row is all zeros (the taps come
    // from bytes: no negative zero, no NaN)
    auto load = [&](f2 (&D)[N / 2], int iy) -> uint32_t {
      const char *r = reinterpret_cast<const char *>(L + (iy - vy_first) * ls);
      uint32_t nz = 0;
#pragma unroll
      for (int h = 0; h < N / 2; ++h) {
        const taps a = *reinterpret_cast<const taps *>(r + oa[2 * h]), c = *reinterpret_cast<const taps *>(r + oa[2 * h + 1]);
        D[h] = f2{fmaf(a.x, wa[2 * h], a.y * wb[2 * h]), fmaf(c.x, wa[2 * h + 1], c.y * wb[2 * h + 1])};
        nz |= __float_as_uint(D[h].x) | __float_as_uint(D[h].y);
      }
      return nz;
    };
    auto emit = [&](const f2 (&U)[N / 2], const f2 (&W)[N / 2], uint32_t nz, float ly1, int y) {
      const float ly0 = 1.0f - ly1;
      const f2 w0v = {ly0, ly0}, w1v = {ly1, ly1};
      const int yo = S.flip_ud ? S.dh - 1 - y : y;
      uint8_t *row = dst + (size_t)yo * S.dw;
      uint32_t word[SUBS];
      // masks are mostly background: where both source rows are zero under every lane of the wave, the output is zero
      // (fma(0, w, 0 * w) = 0) and only the stores remain
      if (interior && __builtin_amdgcn_ballot_w64(nz != 0) == 0) {
#pragma unroll
        for (int j = 0; j < SUBS; ++j) __builtin_nontemporal_store(0u, reinterpret_cast<u32_anywhere *>(row + xo[j]));
        return;
      }
#pragma unroll
      for (int j = 0; j < SUBS; ++j) {
        // aten: fma(upper, w_upper, lower * w_lower), two pixels per packed instruction (each half rounds like the scalar
        // form); `.to(uint8)`: truncation, the values stay inside 0..255
        const f2 v0 = __builtin_elementwise_fma(U[2 * j], w0v, W[2 * j] * w1v), v1 = __builtin_elementwise_fma(U[2 * j + 1], w0v, W[2 * j + 1] * w1v);
        const uint32_t b0 = (uint32_t)(int)v0.x, b1 = (uint32_t)(int)v0.y, b2 = (uint32_t)(int)v1.x, b3 = (uint32_t)(int)v1.y;
        word[j] = FLIP ? (b3 | (b2 << 8)) | ((b1 | (b0 << 8)) << 16) : (b0 | (b1 << 8)) | ((b2 | (b3 << 8)) << 16);
      }
      if (interior) {
#pragma unroll
        for (int j = 0; j < SUBS; ++j) __builtin_nontemporal_store(word[j], reinterpret_cast<u32_anywhere *>(row + xo[j]));
      } else {
#pragma unroll
        for (int j = 0; j < SUBS; ++j) {
          const int x = (g0 + j * lpr + lc) * 4;
          if (x + 4 <= S.dw) {
            __builtin_nontemporal_store(word[j], reinterpret_cast<u32_anywhere *>(row + xo[j]));
          } else {
            for (int k = 0; k < 4 && x + k < S.dw; ++k)   // the row ends inside the run
              row[FLIP ? S.dw - 1 - (x + k) : x + k] = (uint8_t)(word[j] >> (8 * (FLIP ? 3 - k : k)));
          }
        }
      }
    };
    f2 A[N / 2], B[N / 2];
    uint32_t nzA = 0, nzB = 0;
    int curA = -1, curB = -1;
    for (int y = y_first; y < y_last; y += ystep) {
      const int2 re = rowtab[y - ra];
      const int iy0 = re.x & 0xffff, iy1 = re.x >> 16;
      const float ly1 = __int_as_float(re.y);
      // the upper row is wherever it already is, else it replaces the set that does not hold the lower row
      const bool upA = curA == iy0 || (curB != iy0 && curB == iy1) || (curB != iy0 && curA != iy1);
      if (upA) {
        if (curA != iy0) { nzA = load(A, iy0); curA = iy0; }
        if (iy1 == iy0) emit(A, A, nzA, ly1, y);
        else {
          if (curB != iy1) { nzB = load(B, iy1); curB = iy1; }
          emit(A, B, nzA | nzB, ly1, y);
        }
      } else {
        if (curB != iy0) { nzB = load(B, iy0); curB = iy0; }
        if (iy1 == iy0) emit(B, B, nzB, ly1, y);
        else {
          if (curA != iy1) { nzA = load(A, iy1); curA = iy1; }
          emit(B, A, nzA | nzB, ly1, y);
        }
      }
    }
  }
}

// fp32 outputs whose rows are not whole 128-byte lines (dw % 32 != 0: 1916, 1080, 500 ...).  A wave's 16-byte stores are
// issued a quarter wave at a time; 16 lanes x 16 bytes that do not start on a line boundary touch three lines instead
// of two, and every row starts at a different offset inside its line, so no fixed lane -> column assignment is aligned
// (revert_resize_kernel on such rows: 1.5x the time of the aligned width next to it).  Here the arithmetic keeps the
// fixed assignment -- a lane owns up to ITER chunk columns and walks down the rows, the horizontally interpolated
// source-row pair staying in registers -- but the results of `rs` output rows go to LDS, laid out exactly as they lie
// in memory (whole output rows are written, so these rows are ONE contiguous byte range of the mask's plane); then all
// lanes copy that range out in 16-byte chunks cut by ABSOLUTE address, 8 lanes per 128-byte line.  The workgroup is `cl`
// lanes (rounded up to whole waves), one per chunk column of the first ITER-th of the row.  200 masks 640x360 ->
// 1080x1916: 495 -> 334 us (1080x1920 on revert_resize_kernel: 325); -> 1920x1080: 477 -> 366-420; 50 -> 2160x3836:
// 501 -> 377-400.  Used where the stage is bound by its stores (lmi_revert_masks_stage has the rule).
template <int ITER>
__global__ __launch_bounds__(256) void revert_resize_lines_kernel(MaskStage S, int rows_per_block, int src_floats, int rs, int cl) {
  constexpr int N = 4;
  extern __shared__ __attribute__((aligned(16))) float L[];
  const long long mi = blockIdx.y;
  const int ra = blockIdx.x * rows_per_block, rb = min(ra + rows_per_block, S.dh);
  if (ra >= rb) return;
  const float *src = reinterpret_cast<const float *>(S.src) + (size_t)mi * S.sh * S.sw;
  float *dst = reinterpret_cast<float *>(S.dst) + (size_t)mi * S.dh * S.dw;
  const float scale_h = (float)S.vh / (float)S.dh, scale_w = (float)S.vw / (float)S.dw;
  const int ls = (S.vw + 1 + 3) & ~3;
  const int vy_first = stage_source_rows<float>(S, src, ra, rb, L, ls);
  float *Lout = L + src_floats;

  const int tid = threadIdx.x;
  const int c = tid;                               // column lane (the workgroup is cl lanes rounded up to whole waves)
  const bool active = c < cl;
  const int chunks = (S.dw + N - 1) / N;
  const bool vec = (S.dw & 3) == 0;                // chunks are whole and 16-byte aligned inside Lout
  int oa[ITER][N];
  float wa[ITER][N], wb[ITER][N], tu[ITER][N], tl[ITER][N];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int x0 = (c + it * cl) * N;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      int ia, ib;
      bilinear_src(scale_w, min(x0 + k, S.dw - 1), S.vw, ia, ib, wa[it][k], wb[it][k]);
      oa[it][k] = ia * 4;
      tu[it][k] = 0.0f; tl[it][k] = 0.0f;
    }
  }
  typedef float taps __attribute__((ext_vector_type(2), aligned(4)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef f4 f4_a4 __attribute__((aligned(4)));
  int cur0 = -1, cur1 = -1;
  const uintptr_t base = reinterpret_cast<uintptr_t>(dst);
  for (int ys = ra; ys < rb; ys += rs) {
    const int ye = min(ys + rs, rb);
    if (active) {
      for (int y = ys; y < ye; ++y) {
        int iy0, iy1; float ly0, ly1;
        bilinear_src(scale_h, y, S.vh, iy0, iy1, ly0, ly1);
        const bool fresh = iy0 != cur0 || iy1 != cur1, shift = iy0 == cur1, same = iy1 == iy0;
        const float *r0 = L + (iy0 - vy_first) * ls, *r1 = L + (iy1 - vy_first) * ls;
        // memory order of the sub-strip: rows [ys, ye), or their mirror images [dh - ye, dh - ys)
        const int lrow = S.flip_ud ? ye - 1 - y : y - ys;
        float *orow = Lout + (size_t)lrow * S.dw;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int chunk = c + it * cl;
          if (chunk >= chunks) continue;
          const int x0 = chunk * N;
          if (fresh) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
              if (shift) tu[it][k] = tl[it][k];
              else {
                const taps a = *reinterpret_cast<const taps *>(reinterpret_cast<const char *>(r0) + oa[it][k]);
                tu[it][k] = fmaf(a.x, wa[it][k], a.y * wb[it][k]);
              }
              if (same) tl[it][k] = tu[it][k];
              else {
                const taps a = *reinterpret_cast<const taps *>(reinterpret_cast<const char *>(r1) + oa[it][k]);
                tl[it][k] = fmaf(a.x, wa[it][k], a.y * wb[it][k]);
              }
            }
          }
          float v[N];
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = fmaf(tu[it][k], ly0, tl[it][k] * ly1);
          if (vec) {
            if (!S.flip_lr) *reinterpret_cast<f4 *>(orow + x0) = f4{v[0], v[1], v[2], v[3]};
            else *reinterpret_cast<f4 *>(orow + (S.dw - x0 - N)) = f4{v[3], v[2], v[1], v[0]};
          } else {
            for (int k = 0; k < N && x0 + k < S.dw; ++k) orow[S.flip_lr ? S.dw - 1 - (x0 + k) : x0 + k] = v[k];
          }
        }
        cur0 = iy0; cur1 = iy1;
      }
    }
    __syncthreads();
    // copy-out: floats [ga, ga + cnt) of the plane
    const long long ga = (long long)(S.flip_ud ? S.dh - ye : ys) * S.dw;
    const int cnt = (ye - ys) * S.dw;
    const uintptr_t a0 = base + (uintptr_t)ga * 4, a1 = a0 + (uintptr_t)cnt * 4;
    for (uintptr_t addr = (a0 & ~(uintptr_t)127) + (uintptr_t)tid * 16; addr < a1; addr += (uintptr_t)blockDim.x * 16) {
      const int idx = (int)(((long long)addr - (long long)a0) / 4);   // first float of the chunk inside Lout; a0 and addr are both multiples of 4
      if (idx >= 0 && idx + N <= cnt) {
        const f4 o = *reinterpret_cast<const f4_a4 *>(Lout + idx);
        __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(addr));
      } else {
        for (int k = 0; k < N; ++k)
          if (idx + k >= 0 && idx + k < cnt) reinterpret_cast<float *>(addr)[k] = Lout[idx + k];
      }
    }
    __syncthreads();
  }
}

// rescale_masks (lmi_utils/postprocess_utils/mask_utils.py:12-140): paste N soft masks of a
// fixed resolution (e.g. 28x28) into their boxes of an (H, W) image and threshold.  Per output pixel, the
// reference's arithmetic: g = ((p + 0.5) - lo) / (hi - lo) * 2 - 1, aten grid_sample (bilinear, zero padding,
// align_corners=False): i = ((g + 1) * size - 1) / 2, taps floor(i), floor(i)+1 weighted (1-f), f, out-of-range
// taps contribute 0; `>= threshold`.  Pixels whose taps all fall outside the mask are zero without reading it.
// grid: x = 16-pixel column chunks in groups of 64 lanes, y = rows / 4, z = mask.  One 16-byte store per lane.
struct PasteParams {
  const float *masks;   // (n, hm, wm)
  const float *boxes;   // (n, box_stride) x0,y0,x1,y1 first
  uint8_t *out;         // (n, H, W) 0/1
  int hm, wm, box_stride, H, W;
  float threshold;   // < 0: the reference's visualisation mode, (value * 255).to(uint8) instead of `>= threshold`
};

// One output byte from the sampled value: `value >= threshold` -> 0 / 1 (a NaN from a degenerate box compares false,
// like torch), or for threshold < 0 (mask_utils.py:131-133) the value scaled to 0..255 and truncated like `.to(uint8)`.
__device__ __forceinline__ uint8_t paste_value(float acc, float threshold) {
  return threshold >= 0.0f ? (acc >= threshold ? 1 : 0) : (uint8_t)(int)(acc * 255.0f);
}
// ... and of a pixel whose taps all lie outside the mask (sampled value exactly 0)
__device__ __forceinline__ uint8_t paste_outside(float threshold) { return (threshold >= 0.0f && 0.0f >= threshold) ? 1 : 0; }

__global__ __launch_bounds__(256) void paste_masks_kernel(PasteParams P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = blockIdx.y * 4 + wave;
  const int xq = (blockIdx.x * 64 + lane) * 16;
  if (y >= P.H || xq >= P.W) return;
  const int i = blockIdx.z;
  const float *b = P.boxes + (size_t)i * P.box_stride;
  const float bx0 = b[0], by0 = b[1], bx1 = b[2], by1 = b[3];
  const float *m = P.masks + (size_t)i * P.hm * P.wm;
  uint8_t *o = P.out + ((size_t)i * P.H + y) * P.W + xq;

  // Beyond the box's sampling support (fx in [-1, wm) <=> x + 0.5 within half a mask texel of the box; same for y)
  // every tap lies outside the mask and the sum is exactly 0: those chunks -- most of the frame -- are stored without
  // the per-pixel arithmetic.  The margin (a whole texel + 2 px) is far wider than any rounding; a degenerate or NaN
  // box fails the `> 0` tests and takes the general path.
  const float bw = bx1 - bx0, bh = by1 - by0;
  if (bw > 0.0f && bh > 0.0f) {
    // (v_rcp_f32: the margin need not be exact, and two IEEE divisions per lane were a third of this path)
    const float mx = bw * __builtin_amdgcn_rcpf((float)P.wm) + 2.0f, my = bh * __builtin_amdgcn_rcpf((float)P.hm) + 2.0f;
    if ((float)(xq + 16) + mx < bx0 || (float)xq - mx > bx1 || (float)(y + 1) + my < by0 || (float)y - my > by1) {
      uint8_t z[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) z[k] = paste_outside(P.threshold);
      if (xq + 16 <= P.W) {
        Px<uint8_t>::store(o, z);
      } else {
        for (int k = 0; k < 16 && xq + k < P.W; ++k) o[k] = z[k];
      }
      return;
    }
  }

  const float gy = (((float)y + 0.5f) - by0) / (by1 - by0) * 2.0f - 1.0f;
  const float fy = ((gy + 1.0f) * (float)P.hm - 1.0f) / 2.0f;
  const float fy0 = floorf(fy);
  const float wy1 = fy - fy0, wy0 = 1.0f - wy1;
  const int iy0 = (int)fmaxf(fminf(fy0, 1.0e9f), -1.0e9f);
  const bool r0 = iy0 >= 0 && iy0 < P.hm, r1 = iy0 + 1 >= 0 && iy0 + 1 < P.hm;
  uint8_t v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int x = xq + k;
    const float gx = (((float)x + 0.5f) - bx0) / (bx1 - bx0) * 2.0f - 1.0f;
    const float fx = ((gx + 1.0f) * (float)P.wm - 1.0f) / 2.0f;
    const float fx0 = floorf(fx);
    const float wx1 = fx - fx0, wx0 = 1.0f - wx1;
    const int ix0 = (int)fmaxf(fminf(fx0, 1.0e9f), -1.0e9f);
    const bool c0 = ix0 >= 0 && ix0 < P.wm, c1 = ix0 + 1 >= 0 && ix0 + 1 < P.wm;
    // +0, or NaN when a sampling coordinate is not finite (degenerate box): aten multiplies the zero it reads for a tap
    // outside the mask by that tap's weight, so NaN weights reach the sum and `>= threshold` is false for every threshold
    float acc = (wy0 * wx0 + wy1 * wx1) * 0.0f;
    if (r0 && c0) acc += m[iy0 * P.wm + ix0] * (wy0 * wx0);
    if (r0 && c1) acc += m[iy0 * P.wm + ix0 + 1] * (wy0 * wx1);
    if (r1 && c0) acc += m[(iy0 + 1) * P.wm + ix0] * (wy1 * wx0);
    if (r1 && c1) acc += m[(iy0 + 1) * P.wm + ix0 + 1] * (wy1 * wx1);
    v[k] = paste_value(acc, P.threshold);
  }
  if (xq + 16 <= P.W) {
    Px<uint8_t>::store(o, v);
  } else {
    for (int k = 0; k < 16 && xq + k < P.W; ++k) o[k] = v[k];
  }
}

// The same paste for images whose rows are NOT whole 16-byte chunks (W % 16 != 0, W >= 16): row-relative chunks then
// start at any address and every row ends in a chunk written byte by byte (1080 x 1916: 442 us against 169 us at
// 1920).  Here the whole output (n, H, W) is cut into 16-byte chunks by ABSOLUTE address instead: one lane per chunk,
// every store aligned; a chunk may run over the end of a row into the next one (or into the next mask's first row),
// so a lane carries the row constants of both.  The per-pixel arithmetic is paste_masks_kernel's, operation for
// operation.
struct PasteRow {
  const float *m;
  float bx0, bx1, by0, by1, wy0, wy1;
  int iy0, y;
  bool r0, r1;
};

__device__ __forceinline__ void paste_row(const PasteParams &P, long long i, int y, PasteRow &R) {
  const float *b = P.boxes + (size_t)i * P.box_stride;
  R.bx0 = b[0]; R.by0 = b[1]; R.bx1 = b[2]; R.by1 = b[3];
  R.m = P.masks + (size_t)i * P.hm * P.wm;
  R.y = y;
  const float gy = (((float)y + 0.5f) - R.by0) / (R.by1 - R.by0) * 2.0f - 1.0f;
  const float fy = ((gy + 1.0f) * (float)P.hm - 1.0f) / 2.0f;
  const float fy0 = floorf(fy);
  R.wy1 = fy - fy0; R.wy0 = 1.0f - R.wy1;
  R.iy0 = (int)fmaxf(fminf(fy0, 1.0e9f), -1.0e9f);
  R.r0 = R.iy0 >= 0 && R.iy0 < P.hm;
  R.r1 = R.iy0 + 1 >= 0 && R.iy0 + 1 < P.hm;
}

// every tap of the pixels [xa, xb) of the row lies outside the mask (same margins as paste_masks_kernel)
__device__ __forceinline__ bool paste_row_outside(const PasteParams &P, const PasteRow &R, int xa, int xb) {
  const float bw = R.bx1 - R.bx0, bh = R.by1 - R.by0;
  if (!(bw > 0.0f && bh > 0.0f)) return false;
  const float mx = bw / (float)P.wm + 2.0f, my = bh / (float)P.hm + 2.0f;
  return (float)xb + mx < R.bx0 || (float)xa - mx > R.bx1 || (float)(R.y + 1) + my < R.by0 || (float)R.y - my > R.by1;
}

__device__ __forceinline__ uint8_t paste_pixel(const PasteParams &P, const PasteRow &R, int x) {
  const float gx = (((float)x + 0.5f) - R.bx0) / (R.bx1 - R.bx0) * 2.0f - 1.0f;
  const float fx = ((gx + 1.0f) * (float)P.wm - 1.0f) / 2.0f;
  const float fx0 = floorf(fx);
  const float wx1 = fx - fx0, wx0 = 1.0f - wx1;
  const int ix0 = (int)fmaxf(fminf(fx0, 1.0e9f), -1.0e9f);
  const bool c0 = ix0 >= 0 && ix0 < P.wm, c1 = ix0 + 1 >= 0 && ix0 + 1 < P.wm;
  float acc = (R.wy0 * wx0 + R.wy1 * wx1) * 0.0f;   // +0, or NaN for a coordinate that is not finite (see paste_masks_kernel)
  if (R.r0 && c0) acc += R.m[R.iy0 * P.wm + ix0] * (R.wy0 * wx0);
  if (R.r0 && c1) acc += R.m[R.iy0 * P.wm + ix0 + 1] * (R.wy0 * wx1);
  if (R.r1 && c0) acc += R.m[(R.iy0 + 1) * P.wm + ix0] * (R.wy1 * wx0);
  if (R.r1 && c1) acc += R.m[(R.iy0 + 1) * P.wm + ix0 + 1] * (R.wy1 * wx1);
  return paste_value(acc, P.threshold);
}

// The paste in two passes (r03; masks up to 60 x 60): paste_masks_kernel spends ~80 instructions on
// every pixel inside a box's sampling support (two IEEE divisions, four predicated global loads) and ~40 per lane on the
// chunks outside it, which are most of the frame.
//  paste_fill_kernel  the constant `0 >= threshold` for the whole output at the memset rate (one workgroup per CU,
//                     grid-stride, non-temporal 16-byte stores: the pattern of mask_zero_fill_kernel);
//  paste_box_kernel   only the chunks that meet a box's support: grid (strips of 64 rows, masks), workgroups off the
//                     support leave at once; the mask sits in LDS inside a border of two zero texels (a tap outside
//                     the mask then reads a zero instead of being predicated away: `acc += 0 * w` changes nothing);
//                     every lane owns one 16-pixel chunk column, does its pixels' x arithmetic once and walks down the
//                     strip's rows, whose y arithmetic comes from a table; per pixel and row what remains is four taps
//                     (two 8-byte LDS reads), four weight products and the sum, in paste_masks_kernel's order.
// A box that is degenerate or not finite (NaN compares false throughout) keeps paste_pixel over its whole frame.
__global__ __launch_bounds__(256) void paste_fill_kernel(uint8_t *out, long long nbytes, uint32_t word) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 v = {word, word, word, word};
  u4 *o = reinterpret_cast<u4 *>(out);
  const long long n = nbytes >> 4, stride = (long long)gridDim.x * 256;
  for (long long base = (long long)blockIdx.x * 256; base < n; base += stride) {   // the waves kept in step as in mask_zero_fill_kernel
    const long long g = base + threadIdx.x;
    if (g < n) __builtin_nontemporal_store(v, o + g);
    __builtin_amdgcn_s_barrier();
  }
  if (blockIdx.x == gridDim.x - 1 && (long long)threadIdx.x < (nbytes & 15)) out[(n << 4) + threadIdx.x] = (uint8_t)word;   // rows that are not whole chunks may leave a tail
}

constexpr int kPasteStrip = 32;    // rows per workgroup
constexpr int kPasteMaskMax = 60;  // largest mask side the LDS buffer holds (64 x 64 floats with the border)

__global__ __launch_bounds__(256) void paste_box_kernel(PasteParams P) {
  __shared__ __attribute__((aligned(16))) float M[(kPasteMaskMax + 4) * (kPasteMaskMax + 4)];
  __shared__ float4 rowtab[kPasteStrip];   // per strip row: {padded upper row index (as int bits), wy0, wy1, -}
  const int i = blockIdx.y;
  const float *b = P.boxes + (size_t)i * P.box_stride;
  const float bx0 = b[0], by0 = b[1], bx1 = b[2], by1 = b[3];
  const float bw = bx1 - bx0, bh = by1 - by0;
  const int ra = blockIdx.x * kPasteStrip, rb = min(ra + kPasteStrip, P.H);
  uint8_t *out = P.out + (size_t)i * P.H * P.W;
  const bool regular = bw > 0.0f && bh > 0.0f && bw < 1.0e8f && bh < 1.0e8f && fabsf(bx0) < 1.0e8f && fabsf(by0) < 1.0e8f;
  if (!regular) {   // the general path, pixel by pixel over the strip (uniform over the workgroup)
    for (int y = ra + (threadIdx.x >> 6); y < rb; y += 4) {
      PasteRow R;
      paste_row(P, i, y, R);
      for (int xq = (threadIdx.x & 63) * 16; xq < P.W; xq += 64 * 16) {
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = paste_pixel(P, R, xq + k);
        if (xq + 16 <= P.W) Px<uint8_t>::store(out + (size_t)y * P.W + xq, v);
        else for (int k = 0; k < 16 && xq + k < P.W; ++k) out[(size_t)y * P.W + xq + k] = v[k];
      }
    }
    return;
  }
  // the support, with paste_masks_kernel's margins: rows and chunk columns outside it hold the constant already
  const float mx = bw / (float)P.wm + 2.0f, my = bh / (float)P.hm + 2.0f;
  const int ya = max(ra, (int)floorf(by0 - my) - 1), yb = min(rb, (int)ceilf(by1 + my) + 2);
  if (ya >= yb) return;
  const int ca = max(0, ((int)floorf(bx0 - mx) - 1) >> 4), cb = min((P.W + 15) >> 4, (((int)ceilf(bx1 + mx) + 2) + 15) >> 4);
  if (ca >= cb) return;
  const int pw = P.wm + 4;
  const float *m = P.masks + (size_t)i * P.hm * P.wm;
  for (int t = threadIdx.x; t < (P.hm + 4) * pw; t += 256) {
    const int r = t / pw - 2, c = t - (r + 2) * pw - 2;
    M[t] = (r >= 0 && r < P.hm && c >= 0 && c < P.wm) ? m[r * P.wm + c] : 0.0f;
  }
  if ((int)threadIdx.x < yb - ya) {
    const int y = ya + threadIdx.x;
    const float gy = (((float)y + 0.5f) - by0) / (by1 - by0) * 2.0f - 1.0f;
    const float fy = ((gy + 1.0f) * (float)P.hm - 1.0f) / 2.0f;
    const float fy0 = floorf(fy);
    const float wy1 = fy - fy0, wy0 = 1.0f - wy1;
    const int iy0 = (int)fmaxf(fminf(fy0, 1.0e9f), -1.0e9f);
    rowtab[threadIdx.x] = make_float4(__int_as_float(min(max(iy0 + 2, 0), P.hm + 2) * pw * 4), wy0, wy1, 0.0f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = min(cb - ca, 64);
  const int R = 64 / lpr, rsub = lane / lpr;
  const int lc = lane - rsub * lpr;
  if (rsub >= R) return;
  const int q = (yb - ya + 3) >> 2;
  const int y0w = ya + wave * q;
  typedef float taps __attribute__((ext_vector_type(2), aligned(4)));
  for (int chunk = ca + lc; chunk < cb; chunk += lpr) {
    const int xq = chunk * 16;
    int ox[16];            // byte offset of the left tap's column inside a padded row
    float wx0[16], wx1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int x = xq + k;
      const float gx = (((float)x + 0.5f) - bx0) / (bx1 - bx0) * 2.0f - 1.0f;
      const float fx = ((gx + 1.0f) * (float)P.wm - 1.0f) / 2.0f;
      const float fx0 = floorf(fx);
      wx1[k] = fx - fx0; wx0[k] = 1.0f - wx1[k];
      const int ix0 = (int)fmaxf(fminf(fx0, 1.0e9f), -1.0e9f);
      ox[k] = min(max(ix0 + 2, 0), P.wm + 2) * 4;
    }
    for (int y = y0w + rsub; y < min(y0w + q, yb); y += R) {
      const float4 rt = rowtab[y - ya];
      const char *r0 = reinterpret_cast<const char *>(M) + __float_as_int(rt.x), *r1 = r0 + pw * 4;
      const float wy0 = rt.y, wy1 = rt.z;
      uint8_t v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const taps u = *reinterpret_cast<const taps *>(r0 + ox[k]), l = *reinterpret_cast<const taps *>(r1 + ox[k]);
        float acc = 0.0f;
        acc += u.x * (wy0 * wx0[k]);
        acc += u.y * (wy0 * wx1[k]);
        acc += l.x * (wy1 * wx0[k]);
        acc += l.y * (wy1 * wx1[k]);
        v[k] = paste_value(acc, P.threshold);
      }
      // 16-byte stores at whatever address the row gives (rows that are not whole chunks); a chunk the row ends in goes out byte by byte
      if (xq + 16 <= P.W) Px<uint8_t>::store(out + (size_t)y * P.W + xq, v);
      else for (int k = 0; k < 16 && xq + k < P.W; ++k) out[(size_t)y * P.W + xq + k] = v[k];
    }
  }
}

__global__ __launch_bounds__(256) void paste_masks_flat_kernel(PasteParams P, long long n) {
  const long long HW = (long long)P.H * P.W, total = n * HW;
  const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
  if (p0 >= total) return;
  const long long i = p0 / HW;
  const int rem = (int)(p0 - i * HW);
  const int y = rem / P.W, x = rem - y * P.W;
  const int lenA = min(16, P.W - x);   // pixels of this chunk in row (i, y); the rest open the next row (W >= 16: one row at most)
  PasteRow A, B;
  paste_row(P, i, y, A);
  long long iB = i;
  int yB = y + 1;
  if (yB == P.H) { yB = 0; iB = i + 1; }
  const bool hasB = lenA < 16 && iB < n;
  B = A;
  if (hasB) paste_row(P, iB, yB, B);
  uint8_t *o = P.out + p0;
  uint8_t v[16];
  if (paste_row_outside(P, A, x, x + lenA) && (!hasB || paste_row_outside(P, B, 0, 16 - lenA))) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = paste_outside(P.threshold);
  } else if (lenA == 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = paste_pixel(P, A, x + k);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = k < lenA ? paste_pixel(P, A, x + k) : paste_pixel(P, B, k - lenA);
  }
  if (p0 + 16 <= total) {
    Px<uint8_t>::store(o, v);
  } else {
    for (int k = 0; k < 16 && p0 + k < total; ++k) o[k] = v[k];
  }
}

}  // namespace

extern "C" {

int lmi_revert_points_f32(const float *pts, int64_t n_pairs, const lmi_point_op *ops, int n_ops, float *out, void *stream) {
  LMI_REQUIRE(n_pairs >= 0 && n_ops >= 0, LMI_ERR_BAD_ARG, "lmi_revert_points_f32: negative count");
  LMI_REQUIRE(n_ops <= LMI_MAX_POINT_OPS, LMI_ERR_LIMIT, "lmi_revert_points_f32: %d operations > %d", n_ops, LMI_MAX_POINT_OPS);
  if (n_pairs == 0) return LMI_OK;
  LMI_REQUIRE(pts && out && (ops || n_ops == 0), LMI_ERR_BAD_ARG, "lmi_revert_points_f32: NULL pointer");
  LMI_REQUIRE(n_pairs <= (int64_t)0x7fffffff * 128, LMI_ERR_LIMIT, "lmi_revert_points_f32: too many points");
  PointOps P;
  P.n = n_ops;
  for (int k = 0; k < n_ops; ++k) {
    LMI_REQUIRE(ops[k].kind >= LMI_POINT_DIV && ops[k].kind <= LMI_POINT_FLIP_XY, LMI_ERR_BAD_ARG,
                "lmi_revert_points_f32: unsupported operation kind %d", ops[k].kind);
    P.kind[k] = ops[k].kind; P.x[k] = ops[k].x; P.y[k] = ops[k].y;
  }
  const unsigned grid = (unsigned)((n_pairs + 255) / 256);
  hipLaunchKernelGGL(revert_points_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pts, out, (long long)n_pairs, P);
  return lmi_check_launch("revert_points_kernel");
}

int lmi_revert_masks_stage(const void *src, int dtype, int64_t m, int src_h, int src_w, int fit_h, int fit_w, int resize_h,
                           int resize_w, int flip_lr, int flip_ud, void *dst, void *stream) {
  LMI_REQUIRE(dtype == LMI_F32 || dtype == LMI_U8, LMI_ERR_BAD_ARG, "lmi_revert_masks_stage: dtype must be LMI_F32 or LMI_U8");
  LMI_REQUIRE(m >= 0 && m <= 65535, LMI_ERR_LIMIT, "lmi_revert_masks_stage: %lld masks not in 0..65535", (long long)m);
  LMI_REQUIRE(src_h > 0 && src_w > 0, LMI_ERR_BAD_ARG, "lmi_revert_masks_stage: bad source size %dx%d", src_h, src_w);
  MaskStage S;
  S.src = src; S.dst = dst; S.m = m; S.sh = src_h; S.sw = src_w;
  // fit_im_to_size: centred; when growing pad_before = (want - have) // 2, when shrinking crop_before = (have - want) // 2
  S.vh = fit_h > 0 ? fit_h : src_h;
  S.vw = fit_w > 0 ? fit_w : src_w;
  S.oy = S.vh >= src_h ? -((S.vh - src_h) / 2) : (src_h - S.vh) / 2;
  S.ox = S.vw >= src_w ? -((S.vw - src_w) / 2) : (src_w - S.vw) / 2;
  S.resize = (resize_h > 0 || resize_w > 0) ? 1 : 0;
  LMI_REQUIRE(!S.resize || (resize_h > 0 && resize_w > 0), LMI_ERR_BAD_ARG, "lmi_revert_masks_stage: resize needs both sides");
  S.dh = S.resize ? resize_h : S.vh;
  S.dw = S.resize ? resize_w : S.vw;
  S.flip_lr = flip_lr ? 1 : 0; S.flip_ud = flip_ud ? 1 : 0;
  if (m == 0) return LMI_OK;
  LMI_REQUIRE(src && dst, LMI_ERR_BAD_ARG, "lmi_revert_masks_stage: NULL pointer");
  const int n = dtype == LMI_F32 ? 4 : 16;
  dim3 grid(((S.dw + n - 1) / n + 63) / 64, (S.dh + 3) / 4, (unsigned)m);
  LMI_REQUIRE(grid.y <= 65535, LMI_ERR_LIMIT, "lmi_revert_masks_stage: output height %d too large", S.dh);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // fp32 rows that are not whole 128-byte lines, enlarged >= 6x in area (where the stage is bound by its stores; below
  // that, and on narrow outputs, the extra pass through LDS costs what the aligned stores save: 720p 514 vs 496 us per
  // 400 masks, measured): revert_resize_lines_kernel
  if (S.resize && S.dh + S.dw > 128 && dtype == LMI_F32 && (S.dw & 31) != 0 && S.dw >= 512 &&
      (long long)S.dh * S.dw >= 6LL * S.vh * S.vw) {
    // a lane owns `it` chunk columns: the fewest that make a workgroup of <= 256 lanes cover a row
    const int chunks = (S.dw + 3) / 4;
    const int src_kb = 16, out_kb = 16;
    int it = 0, cl = 0;
    for (int i = 1; i <= 4 && it == 0; ++i) {
      const int c = (chunks + i - 1) / i;
      if (c <= 256) { it = i; cl = c; }
    }
    if (it > 0) {
      const int lanes = (cl + 63) & ~63;
      int rpb = 64, lds_rows = 0;
      for (;; --rpb) {
        lds_rows = (int)((double)rpb * S.vh / S.dh) + 3;
        if (lds_rows > S.vh) lds_rows = S.vh;
        if ((size_t)lds_rows * ((S.vw + 4) & ~3) * sizeof(float) <= (size_t)src_kb * 1024 || rpb == 1) break;
      }
      const size_t src_floats = (size_t)lds_rows * ((S.vw + 4) & ~3);
      int rs = (out_kb * 1024) / (S.dw * 4);
      if (rs > rpb) rs = rpb;
      if (rs >= 1 && src_floats * 4 <= (size_t)src_kb * 1024) {
        const size_t lds = src_floats * 4 + (size_t)rs * S.dw * 4 + 16;
        dim3 g2((S.dh + rpb - 1) / rpb, (unsigned)m);
        switch (it) {
          case 1: hipLaunchKernelGGL(revert_resize_lines_kernel<1>, g2, dim3(lanes), lds, s, S, rpb, (int)src_floats, rs, cl); break;
          case 2: hipLaunchKernelGGL(revert_resize_lines_kernel<2>, g2, dim3(lanes), lds, s, S, rpb, (int)src_floats, rs, cl); break;
          case 3: hipLaunchKernelGGL(revert_resize_lines_kernel<3>, g2, dim3(lanes), lds, s, S, rpb, (int)src_floats, rs, cl); break;
          default: hipLaunchKernelGGL(revert_resize_lines_kernel<4>, g2, dim3(lanes), lds, s, S, rpb, (int)src_floats, rs, cl); break;
        }
        return lmi_check_launch("revert_resize_lines_kernel");
      }
    }
  }
  if (S.resize && S.dh + S.dw > 128) {
    // rows per strip: the largest power of two <= 64 whose staged source rows fit 48 KiB of LDS
    int rpb = 64, lds_rows = 0;
    for (;; rpb >>= 1) {
      lds_rows = (int)((double)rpb * S.vh / S.dh) + 3;
      if (lds_rows > S.vh) lds_rows = S.vh;
      if ((size_t)lds_rows * ((S.vw + 4) & ~3) * sizeof(float) <= 48 * 1024 || rpb == 1) break;
    }
    const size_t lds = (size_t)lds_rows * ((S.vw + 4) & ~3) * sizeof(float);
    if (lds <= 64 * 1024) {
      dim3 g2((S.dh + rpb - 1) / rpb, (unsigned)m);
      if (dtype == LMI_F32) hipLaunchKernelGGL(revert_resize_kernel<float>, g2, dim3(256), lds, s, S, rpb);
      // uint8: enlargements of 16x in area and more are bound by their stores, and on rows that are not whole 64-byte
      // blocks the 16-byte stores of the generic kernel hold up better than 4-byte ones (4K - 4 columns: 301 vs 358 us)
      else if ((S.dw & 63) != 0 && (long long)S.dh * S.dw >= 16LL * S.vh * S.vw) hipLaunchKernelGGL(revert_resize_kernel<uint8_t>, g2, dim3(256), lds, s, S, rpb);
      else if (S.flip_lr) hipLaunchKernelGGL(revert_resize_u8_kernel<true>, g2, dim3(256), lds, s, S, rpb);
      else hipLaunchKernelGGL(revert_resize_u8_kernel<false>, g2, dim3(256), lds, s, S, rpb);
      return lmi_check_launch("revert_resize_kernel");
    }
  }
  if (dtype == LMI_F32) hipLaunchKernelGGL(revert_masks_kernel<float>, grid, dim3(256), 0, s, S);
  else hipLaunchKernelGGL(revert_masks_kernel<uint8_t>, grid, dim3(256), 0, s, S);
  return lmi_check_launch("revert_masks_kernel");
}

// CU count of the current device, asked once per device (the paste's fill launches one workgroup per CU)
static int paste_cu_count() {
  static std::mutex mu;
  static int cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if (cache[dev] == 0) {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    cache[dev] = ncu;
  }
  return cache[dev];
}

int lmi_paste_masks(const float *masks, int64_t n, int mask_h, int mask_w, const float *boxes, int box_stride, int img_h,
                    int img_w, float threshold, uint8_t *out, void *stream) {
  LMI_REQUIRE(n >= 0 && n <= 65535, LMI_ERR_LIMIT, "lmi_paste_masks: %lld masks not in 0..65535", (long long)n);
  LMI_REQUIRE(mask_h > 0 && mask_w > 0 && img_h > 0 && img_w > 0 && box_stride >= 4, LMI_ERR_BAD_ARG,
              "lmi_paste_masks: bad shape mask=%dx%d image=%dx%d box_stride=%d", mask_h, mask_w, img_h, img_w, box_stride);
  if (n == 0) return LMI_OK;
  LMI_REQUIRE(masks && boxes && out, LMI_ERR_BAD_ARG, "lmi_paste_masks: NULL pointer");
  PasteParams P;
  P.masks = masks; P.boxes = boxes; P.out = out; P.hm = mask_h; P.wm = mask_w; P.box_stride = box_stride;
  P.H = img_h; P.W = img_w; P.threshold = threshold;
  dim3 grid(((img_w + 15) / 16 + 63) / 64, (img_h + 3) / 4, (unsigned)n);
  LMI_REQUIRE(grid.y <= 65535, LMI_ERR_LIMIT, "lmi_paste_masks: image height %d too large", img_h);
  if (mask_h <= kPasteMaskMax && mask_w <= kPasteMaskMax && lmi_is_aligned(out, 16) && (img_h + kPasteStrip - 1) / kPasteStrip <= 65535) {
    // the constant everywhere at the memset rate, then the chunks under each box's support
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int cus = paste_cu_count();
    const uint32_t word = (threshold >= 0.0f && 0.0f >= threshold) ? 0x01010101u : 0u;
    hipLaunchKernelGGL(paste_fill_kernel, dim3((unsigned)cus), dim3(256), 0, s, out, (long long)n * img_h * img_w, word);
    hipLaunchKernelGGL(paste_box_kernel, dim3((img_h + kPasteStrip - 1) / kPasteStrip, (unsigned)n), dim3(256), 0, s, P);
    return lmi_check_launch("paste_box_kernel");
  }
  if ((img_w & 15) != 0 && img_w >= 16 && (long long)img_h * img_w < (1LL << 31) && lmi_is_aligned(out, 16)) {   // rows that are not whole chunks: chunks by absolute address
    const long long chunks = ((long long)n * img_h * img_w + 15) / 16;
    const long long blocks = (chunks + 255) / 256;
    LMI_REQUIRE(blocks <= 0x7fffffffLL, LMI_ERR_LIMIT, "lmi_paste_masks: output too large");
    hipLaunchKernelGGL(paste_masks_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, (long long)n);
    return lmi_check_launch("paste_masks_flat_kernel");
  }
  hipLaunchKernelGGL(paste_masks_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P);
  return lmi_check_launch("paste_masks_kernel");
}

}  // extern "C"
