// Segmentation mask assembly for gfx950.
//
// Replaces `ops.process_mask_native(proto[i], pred[:, 6:], pred[:, :4], orig_img.shape[:2])` and
// `masks[M]` (object_detectors/yolov8_lmi/model.py:235-236; in-tree YOLOv5 analogue
// object_detectors/yolov5_lmi/trt/yolov5_trt.py:373-415).  Semantics: oracle/masks.py.
//
// Kernels
//  mask_logits_rows_kernel  logits[slot][y][x] = sum_k coef[slot][k] * proto[b][k][y][x], only for the rows
//                        of the letterbox crop window that lie under the survivor's box, as a k-ordered
//                        fmaf chain on the vector ALU (v_pk_fma_f32); proto is read once.
//  mask_gemm_kernel      the same product as dense 32-survivor tiles on the exact-fp32 MFMA
//                        (v_mfma_f32_32x32x2_f32); kept for odd prototype widths.
//  The upsample is HBM-write-bound (m*H0*W0 bytes of uint8 per image, ~97 % of them zeros outside the boxes) and
//  runs as two passes when output rows are whole 16-byte chunks:
//  mask_slot_table_kernel  per slot, the integer pixel ranges of its box.
//  mask_zero_fill_kernel   zeros for all live slots from ONE workgroup per CU: the write pattern that reaches the
//                        memset rate (6.4 TB/s; profiles/r01/write_ceiling.md).
//  mask_box_kernel       the chunks that meet a box: bilinear (align_corners=False) -> box crop -> `> 0`
//                        from logits rows staged in LDS, 16 bytes per lane per store.
//  mask_upsample_kernel  the same result in one pass (per survivor, per strip of output rows, zeros included),
//                        for row lengths that are not whole chunks; `LMI_MASK_UPSAMPLE=single` selects it for
//                        every shape (measurement).
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct MaskGeom {
  int mh, mw;                    // proto size
  int top, bottom, left, right;  // scale_masks crop window in proto space
  int h0, w0;                    // original frame size
};

struct GemmParams {
  const float *proto;  // (batch, 32, mh, mw)
  const float *det;    // (batch, max_det, det_stride)
  const int *slot_map;
  const int *slot_base;
  float *logits;  // (capacity, wh*mw)
  int batch, max_det, det_stride;
  long long capacity;
  MaskGeom g;
  int tiles_per_block;  // pixel tiles (32*V px) per wave
  int4 *table;          // per-slot pixel ranges of the box {xa, xb, ya, yb} for mask_box_kernel
};

// aten upsample_bilinear2d, align_corners=False: source index / weights of output index d
__device__ __forceinline__ void bilinear_src(float scale, int d, int in_n, int &i0, int &i1, float &l0, float &l1) {
  float s = scale * ((float)d + 0.5f) - 0.5f;
  s = fmaxf(s, 0.0f);
  i0 = (int)s;
  i1 = i0 + ((i0 < in_n - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

// Integer pixel ranges of a box: x >= x1 && x < x2  <=>  ceil(x1) <= x < ceil(x2) for integer x (same for y),
// clamped to the frame; all zero when the box holds no pixel (NaN included).
__device__ __forceinline__ int4 box_ranges(float bx1, float by1, float bx2, float by2, int H0, int W0) {
  if (!(bx1 < bx2 && by1 < by2)) return make_int4(0, 0, 0, 0);
  const int xa = (int)ceilf(fminf(fmaxf(bx1, 0.0f), (float)W0)), xb = (int)ceilf(fminf(fmaxf(bx2, 0.0f), (float)W0));
  const int ya = (int)ceilf(fminf(fmaxf(by1, 0.0f), (float)H0)), yb = (int)ceilf(fminf(fmaxf(by2, 0.0f), (float)H0));
  if (xa >= xb || ya >= yb) return make_int4(0, 0, 0, 0);
  return make_int4(xa, xb, ya, yb);
}

// grid: x = pixel-tile groups, y = survivor tile (32 slots), z = image
template <int V>
__global__ __launch_bounds__(256) void mask_gemm_kernel(GemmParams G) {
  const int b = blockIdx.z;
  const int base = G.slot_base[b];
  const int m_b = G.slot_base[b + 1] - base;
  const int row0 = blockIdx.y * 32;
  if (row0 >= m_b) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wh = G.g.bottom - G.g.top;

  // Logits rows each survivor's upsample will read: the bilinear taps of the output rows inside its box
  // (same arithmetic as mask_upsample_kernel).  Only those are computed and written.
  __shared__ int rlo[32], rhi[32], runion[2];
  if (threadIdx.x < 32) {
    int lo = 1, hi = 0;
    const int i = row0 + (int)threadIdx.x;
    if (i < m_b) {
      const float *d = G.det + (size_t)G.slot_map[base + i] * G.det_stride;
      const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3];
      const int ya = max(0, (int)ceilf(by1)), yb = min(G.g.h0, (int)ceilf(by2));
      if (ya < yb && bx1 < bx2) {
        const float scale_h = (float)wh / (float)G.g.h0;
        int i0, i1; float l0, l1;
        bilinear_src(scale_h, ya, wh, i0, i1, l0, l1);
        lo = i0;
        bilinear_src(scale_h, yb - 1, wh, i0, i1, l0, l1);
        hi = i1;
      }
    }
    rlo[threadIdx.x] = lo;
    rhi[threadIdx.x] = hi;
    // union over the tile's survivors (wave 0, lanes 0..31 active)
    int ulo = lo <= hi ? lo : 0x7fffffff, uhi = lo <= hi ? hi : -1;
    for (int o = 16; o > 0; o >>= 1) {
      ulo = min(ulo, __shfl_xor(ulo, o));
      uhi = max(uhi, __shfl_xor(uhi, o));
    }
    if (threadIdx.x == 0) { runion[0] = ulo; runion[1] = uhi; }
  }
  __syncthreads();
  const int ulo = runion[0], uhi = runion[1];
  if (ulo > uhi) return;
  const int npx = wh * G.g.mw;
  const int plane = G.g.mh * G.g.mw;
  const float *pb = G.proto + (size_t)b * 32 * plane + (size_t)G.g.top * G.g.mw;

  // A fragments: coef[row0 + (lane&31)][2s + (lane>>5)], s = 0..15
  float a[16];
  {
    const int i = row0 + (lane & 31);
    const float *c = nullptr;
    if (i < m_b) {
      const int e = G.slot_map[base + i];
      c = G.det + (size_t)e * G.det_stride + 6;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = c ? c[2 * s + (lane >> 5)] : 0.0f;
  }

  // Tile = 32 survivors x 32*V pixels.  With V = 4 a lane owns 4 ADJACENT pixels (one 16-byte load per k row, one
  // 16-byte store per survivor row) and the 4 MFMA chains e = 0..3 use pixel 4n+e as column n: 16 KiB of loads
  // in flight per wave instead of 4 KiB.
  constexpr int TP = 32 * V;
  const int tile0 = (blockIdx.x * 4 + wave) * G.tiles_per_block;
  for (int t = 0; t < G.tiles_per_block; ++t) {
    const int p0 = (tile0 + t) * TP;
    if (p0 >= npx) break;
    if (p0 / G.g.mw > uhi || min(p0 + TP - 1, npx - 1) / G.g.mw < ulo) continue;  // no survivor of this tile reads these rows
    const int p = p0 + V * (lane & 31);
    const bool pv = p < npx;          // npx % V == 0: a lane's V pixels are valid together and share a row
    const int prow = p / G.g.mw;
    floatx16 acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][r] = 0.0f;
    float bf[16][V];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float *q = pb + (size_t)(2 * s + (lane >> 5)) * plane + p;
      if constexpr (V == 4) {
        const float4 v = pv ? *reinterpret_cast<const float4 *>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        bf[s][0] = v.x; bf[s][1] = v.y; bf[s][2] = v.z; bf[s][3] = v.w;
      } else {
        bf[s][0] = pv ? *q : 0.0f;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[s][e], acc[e], 0, 0, 0);
    // D: col n = lane&31 (pixel V*n+e of the tile) ; row (survivor) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (pv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int li = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int i = row0 + li;
        const long long slot = (long long)base + i;
        if (i < m_b && slot < G.capacity && prow >= rlo[li] && prow <= rhi[li]) {
          float *o = G.logits + (size_t)slot * npx + p;
          if constexpr (V == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
          else *o = acc[0][r];
        }
      }
    }
  }
}

// Row-sparse variant of the GEMM on the vector ALU, used when prototype rows are whole 8-byte pairs.
// A survivor's upsample reads only the logits rows under its box, typically a tenth of the window, so the dense
// 32-survivor MFMA tile above spends most of its issue slots on rows nobody reads (measured 82 us per seg640
// launch, MFMA-issue bound).  Here a wave keeps 128 consecutive window pixels x 32 channels in registers (2 px per
// lane) and walks the image's survivors, doing the k-ordered fmaf chain only for those whose row range meets the
// wave's pixels; coefficients and row ranges sit in LDS and are read as broadcasts.
// grid: x = groups of 4 pixel tiles (one per wave), y = image
constexpr int kSurvBatch = 128;  // survivors staged per pass: 128 * (32 coef + 2 range) * 4 B = 17 KiB of LDS

__global__ __launch_bounds__(256) void mask_logits_rows_kernel(GemmParams G) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float coef[kSurvBatch][32];
  __shared__ int rlo[kSurvBatch], rhi[kSurvBatch];

  const int b = blockIdx.y;
  const int base = G.slot_base[b];
  const int m_b = G.slot_base[b + 1] - base;
  if (m_b <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wh = G.g.bottom - G.g.top, mw = G.g.mw;
  const int npx = wh * mw;
  const int plane = G.g.mh * mw;
  const float *pb = G.proto + (size_t)b * 32 * plane + (size_t)G.g.top * mw;
  const float scale_h = (float)wh / (float)G.g.h0;

  const int p0 = (blockIdx.x * 4 + wave) * 128;
  const int p = p0 + 2 * lane;
  const bool pv = p < npx;                 // mw even: a lane's two pixels are valid together and share a row
  const int prow = pv ? p / mw : -1;
  const int ra = p0 < npx ? p0 / mw : 0x7fffffff;   // rows this wave touches
  const int rb = p0 < npx ? min(p0 + 127, npx - 1) / mw : -1;

  auto row_range = [&](const float *d, int &lo, int &hi) {
    const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3];
    lo = 1; hi = 0;
    const int ya = max(0, (int)ceilf(by1)), yb = min(G.g.h0, (int)ceilf(by2));
    if (ya < yb && bx1 < bx2) {  // same arithmetic as mask_upsample_kernel
      int i0, i1; float l0, l1;
      bilinear_src(scale_h, ya, wh, i0, i1, l0, l1);
      lo = i0;
      bilinear_src(scale_h, yb - 1, wh, i0, i1, l0, l1);
      hi = i1;
    }
  };

  // Issued before anything depends on the survivor list, so the loads overlap the staging below.  Rows nobody
  // reads are loaded too: with ~47 boxes per image nearly every window row is under some box.
  float2v bf[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float2v z = {0.0f, 0.0f};
    bf[k] = pv ? *reinterpret_cast<const float2v *>(pb + (size_t)k * plane + p) : z;
  }

  for (int s0 = 0; s0 < m_b; s0 += kSurvBatch) {
    const int ns = min(kSurvBatch, m_b - s0);
    __syncthreads();  // the previous batch's readers are done
    // stage coefficients (one 128-byte row per survivor) and row ranges
    for (int idx = threadIdx.x; idx < ns * 32; idx += 256) {
      const int i = idx >> 5, k = idx & 31;
      coef[i][k] = G.det[(size_t)G.slot_map[base + s0 + i] * G.det_stride + 6 + k];
    }
    if (threadIdx.x < ns) {
      int lo, hi;
      const float *d = G.det + (size_t)G.slot_map[base + s0 + threadIdx.x] * G.det_stride;
      row_range(d, lo, hi);
      rlo[threadIdx.x] = lo;
      rhi[threadIdx.x] = hi;
      const long long slot = (long long)base + s0 + threadIdx.x;   // the upsample's box pass reads this record
      if (blockIdx.x == 0 && slot < G.capacity) G.table[slot] = box_ranges(d[0], d[1], d[2], d[3], G.g.h0, G.g.w0);
    }
    __syncthreads();

    for (int i = 0; i < ns; ++i) {
      const int lo = rlo[i], hi = rhi[i];
      if (hi < ra || lo > rb) continue;  // wave-uniform
      const long long slot = (long long)base + s0 + i;
      if (slot >= G.capacity) continue;
      float2v acc = {0.0f, 0.0f};
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 c = *reinterpret_cast<const float4 *>(&coef[i][4 * k4]);
        const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2v cv = {cc[j], cc[j]};
          acc = __builtin_elementwise_fma(cv, bf[4 * k4 + j], acc);
        }
      }
      if (prow >= lo && prow <= hi) *reinterpret_cast<float2v *>(G.logits + (size_t)slot * npx + p) = acc;
    }
  }
}

struct UpParams {
  const float *det;
  const int *slot_map;
  const int *slot_base;
  const float *logits;
  void *out;
  int batch, max_det, det_stride;
  long long capacity;
  MaskGeom g;
  int rows_per_block;
  int lds_rows;  // logits rows the LDS buffer can hold
  const int4 *table;  // split path: per-slot pixel ranges of the box {xa, xb, ya, yb}
  int *overflow;
};

template <typename OutT>
struct MaskOut;
template <>
struct MaskOut<uint8_t> {
  static constexpr int PX = 16;
  static __device__ __forceinline__ void store_vec(uint8_t *p, const uint32_t bits) {
    uint4 v;
    // expand 16 bits to 16 bytes of 0/1
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t n = (bits >> (4 * q)) & 0xfu;
      w[q] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
    }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    *reinterpret_cast<uint4 *>(p) = v;
  }
  static __device__ __forceinline__ void store_vec_nt(uint8_t *p, const uint32_t bits) {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t n = (bits >> (4 * q)) & 0xfu;
      w[q] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
    }
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v = {w[0], w[1], w[2], w[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(p));
  }
  static __device__ __forceinline__ void store1(uint8_t *p, bool on) { *p = on ? 1 : 0; }
};
template <>
struct MaskOut<float> {
  static constexpr int PX = 4;
  static __device__ __forceinline__ void store_vec(float *p, const uint32_t bits) {
    *reinterpret_cast<float4 *>(p) = make_float4((bits & 1u) ? 1.f : 0.f, (bits & 2u) ? 1.f : 0.f, (bits & 4u) ? 1.f : 0.f, (bits & 8u) ? 1.f : 0.f);
  }
  static __device__ __forceinline__ void store_vec_nt(float *p, const uint32_t bits) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {(bits & 1u) ? 1.f : 0.f, (bits & 2u) ? 1.f : 0.f, (bits & 4u) ? 1.f : 0.f, (bits & 8u) ? 1.f : 0.f};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
  }
  static __device__ __forceinline__ void store1(float *p, bool on) { *p = on ? 1.0f : 0.0f; }
};

// one record per live slot {xa, xb, ya, yb}: lets mask_box_kernel decide with one load whether a strip has work.
// mask_logits_rows_kernel writes the records itself; this kernel serves the MFMA variant of the logits.
__global__ __launch_bounds__(256) void mask_slot_table_kernel(GemmParams G) {
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= min((long long)G.slot_base[G.batch], G.capacity)) return;
  const float *d = G.det + (size_t)G.slot_map[s] * G.det_stride;
  G.table[s] = box_ranges(d[0], d[1], d[2], d[3], G.g.h0, G.g.w0);
}

// Zeros for every live slot, written the way that reaches the memset rate on MI355X (profiles/r01/write_ceiling.md):
// ONE workgroup per CU at most, grid-stride over 16-byte chunks, so that at any moment the chip writes one compact
// ~1 MiB window that sweeps the buffer in address order (6.36 TB/s; thousands of workgroups each streaming its own
// strip stay at 5.5-5.7 TB/s).  With a single wave per SIMD nothing hides latency, so the loop is kept to a handful
// of instructions: no box test here -- mask_upsample_kernel (box_only) overwrites the few per cent of chunks that
// meet a box afterwards.
__global__ __launch_bounds__(256) void mask_zero_fill_kernel(const int *__restrict__ slot_base, int batch, long long capacity,
                                                            long long chunks_per_slot, uint4 *__restrict__ out, int *__restrict__ overflow) {
  const long long total_all = slot_base[batch];
  if (blockIdx.x == 0 && threadIdx.x == 0 && overflow) *overflow = total_all > capacity ? (int)(total_all - capacity) : 0;
  const long long total = total_all < capacity ? total_all : capacity;
  const long long n = total * chunks_per_slot;
  const long long stride = (long long)gridDim.x * 256;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 zero = {0u, 0u, 0u, 0u};
  u4 *o = reinterpret_cast<u4 *>(out);
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n; g += stride) {
    __builtin_nontemporal_store(zero, o + g);   // plain stores run as fast here but leave write-backs that slow the next kernels
  }
}

// Second pass of the split path: the chunks that meet a slot's box, and nothing else.  grid: x = row strips, y = slot.
// Every lane owns one 16-byte chunk column of the box and walks down the strip's box rows, so the horizontal taps and
// weights of its pixels are computed once and all lanes of a wave have work (a lane-per-chunk sweep over whole rows
// leaves ~3/4 of the lanes idle on boxes ~1/7 of the frame wide).  Same arithmetic as mask_upsample_kernel.
template <typename OutT>
__global__ __launch_bounds__(256) void mask_box_kernel(UpParams U) {
  using MO = MaskOut<OutT>;
  constexpr int PX = MO::PX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *L = reinterpret_cast<float *>(smem);

  const long long slot = blockIdx.y;
  if (slot >= U.slot_base[U.batch]) return;   // beyond the live slots (the grid spans the buffer's capacity)
  const int4 t = U.table[slot];               // all zero for a box that holds no pixel
  const int ra = max((int)blockIdx.x * U.rows_per_block, t.z), rb = min(((int)blockIdx.x + 1) * U.rows_per_block, t.w);
  if (ra >= rb) return;           // most workgroups: no row of the box in this strip
  const int H0 = U.g.h0, W0 = U.g.w0;
  const int wh = U.g.bottom - U.g.top, ww = U.g.right - U.g.left, mw = U.g.mw;
  const float scale_h = (float)wh / (float)H0, scale_w = (float)ww / (float)W0;
  OutT *out = reinterpret_cast<OutT *>(U.out) + (size_t)slot * H0 * W0;

  const int ca = t.x / PX, cb = (t.y + PX - 1) / PX;   // chunk columns that meet the box (rows are whole chunks)
  // Staged: the logits rows this strip's box rows read, and of those only the columns its chunk columns read, plus
  // one more column (the last logits column repeated past the right border): the right-hand tap of a pixel is then
  // always the next float -- ix1 == ix0 only at the right border, where both taps read the last column -- and one
  // 8-byte LDS read fetches both.
  int c_lo, ls, ly_first;
  {
    int i0, i1, j0, j1; float l0, l1;
    bilinear_src(scale_w, ca * PX, ww, c_lo, i1, l0, l1);
    bilinear_src(scale_w, cb * PX - 1, ww, j0, j1, l0, l1);
    ls = j0 + 2 - c_lo;                   // columns c_lo .. j0 + 1
    bilinear_src(scale_h, ra, wh, i0, i1, l0, l1);
    ly_first = i0;
    bilinear_src(scale_h, rb - 1, wh, j0, j1, l0, l1);
    const int nrows = j1 - ly_first + 1;  // <= U.lds_rows by construction on the host
    const float *src = U.logits + (size_t)slot * wh * mw + (size_t)ly_first * mw + U.g.left;
    for (int idx = threadIdx.x; idx < nrows * ls; idx += 256) {
      const int r = idx / ls, c = idx - r * ls;
      L[idx] = src[(size_t)r * mw + min(c_lo + c, ww - 1)];
    }
    __syncthreads();
  }

  const int lpr = min(cb - ca, 256);                   // lanes per row
  const int groups = 256 / lpr;                        // row groups: each takes consecutive rows of the strip
  const int lr = threadIdx.x / lpr, lc = threadIdx.x - lr * lpr;
  if (lr >= groups) return;
  const int rows_per_group = (rb - ra + groups - 1) / groups;
  const int y_first = ra + lr * rows_per_group, y_last = min(y_first + rows_per_group, rb);
  for (int chunk = ca + lc; chunk < cb; chunk += lpr) {
    const int x0 = chunk * PX;
    typedef float f2 __attribute__((ext_vector_type(2)));
    int oa[PX];   // byte offset of the left tap within a staged row
    f2 wx[PX];    // its weights {left, right}
    uint32_t xmask = 0;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      int ia, ib;
      float la, lb;
      bilinear_src(scale_w, x0 + k, ww, ia, ib, la, lb);
      oa[k] = (ia - c_lo) * 4;
      wx[k] = f2{la, lb};
      if (x0 + k >= t.x && x0 + k < t.y) xmask |= 1u << k;
    }
    typedef float taps __attribute__((ext_vector_type(2), aligned(4)));
    // horizontal interpolation of one staged row at pixel k: depends on the source row only, so it is kept in
    // registers across the consecutive output rows that share a source row pair (6-7 rows at 1080p from 160x160)
    auto hrow = [&](int iy, int k) -> float {
      const taps a = *reinterpret_cast<const taps *>(reinterpret_cast<const char *>(L + (iy - ly_first) * ls) + oa[k]);
      const f2 p = f2{a.x, a.y} * wx[k];   // packed fp32 multiply, each half rounded as the scalar form
      return p.x + p.y;
    };
    f2 tt[PX];   // {upper, lower} source row, interpolated horizontally
    int cur0 = -1, cur1 = -1;
    for (int y = y_first; y < y_last; ++y) {
      int iy0, iy1; float ly0, ly1;
      bilinear_src(scale_h, y, wh, iy0, iy1, ly0, ly1);
      if (iy0 != cur0 || iy1 != cur1) {
        if (iy0 == cur1) {
#pragma unroll
          for (int k = 0; k < PX; ++k) tt[k].x = tt[k].y;
        } else {
#pragma unroll
          for (int k = 0; k < PX; ++k) tt[k].x = hrow(iy0, k);
        }
        if (iy1 == iy0) {
#pragma unroll
          for (int k = 0; k < PX; ++k) tt[k].y = tt[k].x;
        } else {
#pragma unroll
          for (int k = 0; k < PX; ++k) tt[k].y = hrow(iy1, k);
        }
        cur0 = iy0; cur1 = iy1;
      }
      const f2 lyv = {ly0, ly1};
      uint32_t bits = 0;
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        const f2 q = tt[k] * lyv;
        if (q.x + q.y > 0.0f) bits |= 1u << k;
      }
      OutT *p = out + (size_t)y * W0 + x0;
      MO::store_vec_nt(p, bits & xmask);
    }
  }
}

// grid: x = row strips, y = slot
template <typename OutT>
__global__ __launch_bounds__(256) void mask_upsample_kernel(UpParams U) {
  using MO = MaskOut<OutT>;
  constexpr int PX = MO::PX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *L = reinterpret_cast<float *>(smem);

  const long long slot = blockIdx.y;
  const int total = U.slot_base[U.batch];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && U.overflow)
    *U.overflow = total > U.capacity ? (int)(total - U.capacity) : 0;  // survivors that did not fit
  if (slot >= total || slot >= U.capacity) return;
  const int e = U.slot_map[slot];
  const float *d = U.det + (size_t)e * U.det_stride;
  const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3];
  const int H0 = U.g.h0, W0 = U.g.w0;
  const int wh = U.g.bottom - U.g.top, ww = U.g.right - U.g.left, mw = U.g.mw;
  const float scale_h = (float)wh / (float)H0, scale_w = (float)ww / (float)W0;
  OutT *out = reinterpret_cast<OutT *>(U.out) + (size_t)slot * H0 * W0;

  const int y_begin = blockIdx.x * U.rows_per_block;
  const int y_end = min(y_begin + U.rows_per_block, H0);
  // rows of this strip that intersect the box: y >= by1 && y < by2
  int ya = max(y_begin, (int)ceilf(by1));
  int yb = min(y_end, (int)ceilf(by2));  // first integer y with y >= by2
  const bool any = ya < yb && bx1 < bx2;

  int ly_first = 0;
  if (any) {
    int i0, i1; float l0, l1;
    bilinear_src(scale_h, ya, wh, i0, i1, l0, l1);
    ly_first = i0;
    int j0, j1;
    bilinear_src(scale_h, yb - 1, wh, j0, j1, l0, l1);
    const int nrows = j1 - ly_first + 1;  // <= U.lds_rows by construction on the host
    const float *src = U.logits + (size_t)slot * wh * mw + (size_t)ly_first * mw + U.g.left;
    for (int idx = threadIdx.x; idx < nrows * ww; idx += 256) {
      const int r = idx / ww, c = idx - r * ww;
      L[idx] = src[(size_t)r * mw + c];
    }
    __syncthreads();
  }

  const int chunks = (W0 + PX - 1) / PX;
  const bool vec_ok = (W0 % PX) == 0;
  const int nwork = (y_end - y_begin) * chunks;
  for (int wi = threadIdx.x; wi < nwork; wi += 256) {
    const int y = y_begin + wi / chunks;
    const int x0 = (wi % chunks) * PX;
    OutT *p = out + (size_t)y * W0 + x0;
    uint32_t bits = 0;
    const float fy = (float)y;
    const bool row_in = any && (fy >= by1) && (fy < by2);
    if (row_in && (float)(x0 + PX - 1) >= bx1 && (float)x0 < bx2) {
      int iy0, iy1; float ly0, ly1;
      bilinear_src(scale_h, y, wh, iy0, iy1, ly0, ly1);
      const float *r0 = L + (iy0 - ly_first) * ww;
      const float *r1 = L + (iy1 - ly_first) * ww;
#pragma unroll 4
      for (int k = 0; k < PX; ++k) {
        const int x = x0 + k;
        const float fx = (float)x;
        if (x < W0 && fx >= bx1 && fx < bx2) {
          int ix0, ix1; float lx0, lx1;
          bilinear_src(scale_w, x, ww, ix0, ix1, lx0, lx1);
          const float t0 = r0[ix0] * lx0 + r0[ix1] * lx1;
          const float t1 = r1[ix0] * lx0 + r1[ix1] * lx1;
          const float v = t0 * ly0 + t1 * ly1;
          if (v > 0.0f) bits |= 1u << k;
        }
      }
    }
    if (vec_ok) {
      MO::store_vec_nt(p, bits);  // streaming stores: masks are written once, never re-read here (+2 % measured)
    } else {
      for (int k = 0; k < PX && x0 + k < W0; ++k) MO::store1(p + k, (bits >> k) & 1u);
    }
  }
}

void crop_window(int mh, int mw, int h0, int w0, MaskGeom &g) {
  // ultralytics scale_masks: gain = min(mh/h0, mw/w0); pad = (mw - w0*gain)/2, (mh - h0*gain)/2;
  // top,left = int(pad_y), int(pad_x); bottom,right = int(mh - pad_y), int(mw - pad_x)
  const double gh = (double)mh / (double)h0, gw = (double)mw / (double)w0;
  const double gain = gh < gw ? gh : gw;
  const double pad_x = ((double)mw - (double)w0 * gain) / 2.0;
  const double pad_y = ((double)mh - (double)h0 * gain) / 2.0;
  g.mh = mh; g.mw = mw; g.h0 = h0; g.w0 = w0;
  g.top = (int)pad_y; g.left = (int)pad_x;
  g.bottom = (int)((double)mh - pad_y); g.right = (int)((double)mw - pad_x);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int64_t lmi_mask_workspace_bytes(int64_t capacity_masks, int mh, int mw) {
  if (capacity_masks <= 0 || mh <= 0 || mw <= 0) return 0;
  // logits (capacity x mh x mw fp32) + one int4 box record per slot
  return (int64_t)(align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256) + align_up((size_t)capacity_masks * sizeof(int4), 256));
}

static int mask_check_common(const char *fn, int batch, int nm, int mh, int mw, int h0, int w0, int max_det, int det_stride,
                             int64_t capacity_masks, MaskGeom &g) {
  LMI_REQUIRE(nm == LMI_MAX_MASK_DIM, LMI_ERR_LIMIT, "%s: nm=%d, only %d prototype channels are supported", fn, nm, LMI_MAX_MASK_DIM);
  LMI_REQUIRE(batch > 0 && batch <= 65535 && mh > 0 && mw > 0 && h0 > 0 && w0 > 0 && max_det > 0 && det_stride >= 6 + nm, LMI_ERR_BAD_ARG,
              "%s: bad shape batch=%d proto=%dx%d orig=%dx%d max_det=%d det_stride=%d", fn, batch, mh, mw, h0, w0, max_det, det_stride);
  LMI_REQUIRE(capacity_masks > 0 && capacity_masks <= 65535, LMI_ERR_LIMIT, "%s: capacity_masks %lld not in 1..65535", fn, (long long)capacity_masks);
  crop_window(mh, mw, h0, w0, g);
  const int wh = g.bottom - g.top, ww = g.right - g.left;
  LMI_REQUIRE(wh > 0 && ww > 0 && g.top >= 0 && g.left >= 0 && g.bottom <= mh && g.right <= mw, LMI_ERR_BAD_ARG, "%s: empty crop window", fn);
  return LMI_OK;
}

int lmi_mask_logits(const float *proto, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                    const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, int64_t capacity_masks,
                    void *workspace, int64_t workspace_bytes, void *stream) {
  LMI_REQUIRE(proto && det && slot_map && slot_base && workspace, LMI_ERR_BAD_ARG, "lmi_mask_logits: NULL pointer");
  MaskGeom g;
  int rc = mask_check_common("lmi_mask_logits", batch, nm, mh, mw, h0, w0, max_det, det_stride, capacity_masks, g);
  if (rc) return rc;
  LMI_REQUIRE(lmi_is_aligned(workspace, 256) && lmi_is_aligned(proto, 4), LMI_ERR_ALIGNMENT, "lmi_mask_logits: workspace must be 256-byte aligned");
  LMI_REQUIRE(workspace_bytes >= lmi_mask_workspace_bytes(capacity_masks, mh, mw), LMI_ERR_WORKSPACE, "lmi_mask_logits: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)lmi_mask_workspace_bytes(capacity_masks, mh, mw));
  GemmParams G;
  G.proto = proto; G.det = det; G.slot_map = slot_map; G.slot_base = slot_base; G.logits = reinterpret_cast<float *>(workspace);
  G.batch = batch; G.max_det = max_det; G.det_stride = det_stride; G.capacity = capacity_masks; G.g = g;
  G.table = reinterpret_cast<int4 *>(reinterpret_cast<char *>(workspace) + align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256));
  const int npx = (g.bottom - g.top) * mw;
  // LMI_MASK_LOGITS=mfma selects the dense MFMA tiles for every shape (a measurement knob: bench.py --mask-logits mfma)
  const char *mode = getenv("LMI_MASK_LOGITS");
  const bool force_mfma = mode && strcmp(mode, "mfma") == 0;
  if (!force_mfma && (mw % 2) == 0 && lmi_is_aligned(proto, 8)) {
    const int tiles = (npx + 127) / 128;
    hipLaunchKernelGGL(mask_logits_rows_kernel, dim3((tiles + 3) / 4, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
    return lmi_check_launch("mask_logits_rows_kernel");
  }
  hipLaunchKernelGGL(mask_slot_table_kernel, dim3((unsigned)((capacity_masks + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
  // dense MFMA tiles: 4 adjacent pixels per lane when rows are whole 16-byte vectors, else one pixel per lane
  const bool vec4 = (mw % 4) == 0 && lmi_is_aligned(proto, 16);
  const int ntiles = (npx + (vec4 ? 127 : 31)) / (vec4 ? 128 : 32);
  G.tiles_per_block = vec4 ? 1 : 4;
  const int gx = (ntiles + 4 * G.tiles_per_block - 1) / (4 * G.tiles_per_block);
  dim3 ggrid(gx, (max_det + 31) / 32, batch);
  if (vec4) hipLaunchKernelGGL(mask_gemm_kernel<4>, ggrid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
  else hipLaunchKernelGGL(mask_gemm_kernel<1>, ggrid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G);
  return lmi_check_launch("mask_gemm_kernel");
}

namespace {

struct UpCall {
  UpParams U;
  MaskGeom g;
  hipStream_t s;
  int px;       // pixels per 16-byte chunk
  bool split;   // rows are whole chunks: zero fill + box pass
};

int up_setup(const char *fn, void *workspace, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
             const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, void *out_masks, int out_dtype,
             int64_t capacity_masks, int32_t *overflow, void *stream, UpCall &C) {
  LMI_REQUIRE(workspace && det && slot_map && slot_base && out_masks, LMI_ERR_BAD_ARG, "%s: NULL pointer", fn);
  int rc = mask_check_common(fn, batch, nm, mh, mw, h0, w0, max_det, det_stride, capacity_masks, C.g);
  if (rc) return rc;
  LMI_REQUIRE(out_dtype == LMI_U8 || out_dtype == LMI_F32, LMI_ERR_BAD_ARG, "%s: out_dtype must be LMI_U8 or LMI_F32", fn);
  LMI_REQUIRE(lmi_is_aligned(out_masks, 16) && lmi_is_aligned(workspace, 256), LMI_ERR_ALIGNMENT,
              "%s: out_masks must be 16-byte and workspace 256-byte aligned", fn);
  UpParams &U = C.U;
  U.det = det; U.slot_map = slot_map; U.slot_base = slot_base; U.logits = reinterpret_cast<const float *>(workspace); U.out = out_masks;
  U.batch = batch; U.max_det = max_det; U.det_stride = det_stride; U.capacity = capacity_masks; U.g = C.g; U.overflow = overflow;
  U.table = reinterpret_cast<int4 *>(reinterpret_cast<char *>(workspace) + align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256));
  U.rows_per_block = 0; U.lds_rows = 0;
  C.s = reinterpret_cast<hipStream_t>(stream);
  C.px = out_dtype == LMI_U8 ? 16 : 4;
  C.split = (w0 % C.px) == 0;
  return LMI_OK;
}

// rows per strip: the largest power of two <= start whose staged logits rows fit 48 KiB of LDS
int up_strip(UpCall &C, int start, int extra_cols) {
  const int wh = C.g.bottom - C.g.top, ww = C.g.right - C.g.left;
  int rpb = start, lds_rows = 0;
  for (;; rpb >>= 1) {
    lds_rows = (int)((double)rpb * (double)wh / (double)C.g.h0) + 3;
    if (lds_rows > wh) lds_rows = wh;
    if ((size_t)lds_rows * (ww + extra_cols) * sizeof(float) <= 48 * 1024 || rpb == 1) break;
  }
  if ((size_t)lds_rows * (ww + extra_cols) * sizeof(float) > 64 * 1024) {
    lmi_set_error("lmi_mask_upsample: prototype rows too wide for LDS (%d)", ww);
    return LMI_ERR_LIMIT;
  }
  C.U.rows_per_block = rpb; C.U.lds_rows = lds_rows;
  return LMI_OK;
}

// zeros for every live slot at the memset rate: one workgroup per CU, the compact write window (see the kernel)
int up_fill(const int32_t *slot_base, int batch, int h0, int w0, int px, void *out_masks, int64_t capacity_masks, int32_t *overflow,
            hipStream_t s) {
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
    ncu = 256;
  hipLaunchKernelGGL(mask_zero_fill_kernel, dim3(ncu), dim3(256), 0, s, slot_base, batch, (long long)capacity_masks,
                     (long long)h0 * (w0 / px), reinterpret_cast<uint4 *>(out_masks), overflow);
  return lmi_check_launch("mask_zero_fill_kernel");
}

// the chunks that meet a box; ALU-bound, not store-bound: tall strips, so that a lane's horizontal taps serve many rows
int up_boxes(UpCall &C, int out_dtype) {
  int rc = up_strip(C, 256, 1);
  if (rc) return rc;
  const size_t lds = (size_t)C.U.lds_rows * (C.g.right - C.g.left + 1) * sizeof(float);
  dim3 grid((C.g.h0 + C.U.rows_per_block - 1) / C.U.rows_per_block, (unsigned)C.U.capacity);
  if (out_dtype == LMI_U8)
    hipLaunchKernelGGL(mask_box_kernel<uint8_t>, grid, dim3(256), lds, C.s, C.U);
  else
    hipLaunchKernelGGL(mask_box_kernel<float>, grid, dim3(256), lds, C.s, C.U);
  return lmi_check_launch("mask_box_kernel");
}

}  // namespace

int lmi_mask_upsample(void *workspace, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                      const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, void *out_masks, int out_dtype,
                      int64_t capacity_masks, int32_t *overflow, void *stream) {
  UpCall C;
  int rc = up_setup("lmi_mask_upsample", workspace, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, out_masks,
                    out_dtype, capacity_masks, overflow, stream, C);
  if (rc) return rc;
  // Split path (rows are whole 16-byte chunks): zeros for every live slot at the memset rate, then mask_box_kernel
  // overwrites only the chunks that meet a box.  LMI_MASK_UPSAMPLE=single keeps the one-kernel path (measurement).
  const char *single = getenv("LMI_MASK_UPSAMPLE");
  if (C.split && !(single && strcmp(single, "single") == 0)) {
    rc = up_fill(slot_base, batch, h0, w0, C.px, out_masks, capacity_masks, overflow, C.s);
    if (rc) return rc;
    return up_boxes(C, out_dtype);
  }
  rc = up_strip(C, 16, 0);
  if (rc) return rc;
  const size_t lds = (size_t)C.U.lds_rows * (C.g.right - C.g.left) * sizeof(float);
  dim3 ugrid((h0 + C.U.rows_per_block - 1) / C.U.rows_per_block, (unsigned)capacity_masks);
  if (out_dtype == LMI_U8)
    hipLaunchKernelGGL(mask_upsample_kernel<uint8_t>, ugrid, dim3(256), lds, C.s, C.U);
  else
    hipLaunchKernelGGL(mask_upsample_kernel<float>, ugrid, dim3(256), lds, C.s, C.U);
  return lmi_check_launch("mask_upsample_kernel");
}

int lmi_process_mask(const float *proto, int batch, int nm, int mh, int mw, const float *det, int max_det,
                     int det_stride, const int32_t *slot_map, const int32_t *slot_base, int h0, int w0,
                     void *out_masks, int out_dtype, int64_t capacity_masks, int32_t *overflow, void *workspace,
                     int64_t workspace_bytes, void *stream) {
  int rc = lmi_mask_logits(proto, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, capacity_masks,
                           workspace, workspace_bytes, stream);
  if (rc) return rc;
  return lmi_mask_upsample(workspace, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, out_masks,
                           out_dtype, capacity_masks, overflow, stream);
}

}  // extern "C"
