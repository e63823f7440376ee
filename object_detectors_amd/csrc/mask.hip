// Segmentation mask assembly for gfx950.
//
// Replaces `ops.process_mask_native(proto[i], pred[:, 6:], pred[:, :4], orig_img.shape[:2])` and
// `masks[M]` (object_detectors/yolov8_lmi/model.py:235-236; in-tree YOLOv5 analogue
// object_detectors/yolov5_lmi/trt/yolov5_trt.py:373-415).  Semantics: oracle/masks.py.
//
// Kernels
//  mask_logits_rows_kernel  logits[slot][y][x] = sum_k coef[slot][k] * proto[b][k][y][x], only for the rows and
//                        columns of the letterbox crop window that the survivor's box can read, as a k-ordered
//                        fmaf chain on the vector ALU (v_pk_fma_f32); proto is read once.
//  mask_gemm_kernel      the same product as dense 32-survivor tiles on the exact-fp32 MFMA
//                        (v_mfma_f32_32x32x2_f32); kept for odd prototype widths.
//  The upsample is HBM-write-bound (m*H0*W0 bytes of uint8 per image, ~97 % of them zeros outside the boxes) and
//  runs as two passes:
//  mask_slot_table_kernel  per slot, the integer pixel ranges of its box.
//  mask_zero_fill_kernel   zeros for all live slots from ONE workgroup per CU: the write pattern that reaches the
//                        memset rate (6.4 TB/s; profiles/r01/write_ceiling.md).
//  mask_box_kernel       the chunks that meet a box: bilinear (align_corners=False) -> box crop -> `> 0`
//                        from logits rows staged in LDS, 16 bytes per lane per store (rows that are not whole
//                        16-byte chunks: the same stores at whatever address, the row's last chunk pixel by pixel).
//  mask_upsample_kernel  the same result in one pass (per survivor, per strip of output rows, zeros included),
//                        for tiny frames (h0 + w0 <= 128, where aten interpolates with its 4-weight kernel);
//                        LMI_MASK_UPSAMPLE_SINGLE selects it for every shape.
// Every kernel takes the geometry per image (MaskBatch): a batch of differently sized originals
// (object_detectors/yolov8_lmi/model.py:215-236 loops over them) is one launch, masks packed back to back.
#include <hip/hip_fp16.h>

#include <mutex>

#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct MaskGeom {
  int mh, mw;                    // proto size
  int top, bottom, left, right;  // scale_masks crop window in proto space
  int h0, w0;                    // original frame size
};

// ultralytics scale_masks: gain = min(mh/h0, mw/w0); pad = (mw - w0*gain)/2, (mh - h0*gain)/2;
// top,left = int(pad_y), int(pad_x); bottom,right = int(mh - pad_y), int(mw - pad_x)   (Python floats: fp64)
__host__ __device__ inline void crop_window(int mh, int mw, int h0, int w0, MaskGeom &g) {
  const double gh = (double)mh / (double)h0, gw = (double)mw / (double)w0;
  const double gain = gh < gw ? gh : gw;
  const double pad_x = ((double)mw - (double)w0 * gain) / 2.0;
  const double pad_y = ((double)mh - (double)h0 * gain) / 2.0;
  g.mh = mh; g.mw = mw; g.h0 = h0; g.w0 = w0;
  g.top = (int)pad_y; g.left = (int)pad_x;
  g.bottom = (int)((double)mh - pad_y); g.right = (int)((double)mw - pad_x);
}

// The batch's originals: one size for all (images == nullptr; the geometry comes ready from the host), or a device
// table with one record per image.
struct MaskBatch {
  const lmi_mask_image *images;
  MaskGeom g;
};
__device__ __forceinline__ MaskGeom image_geom(const MaskBatch &M, int b) {
  if (M.images == nullptr) return M.g;
  MaskGeom g;
  crop_window(M.g.mh, M.g.mw, M.images[b].h0, M.images[b].w0, g);
  return g;
}
// element offset of a slot's mask in the output: slot * H0 * W0, or behind the image's first mask
__device__ __forceinline__ long long mask_offset(const MaskBatch &M, const MaskGeom &g, int b, long long slot, int slot_base_b) {
  if (M.images == nullptr) return slot * (long long)g.h0 * g.w0;
  return (long long)M.images[b].out_offset + (slot - slot_base_b) * (long long)g.h0 * g.w0;
}

struct GemmParams {
  const void *proto;   // (batch, 32, mh, mw), fp32 or (rows kernel only) fp16
  const float *det;    // (batch, max_det, det_stride)
  const int *slot_map;
  const int *slot_base;
  float *logits;  // (capacity, wh*mw)
  int batch, max_det, det_stride;
  long long capacity;
  MaskBatch M;
  int lstride;          // floats per slot in `logits` (mh * mw)
  int tiles_per_block;  // pixel tiles (32*V px) per wave
  int4 *table;          // per-slot pixel ranges of the box {xa, xb, ya, yb} for mask_box_kernel
};

// aten upsample_bilinear2d, align_corners=False, as torch's x86 CPU build computes it (oracle/geometry.py; the same
// forms as revert.hip): the source coordinate is ONE fused multiply-add, and each pair of taps is combined as
// fma(a, wa, b * wb), horizontal pairs first (aten's separable kernel) -- only these reproduce F.interpolate bit for
// bit.  Outputs with h0 + w0 <= 128 take aten's per-pixel 4-weight kernel (mask_upsample_kernel only).
__device__ __forceinline__ void bilinear_src(float scale, int d, int in_n, int &i0, int &i1, float &l0, float &l1) {
  float s = fmaf(scale, (float)d + 0.5f, -0.5f);
  s = fmaxf(s, 0.0f);
  i0 = min((int)s, in_n - 1);
  i1 = i0 + ((i0 < in_n - 1) ? 1 : 0);
  l1 = fminf(fmaxf(s - (float)i0, 0.0f), 1.0f);
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
  const MaskGeom g = image_geom(G.M, b);
  const int wh = g.bottom - g.top;

  // Logits rows each survivor's upsample will read: the bilinear taps of the output rows inside its box
  // (same arithmetic as mask_upsample_kernel).  Only those are computed and written.
  __shared__ int rlo[32], rhi[32], runion[2];
  if (threadIdx.x < 32) {
    int lo = 1, hi = 0;
    const int i = row0 + (int)threadIdx.x;
    if (i < m_b) {
      const float *d = G.det + (size_t)G.slot_map[base + i] * G.det_stride;
      const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3];
      const int ya = max(0, (int)ceilf(by1)), yb = min(g.h0, (int)ceilf(by2));
      if (ya < yb && bx1 < bx2) {
        const float scale_h = (float)wh / (float)g.h0;
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
  const int npx = wh * g.mw;
  const int plane = g.mh * g.mw;
  const float *pb = reinterpret_cast<const float *>(G.proto) + (size_t)b * 32 * plane + (size_t)g.top * g.mw;

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
    if (p0 / g.mw > uhi || min(p0 + TP - 1, npx - 1) / g.mw < ulo) continue;  // no survivor of this tile reads these rows
    const int p = p0 + V * (lane & 31);
    const bool pv = p < npx;          // npx % V == 0: a lane's V pixels are valid together and share a row
    const int prow = p / g.mw;
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
          float *o = G.logits + (size_t)slot * G.lstride + p;
          if constexpr (V == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
          else *o = acc[0][r];
        }
      }
    }
  }
}

// Box-sparse variant of the GEMM on the vector ALU, used when prototype rows are whole 8-byte pairs.
// A survivor's upsample reads only the logits under its box -- typically a tenth of the window's rows and a sixth of
// its columns -- so the dense 32-survivor MFMA tile above spends most of its issue slots on values nobody reads
// (measured 82 us per seg640 launch, MFMA-issue bound).  Here a wave keeps a tile of 32 columns x 4 rows of the window
// x 32 channels in registers (2 adjacent px per lane; every row segment of a load is one 128-byte line) and walks the
// image's survivors, doing the k-ordered fmaf chain only for those whose logits rows AND columns meet the tile;
// coefficients and ranges sit in LDS and are read as broadcasts.  The column range is the one mask_box_kernel and
// mask_upsample_kernel can read at most: the box's pixel columns widened to 64-pixel blocks, through the same
// source-index arithmetic, plus the right-hand tap.
// grid: x = groups of 4 tiles (one per wave), y = image
constexpr int kSurvBatch = 128;  // survivors staged per pass: 128 * (32 coef + 4 range) * 4 B = 18 KiB of LDS
constexpr int kLogitTileW = 32, kLogitTileH = 4;

// PT: float, or __half for the prototypes of an fp16 model (`fp16=True`, object_detectors/yolov8_lmi/model.py:68): every
// value is promoted to fp32 as it is loaded, the result of `protos.float()` without that pass over the tensor.
template <typename PT>
__global__ __launch_bounds__(256, 4) void mask_logits_rows_kernel(GemmParams G) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float coef[kSurvBatch][32];
  __shared__ int4 rng[kSurvBatch];   // logits rows {lo, hi} and prototype columns {lo, hi} a survivor's upsample reads
  __shared__ int emap[kSurvBatch];   // det row of each staged survivor

  const int b = blockIdx.y;
  const int base = G.slot_base[b];
  const int m_b = G.slot_base[b + 1] - base;
  if (m_b <= 0) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const MaskGeom g = image_geom(G.M, b);
  const int wh = g.bottom - g.top, ww = g.right - g.left, mw = g.mw;
  const int plane = g.mh * mw;
  const PT *pb = reinterpret_cast<const PT *>(G.proto) + (size_t)b * 32 * plane + (size_t)g.top * mw;
  const float scale_h = (float)wh / (float)g.h0, scale_w = (float)ww / (float)g.w0;
  const int tiles_x = (mw + kLogitTileW - 1) / kLogitTileW;
  // ragged batch: the grid spans the largest window
  if (blockIdx.x > 0 && ((int)blockIdx.x * 4 / tiles_x) * kLogitTileH >= wh) return;

  const int tile = blockIdx.x * 4 + wave;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int ra = ty * kLogitTileH, rb = ra + kLogitTileH - 1;      // rows and columns of this wave's tile
  const int ca = tx * kLogitTileW, cb = ca + kLogitTileW - 1;
  const int prow = ra + (lane >> 4), pcol = ca + 2 * (lane & 15);
  const bool pv = prow < wh && pcol < mw;      // mw even: a lane's two pixels are valid together
  const int p = prow * mw + pcol;

  // The staging below is a chain of dependent loads (slot_base -> slot_map -> det row).  Loads return in order, so a
  // slot_map load issued after the 32 prototype loads would only be seen once all of those have landed, and the det
  // rows one more round trip later: the first batch's slot_map entries are requested BEFORE the prototype loads.
  int e_first = 0;
  if ((int)threadIdx.x < min(kSurvBatch, m_b)) e_first = G.slot_map[base + threadIdx.x];

  // Issued before anything depends on the survivor list, so the loads overlap the staging below.  Tiles nobody
  // reads are loaded too: with ~47 boxes per image nearly every tile is under some box.
  // Buffer loads: one VGPR offset for the lane's pixel pair, the channel's plane as the scalar offset -- 32 separate
  // 64-bit addresses would cost 64 VGPRs at the point where all loads are in flight.  A lane beyond the window gets an
  // offset past num_records: its loads are dropped and return 0.
  float2v bf[32];
  {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(reinterpret_cast<uintptr_t>(pb) & 0xffffffffu));
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(reinterpret_cast<uintptr_t>(pb) >> 32));
    const void *ubase = reinterpret_cast<const void *>(((uintptr_t)hi << 32) | lo);
    constexpr uint32_t esz = (uint32_t)sizeof(PT);
    const uint32_t bytes = (uint32_t)((31u * (uint32_t)plane + (uint32_t)(wh * mw)) * esz);   // this image's window, channel 0 .. 31
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    const uint32_t voff = pv ? (uint32_t)p * esz : 0xfffffff0u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if constexpr (sizeof(PT) == 4) {
        auto t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, (uint32_t)k * (uint32_t)plane * esz, 0);
        const uint32_t t0 = t[0], t1 = t[1];   // (a bit_cast applied to a vector element reads element 0 both times)
        bf[k] = float2v{__uint_as_float(t0), __uint_as_float(t1)};
      } else {   // the lane's pixel pair is one dword of two halves
        const uint32_t t = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)k * (uint32_t)plane * esz, 0);
        const __half2 h = __builtin_bit_cast(__half2, t);
        bf[k] = float2v{__low2float(h), __high2float(h)};
      }
    }
  }

  for (int s0 = 0; s0 < m_b; s0 += kSurvBatch) {
    const int ns = min(kSurvBatch, m_b - s0);
    __syncthreads();  // the previous batch's readers are done
    if ((int)threadIdx.x < ns) emap[threadIdx.x] = s0 == 0 ? e_first : G.slot_map[base + s0 + threadIdx.x];
    __syncthreads();
    // stage coefficients (one 128-byte row per survivor) and ranges
    for (int idx = threadIdx.x; idx < ns * 32; idx += 256) {
      const int i = idx >> 5, k = idx & 31;
      coef[i][k] = G.det[(size_t)emap[i] * G.det_stride + 6 + k];
    }
    if (threadIdx.x < ns) {
      const float *d = G.det + (size_t)emap[threadIdx.x] * G.det_stride;
      const int4 t = box_ranges(d[0], d[1], d[2], d[3], g.h0, g.w0);   // {xa, xb, ya, yb}, all zero for an empty box
      int4 r = make_int4(1, 0, 1, 0);
      if (t.y > t.x) {  // same arithmetic as the upsample kernels
        int i0, i1; float l0, l1;
        bilinear_src(scale_h, t.z, wh, i0, i1, l0, l1);
        r.x = i0;
        bilinear_src(scale_h, t.w - 1, wh, i0, i1, l0, l1);
        r.y = i1;
        bilinear_src(scale_w, t.x & ~63, ww, i0, i1, l0, l1);
        r.z = g.left + i0;
        bilinear_src(scale_w, min((t.y + 63) & ~63, g.w0) - 1, ww, i0, i1, l0, l1);
        r.w = g.left + min(i0 + 1, ww - 1);
      }
      rng[threadIdx.x] = r;
      const long long slot = (long long)base + s0 + threadIdx.x;   // the upsample's box pass reads this record
      if (blockIdx.x == 0 && slot < G.capacity) G.table[slot] = t;
    }
    __syncthreads();

#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)   // interleaved two-fold the loop holds two survivors' coefficients: +32 VGPRs
    for (int i = 0; i < ns; ++i) {
      const int4 r = rng[i];
      if (r.y < ra || r.x > rb || r.w < ca || r.z > cb) continue;  // wave-uniform
      const long long slot = (long long)base + s0 + i;
      if (slot >= G.capacity) continue;
      float2v acc = {0.0f, 0.0f};
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        // the second half's coefficients are not read before the first half is used up: 16 fewer live VGPRs, which is
        // what brings the kernel to 128 (4 waves per SIMD) without spills
        if (k4 == 4) __builtin_amdgcn_sched_barrier(0);
        const float4 c = *reinterpret_cast<const float4 *>(&coef[i][4 * k4]);
        const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2v cv = {cc[j], cc[j]};
          acc = __builtin_elementwise_fma(cv, bf[4 * k4 + j], acc);
        }
      }
      if (pv && prow >= r.x && prow <= r.y && pcol + 1 >= r.z && pcol <= r.w)
        *reinterpret_cast<float2v *>(G.logits + (size_t)slot * G.lstride + p) = acc;
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
  long long capacity;      // slots the output (and the logits workspace) can hold
  long long out_elems;     // ragged originals: elements the output can hold
  MaskBatch M;
  int lstride;             // floats per slot in `logits`
  int rows_per_block;
  int lds_rows;  // logits rows the LDS buffer can hold
  int strips;    // row strips per slot: the grid is one-dimensional, block = slot * strips + strip (grid.y would stop at 65535 slots)
  const int4 *table;  // split path: per-slot pixel ranges of the box {xa, xb, ya, yb}
  int *overflow;
};

// image of a slot: slot_map[slot] = b * max_det + i
__device__ __forceinline__ int slot_image(const UpParams &U, long long slot) { return U.slot_map[slot] / U.max_det; }

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
  // `v > 0` for 16 interpolated pixels straight into 16 bytes of 0/1 without compares (v_cmp + v_cndmask pairs and
  // their SGPR hazards were most of the box pass): a float's bits read as a signed integer are positive for exactly
  // the values > 0, denormals included, and zero or negative for +-0 and everything below; v_med3_i32 clamps that to
  // 0..1 and v_lshl_or assembles the bytes.  (A NaN with a clear sign bit would read as 1: the caller keeps NaNs out
  // of the interpolated rows, see hrow in mask_box_kernel.)  allow: bit k = pixel k lies inside the box's columns.
  typedef float f2v __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ uint32_t pos01(float v) {
    uint32_t r;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(v));
    return r;
  }
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 u4_anywhere __attribute__((aligned(1)));
  // mode 0: p is 16-byte aligned; 1: any address (rows that are not whole chunks; the target runs in unaligned-access
  // mode, one store instruction either way); 2: only the first n pixels, one by one (the row ends inside the chunk)
  static __device__ __forceinline__ void emit_nt(uint8_t *p, const f2v (&v)[8], uint32_t allow, int mode = 0, int n = 16) {
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t d = pos01(v[2 * q].x);
      d |= pos01(v[2 * q].y) << 8;
      d |= pos01(v[2 * q + 1].x) << 16;
      d |= pos01(v[2 * q + 1].y) << 24;
      const uint32_t n = (allow >> (4 * q)) & 0xfu;
      const uint32_t m = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);   // loop-invariant: hoisted out of the row loop
      w[q] = d & m;
    }
    u4 o = {w[0], w[1], w[2], w[3]};
    if (mode == 0) {
      __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(p));
    } else if (mode == 1) {
      __builtin_nontemporal_store(o, reinterpret_cast<u4_anywhere *>(p));
    } else {
      for (int k = 0; k < n; ++k) p[k] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
  }
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
  typedef float f2v __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void emit_nt(float *p, const f2v (&v)[2], uint32_t allow, int mode = 0, int n = 4) {
    uint32_t bits = 0;
    if (v[0].x > 0.0f) bits |= 1u;
    if (v[0].y > 0.0f) bits |= 2u;
    if (v[1].x > 0.0f) bits |= 4u;
    if (v[1].y > 0.0f) bits |= 8u;
    bits &= allow;
    if (mode == 0) {
      store_vec_nt(p, bits);
    } else if (mode == 1) {   // 4-byte aligned, as every float is
      typedef float f4 __attribute__((ext_vector_type(4)));
      typedef f4 f4_anywhere __attribute__((aligned(4)));
      f4 t = {(bits & 1u) ? 1.f : 0.f, (bits & 2u) ? 1.f : 0.f, (bits & 4u) ? 1.f : 0.f, (bits & 8u) ? 1.f : 0.f};
      __builtin_nontemporal_store(t, reinterpret_cast<f4_anywhere *>(p));
    } else {
      for (int k = 0; k < n; ++k) p[k] = ((bits >> k) & 1u) ? 1.0f : 0.0f;
    }
  }
};

// one record per live slot {xa, xb, ya, yb}: lets mask_box_kernel decide with one load whether a strip has work.
// mask_logits_rows_kernel writes the records itself; this kernel serves the MFMA variant of the logits.
__global__ __launch_bounds__(256) void mask_slot_table_kernel(GemmParams G) {
  const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= min((long long)G.slot_base[G.batch], G.capacity)) return;
  const int e = G.slot_map[s];
  const float *d = G.det + (size_t)e * G.det_stride;
  const MaskGeom g = image_geom(G.M, e / G.max_det);
  G.table[s] = box_ranges(d[0], d[1], d[2], d[3], g.h0, g.w0);
}

// Zeros for every live slot, written the way that reaches the memset rate on MI355X (profiles/r01/write_ceiling.md):
// ONE workgroup per CU at most, grid-stride over 16-byte chunks, so that at any moment the chip writes one compact
// ~1 MiB window that sweeps the buffer in address order (6.36 TB/s; thousands of workgroups each streaming its own
// strip stay at 5.5-5.7 TB/s).  With a single wave per SIMD nothing hides latency, so the loop is kept to a handful
// of instructions: no box test here -- mask_upsample_kernel (box_only) overwrites the few per cent of chunks that
// meet a box afterwards.
__global__ __launch_bounds__(256) void mask_zero_fill_kernel(const int *__restrict__ slot_base, int batch, long long capacity,
                                                            long long bytes_per_slot, long long ragged_bytes, uint4 *__restrict__ out,
                                                            int *__restrict__ overflow) {
  long long nbytes = ragged_bytes;   // ragged originals: the host laid the masks out and knows where they end
  if (ragged_bytes < 0) {
    const long long total_all = slot_base[batch];
    if (blockIdx.x == 0 && threadIdx.x == 0 && overflow) *overflow = total_all > capacity ? (int)(total_all - capacity) : 0;
    const long long total = total_all < capacity ? total_all : capacity;
    nbytes = total * bytes_per_slot;
  } else if (blockIdx.x == 0 && threadIdx.x == 0 && overflow) {
    *overflow = 0;
  }
  const long long n = nbytes >> 4;   // whole 16-byte chunks; frames whose rows are not whole chunks may leave a tail
  const long long stride = (long long)gridDim.x * 256;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 zero = {0u, 0u, 0u, 0u};
  u4 *o = reinterpret_cast<u4 *>(out);
  // r03: the workgroup's four waves are kept in step by one bare s_barrier per trip (the instruction alone: no wait for
  // the stores, which a __syncthreads() would add).  Left to themselves the waves drift apart and the window frays:
  // 6.28-6.37 TB/s without, 6.46-6.54 with the barrier on the same 6.18 GB (scratch/fill_skip.hip,
  // profiles/r03/fill_pacing.md; hipMemset: 6.25-6.40).  The trip count is the same for every lane of the workgroup.
  for (long long base = (long long)blockIdx.x * 256; base < n; base += stride) {
    const long long g = base + threadIdx.x;
    if (g < n) __builtin_nontemporal_store(zero, o + g);   // plain stores run as fast here but leave write-backs that slow the next kernels
    __builtin_amdgcn_s_barrier();
  }
  if (blockIdx.x == gridDim.x - 1 && (long long)threadIdx.x < (nbytes & 15))
    reinterpret_cast<unsigned char *>(out)[(n << 4) + threadIdx.x] = 0;
}

// Second pass of the split path: the chunks that meet a slot's box, and nothing else.  grid: one workgroup per (slot, row strip), slot-major.
// Every lane owns one 16-byte chunk column of the box and walks down the strip's box rows, so the horizontal
// weights of its pixels are computed once and all lanes of a wave have work (a lane-per-chunk sweep over whole rows
// leaves ~3/4 of the lanes idle on boxes ~1/7 of the frame wide).  Same arithmetic as mask_upsample_kernel.
// CHUNKED: every image's rows and mask starts are whole 16-byte chunks (the common case: aligned stores only).
template <typename OutT, bool CHUNKED>
__global__ __launch_bounds__(256, 4) void mask_box_kernel(UpParams U) {   // 4 waves per SIMD: at most 128 VGPRs
  using MO = MaskOut<OutT>;
  constexpr int PX = MO::PX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int2 rowtab[256];   // per strip row: {upper | lower << 16 source row, fp32 bits of the lower row's weight}
  float *L = reinterpret_cast<float *>(smem);

  const unsigned slot_u = blockIdx.x / (unsigned)U.strips, strip = blockIdx.x - slot_u * (unsigned)U.strips;
  const long long slot = slot_u;
  if (slot >= U.slot_base[U.batch]) return;   // beyond the live slots (the grid spans the buffer's capacity)
  if (slot >= U.capacity) return;
  const int4 t = U.table[slot];               // all zero for a box that holds no pixel
  const int ra = max((int)strip * U.rows_per_block, t.z), rb = min(((int)strip + 1) * U.rows_per_block, t.w);
  if (ra >= rb) return;           // most workgroups: no row of the box in this strip
  const int b = U.M.images ? slot_image(U, slot) : 0;
  const MaskGeom g = image_geom(U.M, b);
  const int H0 = g.h0, W0 = g.w0;
  const int wh = g.bottom - g.top, ww = g.right - g.left, mw = g.mw;
  const float scale_h = (float)wh / (float)H0, scale_w = (float)ww / (float)W0;
  const long long off = mask_offset(U.M, g, b, slot, U.M.images ? U.slot_base[b] : 0);
  if (U.M.images && off + (long long)H0 * W0 > U.out_elems) return;
  OutT *out = reinterpret_cast<OutT *>(U.out) + off;

  int ca = t.x / PX, cb = (t.y + PX - 1) / PX;   // chunk columns that meet the box (rows are whole chunks)
  // ... widened to whole 64-byte blocks where the rows allow it: a store that covers part of a block costs more than
  // the bytes it saves (this pass's stores alone, no arithmetic: 71 us as the boxes fall, 50 us on whole 128-byte
  // lines although those are ~40 % more bytes).  The added chunks lie outside the box and come out as zeros through
  // the same arithmetic (a branch for them serialises against the other lanes' work: slower); 64 bytes is where
  // the cheaper stores and the extra arithmetic balance (84 us unaligned, 78 / 72 / 85 us at 32 / 64 / 128 bytes).
  constexpr int kBlock = 64 / 16;
  if (((size_t)W0 * sizeof(OutT)) % 64 == 0 && ((size_t)off * sizeof(OutT)) % 64 == 0) {
    ca = ca / kBlock * kBlock;
    cb = (cb + kBlock - 1) / kBlock * kBlock;
  }
  // Rows (or mask starts) that are not whole 16-byte chunks: the chunk columns stay relative to the row start, the
  // stores go to whatever address that gives, and a chunk the row ends in is written pixel by pixel.
  const bool chunked = CHUNKED || ((W0 % PX) == 0 && (off % PX) == 0);
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
    const float *src = U.logits + (size_t)slot * U.lstride + (size_t)ly_first * mw + g.left;
    for (int idx = threadIdx.x; idx < nrows * ls; idx += 256) {
      const int r = idx / ls, c = idx - r * ls;
      L[idx] = src[(size_t)r * mw + min(c_lo + c, ww - 1)];
    }
    // the vertical source pair and weight of every row of the strip, once (every lane would redo them per row)
    if ((int)threadIdx.x < rb - ra) {
      int i0r, i1r; float l0r, l1r;
      bilinear_src(scale_h, ra + threadIdx.x, wh, i0r, i1r, l0r, l1r);
      rowtab[threadIdx.x] = make_int2(i0r | (i1r << 16), __float_as_int(l1r));
    }
    __syncthreads();
  }

  // Each wave takes a quarter of the strip's box rows and sweeps it in address order, R adjacent rows per store
  // instruction (a box ~300 bytes wide: 3 rows).  The pass is bound by its stores (no stores: 53 us, stores alone:
  // 88 us of 91 on seg640), and lanes that each walk their own run of rows -- more reuse of the horizontal taps --
  // scatter one instruction's stores over rows far apart: 91 us against 84 us this way.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpr = min(cb - ca, 64);                    // lanes per row
  const int R = 64 / lpr, rsub = lane / lpr;
  const int lc = lane - rsub * lpr;
  if (rsub >= R) return;
  const int q = (rb - ra + 3) >> 2;
  const int y0w = ra + wave * q;
  const int y_first = y0w + rsub, y_last = min(y0w + q, rb), ystep = R;
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
      // aten: fma(left, w_left, right * w_right).  fmaxf drops a NaN (non-finite prototypes or coefficients) for a
      // negative number: `NaN > 0` is false in the reference too, and the byte assembly below reads float bits as integers
      return fmaxf(fmaf(a.x, wx[k].x, a.y * wx[k].y), -3.0e38f);
    };
    f2 tu[PX / 2], tl[PX / 2];   // upper / lower source row, interpolated horizontally; adjacent pixels paired for packed math
    int cur0 = -1, cur1 = -1;
    for (int y = y_first; y < y_last; y += ystep) {
      const int2 re = rowtab[y - ra];
      const int iy0 = re.x & 0xffff, iy1 = re.x >> 16;
      const float ly1 = __int_as_float(re.y), ly0 = 1.0f - ly1;
      if (iy0 != cur0 || iy1 != cur1) {
        if (iy0 == cur1) {
#pragma unroll
          for (int h = 0; h < PX / 2; ++h) tu[h] = tl[h];
        } else {
#pragma unroll
          for (int h = 0; h < PX / 2; ++h) tu[h] = f2{hrow(iy0, 2 * h), hrow(iy0, 2 * h + 1)};
        }
        if (iy1 == iy0) {
#pragma unroll
          for (int h = 0; h < PX / 2; ++h) tl[h] = tu[h];
        } else {
#pragma unroll
          for (int h = 0; h < PX / 2; ++h) tl[h] = f2{hrow(iy1, 2 * h), hrow(iy1, 2 * h + 1)};
        }
        cur0 = iy0; cur1 = iy1;
      }
      const f2 w0v = {ly0, ly0}, w1v = {ly1, ly1};
      f2 v[PX / 2];
#pragma unroll
      for (int h = 0; h < PX / 2; ++h)   // aten: fma(upper, w_upper, lower * w_lower), two pixels per packed instruction (each half rounds like the scalar form)
        v[h] = __builtin_elementwise_fma(tu[h], w0v, tl[h] * w1v);
      OutT *o = out + (size_t)y * W0 + x0;
      if (CHUNKED || chunked) MO::emit_nt(o, v, xmask);
      else if (x0 + PX <= W0) MO::emit_nt(o, v, xmask, 1);
      else MO::emit_nt(o, v, xmask, 2, W0 - x0);
    }
  }
}

// grid: one workgroup per (slot, row strip), slot-major
template <typename OutT>
__global__ __launch_bounds__(256) void mask_upsample_kernel(UpParams U) {
  using MO = MaskOut<OutT>;
  constexpr int PX = MO::PX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *L = reinterpret_cast<float *>(smem);

  const unsigned slot_u = blockIdx.x / (unsigned)U.strips, strip = blockIdx.x - slot_u * (unsigned)U.strips;
  const long long slot = slot_u;
  const int total = U.slot_base[U.batch];
  if (blockIdx.x == 0 && threadIdx.x == 0 && U.overflow)
    *U.overflow = total > U.capacity ? (int)(total - U.capacity) : 0;  // survivors that did not fit
  if (slot >= total || slot >= U.capacity) return;
  const int e = U.slot_map[slot];
  const int b = e / U.max_det;
  const float *d = U.det + (size_t)e * U.det_stride;
  const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3];
  const MaskGeom g = image_geom(U.M, b);
  const int H0 = g.h0, W0 = g.w0;
  const int wh = g.bottom - g.top, ww = g.right - g.left, mw = g.mw;
  const float scale_h = (float)wh / (float)H0, scale_w = (float)ww / (float)W0;
  const long long off = mask_offset(U.M, g, b, slot, U.M.images ? U.slot_base[b] : 0);
  if (U.M.images && off + (long long)H0 * W0 > U.out_elems) return;
  OutT *out = reinterpret_cast<OutT *>(U.out) + off;
  const bool weights4 = H0 + W0 <= 128;   // aten's kernel choice (oracle/geometry.py:aten_linear_variant)

  const int y_begin = (int)strip * U.rows_per_block;
  if (y_begin >= H0) return;              // ragged batch: the grid spans the tallest frame
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
    const float *src = U.logits + (size_t)slot * U.lstride + (size_t)ly_first * mw + g.left;
    for (int idx = threadIdx.x; idx < nrows * ww; idx += 256) {
      const int r = idx / ww, c = idx - r * ww;
      L[idx] = src[(size_t)r * mw + c];
    }
    __syncthreads();
  }

  const int chunks = (W0 + PX - 1) / PX;
  const bool vec_ok = (W0 % PX) == 0 && (off % PX) == 0;
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
          float v;
          if (weights4) {
            const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
            v = fmaf(w11, r1[ix1], fmaf(w10, r1[ix0], fmaf(w00, r0[ix0], w01 * r0[ix1])));
          } else {
            const float t0 = fmaf(r0[ix0], lx0, r0[ix1] * lx1);
            const float t1 = fmaf(r1[ix0], lx0, r1[ix1] * lx1);
            v = fmaf(t0, ly0, t1 * ly1);
          }
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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// CU count of the current device, asked once per device (the zero fill launches one workgroup per CU)
int device_cu_count() {
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

// Validated shape of one call: the batch's geometry for the kernels and its extents for the launch.
struct MaskCall {
  MaskBatch M;
  int max_h0;          // tallest original
  int max_ww;          // widest crop window
  int max_wh;          // tallest crop window
  double max_ratio;    // largest wh / h0: logits rows per output row
  bool separable;      // every image: h0 + w0 > 128 (aten's separable kernel; smaller frames take its 4-weight form)
  bool chunked;        // every image: rows and mask starts are whole 16-byte chunks
  long long end_elems; // ragged: one past the last mask element
};

int mask_check_common(const char *fn, int batch, int nm, int mh, int mw, int h0, int w0, const lmi_mask_image *images_host,
                      const lmi_mask_image *images_dev, int max_det, int det_stride, int64_t capacity_masks, int px, MaskCall &C) {
  LMI_REQUIRE(nm == LMI_MAX_MASK_DIM, LMI_ERR_LIMIT, "%s: nm=%d, only %d prototype channels are supported", fn, nm, LMI_MAX_MASK_DIM);
  LMI_REQUIRE(batch > 0 && batch <= 65535 && mh > 0 && mw > 0 && max_det > 0 && det_stride >= 6 + nm, LMI_ERR_BAD_ARG,
              "%s: bad shape batch=%d proto=%dx%d max_det=%d det_stride=%d", fn, batch, mh, mw, max_det, det_stride);
  LMI_REQUIRE(capacity_masks > 0 && capacity_masks <= (int64_t)65535 * 1024, LMI_ERR_LIMIT, "%s: capacity_masks %lld not in 1..%lld", fn, (long long)capacity_masks, (long long)65535 * 1024);
  LMI_REQUIRE((images_host == nullptr) == (images_dev == nullptr), LMI_ERR_BAD_ARG, "%s: images_host and images_dev go together", fn);
  C.M.images = images_dev;
  C.max_h0 = 0; C.max_ww = 0; C.max_wh = 0; C.max_ratio = 0.0; C.separable = true; C.chunked = true; C.end_elems = 0;
  const int n = images_host ? batch : 1;
  for (int b = 0; b < n; ++b) {
    const int ih = images_host ? images_host[b].h0 : h0, iw = images_host ? images_host[b].w0 : w0;
    LMI_REQUIRE(ih > 0 && iw > 0, LMI_ERR_BAD_ARG, "%s: image %d: bad original size %dx%d", fn, b, ih, iw);
    MaskGeom g;
    crop_window(mh, mw, ih, iw, g);
    const int wh = g.bottom - g.top, ww = g.right - g.left;
    LMI_REQUIRE(wh > 0 && ww > 0 && g.top >= 0 && g.left >= 0 && g.bottom <= mh && g.right <= mw, LMI_ERR_BAD_ARG, "%s: image %d: empty crop window", fn, b);
    if (b == 0) C.M.g = g;   // uniform originals: the kernels take it from here
    if (ih > C.max_h0) C.max_h0 = ih;
    if (ww > C.max_ww) C.max_ww = ww;
    if (wh > C.max_wh) C.max_wh = wh;
    const double ratio = (double)wh / (double)ih;
    if (ratio > C.max_ratio) C.max_ratio = ratio;
    if (images_host) {
      const lmi_mask_image &im = images_host[b];
      LMI_REQUIRE(im.n_masks >= 0 && im.out_offset >= 0, LMI_ERR_BAD_ARG, "%s: image %d: negative mask count / offset", fn, b);
      const long long end = (long long)im.out_offset + (long long)im.n_masks * ih * iw;
      if (im.n_masks > 0 && end > C.end_elems) C.end_elems = end;
    }
    C.separable = C.separable && ih + iw > 128;
    C.chunked = C.chunked && (iw % px) == 0 && (!images_host || (images_host[b].out_offset % px) == 0);
  }
  if (images_host) { C.M.g.mh = mh; C.M.g.mw = mw; }
  return LMI_OK;
}

}  // namespace

extern "C" {

int64_t lmi_mask_workspace_bytes(int64_t capacity_masks, int mh, int mw) {
  if (capacity_masks <= 0 || mh <= 0 || mw <= 0) return 0;
  // logits (capacity x mh x mw fp32) + one int4 box record per slot
  return (int64_t)(align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256) + align_up((size_t)capacity_masks * sizeof(int4), 256));
}

int lmi_mask_logits_ex(const void *proto, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                       const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, const lmi_mask_image *images_host,
                       const lmi_mask_image *images_dev, int64_t capacity_masks, void *workspace, int64_t workspace_bytes, int flags,
                       void *stream) {
  LMI_REQUIRE(proto && det && slot_map && slot_base && workspace, LMI_ERR_BAD_ARG, "lmi_mask_logits: NULL pointer");
  MaskCall C;
  int rc = mask_check_common("lmi_mask_logits", batch, nm, mh, mw, h0, w0, images_host, images_dev, max_det, det_stride, capacity_masks, 1, C);
  if (rc) return rc;
  LMI_REQUIRE(lmi_is_aligned(workspace, 256) && lmi_is_aligned(proto, 4), LMI_ERR_ALIGNMENT, "lmi_mask_logits: workspace must be 256-byte aligned");
  LMI_REQUIRE(workspace_bytes >= lmi_mask_workspace_bytes(capacity_masks, mh, mw), LMI_ERR_WORKSPACE, "lmi_mask_logits: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)lmi_mask_workspace_bytes(capacity_masks, mh, mw));
  GemmParams G;
  G.proto = proto; G.det = det; G.slot_map = slot_map; G.slot_base = slot_base; G.logits = reinterpret_cast<float *>(workspace);
  G.batch = batch; G.max_det = max_det; G.det_stride = det_stride; G.capacity = capacity_masks; G.M = C.M; G.lstride = mh * mw;
  G.table = reinterpret_cast<int4 *>(reinterpret_cast<char *>(workspace) + align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256));
  const int npx = C.max_wh * mw;   // the largest window of the batch; tiles beyond an image's own window leave at once
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool half = (flags & LMI_MASK_PROTO_F16) != 0;
  if (!(flags & LMI_MASK_LOGITS_MFMA) && (mw % 2) == 0 && lmi_is_aligned(proto, half ? 4 : 8)) {
    const int tiles = ((mw + kLogitTileW - 1) / kLogitTileW) * ((C.max_wh + kLogitTileH - 1) / kLogitTileH);
    if (half) hipLaunchKernelGGL(mask_logits_rows_kernel<__half>, dim3((tiles + 3) / 4, batch), dim3(256), 0, s, G);
    else hipLaunchKernelGGL(mask_logits_rows_kernel<float>, dim3((tiles + 3) / 4, batch), dim3(256), 0, s, G);
    return lmi_check_launch("mask_logits_rows_kernel");
  }
  LMI_REQUIRE(!half, LMI_ERR_BAD_ARG, "lmi_mask_logits: fp16 prototypes are served by the row kernel only (even width, 4-byte aligned, "
              "no LMI_MASK_LOGITS_MFMA): convert them to fp32 for this shape");
  hipLaunchKernelGGL(mask_slot_table_kernel, dim3((unsigned)((capacity_masks + 255) / 256)), dim3(256), 0, s, G);
  // dense MFMA tiles: 4 adjacent pixels per lane when rows are whole 16-byte vectors, else one pixel per lane
  const bool vec4 = (mw % 4) == 0 && lmi_is_aligned(proto, 16);
  const int ntiles = (npx + (vec4 ? 127 : 31)) / (vec4 ? 128 : 32);
  G.tiles_per_block = vec4 ? 1 : 4;
  const int gx = (ntiles + 4 * G.tiles_per_block - 1) / (4 * G.tiles_per_block);
  dim3 ggrid(gx, (max_det + 31) / 32, batch);
  if (vec4) hipLaunchKernelGGL(mask_gemm_kernel<4>, ggrid, dim3(256), 0, s, G);
  else hipLaunchKernelGGL(mask_gemm_kernel<1>, ggrid, dim3(256), 0, s, G);
  return lmi_check_launch("mask_gemm_kernel");
}

// rows per strip: the largest power of two <= start whose staged logits rows fit 48 KiB of LDS
static int up_strip(const MaskCall &C, UpParams &U, int start, int extra_cols) {
  int rpb = start, lds_rows = 0;
  for (;; rpb >>= 1) {
    lds_rows = (int)((double)rpb * C.max_ratio) + 3;
    if (lds_rows > C.M.g.mh) lds_rows = C.M.g.mh;
    if ((size_t)lds_rows * (C.max_ww + extra_cols) * sizeof(float) <= 48 * 1024 || rpb == 1) break;
  }
  if ((size_t)lds_rows * (C.max_ww + extra_cols) * sizeof(float) > 64 * 1024) {
    lmi_set_error("lmi_mask_upsample: prototype rows too wide for LDS (%d)", C.max_ww);
    return LMI_ERR_LIMIT;
  }
  U.rows_per_block = rpb; U.lds_rows = lds_rows;
  return LMI_OK;
}

int lmi_mask_upsample_ex(void *workspace, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                         const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, const lmi_mask_image *images_host,
                         const lmi_mask_image *images_dev, void *out_masks, int out_dtype, int64_t capacity_masks, int64_t out_elems,
                         int32_t *overflow, int flags, void *stream) {
  const char *fn = "lmi_mask_upsample";
  LMI_REQUIRE(workspace && det && slot_map && slot_base && out_masks, LMI_ERR_BAD_ARG, "%s: NULL pointer", fn);
  LMI_REQUIRE(out_dtype == LMI_U8 || out_dtype == LMI_F32, LMI_ERR_BAD_ARG, "%s: out_dtype must be LMI_U8 or LMI_F32", fn);
  const int px = out_dtype == LMI_U8 ? 16 : 4;   // pixels per 16-byte chunk
  MaskCall C;
  int rc = mask_check_common(fn, batch, nm, mh, mw, h0, w0, images_host, images_dev, max_det, det_stride, capacity_masks, px, C);
  if (rc) return rc;
  LMI_REQUIRE(lmi_is_aligned(out_masks, 16) && lmi_is_aligned(workspace, 256), LMI_ERR_ALIGNMENT,
              "%s: out_masks must be 16-byte and workspace 256-byte aligned", fn);
  LMI_REQUIRE(images_host == nullptr || C.end_elems <= out_elems, LMI_ERR_WORKSPACE, "%s: the masks end at element %lld, out_masks holds %lld",
              fn, (long long)C.end_elems, (long long)out_elems);
  UpParams U;
  U.det = det; U.slot_map = slot_map; U.slot_base = slot_base; U.logits = reinterpret_cast<const float *>(workspace); U.out = out_masks;
  U.batch = batch; U.max_det = max_det; U.det_stride = det_stride; U.capacity = capacity_masks; U.out_elems = out_elems; U.M = C.M;
  U.lstride = mh * mw; U.overflow = overflow;
  U.table = reinterpret_cast<int4 *>(reinterpret_cast<char *>(workspace) + align_up((size_t)capacity_masks * mh * mw * sizeof(float), 256));
  U.rows_per_block = 0; U.lds_rows = 0; U.strips = 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  // Split path: zeros for every live slot at the memset rate -- one workgroup per
  // CU, the compact write window -- then mask_box_kernel overwrites only the chunks that meet a box (ALU-bound, not
  // store-bound: tall strips, so that a lane's horizontal taps serve many rows).
  if (C.separable && !(flags & LMI_MASK_UPSAMPLE_SINGLE)) {
    const long long esz = out_dtype == LMI_U8 ? 1 : 4;
    const long long ragged_bytes = images_host ? C.end_elems * esz : -1;
    hipLaunchKernelGGL(mask_zero_fill_kernel, dim3(device_cu_count()), dim3(256), 0, s, slot_base, batch, (long long)capacity_masks,
                       (long long)C.M.g.h0 * C.M.g.w0 * esz, ragged_bytes, reinterpret_cast<uint4 *>(out_masks), overflow);
    rc = lmi_check_launch("mask_zero_fill_kernel");
    if (rc) return rc;
    rc = up_strip(C, U, 256, 1);
    if (rc) return rc;
    const size_t lds = (size_t)U.lds_rows * (C.max_ww + 1) * sizeof(float);
    U.strips = (C.max_h0 + U.rows_per_block - 1) / U.rows_per_block;
    LMI_REQUIRE((long long)U.strips * capacity_masks <= 0x7fffffffLL, LMI_ERR_LIMIT, "%s: %lld slots x %d row strips exceed the grid", fn, (long long)capacity_masks, U.strips);
    dim3 grid((unsigned)((long long)U.strips * capacity_masks));
    if (out_dtype == LMI_U8) {
      if (C.chunked) hipLaunchKernelGGL((mask_box_kernel<uint8_t, true>), grid, dim3(256), lds, s, U);
      else hipLaunchKernelGGL((mask_box_kernel<uint8_t, false>), grid, dim3(256), lds, s, U);
    } else {
      if (C.chunked) hipLaunchKernelGGL((mask_box_kernel<float, true>), grid, dim3(256), lds, s, U);
      else hipLaunchKernelGGL((mask_box_kernel<float, false>), grid, dim3(256), lds, s, U);
    }
    return lmi_check_launch("mask_box_kernel");
  }
  rc = up_strip(C, U, 16, 0);
  if (rc) return rc;
  const size_t lds = (size_t)U.lds_rows * C.max_ww * sizeof(float);
  U.strips = (C.max_h0 + U.rows_per_block - 1) / U.rows_per_block;
  LMI_REQUIRE((long long)U.strips * capacity_masks <= 0x7fffffffLL, LMI_ERR_LIMIT, "%s: %lld slots x %d row strips exceed the grid", fn, (long long)capacity_masks, U.strips);
  dim3 ugrid((unsigned)((long long)U.strips * capacity_masks));
  if (out_dtype == LMI_U8) hipLaunchKernelGGL(mask_upsample_kernel<uint8_t>, ugrid, dim3(256), lds, s, U);
  else hipLaunchKernelGGL(mask_upsample_kernel<float>, ugrid, dim3(256), lds, s, U);
  return lmi_check_launch("mask_upsample_kernel");
}

int lmi_process_mask_ex(const void *proto, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                        const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, const lmi_mask_image *images_host,
                        const lmi_mask_image *images_dev, void *out_masks, int out_dtype, int64_t capacity_masks, int64_t out_elems,
                        int32_t *overflow, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
  int rc = lmi_mask_logits_ex(proto, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, images_host, images_dev,
                              capacity_masks, workspace, workspace_bytes, flags, stream);
  if (rc) return rc;
  return lmi_mask_upsample_ex(workspace, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, images_host, images_dev,
                              out_masks, out_dtype, capacity_masks, out_elems, overflow, flags, stream);
}

int lmi_mask_logits(const float *proto, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                    const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, int64_t capacity_masks,
                    void *workspace, int64_t workspace_bytes, void *stream) {
  return lmi_mask_logits_ex(proto, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, nullptr, nullptr,
                            capacity_masks, workspace, workspace_bytes, 0, stream);
}

int lmi_mask_upsample(void *workspace, int batch, int nm, int mh, int mw, const float *det, int max_det, int det_stride,
                      const int32_t *slot_map, const int32_t *slot_base, int h0, int w0, void *out_masks, int out_dtype,
                      int64_t capacity_masks, int32_t *overflow, void *stream) {
  return lmi_mask_upsample_ex(workspace, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, nullptr, nullptr,
                              out_masks, out_dtype, capacity_masks, 0, overflow, 0, stream);
}

int lmi_process_mask(const float *proto, int batch, int nm, int mh, int mw, const float *det, int max_det,
                     int det_stride, const int32_t *slot_map, const int32_t *slot_base, int h0, int w0,
                     void *out_masks, int out_dtype, int64_t capacity_masks, int32_t *overflow, void *workspace,
                     int64_t workspace_bytes, void *stream) {
  return lmi_process_mask_ex(proto, batch, nm, mh, mw, det, max_det, det_stride, slot_map, slot_base, h0, w0, nullptr, nullptr,
                             out_masks, out_dtype, capacity_masks, 0, overflow, workspace, workspace_bytes, 0, stream);
}

}  // extern "C"
