// Fused letterbox for gfx950: uint8 HWC frames -> cv2.INTER_LINEAR fixed-point resize ->
// centred constant pad -> optional BGR<->RGB -> /255 -> NCHW fp32|fp16, one pass over HBM.
// Also the plain `Yolov8.preprocess` transpose/normalise kernel.
//
// Replaces (reference paths): object_detectors/yolov8_lmi/run_model.py:79-87 (cv2 resize chain),
// lmi_utils/gadget_utils/pipeline_utils.py:145-173 (centred pad) and
// object_detectors/yolov8_lmi/model.py:112-134 (HWC->NCHW, cast, /255).
//
// Kernels
//  letterbox_tables_kernel  one lane per destination column / row: OpenCV's coefficient tables
//                           (fp64 coordinate -> fp32 fraction -> int16 weights) written once per
//                           launch into the caller's workspace; the main kernel reads them through
//                           L2 instead of redoing two fp64 divisions per tile.
//  letterbox_kernel         HBM-bound.  Per output tile (TW x 8 pixels) a workgroup stages the two
//                           source rows of every output row into LDS with 16-byte aligned,
//                           range-checked buffer loads (coalesced 128-bit), then every lane produces
//                           4 horizontally adjacent pixels x 3 channels in integer fixed point and
//                           writes one 128-bit store per channel plane.  One barrier per tile.
// Algorithmic bytes per image: unique source taps + 3*dst_h*dst_w*sizeof(out)  (SURVEY.md 8(d)).
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int kTH = 8;  // output rows per tile of the default shape

struct LbParams {
  const lmi_frame *frames;  // [device] ragged batch, or nullptr for the uniform form below
  const uint8_t *src;
  int h, w;
  long long row_stride, image_stride;
  int new_w, new_h, pad_l, pad_t;
  void *dst;
  int dst_h, dst_w;
  int pad_value;
  int swap_rb;
  int tiles_x, tiles_y;
  int pitch;        // LDS bytes per staged source row (multiple of 16)
  int2 *xtab;       // [sets][xw]     {source pixel or -1 (border column), a0 | a1 << 16}
  int4 *ytab;       // [sets][dst_h]  {source row 0, source row 1 (or -1,-1: border row), b0 | b1 << 16, 0}
  int xw;           // dst_w rounded up to 4
  int torch_one_thread;  // torch flavour: aten's kernel choice on exactly one thread (always the 4-weight form)
  int torch;             // tables and arithmetic of the torch flavour (aten bilinear on the fp32 cast, truncating cast)
};

// One axis of OpenCV's INTER_LINEAR table (resize.cpp, cv::hal::resize): see oracle/cv2_resize.py.
// fp64 scale and source coordinate, fp32 fraction, int16 weights rounded half-to-even.
__device__ __forceinline__ void cv_linear_coef(int src_n, int dst_n, int d, bool is_x, int &s, int &w0, int &w1) {
  double inv_scale = (double)dst_n / (double)src_n;
  double scale = 1.0 / inv_scale;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f = f - (float)si;
  if (is_x) {
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= src_n - 1) { f = 0.f; si = src_n - 1; }
  }
  w0 = (int)rintf((1.0f - f) * 2048.0f);
  w1 = (int)rintf(f * 2048.0f);
  s = si;
}

// aten upsample_bilinear2d (align_corners=False) source index and weights as torch's x86 CPU build computes them
// (oracle/geometry.py:resize_bilinear): the coordinate is one fused multiply-add, clamped at 0.
__device__ __forceinline__ void aten_linear_src(float scale, int d, int in_n, int &i0, int &i1, float &l0, float &l1) {
  float s = fmaxf(fmaf(scale, (float)d + 0.5f, -0.5f), 0.0f);
  i0 = min((int)s, in_n - 1);
  i1 = i0 + ((i0 < in_n - 1) ? 1 : 0);
  l1 = fminf(fmaxf(s - (float)i0, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}

__device__ __forceinline__ void frame_of(const LbParams &P, int b, const uint8_t *&src, int &h, int &w, long long &stride,
                                         int &new_w, int &new_h, int &pad_l, int &pad_t) {
  if (P.frames != nullptr) {
    const lmi_frame f = P.frames[b];
    src = reinterpret_cast<const uint8_t *>(f.src);
    h = f.h; w = f.w; stride = f.row_stride;
    new_w = f.new_w; new_h = f.new_h; pad_l = f.pad_l; pad_t = f.pad_t;
  } else {
    src = P.src + (long long)b * P.image_stride;
    h = P.h; w = P.w; stride = P.row_stride;
    new_w = P.new_w; new_h = P.new_h; pad_l = P.pad_l; pad_t = P.pad_t;
  }
}

// grid (ceil((xw + dst_h) / 256), table sets)
__global__ __launch_bounds__(256) void letterbox_tables_kernel(LbParams P) {
  const int set = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const uint8_t *src;
  int h, w, new_w, new_h, pad_l, pad_t;
  long long stride;
  frame_of(P, set, src, h, w, stride, new_w, new_h, pad_l, pad_t);
  if (i < P.xw) {
    const int dx = i - pad_l;
    int s = -1, a0 = 0, a1 = 0;
    if (dx >= 0 && dx < new_w && i < P.dst_w) {
      if (P.torch) {   // {left source pixel, fp32 bits of the right-hand weight}
        int i1; float l0, l1;
        aten_linear_src((float)w / (float)new_w, dx, w, s, i1, l0, l1);
        P.xtab[(size_t)set * P.xw + i] = make_int2(s, __float_as_int(l1));
        return;
      }
      cv_linear_coef(w, new_w, dx, true, s, a0, a1);
    }
    P.xtab[(size_t)set * P.xw + i] = make_int2(s, (int)((uint32_t)a0 | ((uint32_t)a1 << 16)));
  } else if (i < P.xw + P.dst_h) {
    const int y = i - P.xw;
    const int dy = y - pad_t;
    int r0 = -1, r1 = -1, b0 = 0, b1 = 0;
    if (dy >= 0 && dy < new_h && P.torch) {   // {upper row, lower row, fp32 bits of the lower row's weight}
      float l0, l1;
      aten_linear_src((float)h / (float)new_h, dy, h, r0, r1, l0, l1);
      P.ytab[(size_t)set * P.dst_h + y] = make_int4(r0, r1, __float_as_int(l1), 0);
      return;
    }
    if (dy >= 0 && dy < new_h) {
      int s;
      cv_linear_coef(h, new_h, dy, false, s, b0, b1);
      r0 = min(max(s, 0), h - 1);      // rows are clamped, their weights are kept (OpenCV's vertical pass)
      r1 = min(max(s + 1, 0), h - 1);
    }
    P.ytab[(size_t)set * P.dst_h + y] = make_int4(r0, r1, (int)((uint32_t)b0 | ((uint32_t)b1 << 16)), 0);
  }
}

// v / 255 with IEEE rounding for integer v in 0..255 WITHOUT a division: q = v * (1/255) is off by at most one
// ulp and one Newton step with exact fused residual repairs it (verified for all 256 inputs in the tests).
__device__ __forceinline__ float div255(int v) {
  const float fv = (float)v;
  const float r = 1.0f / 255.0f;  // constant-folded, correctly rounded
  const float q = fv * r;
  const float rem = fmaf(-q, 255.0f, fv);
  return fmaf(rem, r, q);
}

// Register-level view of one output element: fp32 as float, fp16 as its bit pattern in a
// 32-bit register (keeps the per-lane tile in VGPRs instead of a scratch array of __half).
template <typename OutT>
struct OutTraits;
template <>
struct OutTraits<float> {
  using reg = float;
  static __device__ __forceinline__ reg make(int v) { return div255(v); }
  static __device__ __forceinline__ void store1(float *p, reg v) { *p = v; }
  static __device__ __forceinline__ void store4nt(float *p, const reg v[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p));
  }
  static __device__ __forceinline__ void store4(float *p, const reg v[4], bool vec) {
    if (vec) {
      *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
    }
  }
};
template <>
struct OutTraits<__half> {
  using reg = uint32_t;
  // torch-CPU semantics: the fp32 quotient rounded to fp16
  static __device__ __forceinline__ reg make(int v) { return (uint32_t)__half_as_ushort(__float2half_rn(div255(v))); }
  static __device__ __forceinline__ void store1(__half *p, reg v) { *reinterpret_cast<uint16_t *>(p) = (uint16_t)v; }
  static __device__ __forceinline__ void store4nt(__half *p, const reg v[4]) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 t = {v[0] | (v[1] << 16), v[2] | (v[3] << 16)};
    __builtin_nontemporal_store(t, reinterpret_cast<u2 *>(p));
  }
  static __device__ __forceinline__ void store4(__half *p, const reg v[4], bool vec) {
    if (vec) {
      *reinterpret_cast<uint2 *>(p) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    } else {
      uint16_t *q = reinterpret_cast<uint16_t *>(p);
      q[0] = (uint16_t)v[0]; q[1] = (uint16_t)v[1]; q[2] = (uint16_t)v[2]; q[3] = (uint16_t)v[3];
    }
  }
};

// Workgroup = (TW/4) x TH lanes; lane (g, r) produces output pixels x = tile_x0 + 4g .. +3 of
// row tile_y0 + r.  LDS holds only the staged source rows.
// KF = 16-byte staging loads a lane keeps in flight: as many as a staged row has chunks per lane (4K -> 640: 10), no
// more -- a load past the row's span is dropped by the range check but still costs its issue slot.
// TORCH: the tables hold aten's source indices and fp32 weights, the arithmetic is aten's CPU bilinear on the fp32 cast
// with a truncating cast back (the reference helpers' resize_image); staging, tiles and stores are
// the same.
// CH: bytes per source pixel, 3 (HWC colour) or 1 (a grey plane: the callers replicate it to three channels BEFORE they
// resize -- cv2.cvtColor(GRAY2BGR), object_detectors/yolov8_lmi/run_model.py:77-78; `expand(-1,-1,3)`,
// model.py:125-126 -- and the resize treats channels independently, so resampling the plane once and writing the value
// to the three output planes is the same tensor for a third of the source bytes).
template <typename OutT, int TW, int TH, int KF, bool TORCH, int CH>
__global__ __launch_bounds__((TW / 4) * TH) void letterbox_kernel(LbParams P) {
  constexpr int NT = (TW / 4) * TH;
  extern __shared__ __attribute__((aligned(16))) unsigned char rows[];
  using TR = OutTraits<OutT>;
  using reg_t = typename TR::reg;

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int tx = blockIdx.x % P.tiles_x;
  const int ty = blockIdx.x / P.tiles_x;
  const int x_t0 = tx * TW, y_t0 = ty * TH;

  const uint8_t *src;
  int h, w, new_w, new_h, pad_l, pad_t;
  long long stride;
  frame_of(P, b, src, h, w, stride, new_w, new_h, pad_l, pad_t);
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);   // the frame's start within its 16-byte chunk
  const int set = P.frames != nullptr ? b : 0;
  const int2 *xt = P.xtab + (size_t)set * P.xw;
  const int4 *yt = P.ytab + (size_t)set * P.dst_h;

  OutT *dst = reinterpret_cast<OutT *>(P.dst) + (long long)b * 3 * P.dst_h * P.dst_w;
  const long long plane = (long long)P.dst_h * P.dst_w;
  const bool vec = ((P.dst_w & 3) == 0);

  const int g = tid % (TW / 4), r = tid / (TW / 4);
  const int x0 = x_t0 + 4 * g, y = y_t0 + r;
  // aten's choice between its two CPU kernels for a 3-channel frame: by output size on >= 2 threads, always the
  // 4-weight form on one
  const bool weights4 = TORCH && (P.torch_one_thread || new_h + new_w <= 128);
  const reg_t pv = TR::make(P.pad_value);

  // content rectangle clipped to this tile
  const int cx0 = max(x_t0, pad_l), cx1 = min(min(x_t0 + TW, P.dst_w), pad_l + new_w);
  const int cy0 = max(y_t0, pad_t), cy1 = min(min(y_t0 + TH, P.dst_h), pad_t + new_h);

  if (cx0 >= cx1 || cy0 >= cy1) {  // tile lies entirely in the border
    if (y < P.dst_h && x0 < P.dst_w) {
      const reg_t v[4] = {pv, pv, pv, pv};
      for (int c = 0; c < 3; ++c) {
        OutT *p = dst + c * plane + (long long)y * P.dst_w + x0;
        if (x0 + 3 < P.dst_w) { if (vec) TR::store4nt(p, v); else TR::store4(p, v, false); }
        else for (int k = 0; x0 + k < P.dst_w; ++k) TR::store1(p + k, pv);
      }
    }
    return;
  }

  // ---- table entries of this lane (issued first: their latency hides behind the staging) -------------
  const bool live = (y < P.dst_h) && (x0 < P.dst_w);
  int4 ye = make_int4(-1, -1, 0, 0);
  int4 xe01 = make_int4(-1, 0, -1, 0), xe23 = make_int4(-1, 0, -1, 0);
  if (live) {
    ye = yt[y];
    const int4 *xq = reinterpret_cast<const int4 *>(xt + x0);  // x0 % 4 == 0 and xw % 4 == 0: 32-byte aligned pairs
    xe01 = xq[0];
    xe23 = xq[1];
  }

  // ---- stage source rows -------------------------------------------------------------------------
  const int sx_min = xt[cx0].x;
  const int sx_max = min(xt[cx1 - 1].x + 1, w - 1);
  const int span_b0 = sx_min * CH;                             // first needed byte within a row
  const int span_bytes = (sx_max - sx_min + 1) * CH;
  const int nchunk = (span_bytes + 15 + 15) >> 4;              // covers any 0..15 byte alignment shift
  {
    // Range-checked loads are dropped whole when they cross num_records, so the range is the frame
    // rounded up to the 16-byte chunk size.  The buffer's base is the frame's address rounded DOWN to 16 bytes (a
    // frame may start at any byte: frame k of a packed batch of 375x500 images does; `mis` shifts every offset),
    // hence the rounded end is 16-byte aligned too: the first chunk stays inside the page that holds the frame's
    // first byte and the last one inside the page of its last byte -- no access can fault.
    const uint32_t frame_bytes = (uint32_t)(((long long)(h - 1) * stride + (long long)CH * w + mis + 15) & ~15LL);
    const uintptr_t abase = reinterpret_cast<uintptr_t>(src) - mis;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(abase & 0xffffffffu));
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    const void *ubase = reinterpret_cast<const void *>(((uintptr_t)hi << 32) | lo);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0,
                                                  __builtin_amdgcn_readfirstlane(frame_bytes), 0x00020000);
    // NT / (2*TH) lanes per staged row, each striding over that row's 16-byte chunks
    constexpr int LPS = NT / (2 * TH);
    const int slot = tid / LPS;
    const int sy = y_t0 + (slot >> 1);
    int row = -1;
    if (sy < P.dst_h) {
      const int4 e = yt[sy];
      row = (slot & 1) ? e.y : e.x;
      // a source row whose vertical weight is 0 contributes 0 to the fixed-point sum whatever it holds: it is not
      // fetched (the lanes then multiply stale LDS bytes by 0).  Exact integer down-scales such as 1080p -> 640x360
      // (fy = 3 dy + 1) need only one source row per output row this way: half the reads.
      const uint32_t bw = (uint32_t)e.z;
      if constexpr (TORCH) {   // lower row with weight +0.0: t1 * 0 adds nothing (the staged bytes are finite as floats)
        if ((slot & 1) && bw == 0u) row = -1;
      } else {
        if (((slot & 1) ? (bw >> 16) : (bw & 0xffffu)) == 0u) row = -1;
      }
    }
    if (row >= 0) {
      const uint32_t goff = (uint32_t)((long long)row * stride + span_b0) + mis;
      const uint32_t aligned = goff & ~15u;
      // KF loads in flight per lane before the first LDS write: the loop is latency-bound otherwise (4K frames need 10
      // chunks per lane: 169.5 us with 4 in flight, 159.3 us with 12).  Loads past the row's span get an out-of-range
      // offset (dropped by the range check, no traffic) and their LDS writes are skipped.
      constexpr uint32_t kOut = 0xfffffff0u;  // beyond num_records: the load is dropped without touching memory
      for (int ch = tid % LPS; ch < nchunk; ch += KF * LPS) {
        using chunk_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(rsrc, 0, 0, 0));
        chunk_t v[KF];
#pragma unroll
        for (int j = 0; j < KF; ++j)
          v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ch + j * LPS < nchunk ? aligned + 16u * (ch + j * LPS) : kOut, 0, 0);
        unsigned char *q = rows + slot * P.pitch + 16 * ch;
#pragma unroll
        for (int j = 0; j < KF; ++j)
          if (ch + j * LPS < nchunk) *reinterpret_cast<chunk_t *>(q + 16 * j * LPS) = v[j];
      }
    }
  }
  __syncthreads();

  // ---- resample ---------------------------------------------------------------------------------
  if (!live) return;
  reg_t out[3][4];
  if (ye.x < 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) out[c][k] = pv;
  } else {
    const uint32_t bw = (uint32_t)ye.z;
    const int b0 = (int)(bw & 0xffffu), b1 = (int)(bw >> 16);
    // byte offsets inside the staged rows (the slot's data starts `shift` bytes into its LDS row); every
    // LDS access indexes the one `rows` array so the compiler keeps them as ds_read, not flat loads
    const uint32_t *rw = reinterpret_cast<const uint32_t *>(rows);
    const int shift0 = (int)(((uint32_t)((long long)ye.x * stride + span_b0) + mis) & 15u);
    const int shift1 = (int)(((uint32_t)((long long)ye.y * stride + span_b0) + mis) & 15u);
    const int base0 = (2 * r) * P.pitch + shift0 - span_b0;
    const int base1 = (2 * r + 1) * P.pitch + shift1 - span_b0;
    const int xs_[4] = {xe01.x, xe01.z, xe23.x, xe23.z};
    const int xa_[4] = {xe01.y, xe01.w, xe23.y, xe23.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int s = xs_[k];
      if (s < 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c][k] = pv;
        continue;
      }
      const uint32_t aw = (uint32_t)xa_[k];
      const int a0 = (int)(aw & 0xffffu), a1 = (int)(aw >> 16);
      if (TORCH ? (aw | bw) == 0u : (a1 | b1) == 0) {   // torch flavour: both second weights +0.0 -> (int)(float)tap = tap
        // Both second weights are 0 (a0 = b0 = 2048): the fixed-point sum is the left upper tap itself,
        // ((2048 * ((s * 2048) >> 4)) >> 16 + 2) >> 2 = s.  Every pixel of an exact odd reduction (1080p -> 640x360:
        // fx = 3 dx + 1) or of a frame already at model size takes this branch: one tap, no arithmetic.
        const int p = base0 + s * CH;
        const int wi = p >> 2;
        const uint32_t t = __builtin_amdgcn_alignbyte(rw[wi + 1], rw[wi], (uint32_t)(p & 3));
        if constexpr (CH == 1) {
          out[0][k] = out[1][k] = out[2][k] = TR::make((int)(t & 0xffu));
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) out[c][k] = TR::make((int)((t >> (8 * c)) & 0xffu));
        }
        continue;
      }
      uint32_t px[2][2];  // [row][0: bytes p..p+3, 1: bytes p+4..p+7]
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int p = (rr == 0 ? base0 : base1) + s * CH;
        const int wi = p >> 2;
        const uint32_t sh = (uint32_t)(p & 3);
        if constexpr (CH == 1) {   // bytes p, p + 1 only
          px[rr][0] = __builtin_amdgcn_alignbyte(rw[wi + 1], rw[wi], sh);
          px[rr][1] = 0u;
        } else {
          const uint32_t d0 = rw[wi], d1 = rw[wi + 1], d2 = rw[wi + 2];
          px[rr][0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
          px[rr][1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        // pixel s: byte c ; pixel s+1: byte CH+c
        const int s00 = (int)((px[0][0] >> (8 * c)) & 0xffu);
        const int s10 = (int)((px[1][0] >> (8 * c)) & 0xffu);
        const int s01 = CH == 1 ? (int)((px[0][0] >> 8) & 0xffu) : (c == 0) ? (int)(px[0][0] >> 24) : (int)((px[0][1] >> (8 * (c - 1))) & 0xffu);
        const int s11 = CH == 1 ? (int)((px[1][0] >> 8) & 0xffu) : (c == 0) ? (int)(px[1][0] >> 24) : (int)((px[1][1] >> (8 * (c - 1))) & 0xffu);
        if constexpr (TORCH) {
          // aten: at the right border both horizontal taps are the last pixel; a lower row that was not staged has
          // weight 0 and whatever finite bytes the LDS holds
          const bool edge = s >= w - 1;
          const float lx1 = __uint_as_float(aw), lx0 = 1.0f - lx1, ly1 = __uint_as_float(bw), ly0 = 1.0f - ly1;
          const float i00 = (float)s00, i01 = (float)(edge ? s00 : s01), i10 = (float)s10, i11 = (float)(edge ? s10 : s11);
          float v;
          if (weights4) {
            const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
            v = fmaf(w11, i11, fmaf(w10, i10, fmaf(w00, i00, w01 * i01)));
          } else {
            const float t0 = fmaf(i00, lx0, i01 * lx1), t1 = fmaf(i10, lx0, i11 * lx1);
            v = fmaf(t0, ly0, t1 * ly1);
          }
          out[c][k] = TR::make((int)v);   // `.to(uint8)`: truncation (the value lies in [0, 255.00002])
        } else {
          // 24-bit multiplies (full rate; the 32-bit v_mul_lo_u32 the compiler picks for plain ints is quarter rate):
          // taps <= 255, weights <= 2048, h >> 4 <= 32640
          const uint32_t h0 = __umul24((uint32_t)s00, (uint32_t)a0) + __umul24((uint32_t)s01, (uint32_t)a1);
          const uint32_t h1 = __umul24((uint32_t)s10, (uint32_t)a0) + __umul24((uint32_t)s11, (uint32_t)a1);
          const int v = (int)(((__umul24((uint32_t)b0, h0 >> 4) >> 16) + (__umul24((uint32_t)b1, h1 >> 4) >> 16) + 2u) >> 2);
          out[c][k] = TR::make(v);
        }
      }
      if constexpr (CH == 1) out[1][k] = out[2][k] = out[0][k];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    OutT *p = dst + (P.swap_rb ? 2 - c : c) * plane + (long long)y * P.dst_w + x0;
    if (x0 + 3 < P.dst_w) {
      // streaming (non-temporal) stores: the tensor is written once and not re-read here; measured 162 -> 110 us
      // per 64 x 1080p launch against write-allocating stores
      if (vec) TR::store4nt(p, out[c]); else TR::store4(p, out[c], false);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (x0 + k < P.dst_w) TR::store1(p + k, out[c][k]);
    }
  }
}

// `Yolov8.preprocess`: HWC uint8 (C = 1 or 3) -> CHW /255.  One lane = 4 pixels of one row.
template <typename OutT>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ src, int h, int w, int c,
                                                        long long row_stride, OutT *__restrict__ dst) {
  const int groups = (w + 3) >> 2;
  const long long total = (long long)h * groups;
  const long long plane = (long long)h * w;
  const bool vec = ((w & 3) == 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(i / groups), x0 = (int)(i - (long long)y * groups) * 4;
    const uint8_t *p = src + (long long)y * row_stride + (long long)x0 * c;
    const int n = min(4, w - x0);
    using TR = OutTraits<OutT>;
    typename TR::reg o[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < n) {
        if (c == 3) {
          o[0][k] = TR::make(p[3 * k]);
          o[1][k] = TR::make(p[3 * k + 1]);
          o[2][k] = TR::make(p[3 * k + 2]);
        } else {
          o[0][k] = o[1][k] = o[2][k] = TR::make(p[k]);
        }
      } else {
        o[0][k] = o[1][k] = o[2][k] = TR::make(0);
      }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      OutT *q = dst + ch * plane + (long long)y * w + x0;
      if (n == 4) {
        TR::store4(q, o[ch], vec);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < n) TR::store1(q + k, o[ch][k]);
      }
    }
  }
}

// The LMI-flavoured caller-side resize: `resize_image` = F.interpolate(bilinear) on the
// fp32 cast of the frame, truncating cast back to uint8, then `fit_im_to_size` = centred constant pad, or centred
// crop where the content is larger than the destination (lmi_utils/gadget_utils/pipeline_utils.py:23-142, chained
// by ultralytics_lmi/yolo/run_model.py:87-100), then preprocess.  The arithmetic is aten's CPU kernel as built for
// x86 (oracle/geometry.py:resize_bilinear, pinned by tests/golden/letterbox_torch.npz): source coordinate
// fma(scale, d + 0.5, -0.5) clamped at 0; then the per-pixel 4-weight kernel aten runs when out_h + out_w <= 128,
// else its separable kernel (horizontal pairs, then vertical, each fma(a, wa, b*wb)).
// One lane = 4 output pixels x 3 channels; taps are read straight from L2/HBM (12 bytes per tap pair).
template <typename OutT, int CH>
__global__ __launch_bounds__(256) void letterbox_torch_kernel(LbParams P) {
  using TR = OutTraits<OutT>;
  using reg_t = typename TR::reg;
  const int b = blockIdx.z;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  if (y >= P.dst_h || x0 >= P.dst_w) return;
  const uint8_t *src;
  int h, w, new_w, new_h, pad_l, pad_t;
  long long stride;
  frame_of(P, b, src, h, w, stride, new_w, new_h, pad_l, pad_t);
  // aten's choice between its two CPU kernels for a 3-channel frame: by output size on >= 2 threads, always the
  // 4-weight form on one
  const bool weights4 = P.torch_one_thread || new_h + new_w <= 128;
  const reg_t pv = TR::make(P.pad_value);
  reg_t out[3][4];
  const int dy = y - pad_t;
  const bool row_in = dy >= 0 && dy < new_h;
  int iy0 = 0, iy1 = 0;
  float ly0 = 0.f, ly1 = 0.f;
  if (row_in) aten_linear_src((float)h / (float)new_h, dy, h, iy0, iy1, ly0, ly1);
  const float scale_w = (float)w / (float)new_w;
  const uint8_t *r0 = src + (long long)iy0 * stride, *r1 = src + (long long)iy1 * stride;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int dx = x0 + k - pad_l;
    if (!row_in || dx < 0 || dx >= new_w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[c][k] = pv;
      continue;
    }
    int ix0, ix1;
    float lx0, lx1;
    aten_linear_src(scale_w, dx, w, ix0, ix1, lx0, lx1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float i00 = (float)r0[ix0 * CH + c], i01 = (float)r0[ix1 * CH + c];
      const float i10 = (float)r1[ix0 * CH + c], i11 = (float)r1[ix1 * CH + c];
      float v;
      if (weights4) {
        const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
        v = fmaf(w11, i11, fmaf(w10, i10, fmaf(w00, i00, w01 * i01)));
      } else {
        const float t0 = fmaf(i00, lx0, i01 * lx1), t1 = fmaf(i10, lx0, i11 * lx1);
        v = fmaf(t0, ly0, t1 * ly1);
      }
      out[c][k] = TR::make((int)v);   // `.to(uint8)`: truncation (the value lies in [0, 255.00002])
    }
    if constexpr (CH == 1) out[1][k] = out[2][k] = out[0][k];
  }
  const long long plane = (long long)P.dst_h * P.dst_w;
  OutT *dst = reinterpret_cast<OutT *>(P.dst) + (long long)b * 3 * plane;
  const bool vec = (P.dst_w & 3) == 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    OutT *p = dst + (P.swap_rb ? 2 - c : c) * plane + (long long)y * P.dst_w + x0;
    if (x0 + 3 < P.dst_w) {
      if (vec) TR::store4nt(p, out[c]); else TR::store4(p, out[c], false);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (x0 + k < P.dst_w) TR::store1(p + k, out[c][k]);
    }
  }
}

template <typename OutT>
int launch_letterbox_torch(const LbParams &P, int batch, int ch, hipStream_t stream) {
  dim3 grid((P.dst_w + 255) / 256, (P.dst_h + 3) / 4, batch);
  if (ch == 1) hipLaunchKernelGGL((letterbox_torch_kernel<OutT, 1>), grid, dim3(256), 0, stream, P);
  else hipLaunchKernelGGL((letterbox_torch_kernel<OutT, 3>), grid, dim3(256), 0, stream, P);
  return lmi_check_launch("letterbox_torch_kernel");
}

int lds_pitch_for(int w, int new_w, int tw, int ch) {
  // widest source span (pixels) a tile of `tw` output columns can touch, +1 neighbour, +alignment slack
  double scale = (double)w / (double)new_w;
  long span_px = (long)(scale * tw) + 3;
  if (span_px > w) span_px = w;
  long bytes = span_px * ch + 15 + 15 + 12;  // alignment shift + chunk rounding + 3-dword tap window
  return (int)((bytes + 15) & ~15L);
}

size_t lds_bytes_for(int pitch, int th = kTH) { return (size_t)2 * th * pitch; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct TabLayout {
  size_t off_y, total;
  int xw;
};

TabLayout tab_layout(int sets, int dst_h, int dst_w) {
  TabLayout L;
  L.xw = (dst_w + 3) & ~3;
  L.off_y = align_up(sizeof(int2) * (size_t)sets * L.xw, 256);
  L.total = align_up(L.off_y + sizeof(int4) * (size_t)sets * dst_h, 256);
  return L;
}

template <typename OutT, bool TORCH, int CH>
int launch_letterbox_ch(const LbParams &P0, int batch, int sets, int max_pitch128, int max_pitch32, void *workspace,
                        int64_t workspace_bytes, hipStream_t stream, int plan) {
  LbParams P = P0;
  const TabLayout L = tab_layout(sets, P.dst_h, P.dst_w);
  LMI_REQUIRE(workspace != nullptr && lmi_is_aligned(workspace, 256), LMI_ERR_ALIGNMENT, "lmi_letterbox_u8: workspace must be 256-byte aligned");
  LMI_REQUIRE(workspace_bytes >= (int64_t)L.total, LMI_ERR_WORKSPACE, "lmi_letterbox_u8: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)L.total);
  P.xw = L.xw;
  P.torch = TORCH ? 1 : 0;
  P.xtab = reinterpret_cast<int2 *>(workspace);
  P.ytab = reinterpret_cast<int4 *>(reinterpret_cast<unsigned char *>(workspace) + L.off_y);
  // plan / execute (r03): the tables depend on the sizes alone -- a caller that letterboxes one geometry again and again
  // builds them once (LMI_LB_TABLES_ONLY) and leaves the table pass out of every call after that (LMI_LB_TABLES_READY)
  if (plan != LMI_LB_TABLES_READY) {
    hipLaunchKernelGGL(letterbox_tables_kernel, dim3((L.xw + P.dst_h + 255) / 256, sets), dim3(256), 0, stream, P);
    const int rc = lmi_check_launch("letterbox_tables_kernel");
    if (rc || plan == LMI_LB_TABLES_ONLY) return rc;
  }

  const size_t kMaxLds = 64 * 1024;
  if (lds_bytes_for(max_pitch128) <= kMaxLds) {
    P.pitch = max_pitch128;
    P.tiles_x = (P.dst_w + 127) / 128;
    P.tiles_y = (P.dst_h + kTH - 1) / kTH;
    dim3 grid(P.tiles_x * P.tiles_y, batch);
    // chunks per lane of a staged row: pitch / 16 bytes over the 16 lanes that share the row
    if (P.pitch > 8 * 256) hipLaunchKernelGGL((letterbox_kernel<OutT, 128, kTH, 12, TORCH, CH>), grid, dim3(32 * kTH), lds_bytes_for(P.pitch), stream, P);
    else if (P.pitch > 4 * 256) hipLaunchKernelGGL((letterbox_kernel<OutT, 128, kTH, 8, TORCH, CH>), grid, dim3(32 * kTH), lds_bytes_for(P.pitch), stream, P);
    else hipLaunchKernelGGL((letterbox_kernel<OutT, 128, kTH, 4, TORCH, CH>), grid, dim3(32 * kTH), lds_bytes_for(P.pitch), stream, P);
  } else if (lds_bytes_for(max_pitch32) <= 2 * kMaxLds) {
    P.pitch = max_pitch32;
    P.tiles_x = (P.dst_w + 31) / 32;
    P.tiles_y = (P.dst_h + kTH - 1) / kTH;
    dim3 grid(P.tiles_x * P.tiles_y, batch);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(letterbox_kernel<OutT, 32, kTH, 4, TORCH, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)kMaxLds);
    hipLaunchKernelGGL((letterbox_kernel<OutT, 32, kTH, 4, TORCH, CH>), grid, dim3(8 * kTH), lds_bytes_for(P.pitch), stream, P);
  } else {
    lmi_set_error("lmi_letterbox_u8: down-scale factor too large for the LDS row tiles");
    return LMI_ERR_LIMIT;
  }
  return lmi_check_launch("letterbox_kernel");
}

template <typename OutT, bool TORCH>
int launch_letterbox(const LbParams &P, int batch, int sets, int ch, int max_pitch128, int max_pitch32, void *workspace,
                     int64_t workspace_bytes, hipStream_t stream, int plan) {
  return ch == 1 ? launch_letterbox_ch<OutT, TORCH, 1>(P, batch, sets, max_pitch128, max_pitch32, workspace, workspace_bytes, stream, plan)
                 : launch_letterbox_ch<OutT, TORCH, 3>(P, batch, sets, max_pitch128, max_pitch32, workspace, workspace_bytes, stream, plan);
}

int check_common(const void *dst, int dst_h, int dst_w, int dst_dtype, int pad_value, int &mode, int batch, int &ch, int &plan) {
  ch = (mode & LMI_SRC_GRAY) ? 1 : 3;   // the resampler is the low byte of `mode`
  plan = mode & (LMI_LB_TABLES_ONLY | LMI_LB_TABLES_READY);
  mode &= ~(LMI_SRC_GRAY | LMI_LB_TABLES_ONLY | LMI_LB_TABLES_READY);
  LMI_REQUIRE(plan != (LMI_LB_TABLES_ONLY | LMI_LB_TABLES_READY), LMI_ERR_BAD_ARG, "lmi_letterbox_u8: LMI_LB_TABLES_ONLY and LMI_LB_TABLES_READY exclude each other");
  LMI_REQUIRE(dst != nullptr || plan == LMI_LB_TABLES_ONLY, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: dst is NULL");
  LMI_REQUIRE(batch > 0 && batch <= 65535, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: batch %d out of range 1..65535", batch);
  LMI_REQUIRE(dst_h > 0 && dst_w > 0, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: bad destination size %dx%d", dst_h, dst_w);
  LMI_REQUIRE(dst_dtype == LMI_F32 || dst_dtype == LMI_F16, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: dst_dtype must be LMI_F32 or LMI_F16");
  LMI_REQUIRE(pad_value >= 0 && pad_value <= 255, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: pad_value %d not in 0..255", pad_value);
  LMI_REQUIRE(mode == LMI_RESIZE_CV2_LINEAR || mode == LMI_RESIZE_TORCH_BILINEAR || mode == LMI_RESIZE_TORCH_BILINEAR_1T, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: unknown resize mode %d", mode);
  LMI_REQUIRE(lmi_is_aligned(dst, 16), LMI_ERR_ALIGNMENT, "lmi_letterbox_u8: dst must be 16-byte aligned");   // (NULL passes)
  return LMI_OK;
}

int check_frame(const void *src, int h, int w, long long stride, int new_w, int new_h, int pad_l, int pad_t, int dst_h, int dst_w, int i,
                int ch, bool may_crop = false, bool tables_only = false) {
  LMI_REQUIRE(src != nullptr || tables_only, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: frame %d: src is NULL", i);
  LMI_REQUIRE(h > 0 && w > 0 && stride >= (long long)ch * w, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: frame %d: bad size/stride (%d,%d,%lld)", i, h, w, stride);
  LMI_REQUIRE((long long)(h - 1) * stride + (long long)ch * w < (1LL << 32) - 32, LMI_ERR_LIMIT, "lmi_letterbox_u8: frame %d larger than 4 GiB", i);
  // the torch flavour may place the content partly outside the destination (fit_im_to_size crops, centred)
  LMI_REQUIRE(new_w > 0 && new_h > 0 && (may_crop || (pad_l >= 0 && pad_t >= 0 && pad_l + new_w <= dst_w && pad_t + new_h <= dst_h)),
              LMI_ERR_BAD_ARG, "lmi_letterbox_u8: frame %d: content %dx%d at (%d,%d) does not fit %dx%d", i, new_w, new_h, pad_l, pad_t, dst_w, dst_h);
  return LMI_OK;
}

}  // namespace

extern "C" {

int64_t lmi_letterbox_workspace_bytes(int table_sets, int dst_h, int dst_w) {
  if (table_sets <= 0 || dst_h <= 0 || dst_w <= 0) return 0;
  return (int64_t)tab_layout(table_sets, dst_h, dst_w).total;
}

int lmi_letterbox_u8(const lmi_frame *frames_host, const lmi_frame *frames_dev, int batch, void *dst, int dst_h,
                     int dst_w, int dst_dtype, int pad_value, int swap_rb, int mode, void *workspace,
                     int64_t workspace_bytes, void *stream) {
  int ch = 3, plan = 0;
  int rc = check_common(dst, dst_h, dst_w, dst_dtype, pad_value, mode, batch, ch, plan);
  if (rc) return rc;
  LMI_REQUIRE(frames_host != nullptr && frames_dev != nullptr, LMI_ERR_BAD_ARG, "lmi_letterbox_u8: frames is NULL");
  int p128 = 0, p32 = 0;
  for (int i = 0; i < batch; ++i) {
    const lmi_frame &f = frames_host[i];
    rc = check_frame(reinterpret_cast<const void *>(f.src), f.h, f.w, f.row_stride, f.new_w, f.new_h, f.pad_l, f.pad_t, dst_h, dst_w, i,
                     ch, mode != LMI_RESIZE_CV2_LINEAR, plan == LMI_LB_TABLES_ONLY);
    if (rc) return rc;
    int a = lds_pitch_for(f.w, f.new_w, 128, ch), b = lds_pitch_for(f.w, f.new_w, 32, ch);
    if (a > p128) p128 = a;
    if (b > p32) p32 = b;
  }
  LbParams P{};
  P.frames = frames_dev;
  P.dst = dst; P.dst_h = dst_h; P.dst_w = dst_w; P.pad_value = pad_value; P.swap_rb = swap_rb ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  P.torch_one_thread = mode == LMI_RESIZE_TORCH_BILINEAR_1T ? 1 : 0;
  if (mode != LMI_RESIZE_CV2_LINEAR) {
    // with a workspace for the tables: the LDS-staged kernel (249 -> ~100 us per 64 x 1080p); without one, the
    // table-free kernel that reads its taps straight from global memory (the entry's contract before r02)
    if (workspace == nullptr || workspace_bytes < lmi_letterbox_workspace_bytes(batch, dst_h, dst_w)) {
      LMI_REQUIRE(plan == 0, LMI_ERR_WORKSPACE, "lmi_letterbox_u8: LMI_LB_TABLES_ONLY / _READY need a workspace for the tables");
      return dst_dtype == LMI_F32 ? launch_letterbox_torch<float>(P, batch, ch, s) : launch_letterbox_torch<__half>(P, batch, ch, s);
    }
    return dst_dtype == LMI_F32 ? launch_letterbox<float, true>(P, batch, batch, ch, p128, p32, workspace, workspace_bytes, s, plan)
                                : launch_letterbox<__half, true>(P, batch, batch, ch, p128, p32, workspace, workspace_bytes, s, plan);
  }
  return dst_dtype == LMI_F32 ? launch_letterbox<float, false>(P, batch, batch, ch, p128, p32, workspace, workspace_bytes, s, plan)
                              : launch_letterbox<__half, false>(P, batch, batch, ch, p128, p32, workspace, workspace_bytes, s, plan);
}

int lmi_letterbox_u8_uniform(const void *src, int batch, int h, int w, int64_t row_stride, int64_t image_stride,
                             void *dst, int dst_h, int dst_w, int dst_dtype, int pad_value, int swap_rb, int stretch,
                             int mode, void *workspace, int64_t workspace_bytes, void *stream) {
  int ch = 3, plan = 0;
  int rc = check_common(dst, dst_h, dst_w, dst_dtype, pad_value, mode, batch, ch, plan);
  if (rc) return rc;
  // this form derives the geometry itself, by the cv2 / ultralytics rules (round(), centred pad); the reference
  // helpers behind the torch flavours size and pad differently (int() truncation, pipeline_utils.py:36-39,110-131),
  // so those go through lmi_letterbox_u8 with the geometry in the frame descriptors
  LMI_REQUIRE(mode == LMI_RESIZE_CV2_LINEAR, LMI_ERR_BAD_ARG,
              "lmi_letterbox_u8_uniform: mode %d needs the helpers' own geometry: use lmi_letterbox_u8 with frame descriptors", mode);
  int32_t g[6];
  rc = lmi_letterbox_geometry(h, w, dst_h, dst_w, stretch, g);
  if (rc) return rc;
  rc = check_frame(src, h, w, row_stride, g[0], g[1], g[2], g[4], dst_h, dst_w, 0, ch, false, plan == LMI_LB_TABLES_ONLY);
  if (rc) return rc;
  LMI_REQUIRE(batch == 1 || image_stride >= (int64_t)(h - 1) * row_stride + (int64_t)ch * w, LMI_ERR_BAD_ARG,
              "lmi_letterbox_u8_uniform: image_stride %lld must cover a frame", (long long)image_stride);
  LbParams P{};
  P.frames = nullptr;
  P.src = reinterpret_cast<const uint8_t *>(src);
  P.h = h; P.w = w; P.row_stride = row_stride; P.image_stride = image_stride;
  P.new_w = g[0]; P.new_h = g[1]; P.pad_l = g[2]; P.pad_t = g[4];
  P.dst = dst; P.dst_h = dst_h; P.dst_w = dst_w; P.pad_value = pad_value; P.swap_rb = swap_rb ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int p128 = lds_pitch_for(w, g[0], 128, ch), p32 = lds_pitch_for(w, g[0], 32, ch);
  return dst_dtype == LMI_F32 ? launch_letterbox<float, false>(P, batch, 1, ch, p128, p32, workspace, workspace_bytes, s, plan)
                              : launch_letterbox<__half, false>(P, batch, 1, ch, p128, p32, workspace, workspace_bytes, s, plan);
}

int lmi_preprocess_u8(const void *src, int h, int w, int c, int64_t row_stride, void *dst, int dst_dtype, void *stream) {
  LMI_REQUIRE(src != nullptr && dst != nullptr, LMI_ERR_BAD_ARG, "lmi_preprocess_u8: NULL pointer");
  LMI_REQUIRE(h > 0 && w > 0 && (c == 1 || c == 3), LMI_ERR_BAD_ARG, "lmi_preprocess_u8: bad shape (%d,%d,%d); C must be 1 or 3", h, w, c);
  LMI_REQUIRE(row_stride >= (int64_t)w * c, LMI_ERR_BAD_ARG, "lmi_preprocess_u8: row_stride %lld < %d", (long long)row_stride, w * c);
  LMI_REQUIRE(dst_dtype == LMI_F32 || dst_dtype == LMI_F16, LMI_ERR_BAD_ARG, "lmi_preprocess_u8: dst_dtype must be LMI_F32 or LMI_F16");
  LMI_REQUIRE(lmi_is_aligned(dst, 16), LMI_ERR_ALIGNMENT, "lmi_preprocess_u8: dst must be 16-byte aligned");
  const long long total = (long long)h * ((w + 3) / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dst_dtype == LMI_F32)
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(src), h, w, c, (long long)row_stride, reinterpret_cast<float *>(dst));
  else
    hipLaunchKernelGGL(preprocess_kernel<__half>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(src), h, w, c, (long long)row_stride, reinterpret_cast<__half *>(dst));
  return lmi_check_launch("preprocess_kernel");
}

}  // extern "C"
