// `segments` of the result dict: `ops.masks2segments(masks)`
// (object_detectors/yolov8_lmi/model.py:238-243; in-tree form yolov5_lmi/trt/yolov5_trt.py:418-431), which the
// reference computes on the CPU with cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) after copying every
// full-resolution mask to the host.
//
// Here the masks stay on the GPU: pack_mask_boxes_kernel turns each mask's box region (a mask is zero outside its
// box by construction) into one bit per pixel with wave ballots -- ~1/50 of the bytes of the uint8 masks for the
// device->host copy -- and the border following itself (sequential by nature) runs in C++ on the host cores, all
// masks of the batch in parallel threads.  Semantics: oracle/segments.py (Suzuki-Abe outer borders of the
// hole-filled, 8-connected components; start = raster-first pixel; collinear runs collapsed).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct PackParams {
  const void *masks;        // (m, H, W) uint8 or fp32
  const int4 *boxes;        // per mask: x0, y0, x1, y1 (integer, half-open, inside the image)
  const long long *offset;  // per mask: first 64-bit word of its packed rows
  unsigned long long *out;
  int H, W;
};

// grid: x = row groups of 4 (one wave per row), y = mask
template <typename T>
__global__ __launch_bounds__(256) void pack_mask_boxes_kernel(PackParams P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.y;
  const int4 b = P.boxes[i];
  const int rows = b.w - b.y, cols = b.z - b.x;
  const int r = blockIdx.x * 4 + wave;
  if (r >= rows || cols <= 0) return;
  const int row_words = (cols + 63) >> 6;
  const T *src = reinterpret_cast<const T *>(P.masks) + ((size_t)i * P.H + (b.y + r)) * P.W + b.x;
  unsigned long long *dst = P.out + P.offset[i] + (size_t)r * row_words;
  for (int w = 0; w < row_words; ++w) {
    const int c = w * 64 + lane;
    const bool on = c < cols && src[c] != (T)0;
    const unsigned long long bits = __ballot(on);
    if (lane == 0) dst[w] = bits;
  }
}

// ---- host: border following on one packed crop ---------------------------------------------------
// 8-neighbourhood clockwise from west, (dy, dx) with y down
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
const int kDx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

inline int dir_index(int dy, int dx) {
  for (int k = 0; k < 8; ++k)
    if (kDy[k] == dy && kDx[k] == dx) return k;
  return 0;
}

struct Contours {
  std::vector<int32_t> xy;      // x, y pairs of every kept point, contours back to back
  std::vector<int32_t> length;  // points per contour
};

void trace_crop(const unsigned long long *bits, int rows, int cols, int row_words, Contours &out) {
  const int ph = rows + 2, pw = cols + 2;
  // f: 0 background reachable from outside, 1 foreground or enclosed hole (filled), 2 labelled
  std::vector<uint8_t> fg((size_t)ph * pw, 0), filled((size_t)ph * pw, 1);
  bool any = false;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      if ((bits[(size_t)r * row_words + (c >> 6)] >> (c & 63)) & 1ull) { fg[(size_t)(r + 1) * pw + c + 1] = 1; any = true; }
  if (!any) return;
  // fill holes: background 4-connected to the outside is 0, everything else 1 (scipy binary_fill_holes)
  std::vector<int> stack;
  stack.reserve(1024);
  filled[0] = 0;
  stack.push_back(0);
  while (!stack.empty()) {
    const int p = stack.back();
    stack.pop_back();
    const int y = p / pw, x = p - y * pw;
    const int ny[4] = {y - 1, y + 1, y, y}, nx[4] = {x, x, x - 1, x + 1};
    for (int k = 0; k < 4; ++k) {
      if (ny[k] < 0 || ny[k] >= ph || nx[k] < 0 || nx[k] >= pw) continue;
      const int q = ny[k] * pw + nx[k];
      if (filled[q] && !fg[q]) { filled[q] = 0; stack.push_back(q); }
    }
  }
  // components of `filled` (8-connected) in raster order of their first pixel; trace each outer border
  std::vector<uint8_t> seen((size_t)ph * pw, 0);
  for (int i = 1; i <= rows; ++i) {
    for (int j = 1; j <= cols; ++j) {
      const int p0 = i * pw + j;
      if (!filled[p0] || seen[p0]) continue;
      seen[p0] = 1;
      stack.push_back(p0);
      while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int y = p / pw, x = p - y * pw;
        for (int k = 0; k < 8; ++k) {
          const int q = (y + kDy[k]) * pw + (x + kDx[k]);   // the zero pad ring keeps this in range
          if (filled[q] && !seen[q]) { seen[q] = 1; stack.push_back(q); }
        }
      }
      // Suzuki-Abe step 3.1: clockwise from the west neighbour, first nonzero neighbour
      int fi = -1, fj = -1;
      for (int k = 0; k < 8; ++k)
        if (filled[(i + kDy[k]) * pw + (j + kDx[k])]) { fi = i + kDy[k]; fj = j + kDx[k]; break; }
      const size_t first_pt = out.xy.size();
      if (fi < 0) {  // isolated pixel
        out.xy.push_back(j - 1); out.xy.push_back(i - 1);
        out.length.push_back(1);
        continue;
      }
      std::vector<int32_t> pts;   // x, y, then the direction index leaving the point
      int i2 = fi, j2 = fj, i3 = i, j3 = j;
      for (;;) {
        const int k0 = dir_index(i2 - i3, j2 - j3);
        int i4 = i3, j4 = j3, dk = 0;
        for (int s = 1; s <= 8; ++s) {  // counter-clockwise around (i3,j3), starting after (i2,j2)
          const int k = (k0 - s + 16) & 7;
          if (filled[(i3 + kDy[k]) * pw + (j3 + kDx[k])]) { i4 = i3 + kDy[k]; j4 = j3 + kDx[k]; dk = k; break; }
        }
        pts.push_back(j3 - 1); pts.push_back(i3 - 1); pts.push_back(dk);
        if (i4 == i && j4 == j && i3 == fi && j3 == fj) break;
        i2 = i3; j2 = j3; i3 = i4; j3 = j4;
      }
      const int n = (int)(pts.size() / 3);
      if (n <= 2) {
        for (int k = 0; k < n; ++k) { out.xy.push_back(pts[3 * k]); out.xy.push_back(pts[3 * k + 1]); }
      } else {  // CHAIN_APPROX_SIMPLE: keep a point when the direction leaving it differs from the one entering it
        for (int k = 0; k < n; ++k) {
          const int prev = (k + n - 1) % n;
          if (pts[3 * k + 2] != pts[3 * prev + 2]) { out.xy.push_back(pts[3 * k]); out.xy.push_back(pts[3 * k + 1]); }
        }
        if (out.xy.size() == first_pt) { out.xy.push_back(pts[0]); out.xy.push_back(pts[1]); }
      }
      out.length.push_back((int32_t)((out.xy.size() - first_pt) / 2));
    }
  }
}

}  // namespace

extern "C" {

int lmi_pack_mask_boxes(const void *masks, int dtype, int64_t m, int img_h, int img_w, const int32_t *boxes,
                        const int64_t *word_offset, uint64_t *out_words, void *stream) {
  LMI_REQUIRE(dtype == LMI_U8 || dtype == LMI_F32, LMI_ERR_BAD_ARG, "lmi_pack_mask_boxes: dtype must be LMI_U8 or LMI_F32");
  LMI_REQUIRE(m >= 0 && m <= 65535, LMI_ERR_LIMIT, "lmi_pack_mask_boxes: %lld masks not in 0..65535", (long long)m);
  LMI_REQUIRE(img_h > 0 && img_w > 0, LMI_ERR_BAD_ARG, "lmi_pack_mask_boxes: bad image size");
  if (m == 0) return LMI_OK;
  LMI_REQUIRE(masks && boxes && word_offset && out_words, LMI_ERR_BAD_ARG, "lmi_pack_mask_boxes: NULL pointer");
  PackParams P;
  P.masks = masks; P.boxes = reinterpret_cast<const int4 *>(boxes); P.offset = reinterpret_cast<const long long *>(word_offset);
  P.out = reinterpret_cast<unsigned long long *>(out_words); P.H = img_h; P.W = img_w;
  dim3 grid((img_h + 3) / 4, (unsigned)m);  // rows beyond a box's height exit at once
  LMI_REQUIRE(grid.x <= 0x7fffffffu, LMI_ERR_LIMIT, "lmi_pack_mask_boxes: image too tall");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == LMI_U8) hipLaunchKernelGGL(pack_mask_boxes_kernel<uint8_t>, grid, dim3(256), 0, s, P);
  else hipLaunchKernelGGL(pack_mask_boxes_kernel<float>, grid, dim3(256), 0, s, P);
  return lmi_check_launch("pack_mask_boxes_kernel");
}

int64_t lmi_trace_contours_host(const uint64_t *words, int64_t m, const int32_t *boxes, const int64_t *word_offset,
                                int strategy_all, int threads, int32_t *out_xy, int64_t capacity_points,
                                int64_t *points_per_mask) {
  if (m < 0 || (m > 0 && (!words || !boxes || !word_offset || !points_per_mask))) {
    lmi_set_error("lmi_trace_contours_host: bad argument");
    return LMI_ERR_BAD_ARG;
  }
  std::vector<std::vector<int32_t>> result((size_t)m);
  std::atomic<int64_t> next(0);
  auto work = [&]() {
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= m) break;
      const int32_t *b = boxes + 4 * i;
      const int rows = b[3] - b[1], cols = b[2] - b[0];
      if (rows <= 0 || cols <= 0) continue;
      Contours c;
      trace_crop(reinterpret_cast<const unsigned long long *>(words) + word_offset[i], rows, cols, (cols + 63) >> 6, c);
      if (c.length.empty()) continue;
      size_t begin = 0, end = c.xy.size() / 2;
      if (!strategy_all) {  // the longest contour, first one on ties
        size_t best = 0, pos = 0, best_pos = 0;
        for (size_t k = 0; k < c.length.size(); ++k) {
          if (c.length[k] > c.length[best]) { best = k; best_pos = pos; }
          pos += c.length[k];
        }
        begin = best_pos; end = best_pos + c.length[best];
      }
      std::vector<int32_t> &r = result[(size_t)i];
      r.reserve(2 * (end - begin));
      for (size_t k = begin; k < end; ++k) { r.push_back(c.xy[2 * k] + b[0]); r.push_back(c.xy[2 * k + 1] + b[1]); }
    }
  };
  // a worker thread costs ~20 us to start and join, a mask ~40 us to trace: one thread per 8 masks, `threads` at most
  int nt = threads > 0 ? threads : 1;
  if ((int64_t)nt > (m + 7) / 8) nt = (int)((m + 7) / 8 > 0 ? (m + 7) / 8 : 1);
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  int64_t total = 0;
  for (int64_t i = 0; i < m; ++i) { points_per_mask[i] = (int64_t)(result[(size_t)i].size() / 2); total += points_per_mask[i]; }
  if (total <= capacity_points && out_xy) {
    int64_t pos = 0;
    for (int64_t i = 0; i < m; ++i) {
      if (!result[(size_t)i].empty()) memcpy(out_xy + 2 * pos, result[(size_t)i].data(), result[(size_t)i].size() * sizeof(int32_t));
      pos += points_per_mask[i];
    }
  }
  return total;  // > capacity_points: nothing was written, call again with a larger buffer
}

}  // extern "C"
